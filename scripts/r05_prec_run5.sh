#!/bin/bash
# precedence parity + rates after a change of the wave-wide evaluation (sf_precedence.h); optional stage probe (build/libsf_peval.so)
cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_precedence.py tests/test_gpu_precedence_leaf.py tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
for cfg in "50 20 2048" "100 20 1024" "20 10 2048"; do set -- $cfg; echo "four-leaf $cfg: $(timeout 300 python scripts/precedence_bench.py $1 $2 $3 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,2),'M', d['replica0_matches_oracle'], d['kernel_ms_per_launch'])")"; done
echo "nine-leaf 50 20: $(timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-300)"
echo "nine-leaf 20 10: $(timeout 300 python scripts/prec_policy_launches.py 20 10 2048 10 3 2>&1 | tail -1 | cut -c1-300)"
for r in 256 512; do echo "C4 + makespan $r: $(SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/c4_makespan_rate.py $r 5 2 2>&1 | grep -E "launch|^\{" | cut -c1-250)"; done
[ -f build/libsf_peval.so ] && for c in "50 20 2048" "100 20 1024"; do SF_AMD_LIB=build/libsf_peval.so timeout 300 python scripts/peval_probe.py $c 5 2>&1 | tail -1; done
