#!/bin/bash
# quick check after a change of prec_eval: the precedence parity file, two rates, the stage probe (build/libsf_peval.so)
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_precedence.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3
for cfg in "50 20 2048" "100 20 1024"; do set -- $cfg; echo "four-leaf $cfg: $(timeout 300 python scripts/precedence_bench.py $1 $2 $3 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,2),'M', d['replica0_matches_oracle'], d['kernel_ms_per_launch'])")"; done
echo "nine-leaf 50 20: $(timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-300)"
[ -f build/libsf_peval.so ] && for c in "50 20 2048"; do SF_AMD_LIB=build/libsf_peval.so timeout 300 python scripts/peval_probe.py $c 5 2>&1 | tail -1; done
