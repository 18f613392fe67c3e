#!/bin/bash
# A/B of the wave-wide precedence evaluation (sf_precedence.h, prec_eval): parity first, then rates per library / setting
#   default = this tree; build/libsf_nochain.so = the library before the change (round-5 final check)
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_precedence.py tests/test_gpu_precedence_leaf.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/precedence_bench.py 50 20 2048 2 1 2>&1 | grep -m1 "generic engine launch"
SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/precedence_bench.py 100 20 1024 2 1 2>&1 | grep -m1 "generic engine launch"
run() {  # label, env assignments...
  local label=$1; shift
  echo "== $label"
  for cfg in "50 20 2048" "100 20 1024" "20 10 2048"; do
    set -- $cfg
    echo "four-leaf $cfg: $(env $ENVS timeout 300 python scripts/precedence_bench.py $1 $2 $3 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,2),'M', d['replica0_matches_oracle'], d['kernel_ms_per_launch'])")"
  done
  echo "nine-leaf 50 20: $(env $ENVS timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-300)"
  echo "nine-leaf 20 10: $(env $ENVS timeout 300 python scripts/prec_policy_launches.py 20 10 2048 10 3 2>&1 | tail -1 | cut -c1-300)"
}
ENVS="X=1" run "default"
[ -n "$AB_ALL" ] && ENVS="SF_AMD_PREC_STATIC_SLIM=0" run "no slim static copy"
[ -n "$AB_ALL" ] && [ -f build/libsf_nochain.so ] && ENVS="SF_AMD_LIB=build/libsf_nochain.so" run "library before the change"
[ -f build/libsf_peval.so ] && for c in "50 20 2048" "100 20 1024"; do SF_AMD_LIB=build/libsf_peval.so timeout 300 python scripts/peval_probe.py $c 5 2>&1 | tail -1; done
