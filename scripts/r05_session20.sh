#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s20; mkdir -p $O
for v in prod preceager; do lib=$R/build/libsf_$v.so; [ $v = prod ] && lib=$R/solverforge_amd/libsolverforge_amd.so
  for shop in "20 10 2048 10 4" "50 20 2048 4 3"; do echo "$v $shop $(SF_AMD_LIB=$lib timeout 600 python scripts/precedence_bench.py $shop policy9 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('gpu_moves_per_s','kernel_ms_per_launch','replica0_matches_oracle','replica0_matches_oracle_first_steps')})")" | tee -a $O/prec.txt; done; done
SF_AMD_LIB=$R/build/libsf_preceager.so timeout 1200 python -m pytest tests/test_gpu_precedence_leaf.py tests/test_gpu_precedence.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
