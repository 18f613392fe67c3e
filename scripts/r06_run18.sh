#!/bin/bash
# round 6, eighteenth GPU call: scalar units built twice (without the interpreted joins / with them at W = 8): parity, A/B against the single build (build/libsf_g3.so, W = 4)
# and against the interpreter unit at W = 4 (build/libsf_irw4.so)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r18; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pair_ir.py tests/test_gpu_scalar.py tests/test_gpu_anneal.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $O/tests.txt
for cfg in "build/libsf_g3.so 0" "solverforge_amd/libsolverforge_amd.so 0" "build/libsf_g3.so 1" "solverforge_amd/libsolverforge_amd.so 1" "build/libsf_irw4.so 1"; do
  set -- $cfg
  for pol in la sa; do
    echo "$1 interpret=$2 graph $pol: $(SF_AMD_LIB=$R/$1 SF_AMD_IR_INTERPRET=$2 timeout 300 python scripts/graph_bench.py 3072 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/pair_ir_ab.txt
  done
done
