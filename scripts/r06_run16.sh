#!/bin/bash
# round 6, sixteenth GPU call: the interpreted pair-predicate join with W = 4 partners side by side (pair_program_holds2_w): parity, A/B against the
# library before it (build/libsf_g2.so) on the specialised and the interpreted path; the launch facts of C4
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r16; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pair_ir.py tests/test_gpu_scalar.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $O/tests.txt
for lib in build/libsf_g2.so solverforge_amd/libsolverforge_amd.so; do
  for ip in 0 1; do
    for pol in la sa; do
      echo "$lib interpret=$ip graph $pol: $(SF_AMD_LIB=$R/$lib SF_AMD_IR_INTERPRET=$ip timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/pair_ir_ab.txt
    done
  done
done
SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/jobshop_bench.py 1024 2>&1 | grep "\[sf\]" | head -3 | tee $O/c4_launch.txt
