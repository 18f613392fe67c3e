#!/bin/bash
# round 6, twenty-fourth GPU call: the interpreter with 2 x 4 partner ids / values in flight per pass (the program still runs on four at a time) vs 1 x 4
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r24; mkdir -p $O; export TMPDIR=/tmp
SF_AMD_LIB=$R/build/libsf_irg2.so timeout 600 python -m pytest tests/test_gpu_pair_ir.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/tests.txt
for lib in build/libsf_irg1.so build/libsf_irg2.so build/libsf_irg1.so build/libsf_irg2.so; do
  for pol in la sa; do
    echo "$lib interpret=1 graph $pol: $(SF_AMD_LIB=$R/$lib SF_AMD_IR_INTERPRET=1 timeout 300 python scripts/graph_bench.py 3072 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/ab.txt
  done
done
