#!/bin/bash
# round 6, twenty-second GPU call: partner ids in flight per pass of the specialised join (8 / 16 / 24 / 32) and the interpreter with the next batch's ids prefetched
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r22; mkdir -p $O; export TMPDIR=/tmp
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_cw8.so build/libsf_cw24.so build/libsf_cw32.so solverforge_amd/libsolverforge_amd.so; do
  for pol in la sa; do
    echo "$lib interpret=0 graph $pol: $(SF_AMD_LIB=$R/$lib SF_AMD_IR_INTERPRET=0 timeout 300 python scripts/graph_bench.py 3072 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/ab.txt
  done
done
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_irpf.so; do
  for pol in la sa; do
    echo "$lib interpret=1 graph $pol: $(SF_AMD_LIB=$R/$lib SF_AMD_IR_INTERPRET=1 timeout 300 python scripts/graph_bench.py 3072 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/ab.txt
  done
done
