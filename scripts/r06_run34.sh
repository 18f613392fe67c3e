#!/bin/bash
# round 6: the specialised scalar kernel built for four workgroups per CU (128 registers, 16 replicas per CU) vs three, now that it no longer spills
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r34; mkdir -p $O; export TMPDIR=/tmp
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_sc4.so; do
  for rep in 3072 4096 12288; do
    for pol in la sa; do
      echo "$lib $rep graph $pol: $(SF_AMD_LIB=$R/$lib timeout 300 python scripts/graph_bench.py $rep 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/ab.txt
    done
  done
done
