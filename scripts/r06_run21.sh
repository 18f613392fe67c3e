#!/bin/bash
# round 6, twenty-first GPU call: the six-leaf FAST kernel built for five workgroups per CU (96 registers, 20 replicas per CU if the slice allows) vs four
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r21; mkdir -p $O; export TMPDIR=/tmp
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_b5.so; do
  SF_AMD_LIB=$R/$lib SF_AMD_DEBUG_LAUNCH=1 timeout 400 python scripts/m2_probe.py 24576 default6 40 4 100000 2>$O/err_$(basename $lib).txt | tail -1 | cut -c1-260 | sed "s|^|$lib |" | tee -a $O/b5.txt
  grep "\[sf\]" $O/err_$(basename $lib).txt | head -2 | tee -a $O/b5.txt
done
SF_AMD_LIB=$R/build/libsf_b5.so timeout 600 python -m pytest tests/test_gpu_union.py tests/test_gpu_kopt.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/b5.txt
