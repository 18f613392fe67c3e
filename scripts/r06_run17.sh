#!/bin/bash
# round 6, seventeenth GPU call: slots in flight per lane in the ruin trial's first scan (SF_RV2_U = 2 / 4 / 8), seven-leaf policy, short-step and long-step regime
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r17; mkdir -p $O; export TMPDIR=/tmp
for lib in build/libsf_rv2u2.so solverforge_amd/libsolverforge_amd.so build/libsf_rv2u8.so; do
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py 6144 default 8 8 30000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib early |" | tee -a $O/rv2_u.txt
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py 12288 default 40 4 100000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib late |" | tee -a $O/rv2_u.txt
done
