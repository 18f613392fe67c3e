#!/bin/bash
# round 6, nineteenth GPU call: the ruin trial's re-pricing with every element's gathers issued together: ruin parity + fuzz, A/B against the library before (build/libsf_g4.so)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r19; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ruin.py tests/test_gpu_union.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $O/tests.txt
SF_FUZZ_MODEL=cvrp timeout 200 python scripts/fuzz_parity.py 90 65000 > $O/fuzz_cvrp.json 2> $O/fuzz.err; tail -c 200 $O/fuzz_cvrp.json; echo
for lib in build/libsf_g4.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py 6144 default 8 8 30000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib early |" | tee -a $O/ruin_ab.txt
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py 12288 default 40 4 100000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib late |" | tee -a $O/ruin_ab.txt
done
