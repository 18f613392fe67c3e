#!/bin/bash
# round 6: the scalar trial code compiled out of the FAST (list-only) generic kernels: six- / seven-leaf rates against the library before (build/libsf_g5.so), parity, fuzz
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r32; mkdir -p $O; export TMPDIR=/tmp
for lib in build/libsf_g5.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 400 python scripts/m2_probe.py 24576 default6 60 4 100000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib six |" | tee -a $O/ab.txt
  SF_AMD_LIB=$R/$lib timeout 400 python scripts/m2_probe.py 24576 default 40 4 100000 2>&1 | tail -1 | cut -c1-260 | sed "s|^|$lib seven |" | tee -a $O/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_union.py tests/test_gpu_kopt.py tests/test_gpu_ruin.py tests/test_gpu_mixed.py tests/test_gpu_cvrp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/tests.txt
SF_FUZZ_MODEL=cvrp timeout 150 python scripts/fuzz_parity.py 60 70000 > $O/fuzz_cvrp.json 2>> $O/fuzz.err; tail -c 160 $O/fuzz_cvrp.json; echo
SF_FUZZ_MODEL=assignment timeout 150 python scripts/fuzz_parity.py 60 71000 > $O/fuzz_assignment.json 2>> $O/fuzz.err; tail -c 160 $O/fuzz_assignment.json; echo
