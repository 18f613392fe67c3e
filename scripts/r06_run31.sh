#!/bin/bash
# round 6: a second (entity, value) cost matrix on another score level (uni programs on two levels): new tests, C2 A/B against the library before (build/libsf_g5.so),
# the six-leaf rate, then the whole GPU suite
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r31; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_uni_program.py tests/test_gpu_assignment.py -x -q -m gpu 2>&1 | tail -6 | tee $O/uni_tests.txt
for lib in build/libsf_g5.so solverforge_amd/libsolverforge_amd.so build/libsf_g5.so solverforge_amd/libsolverforge_amd.so; do
  for pol in la sa; do
    echo "$lib graph $pol: $(SF_AMD_LIB=$R/$lib timeout 300 python scripts/graph_bench.py 3072 60 6 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/ab.txt
  done
done
timeout 400 python scripts/m2_probe.py 24576 default6 60 4 100000 2>&1 | tail -1 | cut -c1-260 | tee $O/six_leaf.txt
SF_FUZZ_MODEL=assignment timeout 150 python scripts/fuzz_parity.py 60 69000 > $O/fuzz_assignment.json 2>> $O/fuzz.err; tail -c 200 $O/fuzz_assignment.json; echo
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
