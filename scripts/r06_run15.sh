#!/bin/bash
# round 6, fifteenth GPU call: the join of the two planning classes under the host-driven entry points (new test), whole GPU suite, fuzz (job shop family)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r15; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu -k "two_class" 2>&1 | tail -15 | tee $O/join_tests.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
SF_FUZZ_MODEL=jobshop timeout 240 python scripts/fuzz_parity.py 120 64000 > $O/fuzz_jobshop.json 2> $O/fuzz.err; tail -c 200 $O/fuzz_jobshop.json; echo
