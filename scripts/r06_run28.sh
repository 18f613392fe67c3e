#!/bin/bash
# round 6, twenty-eighth GPU call: uni filters / weights as programs compiled into the value-cost matrix (new tests), the value-cost / assignment family around it
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r28; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_uni_program.py -x -q -m gpu 2>&1 | tail -15 | tee $O/uni_tests.txt
timeout 600 python -m pytest tests/test_gpu_assignment.py tests/test_gpu_scalar.py tests/test_gpu_validation.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/tests.txt
