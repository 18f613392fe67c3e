#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s26; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -25 | tee $O/mixed.txt
timeout 600 python -m pytest tests/test_default_policy.py tests/test_gpu_pair_ir.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/other.txt
