#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s8; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
for v in v2_prod szA szAB szAC szABC; do echo "$v $(SF_AMD_LIB=$R/build/libsf_$v.so timeout 300 python scripts/generic_step_time.py 2048 $L7 2>&1 | tail -1)" | tee -a $O/variants.txt; done
SF_AMD_LIB=$R/build/libsf_szABC.so timeout 900 python -m pytest tests/test_gpu_ruin.py tests/test_gpu_union.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
