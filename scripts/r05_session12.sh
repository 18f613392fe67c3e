#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s12; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
for ru in 2,5,0 2,5,1 2,5,5 2,5,10 2,5,16 1,1,10 2,2,10 3,3,10 5,5,10; do echo "$ru $(SF_AMD_LIB=$R/build/libsf_prod3.so timeout 300 python scripts/generic_step_time.py 2048 $L7 300 $ru 2>&1 | tail -1)" | tee -a $O/sweep.txt; done
