#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s24; mkdir -p $O
for rep in 512 1280 2560; do echo "C4+makespan $(timeout 900 python scripts/c4_makespan_rate.py $rep 5 2 2>&1 | tail -1 | cut -c1-400)" | tee -a $O/c4mk.txt; done
