#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s7; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
probe() { name=$1; lib=$2; shift 2; SF_AMD_LIB=$R/build/$lib timeout 600 python scripts/phase_probe_generic.py "$@" > $O/$name.txt 2>&1; echo "== $name"; tail -3 $O/$name.txt | cut -c1-260; }
probe v2c_mps10 libsf_v2ph.so 2048 $L7 0 10
probe v2c_late  libsf_v2ph.so 2048 $L7 1500 10
