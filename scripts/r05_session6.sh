#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s6; mkdir -p $O
SF_AMD_LIB=$R/build/libsf_rv2chk.so timeout 900 python scripts/ruin_v2_check.py 60 > $O/rv2_check.jsonl 2> $O/rv2_check.err; tail -2 $O/rv2_check.jsonl; tail -3 $O/rv2_check.err
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
probe() { name=$1; lib=$2; shift 2; SF_AMD_LIB=$R/build/$lib timeout 600 python scripts/phase_probe_generic.py "$@" > $O/$name.txt 2>&1; echo "== $name"; tail -2 $O/$name.txt | cut -c1-260; }
probe v2b_mps10 libsf_v2ph.so 2048 $L7 0 10
probe v2b_mps0  libsf_v2ph.so 2048 $L7 0 0
probe v2b_late  libsf_v2ph.so 2048 $L7 1500 10
