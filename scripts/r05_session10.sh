#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s10; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
probe() { name=$1; lib=$2; shift 2; SF_AMD_LIB=$R/build/$lib timeout 600 python scripts/phase_probe_generic.py "$@" > $O/$name.txt 2>&1; echo "== $name"; tail -3 $O/$name.txt | cut -c1-260; }
probe p7_early libsf_phase.so 2048 $L7 0 10
probe p7_mps0  libsf_phase.so 2048 $L7 0 0
probe p7_late  libsf_phase.so 2048 $L7 1500 10
