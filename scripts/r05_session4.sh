#!/bin/bash
# round 5, session 4: where the seven-leaf step goes -- RUIN instantiation with an empty ruin leaf vs the six-leaf kernel, eager-recreate variants
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s4; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
probe() { name=$1; lib=$2; shift 2; SF_AMD_LIB=$R/build/$lib timeout 600 python scripts/phase_probe_generic.py "$@" > $O/$name.txt 2>&1; echo "== $name"; tail -4 $O/$name.txt | cut -c1-260; }
probe base_mps0   libsf_phase.so 2048 $L7 0 0
probe base_mps1   libsf_phase.so 2048 $L7 0 1
probe six_2048    libsf_phase.so 2048 $L6 0
probe eager_mps10 libsf_eager.so 2048 $L7 0 10
probe eager_mps0  libsf_eager.so 2048 $L7 0 0
probe inl_mps10   libsf_eager_inl.so 2048 $L7 0 10
probe inl_mps0    libsf_eager_inl.so 2048 $L7 0 0
timeout 900 python -m pytest tests/test_gpu_ruin.py tests/test_gpu_union.py tests/test_gpu_kopt.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
timeout 300 python scripts/solve60.py 20 2048 $L7 30000 savings_capacity 2>/dev/null | tail -2 | cut -c1-600
