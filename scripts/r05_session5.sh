#!/bin/bash
# round 5, session 5: the list-preserving ruin trial (sf_ruin_v2.h): differential check against sf_ruin.h, parity vs the oracle, phase probe, sustained rate
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s5; mkdir -p $O
SF_AMD_LIB=$R/build/libsf_rv2chk.so timeout 900 python scripts/ruin_v2_check.py 60 > $O/rv2_check.jsonl 2> $O/rv2_check.err; cat $O/rv2_check.jsonl; tail -3 $O/rv2_check.err
timeout 900 python -m pytest tests/test_gpu_ruin.py tests/test_gpu_union.py tests/test_gpu_kopt.py tests/test_gpu_cvrp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
probe() { name=$1; lib=$2; shift 2; SF_AMD_LIB=$R/build/$lib timeout 600 python scripts/phase_probe_generic.py "$@" > $O/$name.txt 2>&1; echo "== $name"; tail -4 $O/$name.txt | cut -c1-260; }
probe v2_mps10 libsf_v2ph.so 2048 $L7 0 10
probe v2_late  libsf_v2ph.so 2048 $L7 1500 10
timeout 300 python scripts/solve60.py 20 2048 $L7 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('7-leaf 20s 2048:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s', g['ls_steps_per_replica'])"
timeout 300 python scripts/solve60.py 20 4096 $L7 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('7-leaf 20s 4096:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s', g['ls_steps_per_replica'])"
