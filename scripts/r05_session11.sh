#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s11; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
SF_AMD_LIB=$R/build/libsf_rv2chk.so timeout 900 python scripts/ruin_v2_check.py 60 > $O/rv2_check.jsonl 2> $O/rv2_check.err; tail -1 $O/rv2_check.jsonl; tail -3 $O/rv2_check.err
echo "prod3 7-leaf $(SF_AMD_LIB=$R/build/libsf_prod3.so timeout 300 python scripts/generic_step_time.py 2048 $L7 2>&1 | tail -1)" | tee -a $O/times.txt
SF_AMD_LIB=$R/build/libsf_prod3.so timeout 900 python -m pytest tests/test_gpu_ruin.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
SF_AMD_LIB=$R/build/libsf_prod3.so timeout 300 python scripts/solve60.py 20 2048 $L7 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('7-leaf 20s 2048:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s', g['ls_steps_per_replica'])" | tee -a $O/times.txt
SF_AMD_LIB=$R/build/libsf_prod3.so timeout 600 python scripts/deep_parity_ruin.py 120 2>&1 | tail -3 | cut -c1-200
