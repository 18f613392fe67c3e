#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s13; mkdir -p $O
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
timeout 900 python -m pytest tests/test_gpu_ruin.py tests/test_gpu_union.py tests/test_gpu_kopt.py tests/test_gpu_budget.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
timeout 600 python scripts/deep_parity_ruin.py 120 2>&1 | tail -3 | cut -c1-220 | tee $O/deep.txt
for rep in 2048 3072 6144; do echo "7-leaf $rep $(SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/generic_step_time.py $rep $L7 300 2>&1 | tail -2 | tr '\n' ' ')" | tee -a $O/times.txt; done
for rep in 3072 6144; do timeout 300 python scripts/solve60.py 20 $rep $L7 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('7-leaf 20s $rep:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s', g['ls_steps_per_replica'])" | tee -a $O/times.txt; done
