#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s14; mkdir -p $O
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
for v in prod six_ng3 six_ng4; do lib=$R/build/libsf_$v.so; [ $v = prod ] && lib=$R/solverforge_amd/libsolverforge_amd.so
  for rep in 3072 4096 12288; do echo "$v 6-leaf $rep $(SF_AMD_LIB=$lib SF_AMD_DEBUG_LAUNCH=1 timeout 300 python scripts/generic_step_time.py $rep $L6 300 2>&1 | tail -2 | tr '\n' ' ' | sed 's/.*resident\/CU=/resident=/')" | tee -a $O/times.txt; done
  SF_AMD_LIB=$lib timeout 300 python scripts/solve60.py 20 12288 $L6 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('$v 6-leaf 20s 12288:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s', g['ls_steps_per_replica'])" | tee -a $O/times.txt
done
SF_AMD_LIB=$R/build/libsf_six_ng4.so timeout 900 python -m pytest tests/test_gpu_union.py tests/test_gpu_kopt.py tests/test_gpu_budget.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
