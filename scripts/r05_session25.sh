#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s25; mkdir -p $O
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
for v in prod six5; do lib=$R/build/libsf_$v.so; [ $v = prod ] && lib=$R/solverforge_amd/libsolverforge_amd.so
  for rep in 12288 20480; do SF_AMD_LIB=$lib timeout 300 python scripts/solve60.py 15 $rep $L6 30000 savings_capacity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gpu']; print('$v 6-leaf 15s $rep:', g['best_score'], round(g['moves_per_s']/1e9,3), 'G moves/s')" | tee -a $O/six.txt; done; done
