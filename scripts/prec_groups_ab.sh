#!/bin/bash
# A/B of the grouped precedence trial evaluator: moves/s and best score of scripts/precedence_solve60.py per SF_AMD_PREC_GROUPS value.
# usage: prec_groups_ab.sh seconds jobs machines replicas "groups..." [policy]
S=$1; J=$2; M=$3; R=$4; GS=$5; POL=$6
for g in $GS; do
  if [ -n "$POL" ]; then ARGS="feasible policy"; else ARGS=""; fi
  SF_AMD_PREC_GROUPS=$g timeout 300 python scripts/precedence_solve60.py $S $J $M $R $ARGS 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('J%sxM%s R%s groups %s %s: %.1f M moves/s best %s steps/replica %s | cpu %s' % ($J,$M,$R,'$g','$POL',d['gpu']['moves_per_s']/1e6,d['gpu']['best_score'],d['gpu']['ls_steps_per_replica'],d.get('cpu',{}).get('best_score')))"
done
