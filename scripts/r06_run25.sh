#!/bin/bash
# round 6, twenty-fifth GPU call: do the M2 leg's counter passes survive launches of 24,576 replicas of k_mixed_search_wave?
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r25; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tuned-seconds 0 --c5-seconds 0 --solve-seconds 30 --solve-replicas 24576 2>$O/err.txt | tail -1 > $O/bench.json
python - <<'P' | tee $O/summary.txt
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_r25/bench.json').read())
s=d['extra']['side_configs']['cvrp1000_default_list_policy']; r=s['roofline']
print(round(s['moves_per_s_rank0']/1e9,3), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('bound','frac','salu_frac','valu_frac','hbm_frac','replay_matches_leg','pmc_source','avg_launch_ms')})
print(r.get('kernel_resources',{}).get('failed_passes'))
P
