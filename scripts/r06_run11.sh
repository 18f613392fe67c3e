#!/bin/bash
# round 6, eleventh GPU call: bench.py with the counter record of the M2 leg's kernel (short legs), launch shapes of the M2 leg and of the M1 leg,
# counter profile of the precedence kernel (50 x 20 nine-leaf) on the library that ships
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r11; mkdir -p $O; export TMPDIR=/tmp
export SF_COMMIT=$(cat $R/build/commit.txt 2>/dev/null)  # (.git does not travel: the container writes HEAD there before the call)
timeout 600 python bench.py --steps 20 --warmup 5 --solve-seconds 20 --tuned-seconds 0 --c5-seconds 3 2>$O/bench.err | tail -1 > $O/bench_short.json
python - <<'P' | tee $O/bench_short_summary.txt
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_r11/bench_short.json').read())
print('value', round(d['value']/1e9,2), 'roof', d['roofline']['bound'], round(d['roofline']['frac'] or 0,3))
s=d['extra']['side_configs']['cvrp1000_default_list_policy']
r=s['roofline']; print('m2', round(s['moves_per_s_rank0']/1e9,2), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ('counters_per_launch','kernel_resources','window')}); print(r.get('window'))
P
for cfg in "6144 30000" "6144 100000" "9216 30000" "12288 30000" "12288 100000"; do
  set -- $cfg
  timeout 400 python scripts/m2_probe.py $1 default 250 6 $2 2>&1 | tail -1 | cut -c1-330 | tee -a $O/m2_shapes.jsonl
done
B="python bench.py --no-pmc --solve-seconds 0 --steps 12 --warmup 4 --no-cpu-baseline"
for rep in 24576 36864 49152; do
  timeout 300 $B --replicas $rep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('replicas $rep', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/m1_shapes.txt
done
timeout 900 python scripts/pmc_run.py k_mixed_search_wave 1 $O/prec_pmc.json -- python $R/scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-600
