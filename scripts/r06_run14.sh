#!/bin/bash
# round 6, fourteenth GPU call: C5 / C4 launch shapes, then the default bench.py line with the new launch shapes (M1 98,304 replicas, M2 12,288 x 100,000) and
# its rocprofv3 kernel-trace summary
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r14; mkdir -p $O; export TMPDIR=/tmp
C5="python bench.py --customers 5000 --vehicles 500 --ls-steps 100 --steps 5 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc"
for rep in 2816 5632 11264; do
  timeout 300 $C5 --replicas $rep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 replicas $rep', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/shapes.txt
done
for rep in 1024 2048 4096; do
  echo "c4 replicas $rep: $(timeout 300 python scripts/jobshop_bench.py $rep 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e9,3),'G', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/shapes.txt
done
for rep in 3072 6144 12288; do
  echo "c2 la replicas $rep: $(timeout 300 python scripts/graph_bench.py $rep 100 10 la 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e9,3),'G', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/shapes.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'P' | tee $O/bench_summary.txt
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_r14/bench.json').read())
print('value', round(d['value']/1e9,2), 'ms_per_step', d['ms_per_step'], 'roof', d['roofline']['bound'], d['roofline']['frac'], 'match', d['extra'].get('replica0_matches_cpu_oracle'), 'cpu', d['cpu_baseline'].get('value'))
b=d['extra']['best_score_at_60s']; print('m2', b['gpu'], b['cpu_oracle'].get('best_score') if b.get('cpu_oracle') else None, b['gpu_moves_per_s_rank0'], b['seconds_to_cpu_best'])
s=d['extra']['side_configs']; r=s['cvrp1000_default_list_policy']['roofline']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ('counters_per_launch','kernel_resources','window','pmc_source','kernel')}); print(s.get('cvrp5000_nearby2'))
P
