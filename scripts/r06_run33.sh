#!/bin/bash
# round 6, last validation of the library at HEAD: smoke, whole GPU suite, the default bench line
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r33; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'P' | tee $O/bench_summary.txt
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_r33/bench.json').read())
print('value', round(d['value']/1e9,2), 'ms_per_step', round(d['ms_per_step'],2), 'roof', d['roofline']['bound'], round(d['roofline']['frac'] or 0,3), 'match', d['extra'].get('replica0_matches_cpu_oracle'), 'cpu', d['cpu_baseline'].get('value'))
b=d['extra']['best_score_at_60s']; print('m2', b['gpu'], b['cpu_oracle'].get('best_score') if b.get('cpu_oracle') else None, round(b['gpu_moves_per_s_rank0']/1e9,2), b['seconds_to_cpu_best'])
s=d['extra']['side_configs']; r=s['cvrp1000_default_list_policy']['roofline']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('bound','frac','salu_frac','valu_frac','hbm_frac','replay_matches_leg')}); print(round(s['cvrp5000_nearby2']['moves_per_s_rank0']/1e9,2))
P
