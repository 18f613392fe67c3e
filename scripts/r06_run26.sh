#!/bin/bash
# round 6, final validation of the library at HEAD: smoke, whole GPU suite, differential fuzz (three families), the default bench line, its rocprofv3 kernel-trace
# summary, counter profiles of C2 / C4 on this library
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r26; mkdir -p $O; export TMPDIR=/tmp
export SF_COMMIT=$(cat $R/build/commit.txt 2>/dev/null)
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
for fam in cvrp jobshop precedence; do SF_FUZZ_MODEL=$fam timeout 200 python scripts/fuzz_parity.py 90 66000 > $O/fuzz_$fam.json 2>> $O/fuzz.err; tail -c 160 $O/fuzz_$fam.json; echo; done
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'P' | tee $O/bench_summary.txt
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_r26/bench.json').read())
print('value', round(d['value']/1e9,2), 'ms_per_step', round(d['ms_per_step'],2), 'roof', d['roofline']['bound'], round(d['roofline']['frac'] or 0,3), 'match', d['extra'].get('replica0_matches_cpu_oracle'), 'cpu', d['cpu_baseline'].get('value'))
b=d['extra']['best_score_at_60s']; print('m2', b['gpu'], b['cpu_oracle'].get('best_score') if b.get('cpu_oracle') else None, round(b['gpu_moves_per_s_rank0']/1e9,2), b['seconds_to_cpu_best'])
s=d['extra']['side_configs']; r=s['cvrp1000_default_list_policy']['roofline']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('bound','frac','salu_frac','valu_frac','hbm_frac','replay_matches_leg')}); print(round(s['cvrp5000_nearby2']['moves_per_s_rank0']/1e9,2))
P
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --solve-seconds 0 --no-cpu-baseline > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; find $O/trace -name "*.db" -delete; find $O/trace -name "*kernel_trace.csv" -delete; head -3 $O/bench_kernel_stats.csv | cut -c1-200
cd $R
for pol in la sa; do
  timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 > $O/graph_$pol.json
  timeout 600 python scripts/pmc_run.py k_scalar_search_wave 1 $O/graph_${pol}_pmc.json -- python $R/scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | cut -c1-300
done
timeout 300 python scripts/jobshop_bench.py 2>&1 | tail -1 > $O/jobshop.json
timeout 600 python scripts/pmc_run.py k_mixed_search_wave 1 $O/jobshop_pmc.json -- python $R/scripts/jobshop_bench.py 2>&1 | tail -1 | cut -c1-300
