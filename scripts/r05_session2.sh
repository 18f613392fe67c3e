#!/bin/bash
# round 5, second GPU session: parity of the internal node numbering, C5 A/B, the SALU ceiling, phase probes of the generic engine late in a search
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
hipcc --offload-arch=gfx950 -O3 scripts/salu_microbench.hip -o /tmp/salu_microbench 2> $O/salu_build.err && timeout 120 /tmp/salu_microbench > $O/salu_microbench.jsonl 2> $O/salu_microbench.err; cat $O/salu_microbench.jsonl
C5="python bench.py --customers 5000 --vehicles 500 --replicas 1280 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline"
for f in 1 0; do SF_AMD_RENUMBER=$f timeout 400 $C5 > $O/c5_renumber$f.json 2> $O/c5_renumber$f.err; python - <<PY
import json
d=json.loads(open("$O/c5_renumber$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("C5 renumber=$f", round(d["value"]/1e9,3), "G moves/s; traffic", r.get("traffic"), "hbm_frac", r.get("hbm_frac"), "shares", r.get("wave_cycle_shares"), "bytes/cand", (r.get("traffic") or 0)/max(r["candidates_scored_per_launch"],1))
PY
done
for rep in 2560 5120; do SF_AMD_RENUMBER=1 timeout 300 python bench.py --customers 5000 --vehicles 500 --replicas $rep --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 renumbered replicas $rep', round(d['value']/1e9,3))"; done
for f in 1 0; do SF_AMD_RENUMBER=$f timeout 300 python bench.py --solve-seconds 0 --no-cpu-baseline --no-pmc --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 renumber=$f', round(d['value']/1e9,3))"; done
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 600 python scripts/phase_probe_generic.py 2048 $L7 0 > $O/phase7_early.txt 2>&1; tail -8 $O/phase7_early.txt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 900 python scripts/phase_probe_generic.py 2048 $L7 1500 > $O/phase7_late.txt 2>&1; tail -8 $O/phase7_late.txt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 900 python scripts/phase_probe_generic.py 3072 $L6 1500 > $O/phase6_late.txt 2>&1; tail -4 $O/phase6_late.txt
