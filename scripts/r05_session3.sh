#!/bin/bash
# round 5, third GPU session: mode 6 (node table in HBM) parity + C5 A/B, SALU ceiling, generic-engine phase probes, the full bench line
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_migrate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
hipcc --offload-arch=gfx950 -O3 scripts/salu_microbench.hip -o /tmp/salu_microbench 2> $O/salu_build.err && timeout 60 /tmp/salu_microbench > $O/salu_microbench.jsonl 2> $O/salu_microbench.err; cat $O/salu_microbench.jsonl
for ng in 1 0; do for rep in 2048 4096; do SF_AMD_NODE_GLOBAL=$ng timeout 300 python bench.py --customers 5000 --vehicles 500 --replicas $rep --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 node_global=$ng replicas $rep', round(d['value']/1e9,3))"; done; done
timeout 400 python bench.py --customers 5000 --vehicles 500 --replicas 2048 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 2500 $O/c5_bench.json
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 600 python scripts/phase_probe_generic.py 2048 $L7 0 > $O/phase7_early.txt 2>&1; tail -8 $O/phase7_early.txt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 900 python scripts/phase_probe_generic.py 2048 $L7 1500 > $O/phase7_late.txt 2>&1; tail -8 $O/phase7_late.txt
SF_AMD_LIB=$R/build/libsf_phase.so timeout 900 python scripts/phase_probe_generic.py 3072 $L6 1500 > $O/phase6_late.txt 2>&1; tail -4 $O/phase6_late.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json
