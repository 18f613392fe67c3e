#!/bin/bash
# round 5, first GPU session: parity of the touched wave engine, the SALU ceiling, the PMC probe, baselines of C2 / C4 / C5 / generic engine
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu 2>&1 | tail -2 | tee $O/parity.txt
timeout 120 build/salu_microbench | tee $O/salu_microbench.jsonl
timeout 1500 bash scripts/r05_pmc_probe.sh 2>&1 | tail -20
for pol in la sa; do timeout 300 python scripts/graph_bench.py 2048 100 10 $pol > $O/graph_$pol.json 2> $O/graph_$pol.err; tail -c 1500 $O/graph_$pol.json; done
timeout 300 python scripts/jobshop_bench.py > $O/jobshop.json 2> $O/jobshop.err; tail -c 1500 $O/jobshop.json
timeout 300 python bench.py --customers 5000 --vehicles 500 --replicas 1280 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 3000 $O/c5_bench.json
