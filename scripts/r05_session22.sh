#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_migrate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
for rep in 2048 2816 5632; do timeout 300 python bench.py --customers 5000 --vehicles 500 --replicas $rep --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 replicas $rep', round(d['value']/1e9,3), round(d['roofline']['avg_launch_ms'],2))" | tee -a $O/c5.txt; done
timeout 400 python bench.py --customers 5000 --vehicles 500 --replicas 2816 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5_bench.err; python -c "
import json; d=json.loads(open('$O/c5_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value']/1e9, r.get('traffic_bytes_per_candidate'), r.get('hbm_frac'), r.get('wave_cycle_shares'), r['kernel'][:70], (r.get('kernel_resources') or {}).get('failed_passes'))"
