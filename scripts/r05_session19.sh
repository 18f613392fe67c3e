#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
for i in 1 2; do timeout 300 $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanehash', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],2), d['extra'].get('replica0_matches_cpu_oracle'))" | tee -a $O/wave.txt; done
timeout 600 python scripts/salu_fit.py $O/salu_fit.json 2>&1 | tail -22 | head -12
