#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_migrate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
for cfg in "8 32768" "8 24576" "8 16384" "6 24576" "8 65536"; do set -- $cfg; SF_AMD_WAVE_WPE=$1 timeout 300 $B --replicas $2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpe $1 replicas $2', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],2), d['extra'].get('replica0_matches_cpu_oracle'), d['roofline']['kernel'][:60])" | tee -a $O/wave.txt; done
