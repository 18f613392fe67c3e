#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s21; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pair_ir.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pair_ir.txt
timeout 1200 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_mixed.py tests/test_gpu_runs.py tests/test_gpu_grouped.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
for ie in 0 1; do for pol in la sa; do echo "interpret=$ie graph $pol $(SF_AMD_IR_INTERPRET=$ie timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M moves/s', d['replica0_matches_indexed_cpu'])")" | tee -a $O/graph_ir.txt; done; done
