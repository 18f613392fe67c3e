#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s28; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_anneal.py tests/test_gpu_foragers.py tests/test_gpu_nearby_scalar.py tests/test_gpu_value_lists.py tests/test_gpu_pair_ir.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/parity.txt
for pol in sa la; do echo "graph $pol $(timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M moves/s', round(d['gpu_steps_per_s']/1e6,2),'M steps/s', d['replica0_matches_indexed_cpu'], d['fill_calls_per_step'])")" | tee -a $O/graph.txt; done
SF_FUZZ_MODEL=graph timeout 200 python scripts/fuzz_parity.py 100 15000 2>/dev/null | tail -c 300
