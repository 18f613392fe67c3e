#!/bin/bash
# round 6, twenty-third GPU call: the 60 s M2 leg sustained at three replica counts; the specialised scalar units at eight partner ids per pass (parity + rate)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r23; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_pair_ir.py tests/test_gpu_anneal.py tests/test_gpu_grouped.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -2 | tee $O/tests.txt
for pol in la sa; do echo "graph $pol: $(timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/graph.txt; done
for rep in 12288 18432 24576; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --tuned-seconds 0 --c5-seconds 0 --solve-seconds 60 --solve-replicas $rep 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); b=d['extra']['best_score_at_60s']
print('m2 replicas $rep', b['gpu'], round(b['gpu_moves_per_s_rank0']/1e9,3), 'launches', b['gpu_launches'], 'steps', b['gpu_ls_steps_rank0'])" | tee -a $O/m2_sustained.txt
done
