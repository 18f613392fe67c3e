#!/bin/bash
# A/B of wave-kernel variants on the GPU box: parity tests of the list engines, then short bench lines.
# Usage: bash scripts/ab_wave.sh <tag> [lib1.so lib2.so ...]   (default: the in-tree library, MODE 2 vs SF_AMD_NO_SMALL=1)
tag=${1:-ab}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; [ -x build/exp/dpp_run ] && build/exp/dpp_run
timeout 600 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
if [ $# -eq 0 ]; then
  for i in 1 2; do
    $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small ', d['value']/1e9, d['roofline']['avg_launch_ms'], d['extra']['replica0_matches_cpu_oracle'])"
    SF_AMD_NO_SMALL=1 $B --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode1 ', d['value']/1e9, d['roofline']['avg_launch_ms'])"
  done
else
  for lib in "$@"; do
    for i in 1 2; do
      SF_AMD_LIB=$R/$lib $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value']/1e9, d['roofline']['avg_launch_ms'], d['extra'].get('replica0_matches_cpu_oracle'))"
    done
  done
fi
