#!/bin/bash
# A/B of library variants on the GPU box (scripts/wave_variant.sh builds them): CVRP parity tests of the list engines against the
# oracle for every variant, then two short bench lines each.   usage: ab_libs.sh <lib.so> [<lib.so> ...]   (paths relative to the repo)
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
for lib in "$@"; do
  SF_AMD_LIB=$R/$lib timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu 2>&1 | tail -1
  for i in 1 2; do
    SF_AMD_LIB=$R/$lib $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3), d['extra'].get('replica0_matches_cpu_oracle'))"
  done
done
