#!/bin/bash
# round 6, sixth GPU call: the paired pass with an uneven lane split (SF_PAIR_SPLIT): parity of the 24-lane split, A/B of 32 / 28 / 24 / 20 against
# the previous commit's kernel, event counts of the generation
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r6; mkdir -p $O; export TMPDIR=/tmp
SF_AMD_LIB=$R/build/libsf_ps24.so timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests_ps24.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline"
for lib in wbase ps32 ps28 ps24 ps20 wbase ps24; do
  SF_AMD_LIB=$R/build/libsf_$lib.so timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
done
SF_AMD_LIB=$R/build/libsf_gencount.so timeout 300 python scripts/gen_count.py 6144 4 2>&1 | tail -1 | tee $O/gen_count.json
