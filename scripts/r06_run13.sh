#!/bin/bash
# round 6, thirteenth GPU call: replicas per workgroup (1 / 2 / 4 waves) for the wave engine and the generic engine at equal residency
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r13; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-pmc --solve-seconds 0 --steps 12 --warmup 4 --no-cpu-baseline"
for w in 4 2 1 4 1; do
  SF_AMD_WAVE_WPB=$w timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wave wpb $w', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/wpb.txt
done
SF_AMD_WAVE_WPB=1 timeout 600 python -m pytest tests/test_gpu_cvrp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/wpb.txt
for w in 4 2 1; do
  SF_AMD_MIXED_WPB=$w timeout 400 python scripts/m2_probe.py 12288 default 120 4 100000 2>&1 | tail -1 | cut -c1-250 | sed "s/^/mixed7 wpb $w /" | tee -a $O/wpb.txt
  SF_AMD_MIXED_WPB=$w timeout 400 python scripts/m2_probe.py 12288 default6 120 4 100000 2>&1 | tail -1 | cut -c1-250 | sed "s/^/mixed6 wpb $w /" | tee -a $O/wpb.txt
done
C5="python bench.py --customers 5000 --vehicles 500 --replicas 2816 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc"
for w in 4 1; do
SF_AMD_WAVE_WPB=$w timeout 300 $C5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 wpb $w', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/wpb.txt
done
