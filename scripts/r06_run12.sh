#!/bin/bash
# round 6, twelfth GPU call: more launch shapes (M1, M2 seven- and six-leaf), the marginal cost of a ruin trial (ruin candidates per step 10 / 5 / 1)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r12; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-pmc --solve-seconds 0 --steps 8 --warmup 3 --no-cpu-baseline"
for rep in 73728 98304; do
  timeout 300 $B --replicas $rep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('replicas $rep', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/m1_shapes.txt
done
for cfg in "24576 default 100000 60" "12288 default 300000 30" "12288 default6 100000 100" "24576 default6 100000 60" "24576 default6 30000 150"; do
  set -- $cfg
  timeout 500 python scripts/m2_probe.py $1 $2 $4 4 $3 2>&1 | tail -1 | cut -c1-330 | tee -a $O/m2_shapes.jsonl
done
for mps in 10 5 1; do
  SF_PROBE_RUIN_MPS=$mps timeout 400 python scripts/m2_probe.py 6144 default 250 6 30000 2>&1 | tail -1 | cut -c1-360 | tee -a $O/ruin_mps.jsonl
done
