#!/bin/bash
# round 6: the whole GPU suite + smoke on the library with the uni-program compile step (the kernels are the validated ones of scripts/r06_run26.sh; the C-ABI unit changed)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r29; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-pmc --solve-seconds 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench short', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee $O/bench_short.txt
