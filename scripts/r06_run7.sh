#!/bin/bash
# round 6, seventh GPU call: COMPACT wave kernels with arithmetic route ranks (RouteArith: no per-step rank table, no rtab in HBM at CVRP-5000):
# parity, A/B at CVRP-1000 against the previous kernel, the CVRP-5000 line with counters
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r7; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline"
for lib in build/libsf_wbase.so solverforge_amd/libsolverforge_amd.so build/libsf_wbase.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
done
C5="python bench.py --customers 5000 --vehicles 500 --replicas 2816 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline"
for lib in build/libsf_wbase.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $C5 --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
done
timeout 900 $C5 2>$O/c5.err | tail -1 > $O/c5_bench.json; cut -c1-1500 $O/c5_bench.json
