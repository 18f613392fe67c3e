#!/bin/bash
# round 6, eighth GPU call: generic engine with the scheduler's cycle order cached across the batches of a step; wave kernels with the route table
# back in LDS for the LDS-resident layouts (RouteArith for the NODEG layout only): whole GPU suite, fuzz, M2 rates, A/B
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r8; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
SF_FUZZ_MODEL=cvrp timeout 300 python scripts/fuzz_parity.py 150 62000 > $O/fuzz_cvrp.json 2> $O/fuzz.err; tail -c 200 $O/fuzz_cvrp.json; echo
for cfg in "6144 default" "12288 default6"; do
  set -- $cfg
  timeout 300 python scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | tee -a $O/m2_late.jsonl
  timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | tee -a $O/m2_early.jsonl
done
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline"
for lib in build/libsf_wbase.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
done
C5="python bench.py --customers 5000 --vehicles 500 --replicas 2816 --ls-steps 100 --steps 6 --warmup 2 --solve-seconds 0 --no-cpu-baseline --no-pmc"
timeout 300 $C5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
