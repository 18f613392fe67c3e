#!/bin/bash
# round 6, fourth GPU call: parity + A/B of the wave engine's generation changes, scalar-instruction fit, counter profiles of the generic engine in the
# LATE M2 regime (what the driver's 60 s leg prices)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r4; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_wave_base.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3), d['extra'].get('replica0_matches_cpu_oracle'))" | tee -a $O/ab.txt
done
timeout 400 python scripts/salu_fit.py $O/salu_fit.json 2>&1 | tail -30 > $O/salu_fit.log
for cfg in "6144 default 7" "12288 default6 6"; do
  set -- $cfg
  timeout 900 python scripts/pmc_run.py k_mixed_search_wave 250 $O/generic_${3}leaf_late_pmc.json -- python $R/scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | cut -c1-400 | tee -a $O/log.txt
done
