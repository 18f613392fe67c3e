#!/bin/bash
# round 6, third GPU call: parity of the wave-engine changes (resolve fast path, merged item), A/B, and the LATE M2 regime of the generic engine
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r3; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_provider_step.py tests/test_gpu_grouped.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5"
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_wave_base.so solverforge_amd/libsolverforge_amd.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3), d['extra'].get('replica0_matches_cpu_oracle'))" | tee -a $O/ab.txt
done
timeout 400 python scripts/salu_fit.py $O/salu_fit.json 2>&1 | tail -30 > $O/salu_fit.log
for cfg in "6144 default" "12288 default6"; do
  set -- $cfg
  timeout 300 python scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | tee -a $O/m2_late_rates.jsonl
  SF_AMD_LIB=$R/build/libsf_phase.so timeout 400 python scripts/m2_probe.py $1 $2 200 6 2>&1 | tail -1 | tee -a $O/m2_late_phases.jsonl
done
