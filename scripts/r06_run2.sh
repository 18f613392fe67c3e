#!/bin/bash
# round 6, second GPU call: parity of the rewritten wave replay + the cursor step, A/B against the round-5 wave kernels, scalar-instruction fit,
# and the M2 regime of the generic engine (rates + phase shares)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cvrp.py tests/test_gpu_budget.py tests/test_gpu_foragers.py tests/test_gpu_provider_step.py tests/test_gpu_grouped.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline"
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_wave_base.so solverforge_amd/libsolverforge_amd.so build/libsf_wave_base.so; do
  SF_AMD_LIB=$R/$lib timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e9,2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/ab.txt
done
timeout 600 python scripts/salu_fit.py $O/salu_fit.json 2>&1 | tail -30 > $O/salu_fit.log
for cfg in "6144 default" "12288 default6"; do
  set -- $cfg
  timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | tee -a $O/m2_rates.jsonl
  SF_AMD_LIB=$R/build/libsf_phase.so timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | tee -a $O/m2_phases.jsonl
done
