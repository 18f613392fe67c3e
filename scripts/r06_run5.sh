#!/bin/bash
# round 6, fifth GPU call: the generic FAST kernels' scoring stage (deltas computed one leaf at a time, lump fills): whole GPU suite, fuzz, M2 rates
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r5; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 | tee $O/tests.txt
SF_FUZZ_MODEL=cvrp timeout 300 python scripts/fuzz_parity.py 200 61000 > $O/fuzz_cvrp.json 2> $O/fuzz.err; tail -c 300 $O/fuzz_cvrp.json; echo
for cfg in "6144 default" "12288 default6"; do
  set -- $cfg
  timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | tee -a $O/m2_early.jsonl
  SF_AMD_MIXED_NO_PRE_EVAL=1 timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | tee -a $O/m2_early_nopre.jsonl
  timeout 300 python scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | tee -a $O/m2_late.jsonl
  SF_AMD_MIXED_NO_PRE_EVAL=1 timeout 300 python scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | tee -a $O/m2_late_nopre.jsonl
done
