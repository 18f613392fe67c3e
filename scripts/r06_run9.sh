#!/bin/bash
# round 6, ninth GPU call: precedence stages with by-reference parameter blocks, force-inlined GCarve / ScalarModel::tables, KoptS setters as value selects:
# whole GPU suite + fuzz (cvrp, precedence), M2 rates of the library before (build/libsf_g1.so: no KoptS change) and after, precedence rates
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r9; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
SF_FUZZ_MODEL=cvrp timeout 240 python scripts/fuzz_parity.py 100 63000 > $O/fuzz_cvrp.json 2> $O/fuzz.err; tail -c 200 $O/fuzz_cvrp.json; echo
SF_FUZZ_MODEL=precedence timeout 240 python scripts/fuzz_parity.py 100 63500 > $O/fuzz_prec.json 2>> $O/fuzz.err; tail -c 200 $O/fuzz_prec.json; echo
for lib in build/libsf_g1.so solverforge_amd/libsolverforge_amd.so; do
for cfg in "6144 default" "12288 default6"; do
  set -- $cfg
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py $1 $2 250 8 2>&1 | tail -1 | sed "s|^|$lib |" | tee -a $O/m2_late.jsonl
  SF_AMD_LIB=$R/$lib timeout 300 python scripts/m2_probe.py $1 $2 8 8 2>&1 | tail -1 | sed "s|^|$lib |" | tee -a $O/m2_early.jsonl
done
done
for cfg in "50 20 2048"; do set -- $cfg; echo "four-leaf $cfg: $(timeout 300 python scripts/precedence_bench.py $1 $2 $3 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | cut -c1-400)" | tee -a $O/prec_rates.txt; done
echo "nine-leaf 50 20: $(timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-300)" | tee -a $O/prec_rates.txt
echo "c4 makespan: $(timeout 300 python scripts/c4_makespan_rate.py 256 5 2 2>&1 | tail -1 | cut -c1-300)" | tee -a $O/prec_rates.txt
