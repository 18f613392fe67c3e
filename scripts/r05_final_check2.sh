#!/bin/bash
# last check of the round after the route-graph filter took the unordered evaluation: the whole GPU suite, smoke, the precedence rates, a precedence fuzz
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_final2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_tests.txt
echo "nine-leaf 50 20: $(timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | cut -c1-300)" | tee $O/prec_rates.txt
echo "nine-leaf 100 20: $(timeout 300 python scripts/prec_policy_launches.py 100 20 1024 5 2 2>&1 | tail -1 | cut -c1-300)" | tee -a $O/prec_rates.txt
echo "four-leaf 50 20: $(timeout 300 python scripts/precedence_bench.py 50 20 2048 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | cut -c1-420)" | tee -a $O/prec_rates.txt
SF_FUZZ_MODEL=precedence timeout 200 python scripts/fuzz_parity.py 150 33000 > $O/fuzz_parity_precedence.json 2> $O/fuzz.err; tail -c 300 $O/fuzz_parity_precedence.json; echo
