#!/bin/bash
# round 6, first GPU call: counter profiles of the generic engine on CVRP-1000 (seven / six leaves, one residency and the M2 replica counts),
# and of C2 / C4 on the library that ships.  usage (from the container): gpurun -- "SF_COMMIT=$(git rev-parse --short HEAD) bash scripts/r06_profile1.sh"
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_p1; mkdir -p $O; export TMPDIR=/tmp
L7=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin
L6=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
for cfg in "7 3072 $L7" "7 6144 $L7" "6 4096 $L6" "6 12288 $L6"; do
  set -- $cfg
  echo "== $1-leaf $2 replicas" | tee -a $O/log.txt
  timeout 300 python scripts/generic_step_time.py $2 $3 600 2>&1 | tail -1 | tee $O/generic_${1}leaf_$2_rate.json
  timeout 900 python scripts/pmc_run.py k_mixed_search_wave 2 $O/generic_${1}leaf_$2_pmc.json -- python $R/scripts/generic_step_time.py $2 $3 600 2>&1 | tail -1 | cut -c1-600 | tee -a $O/log.txt
done
for pol in la sa; do
  timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | tee $O/graph_$pol.json
  timeout 900 python scripts/pmc_run.py k_scalar_search_wave 1 $O/graph_${pol}_pmc.json -- python $R/scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | cut -c1-600 | tee -a $O/log.txt
done
timeout 300 python scripts/jobshop_bench.py 2>&1 | tail -1 | tee $O/jobshop.json
timeout 900 python scripts/pmc_run.py k_mixed_search_wave 1 $O/jobshop_pmc.json -- python $R/scripts/jobshop_bench.py 2>&1 | tail -1 | cut -c1-600 | tee -a $O/log.txt
