#!/bin/bash
# round 5, session 16: C2 / C4 re-measured with PMC, scalar engine at 12 / 16 waves per CU
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s16; mkdir -p $O
for v in prod sc3 sc4; do lib=$R/build/libsf_$v.so; [ $v = prod ] && lib=$R/solverforge_amd/libsolverforge_amd.so
  for rep in 2048 3072 4096; do for pol in sa la; do echo "$v graph $pol $rep $(SF_AMD_LIB=$lib timeout 300 python scripts/graph_bench.py $rep 100 10 $pol 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,1),'M moves/s', round(d['gpu_steps_per_s']/1e6,2),'M steps/s', d['kernel_ms_per_launch'], d['replica0_matches_indexed_cpu'])")" | tee -a $O/graph.txt; done; done; done
cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/scripts/pmc_run.py k_scalar_search_wave 1 $O/graph_la_pmc.json -- python $R/scripts/graph_bench.py 2048 100 10 la | tail -1 | cut -c1-700
timeout 600 python $R/scripts/pmc_run.py k_scalar_search_wave 1 $O/graph_sa_pmc.json -- python $R/scripts/graph_bench.py 2048 100 10 sa | tail -1 | cut -c1-700
timeout 600 python $R/scripts/pmc_run.py k_mixed_search_wave 1 $O/jobshop_pmc.json -- python $R/scripts/jobshop_bench.py | tail -1 | cut -c1-700
