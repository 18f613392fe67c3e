#!/bin/bash
# rocprofv3 kernel stats + two PMC passes of the PREC x RUIN instantiation under the nine-leaf default policy of a slot with precedence
# hooks (job shop 20 x 10, 2,048 replicas, LDS scratch).  Usage (via gpurun): bash scripts/r03_prec_policy_profile.sh <tag>
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag/prec_policy
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/kt -- python $R/scripts/precedence_bench.py 20 10 2048 10 2 policy9 > $O/bench.json 2> $O/err_kt.log
find $O/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/prec_policy_kernel_stats.csv \;
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp -f csv -d $O/pmc_$i -- python $R/scripts/precedence_bench.py 20 10 2048 10 1 policy9 > /dev/null 2> $O/err_$i.log
done
python $R/scripts/pmc_dump.py $O k_mixed_search_wave > $R/gpurun_out/$tag/prec_policy_pmc.json 2>/dev/null
tail -1 $O/bench.json | cut -c1-400
head -5 $R/gpurun_out/$tag/prec_policy_kernel_stats.csv | cut -c1-300
cat $R/gpurun_out/$tag/prec_policy_pmc.json | cut -c1-600
