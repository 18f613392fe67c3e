#!/bin/bash
# rocprofv3 PMC passes (instruction mix, activity) of the generic engine on the six-leaf CVRP-1000 union.
# Usage (via gpurun): bash scripts/pmc_generic.sh <tag>   -> gpurun_out/<tag>/generic_pmc.json
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag/gpmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
U=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o g6 -- python $R/scripts/union_probe.py 3072 100 3 $U > $O/probe.json 2> $O/trace_err.log
find $O/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/generic_kernel_stats.csv \;
tail -1 $O/probe.json | cut -c1-300
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -f csv -d $O/pmc_$i -- python $R/scripts/union_probe.py 3072 100 3 $U > /dev/null 2> $O/err_$i.log
done
python $R/scripts/pmc_dump.py $O k_mixed_search_wave > $R/gpurun_out/$tag/generic_pmc.json
cat $R/gpurun_out/$tag/generic_pmc.json | python -c "
import json,sys; d=json.load(sys.stdin); print({k:(round(v['mean']) if isinstance(v,dict) else v) for k,v in d.items()})"
