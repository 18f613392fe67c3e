#!/bin/bash
# rocprofv3 PMC passes of the generic engine's PREC instantiation (job shop 50x20 with the makespan objective, LDS scratch).
# Usage (via gpurun): bash scripts/pmc_precedence.sh <tag>   -> gpurun_out/<tag>/precedence_pmc.json
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag/prec_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -f csv -d $O/pmc_$i -- python $R/scripts/precedence_bench.py 50 20 2048 2 1 > /dev/null 2> $O/err_$i.log
done
python $R/scripts/pmc_dump.py $O k_mixed_search_wave > $R/gpurun_out/$tag/precedence_pmc.json
python -c "
import json; d=json.load(open('$R/gpurun_out/$tag/precedence_pmc.json')); print({k:(round(v['mean']) if isinstance(v,dict) else v) for k,v in d.items()})"
