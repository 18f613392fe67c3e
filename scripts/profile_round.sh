#!/bin/bash
# rocprofv3 evidence for the headline bench (run on the GPU box via gpurun): kernel trace + stats, then one PMC pass per
# counter group (FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes).  scripts/summarize_profile.py turns
# gpurun_out/<tag> into profiles/<tag>_*.  Usage: bash scripts/profile_round.sh <tag>
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- $CMD > $O/bench_prof.json 2> $O/bench_prof.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -f csv -d $O/pmc_$i -o b -- $CMD > /dev/null 2> $O/err_pmc_$i.log
done
ls $O
