cd /tmp && export TMPDIR=/tmp
# usage: [PREC_CFG="50 20 2048 10 4"] [PREC_OUT=r05_prec] prec_profile.sh   (defaults: the round-4 profile, 20 x 10)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PREC_OUT:-r04_prec}; mkdir -p $O
CMD="python $R/scripts/prec_policy_launches.py ${PREC_CFG:-20 10 2048 10 4}"
$CMD > $O/plain.json 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$n -o p -- $CMD > $O/pmc_$n.log 2>&1 || echo "pass $n rc=$?"
done
python $R/scripts/pmc_dump.py $O k_mixed_search_wave > $O/pmc.json; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/plain.json | tail -1; tail -5 $O/trace.log; find $O/trace -name "*kernel_stats.csv" | head -1 | xargs head -5; head -c 1500 $O/pmc.json; du -sh $O
