# precedence constraint, trial scoring variants (one MI355X): LDS full evaluation, HBM lane-per-trial sweep (default with HBM scratch),
# HBM full evaluation (SF_AMD_PREC_NO_SWEEP=1); C4 = mixed job shop 500 x 20 + makespan (HBM scratch by size)
for cfg in "50 20 2048 5 2" "100 20 1024 5 2"; do
  echo "LDS full:  $(timeout 300 python scripts/precedence_bench.py $cfg 2>&1 | tail -1 | cut -c1-330)"
  echo "HBM sweep: $(SF_AMD_PREC_HBM=1 timeout 300 python scripts/precedence_bench.py $cfg 2>&1 | tail -1 | cut -c1-330)"
done
echo "C4+makespan sweep: $(timeout 600 python scripts/jobshop_bench.py 1024 5 2 constructed makespan 2>&1 | tail -1 | cut -c1-900)"
