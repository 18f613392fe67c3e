# precedence constraint: incremental trial refresh (HBM scratch) vs the LDS full evaluation; C4 + makespan
mkdir -p gpurun_out/r03f
for cfg in "50 20 2048 5 2" "100 20 1024 5 2"; do
  echo "LDS/auto: $(python scripts/precedence_bench.py $cfg 2>&1 | tail -1 | cut -c1-420)"
  echo "HBM inc:  $(SF_AMD_PREC_HBM=1 python scripts/precedence_bench.py $cfg 2>&1 | tail -1 | cut -c1-420)"
done
echo "C4+makespan: $(python scripts/jobshop_bench.py 1024 5 2 constructed makespan 2>&1 | tail -1 | cut -c1-700)"
