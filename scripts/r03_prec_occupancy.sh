# PREC instantiations at four workgroups per CU (MODE 2) vs two (SF_AMD_PREC_NO_OCC=1): nine-leaf policy, one MI355X
for cfg in "10 5 4096 20 2" "20 10 4096 10 2" "50 20 2048 3 2"; do
  echo "MODE 2 / auto: $(timeout 400 python scripts/precedence_bench.py $cfg policy9 2>&1 | tail -1 | cut -c1-420)"
  echo "MODE 0 forced: $(SF_AMD_PREC_NO_OCC=1 timeout 400 python scripts/precedence_bench.py $cfg policy9 2>&1 | tail -1 | cut -c1-420)"
done
