# FAST instantiation of the generic engine (six-leaf default-policy union): occupancy variants (diagnostic builds in csrc/_diag)
U=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
D=solverforge_amd/csrc/_diag
for N in 1000; do
  for cfg in "solverforge_amd/libsolverforge_amd.so 3072" "solverforge_amd/libsolverforge_amd.so 6144" "$D/libsf_fast2.so 2048" "$D/libsf_fast2.so 4096" "$D/libsf_fast2_ringlds.so 2048" "$D/libsf_fast4.so 4096"; do
    set -- $cfg
    echo "N=$N lib=$1 R=$2: $(SF_AMD_LIB=$PWD/$1 python scripts/union_probe.py $2 100 3 $U $N 2>&1 | tail -1 | cut -c1-300)"
  done
done
