#!/bin/bash
# round 6: the differential fuzz with the round's new entry points in its pool -- uni programs in the assignment family, sf_step_evaluate / sf_apply under the two-class join
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r30; mkdir -p $O; export TMPDIR=/tmp
for fam in assignment jobshop; do SF_FUZZ_MODEL=$fam timeout 200 python scripts/fuzz_parity.py 100 67000 > $O/fuzz_$fam.json 2>> $O/fuzz.err; tail -c 700 $O/fuzz_$fam.json; echo; done
timeout 200 python scripts/fuzz_parity.py 100 68000 > $O/fuzz_all.json 2>> $O/fuzz.err; tail -c 300 $O/fuzz_all.json; echo
