#!/bin/bash
# round 6: counter profiles of the generic engine's FAST kernels on the round's last library (seven- and six-leaf, long-step regime, the M2 launch shape)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r36; mkdir -p $O; export TMPDIR=/tmp
export SF_COMMIT=$(cat $R/build/commit.txt 2>/dev/null)
for cfg in "24576 default 7" "24576 default6 6"; do
  set -- $cfg
  timeout 300 python scripts/m2_probe.py $1 $2 40 4 100000 2>&1 | tail -1 | cut -c1-330 | tee $O/generic_${3}leaf_rate.json
  timeout 800 python scripts/pmc_run.py k_mixed_search_wave 40 $O/generic_${3}leaf_pmc.json -- python $R/scripts/m2_probe.py $1 $2 40 4 100000 2>&1 | tail -1 | cut -c1-300
done
