#!/bin/bash
# One measurement pass on the GPU box: bench line, rocprofv3 kernel stats of the same command, generic-engine union probe
# with its kernel stats, 60 s solves (work-balanced launches).  Usage (via gpurun): bash scripts/measure_round.sh <tag>
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
U=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
python $R/scripts/union_probe.py 2048 100 5 $U > $O/union.json 2> $O/union.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_union -o union -- python $R/scripts/union_probe.py 2048 100 3 $U > /dev/null 2> $O/union_prof.err
python $R/scripts/solve60.py 60 4096 nearby_change,nearby_swap 100000 > $O/solve60.json 2> $O/solve60.err
python $R/scripts/solve60.py 60 2048 $U 30000 > $O/solve60_union.json 2> $O/solve60_union.err
find $O -name "*_kernel_stats.csv" | head; tail -c 1500 $O/bench.json; echo; cut -c1-330 $O/union.json
