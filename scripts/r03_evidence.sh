#!/bin/bash
# Round-3 evidence pass on the GPU box: default bench line (live PMC), rocprofv3 kernel stats of the same command, 60 s solves with the
# reference's default list policy (six and seven leaves) from the savings start, a 300 s differential fuzz run.
tag=${1:-r03h}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
U=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --solve-seconds 0 > $O/bench_prof.json 2> $O/bench_prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python $R/scripts/solve60.py 60 3072 $U 30000 savings_capacity > $O/solve60_6leaf.json 2> $O/solve60_6leaf.err
python $R/scripts/solve60.py 60 1280 $U,ruin 30000 savings_capacity > $O/solve60_7leaf.json 2> $O/solve60_7leaf.err
python $R/scripts/fuzz_parity.py 300 30000 > $O/fuzz_parity.json 2> $O/fuzz_parity.err
tail -c 600 $O/bench.json; echo; tail -c 400 $O/solve60_6leaf.json; echo; tail -c 400 $O/solve60_7leaf.json; echo; tail -c 500 $O/fuzz_parity.json
