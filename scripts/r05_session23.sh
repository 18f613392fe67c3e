#!/bin/bash
# round 5 end-game: whole GPU suite + fuzz + the default bench line + rocprofv3 kernel trace of the default command, all on the committed library
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s23; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_tests.txt
timeout 400 python scripts/fuzz_parity.py 200 7000 > $O/fuzz_parity.json 2> $O/fuzz_parity.err; tail -c 400 $O/fuzz_parity.json; echo
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -o b -- python $R/bench.py --no-pmc > $O/bench_traced.json 2> $O/ktrace.err
find $O/ktrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -6 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/ktrace
