#!/bin/bash
# Round-2 measurement pass on the GPU box (via gpurun): GPU tests, the default bench line (live rocprofv3 --pmc child
# passes + 60 s solve leg), and the rocprofv3 kernel trace / stats of the same timed launches.
# Usage: bash scripts/r02_measure.sh <tag> [skip-tests]
tag=${1:-r02x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
if [ -z "$2" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
  tail -3 $O/gputests.log
fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --no-pmc --solve-seconds 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
python $R/scripts/r02_summarize.py $O $tag
tail -c 3000 $O/bench.json
