#!/bin/bash
# rocprofv3 kernel trace of bench.py at its default flags (20 timed launches after 3 warm-up launches): the mean duration of the timed
# launches from the trace beside the live figure of the bench line.  Usage (via gpurun): bash scripts/r03_bench_profile.sh <tag>
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/prof2 -o bench -- python $R/bench.py --no-cpu-baseline --no-pmc --solve-seconds 0 > $O/bench_prof2.json 2> $O/bench_prof2.err
find $O/prof2 -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_default_flags.csv \;
python - <<PY
import csv, glob, json
f = glob.glob("$O/prof2/**/bench_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_list_search_wave" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
line = json.loads(open("$O/bench_prof2.json").read().strip().split("\n")[-1])
out = {"kernel": rows[0]["Kernel_Name"][:60], "calls": len(d), "all_calls_mean_ms": sum(d) / len(d), "timed_last20_mean_ms": sum(d[-20:]) / 20,
       "warmup_ms": d[:3], "bench_line_avg_launch_ms": line["roofline"]["avg_launch_ms"], "bench_line_value": line["value"]}
open("$O/bench_trace_vs_live.json", "w").write(json.dumps(out))
print(json.dumps(out))
PY
