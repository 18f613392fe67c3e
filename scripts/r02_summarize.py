#!/usr/bin/env python
"""Turns one scripts/r02_measure.sh output directory into the committed evidence under profiles/:
<tag>_bench.json (the bench line), <tag>_<engine>_pmc.json (per-launch PMC means of the timed launches + kernel trace
durations of the same launches), <tag>_kernel_stats.csv (rocprofv3 --stats).  Usage: r02_summarize.py <dir> <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
line = [l for l in open(os.path.join(src, "bench.json")).read().splitlines() if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(prof, f"{tag}_bench.json"), "w"), indent=1)
engine = bench["config"]["engine"]
roof = bench["roofline"]
out = {"_command": "python bench.py (default flags): rocprofv3 --pmc child passes of the same command, mean over the timed launches",
       "_kernel": roof["kernel"], "_pmc_source": roof.get("pmc_source"), "_kernel_resources": roof.get("kernel_resources"),
       "_config": dict(bench["config"], steps=bench["steps"], warmup=bench["warmup"]),
       "_bench_avg_launch_ms": roof["avg_launch_ms"], "_bench_value": bench["value"],
       "_frac": {k: roof.get(k) for k in ("bound", "frac", "valu_frac", "salu_frac", "lds_issue_frac", "hbm_frac", "wave_cycle_shares", "effective_clock_ghz")}}
for k, v in (roof.get("counters_per_launch") or {}).items():
    out[k] = {"launches": bench["steps"], "mean_per_launch": v}
for f in glob.glob(os.path.join(src, "prof", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(prof, f"{tag}_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "prof", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_list_search" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    k = bench["steps"]
    pl = [l for l in open(os.path.join(src, "bench_prof.json")).read().splitlines() if l.startswith("{")]
    pb = json.loads(pl[-1]) if pl else None
    out["_kernel_trace"] = {"launches": len(durs), "avg_ms_all_launches": sum(durs) / max(len(durs), 1), "timed_launches": k,
                            "avg_ms_timed_launches": sum(durs[-k:]) / max(len(durs[-k:]), 1),
                            "bench_avg_launch_ms_same_run": pb["roofline"]["avg_launch_ms"] if pb else None}
json.dump(out, open(os.path.join(prof, f"{tag}_{engine}_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("_")}, indent=1))
