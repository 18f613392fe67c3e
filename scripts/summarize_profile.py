#!/usr/bin/env python
"""Collects the rocprofv3 outputs of one gpurun profiling call (gpurun_out/<dir>) into profiles/:
kernel stats CSV, a per-launch PMC summary of the search kernel, and the HBM-traffic record
bench.py reports as roofline.traffic.  Usage: summarize_profile.py <gpurun_out dir> <tag> <engine>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, engine = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
shutil.copy(os.path.join(src, "prof", "bench_kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench_prof.json")).read().strip().splitlines()[-1])
out = {"_command": "rocprofv3 --pmc <counters> -f csv -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline "
                   "(one pass per counter group; FETCH_SIZE and WRITE_SIZE in separate passes)",
       "_kernel": bench["roofline"]["kernel"], "_config": bench["config"]}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(os.path.join(d, "b_counter_collection.csv"))):
        if "k_list_search" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            out["_vgpr"] = int(row["VGPR_Count"]); out["_sgpr"] = int(row["SGPR_Count"])
            out["_lds_block_size"] = int(row["LDS_Block_Size"]); out["_scratch"] = int(row["Scratch_Size"])
    for c, vals in agg.items():
        out[c] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
# rocprofv3's kernel_stats averages every launch of the run (warm-up launches included, and those
# run the cheaper early search steps); bench.py times only the last `steps` launches.  Average the
# same launches from the kernel trace so the two numbers can be compared directly.
trace = os.path.join(src, "prof", "bench_kernel_trace.csv")
if os.path.exists(trace):
    rows = [r for r in csv.DictReader(open(trace)) if "k_list_search" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    k = bench["steps"]
    out["_kernel_trace"] = {"launches": len(durs), "avg_ms_all_launches": sum(durs) / max(len(durs), 1),
                            "timed_launches": k, "avg_ms_timed_launches": sum(durs[-k:]) / max(len(durs[-k:]), 1),
                            "bench_avg_launch_ms_same_run": bench["roofline"]["avg_launch_ms"]}
json.dump(out, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # rocprofv3 units: KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of
    # the bytes of wide (16 B/lane) coalesced streaming reads; this kernel's loads are 4-8 B/lane
    # gathers and 8 B/lane chunk reads, which are uncalibrated, so both the raw and the doubled read
    # figure are recorded and bench.py reports the conservative (doubled) one.
    f, w = out["FETCH_SIZE"]["mean_per_launch"] * 1024, out["WRITE_SIZE"]["mean_per_launch"] * 1024
    json.dump({"engine": engine, "replicas": bench["config"]["replicas_per_gpu"],
               "ls_steps": bench["config"]["ls_steps_per_launch"],
               "fetch_bytes_per_launch_raw": f, "write_bytes_per_launch": w,
               "hbm_bytes_per_launch": 2 * f + w,
               "source": f"profiles/{tag}_pmc.json"},
              open(os.path.join(prof, f"pmc_traffic_{engine}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
