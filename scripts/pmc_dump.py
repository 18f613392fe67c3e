#!/usr/bin/env python
"""Mean per-launch PMC counters of kernels matching a substring, over every pmc_* pass directory.
Usage: pmc_dump.py <dir with pmc_*/ subdirs> <kernel substring>"""
import collections, csv, glob, json, os, sys
src, pat = sys.argv[1], sys.argv[2]
out = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                out["_vgpr"] = int(row["VGPR_Count"]); out["_sgpr"] = int(row["SGPR_Count"])
                out["_lds"] = int(row["LDS_Block_Size"]); out["_scratch"] = int(row["Scratch_Size"])
        for c, v in agg.items():
            out[c] = {"launches": len(v), "mean": sum(v) / len(v), "last": v[-1]}
print(json.dumps(out, indent=1))
