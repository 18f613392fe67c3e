"""Folds the output of scripts/prec_profile.sh (gpurun_out/r04_prec: plain.json, the rocprofv3 kernel stats, pmc.json) into the committed
profile.  usage: prec_profile_summary.py <dir> <out.json>"""
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(os.path.join(src, "pmc.json")))
plain = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
stats = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_mixed_search_wave" in row["Name"]:
            stats = {"kernel": row["Name"].split("(")[0], "calls": int(row["Calls"]), "average_ms": float(row["AverageNs"]) / 1e6, "min_ms": float(row["MinNs"]) / 1e6,
                     "max_ms": float(row["MaxNs"]) / 1e6, "share_of_gpu_time": float(row["Percentage"]) / 100}
m = lambda k: d[k]["mean"]
out = {"command": "python scripts/prec_policy_launches.py " + (sys.argv[3] if len(sys.argv) > 3 else "20 10 2048 10 4") + " (scripts/prec_profile.sh: one rocprofv3 --kernel-trace --stats run, four --pmc passes, -f csv)",
       "untimed_run": plain, "kernel_trace": stats,
       "registers": {"arch_vgpr": d["_vgpr"], "sgpr": d["_sgpr"], "static_lds": d["_lds"], "scratch_bytes_per_lane": d["_scratch"]},
       "per_launch_mean": {k: v["mean"] for k, v in d.items() if isinstance(v, dict)},
       "derived": {"issue_share_SQ_ACTIVE_INST_ANY_over_SQ_WAVE_CYCLES": m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES"),
                   "valu_share": m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES"), "salu_share": m("SQ_ACTIVE_INST_SCA") / m("SQ_WAVE_CYCLES"),
                   "lds_share": m("SQ_ACTIVE_INST_LDS") / m("SQ_WAVE_CYCLES"), "lds_bank_conflict_over_lds_active": m("SQ_LDS_BANK_CONFLICT") / m("SQ_LDS_IDX_ACTIVE"),
                   "instructions_per_candidate": {k: m("SQ_INSTS_" + k.upper()) / plain["moves_per_launch"] for k in ("valu", "salu", "lds", "vmem_rd")}}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({"kernel_trace": stats, "derived": out["derived"]}, indent=1))
