"""Where the scalar instructions of the wave kernel go: SQ_INSTS_SALU / SQ_INSTS_VALU of bench.py's timed launches at several (max_nearby,
AcceptedCount limit) settings, fitted as  a x steps + b x sources + c x replay batches  (batches = candidates scored / 64).
usage: salu_fit.py <out.json>     (runs rocprofv3 --pmc children of bench.py at 6,144 replicas)"""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for k, lim in [(20, 256), (5, 256), (40, 256), (20, 64), (20, 1024), (10, 128)]:
    base = tempfile.mkdtemp(prefix="salufit_", dir="/tmp")
    work = os.path.join(base, "work.json")
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "--kernel-include-regex", "k_list_search_wave", "-f", "csv", "-d", base, "-o", "b", "--", sys.executable,
           os.path.join(R, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3", "--replicas", "6144", "--ls-steps", "200", "--max-nearby", str(k), "--accepted-limit", str(lim),
           "--pmc-child", "--pmc-child-out", work]
    ok = False
    for attempt in range(3):
        try:
            pr = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            ok = pr.returncode == 0
        except subprocess.TimeoutExpired:
            ok = False
        if ok:
            break
    if not ok:
        continue
    per = {}
    for f in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    w = json.load(open(work))
    salu = sum(v for _, v in sorted(per["SQ_INSTS_SALU"])[3:]); valu = sum(v for _, v in sorted(per["SQ_INSTS_VALU"])[3:])
    rows.append({"max_nearby": k, "limit": lim, "salu": salu, "valu": valu, "steps": w["ls_steps"], "sources": w["sources_scanned"], "batches": w["candidates_scored"] / 64.0,
                 "moves": w["moves_evaluated"]})
    shutil.rmtree(base, ignore_errors=True)
    print(json.dumps(rows[-1]), flush=True)
A = np.array([[r["steps"], r["sources"], r["batches"]] for r in rows], dtype=np.float64)
fit = {}
for name in ("salu", "valu"):
    y = np.array([r[name] for r in rows], dtype=np.float64)
    coef, res, rank, sv = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    fit[name] = {"per_step": coef[0], "per_source": coef[1], "per_replay_batch": coef[2], "max_rel_err": float(np.max(np.abs(pred - y) / y))}
    d = rows[0]
    fit[name]["share_at_default"] = {"steps": coef[0] * d["steps"] / d[name], "sources": coef[1] * d["sources"] / d[name], "batches": coef[2] * d["batches"] / d[name]}
json.dump({"rows": rows, "fit": fit}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(fit, indent=1))
