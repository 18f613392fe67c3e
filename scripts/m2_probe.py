"""The M2 regime of bench.py in a few seconds: CVRP-1000, Clarke-Wright (capacity hook) + ListKOpt start, the default list policy on the generic
N-leaf engine, work-balanced launches (sf_solve_moves) -- near a local optimum a step consumes thousands of candidates, which is what the driver's
60 s leg prices (`extra.side_configs.cvrp1000_default_list_policy`); scripts/generic_step_time.py measures the round-robin start instead.
usage: m2_probe.py <replicas> <leaves,comma | default | default6> [warm launches=8] [timed launches=8] [budget=30000]
With a -DSF_PHASE_PROFILE library (SF_AMD_LIB) the per-phase shader-clock shares of the timed launches are printed too
(0 step start + ruin trials, 1 fill nearby / plain, 2 fill sublist, 3 fill reverse / ruin, 4 fill 3-opt, 5 scheduler layout, 6 replay, 7 commit)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib

POL = {"default": ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin"),
       "default6": ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt"),
       "nearby2": ("nearby_change", "nearby_swap")}
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
leaves = POL.get(sys.argv[2], tuple(sys.argv[2].split(","))) if len(sys.argv) > 2 else POL["default"]
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 8
budget = int(sys.argv[5]) if len(sys.argv) > 5 else 30000
p = datasets.make_cvrp(1000, 100, 55, seed=0)
p["routes"] = [[] for _ in p["routes"]]
ruin_mps = int(os.environ.get("SF_PROBE_RUIN_MPS", "10"))  # diagnostics: ruin candidates per step (the default policy's 10): the marginal cost of a ruin trial
d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, ruin=(2, 5, ruin_mps))
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score()
d.construct_list_clarke_wright(0, p["customers"], 1)
start = [int(v) for v in d.construct_list_k_opt(0, 2, 1)[0]]
d.phase_start()
L = _lib.load()
TU = "2_2_1_0" if "ruin" in leaves else "2_2_0_0"
phases = getattr(L, "sf_debug_phases_mixed_" + TU, None)
ph = np.zeros(8, dtype=np.uint64)
for _ in range(warm):
    d.solve_moves(1 << 20, budget, sync=True)
d.profile_solve()
if phases:
    phases(ph.ctypes.data_as(ctypes.c_void_p))
b = d.total_stats()
t0 = time.perf_counter()
for _ in range(timed):
    d.solve_moves(1 << 20, budget, sync=True)
dt = time.perf_counter() - t0
ms, n = d.profile_solve()
a = d.total_stats()
mv = a["moves_evaluated"] - b["moves_evaluated"]
st = a["step_count"] - b["step_count"]
out = {"replicas": R, "leaves": len(leaves), "ruin_moves_per_step": ruin_mps, "start_score": start, "budget": budget, "timed_launches": timed, "kernel_ms_per_launch": round(ms / max(n, 1), 3),
       "G_moves_per_s_wall": round(mv / dt / 1e9, 3), "G_moves_per_s_kernel": round(mv / ms / 1e6, 3), "moves_per_step": round(mv / max(st, 1), 1),
       "steps_per_replica_per_launch": round(st / R / timed, 2), "sources_per_step": round((a["sources_scanned"] - b["sources_scanned"]) / max(st, 1), 1),
       "scored_per_step": round((a["candidates_scored"] - b["candidates_scored"]) / max(st, 1), 1),
       "best": list(max(tuple(int(v) for v in s) for s in d.best_scores()))}
if phases:
    phases(ph.ctypes.data_as(ctypes.c_void_p))
    tot = float(ph.sum())
    out["phase_shares_pct"] = [round(float(x) / tot * 100, 1) for x in ph]
    out["clocks_per_step_per_wave"] = round(tot / max(st, 1))
    for name in ("sf_debug_ruin2_phases_mixed_", "sf_debug_ruin_phases_mixed_"):
        f = getattr(L, name + TU, None)
        if f and "ruin" in leaves:
            ro = np.zeros(8, dtype=np.uint64)
            f(ro.ctypes.data_as(ctypes.c_void_p))
            out[name[9:-7]] = {"clocks_per_step_per_wave": round(float(ro.sum()) / max(st, 1)), "shares_pct": [round(float(x) / max(float(ro.sum()), 1) * 100, 1) for x in ro]}
print(json.dumps(out))
