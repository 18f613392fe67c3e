"""Diagnostic: per-phase shader-clock shares of the generic N-leaf engine on the CVRP-1000 default-policy union (needs a
-DSF_PHASE_PROFILE build passed via SF_AMD_LIB).  Phases: 0 step start + order tables, 1 fill nearby / plain leaves,
2 fill sublist leaves, 3 fill reverse, 4 fill 3-opt, 5 scheduler layout, 6 trial score + acceptor + forager, 7 commit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
leaves = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt")
if leaves[0] == "jobshop":  # BASELINE config 4: mixed job shop 500 x 20, constructed start, 4-leaf union
    d = sfa.build_jobshop(datasets.construct_jobshop(datasets.make_jobshop(500, 20)), n_replicas=R)
else:
    mps = int(sys.argv[4]) if len(sys.argv) > 4 else 10  # ruin leaf: moves_per_step (0 = the leaf declared but empty: what the RUIN instantiation costs by itself)
    d = sfa.build_cvrp(datasets.make_cvrp(1000, 100, 55, seed=0), n_replicas=R, leaves=leaves, ruin=(2, 5, mps))
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
L = _lib.load()
TU = "2_2_1_0" if "ruin" in leaves else ("4_2_0_0" if leaves[0] == "jobshop" else "2_2_0_0")  # the translation unit whose counters are read
phases = getattr(L, "sf_debug_phases_mixed_" + TU)
ruin_phases = getattr(L, "sf_debug_ruin_phases_mixed_" + TU)
out = np.zeros(8, dtype=np.uint64)
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # local-search steps before the probe (late-phase behaviour)
if warm:
    d.solve_steps(warm)
    phases(out.ctypes.data_as(ctypes.c_void_p))
    d.profile_solve()
for it in range(3):
    b = d.total_stats()
    d.solve_steps(100)
    phases(out.ctypes.data_as(ctypes.c_void_p))
    ms, n = d.profile_solve()
    a = d.total_stats()
    tot = out.sum()
    if "ruin" in leaves:  # shader clocks inside ruin_recreate: 0 remove, 1 slot prefix, 2 scan, 3 pick + bookkeeping, 4 placement, 5 undo
        ro = np.zeros(8, dtype=np.uint64)
        ruin_phases(ro.ctypes.data_as(ctypes.c_void_p))
        print("  ruin_recreate cycles/step/wave %.0f" % (ro.sum() / R / 100), "shares %", np.round(ro / max(ro.sum(), 1) * 100, 1))
        if hasattr(L, "sf_debug_ruin2_phases_mixed_" + TU):  # the list-preserving trial (sf_ruin_v2.h): 0 removal, 1 first scans, 2 best lists, 3 re-pricing, 4 pick + placement, 5 restore
            r2 = np.zeros(8, dtype=np.uint64)
            getattr(L, "sf_debug_ruin2_phases_mixed_" + TU)(r2.ctypes.data_as(ctypes.c_void_p))
            print("  ruin_trial_v2 cycles/step/wave %.0f" % (r2.sum() / R / 100), "shares %", np.round(r2 / max(r2.sum(), 1) * 100, 1))
    print("launch", it, "ms %.1f" % ms, "Gmoves/s %.2f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / ms / 1e6),
          "moves/step %.0f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / R / 100),
          "fill calls/step %.0f" % ((a["sources_scanned"] - b["sources_scanned"]) / R / 100),
          "cycles/step/wave %.0f" % (tot / R / 100), "shares %", np.round(out / tot * 100, 1))
