"""Differential check of the list-preserving ruin trial (csrc/sf_ruin_v2.h) against the recreate of csrc/sf_ruin.h: a -DSF_RUIN_V2_CHECK build
(scripts/tu_variant.sh rv2chk "-DSF_RUIN_V2_CHECK -DSF_RUIN_INLINE" mixed_2_2_1_0, passed via SF_AMD_LIB) scores every ruin candidate of a
seven-leaf search with both and counts the disagreements on the device.  usage: ruin_v2_check.py [steps]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L = _lib.load()
chk = L.sf_debug_rv2_check_2_2_1_0
L7 = ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
last = np.zeros(8, dtype=np.uint64)
rows = []
for (n, v, cap, reps, seed, ruin) in [(1000, 100, 55, 1024, 0, (2, 5, 10)), (1000, 100, 40, 512, 1, (2, 6, 10)), (200, 20, 55, 512, 2, (2, 5, 10)), (200, 8, 150, 256, 3, (1, 6, 16)),
                                      (400, 120, 20, 256, 4, (2, 5, 10)), (60, 6, 55, 256, 5, (2, 5, 10)), (300, 30, 1000, 256, 6, (3, 6, 8))]:
    p = datasets.make_cvrp(n, v, cap, seed=seed)
    d = sfa.build_cvrp(p, n_replicas=reps, leaves=L7, ruin=ruin)
    d.configure(sfa.SolverConfig(random_seed=seed))
    d.calculate_score(); d.phase_start()
    d.solve_steps(steps)
    d.sync()
    out = np.zeros(8, dtype=np.uint64)
    assert chk(out.ctypes.data_as(ctypes.c_void_p)) == 0
    delta = out[:3] - last[:3]
    rows.append({"customers": n, "vehicles": v, "capacity": cap, "replicas": reps, "ruin": ruin, "steps": steps, "candidates_checked": int(delta[0]),
                 "mismatches": int(delta[1]), "fallbacks_to_sf_ruin_h": int(delta[2]), "fresh_score_matches": bool((d.fresh_score() == d.calculate_score()).all())})
    if delta[1]:
        rows[-1]["first_mismatch"] = {"replica": int(out[3]), "candidate": int(out[4]), "count": int(out[5]), "v2_soft": int(np.int64(out[6])), "old_soft": int(np.int64(out[7]))}
    last = out.copy()
    d.close()
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"total_checked": int(last[0]), "total_mismatches": int(last[1]), "total_fallbacks": int(last[2])}))
