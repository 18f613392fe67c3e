"""Diagnostic: per-phase shader-clock shares of the generic engine on a job shop under the nine-leaf default policy of a slot with
precedence hooks, default components (needs a -DSF_PHASE_PROFILE build passed via SF_AMD_LIB).  Phases: 0 step start + the leaf's
analysis, 1 fill: precedence leaf / permute / list change / list swap, 2 fill sublist leaves, 3 fill reverse + ruin, 4 fill 3-opt,
5 scheduler layout, 6 trial score + acceptor + forager, 7 commit.   usage: phase_probe_prec.py jobs machines replicas [steps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
J, M, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
p = datasets.make_precedence_shop(J, M, seed=1)
d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=True)
d.configure_default(random_seed=0)
d.calculate_score(); d.phase_start()
L = _lib.load()
phases = getattr(L, "sf_debug_phases_mixed_2_2_1_1")  # a -DSF_PHASE_PROFILE -DSF_PHASE_PREC build re-labels the slots: 0 rest, 1 critical-path leaf, 2 permute + change + swap, 3 reverse, 4 recreate: forward evaluation, 5 tails + closure, 6 slot pricing, 7 list edits + the ruin leaf's bookkeeping
out = np.zeros(8, dtype=np.uint64)
for it in range(3):
    b = d.total_stats()
    d.solve_steps(steps)
    phases(out.ctypes.data_as(ctypes.c_void_p))
    ms, n = d.profile_solve()
    a = d.total_stats()
    tot = max(int(out.sum()), 1)
    print("launch", it, "ms %.1f" % ms, "Mmoves/s %.2f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / ms / 1e3),
          "moves/step %.0f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / R / steps),
          "fill calls/step %.0f" % ((a["sources_scanned"] - b["sources_scanned"]) / R / steps),
          "cycles/step/wave %.0f" % (tot / R / steps), "shares %", np.round(out / tot * 100, 1))
