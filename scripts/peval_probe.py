"""Diagnostic: where one wave-wide precedence evaluation (sf_precedence.h, prec_eval) spends its shader clocks -- needs a
-DSF_PHASE_PROFILE -DSF_PHASE_PEVAL build of the PREC unit passed via SF_AMD_LIB.   usage: peval_probe.py jobs machines replicas [steps] [leaves]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
J, M, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
leaves = tuple((sys.argv[5] if len(sys.argv) > 5 else "list_change,list_swap,sublist_change,list_reverse").split(","))
p = datasets.make_precedence_shop(J, M, seed=1)
d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=False)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
L = _lib.load()
phases = getattr(L, "sf_debug_ruin2_phases_mixed_2_2_0_1")
out = np.zeros(8, dtype=np.uint64)
d.solve_steps(steps); phases(out.ctypes.data_as(ctypes.c_void_p))
for it in range(2):
    b = d.total_stats()
    d.solve_steps(steps)
    phases(out.ctypes.data_as(ctypes.c_void_p))
    ms, n = d.profile_solve(); a = d.total_stats()
    ev = max(int(out[0]), 1)
    print("launch", it, "ms %.1f" % ms, "moves %d" % (a["moves_evaluated"] - b["moves_evaluated"]), "evaluations %d" % ev, "nodes %d" % (int(out[6]) // ev),
          "clocks per evaluation: init %.0f, list pass %.0f, ready scan %.0f, rounds %.0f (%.1f rounds, %.0f clocks each)" % (
              out[1] / ev, out[2] / ev, out[3] / ev, out[4] / ev, out[5] / ev, out[4] / max(int(out[5]), 1)))
