"""Diagnostic: per-phase shader-clock shares of the scalar engine on graph colouring 10k/100k/16 (needs a
-DSF_PHASE_PROFILE build passed via SF_AMD_LIB).  Phases: 0 step start (seeds, permutations, load-balance aggregates),
1 ring fill (change / swap streams), 2 replay (trial deltas, acceptor, forager), 3 commit + step end.
Usage: phase_probe_scalar.py <replicas> <la|sa> <steps per launch>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
policy = sys.argv[2] if len(sys.argv) > 2 else "sa"
ls = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
g = datasets.make_graph(10000, 100000, 16, seed=0)
d = sfa.build_graph_coloring(g, n_replicas=R)
if policy == "sa":
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.SIMULATED_ANNEALING, accepted_count_limit=1, random_seed=0))
else:
    d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
L = _lib.load()
out = np.zeros(8, dtype=np.uint64)
for it in range(3):
    b = d.total_stats()
    d.solve_steps(ls)
    L.sf_debug_phases(out.ctypes.data_as(ctypes.c_void_p))
    ms, n = d.profile_solve()
    a = d.total_stats()
    tot = out.sum()
    steps = a["step_count"] - b["step_count"]
    print("launch", it, "ms %.1f" % ms, "Msteps/s %.1f" % (steps / ms / 1e3), "moves/step %.1f" % ((a["moves_evaluated"] - b["moves_evaluated"]) / steps),
          "scored/step %.1f" % ((a["candidates_scored"] - b["candidates_scored"]) / steps), "fill calls/step %.2f" % ((a["sources_scanned"] - b["sources_scanned"]) / steps),
          "cycles/step/wave %.0f" % (tot / steps), "shares %", np.round(out[:4] / tot * 100, 1))
