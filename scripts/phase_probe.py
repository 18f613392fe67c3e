"""Diagnostic: per-phase shader-clock shares of the wave engine (needs a -DSF_PHASE_PROFILE build
passed via SF_AMD_LIB).  Phases: 0 step start, 1 order tables+first resolve, 2 generation, 3 replay,
4 loop glue, 5 commit/best/LA, 7 inter-step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
p = datasets.make_cvrp(1000, 100, 55, seed=0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = sfa.build_cvrp(p, n_replicas=R)
d.set_engine(2)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
L = _lib.load()
out = np.zeros(8, dtype=np.uint64)
for it in range(4):
    d.solve_steps(200)
    L.sf_debug_phases_wave_2(out.ctypes.data_as(ctypes.c_void_p))
    ms, n = d.profile_solve()
    tot = out.sum()
    print("launch", it, "ms %.1f" % ms, "cycles/step/wave %.0f" % (tot / R / 200), "shares %", np.round(out / tot * 100, 1))
