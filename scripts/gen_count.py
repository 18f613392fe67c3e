"""Event counts of the wave engine's candidate generation on bench.py's M1 workload (CVRP-1000 / 100, round-robin start, 200-step launches): needs a
-DSF_GEN_COUNT library passed via SF_AMD_LIB (scripts/build_variant.sh gencount list_wave_2 "-DSF_GEN_COUNT").
usage: gen_count.py [replicas=6144] [launches=4]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets, _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = datasets.make_cvrp(1000, 100, 55, seed=0)
d = sfa.build_cvrp(p, n_replicas=R)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
L = _lib.load()
out = np.zeros(8, dtype=np.uint64)
for it in range(3):
    d.solve_steps(200)
L.sf_debug_phases_wave_2(out.ctypes.data_as(ctypes.c_void_p))
b = d.total_stats()
for it in range(n):
    d.solve_steps(200)
L.sf_debug_phases_wave_2(out.ctypes.data_as(ctypes.c_void_p))
a = d.total_stats()
src = a["sources_scanned"] - b["sources_scanned"]
names = ["paired_passes", "single_leaf0", "single_leaf1", "rest_calls_pair_leaf0", "rest_calls_pair_leaf1", "rest_iters_leaf0", "rest_iters_leaf1", "rest_iters_empty"]
o = {k: int(v) for k, v in zip(names, out)}
o["sources"] = int(src); o["steps"] = int(a["step_count"] - b["step_count"]); o["moves"] = int(a["moves_evaluated"] - b["moves_evaluated"])
o["per_source"] = {k: round(int(v) / max(src, 1), 4) for k, v in zip(names, out)}
print(json.dumps(o))
