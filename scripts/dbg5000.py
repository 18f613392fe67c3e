import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo
n, v = int(sys.argv[1]), int(sys.argv[2])
p = datasets.make_cvrp(n, v, 55, seed=0)
for eng in (1, 2):
    d = sfa.build_cvrp(p, n_replicas=1); d.set_engine(eng)
    o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = sfo.LEAF_NEARBY_LIST_CHANGE | sfo.LEAF_NEARBY_LIST_SWAP
    o.configure(leaves=bits, random_seed=0)
    d.configure(sfa.SolverConfig(random_seed=0)); d.calculate_score()
    try:
        gm, gs, gd = d.open_cursor(0, 12345, selection_order=3, cap=1 << 19)
    except Exception as ex:
        print("engine", eng, "error", ex); continue
    om = o.enumerate(0, 0, 12345, 3)
    T = lambda m: np.stack([m["kind"], m["a"], m["a_pos"], m["b"], m["b_pos"]], axis=1)
    g, oo = T(gm), T(om)
    k = min(len(g), len(oo))
    bad = np.flatnonzero((g[:k] != oo[:k]).any(axis=1))
    print("engine", eng, "len", len(g), len(oo), "first mismatch", bad[:5])
    if len(bad):
        i = bad[0]
        print(g[max(0,i-2):i+3]); print(oo[max(0,i-2):i+3])
