import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SF_AMD_PREC_HBM"] = "1"
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo
p = datasets.make_precedence_shop(7, 4, seed=5)
leaves = ("list_change", "list_swap")
d = sfa.build_precedence_shop(p, leaves=leaves)
o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
bits = 4 | 8
print("start", d.calculate_score()[0], o.score()[:2])
print("lists", p["sequences"]); print("dur", list(p["durations"])); print("succ", p["successors"])
for order in (0,):
    o.configure(leaves=bits, selection_order=order)
    gm, gs, gd = d.open_cursor(2, 31, selection_order=order, cap=1 << 18)
    om = o.enumerate(0, 2, 31, order)
    os_, od = o.evaluate_moves(om)
    bad = np.flatnonzero((gs != os_[:, :2]).any(axis=1))
    print("n", len(om), "bad", len(bad))
    for i in bad[:25]:
        print(i, om[i], "gpu", gs[i], "oracle", os_[i, :2], "doable", gd[i], od[i])
