import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import solverforge_amd as sfa
from oracle import sfo
import test_gpu_ruin as T
p = T._problem("asym")
d, o = T._mk(sfo, p, ("ruin",), ruin=(3,3,4), seed=4, la_size=5, limit=8)
d.configure(sfa.SolverConfig(random_seed=4, late_acceptance_size=5, accepted_count_limit=8))
print(d.calculate_score()[0], o.score()[:2])
d.phase_start(); o.phase_start()
gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1<<18)
om, os_, of, oap, omv = o.step_traced()
for a,b,c,e in zip(gm, gs, om, os_): print(a, sfo.ruin_positions(a), b, e[:2])
print(p["routes"])
