"""Long-horizon parity of the seven-leaf default list policy (the FAST instantiation with the list-preserving ruin trial AND commit of
csrc/sf_ruin_v2.h): replicas of a fused GPU run vs the CPU oracle, lists / scores / best scores / counters after `steps` steps.
usage: deep_parity_ruin.py [steps=120]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
L7 = ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
BITS = 16 | 32 | 128 | 256 | 64 | 512 | 1024
for name, n, v, cap, seed in [("cvrp120", 120, 12, 55, 2), ("cvrp250_tight", 250, 30, 45, 3), ("cvrp60_few_lists", 60, 4, 200, 4)]:
    p = datasets.make_cvrp(n, v, cap, seed=seed)
    d = sfa.build_cvrp(p, n_replicas=6, leaves=L7)
    d.configure(sfa.SolverConfig(random_seed=40))
    d.calculate_score(); d.phase_start()
    done = 0
    while done < steps:
        k = min(50, steps - done); d.solve_steps(k); done += k
    sc = d.calculate_score()
    res = {"steps": steps, "fresh_equals_incremental": bool((d.fresh_score() == sc).all()), "replicas": {}}
    for r in (0, 4):
        o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(leaves=BITS, max_nearby=20, random_seed=40 + r)
        o.set_kopt(1, 20)
        o.set_ruin()
        o.phase_start()
        t1 = time.perf_counter(); o.steps(steps); ct = time.perf_counter() - t1
        so, sg = o.stats(), d.stats(r)
        res["replicas"][r] = {"score_match": bool((sc[r] == o.score()[:2]).all()), "lists_match": d.working_lists(0, r) == o.get_lists(0),
                              "best_match": bool((d.best_scores()[r] == o.best_score()[:2]).all()),
                              "counters_match": all(sg[k] == so[k] for k in ("moves_evaluated", "moves_accepted", "moves_applied", "score_calculations") if k in so and k in sg),
                              "score": sc[r].tolist(), "cpu_seconds": round(ct, 1)}
    print(json.dumps({name: res}), flush=True)
    d.close()
