"""Long-horizon parity: GPU replicas vs the CPU oracle after tens of thousands of local-search steps
(late-phase behaviour: many sources per step, multi-chunk neighbour walks, exhausted leaves)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
R = 8
out = {}
# (the oracle replays every step on one CPU core: late steps consume thousands of candidates each, so
# the horizon is bounded by the CPU side; results are printed per problem as they complete)
for name, n, v, cap in [("cvrp300_tight", 300, 40, 45), ("cvrp1000", 1000, 100, 55)]:
    p = datasets.make_cvrp(n, v, cap, seed=1)
    d = sfa.build_cvrp(p, n_replicas=R)
    d.configure(sfa.SolverConfig(random_seed=100))
    d.calculate_score(); d.phase_start()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        k = min(1000, steps - done); d.solve_steps(k); done += k
    gt = time.perf_counter() - t0
    sc = d.calculate_score()
    ok_fresh = bool((d.fresh_score() == sc).all())
    res = {"gpu_seconds": gt, "fresh_equals_incremental": ok_fresh, "engine": d.engine(), "replicas": {}}
    for r in ((0, 5) if n < 1000 else (3,)):
        o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(leaves=sfo.LEAF_NEARBY_LIST_CHANGE | sfo.LEAF_NEARBY_LIST_SWAP, random_seed=100 + r)
        o.phase_start()
        t1 = time.perf_counter(); o.steps(steps); ct = time.perf_counter() - t1
        res["replicas"][r] = {"score_match": bool((sc[r] == o.score()[:2]).all()),
                              "lists_match": d.working_lists(0, r) == o.get_lists(0),
                              "moves_match": d.stats(r)["moves_evaluated"] == o.stats()["moves_evaluated"],
                              "best_match": bool((d.best_scores()[r] == o.best_score()[:2]).all()),
                              "score": sc[r].tolist(), "cpu_seconds": ct,
                              "moves_per_step": o.stats()["moves_evaluated"] / steps}
    out[name] = res
    print(json.dumps({name: res}), flush=True)
