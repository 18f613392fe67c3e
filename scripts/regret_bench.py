#!/usr/bin/env python3
"""Wall-clock of the device construction phases on CVRP (one launch each, n_replicas replicas): regret insertion vs cheapest
insertion, with the quality (hard, soft) of the constructed state.  Usage: regret_bench.py [n_replicas]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import solverforge_amd as sfa  # noqa: E402
from solverforge_amd import datasets  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = []
for n, v, cap in ((200, 20, 55), (500, 50, 55), (1000, 100, 55)):
    p = datasets.make_cvrp(n, v, cap, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    row = {"customers": n, "vehicles": v, "n_replicas": R}
    for name in ("cheapest", "regret"):
        d = sfa.build_cvrp(p, n_replicas=R)
        d.calculate_score()
        t0 = time.perf_counter()
        sc = getattr(d, "construct_list_" + name)(0, p["customers"])
        dt = time.perf_counter() - t0
        st = d.stats(0)
        row[name] = {"seconds": round(dt, 4), "score": [int(x) for x in sc[0]], "trials_per_replica": int(st["score_calculations"]),
                     "trials_per_second_all_replicas": round(st["score_calculations"] * R / dt)}
        d.close()
    out.append(row)
    print(json.dumps(row), flush=True)
