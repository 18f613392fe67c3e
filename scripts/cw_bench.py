"""Clarke-Wright construction on the device: wall time per call and the constructed score at C3 / C5 size, the oracle beside it
(bounded: the oracle runs once per size).  python scripts/cw_bench.py [--replicas R]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--sizes", default="1000x100,5000x500")
    ap.add_argument("--oracle", type=int, default=1)
    a = ap.parse_args()
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    for sz in a.sizes.split(","):
        n, v = (int(t) for t in sz.split("x"))
        p = datasets.make_cvrp(n, v, 55, seed=0)
        start = [list(rt) for rt in p["routes"]]
        p["routes"] = [[] for _ in p["routes"]]
        for mode in (1, 0):
            d = sfa.build_cvrp(p, n_replicas=a.replicas)
            d.calculate_score()
            t0 = time.perf_counter()
            sc, flags = d.construct_list_clarke_wright(0, p["customers"], mode)
            t1 = time.perf_counter()
            out = {"customers": n, "vehicles": v, "replicas": a.replicas, "feasible_mode": mode, "seconds": t1 - t0,
                   "savings_entries": n * (n - 1) // 2, "score": sc[0].tolist(), "committed": bool(flags.all()),
                   "routes": sum(1 for rt in d.working_lists(0, 0) if rt)}
            if a.oracle and mode == 1:
                from oracle import sfo

                o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
                t0 = time.perf_counter()
                o.construct_list_clarke_wright(p["customers"], mode)
                out["oracle_seconds"] = time.perf_counter() - t0
                out["matches_oracle"] = d.working_lists(0, 0) == o.get_lists(0)
                o2 = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], start)
                out["round_robin_start_score"] = o2.score()[:2].tolist()
            print(json.dumps(out), flush=True)
            d.close() if hasattr(d, "close") else None


if __name__ == "__main__":
    main()
