"""SURVEY §8(d) metric M1 -- sweep throughput: every candidate of one leaf's FULL neighbourhood scored on a
frozen state (BestScoreForager never quits early, forager.rs:418-420), selection order Original, median of 11
repeats, GPU (all replicas sweep the same frozen state) beside the CPU oracle on one host core."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPEATS = 11
BITS = {"nearby_change": 16, "nearby_swap": 32, "list_reverse": 64, "sublist_change": 128, "sublist_swap": 256, "kopt": 512, "ruin": 1024, "permute": 8192}
p = datasets.make_cvrp(1000, 100, 55, seed=0)
out = {"workload": "CVRP-1000 frozen start state, order Original, BestScore forager", "replicas": R, "leaves": {}}
# round 4: + the ruin leaf (10 candidates per sweep, each a full greedy recreate) and the list permute leaf
for leaf in ["nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin", "permute"]:
    d = sfa.build_cvrp(p, n_replicas=R, leaves=(leaf,))
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.HILL_CLIMBING, forager=sfa.Forager.BEST_SCORE,
                                 selection_order=sfa.SelectionOrder.ORIGINAL, random_seed=0))
    d.calculate_score()
    ms_list, cand = [], 0
    for rep in range(REPEATS + 1):  # every repeat restarts the phase from the committed state: one step = one sweep;
        d.phase_start()             # (the step commits its best move, so re-create the director for a frozen state)
        d.profile_solve()
        d.solve_steps(1)
        ms, n = d.profile_solve()
        st = d.total_stats()
        if rep:
            ms_list.append(ms)
        cand = st["moves_evaluated"]
        d = sfa.build_cvrp(p, n_replicas=R, leaves=(leaf,))
        d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.HILL_CLIMBING, forager=sfa.Forager.BEST_SCORE,
                                     selection_order=sfa.SelectionOrder.ORIGINAL, random_seed=0))
        d.calculate_score()
    gms = statistics.median(ms_list)
    o_t = []
    per = 0
    for rep in range(REPEATS if leaf not in ("sublist_swap", "permute") else 3):
        o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(acceptor=0, forager=2, limit=1, leaves=BITS[leaf], selection_order=0, random_seed=0)
        o.set_ruin()
        o.phase_start()
        t0 = time.perf_counter(); o.steps(1); o_t.append(time.perf_counter() - t0)
        per = o.stats()["moves_evaluated"]
    cs = statistics.median(o_t)
    assert cand == per * R, (leaf, cand, per)
    out["leaves"][leaf] = {"candidates_per_sweep": per, "gpu_ms_per_launch_median": gms,
                           "gpu_candidates_per_s": cand / (gms * 1e-3), "cpu_s_per_sweep_median": cs,
                           "cpu_candidates_per_s": per / cs, "gpu_over_cpu": (cand / (gms * 1e-3)) / (per / cs)}
# the critical-path precedence leaf on a job shop (its own model: the list class carries the precedence constraint)
shop = datasets.make_precedence_shop(20, 10, seed=1)
Rp = min(R, 1024)
def shop_director():
    dd = sfa.build_precedence_shop(shop, n_replicas=Rp, leaves=("precedence",), precedence_policy=True)
    dd.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.HILL_CLIMBING, forager=sfa.Forager.BEST_SCORE, selection_order=sfa.SelectionOrder.ORIGINAL, random_seed=0))
    dd.calculate_score()
    return dd
ms_list, cand = [], 0
for rep in range(4):
    d = shop_director()
    d.phase_start(); d.profile_solve(); d.solve_steps(1)
    ms, n = d.profile_solve()
    if rep:
        ms_list.append(ms)
    cand = d.total_stats()["moves_evaluated"]
o = sfo.Model.precedence_shop(shop["durations"], shop["successors"], shop["sequences"], shop["expected_owner"])
o.configure(acceptor=0, forager=2, limit=1, leaves=16384, selection_order=0, random_seed=0)
o.set_precedence_policy(True)
o.phase_start()
t0 = time.perf_counter(); o.steps(1); cs = time.perf_counter() - t0
per = o.stats()["moves_evaluated"]
assert cand == per * Rp, (cand, per)
gms = statistics.median(ms_list)
out["precedence_leaf_jobshop_20x10"] = {"replicas": Rp, "candidates_per_sweep": per, "gpu_ms_per_launch_median": gms, "gpu_candidates_per_s": cand / (gms * 1e-3),
                                        "cpu_s_per_sweep": cs, "cpu_candidates_per_s": per / cs, "gpu_over_cpu": (cand / (gms * 1e-3)) / (per / cs)}
print(json.dumps(out))
