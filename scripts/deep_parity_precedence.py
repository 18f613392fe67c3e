"""Long-horizon parity of the precedence engine: GPU replicas vs the CPU oracle after a fused run of the nine-leaf default policy of a list
slot with precedence hooks under the reference's default components (LateAcceptance(400) + FirstLastStepScoreImproving(256)), the grouped
trial evaluator at its default width.  One JSON line per shop.  usage: deep_parity_precedence.py ["jobs:machines:steps,..."]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

cases = sys.argv[1] if len(sys.argv) > 1 else "10:5:4000,20:10:160,30:10:50"
leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
BITS = {"precedence": 16384, "permute": 8192, "list_change": 4, "list_swap": 8, "sublist_change": 128, "sublist_swap": 256, "list_reverse": 64, "kopt": 512, "ruin": 1024}
R = 8
for case in cases.split(","):
    J, M, steps = (int(x) for x in case.split(":"))
    p = datasets.make_precedence_shop(J, M, seed=1)
    d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=True)
    cfg = d.configure_default(random_seed=100)
    d.calculate_score(); d.phase_start()
    t0 = time.perf_counter(); done = 0
    while done < steps:
        k = min(500, steps - done); d.solve_steps(k); done += k
    gt = time.perf_counter() - t0
    sc = d.calculate_score()
    res = {"shop": "%dx%d" % (J, M), "steps": steps, "gpu_seconds": gt, "fresh_equals_incremental": bool((d.fresh_score() == sc).all()), "replicas": {}}
    for r in (0, 5):
        o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
        o.configure(leaves=sum(BITS[x] for x in leaves), random_seed=100 + r, la_size=cfg.late_acceptance_size, forager=cfg.forager, limit=cfg.accepted_count_limit)
        o.set_kopt(1, 0); o.set_ruin(); o.set_precedence_policy(True)
        o.phase_start()
        t1 = time.perf_counter(); o.steps(steps); ct = time.perf_counter() - t1
        gs, os_ = d.stats(r), o.stats()
        res["replicas"][r] = {"score_match": bool((sc[r] == o.score()[:2]).all()), "lists_match": d.working_lists(0, r) == o.get_lists(0),
                              "counters_match": all(gs[k] == os_[k] for k in ("step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations")),
                              "best_match": bool((d.best_scores()[r] == o.best_score()[:2]).all()), "score": sc[r].tolist(), "cpu_seconds": ct,
                              "moves_per_step": os_["moves_evaluated"] / steps}
    print(json.dumps(res), flush=True)
