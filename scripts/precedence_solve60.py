"""Best score after `seconds` of wall time on a job shop with the makespan objective (ListPrecedenceMakespanConstraint, list change +
swap + reverse + sublist change, LateAcceptance(400) + AcceptedCount(256)): GPU portfolio (work-balanced launches) vs the CPU oracle's
incremental refresh on one host core, same instance, same start: the feasible step-major schedule (default), or with argv[5] =
"shuffled" every machine sequence permuted (cyclic, hard = -node_count: the all-or-nothing cycle penalty is a plateau that neither
side leaves in 60 s at 200+ nodes -- recorded in profiles/r02e_prec_solve60_shuffled_*.json).
argv: seconds jobs machines replicas [start] [policy]: "policy" = the reference's complete default list policy of a slot with precedence
hooks and no distance meter (critical-path leaf, permute, change, swap, sublist change / swap, reverse, full 3-opt, ruin; the slot's
precedence policy on) instead of the four leaves above, under the reference's default components for such a slot: LateAcceptance(400) +
FirstLastStepScoreImproving(256) (default_local_search/policy.rs:62-71; argv[7] = "accepted_count" keeps round 3's AcceptedCount(256))"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
J = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(sys.argv[3]) if len(sys.argv) > 3 else 10
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
leaves = ("list_change", "list_swap", "sublist_change", "list_reverse")
policy = len(sys.argv) > 6 and sys.argv[6] == "policy"
if policy:
    leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
BITS = {"precedence": 16384, "permute": 8192, "list_change": 4, "list_swap": 8, "sublist_change": 128, "sublist_swap": 256, "list_reverse": 64, "kopt": 512,
        "ruin": 1024}
p = datasets.make_precedence_shop(J, M, seed=1)
feasible = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"]).score()[:2].tolist()
start_kind = sys.argv[5] if len(sys.argv) > 5 else "feasible"
if start_kind == "shuffled":
    rng = np.random.default_rng(7)
    p["sequences"] = [[int(x) for x in rng.permutation(s)] for s in p["sequences"]]  # a cyclic start
d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=policy)
accepted_count = len(sys.argv) > 7 and sys.argv[7] == "accepted_count"
if policy and not accepted_count:
    cfg = d.configure_default(random_seed=0)  # FirstLastStepScoreImproving(256)
else:
    cfg = sfa.SolverConfig(random_seed=0)
    d.configure(cfg)
start = [int(v) for v in d.calculate_score()[0]]
d.phase_start()
t0 = time.perf_counter(); trace = []; k = 0
while time.perf_counter() - t0 < seconds:
    d.solve_moves(1 << 20, 20_000)
    if k % 10 == 0:
        trace.append((round(time.perf_counter() - t0, 1), list(max(tuple(int(v) for v in s) for s in d.best_scores()))))
    k += 1
gt = time.perf_counter() - t0
st = d.total_stats()
gpu = {"seconds": gt, "replicas": R, "best_score": list(max(tuple(int(v) for v in s) for s in d.best_scores())),
       "moves_evaluated": st["moves_evaluated"], "moves_per_s": st["moves_evaluated"] / gt, "ls_steps_per_replica": st["step_count"] // R,
       "trace": trace}
o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
o.configure(leaves=sum(BITS[x] for x in leaves), random_seed=0, la_size=cfg.late_acceptance_size, forager=cfg.forager, limit=cfg.accepted_count_limit)
o.set_kopt(1, 0)
if policy:
    o.set_ruin()
    o.set_precedence_policy(True)
o.phase_start()
t0 = time.perf_counter()
steps = o.steps_timed(seconds)
ct = time.perf_counter() - t0
cpu = {"seconds": ct, "best_score": [int(v) for v in o.best_score()[:2]], "ls_steps": int(steps), "moves_evaluated": o.stats()["moves_evaluated"],
       "moves_per_s": o.stats()["moves_evaluated"] / ct}
print(json.dumps({"workload": "job shop %dx%d, ListPrecedenceMakespan, %s start" % (J, M, start_kind), "leaves": list(leaves),
                  "components": {"acceptor": cfg.acceptor, "late_acceptance_size": cfg.late_acceptance_size, "forager": cfg.forager,
                                 "accepted_count_limit": cfg.accepted_count_limit}, "start_score": start,
                  "step_major_schedule_score": feasible, "gpu": gpu, "cpu_oracle_1core": cpu}))
