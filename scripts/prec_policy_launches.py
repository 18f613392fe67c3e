"""A few launches of the nine-leaf default policy of a list slot with precedence hooks (default components: LateAcceptance(400) +
FirstLastStepScoreImproving(256)) on a job shop -- the command behind the rocprofv3 kernel-trace / --pmc passes of the precedence
instantiation (profiles/r04_prec_*).  usage: prec_policy_launches.py jobs machines replicas steps_per_launch launches"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
J, M, R, steps, K = (int(x) for x in sys.argv[1:6])
leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
p = datasets.make_precedence_shop(J, M, seed=1)
d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=True)
d.configure_default(random_seed=0)
d.calculate_score(); d.phase_start()
d.solve_steps(steps)  # warm-up launch
b = d.total_stats(); t0 = time.perf_counter()
for _ in range(K):
    d.solve_steps(steps)
dt = time.perf_counter() - t0
a = d.total_stats()
print(json.dumps({"workload": "job shop %dx%d nine-leaf policy, default components" % (J, M), "replicas": R, "steps_per_launch": steps, "launches": K,
                  "moves_per_launch": (a["moves_evaluated"] - b["moves_evaluated"]) / K, "ms_per_launch": dt / K * 1e3,
                  "moves_per_s": (a["moves_evaluated"] - b["moves_evaluated"]) / dt, "best": [int(v) for v in d.best_scores().max(axis=0)]}))
