"""Job shop with the makespan objective (ListPrecedenceMakespanConstraint, list change + list swap leaves, LateAcceptance):
moves/s of the generic HIP engine (one full wave-wide Kahn evaluation per trial) vs the CPU oracle's incremental refresh over
the same step window of replica 0.  argv: jobs machines replicas steps_per_launch launches [leaves, comma separated; "policy" = the
critical-path leaf + permute + the rest of the list policy of a slot without a distance meter]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

J = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ls = int(sys.argv[4]) if len(sys.argv) > 4 else 5
K = int(sys.argv[5]) if len(sys.argv) > 5 else 2
leaves = sys.argv[6] if len(sys.argv) > 6 else "list_change,list_swap"
slot_policy = False
if leaves == "policy":
    leaves = "precedence,permute,list_change,list_swap,sublist_change,sublist_swap,list_reverse,kopt"
if leaves == "policy9":  # the complete default policy of a slot with precedence hooks: nine leaves + the slot's precedence policy
    leaves = "precedence,permute,list_change,list_swap,sublist_change,sublist_swap,list_reverse,kopt,ruin"
    slot_policy = True
leaves = tuple(leaves.split(","))
BITS = {"precedence": 16384, "permute": 8192, "list_change": 4, "list_swap": 8, "sublist_change": 128, "sublist_swap": 256, "list_reverse": 64, "kopt": 512, "ruin": 1024}
p = datasets.make_precedence_shop(J, M, seed=1)
d = sfa.build_precedence_shop(p, n_replicas=R, leaves=leaves, precedence_policy=slot_policy)
d.configure(sfa.SolverConfig(random_seed=0))
start = d.calculate_score()[0].tolist()
d.phase_start()
d.solve_steps(ls); d.profile_solve()
warm = d.calculate_score()[0].copy()
b = d.total_stats()
t0 = time.perf_counter()
for _ in range(K): d.solve_steps(ls, sync=False)
d.sync()
dt = time.perf_counter() - t0
ms, n = d.profile_solve()
a = d.total_stats()
moves = a["moves_evaluated"] - b["moves_evaluated"]
o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
o.configure(leaves=sum(BITS[x] for x in leaves), random_seed=0)
o.set_kopt(1, 0)
o.set_ruin()
o.set_precedence_policy(slot_policy)
o.phase_start(); o.steps(ls)
match_warm = bool((warm == o.score()[:2]).all())
m0 = o.stats()["moves_evaluated"]; t1 = time.perf_counter(); done = 0
while done < K * ls and time.perf_counter() - t1 < 20: o.steps(1); done += 1
ct = time.perf_counter() - t1
cm = o.stats()["moves_evaluated"] - m0
match = bool((d.calculate_score()[0] == o.score()[:2]).all()) if done == K * ls else None
print(json.dumps({"workload": "job shop %dx%d, ListPrecedenceMakespan, leaves %s" % (J, M, "+".join(leaves)), "nodes": J * M, "replicas": R,
                  "gpu_moves_per_s": moves / dt, "kernel_ms_per_launch": ms / n, "cpu_oracle_moves_per_s": cm / max(ct, 1e-9), "cpu_steps": done,
                  "replica0_matches_oracle": match, "replica0_matches_oracle_first_%d_steps" % ls: match_warm,
                  "gpu_over_cpu": (moves / dt) / max(cm / max(ct, 1e-9), 1e-9), "start_score": start, "score_replica0": d.calculate_score()[0].tolist(),
                  "best": d.best_scores().max(axis=0).tolist(),
                  "moves_per_step": (a["moves_evaluated"] - b["moves_evaluated"]) / max(a["step_count"] - b["step_count"], 1)}))
