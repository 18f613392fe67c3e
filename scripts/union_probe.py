"""CVRP-1000 with the 3-leaf union nearby change + nearby swap + list reverse (generic N-leaf engine)
vs the CPU oracle on the same step window of replica 0."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
leaves = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("nearby_change", "nearby_swap", "list_reverse")
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
p = datasets.make_cvrp(N, N // 10, 55, seed=0)
d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
d.solve_steps(ls); d.profile_solve()
b = d.total_stats()
t0 = time.perf_counter()
for _ in range(K): d.solve_steps(ls, sync=False)
d.sync(); dt = time.perf_counter() - t0
ms, n = d.profile_solve(); a = d.total_stats()
moves = a["moves_evaluated"] - b["moves_evaluated"]
o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
BITS = {"nearby_change": 16, "nearby_swap": 32, "list_reverse": 64, "sublist_change": 128, "sublist_swap": 256, "list_change": 4, "list_swap": 8, "kopt": 512, "ruin": 1024}
o.configure(leaves=sum(BITS[x] for x in leaves), random_seed=0)
o.set_ruin()
o.phase_start(); o.steps(ls)
m0 = o.stats()["moves_evaluated"]; t1 = time.perf_counter(); done = 0
while done < K * ls and time.perf_counter() - t1 < 20: o.steps(20); done += 20
ct = time.perf_counter() - t1
cm = o.stats()["moves_evaluated"] - m0
match = bool((d.calculate_score()[0] == o.score()[:2]).all()) if done == K * ls else None
print(json.dumps({"workload": f"CVRP-{N} union " + "+".join(leaves), "replicas": R, "gpu_moves_per_s": moves / dt,
                  "kernel_ms_per_launch": ms / n, "cpu_oracle_moves_per_s": cm / ct, "replica0_matches_oracle": match,
                  "gpu_over_cpu": (moves / dt) / (cm / ct), "best": list(max(tuple(int(v) for v in s) for s in d.best_scores()))}))
