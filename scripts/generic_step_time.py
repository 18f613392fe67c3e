"""Wall time per launch of 100 local-search steps of the generic engine on CVRP-1000 (no instrumentation): early in a search and after `warm` steps.
usage: generic_step_time.py <replicas> <leaves,comma> [warm=1500] [ruin min,max,moves_per_step]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
R = int(sys.argv[1]); leaves = tuple(sys.argv[2].split(",")); warm = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
ruin = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (2, 5, 10)  # min count, max count, moves per step
d = sfa.build_cvrp(datasets.make_cvrp(1000, 100, 55, seed=0), n_replicas=R, leaves=leaves, ruin=ruin)
d.configure(sfa.SolverConfig(random_seed=0)); d.calculate_score(); d.phase_start()
out = {"replicas": R, "leaves": len(leaves), "ruin": ruin}
for tag, pre in (("early", 100), ("late", warm)):
    d.solve_steps(pre); d.profile_solve()
    b = d.total_stats()
    for _ in range(3): d.solve_steps(100)
    ms, n = d.profile_solve(); a = d.total_stats()
    mv = a["moves_evaluated"] - b["moves_evaluated"]
    out[tag] = {"ms_per_100_steps": round(ms / n, 2), "moves_per_step": round(mv / R / 300, 1), "G_moves_per_s": round(mv / ms / 1e6, 3)}
print(json.dumps(out))
