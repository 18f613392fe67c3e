"""GPU-only rate of BASELINE config 4 with the makespan objective (mixed job shop 500 x 20 + ListPrecedenceMakespanConstraint, 10,000 nodes, Kahn scratch
in HBM): moves/s over K launches of `ls` steps.  usage: c4_makespan_rate.py <replicas> [ls=5] [K=2]   (parity of this model: tests/test_gpu_mixed.py,
scripts/jobshop_bench.py ... makespan)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
R = int(sys.argv[1]); ls = int(sys.argv[2]) if len(sys.argv) > 2 else 5; K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = datasets.construct_jobshop(datasets.make_jobshop(500, 20))
p["durations"] = (datasets.stream(5, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
d = sfa.build_jobshop(p, n_replicas=R, makespan=True)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
d.solve_steps(ls); d.profile_solve()
b = d.total_stats(); t0 = time.perf_counter()
for _ in range(K): d.solve_steps(ls, sync=False)
d.sync(); dt = time.perf_counter() - t0
ms, n = d.profile_solve(); a = d.total_stats()
print(json.dumps({"replicas": R, "gpu_moves_per_s": (a["moves_evaluated"] - b["moves_evaluated"]) / dt, "kernel_ms_per_launch": ms / n, "ls_steps": ls,
                  "moves_per_step": (a["moves_evaluated"] - b["moves_evaluated"]) / max(a["step_count"] - b["step_count"], 1),
                  "fresh_equals_incremental": bool((d.fresh_score() == d.calculate_score()).all())}))
