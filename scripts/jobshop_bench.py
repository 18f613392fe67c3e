"""BASELINE config 4 (mixed job shop 500 jobs x 20 machines, list + scalar moves, BendableScore<2,1>):
moves/s of the generic N-leaf HIP engine vs the CPU oracle over the same step window of replica 0."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
# start = constructed state (every operation assigned and scheduled); argv[4] = 'empty' keeps the all-unassigned start
p = datasets.make_jobshop(500, 20)
if not (len(sys.argv) > 4 and sys.argv[4] == 'empty'):
    p = datasets.construct_jobshop(p)
MK = len(sys.argv) > 5 and sys.argv[5] == 'makespan'  # add the ListPrecedenceMakespanConstraint (durations from the documented stream)
if MK:
    import numpy as np
    p["durations"] = (datasets.stream(5, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
d = sfa.build_jobshop(p, n_replicas=R, makespan=MK)
d.configure(sfa.SolverConfig(random_seed=0))
d.calculate_score(); d.phase_start()
d.solve_steps(ls); d.profile_solve()
warm_score = d.calculate_score()[0].copy()  # replica 0 after the first `ls` steps: compared with the oracle below (bounded window)
b = d.total_stats()
t0 = time.perf_counter()
for _ in range(K): d.solve_steps(ls, sync=False)
d.sync()
dt = time.perf_counter() - t0
ms, n = d.profile_solve()
a = d.total_stats()
moves = a["moves_evaluated"] - b["moves_evaluated"]
o = sfo.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, durations=p["durations"] if MK else None)
bits = sfo.LEAF_LIST_CHANGE | sfo.LEAF_LIST_SWAP | sfo.LEAF_SCALAR_CHANGE | sfo.LEAF_SCALAR_SWAP
o.configure(leaves=bits, random_seed=0)
o.phase_start(); o.steps(ls)
match_warm = bool((warm_score == o.score()[:3]).all())
m0 = o.stats()["moves_evaluated"]; t1 = time.perf_counter(); done = 0
while done < K * ls and time.perf_counter() - t1 < 20: o.steps(5); done += 5
ct = time.perf_counter() - t1
cm = o.stats()["moves_evaluated"] - m0
match = bool((d.calculate_score()[0] == o.score()[:3]).all()) if done == K * ls else None
# INDEXED CPU baseline (report both): the "same job, same machine" join indexed by job
oi = sfo.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, indexed=True, durations=p["durations"] if MK else None)
oi.configure(leaves=bits, random_seed=0)
oi.phase_start(); oi.steps(ls)
mi0 = oi.stats()["moves_evaluated"]; t2 = time.perf_counter(); done_i = 0
while done_i < K * ls and time.perf_counter() - t2 < 20: oi.steps(5); done_i += 5
cti = time.perf_counter() - t2
cmi = oi.stats()["moves_evaluated"] - mi0
match_i = bool((d.calculate_score()[0] == oi.score()[:3]).all()) if done_i == K * ls else None
print(json.dumps({"workload": "mixed job shop 500x20, Bendable<2,1>" + (" + makespan (ListPrecedenceMakespanConstraint)" if MK else ""), "replicas": R, "gpu_moves_per_s": moves / dt,
                  "kernel_ms_per_launch": ms / n, "cpu_oracle_moves_per_s": cm / ct, "cpu_steps": done, "cpu_indexed_moves_per_s": cmi / cti, "cpu_indexed_steps": done_i,
                  "replica0_matches_indexed_cpu": match_i, "gpu_over_cpu_indexed": (moves / dt) / (cmi / cti),
                  "replica0_matches_oracle": match, "replica0_matches_oracle_first_%d_steps" % ls: match_warm, "gpu_over_cpu": (moves / dt) / (cm / ct),
                  "score_replica0": d.calculate_score()[0].tolist(),
                  "per_step": {k: (a[k] - b[k]) / max(a["step_count"] - b["step_count"], 1)
                               for k in ("moves_evaluated", "candidates_scored", "sources_scanned", "moves_accepted", "moves_applied")}}))
