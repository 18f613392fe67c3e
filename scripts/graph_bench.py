"""BASELINE config 2 (graph colouring 10k nodes / 100k edges / 16 colours, scalar change + swap):
moves/s of the HIP scalar path vs the CPU oracle over the same step window of replica 0."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3072  # (12 per CU since round 5)
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
# "la" = LateAcceptance(400)+AcceptedCount(256) (the list-policy components); "sa" = the reference's default
# for scalar-only models: auto-calibrated SimulatedAnnealing + AcceptedCount(1) (default_local_search/policy.rs:56-77)
policy = sys.argv[4] if len(sys.argv) > 4 else "la"
# fifth argument "empty": the all-unassigned start (the swap stream then scans thousands of equal (None, None) pairs per
# step, a property of the reference's stream); default = the post-construction start of SURVEY 8d (first-fit colouring)
start = sys.argv[5] if len(sys.argv) > 5 else "constructed"
g = datasets.make_graph(10000, 100000, 16, seed=0)
if start != "empty":
    g = datasets.construct_graph(g)
d = sfa.build_graph_coloring(g, n_replicas=R)
if policy == "sa":
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.SIMULATED_ANNEALING, accepted_count_limit=1, random_seed=0))
else:
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
moves = a["moves_evaluated"] - b["moves_evaluated"]; scored = a["candidates_scored"] - b["candidates_scored"]
o = sfo.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
o.configure(leaves=sfo.LEAF_SCALAR_CHANGE | sfo.LEAF_SCALAR_SWAP, random_seed=0, limit=1 if policy == "sa" else 256)
if policy == "sa":
    o.configure_annealing(seed=0)
o.phase_start(); o.steps(ls)
match_warm = bool((warm_score == o.score()[:2]).all())
m0 = o.stats()["moves_evaluated"]; t1 = time.perf_counter(); done = 0
while done < K * ls and time.perf_counter() - t1 < 10: o.steps(10); done += 10
ct = time.perf_counter() - t1
cm = o.stats()["moves_evaluated"] - m0
match = bool((d.calculate_score()[0] == o.score()[:2]).all()) if done == K * ls else None
# INDEXED CPU baseline (report both): the same search with the predicate join indexed by its partner relation; the
# whole timed window fits, so replica 0 is compared with it over all K * ls steps
oi = sfo.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"], indexed=True)
oi.configure(leaves=sfo.LEAF_SCALAR_CHANGE | sfo.LEAF_SCALAR_SWAP, random_seed=0, limit=1 if policy == "sa" else 256)
if policy == "sa":
    oi.configure_annealing(seed=0)
oi.phase_start(); oi.steps(ls)
mi0 = oi.stats()["moves_evaluated"]; t2 = time.perf_counter(); done_i = 0
while done_i < K * ls and time.perf_counter() - t2 < 20: oi.steps(10); done_i += 10
cti = time.perf_counter() - t2
cmi = oi.stats()["moves_evaluated"] - mi0
match_i = bool((d.calculate_score()[0] == oi.score()[:2]).all()) if done_i == K * ls else None
# change candidate 28 + 8*deg = 188 B, swap 36 + 8*(deg u + deg v) = 356 B at deg 20
alg = scored * (188 + 356) / 2
print(json.dumps({"workload": "graph colouring 10k/100k/16", "policy": policy, "start": start,
                  "gpu_steps_per_s": (a["step_count"] - b["step_count"]) / dt, "cpu_steps_per_s": done / ct, "replicas": R, "gpu_moves_per_s": moves / dt,
                  "gpu_candidates_scored_per_s": scored / dt, "kernel_ms_per_launch": ms / n,
                  "alg_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12,
                  "cpu_oracle_moves_per_s": cm / ct, "cpu_steps": done, "cpu_indexed_moves_per_s": cmi / cti, "cpu_indexed_steps": done_i,
                  "replica0_matches_indexed_cpu": match_i, "gpu_over_cpu_indexed": (moves / dt) / (cmi / cti), "replica0_matches_oracle": match, "replica0_matches_oracle_first_%d_steps" % ls: match_warm,
                  "gpu_over_cpu": (moves / dt) / (cm / ct), "score_replica0": d.calculate_score()[0].tolist(), "fill_calls_per_step": (a["sources_scanned"] - b["sources_scanned"]) / max(a["step_count"] - b["step_count"], 1), "moves_per_step": moves / max(a["step_count"] - b["step_count"], 1), "scored_per_step": scored / max(a["step_count"] - b["step_count"], 1)}))
