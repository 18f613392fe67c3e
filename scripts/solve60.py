"""M2 of BASELINE.md: best HardSoftScore after `seconds` of wall time, GPU portfolio vs the CPU oracle
(1 host core), same CVRP-1000 instance and default list policy."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
replicas = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
# optional third argument: comma-separated leaves (default = the nearby change + nearby swap pair); e.g. the default
# list policy without ruin: nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
leaves = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("nearby_change", "nearby_swap")
BITS = {"nearby_change": 16, "nearby_swap": 32, "list_reverse": 64, "sublist_change": 128, "sublist_swap": 256, "kopt": 512,
        "list_change": 4, "list_swap": 8, "ruin": 1024}
# optional fifth argument: "cheapest" = start from the device's list cheapest-insertion construction (both sides) instead of
# the round-robin fill; "savings" = Clarke-Wright with the stock savings hooks (structural feasibility: the reference's default
# construction for the CVRP domain, defaults/stages.rs:257-266), "savings_capacity" = Clarke-Wright with the capacity test
start = sys.argv[5] if len(sys.argv) > 5 else "roundrobin"
# optional sixth / seventh argument: customers, vehicles
n_customers = int(sys.argv[6]) if len(sys.argv) > 6 else 1000
n_vehicles = int(sys.argv[7]) if len(sys.argv) > 7 else 100
p = datasets.make_cvrp(n_customers, n_vehicles, 55, seed=0)
if start != "roundrobin":
    p["routes"] = [[] for _ in p["routes"]]
d = sfa.build_cvrp(p, n_replicas=replicas, leaves=leaves)
d.configure(sfa.SolverConfig(random_seed=0))
start_score = [int(v) for v in d.calculate_score()[0]]
construct_s = 0.0
if start == "cheapest":
    tc = time.perf_counter()
    start_score = [int(v) for v in d.construct_list_cheapest(0, p["customers"])[0]]
    construct_s = time.perf_counter() - tc
elif start in ("savings", "savings_capacity"):
    tc = time.perf_counter()
    d.construct_list_clarke_wright(0, p["customers"], 1 if start == "savings_capacity" else 0)
    start_score = [int(v) for v in d.construct_list_k_opt(0, 2, 1)[0]]  # the default construction's second step
    construct_s = time.perf_counter() - tc
d.phase_start()
t0 = time.perf_counter(); trace = []
# optional fourth argument: candidates per replica per launch (sf_solve_moves: work-balanced launches); 0 = fixed
# step counts per launch (sf_solve_steps)
budget = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
while time.perf_counter() - t0 < seconds:
    if budget:
        d.solve_moves(1 << 20, budget)
    else:
        d.solve_steps(200 if len(leaves) == 2 else 50)
    if len(trace) % 25 == 0:
        trace.append((round(time.perf_counter() - t0, 1), list(max(tuple(int(v) for v in s) for s in d.best_scores()))))
    else:
        trace.append(None)
gt = time.perf_counter() - t0
st = d.total_stats()
gpu = {"seconds": gt, "replicas": replicas, "leaves": list(leaves), "best_score": list(max(tuple(int(v) for v in s) for s in d.best_scores())),
       "moves_evaluated": st["moves_evaluated"], "ls_steps_per_replica": st["step_count"] // replicas, "launch_move_budget": budget,
       "moves_per_s": st["moves_evaluated"] / gt, "start": start, "start_score": start_score, "construction_seconds": construct_s,
       "trace": [t for t in trace if t]}
o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
o.configure(leaves=sum(BITS[x] for x in leaves), max_nearby=20, random_seed=0)
o.set_ruin()
tc = time.perf_counter()
if start == "cheapest":
    o.construct_list_cheapest(p["customers"])
elif start in ("savings", "savings_capacity"):
    o.construct_list_clarke_wright(p["customers"], 1 if start == "savings_capacity" else 0)
    o.construct_list_k_opt(2, 1)
cpu_construct_s = time.perf_counter() - tc
cpu_start_score = [int(v) for v in o.score()[:2]]
o.phase_start()
t0 = time.perf_counter()
steps = o.steps_timed(seconds)
ct = time.perf_counter() - t0
cpu = {"seconds": ct, "best_score": [int(v) for v in o.best_score()[:2]], "ls_steps": int(steps),
       "moves_evaluated": o.stats()["moves_evaluated"], "moves_per_s": o.stats()["moves_evaluated"] / ct,
       "construction_seconds": cpu_construct_s, "start_score": cpu_start_score}
print(json.dumps({"gpu": gpu, "cpu_oracle_1core": cpu}))
