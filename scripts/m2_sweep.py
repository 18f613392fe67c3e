"""M2 knee: best HardSoftScore after `seconds` of wall time on CVRP-1000 from the savings + capacity start, GPU only, one JSON line per
configuration: replica count x elite migration (sf_portfolio_migrate_local: every `period` seconds the worst `frac` of the replicas adopt
the best solutions of the top `elite` replicas).  The CPU leg of the same workload is bench.py's / scripts/solve60.py's.
usage: m2_sweep.py seconds policy "R[:period:frac:elite[:la_size:accepted_count_limit]],R,..." [budget]
  policy = default (seven leaves), default6 (without ruin), nearby2, nearby2_ruin, nearby2_kopt_ruin"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solverforge_amd as sfa
from solverforge_amd import datasets

POL = {"default": ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin"),
       "default6": ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt"),
       "nearby2": ("nearby_change", "nearby_swap"),
       "nearby2_ruin": ("nearby_change", "nearby_swap", "ruin"),
       "nearby2_kopt_ruin": ("nearby_change", "nearby_swap", "kopt", "ruin")}
seconds = float(sys.argv[1])
policy = sys.argv[2]
configs = sys.argv[3].split(",")
budget = int(sys.argv[4]) if len(sys.argv) > 4 else 30_000
p = datasets.make_cvrp(1000, 100, 55, seed=0)
p["routes"] = [[] for _ in p["routes"]]
for cfg in configs:
    f = cfg.split(":")
    R = int(f[0])
    period = float(f[1]) if len(f) > 1 else 0.0
    frac = float(f[2]) if len(f) > 2 else 0.5
    elite = int(f[3]) if len(f) > 3 else max(1, R // 64)
    la = int(f[4]) if len(f) > 4 else 400
    limit = int(f[5]) if len(f) > 5 else 256
    d = sfa.build_cvrp(p, n_replicas=R, leaves=POL[policy])
    d.configure(sfa.SolverConfig(random_seed=0, late_acceptance_size=la, accepted_count_limit=limit))
    if os.environ.get("SF_M2_ENGINE") == "block":  # one workgroup per replica instead of one wavefront (per-chain latency experiment)
        d.set_engine(sfa.Engine.BLOCK)
    d.calculate_score()
    t0 = time.perf_counter()
    d.construct_list_clarke_wright(0, p["customers"], 1)
    start = [int(v) for v in d.construct_list_k_opt(0, 2, 1)[0]]
    d.phase_start()
    trace, migrations, adopted, next_m = [], 0, 0, period
    while time.perf_counter() - t0 < seconds:
        d.solve_moves(1 << 20, budget)
        now = time.perf_counter() - t0
        if period > 0 and now >= next_m and now < seconds - period * 0.5:
            adopted += d.migrate_local(elite, int(R * frac))
            migrations += 1
            next_m += period
        if not trace or now - trace[-1][0] >= 5.0:
            trace.append((round(now, 1), list(max(tuple(int(v) for v in s) for s in d.best_scores()))))
    gt = time.perf_counter() - t0
    st = d.total_stats()
    print(json.dumps({"policy": policy, "replicas": R, "late_acceptance_size": la, "accepted_count_limit": limit, "migration": {"period_s": period, "replace_fraction": frac, "elite": elite, "migrations": migrations,
                                                                    "adopted": adopted} if period > 0 else None,
                      "seconds": gt, "start_score": start, "best_score": list(max(tuple(int(v) for v in s) for s in d.best_scores())),
                      "moves_per_s": st["moves_evaluated"] / gt, "ls_steps_per_replica": st["step_count"] / R, "launch_move_budget": budget,
                      "trace": trace}), flush=True)
    d.close()
