"""Randomised differential run of the list construction phases: random CVRP instances (sizes, capacities from generous to
impossible, asymmetric / unreachable / tied legs, negative and over-capacity demands) x a random partial start state x a random
sequence of phases (Clarke-Wright in either feasibility mode, round robin with order keys / owner hook values, ListKOpt under a
sweep bound, cheapest insertion, regret insertion) on the GPU vs the CPU oracle: lists, committed scores and verdicts after every phase, then a few
local-search steps from the constructed state.  Prints one JSON line; `failures` lists the seeds whose runs diverged (none
expected).  Usage: fuzz_construction.py <seconds> [first_seed]"""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo


def run_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 5, 9, 17, 33, 64, 65, 90, 130, 200])) + int(rng.integers(0, 3))
    v = int(rng.integers(1, 14))
    cap = int(rng.choice([6, 12, 25, 40, 60, 90, 400]))
    p = datasets.make_cvrp(n, v, cap, seed=seed)
    kind = int(rng.integers(0, 5))
    if kind == 1:  # asymmetric + unreachable + negative legs
        r = datasets.stream(seed + 5, p["matrix"].size).reshape(p["matrix"].shape)
        p["matrix"] = (p["matrix"] + (r % np.uint64(23)).astype(np.int64)).astype(np.int64)
        np.fill_diagonal(p["matrix"], 0)
        if n > 10:
            p["matrix"][3, 7] = np.iinfo(np.int64).max
            p["matrix"][0, 4] = np.iinfo(np.int64).max
            p["matrix"][5, 2] = -1
    elif kind == 2:  # heavy ties
        p["matrix"] = (p["matrix"] // 400).astype(np.int64)
    if rng.random() < 0.25:  # negative demands: loads can shrink, confirming passes matter
        neg = rng.choice(np.arange(1, n + 1), max(1, n // 4), replace=False)
        p["demands"][neg] = -rng.integers(1, 9, len(neg)).astype(np.int32)
    if rng.random() < 0.15:  # a customer no vehicle can take
        p["demands"][int(rng.integers(1, n + 1))] = cap + int(rng.integers(1, 50))
    keep = int(rng.integers(0, v + 1)) if rng.random() < 0.5 else 0
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    desc = {"seed": seed, "n": n, "v": v, "cap": cap, "kind": kind, "keep": keep, "phases": []}
    R = int(rng.choice([1, 2, 3]))
    d = sfa.build_cvrp(p, n_replicas=R, max_nearby=10)
    o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.configure(sfa.SolverConfig(random_seed=seed, late_acceptance_size=8, accepted_count_limit=16))
    assert (d.calculate_score()[0] == o.score()[:2]).all(), "start score"

    def missing():
        placed = {c for rt in o.get_lists(0) for c in rt}
        return [i for i, c in enumerate(p["customers"]) if int(c) not in placed]

    for _ in range(int(rng.integers(1, 4))):
        phase = str(rng.choice(["cw0", "cw1", "cw1", "rr", "kopt", "kopt", "cheapest", "regret", "regret"]))
        desc["phases"].append(phase)
        miss = missing()
        if phase in ("cw0", "cw1"):
            mode = int(phase[-1])
            sc, flags = d.construct_list_clarke_wright(0, p["customers"], mode)
            committed, _ = o.construct_list_clarke_wright([int(p["customers"][i]) for i in miss], mode)
            assert all(bool(f) == committed for f in flags), f"{phase}: verdict"
        elif phase == "rr":
            ks = rng.integers(0, 5, n).astype(np.int64) if rng.random() < 0.5 else None
            ow = None
            if rng.random() < 0.5:
                ow = np.full(n, -1, dtype=np.int64)
                pick = rng.choice(n, max(1, n // 3), replace=False)
                ow[pick] = rng.integers(0, v + 2, len(pick))
            sc = d.construct_list_round_robin(0, p["customers"], ks, ow)
            o.construct_list_round_robin([int(p["customers"][i]) for i in miss], None if ks is None else ks[miss], None if ow is None else ow[miss])
        elif phase == "kopt":
            mode, sweeps = int(rng.integers(0, 2)), int(rng.choice([1, 3, 50]))
            sc = d.construct_list_k_opt(0, 2, mode, sweeps)
            o.construct_list_k_opt(2, mode, sweeps)
        elif phase == "regret":
            if len(miss) > 40:  # the oracle prices every slot of every unassigned element every round
                desc["phases"][-1] = "regret-skipped"
                continue
            ks = rng.integers(0, 4, n).astype(np.int64) if rng.random() < 0.5 else None
            ow = None
            if rng.random() < 0.4:
                ow = np.full(n, -1, dtype=np.int64)
                pick = rng.choice(n, max(1, n // 3), replace=False)
                ow[pick] = rng.integers(0, v + 2, len(pick))
            sc = d.construct_list_regret(0, p["customers"], ks, ow)
            o.construct_list_regret([int(p["customers"][i]) for i in miss], None if ks is None else ks[miss], None if ow is None else ow[miss])
        else:
            if len(miss) > 70:  # the oracle's cheapest insertion is cubic
                desc["phases"][-1] = "cheapest-skipped"
                continue
            sc = d.construct_list_cheapest(0, p["customers"])
            o.construct_list_cheapest([int(p["customers"][i]) for i in miss])
        for r in range(R):
            assert d.working_lists(0, r) == o.get_lists(0), f"{phase}: lists of replica {r}"
            assert (sc[r] == o.score()[:2]).all(), f"{phase}: committed score"
        assert (d.fresh_score()[0] == o.score()[:2]).all(), f"{phase}: fresh score"
    if sum(len(rt) for rt in o.get_lists(0)) >= 2:
        o.configure(leaves=sfo.LEAF_NEARBY_LIST_CHANGE | sfo.LEAF_NEARBY_LIST_SWAP, random_seed=seed, la_size=8, limit=16, max_nearby=10)
        d.phase_start(); o.phase_start()
        d.solve_steps(12); o.steps(12)
        assert d.working_lists(0, 0) == o.get_lists(0), "local search after construction"
        assert (d.calculate_score()[0] == o.score()[:2]).all(), "local search score"
    d.close()
    return desc


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); ran = 0; failures = []; by_phase = {}
while time.time() - t0 < budget:
    try:
        desc = run_case(seed)
        for ph in desc["phases"]:
            by_phase[ph] = by_phase.get(ph, 0) + 1
    except sfa.SolverForgeError as e:
        if "SF_ERR_UNSUPPORTED" not in str(e):
            failures.append({"seed": seed, "error": str(e)[:300]})
        else:
            by_phase["unsupported"] = by_phase.get("unsupported", 0) + 1
    except AssertionError as e:
        failures.append({"seed": seed, "error": str(e)[:300]})
    except Exception:
        failures.append({"seed": seed, "error": traceback.format_exc()[-400:]})
    ran += 1; seed += 1
print(json.dumps({"cases": ran, "phases": by_phase, "failures": failures[:20], "n_failures": len(failures)}))
