"""Randomised differential run: random small instances x random policy (leaves, acceptor, forager, limits,
selection order, engine) -> traced steps + fused steps on the GPU vs the CPU oracle.  Prints one JSON line;
`failures` lists the seeds whose runs diverged (none expected).  Usage: fuzz_parity.py <seconds> [first_seed]
SF_FUZZ_MODEL=<family> draws every case from one model family (e.g. precedence)."""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solverforge_amd as sfa
from solverforge_amd import datasets
from oracle import sfo

BITS = {"nearby_change": 16, "nearby_swap": 32, "list_change": 4, "list_swap": 8, "list_reverse": 64,
        "sublist_change": 128, "sublist_swap": 256, "kopt": 512, "ruin": 1024, "change": 1, "swap": 2, "permute": 8192, "precedence": 16384}


def t6(m):
    return np.stack([m["kind"], m["a"], m["a_pos"], m["b"], m["b_pos"], m["value"]], axis=1)


def uni_matrix(terms, weight, scale, cols, table, n, k):
    """numpy twin of the library's host compile of one uni program (csrc/sf_api_scalar.inc: compile_uni_programs): cost[a][v] = scale * max(0, w) where the CNF holds"""
    a = np.arange(n, dtype=np.int64)[:, None]
    v = np.arange(k, dtype=np.int64)[None, :]

    def val(lhs, f, fb, fc):
        if lhs == 0:
            return np.ones((n, k), dtype=np.int64)
        if lhs == 1:
            return cols[f].astype(np.int64)[a] + 0 * v
        if lhs == 2:
            return v + 0 * a
        if lhs == 3:
            return cols[f].astype(np.int64)[v] + 0 * a
        if lhs in (4, 5):
            x = cols[f].astype(np.int64)[a] - cols[fb].astype(np.int64)[v]
            return np.abs(x) if lhs == 5 else x
        rk = cols[f].astype(np.int64)[a] + 0 * v if f >= 0 else a + 0 * v
        ck = cols[fb].astype(np.int64)[v] + 0 * a if fb >= 0 else v + 0 * a
        return table[rk, ck]

    ok = np.ones((n, k), dtype=bool)
    clause, acc = None, None
    for (lhs, cmp_, cl, f, fb, fc, param) in terms:
        x = val(lhs, f, fb, fc)
        h = [x == param, x != param, x < param, x <= param, x > param, x >= param][cmp_]
        if cl != clause:
            if acc is not None:
                ok &= acc
            clause, acc = cl, np.zeros((n, k), dtype=bool)
        acc |= h
    if acc is not None:
        ok &= acc
    w = val(*weight)
    return np.where(ok, scale * np.maximum(w, 0), 0).astype(np.int64)


def run_case(seed):
    rng = np.random.default_rng(seed)
    model = ["cvrp", "cvrp", "cvrp", "graph", "jobshop", "balance", "assignment", "precedence", "precedence", "shift", "shift", "shift"][int(rng.integers(12))]
    model = os.environ.get("SF_FUZZ_MODEL", model)  # a targeted run: every case of one model family
    acceptor = int(rng.choice([0, 1, 1, 3, 4]))  # 4 = DiversifiedLateAcceptance (round 3)
    dla_tol = float(rng.choice([0.0, 0.01, 0.2]))
    forager = int(rng.choice([0, 0, 1, 2, 3, 4]))
    limit = int(rng.choice([1, 2, 7, 40, 256]))
    order = int(rng.choice([0, 3, 3, 4]))
    la = int(rng.choice([1, 3, 50]))
    levels = 2
    desc = {"seed": seed, "model": model, "acceptor": acceptor, "forager": forager, "limit": limit, "order": order}
    if model == "cvrp":
        n = int(rng.integers(8, 70)); v = int(rng.integers(1, 9)); cap = int(rng.integers(10, 80))
        p = datasets.make_cvrp(n, v, cap, seed=seed)
        if rng.random() < 0.4:  # asymmetric + unreachable legs + ties
            r = datasets.stream(seed + 5, p["matrix"].size).reshape(p["matrix"].shape)
            p["matrix"] = (p["matrix"] // int(rng.choice([1, 1, 50])) + (r % np.uint64(int(rng.choice([1, 5])))).astype(np.int64)).astype(np.int64)
            np.fill_diagonal(p["matrix"], 0)
            if n > 10:
                p["matrix"][3, 7] = np.iinfo(np.int64).max
                p["matrix"][5, 2] = -1
        if rng.random() < 0.3 and v > 2:  # empty and one-element routes
            moved = p["routes"][1]
            p["routes"][1] = []
            p["routes"][0] = p["routes"][0] + moved
            if len(p["routes"][2]) > 1:
                p["routes"][0] = p["routes"][0] + p["routes"][2][1:]
                p["routes"][2] = p["routes"][2][:1]
        pool = ["nearby_change", "nearby_swap", "list_change", "list_swap", "list_reverse", "sublist_change", "sublist_swap", "kopt", "ruin", "permute"]
        chosen = set(rng.choice(pool, size=int(rng.integers(1, 8)), replace=False).tolist())
        leaves = tuple(x for x in ["permute", "nearby_change", "list_change", "nearby_swap", "list_swap", "sublist_change", "sublist_swap", "list_reverse",
                                   "kopt", "ruin"] if x in chosen)  # union (declaration) order
        pw = (int(rng.choice([2, 2, 3])), int(rng.choice([3, 4, 5])))
        mn = int(rng.choice([1, 3, 20, 64]))
        kopt = (int(rng.choice([1, 1, 2])), int(rng.choice([0, 2, 20])))
        sub = (1, int(rng.choice([1, 3, 5])))
        engine = int(rng.choice([0, 1, 2])) if set(leaves) <= {"nearby_change", "nearby_swap"} else 0
        ruin = (int(rng.choice([1, 2])), int(rng.choice([2, 5, 6])), int(rng.choice([1, 3, 10, 16])))
        desc.update(n=n, v=v, leaves=leaves, max_nearby=mn, kopt=kopt, sublist=sub, engine=engine, ruin=ruin)
        d = sfa.build_cvrp(p, leaves=leaves, max_nearby=mn, kopt=kopt, sublist_sizes=sub, ruin=ruin, permute=pw)
        if engine:
            d.set_engine(engine)
        o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.set_kopt(*kopt); o.set_sublist_sizes(*sub); o.set_permute(*pw)
        lists = lambda: (d.working_lists(0, 0), o.get_lists(0))
        post_configure = lambda: o.set_ruin(*ruin)
    elif model == "graph":
        n = int(rng.integers(5, 200)); e = int(rng.integers(n, 4 * n)); k = int(rng.integers(2, 9))
        g = datasets.make_graph(n, min(e, n * (n - 1) // 2), k, seed=seed)
        if rng.random() < 0.7:
            g["colors"] = (datasets.stream(seed + 9, n) % np.uint64(k + 1)).astype(np.int64) - 1
        leaves = [("change",), ("swap",), ("change", "swap")][int(rng.integers(3))]
        desc.update(n=n, e=e, k=k, leaves=leaves)
        d = sfa.build_graph_coloring(g, leaves=leaves)
        o = sfo.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        lists = lambda: (d.working_values(0, 0).tolist(), o.get_vars(0, 0).tolist())
    elif model == "balance":  # keyed self-join + grouped sum / excess-over-cap / load_balance collector
        n = int(rng.integers(4, 120)); k = int(rng.integers(2, 12)); cap = int(rng.choice([-1, 25, -2, -2, -3, -3]))
        r = datasets.stream(seed + 3, 2 * n)
        bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
        sizes = (r[n:] % np.uint64(int(rng.choice([9, 1000, 1_000_000])))).astype(np.int64) + 1
        wp = int(rng.choice([0, 3]))
        leaves = [("change",), ("swap",), ("change", "swap")][int(rng.integers(3))]
        desc.update(n=n, k=k, cap=cap, leaves=leaves)
        d = sfa.build_balance(bins, sizes, k, w_pair=wp, cap=cap, leaves=leaves)
        o = sfo.Model.balance(k, bins, sizes, w_pair=wp, cap=cap)
        lists = lambda: (d.working_values(0, 0).tolist(), o.get_vars(0, 0).tolist())
    elif model == "assignment":  # keyed cross-join with a fact class + exists / not-exists per fact row
        n = int(rng.integers(4, 100)); k = int(rng.integers(2, 12))
        r = datasets.stream(seed + 21, n + n * k + k)
        values = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
        cost = (r[n:n + n * k] % np.uint64(7)).astype(np.int64)
        cost[cost < int(rng.choice([0, 3, 6]))] = 0
        row_w = (r[n + n * k:] % np.uint64(20)).astype(np.int64) + 1
        ex_mode, ex_level = int(rng.integers(2)), int(rng.choice([-1, 0, 1]))
        leaves = [("change",), ("swap",), ("change", "swap")][int(rng.integers(3))]
        desc.update(n=n, k=k, ex_mode=ex_mode, ex_level=ex_level, leaves=leaves)
        d = sfa.build_assignment(values, cost.reshape(n, k), k, cost_weight=2, row_w=row_w, ex_mode=ex_mode, ex_level=ex_level, ex_weight=3,
                                 leaves=leaves)
        ocost, ocw, ocost2 = cost.reshape(n, k), 2, None
        if rng.random() < 0.5:  # uni filters / weights as programs (round 6): compiled by the library, restated here for the oracle's matrix
            cols = {20: rng.integers(-3, 6, n).astype(np.int32), 21: rng.integers(-3, 6, n).astype(np.int32),  # per entity
                    22: rng.integers(-3, 6, max(n, k)).astype(np.int32)[:max(n, k)], 23: np.arange(max(n, k), dtype=np.int32)}  # per value (long enough either way)
            cols[24] = rng.integers(0, 4, n).astype(np.int32)  # table keys
            cols[25] = rng.integers(0, 4, max(n, k)).astype(np.int32)
            table = rng.integers(-2, 5, (4, 4)).astype(np.int64)
            for fid, c_ in cols.items():
                d.add_fact_column_i32(fid, c_)
            d.add_fact_matrix(26, table)
            ocost, ocw = 2 * cost.reshape(n, k), 1
            n_prog = int(rng.integers(1, 4))
            for _ in range(n_prog):
                terms, cl = [], 0
                for _t in range(int(rng.integers(0, 5))):
                    lhs = int(rng.integers(1, 7))
                    f, fb, fc = -1, -1, -1
                    if lhs == 1:
                        f = int(rng.choice([20, 21]))
                    elif lhs == 3:
                        f = int(rng.choice([22, 23]))
                    elif lhs in (4, 5):
                        f, fb = int(rng.choice([20, 21])), int(rng.choice([22, 23]))
                    elif lhs == 6:
                        f, fb, fc = int(rng.choice([24, -1])) if n <= 4 else 24, int(rng.choice([25, -1])) if k <= 4 else 25, 26
                    terms.append((lhs, int(rng.integers(0, 6)), cl, f, fb, fc, int(rng.integers(-2, 5))))
                    cl += int(rng.integers(0, 2))
                wl = int(rng.choice([0, 1, 3, 5, 6]))
                weight = {0: (0, -1, -1, -1), 1: (1, 20, -1, -1), 3: (3, 22, -1, -1), 5: (5, 21, 23, -1), 6: (6, 24, 25, 26)}[wl]
                scale = int(rng.integers(1, 4))
                plevel = int(rng.choice([1, 1, 0]))  # a hard program now and then: the class's programs fold into one matrix per level (two at most)
                d.add_uni_program(0, terms, weight, level=plevel, scale=scale)
                um = uni_matrix(terms, weight, scale, cols, table, n, k)
                if plevel == 1:
                    ocost = ocost + um
                else:
                    ocost2 = um if ocost2 is None else ocost2 + um
            desc["uni_programs"] = n_prog
        o = sfo.Model.assignment(values, ocost, k, cost_weight=ocw, row_w=row_w, ex_mode=ex_mode, ex_level=ex_level, ex_weight=3, cost2=ocost2,
                                 cost2_level=0 if ocost2 is not None else -1)
        lists = lambda: (d.working_values(0, 0).tolist(), o.get_vars(0, 0).tolist())
        cands = [[(int(e), int(v)) for e, v in zip(rng.integers(0, n, m), rng.integers(-1, k, m))] for m in rng.integers(1, 9, 40)]
        d.calculate_score()
        cs, cd = d.evaluate_candidates(cands)
        ocs, ocd = o.evaluate_compound(cands)
        assert (cd == ocd).all() and (cs == ocs[:, :2]).all(), "compound candidates"
    elif model == "precedence":  # ListPrecedenceMakespanConstraint: list-only shop (cyclic / partly scheduled / wrong-owner starts)
        nj = int(rng.integers(2, 11)); nm = int(rng.integers(2, 7))
        p = datasets.make_precedence_shop(nj, nm, seed=seed, scheduled=rng.random() < 0.9, max_duration=int(rng.choice([1, 9, 1000])))
        if rng.random() < 0.6:  # permute the sequences (cycles), drop some operations, move some to another machine
            seqs = [list(x) for x in p["sequences"]]
            for x in seqs:
                rng.shuffle(x)
            for _ in range(int(rng.integers(0, 4))):
                v = int(rng.integers(nm))
                if seqs[v]:
                    x = seqs[v].pop(int(rng.integers(len(seqs[v]))))
                    if rng.random() < 0.5:
                        seqs[int(rng.integers(nm))].append(x)
            p["sequences"] = seqs
        with_owner = bool(rng.random() < 0.7)
        pool = ["precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin"]  # critical-path leaf, ruin: round 3
        chosen = set(rng.choice(pool, size=int(rng.integers(1, 8)), replace=False).tolist())
        leaves = tuple(x for x in pool if x in chosen)
        policy = bool(rng.random() < 0.5)  # the runtime slot's precedence policy: route-graph filter + ruin hooks
        ruin = (int(rng.choice([1, 2])), int(rng.choice([2, 5, 6])), int(rng.choice([1, 3, 6])))
        groups = str(rng.choice(["auto", "auto", "0", "2", "4", "8", "16"]))  # grouped trial evaluator (round 4): trials per wavefront, read at every launch
        os.environ.pop("SF_AMD_PREC_GROUPS", None)
        if groups != "auto":
            os.environ["SF_AMD_PREC_GROUPS"] = groups
        desc.update(nj=nj, nm=nm, leaves=leaves, with_owner=with_owner, policy=policy, ruin=ruin, groups=groups)
        d = sfa.build_precedence_shop(p, leaves=leaves, with_owner=with_owner, ruin=ruin, precedence_policy=policy)
        o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"] if with_owner else None)
        o.set_kopt(1, 0)
        o.set_precedence_policy(policy)
        post_configure = lambda: o.set_ruin(*ruin)
        lists = lambda: (d.working_lists(0, 0), o.get_lists(0))
    elif model == "shift":  # consecutive-runs collector + complemented / plain grouped count (examples/minimal-shift-scheduling)
        nn = int(rng.integers(2, 10)); nd = int(rng.integers(3, 40)); per = int(rng.integers(1, 4))
        day = np.repeat(np.arange(nd), per).astype(np.int64)
        if rng.random() < 0.3:
            day = day * 2  # gaps: no two points are consecutive
        nurse = (datasets.stream(seed + 31, len(day)) % np.uint64(nn + 1)).astype(np.int64) - 1
        lim = int(rng.choice([0, 1, 2, 5])); cw = int(rng.choice([0, 1, 3])); tgt = int(rng.choice([-1, 0, 4]))
        leaves = [("change",), ("swap",), ("change", "swap")][int(rng.integers(3))]
        desc.update(nn=nn, nd=nd, per=per, limit=lim, cw=cw, target=tgt, leaves=leaves)
        ws = int(rng.choice([1, 7]))
        req = (datasets.stream(seed + 33, len(day)) % np.uint64(3)).astype(np.int64) if rng.random() < 0.5 else None  # required flag / weight
        pres = None
        if rng.random() < 0.4:  # the indexed_presence collector instead of the runs: count / count_in / capped / any_in
            lo = int(rng.integers(0, nd)); hi = int(rng.choice([lo, lo + 1, lo + 5, 4096]))
            pres = (lo if hi != 4096 else int(rng.choice([0, lo])), hi, int(rng.choice([0, 0, 1, 3])))
            if rng.random() < 0.4:  # complement_runs(lo..hi) excess over cap
                pres = (pres[0], min(hi, int(rng.choice([nd, nd + 7, 2 * nd + 3]))), pres[2], 1)
            desc.update(presence=pres)
        d = sfa.build_shift_schedule(nurse, day, nn, limit=lim, w_streak=ws, count_weight=cw, target=tgt, leaves=leaves, required=req, presence=pres)
        o = sfo.Model.shift_schedule(nurse, day, nn, limit=lim, w_streak=ws, count_weight=cw, target=tgt, required=req, presence=pres)
        lists = lambda: (d.working_values(0, 0).tolist(), o.get_vars(0, 0).tolist())
    else:
        nj = int(rng.integers(2, 9)); nm = int(rng.integers(2, 6))
        p = datasets.make_jobshop(nj, nm)
        if rng.random() < 0.8:
            p = datasets.construct_jobshop(p, seed=seed)
        pool = ["list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "change", "swap"]
        chosen = set(rng.choice(pool, size=int(rng.integers(1, 9)), replace=False).tolist())
        leaves = tuple(x for x in pool if x in chosen)
        levels = 3
        desc.update(nj=nj, nm=nm, leaves=leaves)
        mk = bool(rng.random() < 0.5)  # add the makespan objective (precedence constraint on the list class of the mixed model)
        if mk:
            p["durations"] = (datasets.stream(seed + 41, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
            if rng.random() < 0.5 and len(p["sequences"][0]) > 1:
                p["sequences"][0] = p["sequences"][0][::-1]  # against the job order: cycles
        # the join of the two planning classes (an operation on a machine that does not schedule it), on a random level
        own = int(rng.integers(0, 3)) if (not mk and rng.random() < 0.45) else None
        desc.update(makespan=mk, owner_match_level=own)
        d = sfa.build_jobshop(p, leaves=leaves, makespan=mk, owner_match_level=own)
        o = sfo.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, durations=p["durations"] if mk else None, owner_match_level=own)
        o.set_kopt(1, 0)
        lists = lambda: ((d.working_lists(1, 0), d.working_values(0, 0).tolist()), (o.get_lists(1), o.get_vars(0, 0).tolist()))
    bits = sum(BITS[x] for x in leaves)
    # root union: mostly the default policy, sometimes another order / weights (those run in the generic engine; value-keyed
    # scalar models cannot -> SF_ERR_UNSUPPORTED, counted separately)
    union_order, union_weights = -1, None
    if len(leaves) > 1 and rng.random() < 0.35:
        union_order = int(rng.integers(0, 5))
        if union_order >= 3 and rng.random() < 0.6:
            union_weights = [int(w) for w in rng.integers(0, 5, len(leaves))]
            if sum(union_weights) == 0:
                union_weights[0] = 1
        d.configure_union(union_order, union_weights)
    desc.update(union_order=union_order, union_weights=union_weights)
    o.configure(acceptor=1 if acceptor in (3, 4) else acceptor, la_size=la, forager=forager, limit=limit, leaves=bits,
                selection_order=order, random_seed=seed, max_nearby=desc.get("max_nearby", 20), union_order=union_order)
    if union_weights:
        o.set_union_weights(union_weights)
    if model in ("cvrp", "precedence"):
        post_configure()
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=la, forager=forager, accepted_count_limit=limit,
                                 selection_order=order, random_seed=seed))
    if acceptor == 4:
        o.configure_diversified(la, dla_tol)
        d.configure_diversified(dla_tol)
    if acceptor == 3:
        ss = int(rng.choice([1, 9, 128]))
        o.configure_annealing(mode=2, levels=levels, hard_levels=levels - 1, sample_size=ss, seed=seed)
        d.configure_annealing(mode=2, calibration_sample_size=ss, seed=seed)
    assert (d.calculate_score()[0] == o.score()[:levels]).all(), "initial score"
    stats_base = None
    if model == "precedence" and rng.random() < 0.5:  # cheapest insertion of whatever is in no list (round 3: on precedence models; with the policy: downstream order)
        els = rng.permutation(len(p["durations"])).astype(np.uint32)
        placed = {int(x) for sq in o.get_lists(0) for x in sq}
        sc = d.construct_list_cheapest(0, els)
        o.construct_list_cheapest([int(x) for x in els if int(x) not in placed])
        desc["constructed"] = True
        assert (sc[0] == o.score()[:levels]).all(), "constructed score"
        a, b = lists()
        assert a == b, "constructed lists"
        gst, stats_base = d.stats(0), o.stats()  # sf_phase_start zeroes the device's counters; the oracle's run on
        for k2 in ["step_count", "moves_accepted", "moves_applied", "score_calculations"]:
            assert gst[k2] == stats_base[k2], f"construction counter {k2}"
    if model == "jobshop" and desc.get("owner_match_level") is not None:  # round 6: the host-driven entry points price the join too
        om0 = o.enumerate(0, 0, seed, 3)
        if len(om0):
            os0, od0 = o.evaluate_moves(om0)
            gs0, gd0 = d.evaluate_moves(om0)
            assert (gd0 == od0).all() and (gs0[od0 != 0] == os0[od0 != 0, :levels]).all(), "host-driven trial scores under the two-class join"
            if od0.any():
                mv0 = om0[np.flatnonzero(od0)[0]]
                o.apply_move(mv0); d.apply_move(mv0)
                assert (d.calculate_score()[0] == o.score()[:levels]).all() and (d.fresh_score()[0] == o.score()[:levels]).all(), "sf_apply under the two-class join"
    d.phase_start(); o.phase_start()
    for step in range(6 if forager == 2 else 14):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 20)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), f"step {step}: trace length {len(gm)} vs {len(om)}"
        assert (t6(gm) == t6(om)).all(), f"step {step}: candidate order"
        assert (gs == os_[:, :levels]).all(), f"step {step}: trial scores"
        assert (gf == of).all(), f"step {step}: flags"
        assert gap == oap and (not gap or tuple(gmv) == tuple(omv)), f"step {step}: applied move"
        a, b = lists()
        assert a == b, f"step {step}: state"
    n_fused = 5 if forager == 2 else 40
    d.solve_steps(n_fused); o.steps(n_fused)
    a, b = lists()
    assert a == b, "fused state"
    assert (d.calculate_score()[0] == o.score()[:levels]).all() and (d.fresh_score()[0] == o.score()[:levels]).all(), "fused score"
    gst, ost = d.stats(0), o.stats()
    for k2 in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k2] == ost[k2] - (stats_base[k2] if stats_base else 0), f"counter {k2}"
    return desc


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); ran = 0; failures = []; by_model = {}
while time.time() - t0 < budget:
    try:
        desc = run_case(seed)
        by_model[desc["model"]] = by_model.get(desc["model"], 0) + 1
    except sfa.SolverForgeError as e:
        if "SF_ERR_UNSUPPORTED" not in str(e):
            failures.append({"seed": seed, "error": str(e)[:300]})
        else:
            by_model["unsupported"] = by_model.get("unsupported", 0) + 1
    except AssertionError as e:
        failures.append({"seed": seed, "error": str(e)[:300]})
    except Exception as e:
        failures.append({"seed": seed, "error": traceback.format_exc()[-400:]})
    ran += 1; seed += 1
print(json.dumps({"cases": ran, "by_model": by_model, "failures": failures[:20], "n_failures": len(failures)}))
