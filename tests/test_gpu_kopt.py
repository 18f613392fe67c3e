"""GPU parity tests (through the C ABI): the 3-opt leaf of the generic N-leaf engine vs the CPU oracle --
full enumeration (k_opt/full.rs) and the distance-pruned cut state machine (k_opt/nearby.rs,
nearby_state.rs): candidate order, trial scores, committed moves, fused steps; asymmetric matrix with
unreachable legs, massive distance ties, short routes, one long route (HBM key scratch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEAF_BITS = {"nearby_change": 16, "nearby_swap": 32, "list_change": 4, "list_swap": 8, "list_reverse": 64,
             "sublist_change": 128, "sublist_swap": 256, "kopt": 512}


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _problem(kind):
    from solverforge_amd import datasets

    if kind == "plain":
        return datasets.make_cvrp(40, 4, 60, seed=3)
    if kind == "asym":  # asymmetric matrix, unreachable / negative legs, routes of 1, 3, 4, 5 and many elements
        p = datasets.make_cvrp(36, 6, 40, seed=8)
        r = datasets.stream(123, p["matrix"].size).reshape(p["matrix"].shape)
        p["matrix"] = (p["matrix"] + (r % np.uint64(17)).astype(np.int64)).astype(np.int64)
        np.fill_diagonal(p["matrix"], 0)
        p["matrix"][4, 9] = np.iinfo(np.int64).max
        p["matrix"][11, 2] = -3
        p["matrix"][7, :] = np.iinfo(np.int64).max  # node 7 reaches nothing: INFINITY ties in its row
        rt = p["routes"]
        rt[3] = rt[3][:1]
        rt[4] = rt[4][:3]
        rt[5] = rt[5][:4]
        seen = set()
        rt[0] = rt[0] + list(range(1, 37))
        p["routes"] = [[c for c in x if not (c in seen or seen.add(c))] for x in rt]
        return p
    if kind == "ties":  # every distance equal: the stable (distance, position) order decides
        p = datasets.make_cvrp(30, 3, 60, seed=5)
        p["matrix"][:] = 7
        np.fill_diagonal(p["matrix"], 0)
        return p
    if kind == "long":  # one route of 150 elements (> the LDS key buffer), the rest short
        p = datasets.make_cvrp(170, 5, 2000, seed=6)
        allc = [c for rt in p["routes"] for c in rt]
        p["routes"] = [allc[:150], allc[150:155], allc[155:160], allc[160:166], allc[166:]]
        return p
    raise ValueError(kind)


def _mk(oracle, p, leaves, kopt, n_replicas=1, max_nearby=10):
    import solverforge_amd as sfa

    d = sfa.build_cvrp(p, n_replicas=n_replicas, leaves=leaves, kopt=kopt, max_nearby=max_nearby)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.set_kopt(kopt[0], kopt[1])
    bits = 0
    for name in leaves:
        bits |= LEAF_BITS[name]
    return d, o, bits


@pytest.mark.parametrize("problem,kopt", [
    ("plain", (1, 20)), ("plain", (1, 3)), ("plain", (1, 0)), ("plain", (2, 4)), ("plain", (2, 0)),
    ("asym", (1, 20)), ("asym", (1, 2)), ("asym", (1, 0)), ("ties", (1, 5)), ("ties", (1, 64)),
])
def test_kopt_cursor_order_and_trial_scores(oracle, problem, kopt):
    p = _problem(problem)
    d, o, bits = _mk(oracle, p, ("kopt",), kopt)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3, 4):
        o.configure(leaves=bits, selection_order=order)
        for step_index, step_seed in [(0, 0), (5, 77), (123, 0xDEADBEEFCAFEF00D)]:
            gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order, cap=1 << 18)
            om = o.enumerate(0, step_index, step_seed, order)
            assert len(gm) == len(om) > 0, (order, step_index)
            assert (_t(gm) == _t(om)).all(), (order, step_index)
            os_, od = o.evaluate_moves(om)
            assert (gd == od).all() and (gs == os_[:, :2]).all()
            es, ed = d.evaluate_moves(om)  # sf_step_evaluate on host-provided 3-opt moves
            assert (ed == od).all() and (es == os_[:, :2]).all()


def test_kopt_long_route_uses_hbm_scratch(oracle):
    p = _problem("long")
    d, o, bits = _mk(oracle, p, ("kopt",), (1, 6))
    d.calculate_score()
    for order in (0, 3):
        o.configure(leaves=bits, selection_order=order)
        gm, gs, gd = d.open_cursor(2, 99, selection_order=order, cap=1 << 20)
        om = o.enumerate(0, 2, 99, order)
        assert len(gm) == len(om) > 1000
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om[:5000])
        assert (gd[:5000] == od).all() and (gs[:5000] == os_[:, :2]).all()


@pytest.mark.parametrize("problem,leaves,kopt", [
    ("asym", ("kopt",), (1, 20)),
    ("asym", ("kopt",), (1, 0)),
    ("plain", ("nearby_change", "nearby_swap", "kopt"), (1, 20)),
    # the default list policy without ruin (policy/list.rs:24-33)
    ("plain", ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt"), (1, 20)),
    ("asym", ("list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt"), (1, 0)),
])
def test_kopt_apply_and_traced_steps(oracle, problem, leaves, kopt):
    import solverforge_amd as sfa

    p = _problem(problem)
    d, o, bits = _mk(oracle, p, leaves, kopt)
    d.calculate_score()
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=40, max_nearby=10)
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=5, accepted_count_limit=40))
    rng = np.random.default_rng(5)
    for it in range(8):  # committed 3-opt moves through sf_apply, every pattern
        mv = o.enumerate(512, it, 9 + it, 3)
        mv = mv[mv["value"] == it % 7]
        mv = mv[rng.integers(len(mv))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0), it
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    applied_kopt = 0
    for step in range(25):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
            applied_kopt += int(gmv["kind"] == 7)
        assert d.working_lists(0, 0) == o.get_lists(0), step
    assert applied_kopt > 0
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


def test_jobshop_default_policy_eight_leaf_union(oracle):
    """Mixed job shop with the reference's whole default policy minus ruin: list change, list swap, sublist
    change, sublist swap, reverse, unbounded 3-opt, scalar change, scalar swap (8 leaves, StratifiedRandom)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_jobshop(8, 4)
    n = p["n_ops"]
    r = datasets.stream(3, 3 * n)
    p["machine_idx"] = (r[:n] % np.uint64(5)).astype(np.int64) - 1
    seqs = [[] for _ in range(4)]
    for op in range(n):
        where = int(r[n + op] % np.uint64(5))
        if where < 4:
            seqs[where].append(op)
    p["sequences"] = seqs
    leaves = ("list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "change", "swap")
    d = sfa.build_jobshop(p, leaves=leaves)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
    o.set_kopt(1, 0)
    bits = 4 | 8 | 128 | 256 | 64 | 512 | 1 | 2
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    for order in (0, 3):
        o.configure(leaves=bits, selection_order=order)
        gm, gs, gd = d.open_cursor(3, 17, selection_order=order, cap=1 << 20)
        om = o.enumerate(0, 3, 17, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :3]).all()
    o.configure(leaves=bits, random_seed=6, la_size=7, limit=48)
    d.configure(sfa.SolverConfig(random_seed=6, late_acceptance_size=7, accepted_count_limit=48))
    d.phase_start()
    o.phase_start()
    kinds = set()
    for step in range(30):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 20)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :3]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
            kinds.add(int(gmv["kind"]))
        assert d.working_lists(1, 0) == o.get_lists(1), step
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all(), step
    assert len(kinds) >= 3
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(1, 0) == o.get_lists(1)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.fresh_score()[0] == o.score()[:3]).all()


def test_kopt_selector_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    for bad in [dict(k=2), dict(max_nearby=65), dict(min_segment_len=0)]:
        d = sfa.build_cvrp(p, leaves=())
        with pytest.raises(sfa.SolverForgeError):
            d.add_kopt_selector(0, **bad)
