"""GPU parity tests (through the C ABI): the root union's selection orders and weights (UnionMoveSelectorConfig; scheduler
heuristic/selector/decorator/vec_union.rs:190-365) -- Sequential, RoundRobin, RotatingRoundRobin, Random and StratifiedRandom,
weighted children (smooth weighted round-robin, weighted random draws, a zero weight) -- vs the oracle's UnionScheduler:
candidate order with the child index of every pull, trial scores, committed moves, fused steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEAF_BITS = {"nearby_change": 16, "nearby_swap": 32, "list_change": 4, "list_swap": 8, "list_reverse": 64,
             "sublist_change": 128, "sublist_swap": 256, "kopt": 512, "ruin": 1024}


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


CASES = [
    (0, None), (1, None), (2, None), (3, None), (4, None),
    (3, [3, 1, 2, 1, 5]), (4, [3, 1, 2, 1, 5]), (4, [1, 0, 4, 1, 1]), (3, [0, 2, 0, 1, 1]), (4, [7, 7, 7, 7, 7]),
]


@pytest.mark.parametrize("order,weights", CASES)
def test_union_orders_and_weights_cvrp(oracle, order, weights):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    leaves = ("nearby_change", "nearby_swap", "sublist_change", "list_reverse", "kopt")
    p = datasets.make_cvrp(40, 4, 60, seed=3)
    d = sfa.build_cvrp(p, leaves=leaves, max_nearby=10)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = sum(LEAF_BITS[x] for x in leaves)
    d.configure_union(order, weights)
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=5, accepted_count_limit=40))
    d.calculate_score()
    for sel_order in (3, 0):  # the whole cursor, drained: every child runs dry at some point
        o.configure(leaves=bits, random_seed=3, la_size=5, limit=40, max_nearby=10, union_order=order, selection_order=sel_order)
        if weights:
            o.set_union_weights(weights)
        for step_index, step_seed in [(0, 0), (7, 0xDEADBEEFCAFEF00D)]:
            gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=sel_order, cap=1 << 18)
            om = o.enumerate(0, step_index, step_seed, sel_order)
            assert len(gm) == len(om) > 0, (sel_order, step_index)
            assert (_t(gm) == _t(om)).all(), (sel_order, step_index)
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=40, max_nearby=10, union_order=order)
    if weights:
        o.set_union_weights(weights)
    d.phase_start()
    o.phase_start()
    for step in range(20):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step  # flags carry the child index
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
    d.solve_steps(60)
    o.steps(60)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("order,weights", [(1, None), (3, [3, 1]), (4, [1, 4]), (0, None)])
def test_union_orders_scalar_model(oracle, order, weights):
    """A configured root union sends a scalar-only model to the generic engine too."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(200, 900, 5, seed=3)
    g["colors"] = (datasets.stream(9, 200) % np.uint64(6)).astype(np.int64) - 1
    d = sfa.build_graph_coloring(g)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    d.configure_union(order, weights)
    d.configure(sfa.SolverConfig(random_seed=2, late_acceptance_size=5, accepted_count_limit=30))
    o.configure(leaves=3, random_seed=2, la_size=5, limit=30, union_order=order)
    if weights:
        o.set_union_weights(weights)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(50)
    o.steps(50)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_union_configure_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p)
    with pytest.raises(sfa.SolverForgeError):
        d.configure_union(5)
    with pytest.raises(sfa.SolverForgeError):
        d.configure_union(1, [2, 1])  # weights need Random / StratifiedRandom
    with pytest.raises(sfa.SolverForgeError):
        d.configure_union(4, [-1, 1])
    d.configure_union(4, [2, 1, 1])  # count mismatch surfaces at the launch
    d.calculate_score()
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.solve_steps(1)


def test_configured_union_keeps_declaration_order(oracle):
    """a configured union's children are scheduled (and weighted) in the order of the sf_selector_add calls, not
    in the default policy's kind order.  Sequential drains child 0 first: declared (nearby swap, nearby change), the stream is
    the swap leaf's stream followed by the change leaf's; a zero weight on declared child 0 leaves only the change leaf."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(30, 3, 60, seed=5)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    streams = {}
    for name in ("nearby_swap", "nearby_change"):
        o.configure(leaves=LEAF_BITS[name], random_seed=1, la_size=5, limit=40, max_nearby=8, selection_order=3)
        streams[name] = _t(o.enumerate(0, 2, 99, 3))
    d = sfa.build_cvrp(p, leaves=("nearby_swap", "nearby_change"), max_nearby=8)
    d.configure_union(0, None)  # Sequential
    d.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=5, accepted_count_limit=40))
    d.calculate_score()
    gm, _, _ = d.open_cursor(2, 99, selection_order=3, cap=1 << 18)
    assert (_t(gm) == np.concatenate([streams["nearby_swap"], streams["nearby_change"]])).all()
    d2 = sfa.build_cvrp(p, leaves=("nearby_swap", "nearby_change"), max_nearby=8)
    d2.configure_union(4, [0, 3])  # weights follow the declaration order: the swap leaf is disabled
    d2.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=5, accepted_count_limit=40))
    d2.calculate_score()
    gm2, _, _ = d2.open_cursor(2, 99, selection_order=3, cap=1 << 18)
    assert (_t(gm2) == streams["nearby_change"]).all()


@pytest.mark.parametrize("model", ["bins_sum2", "bins_cap", "bins_fair", "bins_balance", "bins_tri", "assignment", "shift"])
def test_value_keyed_constraints_in_the_generic_engine(oracle, model):
    """the value-keyed nodes (keyed self-join incl. higher arity, grouped sum, load balance, balance,
    keyed cross-join + exists, consecutive runs + complemented sum) with a configured root union, i.e. through the generic N-leaf
    engine (per-value tables in the replica's LDS slice, updated at commit): cursor stream with trial scores, traced and fused
    steps vs the oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    rng = np.random.default_rng(7)
    n, k = 70, 6
    bins = rng.integers(-1, k, n).astype(np.int64)
    sizes = rng.integers(1, 9, n).astype(np.int64)
    if model.startswith("bins"):
        cap = {"bins_sum2": -1, "bins_cap": 25, "bins_fair": -2, "bins_balance": -3, "bins_tri": 25}[model]
        arity = 3 if model == "bins_tri" else 2
        d = sfa.build_balance(bins, sizes, k, n_replicas=2, w_pair=3, cap=cap, arity=arity)
        mk = lambda: oracle.Model.balance(k, bins, sizes, w_pair=3, cap=cap, arity=arity)
    elif model == "assignment":
        cost = rng.integers(0, 20, (n, k)).astype(np.int64)
        cost[rng.random((n, k)) < 0.3] = 0
        row_w = rng.integers(1, 6, k).astype(np.int64)
        d = sfa.build_assignment(bins, cost, k, n_replicas=2, cost_weight=2, row_w=row_w, ex_mode=1, ex_level=1, ex_weight=3)
        mk = lambda: oracle.Model.assignment(bins, cost, k, cost_weight=2, row_w=row_w, ex_mode=1, ex_level=1, ex_weight=3)
    else:
        n_nurses, n_days = 6, 12  # consecutive-runs (streak excess) + complemented workload: minimal-shift-scheduling's constraints
        day = np.repeat(np.arange(n_days), 3).astype(np.int64)
        nurse = rng.integers(-1, n_nurses, len(day)).astype(np.int64)
        kw = dict(limit=2, w_streak=1, count_weight=1, target=4)
        d = sfa.build_shift_schedule(nurse, day, n_nurses, n_replicas=2, **kw)
        mk = lambda: oracle.Model.shift_schedule(nurse, day, n_nurses, **kw)
    o = mk()
    d.configure_union(1, None)  # RoundRobin: a configured union runs in the generic engine
    d.configure(sfa.SolverConfig(random_seed=4, late_acceptance_size=5, accepted_count_limit=30))
    o.configure(leaves=3, random_seed=4, la_size=5, limit=30, union_order=1)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3):
        o.configure(leaves=3, random_seed=4, la_size=5, limit=30, union_order=1, selection_order=order)
        gm, gs, gd = d.open_cursor(2, 55, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 2, 55, order)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all(), order
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all(), order
    o.configure(leaves=3, random_seed=4, la_size=5, limit=30, union_order=1)
    d.phase_start()
    o.phase_start()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(40)
    scores = d.calculate_score()
    for r in range(2):
        o2 = mk()
        o2.configure(leaves=3, random_seed=4 + r, la_size=5, limit=30, union_order=1)
        o2.phase_start()
        o2.steps(52)
        assert (scores[r] == o2.score()[:2]).all(), r
        assert (d.working_values(0, 0, replica=r) == o2.get_vars(0, 0)).all(), r
    assert (d.fresh_score() == scores).all()
