"""GPU parity tests (through the C ABI): HIP scalar-variable hot path (graph colouring, N-queens)
vs the CPU oracle.  Bit-exact integer scores, candidate order, accept flags, applied moves, counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["b"], moves["value"]], axis=1)


def _graph(n=300, e=1500, k=6, seed=3, assign=True):
    from solverforge_amd import datasets

    g = datasets.make_graph(n, e, k, seed=seed)
    if assign:  # a deterministic partial colouring with conflicts and some unassigned nodes
        r = datasets.stream(seed + 99, n)
        g["colors"] = (r % np.uint64(k + 1)).astype(np.int64) - 1
    return g


def _mk(oracle, g, n_replicas=1, leaves=("change", "swap")):
    import solverforge_amd as sfa

    d = sfa.build_graph_coloring(g, n_replicas=n_replicas, leaves=leaves)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    bits = (oracle.LEAF_SCALAR_CHANGE if "change" in leaves else 0) | (oracle.LEAF_SCALAR_SWAP if "swap" in leaves else 0)
    return d, o, bits


def test_graph_initialize_and_fresh_score(oracle):
    g = _graph()
    d, o, _ = _mk(oracle, g)
    s = d.calculate_score()
    assert (s[0] == o.score()[:2]).all() and s[0][0] < 0
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()


@pytest.mark.parametrize("order", [0, 3, 4])
@pytest.mark.parametrize("leaves", [("change",), ("swap",), ("change", "swap")])
def test_graph_cursor_order_and_trial_scores(oracle, order, leaves):
    g = _graph(n=90, e=400, k=5, seed=4)
    d, o, bits = _mk(oracle, g, leaves=leaves)
    o.configure(leaves=bits, selection_order=order)
    d.calculate_score()
    for step_index, step_seed in [(0, 0), (7, 41), (123, 0xDEADBEEFCAFEF00D)]:
        gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order, cap=1 << 17)
        om = o.enumerate(0, step_index, step_seed, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all()
        assert (gs == os_[:, :2]).all()


def test_graph_step_evaluate_and_apply(oracle):
    g = _graph(n=120, e=700, k=4, seed=5)
    d, o, bits = _mk(oracle, g)
    o.configure(leaves=bits)
    d.calculate_score()
    rng = np.random.default_rng(1)
    for it in range(25):
        om = o.enumerate(0, it, 1000 + it, 3)
        os_, od = o.evaluate_moves(om)
        gs, gd = d.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        doable = np.flatnonzero(od)
        mv = om[doable[rng.integers(len(doable))]]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("acceptor,forager,limit", [(1, 0, 64), (0, 0, 3), (1, 1, 1), (0, 2, 1), (1, 3, 0), (1, 4, 0), (1, 4, 5)])
def test_graph_traced_steps(oracle, acceptor, forager, limit):
    import solverforge_amd as sfa

    g = _graph(n=70, e=260, k=4, seed=6)
    d, o, bits = _mk(oracle, g)
    o.configure(acceptor=acceptor, la_size=5, forager=forager, limit=limit, leaves=bits, random_seed=9)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=5, forager=forager,
                                 accepted_count_limit=limit, random_seed=9))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(10 if forager in (2, 3) else 30):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 17)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all(), step
        assert (gf == of).all(), step
        assert (gs == os_[:, :2]).all(), step
        assert gap == oap, step
        if gap:
            assert (gmv["kind"], gmv["a"], gmv["b"], gmv["value"]) == (omv["kind"], omv["a"], omv["b"], omv["value"]), step
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all(), step
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.best_scores()[0] == o.best_score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
              "score_calculations", "moves_not_doable"]:
        assert gst[k] == ost[k], k


def test_graph_fused_multi_replica_from_unassigned(oracle):
    """All nodes start unassigned (the reference example's initial state); several replicas."""
    import solverforge_amd as sfa

    g = _graph(n=400, e=2400, k=8, seed=7, assign=False)
    R = 3
    d, _, bits = _mk(oracle, g, n_replicas=R)
    d.configure(sfa.SolverConfig(random_seed=2))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(40)
    d.solve_steps(35)
    sc = d.calculate_score()
    for r in range(R):
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        o.configure(leaves=bits, random_seed=2 + r)
        o.phase_start()
        o.steps(75)
        assert (sc[r] == o.score()[:2]).all(), r
        assert (d.working_values(0, 0, r) == o.get_vars(0, 0)).all(), r
        assert d.stats(r)["moves_evaluated"] == o.stats()["moves_evaluated"], r
    assert (d.fresh_score() == sc).all()


def test_nqueens_64(oracle):
    """BASELINE config 1 shape (N = 64) on the device: queens predicate join + unassigned."""
    import solverforge_amd as sfa

    n = 64
    rows = (np.arange(n) * 7 % (n + 1)).astype(np.int64) - 1  # some unassigned, many conflicts
    d = sfa.build_nqueens(rows)
    o = oracle.Model.nqueens(rows)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=3)
    d.configure(sfa.SolverConfig(random_seed=3))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    gm, gs, gd = d.open_cursor(1, 5, selection_order=3, cap=1 << 16)
    om = o.enumerate(0, 1, 5, 3)
    assert (_t(gm) == _t(om)).all()
    os_, od = o.evaluate_moves(om)
    assert (gd == od).all() and (gs == os_[:, :2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(50)
    o.steps(50)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


def test_graph_10k_properties(oracle):
    """BASELINE config 2 size (10k nodes / 100k edges, 16 colours): incremental == full
    recalculation, the first steps equal the oracle, scores never exceed 0."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(10000, 100000, 16, seed=0)
    d, o, bits = _mk(oracle, g, n_replicas=2)
    d.configure(sfa.SolverConfig(random_seed=0))
    s0 = d.calculate_score()
    assert (s0[0] == [-10000, 0]).all()
    d.phase_start()
    d.solve_steps(30)
    o.configure(leaves=bits, random_seed=0)
    o.phase_start()
    o.steps(30)
    sc = d.calculate_score()
    assert (sc[0] == o.score()[:2]).all()
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.fresh_score() == sc).all()
    assert (sc <= 0).all() and (sc[:, 0] > -10000).all()


def _balance(n=80, k=7, seed=11):
    from solverforge_amd import datasets

    r = datasets.stream(seed, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    sizes = (r[n:] % np.uint64(9)).astype(np.int64) + 1
    return bins, sizes, k


@pytest.mark.parametrize("cap", [-1, 25, -2, -3])
def test_keyed_selfjoin_and_grouped_sum(oracle, cap):
    """Value-keyed aggregates: pairs sharing a bin (keyed self-join bi node) and group_by(bin,
    sum(size)) with sum^2 / excess-over-cap weights (grouped node + sum collector), or (cap = -2) the load_balance
    collector's unfairness = round(sqrt(sum x^2 - (sum x)^2 / keys)) — the f64 step of the scoring path — or (cap = -3) the
    BalanceConstraint's round(1000 * standard deviation of the per-bin counts): full scores, the
    whole candidate stream with trial scores, committed moves, traced steps and a fused solve."""
    import solverforge_amd as sfa

    bins, sizes, k = _balance()
    d = sfa.build_balance(bins, sizes, k, w_pair=3, cap=cap)
    o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=cap)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=5, la_size=6, limit=40)
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=40))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()
    for order in (0, 3):
        o.configure(leaves=bits, random_seed=5, la_size=6, limit=40, selection_order=order)
        gm, gs, gd = d.open_cursor(2, 99, selection_order=order, cap=1 << 16)
        om = o.enumerate(0, 2, 99, order)
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == os_[:, :2]).all()
    o.configure(leaves=bits, random_seed=5, la_size=6, limit=40)
    rng = np.random.default_rng(3)
    for it in range(10):
        om = o.enumerate(0, it, 7 + it, 3)
        _, od = o.evaluate_moves(om)
        mv = om[np.flatnonzero(od)[rng.integers(int(od.sum()))]]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(40)
    o.steps(40)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


def test_balance_variance_is_not_contracted_on_the_device(oracle):
    """hipcc's default -ffp-contract=fast turned `sum_sq / n - mean * mean` into one v_fma_f64; the library is
    built with -ffp-contract=off.  The counts of tests/test_oracle_golden.py give 13 unfused (reference) and 14 fused; trial
    deltas around that state must agree with the oracle too."""
    import solverforge_amd as sfa
    from test_oracle_golden import BALANCE_FMA_COUNTS, balance_fma_bins

    bins = balance_fma_bins()
    k = len(BALANCE_FMA_COUNTS)
    sizes = np.ones(len(bins), dtype=np.int64)
    d = sfa.build_balance(bins, sizes, k, w_pair=0, cap=-3, balance_base=5)
    o = oracle.Model.balance(k, bins, sizes, w_pair=0, cap=-3, balance_base=5)
    assert d.calculate_score()[0].tolist() == [0, -13] == list(o.score()[:2])
    assert d.fresh_score()[0].tolist() == [0, -13]
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=1, la_size=4, limit=30)
    om = o.enumerate(0, 0, 17, 0)
    os_, od = o.evaluate_moves(om)
    es, ed = d.evaluate_moves(om)
    assert (ed == od).all() and (es == os_[:, :2]).all()


def test_evaluate_each_matches_oracle_per_constraint(oracle):
    """ConstraintSet::evaluate_each: per-constraint score and match count (graph colouring, N-queens, bin balance,
    CVRP, job shop) after some committed steps."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    def check(d, o, levels, steps=15):
        d.calculate_score()
        d.phase_start()
        o.phase_start()
        for it in range(2):
            gs, gc = d.evaluate_each(0)
            os_, oc = o.evaluate_each()
            assert len(gs) == len(os_) > 0
            assert (gs == os_[:, :levels]).all(), (gs, os_)
            assert (gc == oc).all(), (gc, oc)
            assert (gs.sum(axis=0) == d.fresh_score()[0]).all()
            d.solve_steps(steps)
            o.steps(steps)

    g = _graph(n=150, e=600, k=5, seed=8)
    d, o, bits = _mk(oracle, g)
    o.configure(leaves=bits, random_seed=2)
    d.configure(sfa.SolverConfig(random_seed=2))
    check(d, o, 2)

    rows = [int(v) for v in (datasets.stream(5, 12) % np.uint64(13)).astype(np.int64) - 1]
    d = sfa.build_nqueens(rows)
    o = oracle.Model.nqueens(rows)
    o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=3)
    d.configure(sfa.SolverConfig(random_seed=3))
    check(d, o, 2)

    n, nb = 60, 7
    bins = (datasets.stream(11, n) % np.uint64(nb + 1)).astype(np.int64) - 1
    sizes = (datasets.stream(12, n) % np.uint64(9)).astype(np.int64) + 1
    for cap in (-1, 20, -2, -3):
        d = sfa.build_balance(bins, sizes, nb, w_pair=3, cap=cap)
        o = oracle.Model.balance(nb, bins, sizes, 3, cap)
        o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=4)
        d.configure(sfa.SolverConfig(random_seed=4))
        check(d, o, 2)

    p = datasets.make_cvrp(50, 5, 35, seed=2)
    d = sfa.build_cvrp(p)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.configure(leaves=oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP, random_seed=5)
    d.configure(sfa.SolverConfig(random_seed=5))
    check(d, o, 2)

    pj = datasets.make_jobshop(6, 3)
    nops = pj["n_ops"]
    r = datasets.stream(9, 2 * nops)
    pj["machine_idx"] = (r[:nops] % np.uint64(4)).astype(np.int64) - 1
    seqs = [[] for _ in range(3)]
    for op in range(nops):
        w = int(r[nops + op] % np.uint64(4))
        if w < 3:
            seqs[w].append(op)
    pj["sequences"] = seqs
    d = sfa.build_jobshop(pj)
    o = oracle.Model.jobshop(pj["job"], pj["machine_idx"], pj["sequences"], bendable=True)
    o.configure(leaves=4 | 8 | 1 | 2, random_seed=6)
    d.configure(sfa.SolverConfig(random_seed=6))
    check(d, o, 3)


def test_load_balance_large_metrics_and_validation(oracle):
    """load_balance collector with metrics up to 10^6 (radicands ~10^13: the f64 division / sqrt / round must agree bit
    for bit with the CPU), bins emptying and filling (key count changes), multi-replica fused solve; zero metrics are
    refused (the reference skips them, the device's shared count table cannot)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    n, k = 120, 9
    r = datasets.stream(21, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    bins[bins == 4] = -1  # bin 4 starts empty
    sizes = (r[n:] % np.uint64(1_000_000)).astype(np.int64) + 1
    d = sfa.build_balance(bins, sizes, k, n_replicas=3, w_pair=0, cap=-2)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    d.configure(sfa.SolverConfig(random_seed=8, late_acceptance_size=5, accepted_count_limit=30))
    assert (d.calculate_score() == d.fresh_score()).all()
    os_ = []
    for rep in range(3):
        o = oracle.Model.balance(k, bins, sizes, w_pair=0, cap=-2)
        o.configure(leaves=bits, random_seed=8 + rep, la_size=5, limit=30)
        o.phase_start()
        os_.append(o)
    assert (d.calculate_score()[0] == os_[0].score()[:2]).all()
    om = os_[0].enumerate(0, 1, 5, 3)
    ref, od = os_[0].evaluate_moves(om)
    es, ed = d.evaluate_moves(om)
    assert (ed == od).all() and (es == ref[:, :2]).all()
    d.phase_start()
    for chunk in range(4):
        d.solve_steps(50)
        for rep, o in enumerate(os_):
            o.steps(50)
            assert (d.working_values(0, 0, replica=rep) == o.get_vars(0, 0)).all(), (chunk, rep)
            assert (d.calculate_score()[rep] == o.score()[:2]).all(), (chunk, rep)
        assert (d.fresh_score() == d.calculate_score()).all()
    sizes0 = sizes.copy()
    sizes0[3] = 0
    with pytest.raises(sfa.SolverForgeError, match="metrics must be >= 1"):
        sfa.build_balance(bins, sizes0, k, cap=-2).calculate_score()


@pytest.mark.parametrize("arity", [3, 4, 5])
def test_tri_quad_penta_selfjoin(oracle, arity):
    """IncrementalTri / Quad / PentaConstraint (higher_arity/shared.rs): tuples of assigned entities sharing a bin; the
    oracle replays stored tuples through hash sets, the device counts C(members, arity) per value.  Candidate stream
    with trial scores, committed moves, traced steps, fused solve, evaluate_each."""
    import solverforge_amd as sfa

    bins, sizes, k = _balance(n=36, k=4, seed=13 + arity)
    d = sfa.build_balance(bins, sizes, k, w_pair=2, cap=25, arity=arity)
    o = oracle.Model.balance(k, bins, sizes, w_pair=2, cap=25, arity=arity)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=5, la_size=6, limit=30)
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=30))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()
    gs, gc = d.evaluate_each(0)
    os_, oc = o.evaluate_each()
    assert (np.asarray(gs) == np.asarray(os_)[:, :2]).all() and list(gc) == list(oc)
    om = o.enumerate(0, 2, 99, 3)
    ref, od = o.evaluate_moves(om)
    gm, gsc, gd = d.open_cursor(2, 99, selection_order=3, cap=1 << 16)
    assert (_t(gm) == _t(om)).all() and (gd == od).all() and (gsc == ref[:, :2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(10):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
    d.solve_steps(30)
    o.steps(30)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score()[0] == o.score()[:2]).all()
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_balance(bins, sizes, k, arity=6).calculate_score()
