"""GPU parity tests (through the C ABI) of the generic N-leaf engine vs the CPU oracle:
the mixed job shop (scalar + list variable, 4-leaf StratifiedRandom union, BendableScore<2,1>) and
list models with the plain list change / swap streams."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _jobshop(n_jobs=12, n_machines=5, seed=1):
    """Operations with a partial machine assignment and partially filled machine sequences."""
    from solverforge_amd import datasets

    p = datasets.make_jobshop(n_jobs, n_machines)
    n = p["n_ops"]
    r = datasets.stream(seed, 3 * n)
    p["machine_idx"] = (r[:n] % np.uint64(n_machines + 1)).astype(np.int64) - 1
    seqs = [[] for _ in range(n_machines)]
    for op in range(n):
        where = int(r[n + op] % np.uint64(n_machines + 2))
        if where < n_machines:  # some operations stay unscheduled
            seqs[where].append(op)
    p["sequences"] = seqs
    return p


def _mk_jobshop(oracle, p, n_replicas=1, bendable=True):
    import solverforge_amd as sfa

    d = sfa.build_jobshop(p, n_replicas=n_replicas, bendable=bendable)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=bendable)
    bits = oracle.LEAF_LIST_CHANGE | oracle.LEAF_LIST_SWAP | oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    return d, o, bits


@pytest.mark.parametrize("bendable", [True, False])
def test_jobshop_scores(oracle, bendable):
    p = _jobshop()
    d, o, _ = _mk_jobshop(oracle, p, bendable=bendable)
    L = 3 if bendable else 2
    s = d.calculate_score()
    assert s.shape == (1, L)
    assert (s[0] == o.score()[:L]).all() and (s[0] < 0).any()
    assert (d.fresh_score()[0] == o.fresh_score()[:L]).all()


@pytest.mark.parametrize("order", [0, 3, 4])
def test_jobshop_four_leaf_union_order_and_scores(oracle, order):
    """Union of list change, list swap, scalar change, scalar swap (StratifiedRandom, exhaustion of
    the short scalar streams mid-way) + every trial score, BendableScore<2,1>."""
    p = _jobshop(n_jobs=6, n_machines=4, seed=2)
    d, o, bits = _mk_jobshop(oracle, p)
    o.configure(leaves=bits, selection_order=order)
    d.calculate_score()
    for step_index, step_seed in [(0, 0), (7, 41), (3, 0xDEADBEEFCAFEF00D)]:
        gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, step_index, step_seed, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all()
        assert (gs == os_[:, :3]).all()
        es, ed = d.evaluate_moves(om)  # sf_step_evaluate on a mixed batch
        assert (ed == od).all() and (es == os_[:, :3]).all()


@pytest.mark.parametrize("acceptor,forager,limit", [(1, 0, 32), (0, 0, 3), (1, 1, 1), (1, 3, 0), (1, 4, 0), (1, 4, 9)])
def test_jobshop_traced_steps(oracle, acceptor, forager, limit):
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=8, n_machines=4, seed=3)
    d, o, bits = _mk_jobshop(oracle, p)
    o.configure(acceptor=acceptor, la_size=5, forager=forager, limit=limit, leaves=bits, random_seed=4)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=5, forager=forager,
                                 accepted_count_limit=limit, random_seed=4))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(12 if forager == 3 else 30):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all(), step
        assert (gf == of).all(), step
        assert (gs == os_[:, :3]).all(), step
        assert gap == oap, step
        if gap:
            assert tuple(gmv) == tuple(omv), step
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all(), step
        assert d.working_lists(1, 0) == o.get_lists(1), step
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.fresh_score()[0] == o.score()[:3]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
              "score_calculations", "moves_not_doable"]:
        assert gst[k] == ost[k], k


def test_jobshop_apply_and_fused_multi_replica(oracle):
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=20, n_machines=6, seed=5)
    R = 3
    d, o, bits = _mk_jobshop(oracle, p, n_replicas=R)
    o.configure(leaves=bits)
    d.calculate_score()
    rng = np.random.default_rng(2)
    for it in range(12):  # committed moves of all four kinds through sf_apply
        om = o.enumerate(0, it, 50 + it, 3)
        _, od = o.evaluate_moves(om)
        mv = om[np.flatnonzero(od)[rng.integers(int(od.sum()))]]
        o.apply_move(mv)
        for r in range(R):
            d.apply_move(mv, replica=r)
        assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.fresh_score() == d.calculate_score()).all()
    d.configure(sfa.SolverConfig(random_seed=9))
    d.phase_start()
    d.solve_steps(25)
    d.solve_steps(20)
    sc = d.calculate_score()
    for r in range(R):
        o2 = oracle.Model.jobshop(p["job"], o.get_vars(0, 0), o.get_lists(1), bendable=True)
        o2.configure(leaves=bits, random_seed=9 + r)
        o2.phase_start()
        o2.steps(45)
        assert (sc[r] == o2.score()[:3]).all(), r
        assert (d.working_values(0, 0, r) == o2.get_vars(0, 0)).all(), r
        assert d.working_lists(1, r) == o2.get_lists(1), r
    assert (d.fresh_score() == sc).all()


def test_jobshop_c4_size_properties(oracle):
    """BASELINE config 4 size (500 jobs x 20 machines = 10,000 operations, BendableScore): start from
    the reference example's empty state, incremental == full recalculation, first steps == oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_jobshop(500, 20)
    d, o, bits = _mk_jobshop(oracle, p, n_replicas=2)
    d.configure(sfa.SolverConfig(random_seed=0))
    s0 = d.calculate_score()
    assert (s0[0] == [-10000, -10000, 0]).all()
    d.phase_start()
    d.solve_steps(20)
    o.configure(leaves=bits, random_seed=0)
    o.phase_start()
    o.steps(20)
    sc = d.calculate_score()
    assert (sc[0] == o.score()[:3]).all()
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.fresh_score() == sc).all()


@pytest.mark.parametrize("leaves", [("list_change",), ("list_swap",), ("list_change", "list_swap")])
def test_cvrp_plain_list_leaves(oracle, leaves):
    """Plain (non-nearby) list change / swap streams on a CVRP model: full candidate order, trial
    scores with the distance / capacity deltas, and fused steps."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(26, 4, 30, seed=4)
    p["routes"][2] = []  # an empty route in the stream
    seen = set()
    p["routes"][0] = p["routes"][0] + [c for c in range(1, 27) if c % 4 == 3]
    p["routes"] = [[c for c in r if not (c in seen or seen.add(c))] for r in p["routes"]]
    d = sfa.build_cvrp(p, leaves=leaves)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = (oracle.LEAF_LIST_CHANGE if "list_change" in leaves else 0) | (oracle.LEAF_LIST_SWAP if "list_swap" in leaves else 0)
    d.calculate_score()
    for order in (0, 3, 4):
        o.configure(leaves=bits, selection_order=order, random_seed=6)
        gm, gs, gd = d.open_cursor(5, 77, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 5, 77, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
    o.configure(leaves=bits, random_seed=6)
    d.configure(sfa.SolverConfig(random_seed=6))
    d.phase_start()
    o.phase_start()
    d.solve_steps(25)
    o.steps(25)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("leaves", [("list_reverse",), ("list_change", "list_swap", "list_reverse")])
def test_cvrp_list_reverse_leaf(oracle, leaves):
    """Intra-list reversal (2-opt) stream on an ASYMMETRIC matrix with unreachable legs: every leg
    inside the reversed range changes direction; alone and in a 3-leaf union with list change/swap."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(30, 4, 40, seed=8)
    r = datasets.stream(123, p["matrix"].size).reshape(p["matrix"].shape)
    p["matrix"] = (p["matrix"] + (r % np.uint64(17)).astype(np.int64)).astype(np.int64)  # asymmetric
    np.fill_diagonal(p["matrix"], 0)
    p["matrix"][4, 9] = np.iinfo(np.int64).max
    p["matrix"][11, 2] = -3
    p["routes"][3] = p["routes"][3][:1]  # a one-element route (no reversal there)
    seen = set()
    p["routes"][0] = p["routes"][0] + [c for c in range(1, 31)]
    p["routes"] = [[c for c in rt if not (c in seen or seen.add(c))] for rt in p["routes"]]
    d = sfa.build_cvrp(p, leaves=leaves)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = 0
    for name, bit in [("list_change", oracle.LEAF_LIST_CHANGE), ("list_swap", oracle.LEAF_LIST_SWAP),
                      ("list_reverse", oracle.LEAF_LIST_REVERSE)]:
        if name in leaves:
            bits |= bit
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3, 4):
        o.configure(leaves=bits, selection_order=order, random_seed=3)
        gm, gs, gd = d.open_cursor(4, 1234, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 4, 1234, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == os_[:, :2]).all()
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=30)
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=5, accepted_count_limit=30))
    rng = np.random.default_rng(5)
    rev = o.enumerate(oracle.LEAF_LIST_REVERSE, 0, 9, 3)
    for it in range(6):  # committed reversals through sf_apply
        mv = rev[rng.integers(len(rev))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv)[:5] == tuple(omv)[:5], step
        assert d.working_lists(0, 0) == o.get_lists(0), step
    d.solve_steps(30)
    o.steps(30)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("leaves", [
    ("nearby_change", "nearby_swap", "list_reverse"),
    ("nearby_change", "list_change", "nearby_swap", "list_swap", "list_reverse"),
    ("nearby_swap", "list_reverse"),
])
def test_cvrp_nearby_union_with_other_leaves(oracle, leaves):
    """Nearby list change / swap unioned with the reversal and plain list leaves in the generic
    N-leaf engine (StratifiedRandom over up to five children): cursor order, trial scores, traced
    steps and a fused multi-replica solve against the oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(70, 7, 45, seed=10)
    bitmap = {"nearby_change": oracle.LEAF_NEARBY_LIST_CHANGE, "nearby_swap": oracle.LEAF_NEARBY_LIST_SWAP,
              "list_change": oracle.LEAF_LIST_CHANGE, "list_swap": oracle.LEAF_LIST_SWAP,
              "list_reverse": oracle.LEAF_LIST_REVERSE}
    bits = 0
    for name in leaves:
        bits |= bitmap[name]
    R = 2
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, max_nearby=8)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.calculate_score()
    for order in (0, 3):
        o.configure(leaves=bits, selection_order=order, max_nearby=8, random_seed=2)
        gm, gs, gd = d.open_cursor(6, 4242, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 6, 4242, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
    o.configure(leaves=bits, max_nearby=8, random_seed=2, la_size=9, limit=48)
    d.configure(sfa.SolverConfig(random_seed=2, late_acceptance_size=9, accepted_count_limit=48))
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        assert d.working_lists(0, 0) == o.get_lists(0), step
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(0, 0) == o.get_lists(0)
    sc = d.calculate_score()
    assert (sc[0] == o.score()[:2]).all()
    o2 = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o2.configure(leaves=bits, max_nearby=8, random_seed=3, la_size=9, limit=48)
    o2.phase_start()
    o2.steps(55)
    assert d.working_lists(0, 1) == o2.get_lists(0)
    assert (d.fresh_score() == sc).all()


@pytest.mark.parametrize("leaves,sizes", [
    (("sublist_change",), (1, 3)),
    (("sublist_change",), (2, 4)),
    (("nearby_change", "nearby_swap", "sublist_change", "list_reverse"), (1, 3)),
])
def test_cvrp_sublist_change_leaf(oracle, leaves, sizes):
    """Contiguous sublist relocation (Or-opt): full stream order, trial scores with distance and
    capacity deltas (segment demand), committed moves, traced steps; alone and inside a 4-leaf union
    with the nearby leaves and the reversal."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(36, 5, 40, seed=14)
    p["routes"][1] = []  # empty destination / source shorter than the minimum size
    p["routes"][4] = p["routes"][4][:1]
    seen = set()
    p["routes"][0] = p["routes"][0] + [c for c in range(1, 37)]
    p["routes"] = [[c for c in rt if not (c in seen or seen.add(c))] for rt in p["routes"]]
    bitmap = {"nearby_change": oracle.LEAF_NEARBY_LIST_CHANGE, "nearby_swap": oracle.LEAF_NEARBY_LIST_SWAP,
              "sublist_change": oracle.LEAF_SUBLIST_CHANGE, "list_reverse": oracle.LEAF_LIST_REVERSE}
    bits = 0
    for name in leaves:
        bits |= bitmap[name]
    d = sfa.build_cvrp(p, leaves=leaves, max_nearby=6, sublist_sizes=sizes)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.set_sublist_sizes(*sizes)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3, 4):
        o.configure(leaves=bits, selection_order=order, max_nearby=6, random_seed=7)
        gm, gs, gd = d.open_cursor(9, 31337, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 9, 31337, order)
        assert len(gm) == len(om) > 0
        sub = om["kind"] == 5
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all()
        assert (gm["value"][sub] == om["value"][sub]).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == os_[:, :2]).all()
    o.configure(leaves=bits, max_nearby=6, random_seed=7, la_size=5, limit=40)
    d.configure(sfa.SolverConfig(random_seed=7, late_acceptance_size=5, accepted_count_limit=40))
    rng = np.random.default_rng(9)
    for it in range(8):  # committed relocations through sf_apply
        sm = o.enumerate(oracle.LEAF_SUBLIST_CHANGE, it, 5 + it, 3)
        mv = sm[rng.integers(len(sm))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv)[:5] == tuple(omv)[:5], step
            if omv["kind"] == 5:
                assert gmv["value"] == omv["value"]
        assert d.working_lists(0, 0) == o.get_lists(0), step
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("leaves,sizes", [
    (("sublist_swap",), (1, 3)),
    (("sublist_swap",), (2, 2)),
    (("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse"), (1, 3)),
])
def test_cvrp_sublist_swap_leaf(oracle, leaves, sizes):
    """Contiguous sublist exchange (segments of different sizes, intra and inter list): stream order,
    trial scores, committed moves (offsets shift by the size difference), traced steps; alone and in
    the 5-leaf union nearby change + nearby swap + sublist change + sublist swap + reverse — five of
    the seven leaves of the reference's default list policy."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(34, 5, 40, seed=15)
    p["routes"][2] = []
    p["routes"][3] = p["routes"][3][:1]
    seen = set()
    p["routes"][0] = p["routes"][0] + [c for c in range(1, 35)]
    p["routes"] = [[c for c in rt if not (c in seen or seen.add(c))] for rt in p["routes"]]
    bitmap = {"nearby_change": oracle.LEAF_NEARBY_LIST_CHANGE, "nearby_swap": oracle.LEAF_NEARBY_LIST_SWAP,
              "sublist_change": oracle.LEAF_SUBLIST_CHANGE, "sublist_swap": oracle.LEAF_SUBLIST_SWAP,
              "list_reverse": oracle.LEAF_LIST_REVERSE}
    bits = 0
    for name in leaves:
        bits |= bitmap[name]
    d = sfa.build_cvrp(p, leaves=leaves, max_nearby=6, sublist_sizes=sizes)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.set_sublist_sizes(*sizes)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3, 4):
        o.configure(leaves=bits, selection_order=order, max_nearby=6, random_seed=7)
        gm, gs, gd = d.open_cursor(11, 777, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 11, 777, order)
        assert len(gm) == len(om) > 0
        seg = om["kind"] >= 5
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all()
        assert (gm["value"][seg] == om["value"][seg]).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == os_[:, :2]).all()
    o.configure(leaves=bits, max_nearby=6, random_seed=7, la_size=5, limit=40)
    d.configure(sfa.SolverConfig(random_seed=7, late_acceptance_size=5, accepted_count_limit=40))
    rng = np.random.default_rng(10)
    for it in range(10):  # committed exchanges through sf_apply
        sm = o.enumerate(oracle.LEAF_SUBLIST_SWAP, it, 50 + it, 3)
        mv = sm[rng.integers(len(sm))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm)[:, :5] == _t(om)[:, :5]).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv)[:5] == tuple(omv)[:5], step
            if omv["kind"] >= 5:
                assert gmv["value"] == omv["value"]
        assert d.working_lists(0, 0) == o.get_lists(0), step
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("bendable", [True, False])
def test_jobshop_one_byte_values_path(oracle, bendable):
    """Scalar classes of >= 1024 entities and <= 127 values keep one byte per value in LDS (the i8 instantiation of the
    generic engine: job shop 500 x 20 fits 4 waves per CU instead of 3): traced steps and a fused run against the oracle
    on a 1,100-operation job shop (3-level Bendable and 2-level HardSoft instantiations)."""
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=55, n_machines=20, seed=9)
    assert p["n_ops"] >= 1024
    d, o, bits = _mk_jobshop(oracle, p, n_replicas=2, bendable=bendable)
    lv = 3 if bendable else 2
    o.configure(acceptor=1, la_size=5, forager=0, limit=12, leaves=bits, random_seed=6)
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=5, forager=0, accepted_count_limit=12, random_seed=6))
    assert (d.calculate_score()[0] == o.score()[:lv]).all()
    d.phase_start()
    o.phase_start()
    for step in range(4):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om) and (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :lv]).all(), step
        assert gap == oap and (not gap or tuple(gmv) == tuple(omv)), step
    d.solve_steps(12)
    o.steps(12)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert d.working_lists(1, 0) == o.get_lists(1)
    assert (d.calculate_score()[0] == o.score()[:lv]).all() and (d.fresh_score()[0] == o.score()[:lv]).all()


# ---- a join of the TWO planning classes, both sides moving (cross_bi_incremental/incremental.rs:93-137) ---------------------------
def _mk_owner(oracle, p, n_replicas=1, level=1):
    import solverforge_amd as sfa

    d = sfa.build_jobshop(p, n_replicas=n_replicas, bendable=True, owner_match_level=level)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, owner_match_level=level)
    bits = oracle.LEAF_LIST_CHANGE | oracle.LEAF_LIST_SWAP | oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    return d, o, bits


def test_two_class_join_scores_and_each(oracle):
    p = _jobshop(n_jobs=14, n_machines=5, seed=3)
    d, o, _ = _mk_owner(oracle, p)
    s = d.calculate_score()
    assert (s[0] == o.score()[:3]).all() and s[0][1] < 0
    assert (d.fresh_score()[0] == o.fresh_score()[:3]).all()
    # the join's own row of evaluate_each = the count of operations on a machine that does not schedule them
    mi, seqs = np.asarray(p["machine_idx"]), p["sequences"]
    where = {op: m for m, sq in enumerate(seqs) for op in sq}
    expect = sum(1 for op, m in enumerate(mi) if m >= 0 and where.get(op, -1) != m)
    sc, cnt = d.evaluate_each(0)
    assert cnt[-1] == expect and sc[-1][1] == -expect


@pytest.mark.parametrize("order", [0, 3, 4])
def test_two_class_join_trial_scores_of_every_leaf(oracle, order):
    """Every candidate of the four-leaf union: a scalar move changes the A side's key, a list move the B side's filter."""
    p = _jobshop(n_jobs=6, n_machines=4, seed=2)
    d, o, bits = _mk_owner(oracle, p)
    o.configure(leaves=bits, selection_order=order)
    d.calculate_score()
    for step_index, step_seed in [(0, 0), (7, 41), (3, 0xDEADBEEFCAFEF00D)]:
        gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, step_index, step_seed, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all()
        assert (gs == os_[:, :3]).all()


@pytest.mark.parametrize("acceptor,forager,limit", [(1, 0, 256), (0, 0, 4), (1, 1, 1)])
def test_two_class_join_traced_and_fused_steps(oracle, acceptor, forager, limit):
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=10, n_machines=4, seed=6)
    R = 3
    d, o, bits = _mk_owner(oracle, p, n_replicas=R, level=0)
    cfg = sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=5, forager=forager, accepted_count_limit=limit, random_seed=9)
    d.configure(cfg)
    o.configure(leaves=bits, random_seed=9, acceptor=acceptor, la_size=5, forager=forager, limit=limit)
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(8):  # traced steps of replica 0: pulls, trial scores, flags, the committed move
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all(), step
        assert (gf == of).all() and (gs == os_[:, :3]).all(), step
        assert gap == oap, step
        if gap:
            assert tuple(gmv) == tuple(omv), step
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all() and d.working_lists(1, 0) == o.get_lists(1), step
    sc0 = d.calculate_score()[0]
    assert (sc0 == o.score()[:3]).all()
    # fused: every replica against its own oracle
    d2, _, _ = _mk_owner(oracle, p, n_replicas=R, level=0)
    d2.configure(cfg)
    d2.calculate_score()
    d2.phase_start()
    d2.solve_steps(25)
    d2.solve_steps(15)
    sc = d2.calculate_score()
    for r in range(R):
        o2 = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, owner_match_level=0)
        o2.configure(leaves=bits, random_seed=9 + r, acceptor=acceptor, la_size=5, forager=forager, limit=limit)
        o2.phase_start()
        o2.steps(40)
        assert (sc[r] == o2.score()[:3]).all(), r
        assert (d2.working_values(0, 0, r) == o2.get_vars(0, 0)).all(), r
        assert d2.working_lists(1, r) == o2.get_lists(1), r
        assert d2.stats(r)["moves_evaluated"] == o2.stats()["moves_evaluated"], r
    assert (d2.fresh_score() == sc).all()


def test_two_class_join_host_driven_entry_points(oracle):
    """sf_step_evaluate / sf_apply / the compound-candidate entry points price the join from the move's coordinates (round 6): arbitrary records of
    every kind that moves an element between lists or changes a value, then committed moves, against the oracle's hash-indexed node."""
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=9, n_machines=4, seed=4)
    R = 2
    d, o, bits = _mk_owner(oracle, p, n_replicas=R, level=1)
    o.configure(leaves=bits)
    d.calculate_score()
    rng = np.random.default_rng(11)
    n_ops, V = len(p["job"]), len(p["sequences"])
    for it in range(10):
        lists = o.get_lists(1)
        om = o.enumerate(0, it, 90 + it, 3)  # the four-leaf union's candidates ...
        extra = []  # ... plus segment moves and deliberately bad records
        for _ in range(24):
            a, b = (int(v) for v in rng.choice(V, 2, replace=False))
            la, lb = len(lists[a]), len(lists[b])
            if la >= 1:
                i = int(rng.integers(la))
                e = int(rng.integers(i + 1, min(la, i + 3) + 1))
                extra.append((sfa.MoveKind.SUBLIST_CHANGE, a, i, b, int(rng.integers(lb + 1)), e))
            if la >= 1 and lb >= 1:
                i, j = int(rng.integers(la)), int(rng.integers(lb))
                sa, sb = int(rng.integers(1, min(3, la - i) + 1)), int(rng.integers(1, min(3, lb - j) + 1))
                extra.append((sfa.MoveKind.SUBLIST_SWAP, a, i, b, j, sa | (sb << 16)))
            extra.append((sfa.MoveKind.LIST_CHANGE, a, la + 3, b, 0, -1))  # out of range: not doable on either side
        allm = np.concatenate([np.asarray(om, dtype=sfa.MOVE_DTYPE), np.array(extra, dtype=sfa.MOVE_DTYPE)])
        os_, od = o.evaluate_moves(allm)
        for r in range(R):
            gs, gd = d.evaluate_moves(allm, replica=r)
            assert (gd == od).all(), (it, r)
            assert (gs[od != 0] == os_[od != 0, :3]).all(), (it, r)
        cands = [[(int(rng.integers(n_ops)), int(rng.integers(-1, V))) for _ in range(int(rng.integers(1, 4)))] for _ in range(40)]
        cands.append([(0, 1), (0, 2), (0, -1)])  # the same entity three times: the edits chain
        cs, cd = o.evaluate_compound(cands)
        gs, gd = d.evaluate_candidates(cands)
        assert (gd == cd).all() and (gs[cd != 0] == cs[cd != 0, :3]).all(), it
        if it % 3 == 2:  # commit a compound candidate ...
            c = cands[int(np.flatnonzero(cd)[rng.integers(int(cd.sum()))])]
            o.apply_compound(c)
            for r in range(R):
                d.apply_candidate(c, replica=r)
        else:  # ... or a move of the union / a segment move
            ok = np.flatnonzero(od)
            mv = allm[ok[rng.integers(len(ok))]]
            o.apply_move(mv)
            for r in range(R):
                d.apply_move(mv, replica=r)
        sc = d.calculate_score()
        assert (sc[0] == o.score()[:3]).all() and (sc[1] == sc[0]).all(), it
        assert (d.fresh_score() == sc).all(), it
    d.configure(sfa.SolverConfig(random_seed=3))  # and the fused engine carries on from the host-driven state
    d.phase_start()
    d.solve_steps(10)
    assert (d.fresh_score() == d.calculate_score()).all()


def test_two_class_join_validation():
    import solverforge_amd as sfa

    p = _jobshop(n_jobs=6, n_machines=4, seed=2)
    d = sfa.build_jobshop(p, owner_match_level=1)
    d.calculate_score()
    ruin = np.zeros(1, dtype=sfa.MOVE_DTYPE)
    ruin[0] = (sfa.MoveKind.LIST_RUIN, 0, 0, 0, 0, 0)
    with pytest.raises(sfa.SolverForgeError):  # a ruin's recreate does not price the join, host-driven or fused
        d.evaluate_moves(ruin)
    with pytest.raises(sfa.SolverForgeError):  # construction phases do not either
        d.construct_list_cheapest(1, [0])
    d2 = sfa.build_jobshop(p, owner_match_level=1, leaves=("list_change", "change", "ruin"))
    d2.configure(sfa.SolverConfig(random_seed=0))
    d2.calculate_score()
    d2.phase_start()
    with pytest.raises(sfa.SolverForgeError):  # the ruin leaf's recreate does not price it
        d2.solve_steps(1)
