"""GPU parity tests (through the C ABI): HIP list-variable hot path vs the CPU oracle.
Bit-exact: integer HardSoftScore, candidate order, accept flags, applied moves, counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_ENGINE = {"value": 0}


# The block engine (one 1024-thread workgroup per replica) is kept for models whose replica does not fit a wave's LDS slice; AUTO
# never picks it at any BASELINE config.  It stays parity-covered by the traced-step, fused multi-replica
# and CVRP-5000 tests; everything else runs on the wave engine only.
BLOCK_ALSO = {"test_traced_steps_match_oracle", "test_fused_solve_matches_oracle_multi_replica", "test_cvrp_5000_properties"}


def pytest_generate_tests(metafunc):
    if "engine" in metafunc.fixturenames:
        metafunc.parametrize("engine", ["wave", "block"] if metafunc.function.__name__ in BLOCK_ALSO else ["wave"], indirect=True)


@pytest.fixture(autouse=True)
def engine(request):
    _ENGINE["value"] = {"block": 1, "wave": 2}[request.param]
    yield request.param
    _ENGINE["value"] = 0


def _mk(oracle, problem, n_replicas=1, max_nearby=20, leaves=("nearby_change", "nearby_swap")):
    import solverforge_amd as sfa

    d = sfa.build_cvrp(problem, n_replicas=n_replicas, max_nearby=max_nearby, leaves=leaves)
    d.set_engine(_ENGINE["value"])
    o = oracle.Model.cvrp(problem["capacity"], problem["depot"], problem["demands"], problem["matrix"],
                          problem["customers"], problem["routes"])
    bits = 0
    if "nearby_change" in leaves:
        bits |= oracle.LEAF_NEARBY_LIST_CHANGE
    if "nearby_swap" in leaves:
        bits |= oracle.LEAF_NEARBY_LIST_SWAP
    return d, o, bits


def _tuples(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"]], axis=1)


def _small(seed=1, n=40, v=5, cap=30):
    from solverforge_amd import datasets

    return datasets.make_cvrp(n, v, cap, seed=seed)


def test_initialize_and_fresh_score_match_oracle(oracle):
    p = _small()
    d, o, _ = _mk(oracle, p)
    s = d.calculate_score()
    assert s.shape == (1, 2)
    assert (s[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()


@pytest.mark.parametrize("order", [0, 3, 4])  # Original, Random, Shuffled
@pytest.mark.parametrize("leaves", [("nearby_change",), ("nearby_swap",), ("nearby_change", "nearby_swap")])
def test_cursor_order_and_trial_scores(oracle, order, leaves):
    """Candidate stream (seeded order, stable nearby top-k, union scheduling) and every trial score."""
    p = _small(seed=2, n=57, v=6)
    d, o, bits = _mk(oracle, p, leaves=leaves)
    o.configure(leaves=bits, max_nearby=20, selection_order=order)
    d.calculate_score()
    for step_index, step_seed in [(0, 0), (7, 41), (123, 0xDEADBEEFCAFEF00D)]:
        gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order)
        om = o.enumerate(0, step_index, step_seed, order)
        assert len(gm) == len(om) > 0
        assert (_tuples(gm) == _tuples(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all()
        assert (gs == os_[:, :2]).all()


def test_step_evaluate_arbitrary_moves(oracle):
    """sf_step_evaluate on host-provided ListChange/ListSwap batches (full plain neighbourhoods)."""
    p = _small(seed=3, n=30, v=4)
    p["routes"][1] = []  # an empty route: insertion into / removal down to empty
    p["routes"][0] = p["routes"][0] + [c for c in range(1, 31) if c % 4 == 2]
    seen = set()
    p["routes"] = [[c for c in r if not (c in seen or seen.add(c))] for r in p["routes"]]
    d, o, _ = _mk(oracle, p)
    d.calculate_score()
    o.configure(leaves=oracle.LEAF_LIST_CHANGE | oracle.LEAF_LIST_SWAP)
    moves = np.concatenate([o.enumerate(oracle.LEAF_LIST_CHANGE), o.enumerate(oracle.LEAF_LIST_SWAP)])
    assert len(moves) > 1000
    os_, od = o.evaluate_moves(moves)
    gs, gd = d.evaluate_moves(moves)
    assert (gd == od).all()
    assert (gs == os_[:, :2]).all()
    # not-doable coordinates are reported, not scored
    bad = moves[:3].copy()
    bad["a_pos"] = 10_000
    _, bd = d.evaluate_moves(bad)
    assert (bd == 0).all()


def test_apply_matches_oracle(oracle):
    p = _small(seed=4, n=35, v=5)
    d, o, bits = _mk(oracle, p)
    o.configure(leaves=bits)
    d.calculate_score()
    rng = np.random.default_rng(0)
    for it in range(40):
        om = o.enumerate(0, it, 99 + it, 3)
        mv = om[rng.integers(len(om))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("acceptor,forager,limit", [(1, 0, 256), (0, 0, 4), (1, 1, 1), (0, 2, 1),
                                                    # improving foragers (forager/improving.rs): best-ever / last-step, limit 0 = None
                                                    (1, 3, 0), (1, 4, 0), (1, 4, 6), (0, 4, 3)])
def test_traced_steps_match_oracle(oracle, acceptor, forager, limit):
    """Per step: consumed candidates in order, scores, accept flags, the committed move, then state."""
    import solverforge_amd as sfa

    p = _small(seed=5, n=64, v=7, cap=40)
    d, o, bits = _mk(oracle, p)
    o.configure(acceptor=acceptor, la_size=7, forager=forager, limit=limit, leaves=bits, random_seed=11)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=7, forager=forager,
                                 accepted_count_limit=limit, random_seed=11))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    n_steps = 12 if forager in (2, 3) else 40
    for step in range(n_steps):
        gm, gs, gf, gap, gmv = d.solve_step_traced()
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_tuples(gm) == _tuples(om)).all(), step
        assert (gf == of).all(), step
        assert (gs == os_[:, :2]).all(), step
        assert gap == oap, step
        if gap:
            assert tuple(gmv)[:5] == tuple(omv)[:5], step
        assert d.working_lists(0, 0) == o.get_lists(0), step
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.best_scores()[0] == o.best_score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
              "score_calculations", "moves_not_doable"]:
        assert gst[k] == ost[k], k


def test_fused_solve_matches_oracle_multi_replica(oracle):
    """sf_solve_steps: many steps per launch, several replicas (seed + r), explicit step seeds too."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(150, 12, 70, seed=6)
    R = 3
    d, _, bits = _mk(oracle, p, n_replicas=R)
    d.configure(sfa.SolverConfig(random_seed=5))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(30)
    d.solve_steps(45)  # state carries across launches (LA history index, step counters)
    scores = d.calculate_score()
    best = d.best_scores()
    for r in range(R):
        o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(leaves=bits, random_seed=5 + r)
        o.phase_start()
        o.steps(75)
        assert (scores[r] == o.score()[:2]).all(), r
        assert (best[r] == o.best_score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        assert gst["moves_evaluated"] == ost["moves_evaluated"], r
        assert gst["moves_accepted"] == ost["moves_accepted"], r
    assert (d.fresh_score() == scores).all()  # FullAssert: incremental == full recalculation
    # the best snapshot is a real solution with the best score
    assert sum(len(r_) for r_ in d.working_lists(0, 0, best=True)) == 150


def test_explicit_step_seeds(oracle):
    import solverforge_amd as sfa

    p = _small(seed=8, n=48, v=6)
    d, o, bits = _mk(oracle, p)
    seeds = np.array([3, 1 << 63, 77, 0, 123456789, 42, 42, 9], dtype=np.uint64)
    o.configure(leaves=bits, random_seed=1)
    o.set_step_seeds(seeds)
    d.configure(sfa.SolverConfig(random_seed=1))
    d.set_step_seeds(seeds)
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    d.solve_steps(8)
    o.steps(8)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_unreachable_legs_and_ties(oracle):
    """UNREACHABLE / negative matrix entries: excluded from nearby (meters.rs:21-23) and priced
    MAX_SAFE_LEG_COST by distance_cost (problem_data.rs:28-31); many equal distances exercise the
    stable tie order."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(40, 5, 30, seed=9, coord_range=6)  # tiny grid => massive distance ties
    m = p["matrix"]
    m[3, 7] = np.iinfo(np.int64).max
    m[7, 3] = -5
    m[0, 12] = np.iinfo(np.int64).max
    d, o, bits = _mk(oracle, p)
    o.configure(leaves=bits, random_seed=2)
    d.configure(sfa.SolverConfig(random_seed=2))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for step_index, step_seed in [(0, 5), (3, 6)]:
        gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=3)
        om = o.enumerate(0, step_index, step_seed, 3)
        assert (_tuples(gm) == _tuples(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gs == os_[:, :2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(20)
    o.steps(20)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_cvrp_1000_properties(oracle):
    """BASELINE.json size (1000 customers / 100 vehicles): size-independent checks — the element
    multiset is preserved, incremental == full recalculation, best >= start, and the first steps
    equal the oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(1000, 100, 55, seed=0)
    d, o, bits = _mk(oracle, p, n_replicas=4)
    d.configure(sfa.SolverConfig(random_seed=0))
    start = d.calculate_score().copy()
    assert (start[0] == o.score()[:2]).all()
    d.phase_start()
    d.solve_steps(60)
    o.configure(leaves=bits, random_seed=0)
    o.phase_start()
    o.steps(60)
    sc = d.calculate_score()
    assert (sc[0] == o.score()[:2]).all()
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.fresh_score() == sc).all()
    for r in range(4):
        routes = d.working_lists(0, r)
        assert sorted(c for rt in routes for c in rt) == list(range(1, 1001))
        b = d.best_scores()[r]
        assert tuple(b) >= tuple(start[r])


def test_degenerate_ties_wider_than_a_wave(oracle):
    """All customers at one point: every distance group is wider than a 64-lane chunk, so the
    wave engine's exact serial top-k fallback (and the block engine's row scan) must reproduce the
    stable enumeration-order tie break."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(150, 9, 60, seed=12, coord_range=1)
    assert int(p["matrix"].max()) == 0
    d, o, bits = _mk(oracle, p)
    d.configure(sfa.SolverConfig(random_seed=4))
    d.calculate_score()
    for step_index, step_seed in [(0, 1), (5, 99)]:
        for order in (0, 3):
            o.configure(leaves=bits, random_seed=4, selection_order=order)
            gm, gs, gd = d.open_cursor(step_index, step_seed, selection_order=order)
            om = o.enumerate(0, step_index, step_seed, order)
            assert len(gm) == len(om) > 0
            assert (_tuples(gm) == _tuples(om)).all()
    o.configure(leaves=bits, random_seed=4)
    d.phase_start()
    o.phase_start()
    d.solve_steps(10)
    o.steps(10)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("max_nearby", [1, 5, 64])
def test_max_nearby_extremes(oracle, max_nearby):
    import solverforge_amd as sfa

    p = _small(seed=13, n=90, v=8, cap=50)
    d, o, bits = _mk(oracle, p, max_nearby=max_nearby)
    o.configure(leaves=bits, max_nearby=max_nearby, random_seed=3)
    d.configure(sfa.SolverConfig(random_seed=3))
    d.calculate_score()
    gm, gs, gd = d.open_cursor(2, 17, selection_order=3)
    om = o.enumerate(0, 2, 17, 3)
    assert (_tuples(gm) == _tuples(om)).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(15)
    o.steps(15)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_cvrp_5000_properties(oracle):
    """BASELINE config 5 size (5000 customers / 500 vehicles, one portfolio member per replica):
    both engines at the large size — the element multiset is preserved, incremental == full
    recalculation, and the first steps equal the oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(5000, 500, 55, seed=0)
    d, o, bits = _mk(oracle, p, n_replicas=2)
    d.configure(sfa.SolverConfig(random_seed=0))
    start = d.calculate_score().copy()
    assert (start[0] == o.score()[:2]).all()
    d.phase_start()
    d.solve_steps(12)
    o.configure(leaves=bits, random_seed=0)
    o.phase_start()
    o.steps(12)
    sc = d.calculate_score()
    assert (sc[0] == o.score()[:2]).all()
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.fresh_score() == sc).all()
    for r in range(2):
        routes = d.working_lists(0, r)
        assert sorted(c for rt in routes for c in rt) == list(range(1, 5001))
    if _ENGINE["value"] == 2:  # the layout BASELINE config 5 runs on: node -> slot tables in HBM, internal node numbering
        assert d.wave_layout() == (6, True)


@pytest.mark.parametrize("size", [(300, 30, 55, 8), (1000, 100, 55, 4)])
def test_internal_node_numbering_is_invisible(oracle, monkeypatch, size):
    """ListModel::perm: with SF_AMD_RENUMBER=1 the COMPACT wave kernel searches on nodes renumbered along a nearest-neighbour chain
    (u16 matrix, neighbour index, demands, depot; the lists are renamed on the way into LDS and back out).  Nothing may change at
    the boundary: scores, lists, best solutions and counters of every replica equal the run without it, and replica 0 equals the oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    if _ENGINE["value"] != 2:
        pytest.skip("wave engine only")
    n, v, cap, reps = size
    p = datasets.make_cvrp(n, v, cap, seed=7)
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SF_AMD_RENUMBER", flag)
        d, o, bits = _mk(oracle, p, n_replicas=reps)
        d.configure(sfa.SolverConfig(random_seed=3))
        d.calculate_score()
        d.phase_start()
        d.solve_steps(40)
        d.solve_steps(25)  # a second launch: the lists went out and came back in under the caller's ids
        mode, renum = d.wave_layout()
        assert mode >= 3, mode  # the COMPACT slice (the only kernels that take the numbering)
        assert renum == (flag == "1")
        runs[flag] = (d.calculate_score().copy(), [d.working_lists(0, r) for r in range(reps)], d.best_scores().copy(),
                      [d.working_lists(0, r, best=True) for r in range(reps)], [d.stats(r) for r in range(reps)])
        assert (d.fresh_score() == runs[flag][0]).all()
        if flag == "1":
            o.configure(leaves=bits, random_seed=3)
            o.phase_start()
            o.steps(65)
            assert (runs[flag][0][0] == o.score()[:2]).all()
            assert runs[flag][1][0] == o.get_lists(0)
            so = o.stats()
            assert runs[flag][4][0]["moves_evaluated"] == so["moves_evaluated"]
        d.close()
    a, b = runs["1"], runs["0"]
    assert (a[0] == b[0]).all() and a[1] == b[1] and (a[2] == b[2]).all() and a[3] == b[3]
    for sa_, sb_ in zip(a[4], b[4]):
        assert {k: sa_[k] for k in sa_ if k != "sources_scanned"} == {k: sb_[k] for k in sb_ if k != "sources_scanned"}


@pytest.mark.parametrize("renumber", ["0", "1"])
def test_node_table_in_hbm(oracle, monkeypatch, renumber):
    """Launch mode 6 (SF_AMD_NODE_GLOBAL=1 forces it at a small size; CVRP-5000 takes it by itself): the replica's node -> slot table in
    HBM instead of its LDS slice.  Fused multi-launch runs of several replicas == the oracle / the LDS-table run, with and without the
    internal node numbering."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    if _ENGINE["value"] != 2:
        pytest.skip("wave engine only")
    monkeypatch.setenv("SF_AMD_RENUMBER", renumber)
    p = datasets.make_cvrp(400, 40, 55, seed=9)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SF_AMD_NODE_GLOBAL", flag)
        d, o, bits = _mk(oracle, p, n_replicas=6)
        d.configure(sfa.SolverConfig(random_seed=5))
        d.calculate_score()
        d.phase_start()
        for n in (30, 30, 15):
            d.solve_steps(n)
        mode, renum = d.wave_layout()
        assert (mode in (6, 7)) == (flag == "1") and mode >= 3, mode  # (7: the 64-register build, 32 replicas per CU, small models)
        assert renum == (renumber == "1")
        out[flag] = (d.calculate_score().copy(), [d.working_lists(0, r) for r in range(6)], d.best_scores().copy(), [d.stats(r)["moves_evaluated"] for r in range(6)])
        assert (d.fresh_score() == out[flag][0]).all()
        if flag == "1":
            o.configure(leaves=bits, random_seed=5)
            o.phase_start()
            o.steps(75)
            assert (out[flag][0][0] == o.score()[:2]).all() and out[flag][1][0] == o.get_lists(0)
            assert out[flag][3][0] == o.stats()["moves_evaluated"]
        d.close()
    assert (out["1"][0] == out["0"][0]).all() and out["1"][1] == out["0"][1] and (out["1"][2] == out["0"][2]).all() and out["1"][3] == out["0"][3]


@pytest.mark.parametrize("customers,vehicles,coord_range", [(40, 1, 1000), (60, 2, 1000), (90, 3, 1000), (120, 7, 1000), (300, 129, 1000), (150, 9, 1)])
def test_route_ranks_by_arithmetic_small_and_tied(oracle, monkeypatch, customers, vehicles, coord_range):
    """Launch mode 6 computes a route's rank in the leaf's entity order -- ((route - start) x stride^-1) mod V -- instead of reading a per-step table (round 6,
    csrc/sf_list_wave.hip: RouteArith).  Edge cases of that arithmetic against the oracle: one / two / three lists (stride 1, the inverse search's first round),
    more lists than one round of the search holds, and the all-ties instance, whose serial top-k fallback packs the rank-based ordinal behind a wider shift."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    if _ENGINE["value"] != 2:
        pytest.skip("wave engine only")
    monkeypatch.setenv("SF_AMD_NODE_GLOBAL", "1")
    p = datasets.make_cvrp(customers, vehicles, 55 if vehicles > 1 else 10_000, seed=21, coord_range=coord_range)
    d, o, bits = _mk(oracle, p, n_replicas=3)
    d.configure(sfa.SolverConfig(random_seed=8))
    o.configure(leaves=bits, random_seed=8)
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for n in (12, 13):
        d.solve_steps(n)
    o.steps(25)
    mode, _ = d.wave_layout()
    assert mode == 6, mode
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score() == d.calculate_score()).all()
    assert d.stats(0)["moves_evaluated"] == o.stats()["moves_evaluated"]


def test_internal_node_numbering_with_unreachable_and_tied_legs(oracle, monkeypatch):
    """The numbering chain follows the presorted index: unreachable legs end a row early (the chain falls back to the lowest unvisited
    id) and equal distances keep the index's (distance, EXTERNAL id) order, so the degenerate-tie path (which reads the i64 matrix
    under the caller's ids) still sees the reference's enumeration order."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    if _ENGINE["value"] != 2:
        pytest.skip("wave engine only")
    monkeypatch.setenv("SF_AMD_RENUMBER", "1")
    p = datasets.make_cvrp(260, 26, 60, seed=5, coord_range=4)  # a 4 x 4 grid of points: every distance group is wide
    d, o, bits = _mk(oracle, p, n_replicas=3)
    d.configure(sfa.SolverConfig(random_seed=11))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(30)
    o.configure(leaves=bits, random_seed=11)
    o.phase_start()
    o.steps(30)
    mode, renum = d.wave_layout()
    if mode >= 3:  # (a model the COMPACT slice does not take runs without the numbering: nothing to check beyond parity)
        assert renum
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert d.working_lists(0, 0) == o.get_lists(0)


def test_single_level_score_model():
    """A SoftScore-style (1 level) list model runs on the 2-level kernels with a padded zero level:
    incremental == full recalculation, the distance only goes down under HillClimbing, and the level-0
    value equals the soft level of the equivalent HardSoft model."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets
    from solverforge_amd.director import ConstraintKind, GpuScoreDirector, SelectorKind

    p = datasets.make_cvrp(80, 6, 1000, seed=5)  # capacity never binds
    dim = p["matrix"].shape[0]
    d1 = GpuScoreDirector(score_levels=1, hard_levels=0, n_replicas=2)
    d1.set_engine(_ENGINE["value"])
    d1.add_entity_class(0, len(p["routes"]))
    d1.add_list_variable(0, p["routes"], element_capacity=len(p["customers"]), element_id_bound=dim)
    d1.add_fact_matrix(0, p["matrix"])
    d1.add_constraint(ConstraintKind.ROUTE_DISTANCE, 0, fact=0, param=int(p["depot"]), level=0, weight=1)
    d1.add_selector(SelectorKind.NEARBY_LIST_CHANGE, 0, max_nearby=10, fact_meter=0)
    d1.add_selector(SelectorKind.NEARBY_LIST_SWAP, 0, max_nearby=10, fact_meter=0)
    d2 = sfa.build_cvrp(p, n_replicas=2, max_nearby=10)
    d2.set_engine(_ENGINE["value"])
    cfg = sfa.SolverConfig(acceptor=sfa.Acceptor.HILL_CLIMBING, accepted_count_limit=8, random_seed=3)
    for d in (d1, d2):
        d.configure(cfg)
    s1, s2 = d1.calculate_score(), d2.calculate_score()
    assert s1.shape == (2, 1) and (s1[:, 0] == s2[:, 1]).all()
    for d in (d1, d2):
        d.phase_start()
        d.solve_steps(60)
    e1, e2 = d1.calculate_score(), d2.calculate_score()
    assert (e1[:, 0] == e2[:, 1]).all() and (e1[:, 0] > s1[:, 0]).all()
    assert (d1.fresh_score() == e1).all()
    assert d1.working_lists(0, 1) == d2.working_lists(0, 1)

