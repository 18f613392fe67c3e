"""GPU parity tests (through the C ABI): the list ruin leaf of the generic N-leaf engine vs the CPU oracle --
ruin candidates (source list, count, positions from the SmallRng streams), the greedy-recreate trial score of every
candidate, committed ruins, the seven-leaf default list policy (policy/list.rs:24-33) step by step and fused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEAF_BITS = {"nearby_change": 16, "nearby_swap": 32, "list_change": 4, "list_swap": 8, "list_reverse": 64,
             "sublist_change": 128, "sublist_swap": 256, "kopt": 512, "ruin": 1024}
DEFAULT_POLICY = ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _problem(kind):
    from solverforge_amd import datasets

    if kind == "plain":
        return datasets.make_cvrp(40, 4, 60, seed=3)
    if kind == "tight":  # overloaded routes: the hard level decides most placements
        return datasets.make_cvrp(60, 8, 30, seed=11)
    if kind == "ties":  # every distance equal: the first of equal placements (element, list, position order) must win
        p = datasets.make_cvrp(30, 5, 60, seed=5)
        p["matrix"][:] = 7
        np.fill_diagonal(p["matrix"], 0)
        return p
    if kind == "ragged":  # empty lists, single-element lists, one list longer than a wave
        p = datasets.make_cvrp(110, 7, 500, seed=6)
        allc = [c for rt in p["routes"] for c in rt]
        p["routes"] = [allc[:80], [], allc[80:81], allc[81:84], [], allc[84:100], allc[100:]]
        return p
    if kind == "unreach":  # symmetric matrix with unreachable / negative legs: 16-bit leg tables with sentinels, 64-bit deltas
        p = datasets.make_cvrp(40, 4, 60, seed=3)
        big = np.iinfo(np.int64).max
        for a, b, v in [(4, 9, big), (11, 2, -3), (0, 17, big), (30, 31, big)]:
            p["matrix"][a, b] = p["matrix"][b, a] = v
        return p
    if kind == "asym":  # asymmetric matrix with unreachable / negative legs
        p = datasets.make_cvrp(36, 6, 40, seed=8)
        r = datasets.stream(123, p["matrix"].size).reshape(p["matrix"].shape)
        p["matrix"] = (p["matrix"] + (r % np.uint64(17)).astype(np.int64)).astype(np.int64)
        np.fill_diagonal(p["matrix"], 0)
        p["matrix"][4, 9] = np.iinfo(np.int64).max
        p["matrix"][11, 2] = -3
        return p
    raise ValueError(kind)


def _mk(oracle, p, leaves, ruin=(2, 5, 10), n_replicas=1, seed=0, **cfg):
    import solverforge_amd as sfa

    d = sfa.build_cvrp(p, n_replicas=n_replicas, leaves=leaves, ruin=ruin, max_nearby=10)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = sum(LEAF_BITS[x] for x in leaves)
    o.configure(leaves=bits, random_seed=seed, max_nearby=10, **cfg)
    o.set_ruin(ruin[0], ruin[1], ruin[2], variable_name="visits")
    return d, o


def _check_steps(d, o, n, levels=2, expect_kind=None):
    seen = 0
    for step in range(n):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all(), step
        assert (gs == os_[:, :levels]).all(), step
        assert (gf == of).all(), step
        assert gap == oap, step
        if gap:
            assert tuple(gmv) == tuple(omv), step
            seen += int(gmv["kind"] == 8)
        assert d.working_lists(0, 0) == o.get_lists(0), step
    if expect_kind:
        assert seen > 0
    return seen


@pytest.mark.parametrize("problem,ruin", [("plain", (2, 5, 10)), ("tight", (2, 5, 10)), ("ties", (1, 6, 16)), ("ragged", (2, 5, 10)),
                                          ("asym", (3, 3, 4)), ("plain", (1, 1, 3)), ("unreach", (2, 5, 10))])
@pytest.mark.parametrize("general_path", [False, True])
def test_ruin_only_traced_steps(oracle, problem, ruin, general_path, monkeypatch):
    """One-leaf union: every step pulls `moves_per_step` ruin candidates; candidate identity, trial score, accept flag,
    committed move and the state after every step equal the oracle's; then a fused window.  `general_path` forces the
    matrix-gather recreate that asymmetric / wide matrices take (SF_AMD_NO_LEG16) on every problem."""
    import solverforge_amd as sfa

    if general_path:
        monkeypatch.setenv("SF_AMD_NO_LEG16", "1")

    p = _problem(problem)
    d, o = _mk(oracle, p, ("ruin",), ruin=ruin, seed=4, la_size=5, limit=8)
    d.configure(sfa.SolverConfig(random_seed=4, late_acceptance_size=5, accepted_count_limit=8))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    _check_steps(d, o, 30, expect_kind=True)
    d.solve_steps(60)
    o.steps(60)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


@pytest.mark.parametrize("problem", ["plain", "tight", "ragged", "unreach"])
def test_default_list_policy_seven_leaves(oracle, problem):
    """The reference's whole default list policy: nearby change, nearby swap, sublist change, sublist swap, reverse,
    distance-pruned 3-opt, ruin (StratifiedRandom, LateAcceptance + AcceptedCount)."""
    import solverforge_amd as sfa

    p = _problem(problem)
    d, o = _mk(oracle, p, DEFAULT_POLICY, seed=2, la_size=7, limit=48)
    d.configure(sfa.SolverConfig(random_seed=2, late_acceptance_size=7, accepted_count_limit=48))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    _check_steps(d, o, 25)
    d.solve_steps(80)
    o.steps(80)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


def test_ruin_multi_replica_streams_and_relaunch(oracle):
    """Replica r searches with random_seed + r: its ruin stream is scoped_seed(random_seed + r, ...); the per-solve stream
    survives launch boundaries (several short launches == the oracle's single run)."""
    import solverforge_amd as sfa

    p = _problem("plain")
    R = 5
    d, _ = _mk(oracle, p, DEFAULT_POLICY, n_replicas=R, seed=9)
    d.configure(sfa.SolverConfig(random_seed=9, late_acceptance_size=6, accepted_count_limit=32))
    d.calculate_score()
    d.phase_start()
    for n in (3, 1, 7, 12):
        d.solve_steps(n)
    sc = d.calculate_score()
    fs = d.fresh_score()
    assert (sc == fs).all()
    for r in (0, 2, 4):
        _, o = _mk(oracle, p, DEFAULT_POLICY, seed=9 + r, la_size=6, limit=32)
        o.phase_start()
        o.steps(23)
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all(), r


def test_ruin_step_generate_is_a_dry_run(oracle):
    """sf_step_generate (open_cursor) on a union with a ruin leaf peeks at the leaf's next draw: it neither changes the
    state nor consumes the per-solve stream (the following traced step still equals the oracle's)."""
    import solverforge_amd as sfa

    p = _problem("plain")
    d, o = _mk(oracle, p, ("ruin",), seed=1, la_size=5, limit=8)
    d.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=5, accepted_count_limit=8))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    before = d.working_lists(0, 0)
    gm, gs, gd = d.open_cursor(0, int(oracle.lib().sfo_step_seed(1, 0)), selection_order=3, cap=64)
    assert len(gm) == 10 and (gm["kind"] == 8).all() and gd.all()
    assert d.working_lists(0, 0) == before
    om, os_, of, oap, omv = o.step_traced()
    n = len(om)  # the traced step stops at the forager's limit, the dry run drains the cursor
    assert 0 < n <= 10 and (_t(gm)[:n] == _t(om)).all() and (gs[:n] == os_[:, :2]).all()
    o2 = _mk(oracle, p, ("ruin",), seed=1, la_size=5, limit=8)[1]
    o2.phase_start()
    _check_steps(d, o2, 5)


@pytest.mark.parametrize("problem", ["plain", "tight", "asym", "ragged"])
def test_ruin_moves_through_step_evaluate_and_apply(oracle, problem):
    """Host-provided SF_MOVE_LIST_RUIN records (the MoveSelector plugin surface): sf_step_evaluate scores them one wavefront
    each, also inside a batch with other move kinds; sf_apply commits one; not-doable records are reported."""
    import solverforge_amd as sfa

    p = _problem(problem)
    d, o = _mk(oracle, p, ("nearby_change", "ruin"), seed=6, la_size=5, limit=8)
    d.calculate_score()
    rng = np.random.default_rng(8)
    for it in range(6):
        ruins = o.enumerate(1024, it, 11 + it, 3)  # ten ruin candidates of the oracle's stream
        others = o.enumerate(16, it, 11 + it, 3)[:40]
        batch = np.concatenate([ruins[:5], others, ruins[5:]])
        bad = ruins[:3].copy()
        bad["a_pos"][0] = 0  # no element
        bad["b"][1] = 0xFFFF  # position out of range
        bad["a"][2] = 999  # no such list
        batch = np.concatenate([batch, bad])
        gs, gd = d.evaluate_moves(batch)
        os_, od = o.evaluate_moves(batch)
        assert (gd == od).all(), it
        assert (gs == os_[:, :2]).all(), it
        mv = ruins[int(rng.integers(len(ruins)))]
        d.apply_move(mv)
        o.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0), it
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    with pytest.raises(sfa.SolverForgeError):
        d.apply_move(bad[0])


def test_ruin_selector_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    for bad in [dict(min_ruin_count=0), dict(min_ruin_count=4, max_ruin_count=3), dict(max_ruin_count=7), dict(moves_per_step=17),
                dict(moves_per_step=-1)]:
        d = sfa.build_cvrp(p, leaves=())
        with pytest.raises(sfa.SolverForgeError):
            d.add_ruin_selector(0, **bad)
