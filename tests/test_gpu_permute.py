"""GPU parity tests (through the C ABI): the list permute leaf (ListPermuteMoveSelector -- every non-identity permutation of every
window of min..=max consecutive elements; heuristic/selector/list_kernel/permute.rs:22-205, move/list_kernel/permute.rs:22-101) in
the generic N-leaf engine vs the oracle's cursor (pinned to heuristic/selector/tests/list_permute.rs in oracle/test_golden.cpp):
candidate streams with trial scores (generated and host-provided), committed moves, traced and fused steps; a symmetric and an
asymmetric matrix; beside other leaves; on a precedence model."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PERMUTE = 8192
BITS = {"permute": PERMUTE, "nearby_change": 16, "nearby_swap": 32, "list_reverse": 64, "list_change": 4, "list_swap": 8, "sublist_change": 128}


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _cvrp(asym, n=36, v=4, seed=6):
    from solverforge_amd import datasets

    p = datasets.make_cvrp(n, v, 60, seed=seed)
    if asym:
        r = datasets.stream(seed + 5, p["matrix"].size).reshape(p["matrix"].shape)
        p["matrix"] = (p["matrix"] + (r % np.uint64(7)).astype(np.int64)).astype(np.int64)
        np.fill_diagonal(p["matrix"], 0)
        p["matrix"][3, 7] = np.iinfo(np.int64).max
    p["routes"][1] = p["routes"][1][:1]  # a one-element route: no window fits
    return p


@pytest.mark.parametrize("asym", [False, True])
@pytest.mark.parametrize("leaves,window", [(("permute",), (2, 4)), (("permute", "nearby_change", "list_reverse"), (2, 5)), (("permute",), (3, 3))])
def test_streams_scores_and_steps(oracle, asym, leaves, window):
    import solverforge_amd as sfa

    p = _cvrp(asym)
    R = 2
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, max_nearby=8, permute=window)

    def mk(seed):
        o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(leaves=sum(BITS[x] for x in leaves), random_seed=seed, la_size=5, limit=40, max_nearby=8)
        o.set_permute(*window)
        return o

    o = mk(3)
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=5, accepted_count_limit=40))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    bits = sum(BITS[x] for x in leaves)
    for order in (0, 3, 4):
        o.configure(leaves=bits, random_seed=3, la_size=5, limit=40, max_nearby=8, selection_order=order)
        gm, gs, gd = d.open_cursor(4, 123, selection_order=order, cap=1 << 19)
        om = o.enumerate(0, 4, 123, order)
        assert len(gm) == len(om) > 0, order
        assert (_t(gm) == _t(om)).all(), order
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all(), order
        es, ed = d.evaluate_moves(om)  # sf_step_evaluate on host-provided records
        assert (ed == od).all() and (es == os_[:, :2]).all(), order
    assert 9 in set(int(k) for k in _t(gm)[:, 0])
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=40, max_nearby=8)
    rng = np.random.default_rng(1)
    for it in range(5):  # committed permute moves through sf_apply
        mv = o.enumerate(0, it, 9 + it, 3)
        mv = mv[mv["kind"] == 9]
        mv = mv[rng.integers(len(mv))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0), it
        assert (d.calculate_score()[0] == o.score()[:2]).all(), it
        assert (d.fresh_score()[0] == o.score()[:2]).all(), it
    d.phase_start()
    o.phase_start()
    kinds = set()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 19)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
            kinds.add(int(gmv["kind"]))
    d.solve_steps(40)
    o.steps(40)
    scores = d.calculate_score()
    assert (scores[0] == o.score()[:2]).all()
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.fresh_score() == scores).all()
    gst, ost = d.stats(0), o.stats()
    for c in ("step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"):
        assert gst[c] == ost[c], c


def test_permute_on_a_precedence_model(oracle):
    """The leaf the default policy declares for list slots with precedence hooks: trial scores carry the makespan delta."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 4, seed=5)
    leaves = ("permute", "list_change", "list_swap")
    d = sfa.build_precedence_shop(p, leaves=leaves)
    o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
    bits = PERMUTE | 4 | 8
    o.configure(leaves=bits, random_seed=2, la_size=6, limit=30)
    d.configure(sfa.SolverConfig(random_seed=2, late_acceptance_size=6, accepted_count_limit=30))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3):
        o.configure(leaves=bits, random_seed=2, la_size=6, limit=30, selection_order=order)
        gm, gs, gd = d.open_cursor(1, 77, selection_order=order, cap=1 << 19)
        om = o.enumerate(0, 1, 77, order)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == os_[:, :2]).all()
    o.configure(leaves=bits, random_seed=2, la_size=6, limit=30)
    d.phase_start()
    o.phase_start()
    for step in range(10):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 19)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(30)
    o.steps(30)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


def test_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    for bad in ((1, 3), (4, 3), (2, 9)):
        with pytest.raises(sfa.SolverForgeError):
            sfa.build_cvrp(p, leaves=("permute",), permute=bad)
    d = sfa.build_cvrp(p, leaves=("permute",))
    d.calculate_score()
    mv = np.zeros(1, dtype=sfa.director.MOVE_DTYPE)[0]
    mv["kind"], mv["a"], mv["a_pos"], mv["b"], mv["b_pos"], mv["value"] = 9, 0, 0, 0, 3, 0  # rank 0 = identity
    with pytest.raises(sfa.SolverForgeError):
        d.apply_move(mv)
