"""GPU parity tests (through the C ABI): the consecutive-runs collector (stream/collector/runs.rs) as a grouped constraint on the
shift-scheduling model of examples/minimal-shift-scheduling -- full scores, evaluate_each, change / swap trial scores in cursor
order, committed moves, traced and fused steps (LateAcceptance and SimulatedAnnealing), several replicas, limits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _problem(n_nurses, n_days, per_day, seed, unassigned_every=7):
    from solverforge_amd import datasets

    n = n_days * per_day
    day = np.repeat(np.arange(n_days), per_day).astype(np.int64)
    r = datasets.stream(seed, n)
    nurse = (r % np.uint64(n_nurses)).astype(np.int64)
    nurse[::unassigned_every] = -1
    return nurse, day


def _mk(oracle, nurse, day, n_nurses, n_replicas=1, **kw):
    import solverforge_amd as sfa

    d = sfa.build_shift_schedule(nurse, day, n_nurses, n_replicas=n_replicas, **kw)
    o = oracle.Model.shift_schedule(nurse, day, n_nurses, **kw)
    return d, o


def test_complemented_known_answer(oracle):
    """counts 6 / 3 / 0 against the target 4: |6-4| + |3-4| + |0-4| = 7 -- the nurse without shifts is scored (complement)."""
    day = np.array([0, 1, 2, 3, 4, 0, 1, 2, 5, 6])
    nurse = np.array([0, 0, 0, 0, 1, 1, 1, -1, 0, 0])
    d, o = _mk(oracle, nurse, day, 3, limit=2, w_streak=1, count_weight=1, target=4)
    got = d.calculate_score()[0]
    assert got.tolist() == [-1, -9] and (o.score()[:2] == got).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all() and gc[3] == 3


def test_known_answer(oracle):
    """nurse 0 works days {0,1,2,3,5,6}: runs [0..3] (excess 2 over the limit 2) and [5,6]; nurse 1 days {0,1,4}: no excess."""
    day = np.array([0, 1, 2, 3, 4, 0, 1, 2, 5, 6])
    nurse = np.array([0, 0, 0, 0, 1, 1, 1, -1, 0, 0])
    d, o = _mk(oracle, nurse, day, 3, limit=2, w_streak=1, count_weight=1)
    got = d.calculate_score()[0]
    assert got.tolist() == [-1, -47] and (o.score()[:2] == got).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all()
    assert gs[2].tolist() == [0, -2]


@pytest.mark.parametrize("n_nurses,n_days,per_day,limit,w,cw,target", [(3, 10, 2, 2, 1, 0, -1), (5, 14, 3, 3, 7, 2, -1), (8, 28, 4, 0, 1, 0, -1),
                                                                        (4, 30, 1, 5, 3, 1, -1), (6, 14, 2, 2, 1, 1, 4), (9, 7, 3, 1, 2, 3, 0)])
def test_scores_cursor_order_and_trial_scores(oracle, n_nurses, n_days, per_day, limit, w, cw, target):
    nurse, day = _problem(n_nurses, n_days, per_day, seed=n_days)
    if target == 4:
        nurse[nurse == 5] = 2  # a nurse without shifts: the complement scores |0 - target| for her
    d, o = _mk(oracle, nurse, day, n_nurses, limit=limit, w_streak=w, count_weight=cw, target=target)
    got = d.calculate_score()[0]
    assert (got == o.score()[:2]).all()
    assert (d.fresh_score()[0] == got).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all()
    for order in (0, 3, 4):
        o.configure(leaves=3, selection_order=order)
        gm, gsc, gd = d.open_cursor(4, 99, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 4, 99, order)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
        osc, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gsc == osc[:, :2]).all()
        es, ed = d.evaluate_moves(om)  # sf_step_evaluate
        assert (ed == od).all() and (es == osc[:, :2]).all()


@pytest.mark.parametrize("acceptor,target", [("late", -1), ("anneal", -1), ("late", 12)])
def test_apply_traced_and_fused_steps(oracle, acceptor, target):
    """target >= 0: the example's own constraint set (schedule.rs:21-83) -- the complemented |count - target| workload and the
    `required && unassigned` filter (every third shift is optional, some required ones weigh 2)."""
    import solverforge_amd as sfa

    nurse, day = _problem(5, 21, 3, seed=3)
    kw = {}
    if target >= 0:
        req = np.ones(len(nurse), dtype=np.int64)
        req[::3] = 0
        req[1::5] = 2
        kw["required"] = req
    d, o = _mk(oracle, nurse, day, 5, limit=2, w_streak=2, count_weight=1, target=target, **kw)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    if target >= 0:
        gs, gc = d.evaluate_each()
        os_, oc = o.evaluate_each()
        assert (gs == os_[:, :2]).all() and (gc == oc).all()
    rng = np.random.default_rng(1)
    o.configure(leaves=3, selection_order=3)
    for it in range(8):  # committed changes and swaps through sf_apply
        mv = o.enumerate(0, it, 7 + it, 3)
        sc, do = o.evaluate_moves(mv)
        mv = mv[do != 0]
        mv = mv[mv["kind"] == it % 2]
        mv = mv[rng.integers(len(mv))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.calculate_score()[0] == o.score()[:2]).all(), it
        assert (d.fresh_score()[0] == o.score()[:2]).all(), it
    if acceptor == "late":
        o.configure(leaves=3, random_seed=5, la_size=9, limit=30)
        d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=9, accepted_count_limit=30))
    else:
        o.configure(leaves=3, random_seed=5, acceptor=3, forager=0, limit=1)
        o.configure_annealing(mode=2, seed=5)
        d.configure(sfa.SolverConfig(random_seed=5, acceptor=sfa.Acceptor.SIMULATED_ANNEALING, accepted_count_limit=1))
        d.configure_annealing(mode=2, seed=5)
    d.phase_start()
    o.phase_start()
    for step in range(25):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
    d.solve_steps(300)
    o.steps(300)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


def test_multi_replica_and_limits(oracle):
    import solverforge_amd as sfa

    nurse, day = _problem(6, 28, 2, seed=9)
    R = 6
    d, _ = _mk(oracle, nurse, day, 6, n_replicas=R, limit=3, w_streak=1, count_weight=1)
    d.calculate_score()
    d.configure(sfa.SolverConfig(random_seed=40, late_acceptance_size=15, accepted_count_limit=24))
    d.phase_start()
    d.solve_steps(200)
    got = d.calculate_score()
    for r in (0, 2, 5):
        o = oracle.Model.shift_schedule(nurse, day, 6, limit=3, w_streak=1, count_weight=1)
        o.configure(leaves=3, random_seed=40 + r, la_size=15, limit=24)
        o.phase_start()
        o.steps(200)
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r
        assert (got[r] == o.score()[:2]).all(), r
    assert (d.fresh_score() == got).all()
    # limits: negative points, a table beyond the LDS budget, compound candidates
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_shift_schedule(nurse, day - 1, 6).calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_shift_schedule(np.zeros(10, dtype=np.int64), np.arange(10) * 400, 100).calculate_score()


# ---- indexed_presence collector (stream/collector/indexed_presence.rs) as a grouped constraint: SF_C_PRESENCE_VALUE ------------------
def test_presence_known_answer(oracle):
    """The reference's own case (stream/collector/tests/collector.rs:333-356): points {4, 2, 3, 7, 7} -> count 4, count_in(2..5) 3,
    any_in(7..8); here as one nurse's days, a second nurse with days {0, 5}."""
    day = np.array([4, 2, 3, 7, 7, 0, 5])
    nurse = np.array([0, 0, 0, 0, 0, 1, 1])
    # complement_runs(0..7) of {0, 2, 5} = [1], [3, 4], [6] (collector.rs:358-376): nurse 1 below has {0, 5} -> [1..4], [6]
    for presence, want in [((0, 4096, 0), 4 + 2), ((2, 5, 0), 3 + 0), ((7, 8, 1), 1 + 0), ((5, 8, 1), 1 + 1), ((0, 8, 3), 3 + 2), ((3, 3, 0), 0),
                           ((0, 7, 1, 1), (1 + 1) + 3), ((0, 12, 0, 1), (2 + 2 + 4) + (4 + 6)), ((0, 7, 5, 1), 0)]:
        d, o = _mk(oracle, nurse, day, 2, w_streak=1, presence=presence)
        got = d.calculate_score()[0]
        assert got.tolist() == [-1, -want], presence  # hard: the two shifts of nurse 0 on day 7
        assert (o.score()[:2] == got).all() and (d.fresh_score()[0] == got).all()
        gs, gc = d.evaluate_each()
        os_, oc = o.evaluate_each()
        assert (gs == os_[:, :2]).all() and (gc == oc).all()


@pytest.mark.parametrize("n_nurses,n_days,per_day,presence,w,cw", [(3, 10, 2, (0, 4096, 0), 1, 0), (5, 14, 3, (3, 9, 0), 7, 2), (8, 28, 4, (5, 7, 1), 1, 0),
                                                                    (4, 30, 1, (0, 30, 4), 3, 1), (6, 14, 2, (10, 40, 2), 1, 1), (9, 7, 3, (0, 1, 1), 2, 3),
                                                                    (4, 12, 2, (0, 12, 1, 1), 1, 0), (6, 20, 1, (3, 30, 2, 1), 5, 1), (7, 9, 2, (0, 9, 0, 1), 1, 2),
                                                                    (12, 6, 1, (2, 5, 1, 1), 3, 0)])
def test_presence_scores_cursor_order_and_trial_scores(oracle, n_nurses, n_days, per_day, presence, w, cw):
    nurse, day = _problem(n_nurses, n_days, per_day, seed=n_days + 1)
    d, o = _mk(oracle, nurse, day, n_nurses, w_streak=w, count_weight=cw, presence=presence)
    got = d.calculate_score()[0]
    assert (got == o.score()[:2]).all()
    assert (d.fresh_score()[0] == got).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all()
    for order in (0, 3):
        o.configure(leaves=3, selection_order=order)
        gm, gsc, gd = d.open_cursor(4, 99, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 4, 99, order)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
        osc, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gsc == osc[:, :2]).all()
        es, ed = d.evaluate_moves(om)  # sf_step_evaluate
        assert (ed == od).all() and (es == osc[:, :2]).all()


@pytest.mark.parametrize("presence", [(0, 4096, 0), (4, 15, 3), (6, 13, 1), (0, 21, 1, 1), (5, 40, 0, 1)])
def test_presence_apply_traced_and_fused_steps(oracle, presence):
    import solverforge_amd as sfa

    nurse, day = _problem(5, 21, 3, seed=4)
    d, o = _mk(oracle, nurse, day, 5, w_streak=2, count_weight=1, presence=presence)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    rng = np.random.default_rng(2)
    o.configure(leaves=3, selection_order=3)
    for it in range(8):
        mv = o.enumerate(0, it, 7 + it, 3)
        sc, do = o.evaluate_moves(mv)
        mv = mv[do != 0]
        mv = mv[mv["kind"] == it % 2]
        mv = mv[rng.integers(len(mv))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.calculate_score()[0] == o.score()[:2]).all(), it
        assert (d.fresh_score()[0] == o.score()[:2]).all(), it
    o.configure(leaves=3, random_seed=5, la_size=9, limit=30)
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=9, accepted_count_limit=30))
    d.phase_start()
    o.phase_start()
    for step in range(20):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
    d.solve_steps(300)
    o.steps(300)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()


def test_presence_validation():
    import solverforge_amd as sfa

    nurse, day = _problem(3, 6, 2, seed=1)
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_shift_schedule(nurse, day, 3, presence=(5, 2, 0)).calculate_score()  # lo > hi
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_shift_schedule(nurse, day, 3, presence=(0, 5000, 0)).calculate_score()  # hi > 4096
