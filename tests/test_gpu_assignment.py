"""GPU parity tests (through the C ABI): the keyed cross-join of a planning class with a fact class (SF_C_VALUE_COST ≙
cross_bi_incremental::Bi keyed by the planning value) and exists / not-exists of planning entities per fact row
(SF_C_EXISTS_VALUE ≙ IncrementalExistsConstraint) vs the oracle's incremental nodes: full scores, per-constraint
evaluate_each, the whole candidate streams with trial scores, committed moves, compound candidates, traced and fused steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["b"], moves["value"]], axis=1)


def _problem(n=60, k=9, seed=21):
    from solverforge_amd import datasets

    r = datasets.stream(seed, n + n * k + k)
    values = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    cost = (r[n:n + n * k] % np.uint64(7)).astype(np.int64)
    cost[cost < 3] = 0  # the join's filter rejects most pairs
    row_w = (r[n + n * k:] % np.uint64(20)).astype(np.int64) + 1
    return values, cost.reshape(n, k), row_w, k


@pytest.mark.parametrize("ex_mode,ex_level,with_w", [(1, 1, True), (0, 0, True), (1, 1, False), (1, -1, False)])
def test_assignment_cross_join_and_exists(oracle, ex_mode, ex_level, with_w):
    import solverforge_amd as sfa

    values, cost, row_w, k = _problem()
    w = row_w if with_w else None
    d = sfa.build_assignment(values, cost, k, cost_weight=3, row_w=w, ex_mode=ex_mode, ex_level=ex_level, ex_weight=5)
    o = oracle.Model.assignment(values, cost, k, cost_weight=3, row_w=w, ex_mode=ex_mode, ex_level=ex_level, ex_weight=5)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all()
    for order in (0, 3):
        o.configure(leaves=bits, random_seed=5, la_size=6, limit=40, selection_order=order)
        gm, gsc, gd = d.open_cursor(2, 99, selection_order=order, cap=1 << 16)
        om = o.enumerate(0, 2, 99, order)
        assert (_t(gm) == _t(om)).all()
        osc, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gsc == osc[:, :2]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == osc[:, :2]).all()
    rng = np.random.default_rng(3)
    cands = []
    for _ in range(300):  # compound candidates: several entities moving between the same rows inside one candidate
        m = int(rng.integers(1, 9))
        cands.append([(int(e), int(v)) for e, v in zip(rng.integers(0, len(values), m), rng.integers(-1, k, m))])
    cs, cd = d.evaluate_candidates(cands)
    ocs, ocd = o.evaluate_compound(cands)
    assert (cd == ocd).all() and (cs == ocs[:, :2]).all()
    o.configure(leaves=bits, random_seed=5, la_size=6, limit=40)
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=40))
    for it in range(10):
        om = o.enumerate(0, it, 7 + it, 3)
        _, od = o.evaluate_moves(om)
        mv = om[np.flatnonzero(od)[rng.integers(int(od.sum()))]]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    c = [c for c, ok in zip(cands, d.evaluate_candidates(cands)[1]) if ok][0]
    d.apply_candidate(c)
    o.apply_compound(c)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gsc, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, osc, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gsc == osc[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(60)
    o.steps(60)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :2]).all() and (gc == oc).all()


def test_assignment_annealing_multi_replica(oracle):
    """SimulatedAnnealing default policy of scalar models on the new constraint kinds, several replicas."""
    import solverforge_amd as sfa

    values, cost, row_w, k = _problem(n=90, k=11, seed=4)
    d = sfa.build_assignment(values, cost, k, n_replicas=4, row_w=row_w, ex_weight=2)
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.SIMULATED_ANNEALING, forager=sfa.Forager.ACCEPTED_COUNT, accepted_count_limit=1,
                                 random_seed=7))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(300)
    sc = d.calculate_score()
    assert (sc == d.fresh_score()).all()
    for r in (0, 3):
        o = oracle.Model.assignment(values, cost, k, row_w=row_w, ex_weight=2)
        o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, forager=oracle.FORAGER_ACCEPTED_COUNT, limit=1,
                    random_seed=7 + r)
        o.configure_annealing(seed=7 + r)
        o.phase_start()
        o.steps(300)
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r
        assert (sc[r] == o.score()[:2]).all(), r


def test_value_cost_and_exists_validation():
    import solverforge_amd as sfa
    from solverforge_amd.director import ConstraintKind

    values, cost, row_w, k = _problem(n=10, k=4)
    d = sfa.build_assignment(values, cost, k, ex_level=-1)
    d.add_constraint(ConstraintKind.EXISTS_VALUE, 0, fact=-1, param=2, level=1, weight=1)  # mode must be 0 / 1
    with pytest.raises(sfa.SolverForgeError):
        d.calculate_score()
    d = sfa.build_assignment(values, cost[:, :3], 3, ex_level=-1)  # matrix narrower than the declared value range is fine here...
    d2 = sfa.GpuScoreDirector(score_levels=2, hard_levels=1)
    d2.add_entity_class(0, 10)
    d2.add_scalar_variable(0, 0, 4, True, values[:10] % 4)
    d2.add_fact_matrix(0, cost[:10, :3])  # ...but a cost matrix must be [n_rows][n_values]
    d2.add_constraint(ConstraintKind.VALUE_COST, 0, fact=0, level=1, weight=1)
    with pytest.raises(sfa.SolverForgeError):
        d2.calculate_score()
