"""GPU parity tests (through the C ABI) of uni filters / weights as DATA (sf_constraint_add_uni_program, round 6):
for_each(A).filter(pred).penalize(w) with both closures given as small programs over fact columns
(crates/solverforge-scoring/src/constraint/incremental.rs:19-160).  The library compiles each program on the host into the
(entity, value) cost matrix the value-cost paths price; here the same programs are restated in numpy (an independent check of that
compile step) and the folded matrix is handed to the oracle's value-keyed incremental node (the check of everything behind it):
scores, per-constraint rows, the whole candidate streams with trial scores, host-driven moves and compound candidates, committed
moves, traced and fused steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F_SKILL, F_NEED, F_PREF, F_IDENT, F_TABLE, F_COST = 10, 11, 12, 13, 14, 15


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["b"], moves["value"]], axis=1)


def _facts(n=50, k=8, seed=31):
    from solverforge_amd import datasets

    r = datasets.stream(seed, 3 * n + k + 25 + n * k)
    f = {"values": (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1,
         "skill": (r[n:2 * n] % np.uint64(5)).astype(np.int32),
         "pref": (r[2 * n:3 * n] % np.uint64(k + 2)).astype(np.int32) - 2,  # -2 / -1: no preference
         "need": (r[3 * n:3 * n + k] % np.uint64(5)).astype(np.int32),
         "ident": np.arange(k, dtype=np.int32),
         "table": (r[3 * n + k:3 * n + k + 25] % np.uint64(4)).astype(np.int64).reshape(5, 5)}
    f["table"][f["table"] == 1] = 0
    cost = (r[3 * n + k + 25:] % np.uint64(6)).astype(np.int64).reshape(n, k)
    cost[cost < 4] = 0
    f["cost"] = cost
    return f, n, k


def _programs():
    """(terms, weight, scale) per program; numpy twins in _numpy_matrix."""
    import solverforge_amd as sfa

    L, C = sfa.UniLhs, sfa.UniCmp
    return [
        # under-skilled: skill[a] < need[v], costs 3 x (need - skill)
        ([(L.COL_DIFF, C.LT, 0, F_SKILL, F_NEED, -1, 0)], (L.COL_ABSDIFF, F_SKILL, F_NEED, -1), 3),
        # not the preferred value: pref[a] >= 0 AND value != pref[a]  (value as the identity column), constant weight, scale 2
        ([(L.ROW_COL, C.GE, 0, F_PREF, -1, -1, 0), (L.COL_DIFF, C.NE, 1, F_PREF, F_IDENT, -1, 0)], (L.ONE, -1, -1, -1), 2),
        # table[skill][need] != 0 OR value == 0 (one clause), weighted by the table entry
        ([(L.TABLE, C.NE, 0, F_SKILL, F_NEED, F_TABLE, 0), (L.VALUE, C.EQ, 0, -1, -1, -1, 0)], (L.TABLE, F_SKILL, F_NEED, F_TABLE), 1),
        # no filter at all: every assigned entity pays its value's need
        ([], (L.VALUE_COL, F_NEED, -1, -1), 1),
    ]


def _numpy_matrices(f, n, k):
    a = np.arange(n)[:, None]
    v = np.arange(k)[None, :]
    skill, need, pref, tab = f["skill"][a], f["need"][v], f["pref"][a], f["table"]
    m1 = np.where(skill < need, 3 * np.abs(skill - need), 0)
    m2 = np.where((pref >= 0) & (pref != v), 2, 0) * np.ones((n, k), dtype=np.int64)
    te = tab[skill, need]
    m3 = np.where((te != 0) | (v == 0), np.maximum(te, 0), 0)
    m4 = need * np.ones((n, 1), dtype=np.int64)
    p3 = ((te != 0) | (v == 0)) * np.ones((n, k), dtype=bool)
    passes = [skill < need, ((pref >= 0) & (pref != v)) * np.ones((n, k), dtype=bool), p3, np.ones((n, k), dtype=bool)]
    return [m.astype(np.int64) for m in (m1, m2, m3, m4)], passes


HARD_PROGRAM = 1  # index of the program that sits on the hard level in the two-level cases


def _build(f, n, k, with_matrix, n_replicas=1, two_levels=False):
    import solverforge_amd as sfa
    from solverforge_amd.director import ConstraintKind, GpuScoreDirector, SelectorKind

    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, k, True, f["values"])
    for fid, name in ((F_SKILL, "skill"), (F_NEED, "need"), (F_PREF, "pref"), (F_IDENT, "ident")):
        d.add_fact_column_i32(fid, f[name])
    d.add_fact_matrix(F_TABLE, f["table"])
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    if with_matrix:
        d.add_fact_matrix(F_COST, f["cost"])
        d.add_constraint(ConstraintKind.VALUE_COST, 0, fact=F_COST, level=1, weight=5)
    for i, (terms, weight, scale) in enumerate(_programs()):
        d.add_uni_program(0, terms, weight, level=0 if (two_levels and i == HARD_PROGRAM) else 1, scale=scale)
    d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


@pytest.mark.parametrize("with_matrix,two_levels", [(False, False), (True, False), (False, True), (True, True)])
def test_uni_programs_compile_to_the_value_cost_matrix(oracle, with_matrix, two_levels):
    """two_levels: one program is a HARD filter (level 0) beside the soft ones -- the class's programs fold into one matrix per level (at most two)."""
    import solverforge_amd as sfa

    f, n, k = _facts()
    mats, passes = _numpy_matrices(f, n, k)
    soft = [m for i, m in enumerate(mats) if not (two_levels and i == HARD_PROGRAM)]
    folded = sum(soft) + (5 * f["cost"] if with_matrix else 0)
    d = _build(f, n, k, with_matrix, two_levels=two_levels)
    o = oracle.Model.assignment(f["values"], folded, k, cost_weight=1, ex_level=-1, cost2=mats[HARD_PROGRAM] if two_levels else None,
                                cost2_level=0 if two_levels else -1)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.fresh_score()[:2]).all()
    # per-constraint rows: each program (and the matrix) on its own, from the numpy twins
    gs, gc = d.evaluate_each()
    vals = np.asarray(f["values"])
    on = vals >= 0
    rows = ([(5 * f["cost"], f["cost"] != 0)] if with_matrix else []) + list(zip(mats, passes))
    assert len(gs) == 1 + len(rows)
    hard_row = (1 if with_matrix else 0) + HARD_PROGRAM if two_levels else -1
    for i, (m, p) in enumerate(rows):
        lv = 0 if i == hard_row else 1
        assert gs[1 + i][lv] == -int(m[np.flatnonzero(on), vals[on]].sum()) and gs[1 + i][1 - lv] == 0, i
        assert gc[1 + i] == int(p[np.flatnonzero(on), vals[on]].sum()), i
    oe = o.evaluate_each()[0]
    assert gs[1:, 1].sum() == oe[1, 1] and (not two_levels or gs[1:, 0].sum() == oe[2, 0])
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
    cands = [[(int(e), int(v)) for e, v in zip(rng.integers(0, n, m), rng.integers(-1, k, m))] for m in rng.integers(1, 9, 200)]
    cs, cd = d.evaluate_candidates(cands)
    ocs, ocd = o.evaluate_compound(cands)
    assert (cd == ocd).all() and (cs == ocs[:, :2]).all()
    o.configure(leaves=bits, random_seed=5, la_size=6, limit=40)
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=40))
    for it in range(8):
        om = o.enumerate(0, it, 7 + it, 3)
        _, od = o.evaluate_moves(om)
        mv = om[np.flatnonzero(od)[rng.integers(int(od.sum()))]]
        o.apply_move(mv)
        d.apply_move(mv)
        assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(10):
        gm, gsc, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, osc, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gsc == osc[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(50)
    o.steps(50)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score() == d.calculate_score()).all()


def test_uni_program_validation():
    import solverforge_amd as sfa
    from solverforge_amd.director import ConstraintKind, GpuScoreDirector, SelectorKind

    L, C = sfa.UniLhs, sfa.UniCmp
    f, n, k = _facts()

    def fresh():
        d = GpuScoreDirector(score_levels=2, hard_levels=1)
        d.add_entity_class(0, n)
        d.add_scalar_variable(0, 0, k, True, f["values"])
        d.add_fact_column_i32(F_SKILL, f["skill"])
        d.add_fact_column_i32(F_NEED, f["need"])
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
        return d

    d = fresh()
    with pytest.raises(sfa.SolverForgeError):  # unknown operand
        d.add_uni_program(0, [(9, C.EQ, 0, -1, -1, -1, 0)], level=1)
    with pytest.raises(sfa.SolverForgeError):  # clause ids must ascend
        d.add_uni_program(0, [(L.VALUE, C.EQ, 1, -1, -1, -1, 0), (L.VALUE, C.EQ, 0, -1, -1, -1, 1)], level=1)
    with pytest.raises(sfa.SolverForgeError):  # the operand needs its fact
        d.add_uni_program(0, [(L.ROW_COL, C.EQ, 0, -1, -1, -1, 0)], level=1)
    d.add_uni_program(0, [(L.ROW_COL, C.EQ, 0, 77, -1, -1, 0)], level=1)  # a fact that does not exist: found at initialize
    with pytest.raises(sfa.SolverForgeError):
        d.calculate_score()
    d2 = fresh()
    d2.add_uni_program(0, [(L.VALUE_COL, C.GT, 0, F_SKILL, -1, -1, 0)], level=1)  # fine: the column is long enough to be indexed by a value
    d2.add_uni_program(0, [], level=0)  # a second level: fine too (one folded matrix per level)
    d2.calculate_score()
    d3 = GpuScoreDirector(score_levels=3, hard_levels=1)
    d3.add_entity_class(0, n)
    d3.add_scalar_variable(0, 0, k, True, f["values"])
    d3.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    for lv in (0, 1, 2):  # a third level: more than the two matrices a class carries
        d3.add_uni_program(0, [], level=lv)
    with pytest.raises(sfa.SolverForgeError):
        d3.calculate_score()
