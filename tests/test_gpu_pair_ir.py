"""GPU parity tests of the pair-predicate programs (sf_constraint_add_pair_join; stream/join_target.rs:28-110 predicate joins with the
filter closure as data).  (1) The three model-shaped kinds are presets of it: graph colouring, N-queens and the job-shop predicate written
out as programs give the SAME scores, candidate streams and fused trajectories as the CPU oracles of those models -- once on the
specialised loops the compile step picks for these programs, once with SF_AMD_IR_INTERPRET=1 on the interpreter.  (2) Programs no preset
covers are checked against a brute-force count over all pairs (scores, trial deltas, incremental == fresh after fused steps)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t4(moves):
    return np.stack([moves["kind"], moves["a"], moves["b"], moves["value"]], axis=1)


def _graph(n=300, e=1500, k=6, seed=3):
    from solverforge_amd import datasets

    g = datasets.make_graph(n, e, k, seed=seed)
    r = datasets.stream(seed + 99, n)
    g["colors"] = (r % np.uint64(k + 1)).astype(np.int64) - 1
    return g


@pytest.fixture(params=["specialised", "interpreted"])
def mode(request, monkeypatch):
    monkeypatch.setenv("SF_AMD_IR_INTERPRET", "1" if request.param == "interpreted" else "0")
    return request.param


def test_graph_colouring_as_a_program(oracle, mode):
    import solverforge_amd as sfa

    g = _graph()
    d = sfa.build_graph_coloring(g, n_replicas=3, pair_ir=True)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=2)
    d.configure(sfa.SolverConfig(random_seed=2))
    s = d.calculate_score()
    assert (s[0] == o.score()[:2]).all() and s[0][0] < 0
    gm, gs, gd = d.open_cursor(7, 41, selection_order=3, cap=1 << 17)
    om = o.enumerate(0, 7, 41, 3)
    assert (_t4(gm) == _t4(om)).all()
    os_, od = o.evaluate_moves(om)
    assert (gd == od).all() and (gs == os_[:, :2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(40)
    d.solve_steps(20)
    o.steps(60)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.working_values(0, 0, 0) == o.get_vars(0, 0)).all()
    assert d.stats(0)["moves_evaluated"] == o.stats()["moves_evaluated"]
    assert (d.fresh_score() == d.calculate_score()).all()


def test_nqueens_as_a_program(oracle, mode):
    import solverforge_amd as sfa

    n = 48
    rows = (np.arange(n) * 7 % (n + 1)).astype(np.int64) - 1
    d = sfa.build_nqueens(rows, pair_ir=True)
    o = oracle.Model.nqueens(rows)
    o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=3)
    d.configure(sfa.SolverConfig(random_seed=3))
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    gm, gs, gd = d.open_cursor(1, 5, selection_order=3, cap=1 << 16)
    om = o.enumerate(0, 1, 5, 3)
    assert (_t4(gm) == _t4(om)).all()
    os_, od = o.evaluate_moves(om)
    assert (gd == od).all() and (gs == os_[:, :2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(40)
    o.steps(40)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score()[0] == o.score()[:2]).all()


def test_jobshop_predicate_as_a_program(oracle, mode):
    """The mixed job shop (scalar class + list class, generic engine): `same job && same machine` as a program."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_jobshop(10, 4)
    n = p["n_ops"]
    r = datasets.stream(5, 3 * n)
    p["machine_idx"] = (r[:n] % np.uint64(5)).astype(np.int64) - 1
    seqs = [[] for _ in range(4)]
    for op in range(n):
        w = int(r[n + op] % np.uint64(6))
        if w < 4:
            seqs[w].append(op)
    p["sequences"] = seqs
    d = sfa.build_jobshop(p, n_replicas=2, pair_ir=True)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
    bits = oracle.LEAF_LIST_CHANGE | oracle.LEAF_LIST_SWAP | oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=4)
    d.configure(sfa.SolverConfig(random_seed=4))
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(30)
    o.steps(30)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.working_values(0, 0, 0) == o.get_vars(0, 0)).all()
    assert d.working_lists(1, 0) == o.get_lists(1)
    assert (d.fresh_score() == d.calculate_score()).all()


# ---- programs no preset covers: brute force over all pairs -----------------------------------------------------------------------
def _holds(prog, cols, csr, table, l, r, vl, vr):
    from solverforge_amd.director import PairOp as P

    clauses = {}
    for op, cl, fact, fact_b, param in prog:
        if op == P.VALUE_EQ:
            h = vl == vr
        elif op == P.VALUE_NE:
            h = vl != vr
        elif op == P.VALUE_ABSDIFF_LE:
            h = abs(vl - vr) <= param
        elif op == P.VALUE_ABSDIFF_EQ_COL:
            h = abs(vl - vr) == abs(cols[fact][l] - cols[fact][r])
        elif op == P.COL_EQ:
            h = cols[fact][l] == cols[fact][r]
        elif op == P.COL_NE:
            h = cols[fact][l] != cols[fact][r]
        elif op == P.COL_LT:
            h = cols[fact][l] < cols[fact][r]
        elif op == P.COL_ABSDIFF_EQ:
            h = abs(cols[fact][l] - cols[fact][r]) == param
        elif op == P.COL_ABSDIFF_LE:
            h = abs(cols[fact][l] - cols[fact][r]) <= param
        elif op == P.CSR_CONTAINS:
            h = r in csr[fact][l] or l in csr[fact][r]
        else:  # TABLE_NONZERO
            h = table[cols[fact_b][l]][cols[fact_b][r]] != 0
        clauses[cl] = clauses.get(cl, False) or bool(h)
    return all(clauses.values())


def _count(prog, cols, csr, table, vals):
    n = len(vals)
    return sum(1 for l, r in itertools.combinations(range(n), 2) if vals[l] >= 0 and vals[r] >= 0 and _holds(prog, cols, csr, table, l, r, int(vals[l]), int(vals[r])))


def _norm(prog):
    return [tuple(t) + (-1, -1, 0)[len(t) - 2:] if len(t) < 5 else tuple(t) for t in prog]


PROGRAMS = {
    # neighbours whose colours are equal OR adjacent (a clause of two terms behind the partner index)
    "csr_and_value_band": lambda P: [(P.CSR_CONTAINS, 0, 10), (P.VALUE_EQ, 1), (P.VALUE_ABSDIFF_LE, 1, -1, -1, 1)],
    # same group, shift distance within 2, different values  (partner index from COL_EQ, two residual clauses)
    "group_and_band_and_ne": lambda P: [(P.COL_EQ, 0, 11), (P.COL_ABSDIFF_LE, 1, 12, -1, 2), (P.VALUE_NE, 2)],
    # no clause to index by: ordered columns and equal values, or a table says the two kinds clash  (dense scan)
    "dense_lt_or_table": lambda P: [(P.COL_LT, 0, 12), (P.VALUE_EQ, 0), (P.TABLE_NONZERO, 1, 13, 11, 0), (P.VALUE_ABSDIFF_LE, 1, -1, -1, 3)],
    # a CSR term that shares its clause: it cannot drive the index, membership is tested per pair
    "csr_in_a_disjunction": lambda P: [(P.CSR_CONTAINS, 0, 10), (P.COL_ABSDIFF_EQ, 0, 12, -1, 1), (P.VALUE_EQ, 1)],
}


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_program_scores_and_deltas_against_brute_force(name):
    import solverforge_amd as sfa
    from solverforge_amd import datasets
    from solverforge_amd.director import ConstraintKind, GpuScoreDirector, PairOp, SelectorKind

    n, k = 70, 6
    g = datasets.make_graph(n, 260, k, seed=8)
    r = datasets.stream(77, 4 * n)
    vals0 = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    group = (r[n:2 * n] % np.uint64(5)).astype(np.int32)
    shift = (r[2 * n:3 * n] % np.uint64(9)).astype(np.int32)
    table = ((np.arange(25).reshape(5, 5) * 7 + 3) % 4 == 0).astype(np.int64)
    adj = [set(int(x) for x in g["adj"][g["adj_off"][i]:g["adj_off"][i + 1]]) for i in range(n)]
    cols, csr = {11: group, 12: shift}, {10: adj}
    prog = _norm(PROGRAMS[name](PairOp))

    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=2)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, k, True, vals0)
    d.add_fact_csr(10, g["adj_off"], g["adj"])
    d.add_fact_column_i32(11, group)
    d.add_fact_column_i32(12, shift)
    d.add_fact_matrix(13, table)
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    d.add_pair_join(0, prog, level=1, weight=3)
    d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    d.configure(sfa.SolverConfig(random_seed=6))

    def expect(vals):
        return [-int((vals < 0).sum()), -3 * _count(prog, cols, csr, table, vals)]

    s = d.calculate_score()
    assert s[0].tolist() == expect(vals0) and s[0][1] < 0
    # every candidate of one step: the trial score equals the brute-force score of the changed assignment
    gm, gs, gd = d.open_cursor(3, 17, selection_order=3, cap=1 << 15)
    assert len(gm) > 100
    for i in range(0, len(gm), 7):
        mv = gm[i]
        v = vals0.copy()
        if mv["kind"] == 0:
            v[mv["a"]] = mv["value"]
        else:
            v[mv["a"]], v[mv["b"]] = vals0[mv["b"]], vals0[mv["a"]]
        if gd[i]:
            assert gs[i].tolist() == expect(v), (i, mv)
    # fused steps: the incremental score stays the score of the state
    d.phase_start()
    d.solve_steps(25)
    sc = d.calculate_score()
    assert (d.fresh_score() == sc).all()
    for rep in range(2):
        assert sc[rep].tolist() == expect(np.asarray(d.working_values(0, 0, rep), dtype=np.int64))
    assert tuple(d.best_scores()[0]) >= tuple(s[0])


def test_validation():
    import solverforge_amd as sfa
    from solverforge_amd.director import GpuScoreDirector, PairOp

    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=1)
    d.add_entity_class(0, 8)
    d.add_scalar_variable(0, 0, 3, True, np.zeros(8, dtype=np.int64))
    with pytest.raises(sfa.SolverForgeError):
        d.add_pair_join(0, [], level=0)  # empty program
    with pytest.raises(sfa.SolverForgeError):
        d.add_pair_join(0, [(PairOp.VALUE_EQ, 1), (PairOp.VALUE_NE, 0)], level=0)  # clause ids must ascend
    with pytest.raises(sfa.SolverForgeError):
        d.add_pair_join(0, [(PairOp.COL_EQ, 0)], level=0)  # the op needs a fact
    with pytest.raises(sfa.SolverForgeError):
        d.add_pair_join(0, [(99, 0)], level=0)  # unknown op
    with pytest.raises(sfa.SolverForgeError):
        d.add_pair_join(0, [(PairOp.VALUE_EQ, 0)] * 9, level=0)  # too many terms
