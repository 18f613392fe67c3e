"""GPU parity tests (through the C ABI): multi-edit ScalarCandidates (planning/scalar/candidate.rs:85-188) scored and applied
as ONE CompoundScalarMove (heuristic/move/compound_scalar.rs:207-330) vs the CPU oracle -- the ScalarCandidateProvider
plugin surface.  Graph colouring (predicate cross-join), N-queens, bin balance (keyed self-join + grouped sum), mixed job shop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _candidates(rng, n_entities, n_values, n, allow_none=True, max_edits=8):
    """Random candidates: 1..max_edits edits, repeated entities, no-op edits, to-None edits."""
    out = []
    for _ in range(n):
        k = int(rng.integers(1, max_edits + 1))
        ents = rng.integers(0, n_entities, size=k)
        if rng.random() < 0.3 and k > 1:
            ents[-1] = ents[0]  # the same entity edited twice: the later edit wins
        vals = rng.integers(-1 if allow_none else 0, n_values, size=k)
        out.append([(int(e), int(v)) for e, v in zip(ents, vals)])
    return out


def _check(d, o, cands, levels):
    gs, gd = d.evaluate_candidates(cands)
    os_, od = o.evaluate_compound(cands)
    assert (gd == od).all()
    assert (gs == os_[:, :levels]).all()
    return od


def test_compound_graph_coloring(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(300, 1500, 6, seed=3)
    r = datasets.stream(102, 300)
    g["colors"] = (r % np.uint64(7)).astype(np.int64) - 1
    d = sfa.build_graph_coloring(g)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    rng = np.random.default_rng(1)
    cands = _candidates(rng, 300, 6, 400)
    # neighbours edited together, a no-op candidate, an illegal value, an empty candidate, a cancelled edit
    a = 5
    nb = [int(x) for x in g["adj"][g["adj_off"][a]:g["adj_off"][a + 1]][:3]]
    cands += [[(a, 2)] + [(x, 2) for x in nb], [(a, int(g["colors"][a]))], [(a, 6)], [], [(a, 1), (a, int(g["colors"][a]))],
              [(a, -1), (nb[0], -1)]]
    od = _check(d, o, cands, 2)
    assert od[-5] == 0 and od[-4] == 0 and od[-3] == 0  # no-op, illegal value, empty
    assert od[-2] == 1  # an edit that differs from the current value makes the candidate doable even if a later edit cancels it
    for it in range(12):  # committed candidates, then more trials on the new state
        doable = [c for c, ok in zip(cands, od) if ok]
        c = doable[int(rng.integers(len(doable)))]
        d.apply_candidate(c)
        o.apply_compound(c)
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
        cands = _candidates(rng, 300, 6, 100)
        od = _check(d, o, cands, 2)
    with pytest.raises(sfa.SolverForgeError):
        d.apply_candidate([(a, int(d.working_values(0, 0)[a]))])  # not doable
    with pytest.raises(sfa.SolverForgeError):
        d.evaluate_candidates([[(0, 1)] * 9])  # more than 8 edits


def test_compound_nqueens(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    n = 24
    rows = (datasets.stream(5, n) % np.uint64(n + 1)).astype(np.int64) - 1
    d = sfa.build_nqueens(rows)
    o = oracle.Model.nqueens(rows)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    rng = np.random.default_rng(2)
    for it in range(4):
        cands = _candidates(rng, n, n, 200, max_edits=5)
        od = _check(d, o, cands, 2)
        c = [c for c, ok in zip(cands, od) if ok][0]
        d.apply_candidate(c)
        o.apply_compound(c)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("cap,arity", [(-1, 2), (25, 2), (-1, 3)])
def test_compound_value_keyed_tables(oracle, cap, arity):
    """Keyed self-join (pairs / triples sharing a bin) + grouped sum: the candidate's edits shift the per-value tables one
    after the other (several entities entering and leaving the same bin inside one candidate)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    n, k = 80, 7
    r = datasets.stream(11, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    sizes = (r[n:] % np.uint64(9)).astype(np.int64) + 1
    d = sfa.build_balance(bins, sizes, k, w_pair=3, cap=cap, arity=arity)
    o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=cap, arity=arity)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    rng = np.random.default_rng(3)
    for it in range(5):
        cands = _candidates(rng, n, k, 300)
        cands.append([(e, 2) for e in range(8)])  # eight entities into one bin
        od = _check(d, o, cands, 2)
        c = [c for c, ok in zip(cands, od) if ok][int(rng.integers(int(od.sum())))]
        d.apply_candidate(c)
        o.apply_compound(c)
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()


def test_compound_load_balance_is_unsupported():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    n, k = 40, 5
    r = datasets.stream(11, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    sizes = (r[n:] % np.uint64(9)).astype(np.int64) + 1
    d = sfa.build_balance(bins, sizes, k, w_pair=3, cap=-2)
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        d.evaluate_candidates([[(0, 1), (1, 2)]])


def test_compound_on_mixed_jobshop(oracle):
    """Scalar candidates of a mixed model (list class + scalar class share one committed score)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(12, 5))
    d = sfa.build_jobshop(p)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    rng = np.random.default_rng(4)
    for it in range(4):
        cands = _candidates(rng, p["n_ops"], 5, 200, max_edits=6)
        od = _check(d, o, cands, 3)
        c = [c for c, ok in zip(cands, od) if ok][0]
        d.apply_candidate(c)
        o.apply_compound(c)
        assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
        assert (d.calculate_score()[0] == o.score()[:3]).all()
        assert (d.fresh_score()[0] == o.score()[:3]).all()
