"""GPU parity tests (through the C ABI): ValueSource::EntitySlice -- per-entity value lists of a scalar variable
(builder/context/scalar/variable.rs:138-151, sf_schema_set_value_lists) vs the oracle: the change stream draws from the entity's own
list, a swap needs each value in the other row's list (cursor/swap.rs:103-123), compound edits check their entity's list; scalar
engine, generic engine (configured union, mixed job shop), traced + fused steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["b"], moves["value"]], axis=1)


def _lists(n, k, seed, allow_empty=True):
    rng = np.random.default_rng(seed)
    out = []
    for e in range(n):
        m = int(rng.integers(0 if allow_empty else 1, k + 1))
        out.append([int(v) for v in rng.permutation(k)[:m]])  # unsorted on purpose: the list order is the canonical order
    return out


def _graph(oracle, n=120, e=500, k=7, seed=3, union=None):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(n, e, k, seed=seed)
    lists = _lists(n, k, seed + 1)
    colors = np.full(n, -1, dtype=np.int64)
    rng = np.random.default_rng(seed + 2)
    for i in range(n):  # a start that respects the lists (some nodes unassigned)
        if lists[i] and rng.random() < 0.8:
            colors[i] = lists[i][int(rng.integers(len(lists[i])))]
    g["colors"] = colors
    d = sfa.build_graph_coloring(g)
    d.set_value_lists(0, 0, lists)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    o.set_value_lists(lists)
    if union is not None:
        d.configure_union(*union)
    return d, o, lists


@pytest.mark.parametrize("union", [None, (1, None), (4, [2, 1])])
def test_value_lists_streams_and_steps(oracle, union):
    """union None = scalar engine; a configured union sends the same model through the generic engine."""
    import solverforge_amd as sfa

    d, o, lists = _graph(oracle, union=union)
    kw = {} if union is None else {"union_order": union[0]}
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3, 4):
        o.configure(leaves=3, random_seed=5, la_size=6, limit=40, selection_order=order, **kw)
        if union is not None and union[1]:
            o.set_union_weights(union[1])
        gm, gs, gd = d.open_cursor(2, 99, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 2, 99, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all(), order
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        ch = om[om["kind"] == 0]
        assert all(int(m["value"]) == -1 or int(m["value"]) in lists[int(m["a"])] for m in ch)  # only listed values are drawn
    o.configure(leaves=3, random_seed=5, la_size=6, limit=40, **kw)
    if union is not None and union[1]:
        o.set_union_weights(union[1])
    d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=40))
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(60)
    o.steps(60)
    vals = d.working_values(0, 0)
    assert (vals == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score()[0] == o.score()[:2]).all()
    assert all(int(v) == -1 or int(v) in lists[i] for i, v in enumerate(vals))  # the search never leaves the lists


def test_value_lists_compound_candidates(oracle):
    d, o, lists = _graph(oracle)
    d.calculate_score()
    rng = np.random.default_rng(9)
    cands = [[(int(e), int(v)) for e, v in zip(rng.integers(0, 120, m), rng.integers(-1, 7, m))] for m in rng.integers(1, 6, 300)]
    gs, gd = d.evaluate_candidates(cands)
    os_, od = o.evaluate_compound(cands)
    assert (gd == od).all() and (gs == os_[:, :2]).all()
    assert 0 < gd.sum() < len(cands)  # some candidates name a value outside their entity's list


def test_value_lists_mixed_jobshop(oracle):
    """The scalar class of a mixed model (generic engine): every operation may only use its own machines."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(10, 5))
    lists = _lists(p["n_ops"], 5, 4, allow_empty=False)
    for i, l in enumerate(lists):  # the constructed start must be legal
        if int(p["machine_idx"][i]) not in l:
            l.append(int(p["machine_idx"][i]))
    d = sfa.build_jobshop(p)
    d.set_value_lists(0, 0, lists)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
    o.set_value_lists(lists)
    bits = 4 | 8 | 1 | 2
    o.configure(leaves=bits, random_seed=6, la_size=7, limit=48)
    d.configure(sfa.SolverConfig(random_seed=6, late_acceptance_size=7, accepted_count_limit=48))
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    d.phase_start()
    o.phase_start()
    for step in range(15):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om) and (gf == of).all() and (gs == os_[:, :3]).all(), step
    d.solve_steps(40)
    o.steps(40)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all() and d.working_lists(1, 0) == o.get_lists(1)
    assert (d.calculate_score()[0] == o.score()[:3]).all()


def test_value_lists_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(10, 20, 4, seed=1)
    d = sfa.build_graph_coloring(g)
    with pytest.raises(sfa.SolverForgeError):
        d.set_value_lists(0, 0, [[0, 9]] * 10)  # value outside 0..n_values
    with pytest.raises(sfa.SolverForgeError):
        d.set_value_lists(0, 1, [[0]] * 10)  # no such variable
