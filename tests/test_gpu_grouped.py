"""GPU parity tests (through the C ABI): sf_step_decide -- one host-driven local-search step whose cursor is a
GroupedScalarMoveSelector over a candidate-backed ScalarGroup (builder/selector/grouped_scalar.rs:82-176): the provider's output is
ordered, filtered, capped, pulled through acceptor + forager and the pick committed; vs the oracle's grouped_scalar_step.  The policy of
a grouped scalar-only model is DiversifiedLateAcceptance + FirstLastStepScoreImproving without a limit
(runtime/compiler/default_local_search/policy.rs:52-55,63-66)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DLA = 4
COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations", "moves_not_doable"]


def _provider(values, k, rng, n_candidates):
    """A ScalarCandidateProvider: compound moves over the working solution -- pair exchanges, chains, re-colourings -- with empty
    candidates, repeats, illegal values, non-doable ones and double edits of one entity mixed in."""
    n = len(values)
    out = []
    for _ in range(n_candidates):
        kind = int(rng.integers(0, 9))
        a, b, c = (int(x) for x in rng.integers(0, n, 3))
        if kind == 0:
            out.append([])
        elif kind == 1 and out:
            out.append(list(out[int(rng.integers(len(out)))]))  # repeat of an earlier candidate
        elif kind == 2:
            out.append([(a, int(values[a]))])  # changes nothing: not doable
        elif kind == 3:
            out.append([(a, int(rng.integers(0, k))), (a, int(rng.integers(0, k)))])  # two edits on one entity
        elif kind == 4:
            out.append([(a, k + 2)])  # illegal value
        elif kind == 5:
            out.append([(a, int(values[b])), (b, int(values[a]))])  # exchange
        elif kind == 6:
            out.append([(a, int(values[b])), (b, int(values[c])), (c, int(values[a]))])  # rotation
        else:
            out.append([(a, int(rng.integers(-1, k))), (b, int(rng.integers(-1, k)))][: int(rng.integers(1, 3))])
    return out


@pytest.mark.parametrize("model,acceptor,forager,limit,order,cap", [
    ("graph", DLA, 4, 0, 3, 0), ("graph", 1, 0, 5, 4, 12), ("graph", 0, 2, 1, 0, 0), ("bins", DLA, 4, 0, 3, 0), ("bins", 1, 1, 1, 3, 40)])
def test_grouped_selector_steps(oracle, model, acceptor, forager, limit, order, cap):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    rng = np.random.default_rng(5)
    if model == "graph":
        g = datasets.make_graph(60, 200, 5, seed=2)
        g["colors"] = rng.integers(-1, 5, 60).astype(np.int64)
        d = sfa.build_graph_coloring(g)
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        k = 5
    else:
        k = 6
        bins = rng.integers(-1, k, 70).astype(np.int64)
        sizes = rng.integers(1, 9, 70).astype(np.int64)
        d = sfa.build_balance(bins, sizes, k, w_pair=3, cap=25)
        o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=25)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=4, forager=forager, accepted_count_limit=limit, selection_order=order,
                                 random_seed=11))
    o.configure(acceptor=1 if acceptor == DLA else acceptor, la_size=4, forager=forager, limit=limit, selection_order=order, leaves=3, random_seed=11)
    if acceptor == DLA:
        d.configure_diversified(0.05)
        o.configure_diversified(4, 0.05)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    applied = 0
    for step in range(25):
        values = o.get_vars(0, 0)
        assert (d.working_values(0, 0) == values).all(), step
        cands = _provider(values, k, rng, int(rng.integers(0, 90)))
        gk, gs, gf, gsel = d.step_decide(cands, group_name_len=9, max_moves_per_step=cap)
        ok, os_, of, osel = o.step_grouped(cands, group_name_len=9, max_moves_per_step=cap)
        assert (gk == ok).all() and len(gk) == len(ok), step
        assert len(gf) == len(of) and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gsel == osel, step
        applied += gsel >= 0
        assert (d.calculate_score()[0] == o.score()[:2]).all(), step
    assert applied > 3
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    assert (d.best_scores()[0] == o.best_score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c
    # the fused engine continues the same search state (history, step index, seed draws)
    d.solve_steps(10)
    o.steps(10)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("model,acceptor,forager,limit", [("graph", 1, 0, 6), ("graph", 0, 2, 1), ("bins", DLA, 4, 0)])
def test_gated_candidates(oracle, model, acceptor, forager, limit):
    """evaluate_candidate's gates (phase/localsearch/evaluation.rs:75-113): candidates that require a hard improvement
    (hard_score_delta == Improving, phase/hard_delta.rs) or a score improvement are scored and counted but never reach the acceptor
    -- the conflict-repair candidates of the runtime provider cursor carry the first, multi-swaps the second."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    rng = np.random.default_rng(8)
    if model == "graph":
        g = datasets.make_graph(50, 170, 4, seed=3)
        g["colors"] = rng.integers(-1, 4, 50).astype(np.int64)
        d = sfa.build_graph_coloring(g)
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        k = 4
    else:
        k = 5
        bins = rng.integers(-1, k, 60).astype(np.int64)
        sizes = rng.integers(1, 9, 60).astype(np.int64)
        d = sfa.build_balance(bins, sizes, k, w_pair=3, cap=25)
        o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=25)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=4, forager=forager, accepted_count_limit=limit, selection_order=3, random_seed=4))
    o.configure(acceptor=1 if acceptor == DLA else acceptor, la_size=4, forager=forager, limit=limit, selection_order=3, leaves=3, random_seed=4)
    if acceptor == DLA:
        d.configure_diversified(0.05)
        o.configure_diversified(4, 0.05)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    gated_rejections = 0
    for step in range(20):
        values = o.get_vars(0, 0)
        cands = _provider(values, k, rng, int(rng.integers(1, 80)))
        gates = rng.integers(0, 4, len(cands)).astype(np.int32)
        gk, gs, gf, gsel = d.step_decide(cands, group_name_len=5, gates=gates)
        ok, os_, of, osel = o.step_grouped(cands, group_name_len=5, gates=gates)
        assert len(gk) == len(ok) and (gk == ok).all(), step
        assert len(gf) == len(of) and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gsel == osel, step
        gated_rejections += int(((gf & 3) == 1).sum())
        assert (d.calculate_score()[0] == o.score()[:2]).all(), step
    assert gated_rejections > 0
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c


def test_step_decide_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p)
    d.calculate_score()
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.step_decide([[(0, 1)]])  # list model
    g = datasets.make_graph(20, 40, 4, seed=1)
    g["colors"] = np.zeros(20, dtype=np.int64)
    d = sfa.build_graph_coloring(g)
    d.configure(sfa.SolverConfig(acceptor=3))
    d.calculate_score()
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.step_decide([[(0, 1)]])  # SimulatedAnnealing is not carried by the host-driven step
    d.configure(sfa.SolverConfig(acceptor=1))
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.step_decide([[(i % 20, 1) for i in range(9)]])  # more than 8 edits
    kept, sc, fl, sel = d.step_decide([])  # an empty provider: the step still ends
    assert len(kept) == 0 and sel == -1 and d.stats(0)["step_count"] == 1
