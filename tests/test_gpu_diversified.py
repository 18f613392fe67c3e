"""DiversifiedLateAcceptanceAcceptor (phase/localsearch/acceptor/diversified_late_acceptance.rs:40-176), the default acceptor of
grouped scalar-only models (runtime/compiler/default_local_search/policy.rs:52-55, with FirstLastStepScoreImproving and no
accepted-count limit): per-candidate traces and fused multi-replica launches of the scalar, wave and generic engines vs the
oracle's acceptor (pinned to the reference's five tests in oracle/test_golden.cpp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations",
            "moves_not_doable"]
DLA = 4


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _balance(n=90, k=7, seed=11):
    from solverforge_amd import datasets

    r = datasets.stream(seed, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    sizes = (r[n:] % np.uint64(9)).astype(np.int64) + 1
    return bins, sizes, k


@pytest.mark.parametrize("tolerance,forager,limit", [(0.01, 4, 0), (0.2, 0, 30), (0.0, 4, 0), (0.5, 2, 1)])
def test_grouped_scalar_model_traced_and_fused(oracle, tolerance, forager, limit):
    """The grouped scalar-only default policy: DLA(400 -> 6 here) + FirstLastStepScoreImproving(None) on the bin model (keyed
    self-join + grouped sum); the tolerance band is wide at these scores, so the third acceptance rule decides candidates."""
    import solverforge_amd as sfa

    bins, sizes, k = _balance()
    R = 3
    d = sfa.build_balance(bins, sizes, k, n_replicas=R, w_pair=3, cap=25)
    d.configure(sfa.SolverConfig(acceptor=DLA, late_acceptance_size=6, forager=forager, accepted_count_limit=limit, random_seed=5))
    d.configure_diversified(tolerance)
    d.calculate_score()
    d.phase_start()
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=25)
    o.configure(acceptor=1, la_size=6, forager=forager, limit=limit, leaves=bits, random_seed=5)
    o.configure_diversified(6, tolerance)
    o.phase_start()
    for step in range(12):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(25)
    d.solve_steps(25)
    scores = d.calculate_score()
    for r in range(R):
        o = oracle.Model.balance(k, bins, sizes, w_pair=3, cap=25)
        o.configure(acceptor=1, la_size=6, forager=forager, limit=limit, leaves=bits, random_seed=5 + r)
        o.configure_diversified(6, tolerance)
        o.phase_start()
        o.steps(62)
        assert (scores[r] == o.score()[:2]).all(), r
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r
        gst, ost = d.stats(r), o.stats()
        for c in COUNTERS:
            assert gst[c] == ost[c], (r, c)
    assert (d.fresh_score() == scores).all()


@pytest.mark.parametrize("leaves", [("nearby_change", "nearby_swap"), ("nearby_change", "nearby_swap", "sublist_change", "list_reverse")])
def test_cvrp_wave_and_generic_engines(oracle, leaves):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    bits_of = {"nearby_change": 16, "nearby_swap": 32, "list_reverse": 64, "sublist_change": 128}
    p = datasets.make_cvrp(80, 7, 70, seed=4)
    R = 2
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, max_nearby=10)
    d.configure(sfa.SolverConfig(acceptor=DLA, late_acceptance_size=5, forager=0, accepted_count_limit=40, random_seed=9))
    d.configure_diversified(0.02)
    d.calculate_score()
    d.phase_start()
    bits = sum(bits_of[x] for x in leaves)

    def mk(seed):
        o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(acceptor=1, la_size=5, forager=0, limit=40, leaves=bits, max_nearby=10, random_seed=seed)
        o.configure_diversified(5, 0.02)
        o.phase_start()
        return o

    o = mk(9)
    for step in range(10):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(40)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(9 + r)
        o.steps(50)
        assert (scores[r] == o.score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        for c in COUNTERS:
            assert gst[c] == ost[c], (r, c)


def test_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p)
    with pytest.raises(sfa.SolverForgeError):
        d.configure(sfa.SolverConfig(acceptor=DLA, late_acceptance_size=0))
    with pytest.raises(sfa.SolverForgeError):
        d.configure_diversified(float("nan"))
    d.set_engine(1)  # the block engine does not carry the acceptor
    d.configure(sfa.SolverConfig(acceptor=DLA, late_acceptance_size=4))
    d.calculate_score()
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.solve_steps(1)
