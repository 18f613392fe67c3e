"""Improving foragers (phase/localsearch/forager/improving.rs:17-227) in the fused multi-step launches of every engine:
many steps per launch, several replicas, GPU == oracle on scores, state and every counter.  (The per-candidate traces of
these foragers are in the traced-step tests of test_gpu_cvrp / test_gpu_scalar / test_gpu_mixed.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations",
            "moves_not_doable"]
CASES = [(1, 3, 0), (1, 4, 0), (1, 4, 256), (0, 4, 5)]  # (acceptor, forager, accepted_count_limit; 0 = None)


@pytest.mark.parametrize("engine", [1, 2])  # block, wave
@pytest.mark.parametrize("acceptor,forager,limit", CASES)
def test_cvrp_fused_improving_foragers(oracle, engine, acceptor, forager, limit):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(120, 10, 70, seed=8)
    R = 3
    d = sfa.build_cvrp(p, n_replicas=R)
    d.set_engine(engine)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=9, forager=forager, accepted_count_limit=limit, random_seed=21))
    d.calculate_score()
    d.phase_start()
    n = 25 if forager == 3 else 60
    d.solve_steps(n // 2)
    d.solve_steps(n - n // 2)
    scores, best = d.calculate_score(), d.best_scores()
    for r in range(R):
        o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.configure(acceptor=acceptor, la_size=9, forager=forager, limit=limit,
                    leaves=oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP, random_seed=21 + r)
        o.phase_start()
        o.steps(n)
        assert (scores[r] == o.score()[:2]).all(), r
        assert (best[r] == o.best_score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        for k in COUNTERS:
            assert gst[k] == ost[k], (r, k)
    assert (d.fresh_score() == scores).all()


@pytest.mark.parametrize("acceptor,forager,limit", CASES)
def test_graph_fused_improving_foragers(oracle, acceptor, forager, limit):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(200, 900, 5, seed=4)
    r0 = datasets.stream(77, 200)
    g["colors"] = (r0 % np.uint64(6)).astype(np.int64) - 1
    R = 3
    d = sfa.build_graph_coloring(g, n_replicas=R)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=6, forager=forager, accepted_count_limit=limit, random_seed=3))
    d.calculate_score()
    d.phase_start()
    n = 20 if forager == 3 else 50
    d.solve_steps(n)
    scores, best = d.calculate_score(), d.best_scores()
    for r in range(R):
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        o.configure(acceptor=acceptor, la_size=6, forager=forager, limit=limit,
                    leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=3 + r)
        o.phase_start()
        o.steps(n)
        assert (scores[r] == o.score()[:2]).all(), r
        assert (best[r] == o.best_score()[:2]).all(), r
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r
        gst, ost = d.stats(r), o.stats()
        for k in COUNTERS:
            assert gst[k] == ost[k], (r, k)


@pytest.mark.parametrize("acceptor,forager,limit", [(1, 4, 256), (1, 3, 0)])
def test_jobshop_fused_improving_foragers(oracle, acceptor, forager, limit):
    """FirstLastStepScoreImproving(256) is the default forager of a precedence (job shop) model: policy.rs:63-71."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(12, 5), seed=2)
    R = 2
    d = sfa.build_jobshop(p, n_replicas=R)
    d.configure(sfa.SolverConfig(acceptor=acceptor, late_acceptance_size=7, forager=forager, accepted_count_limit=limit, random_seed=6))
    d.calculate_score()
    d.phase_start()
    n = 15 if forager == 3 else 40
    d.solve_steps(n)
    scores = d.calculate_score()
    for r in range(R):
        o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
        o.configure(acceptor=acceptor, la_size=7, forager=forager, limit=limit,
                    leaves=oracle.LEAF_LIST_CHANGE | oracle.LEAF_LIST_SWAP | oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP,
                    random_seed=6 + r)
        o.phase_start()
        o.steps(n)
        assert (scores[r] == o.score()[:3]).all(), r
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r
        assert d.working_lists(1, r) == o.get_lists(1), r
        gst, ost = d.stats(r), o.stats()
        for k in COUNTERS:
            assert gst[k] == ost[k], (r, k)
    assert (d.fresh_score() == scores).all()
