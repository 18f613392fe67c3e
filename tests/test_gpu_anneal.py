"""GPU parity tests (through the C ABI): SimulatedAnnealing acceptor on every search engine vs the
CPU oracle (phase/localsearch/acceptor/simulated_annealing.rs).  Candidate order, trial scores,
accept flags, applied moves, counters and the f64 temperatures must agree exactly; the Boltzmann
test itself uses the device exp(), which may differ from glibc by one ulp -- a flag could only
differ if a draw landed within one ulp of the acceptance probability (never observed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SA = 3


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _graph(n=120, e=500, k=5, seed=11):
    from solverforge_amd import datasets

    g = datasets.make_graph(n, e, k, seed=seed)
    r = datasets.stream(seed + 99, n)
    g["colors"] = (r % np.uint64(k + 1)).astype(np.int64) - 1
    return g


def _configure(d, o, oracle, bits, forager, limit, seed, anneal, levels=2, hard_levels=1, max_nearby=20):
    import solverforge_amd as sfa

    o.configure(acceptor=1, la_size=5, forager=forager, limit=limit, leaves=bits, random_seed=seed, max_nearby=max_nearby)
    o.configure_annealing(levels=levels, hard_levels=hard_levels, seed=seed, **anneal)
    d.configure(sfa.SolverConfig(acceptor=SA, forager=forager, accepted_count_limit=limit, random_seed=seed))
    kw = dict(anneal)
    d.configure_annealing(mode=kw.pop("mode", 2), temperatures=kw.pop("temperatures", ()),
                          decay_rate=kw.pop("decay_rate", 0.999985),
                          hill_climbing_temperature=kw.pop("hill_climbing_temperature", 1.0e-9),
                          never_accept_hard_regression=kw.pop("never_accept_hard", False),
                          calibration_sample_size=kw.pop("sample_size", 128),
                          target_acceptance_probability=kw.pop("target_probability", 0.80),
                          fallback_temperature=kw.pop("fallback_temperature", 1.0), seed=seed)
    assert not kw


def _compare_state(d, o, levels=2):
    gt, gc = d.annealing_state(0)
    ot, _, oc = o.annealing_state()
    assert gc == bool(oc)
    assert (gt.view(np.uint64) == ot[:levels].view(np.uint64)).all(), (gt, ot)


def _traced(d, o, steps, levels=2):
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    accepted_worse = 0
    for step in range(steps):
        last = o.score()[:levels].copy()
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 17)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all(), step
        assert (gs == os_[:, :levels]).all(), step
        assert (gf == of).all(), step
        assert gap == oap, step
        if gap:
            assert tuple(gmv[k] for k in gmv.dtype.names) == tuple(omv[k] for k in omv.dtype.names), step
        for sc, fl in zip(os_[:, :levels], of):
            if fl & 2 and tuple(sc) < tuple(last):
                accepted_worse += 1
        _compare_state(d, o, levels)
    assert (d.calculate_score()[0] == o.score()[:levels]).all()
    assert (d.best_scores()[0] == o.best_score()[:levels]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
              "score_calculations", "moves_not_doable"]:
        assert gst[k] == ost[k], k
    return accepted_worse


ANNEAL_CASES = [
    # the scalar-only default policy: auto-calibrated, AcceptedCount(1) (policy.rs:56-77)
    (dict(mode=2), 0, 1),
    (dict(mode=2, sample_size=24, target_probability=0.5), 0, 7),   # calibration completes inside a chunk
    (dict(mode=0, temperatures=(3.0,), decay_rate=0.97), 0, 4),
    (dict(mode=1, temperatures=(0.5, 40.0), decay_rate=0.9, hill_climbing_temperature=2.0), 1, 1),
    (dict(mode=1, temperatures=(50.0, 50.0), never_accept_hard=True), 0, 16),
    (dict(mode=2, sample_size=5, fallback_temperature=2.5), 2, 1),  # BestScore forager: whole neighbourhood
]


@pytest.mark.parametrize("anneal,forager,limit", ANNEAL_CASES)
def test_graph_annealing_traced_steps(oracle, anneal, forager, limit):
    import solverforge_amd as sfa

    g = _graph()
    d = sfa.build_graph_coloring(g, n_replicas=1)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    _configure(d, o, oracle, oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, forager, limit, 17, anneal)
    worse = _traced(d, o, 12 if forager == 2 else 160)
    if anneal.get("mode") == 0 or anneal.get("sample_size", 128) < 30:
        assert worse > 0  # the Boltzmann branch really ran


def test_graph_annealing_fused_matches_oracle(oracle):
    """Fused multi-step launches (no trace): final values, scores, counters, temperatures, per replica seed."""
    import solverforge_amd as sfa

    g = _graph(n=200, e=900, k=6, seed=5)
    R = 3
    d = sfa.build_graph_coloring(g, n_replicas=R)
    d.configure(sfa.SolverConfig(acceptor=SA, forager=0, accepted_count_limit=1, random_seed=40))
    d.configure_annealing(mode=2, calibration_sample_size=40, seed=40)
    d.calculate_score()
    d.phase_start()
    d.solve_steps(250)
    d.solve_steps(150)
    for r in (0, 2):
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        o.configure(acceptor=1, forager=0, limit=1, leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP,
                    random_seed=40 + r)
        o.configure_annealing(mode=2, sample_size=40, seed=40 + r)
        o.phase_start()
        o.steps(400)
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all()
        assert (d.calculate_score()[r] == o.score()[:2]).all()
        assert (d.best_scores()[r] == o.best_score()[:2]).all()
        gt, gc = d.annealing_state(r)
        ot, _, oc = o.annealing_state()
        assert gc == bool(oc) and (gt.view(np.uint64) == ot[:2].view(np.uint64)).all()
        gst, ost = d.stats(r), o.stats()
        for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
            assert gst[k] == ost[k], k


@pytest.mark.parametrize("engine", [1, 2])  # block, wave
@pytest.mark.parametrize("anneal,limit", [(dict(mode=2, sample_size=30), 8), (dict(mode=0, temperatures=(25.0,), decay_rate=0.99), 3)])
def test_cvrp_annealing_traced_steps(oracle, engine, anneal, limit):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(60, 6, 45, seed=9)
    d = sfa.build_cvrp(p, n_replicas=1, max_nearby=10)
    d.set_engine(engine)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    _configure(d, o, oracle, oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP, 0, limit, 23, anneal, max_nearby=10)
    assert _traced(d, o, 60) > 0


def test_generic_engine_annealing_traced_steps(oracle):
    """Union with plain leaves -> the N-leaf engine."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(40, 5, 40, seed=4)
    leaves = ("nearby_change", "list_swap", "list_reverse")
    d = sfa.build_cvrp(p, n_replicas=1, max_nearby=8, leaves=leaves)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_LIST_SWAP | oracle.LEAF_LIST_REVERSE
    _configure(d, o, oracle, bits, 0, 5, 31, dict(mode=2, sample_size=20), max_nearby=8)
    assert _traced(d, o, 50) > 0


def test_annealing_parameter_validation():
    """assert_simulated_annealing_parameters (simulated_annealing.rs:305-336) -> SF_ERR_INVALID."""
    import solverforge_amd as sfa

    d = sfa.build_nqueens([-1] * 8)
    for bad in [dict(decay_rate=0.0), dict(decay_rate=1.5), dict(hill_climbing_temperature=-1.0),
                dict(mode=1, temperatures=(1.0, float("inf"))), dict(calibration_sample_size=0),
                dict(target_acceptance_probability=1.0), dict(fallback_temperature=float("nan"))]:
        with pytest.raises(sfa.SolverForgeError):
            d.configure_annealing(**bad)
    d.configure_annealing()
