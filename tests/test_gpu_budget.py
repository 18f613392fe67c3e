"""sf_solve_moves (work-balanced launches for wall-clock solves): per-replica parity with the CPU oracle."""
import pytest

import solverforge_amd as sfa
from oracle import sfo
from solverforge_amd import datasets


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [0, 1, 2])
def test_move_budgeted_launch_matches_oracle_per_replica(engine):
    """sf_solve_moves: replicas stop after different step counts (each once its own candidates reach the budget); every
    replica is still exactly the oracle's state after the number of steps it reports, across two launches."""
    p = datasets.make_cvrp(70, 7, 45, seed=6)
    leaves = ("nearby_change", "nearby_swap") if engine else ("nearby_change", "nearby_swap", "sublist_change", "list_reverse", "kopt")
    R = 6
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, max_nearby=8, kopt=(1, 4))
    if engine:
        d.set_engine(engine)
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=7, forager=0, accepted_count_limit=16, random_seed=11))
    d.calculate_score()
    d.phase_start()
    bits = {"nearby_change": 16, "nearby_swap": 32, "sublist_change": 128, "list_reverse": 64, "kopt": 512}
    oracles = []
    for r in range(R):
        o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        o.set_kopt(1, 4)
        o.configure(acceptor=1, la_size=7, forager=0, limit=16, leaves=sum(bits[x] for x in leaves), random_seed=11 + r, max_nearby=8)
        o.phase_start()
        oracles.append([o, 0])
    seen = set()
    for launch, budget in enumerate((400, 900)):
        d.solve_moves(10_000, budget)
        for r in range(R):
            st = d.stats(r)
            o, done = oracles[r]
            o.steps(st["step_count"] - done)
            oracles[r][1] = st["step_count"]
            assert d.working_lists(0, r) == o.get_lists(0), (launch, r)
            assert (d.calculate_score()[r] == o.score()[:2]).all(), (launch, r)
            so = o.stats()
            assert st["moves_evaluated"] == so["moves_evaluated"] and st["step_count"] == so["step_count"]
            seen.add(st["step_count"])
    # the launch ended on the budget, not on max_steps, and replicas got different distances
    assert max(seen) < 10_000 and len(seen) > 1
    with pytest.raises(sfa.SolverForgeError):
        d.solve_moves(10, 0)


@pytest.mark.gpu
def test_portfolio_allgather_and_winner_broadcast_single_rank():
    """RCCL communicator of world size 1 on the one GPU of the box: the all-gather names the best replica and the
    ncclBroadcast of its route CSR returns exactly that replica's best solution."""
    p = datasets.make_cvrp(50, 5, 45, seed=8)
    d = sfa.build_cvrp(p, n_replicas=8)
    d.configure(sfa.SolverConfig(random_seed=3))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(40)
    d.portfolio_init(d.portfolio_unique_id(), 0, 1)
    try:
        best, wr, wrep = d.portfolio_allgather_best()
        scores = [tuple(int(v) for v in s) for s in d.best_scores()]
        assert wr == 0 and tuple(int(v) for v in best) == max(scores) == scores[wrep]
        routes = d.portfolio_broadcast_best(wr, wrep)
        assert routes == d.working_lists(0, wrep, best=True)
        assert sorted(c for r in routes for c in r) == sorted(p["customers"])
        with pytest.raises(sfa.SolverForgeError):
            d.portfolio_broadcast_best(1, 0)  # rank outside the communicator
    finally:
        d.portfolio_destroy()


def _rescore_cvrp(p, routes):
    q = dict(p)
    q["routes"] = [list(r) for r in routes]
    d2 = sfa.build_cvrp(q)
    return tuple(int(v) for v in d2.calculate_score()[0])


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [0, 1, 2])
def test_deferred_best_snapshot_is_the_best_solution(engine):
    """update_best_solution is deferred on the device (written when the search leaves a best state or the launch ends):
    after several launches with worsening stretches the downloaded best solution of every replica, re-scored from
    scratch, has exactly the reported best score; replicas whose working score is below their best hold a different
    solution there."""
    p = datasets.make_cvrp(60, 6, 40, seed=12)
    leaves = ("nearby_change", "nearby_swap") if engine else ("nearby_change", "nearby_swap", "sublist_change", "list_reverse")
    R = 6
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves, max_nearby=6)
    if engine:
        d.set_engine(engine)
    # a short LateAcceptance history lets the search drift away from its best state
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=40, forager=0, accepted_count_limit=4, random_seed=21))
    d.calculate_score()
    d.phase_start()
    drifted = 0
    for launch, steps in enumerate((1, 7, 60, 3, 150)):
        d.solve_steps(steps)
        best = d.best_scores()
        work = d.calculate_score()
        for r in range(R):
            routes = d.working_lists(0, r, best=True)
            assert _rescore_cvrp(p, routes) == tuple(int(v) for v in best[r]), (launch, r)
            if tuple(work[r]) < tuple(best[r]):
                assert routes != d.working_lists(0, r)
                drifted += 1
            # (equal scores do not imply equal solutions: an equal-score move leaves the best snapshot untouched)
    assert drifted > 0


@pytest.mark.gpu
def test_deferred_best_snapshot_scalar_engine():
    g = datasets.construct_graph(datasets.make_graph(80, 300, 4, seed=5))
    d = sfa.build_graph_coloring(g, n_replicas=4)
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=30, forager=0, accepted_count_limit=3, random_seed=2))
    d.calculate_score()
    d.phase_start()
    drifted = 0
    for steps in (1, 9, 80, 200):
        d.solve_steps(steps)
        best, work = d.best_scores(), d.calculate_score()
        for r in range(4):
            vals = d.working_values(0, 0, replica=r, best=True)
            g2 = dict(g)
            g2["colors"] = vals.astype("int64")
            assert tuple(int(v) for v in sfa.build_graph_coloring(g2).calculate_score()[0]) == tuple(int(v) for v in best[r])
            drifted += int(tuple(work[r]) < tuple(best[r]))
    assert drifted > 0
