"""GPU tests of the elite migration EXTENSION (sf_portfolio_migrate_local; no reference counterpart, never part of a parity run): the
adopters hold exactly the elites' best solutions afterwards, every cached aggregate is consistent (incremental == fresh score), the
elites and the bystanders are untouched, and the search continues correctly from the adopted states on both list engines."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIX = ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt")


def _rank(best):
    return sorted(range(len(best)), key=lambda r: tuple(-int(v) for v in best[r]))  # stable: ties to the lower index


@pytest.mark.parametrize("leaves", [("nearby_change", "nearby_swap"), SIX, SIX + ("ruin",)])
def test_adopters_take_the_elites_best_solutions(leaves):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(80, 8, 55, seed=5)
    R = 12
    d = sfa.build_cvrp(p, n_replicas=R, leaves=leaves)
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=20, accepted_count_limit=40))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(40)
    best = d.best_scores()
    order = _rank(best)
    n_elite, n_replace = 2, 5
    before_work = [d.working_lists(0, r) for r in range(R)]
    before_best = [d.working_lists(0, r, best=True) for r in range(R)]
    before_scores = d.calculate_score().copy()
    adopted = d.migrate_local(n_elite, n_replace)
    expect = {}
    for i in range(n_replace):
        a, e = order[R - 1 - i], order[i % n_elite]
        if tuple(best[a]) != tuple(best[e]):
            expect[a] = e
    assert adopted == len(expect) > 0
    scores = d.calculate_score()
    for r in range(R):
        if r in expect:
            e = expect[r]
            assert d.working_lists(0, r) == before_best[e] and d.working_lists(0, r, best=True) == before_best[e], r
            assert (scores[r] == best[e]).all() and (d.best_scores()[r] == best[e]).all(), r
        else:  # elites and bystanders: nothing moved
            assert d.working_lists(0, r) == before_work[r] and d.working_lists(0, r, best=True) == before_best[r], r
            assert (scores[r] == before_scores[r]).all(), r
    assert (d.fresh_score() == scores).all()  # the adopted states' aggregates (loads, cached score) are consistent
    steps0 = [d.stats(r)["step_count"] for r in range(R)]
    d.solve_steps(30)
    scores = d.calculate_score()
    assert (d.fresh_score() == scores).all()
    customers = sorted(int(c) for c in p["customers"])
    for r in range(R):
        assert sorted(x for l in d.working_lists(0, r) for x in l) == customers, r
        assert d.stats(r)["step_count"] == steps0[r] + 30
        assert tuple(d.best_scores()[r]) >= tuple(scores[r])
    # adopters of one elite diverge (their step seeds differ)
    groups = {}
    for a, e in expect.items():
        groups.setdefault(e, []).append(a)
    for e, members in groups.items():
        if len(members) > 1:
            assert len({str(d.working_lists(0, a)) for a in members + [e]}) > 1


def test_migration_on_a_precedence_shop_and_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    q = datasets.make_precedence_shop(6, 4, seed=3)
    R = 8
    d = sfa.build_precedence_shop(q, n_replicas=R, leaves=("list_change", "list_swap", "list_reverse"))
    d.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=10, accepted_count_limit=30))
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        d.migrate_local(1, 2)  # before sf_phase_start
    d.phase_start()
    d.solve_steps(25)
    best = d.best_scores().copy()
    adopted = d.migrate_local(1, 4)
    assert adopted >= 1
    top = max(tuple(int(v) for v in s) for s in best)
    assert sum(1 for s in d.calculate_score() if tuple(int(v) for v in s) == top) >= adopted + 1
    assert (d.fresh_score() == d.calculate_score()).all()
    d.solve_steps(20)
    assert (d.fresh_score() == d.calculate_score()).all()
    for bad in ((0, 1), (1, -1), (5, 4)):
        with pytest.raises(sfa.SolverForgeError):
            d.migrate_local(*bad)
    assert d.migrate_local(2, 0) == 0
    g = sfa.build_graph_coloring(datasets.construct_graph(datasets.make_graph(40, 80, 5, seed=2)), n_replicas=4)
    g.configure(sfa.SolverConfig(random_seed=0, acceptor=sfa.Acceptor.LATE_ACCEPTANCE))
    g.calculate_score()
    g.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        g.migrate_local(1, 1)  # scalar model: unsupported
