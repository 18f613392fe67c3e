"""GPU parity tests (through the C ABI): Clarke-Wright savings construction on the device (sf_construct_list_clarke_wright ≙
ListClarkeWrightPhase, list_clarke_wright/kernel.rs:59-472 with the stock CVRP hook bundle, solverforge-cvrp/src/helpers.rs) vs the
oracle's general hook form (oracle/sfo_clarke_wright.hpp, pinned to the reference's own tests): constructed lists, committed score,
the committed / untouched verdict; structural and capacity feasibility, partial states, ties, unreachable / negative legs, an
asymmetric matrix, completion by savings insertion (succeeding and failing), an over-capacity customer, negative demands; then
local search from the constructed state; C3 size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(kind):
    from solverforge_amd import datasets

    if kind == "plain":
        p = datasets.make_cvrp(60, 6, 55, seed=3)
    elif kind == "infeasible":  # total demand above the fleet: completion finds no insertion -> lists untouched
        p = datasets.make_cvrp(80, 8, 30, seed=11)
    elif kind == "ties":
        p = datasets.make_cvrp(30, 5, 60, seed=5)
        p["matrix"][:] = 7
        np.fill_diagonal(p["matrix"], 0)
    elif kind == "asym":
        p = datasets.make_cvrp(36, 6, 40, seed=8)
        r = datasets.stream(123, p["matrix"].size).reshape(p["matrix"].shape)
        p["matrix"] = (p["matrix"] + (r % np.uint64(17)).astype(np.int64)).astype(np.int64)
        np.fill_diagonal(p["matrix"], 0)
        p["matrix"][4, 9] = np.iinfo(np.int64).max
        p["matrix"][11, 2] = -3
        p["matrix"][0, 7] = np.iinfo(np.int64).max
    elif kind == "completion":  # more savings routes than vehicles, completed by insertion
        p = datasets.make_cvrp(60, 6, 54, seed=10)
    elif kind == "completion_fails":
        p = datasets.make_cvrp(60, 6, 50, seed=0)
    elif kind == "ragged":  # 130 customers: partial last chunk of 64 in every scan
        p = datasets.make_cvrp(130, 9, 90, seed=21)
    elif kind == "over_capacity_customer":  # a singleton no owner can take blocks every merge (route_state.rs:100-103)
        p = datasets.make_cvrp(40, 45, 55, seed=4)
        p["demands"][17] = 99
    elif kind == "negative_demand":  # loads can shrink: a merge of the second pass exists (3 passes in the oracle), nothing is skipped
        p = datasets.make_cvrp(30, 12, 8, seed=41)
        rng = np.random.default_rng(41)
        neg = rng.choice(np.arange(1, 31), 12, replace=False)
        p["demands"][neg] = -rng.integers(1, 9, 12)
    else:
        raise ValueError(kind)
    return p


CASES = [
    ("plain", 0, 0), ("plain", 1, 0), ("infeasible", 0, 0), ("infeasible", 1, 0), ("ties", 0, 0), ("ties", 1, 0), ("asym", 0, 0),
    ("asym", 1, 0), ("completion", 1, 0), ("completion_fails", 1, 0), ("ragged", 1, 0), ("ragged", 0, 3), ("plain", 1, 2),
    ("plain", 0, 5), ("over_capacity_customer", 1, 0), ("over_capacity_customer", 0, 0), ("negative_demand", 1, 0), ("plain", 1, 6),
]


@pytest.mark.parametrize("problem,mode,keep", CASES)
def test_clarke_wright_matches_oracle(oracle, problem, mode, keep):
    """keep = how many of the round-robin start routes stay filled: only the missing customers are routed, onto the empty
    vehicles (kernel.rs:80-91); keep = all vehicles -> no available slot, untouched."""
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    d = sfa.build_cvrp(p, n_replicas=3)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    placed = {c for rt in p["routes"] for c in rt}
    missing = [int(c) for c in p["customers"] if int(c) not in placed]
    sc, flags = d.construct_list_clarke_wright(0, p["customers"], mode)  # the elements already in a list are not routed
    committed, st = o.construct_list_clarke_wright(missing, mode)
    for r in range(3):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
        assert bool(flags[r]) == committed
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    if problem == "completion":
        assert committed and st[4] > 0
    if problem in ("completion_fails", "infeasible") and mode == 1:
        assert not committed and st[4] > 0
    if problem == "negative_demand":
        assert committed and st[3] == 3


def test_clarke_wright_replicas_with_different_states(oracle):
    """Every replica routes ITS missing customers onto ITS empty vehicles: replica 1 starts from lists changed by six committed
    moves."""
    import solverforge_amd as sfa

    p = _problem("plain")
    p["routes"] = [rt if i < 3 else [] for i, rt in enumerate(p["routes"])]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    os_ = [oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"]) for _ in range(2)]
    os_[1].configure(leaves=oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP, max_nearby=10)
    for it in range(6):
        om = os_[1].enumerate(0, it, 7 + it, 3)
        mv = om[(5 * it + 3) % len(om)]
        os_[1].apply_move(mv)
        d.apply_move(mv, replica=1)
    assert d.working_lists(0, 1) == os_[1].get_lists(0) and d.working_lists(0, 0) != d.working_lists(0, 1)
    sc, flags = d.construct_list_clarke_wright(0, p["customers"], 1)
    for r in range(2):
        lists = os_[r].get_lists(0)
        placed = {c for rt in lists for c in rt}
        missing = [int(c) for c in p["customers"] if int(c) not in placed]
        committed, _ = os_[r].construct_list_clarke_wright(missing, 1)
        assert d.working_lists(0, r) == os_[r].get_lists(0), r
        assert (sc[r] == os_[r].score()[:2]).all()
        assert bool(flags[r]) == committed


def test_clarke_wright_then_local_search(oracle):
    import solverforge_amd as sfa

    p = _problem("plain")
    p["routes"] = [[] for _ in p["routes"]]
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_clarke_wright([int(c) for c in p["customers"]], 1)
    leaves = oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP
    o.configure(leaves=leaves, random_seed=1, la_size=8, limit=32, max_nearby=10)
    d = sfa.build_cvrp(p, n_replicas=1, max_nearby=10)
    d.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=8, accepted_count_limit=32))
    d.calculate_score()
    d.construct_list_clarke_wright(0, p["customers"], 1)
    d.phase_start()
    o.phase_start()
    d.solve_steps(40)
    o.steps(40)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_clarke_wright_cvrp_1000(oracle, mode):
    """C3 size (499,500 savings entries): lists == the oracle's, every customer routed exactly once, committed == fresh score."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(1000, 100, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    sc, flags = d.construct_list_clarke_wright(0, p["customers"], mode)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    committed, _ = o.construct_list_clarke_wright(p["customers"], mode)
    assert committed and flags.all()
    lists = d.working_lists(0, 1)
    assert lists == o.get_lists(0)
    assert sorted(c for rt in lists for c in rt) == list(range(1, 1001))
    assert (sc[0] == o.score()[:2]).all() and (d.fresh_score()[1] == sc[1]).all()
    if mode == 1:
        assert sc[0][0] == 0  # capacity-feasible routes


def test_clarke_wright_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(20, 4, 55, seed=1)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=1)
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_clarke_wright(0, [1, 2, 2], 0)  # duplicate declared element
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_clarke_wright(0, [1, 2, 9999], 0)  # id out of range
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_clarke_wright(0, [1, 2, 3], 2)  # feasible_mode
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_clarke_wright(1, [1, 2, 3], 0)  # not the list class
    sc, flags = d.construct_list_clarke_wright(0, [0], 0)  # the depot's value only: nothing to route
    assert not flags.any() and d.working_lists(0, 0) == [[] for _ in range(4)]
    sc, flags = d.construct_list_clarke_wright(0, [5], 1)  # one element: a singleton route, no savings entry
    assert flags.all() and sorted(c for rt in d.working_lists(0, 0) for c in rt) == [5]


# ---- route-local 2-opt polishing (sf_construct_list_k_opt ≙ ListKOptPhase, list_k_opt/kernel.rs:57-220) ------------------------
KOPT_CASES = [
    ("plain", "roundrobin", 1, 1000), ("plain", "roundrobin", 0, 1000), ("plain", "savings", 1, 1000), ("ragged", "roundrobin", 0, 1000),
    ("ties", "roundrobin", 0, 1000), ("asym", "roundrobin", 0, 7), ("asym", "roundrobin", 1, 3), ("infeasible", "roundrobin", 1, 1000),
    ("infeasible", "roundrobin", 0, 1000), ("ragged", "one_route", 0, 1000), ("plain", "one_route", 1, 1000),
]


@pytest.mark.parametrize("problem,start,mode,max_sweeps", KOPT_CASES)
def test_list_k_opt_matches_oracle(oracle, problem, start, mode, max_sweeps):
    """start: the round-robin fill, the savings routes (capacity mode), or every customer in ONE route (> 64 visits: several
    rounds of j per row; mode 1: over capacity, so no reversal is taken).  The asymmetric cases run under a small sweep bound:
    parity holds sweep for sweep whether or not the 2-opt delta converges there."""
    import solverforge_amd as sfa

    p = _problem(problem)
    if start == "one_route":
        p["routes"] = [[int(c) for c in p["customers"]]] + [[] for _ in p["routes"][1:]]
    elif start == "savings":
        p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=3)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    if start == "savings":
        d.construct_list_clarke_wright(0, p["customers"], 1)
        o.construct_list_clarke_wright([int(c) for c in p["customers"]], 1)
    before = o.get_lists(0)
    d.construct_list_k_opt(0, 3, mode, max_sweeps)  # k != 2: scored no-op (also allocates the counters)
    assert d.working_lists(0, 0) == before
    g0, o0 = d.stats(0), o.stats()
    sc = d.construct_list_k_opt(0, 2, mode, max_sweeps)
    st = o.construct_list_k_opt(2, mode, max_sweeps)
    for r in range(3):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] - g0[k] == ost[k] - o0[k], k
    assert int(st[0]) > 0
    if problem == "plain" and start == "roundrobin":
        assert o.get_lists(0) != before and int(st[1]) > 0
    if (problem == "infeasible" or start == "one_route") and mode == 1:
        assert o.get_lists(0) == before and int(st[1]) == 0  # over-capacity routes: every reversal is infeasible
    sc2 = d.construct_list_k_opt(0, 3, mode, max_sweeps)  # k != 2: scored no-op
    assert (sc2 == sc).all() and d.working_lists(0, 0) == o.get_lists(0)


def test_default_construction_pipeline_cvrp_1000(oracle):
    """The reference's default construction of the CVRP domain (defaults/stages.rs:225-266): Clarke-Wright, then ListKOpt; here with
    the capacity test on both.  C3 size, lists == the oracle's."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(1000, 100, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    d.construct_list_clarke_wright(0, p["customers"], 1)
    sc = d.construct_list_k_opt(0, 2, 1)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_clarke_wright(p["customers"], 1)
    st = o.construct_list_k_opt(2, 1)
    assert d.working_lists(0, 1) == o.get_lists(0)
    assert (sc[0] == o.score()[:2]).all() and sc[0].tolist() == [0, -93239] and int(st[1]) == 44
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_k_opt(0, 2, 1, 0)  # max_sweeps >= 1
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_k_opt(0, 2, 2)


def test_clarke_wright_cvrp_5000_properties():
    """C5 size (12,497,500 savings entries, 148 KB of route state per replica): every customer routed once, capacity-feasible
    routes, committed == fresh score, the score the oracle's run of this instance produced (profiles/r02f_cw_bench.jsonl:
    matches_oracle true; the oracle takes 26 s, so it is not re-run here); ListKOpt keeps every route's visit set and does not
    worsen the distance."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(5000, 500, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    sc, flags = d.construct_list_clarke_wright(0, p["customers"], 1)
    assert flags.all() and sc[0].tolist() == [0, -389476] and (sc[1] == sc[0]).all()
    lists = d.working_lists(0, 1)
    assert sorted(c for rt in lists for c in rt) == list(range(1, 5001))
    assert all(sum(int(p["demands"][c]) for c in rt) <= 55 for rt in lists)
    assert (d.fresh_score()[0] == sc[0]).all()
    sc2 = d.construct_list_k_opt(0, 2, 1)
    after = d.working_lists(0, 1)
    assert [sorted(rt) for rt in after] == [sorted(rt) for rt in lists]
    assert sc2[0][0] == 0 and sc2[0][1] >= sc[0][1] and (d.fresh_score()[1] == sc2[1]).all()
