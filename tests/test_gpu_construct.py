"""GPU parity tests (through the C ABI): list cheapest-insertion construction on the device (sf_construct_list_cheapest ≙
ListCheapestInsertionPhase, cheapest/kernel.rs:57-150) vs the oracle: constructed lists, committed score, counters; from empty
lists, from a partial state, ties, unreachable legs, asymmetric matrix; then local search from the constructed state.  Regret
insertion (sf_construct_list_regret ≙ ListRegretInsertionPhase) and round robin the same way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(kind):
    from solverforge_amd import datasets

    if kind == "plain":
        p = datasets.make_cvrp(60, 6, 55, seed=3)
    elif kind == "tight":
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
    else:
        raise ValueError(kind)
    return p


@pytest.mark.parametrize("problem,keep", [("plain", 0), ("tight", 0), ("ties", 0), ("asym", 0), ("plain", 2), ("tight", 5)])
def test_cheapest_insertion_matches_oracle(oracle, problem, keep):
    """keep = how many of the round-robin start routes stay filled (a partial state: only the missing customers are placed)."""
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    d = sfa.build_cvrp(p, n_replicas=3)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    placed = {c for rt in p["routes"] for c in rt}
    missing = [int(c) for c in p["customers"] if int(c) not in placed]
    sc = d.construct_list_cheapest(0, p["customers"])  # the elements already in a list are skipped
    o.construct_list_cheapest(missing)
    for r in range(3):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k
    assert gst["moves_applied"] == len(missing) and gst["moves_generated"] == gst["score_calculations"] > 0
    # local search continues from the constructed state
    leaves = oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP
    o.configure(leaves=leaves, random_seed=1, la_size=8, limit=32, max_nearby=10)
    d2 = sfa.build_cvrp(p, n_replicas=1, max_nearby=10)
    d2.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=8, accepted_count_limit=32))
    d2.calculate_score()
    d2.construct_list_cheapest(0, p["customers"])
    d2.phase_start()
    o.phase_start()
    d2.solve_steps(40)
    o.steps(40)
    assert d2.working_lists(0, 0) == o.get_lists(0)
    assert (d2.calculate_score()[0] == o.score()[:2]).all()


def test_cheapest_insertion_cvrp_1000_properties(oracle):
    """C3 size: every customer placed exactly once, committed == fresh score, and the first 150 placements equal the oracle's."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(1000, 100, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    sc = d.construct_list_cheapest(0, p["customers"])
    lists = d.working_lists(0, 0)
    assert sorted(c for rt in lists for c in rt) == list(range(1, 1001))
    assert (sc == d.fresh_score()).all() and sc[0][0] <= 0
    d3 = sfa.build_cvrp(p, n_replicas=1)
    d3.calculate_score()
    d3.construct_list_cheapest(0, p["customers"][:150])
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_cheapest(p["customers"][:150])
    assert d3.working_lists(0, 0) == o.get_lists(0)


def test_cheapest_insertion_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p)
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_cheapest(0, [999])  # element id out of range
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_cheapest(3, [1])  # not the list class
    d.construct_list_cheapest(0, p["customers"])  # nothing missing: a no-op
    assert d.working_lists(0, 0) == p["routes"]


# ---- regret insertion (sf_construct_list_regret ≙ ListRegretInsertionPhase, regret/kernel/{execute,evaluation,mod}.rs) -------------
@pytest.mark.parametrize("problem,keep", [("plain", 0), ("tight", 0), ("ties", 0), ("asym", 0), ("plain", 2), ("tight", 5)])
def test_regret_insertion_matches_oracle(oracle, problem, keep):
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    d = sfa.build_cvrp(p, n_replicas=3)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    placed = {c for rt in p["routes"] for c in rt}
    missing = [int(c) for c in p["customers"] if int(c) not in placed]
    sc = d.construct_list_regret(0, p["customers"])  # the elements already in a list are not candidates
    o.construct_list_regret(missing)
    for r in range(3):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k
    assert gst["moves_applied"] == len(missing)
    if problem != "ties":  # regret insertion is not cheapest insertion in another order
        d1 = sfa.build_cvrp(p, n_replicas=1)
        d1.calculate_score()
        d1.construct_list_cheapest(0, p["customers"])
        assert d1.working_lists(0, 0) != d.working_lists(0, 0)
    # local search continues from the constructed state
    o.configure(leaves=oracle.LEAF_NEARBY_LIST_CHANGE | oracle.LEAF_NEARBY_LIST_SWAP, random_seed=1, la_size=8, limit=32, max_nearby=10)
    d2 = sfa.build_cvrp(p, n_replicas=1, max_nearby=10)
    d2.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=8, accepted_count_limit=32))
    d2.calculate_score()
    d2.construct_list_regret(0, p["customers"])
    d2.phase_start()
    o.phase_start()
    d2.solve_steps(30)
    o.steps(30)
    assert d2.working_lists(0, 0) == o.get_lists(0)


@pytest.mark.parametrize("n_missing", [1, 2, 3, 4, 5, 9])
def test_regret_insertion_few_elements_and_single_list(oracle, n_missing):
    """Groups of four elements share a pass: every remainder; one list only (the first element of an empty list is Forced)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(14, 1, 400, seed=n_missing)
    p["routes"] = [[]]
    cust = [int(c) for c in p["customers"][:n_missing]]
    d = sfa.build_cvrp(p, n_replicas=2)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.calculate_score()
    sc = d.construct_list_regret(0, cust)
    o.construct_list_regret(cust)
    assert d.working_lists(0, 1) == o.get_lists(0) and (sc[1] == o.score()[:2]).all()
    assert d.stats(1)["score_calculations"] == o.stats()["score_calculations"]


@pytest.mark.parametrize("problem,keep", [("ties", 0), ("plain", 0), ("tight", 3)])
def test_regret_insertion_with_order_keys(oracle, problem, keep):
    """element_order_key: the unassigned elements are ranked by (key, source index); on the all-ties matrix the keys decide
    every round."""
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    n = len(p["customers"])
    ks = np.random.default_rng(5).integers(0, 4, n).astype(np.int64)
    d = sfa.build_cvrp(p, n_replicas=2)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.calculate_score()
    placed = {c for rt in p["routes"] for c in rt}
    miss = [i for i, c in enumerate(p["customers"]) if int(c) not in placed]
    sc = d.construct_list_regret(0, p["customers"], ks)
    o.construct_list_regret([int(p["customers"][i]) for i in miss], ks[miss])
    assert d.working_lists(0, 1) == o.get_lists(0) and (sc[1] == o.score()[:2]).all()
    if problem == "ties":
        d0 = sfa.build_cvrp(p, n_replicas=1)
        d0.calculate_score()
        d0.construct_list_regret(0, p["customers"])
        assert d0.working_lists(0, 0) != d.working_lists(0, 0)
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_regret(0, p["customers"], ks[:-1])


@pytest.mark.parametrize("problem,keep,keys", [("plain", 0, False), ("tight", 3, True), ("ties", 0, False), ("asym", 0, True)])
def test_regret_insertion_with_owner_hook(oracle, problem, keep, keys):
    """The owner hook (list_placement.rs:54-69): unrestricted, fixed (only that list's slots; Forced while it is empty), no valid
    owner (never placed); counters count the candidate slots only."""
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    nv, n = len(p["routes"]), len(p["customers"])
    rng = np.random.default_rng(11)
    ks = rng.integers(0, 5, n).astype(np.int64) if keys else None
    ow = np.full(n, -1, dtype=np.int64)
    pick = rng.choice(n, n // 3, replace=False)
    ow[pick] = rng.integers(0, nv + 2, len(pick))  # nv, nv + 1 = no valid owner
    d = sfa.build_cvrp(p, n_replicas=2)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.calculate_score()
    placed = {c for rt in p["routes"] for c in rt}
    miss = [i for i, c in enumerate(p["customers"]) if int(c) not in placed]
    sc = d.construct_list_regret(0, p["customers"], ks, ow)
    o.construct_list_regret([int(p["customers"][i]) for i in miss], None if ks is None else ks[miss], ow[miss])
    for r in range(2):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
    lists = d.working_lists(0, 0)
    for i in miss:
        c = int(p["customers"][i])
        where = [e for e, rt in enumerate(lists) if c in rt]
        assert where == ([] if ow[i] >= nv else ([int(ow[i])] if ow[i] >= 0 else where)) and (ow[i] >= nv or len(where) == 1)
    gst, ost = d.stats(1), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


def test_regret_insertion_owner_budget_is_refused():
    """All-fixed-owner inputs above the reference's trial budget take its bounded fallbacks (regret/kernel/fallback.rs): not built."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(60, 1, 400, seed=2)
    p["routes"] = [[]]
    d = sfa.build_cvrp(p)
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError, match="SF_ERR_UNSUPPORTED"):
        d.construct_list_regret(0, p["customers"], None, np.zeros(60, dtype=np.int32))  # 60 * 61 * 62 / 6 = 37,820 > 16,384
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_regret(0, p["customers"], None, np.full(60, -2, dtype=np.int32))
    d.construct_list_regret(0, p["customers"][:40], None, np.zeros(40, dtype=np.int32))  # 11,480 trials: the main loop
    assert sorted(d.working_lists(0, 0)[0]) == sorted(int(c) for c in p["customers"][:40])


def test_regret_insertion_cvrp_300_properties(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(300, 30, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    sc = d.construct_list_regret(0, p["customers"])
    lists = d.working_lists(0, 0)
    assert sorted(c for rt in lists for c in rt) == list(range(1, 301))
    assert (sc == d.fresh_score()).all() and sc[0][0] <= 0
    d3 = sfa.build_cvrp(p, n_replicas=1)
    d3.calculate_score()
    d3.construct_list_regret(0, p["customers"][:50])
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_regret(p["customers"][:50])
    assert d3.working_lists(0, 0) == o.get_lists(0)


def test_regret_insertion_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p)
    d.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_regret(0, [999])  # element id out of range
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_regret(0, [3, 4, 3])  # duplicate source key (regret/tests.rs:178-192)
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_regret(3, [1])  # not the list class
    d.construct_list_regret(0, p["customers"])  # nothing missing: a no-op
    assert d.working_lists(0, 0) == p["routes"]
    s = datasets.make_precedence_shop(4, 3, seed=1)
    ds = sfa.build_precedence_shop(s)
    ds.calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        ds.construct_list_regret(0, [0, 1])  # precedence hooks: not built


# ---- round-robin list construction (sf_construct_list_round_robin ≙ ListConstructionPhase, round_robin/kernel.rs:71-175) ----------
@pytest.mark.parametrize("problem,keep,keys,owners", [
    ("plain", 0, False, False), ("plain", 2, False, False), ("tight", 0, True, False), ("plain", 0, False, True),
    ("tight", 3, True, True), ("asym", 0, True, True), ("ties", 5, False, False),
])
def test_round_robin_matches_oracle(oracle, problem, keep, keys, owners):
    """keep = start routes that stay filled (their customers are not candidates); keys = construction order keys with ties
    (source index breaks them); owners = owner hook values: unrestricted, fixed (cursor does not advance), out of range
    (skipped)."""
    import solverforge_amd as sfa

    p = _problem(problem)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    nv = len(p["routes"])
    d = sfa.build_cvrp(p, n_replicas=3)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    d.calculate_score()
    rng = np.random.default_rng(7)
    n = len(p["customers"])
    ks = rng.integers(0, 6, n).astype(np.int64) if keys else None
    ow = None
    if owners:
        ow = np.full(n, -1, dtype=np.int64)
        pick = rng.choice(n, n // 3, replace=False)
        ow[pick] = rng.integers(0, nv + 2, len(pick))  # nv, nv + 1 = no valid owner
    placed = {c for rt in p["routes"] for c in rt}
    miss = [i for i, c in enumerate(p["customers"]) if int(c) not in placed]
    sc = d.construct_list_round_robin(0, p["customers"], ks, ow)
    o.construct_list_round_robin([int(p["customers"][i]) for i in miss], None if ks is None else ks[miss], None if ow is None else ow[miss])
    for r in range(3):
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (sc[r] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k
    if not owners:
        assert gst["moves_applied"] == len(miss)


def test_round_robin_golden_and_c3_start(oracle):
    """The reference's known answer (list_clarke_wright/tests/compiled_parity.rs:468-476: 1..4 over two routes -> [[1, 3], [2, 4]])
    and: the round-robin fill bench.py times M1 on (datasets.make_cvrp) IS this construction from empty routes."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(4, 2, 55, seed=0)
    p["routes"] = [[], []]
    d = sfa.build_cvrp(p, n_replicas=1)
    d.calculate_score()
    d.construct_list_round_robin(0, [1, 2, 3, 4])
    assert d.working_lists(0, 0) == [[1, 3], [2, 4]]
    p = datasets.make_cvrp(1000, 100, 55, seed=0)
    start = [list(map(int, rt)) for rt in p["routes"]]
    p["routes"] = [[] for _ in p["routes"]]
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    sc = d.construct_list_round_robin(0, p["customers"])
    assert d.working_lists(0, 1) == start
    assert sc[1].tolist() == [-97, -548558]
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_round_robin(0, [1, 1])
    with pytest.raises(sfa.SolverForgeError):
        d.construct_list_round_robin(0, [1, 2], None, [-2, 0])
