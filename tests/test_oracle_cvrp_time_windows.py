"""Oracle groundwork for CVRP time windows (CPU only).  The oracle's list k-opt phase takes the stock crate's complete `route_hooks::feasible`
(capacity + time windows; feasible_mode 2) and is pinned here to the reference's end-to-end test for it,
crates/solverforge/tests/list_cvrp_k_opt_time_window.rs:9-41 (fixture: list_cvrp_k_opt_time_window/domain/plan.rs:46-88): the 2-opt reversal that
would shorten [1, 3, 2, 4] to [1, 2, 3, 4] reaches customer 3 after its window closes, so the route must stay as it is.  The oracle's predicate and
the oracle-side `oracle.cvrp_data.route_feasible` (pinned to the crate's own tests in tests/test_cvrp_data.py) are cross-checked on seeded
routes.  The device's `sf_construct_list_k_opt` implements modes 0 and 1; mode 2 is the next step there."""
import numpy as np

from oracle import cvrp_data as cv


def _reference_plan():  # plan.rs:46-88
    d = np.full((5, 5), 100, dtype=np.int64)
    np.fill_diagonal(d, 0)
    for (a, b, v) in [(0, 1, 1), (1, 3, 50), (3, 2, 1), (2, 4, 50), (4, 0, 1), (1, 2, 1), (2, 3, 1), (3, 4, 1)]:
        d[a, b] = v
    t = np.zeros((5, 5), dtype=np.int64)
    t[1, 2] = 10
    t[2, 3] = 10
    return dict(capacity=100, depot=0, demands=[0, 1, 1, 1, 1], matrix=d, lo=[0] * 5, hi=[100, 100, 100, 5, 100], service=[0] * 5, travel=t)


def _model(oracle, p, routes, with_windows=True):
    m = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], list(range(1, len(p["demands"]))), routes)
    if with_windows:
        m.set_time_windows(p["lo"], p["hi"], p["service"], p["travel"], 0)
    return m


def test_stock_cvrp_list_k_opt_rejects_time_window_breaking_reversal(oracle):
    p = _reference_plan()
    m = _model(oracle, p, [[1, 3, 2, 4]])
    st = m.construct_list_k_opt(k=2, feasible_mode=2)
    assert m.get_lists(0) == [[1, 3, 2, 4]] and int(st[1]) == 0  # the reference's assertion: the initial route survives
    # the same phase without the time windows in its hook takes the reversal (capacity alone admits it)
    m1 = _model(oracle, p, [[1, 3, 2, 4]])
    m1.construct_list_k_opt(k=2, feasible_mode=1)
    assert m1.get_lists(0) == [[1, 2, 3, 4]]
    assert m.route_feasible([1, 3, 2, 4]) and not m.route_feasible([1, 2, 3, 4]) and m.route_feasible([])


def test_oracle_predicate_equals_the_host_mirror_on_seeded_routes(oracle):
    rng = np.random.default_rng(11)
    n = 9
    for case in range(40):
        d = rng.integers(1, 30, (n, n)).astype(np.int64)
        np.fill_diagonal(d, 0)
        t = rng.integers(0, 12, (n, n)).astype(np.int64)
        if case % 5 == 0:
            t[rng.integers(0, n), rng.integers(0, n)] = cv.UNREACHABLE
        lo = rng.integers(0, 30, n).astype(np.int64)
        hi = lo + rng.integers(0, 60, n)
        service = rng.integers(0, 6, n).astype(np.int64)
        demands = rng.integers(0, 7, n).astype(np.int32)
        demands[0] = 0
        cap = int(rng.integers(8, 25))
        dep = int(rng.integers(0, 15))
        m = oracle.Model.cvrp(cap, 0, demands, d, list(range(1, n)), [[]])
        m.set_time_windows(lo, hi, service, t, dep)
        data = cv.ProblemData(cap, 0, demands.tolist(), d.tolist(), list(zip(lo.tolist(), hi.tolist())), service.tolist(), t.tolist(), dep)
        plan = cv.VrpSolution([[]], [data])
        for _ in range(60):
            route = rng.permutation(np.arange(1, n))[: rng.integers(0, n)].tolist()
            assert m.route_feasible(route) == cv.route_feasible(plan, 0, route), (case, route)
