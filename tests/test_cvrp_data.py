"""oracle/cvrp_data.py (an oracle-side restatement) against the stock CVRP crate's own tests (crates/solverforge-cvrp/src/tests.rs:111-410), one to one: same
fixture (tests.rs:30-52), same names, same assertions.  CPU only."""
import copy
import math

import pytest

from oracle import cvrp_data as cv
from oracle.cvrp_data import UNREACHABLE


def base_problem_data():  # tests.rs:30-52
    m = [[0, 5, 7, 9], [5, 0, 4, 6], [7, 4, 0, 3], [9, 6, 3, 0]]
    return cv.ProblemData(capacity=10, depot=0, demands=[0, 2, 3, 4], distance_matrix=copy.deepcopy(m), time_windows=[(0, 100), (0, 10), (7, 14), (0, 12)],
                          service_durations=[0, 2, 2, 3], travel_times=copy.deepcopy(m), vehicle_departure_time=0)


def test_solution(routes, data=None):  # TestSolution::new / with_data: one ProblemData per vehicle
    return cv.VrpSolution([list(r) for r in routes], data if data is not None else [base_problem_data() for _ in routes])


test_solution.__test__ = False


def test_helpers_use_problem_data_for_route_owner():
    one = base_problem_data()
    one.depot = 3
    one.distance_matrix[1][3] = 42
    s = test_solution([[1, 2], [3]], [base_problem_data(), one])
    assert cv.route_distance(s, 1, 1, 3) == 42 and cv.depot_for_entity(s, 0) == 0 and cv.depot_for_entity(s, 1) == 3
    assert cv.route_feasible(s, 0, [1, 2])


def test_helpers_handle_empty_fleets():
    s = test_solution([])
    assert cv.route_distance(s, 0, 1, 2) == 0 and cv.depot_for_entity(s, 0) == 0 and cv.savings_metric_class(s, 3) == 3
    assert not cv.route_feasible(s, 0, [1, 2]) and cv.route_feasible(s, 0, [])


def test_savings_metric_class_groups_shared_and_separates_distinct_problem_data():
    shared = base_problem_data()
    s = cv.VrpSolution([[1], [2], [3]], [shared, shared, shared])
    assert cv.savings_metric_class(s, 0) == cv.savings_metric_class(s, 1) == cv.savings_metric_class(s, 2)
    t = test_solution([[1], [2]])
    assert cv.savings_metric_class(t, 0) != cv.savings_metric_class(t, 1)


def test_clarke_wright_adapters_share_exact_cvrp_data_when_requested():
    s = test_solution([[1, 2], [3]])
    assert cv.savings_depot_for_entity(s, 0) == cv.depot_for_entity(s, 0)
    assert cv.savings_distance(s, 0, 1, 2) == cv.route_distance(s, 0, 1, 2)
    assert cv.savings_feasible(s, 0, [1, 2]) == cv.route_feasible(s, 0, [1, 2])


@pytest.mark.parametrize("a,b,route", [(0, 1, [1]), (1, 2, [1, 2]), (1, 0, [1])])
def test_route_feasibility_rejects_unreachable_legs_without_panic(a, b, route):  # depot leg / inter-visit leg / return leg (tests.rs:183-217)
    d = base_problem_data()
    d.distance_matrix[a][b] = UNREACHABLE
    d.travel_times[a][b] = UNREACHABLE
    assert cv.route_feasible(test_solution([route], [d]), 0, route) is False


def test_route_feasibility_rejects_time_and_service_overflow_without_wrapping():
    d = base_problem_data()
    d.vehicle_departure_time = 20
    d.travel_times[0][1] = (1 << 63) - 1 - 10
    assert not cv.route_feasible(test_solution([[1]], [d]), 0, [1])
    d = base_problem_data()
    d.time_windows[1] = (0, (1 << 63) - 1)
    d.service_durations[1] = (1 << 63) - 1
    assert not cv.route_feasible(test_solution([[1]], [d]), 0, [1])


def test_stock_savings_feasibility_stays_structural_for_unreachable_routes():
    d = base_problem_data()
    d.distance_matrix[0][1] = UNREACHABLE
    d.travel_times[0][1] = UNREACHABLE
    s = test_solution([[1]], [d])
    assert cv.savings_feasible(s, 0, [1]) and cv.savings_hooks.feasible(s, 0, [1])


def test_stock_distances_convert_unreachable_or_malformed_legs_to_finite_costs():
    d = base_problem_data()
    d.distance_matrix[0][1] = UNREACHABLE
    s = test_solution([[1]], [d])
    unreachable_cost, malformed_cost = cv.route_distance(s, 0, 0, 1), cv.route_distance(s, 0, 99, 1)
    assert 0 < unreachable_cost < UNREACHABLE and cv.savings_distance(s, 0, 0, 1) == unreachable_cost == malformed_cost


def test_hook_bundles_expose_route_and_savings_semantics():
    s = test_solution([[1, 2], [3]])
    assert cv.route_hooks.get(s, 0) == cv.get_route(s, 0)
    cv.route_hooks.set(s, 1, [2, 1])
    assert cv.get_route(s, 1) == [2, 1]
    assert cv.route_hooks.depot(s, 0) == cv.depot_for_entity(s, 0) and cv.route_hooks.distance(s, 0, 1, 2) == cv.route_distance(s, 0, 1, 2)
    assert cv.route_hooks.feasible(s, 0, [1, 2]) == cv.route_feasible(s, 0, [1, 2])
    assert cv.savings_hooks.depot(s, 0) == cv.savings_depot_for_entity(s, 0) and cv.savings_hooks.distance(s, 0, 1, 2) == cv.savings_distance(s, 0, 1, 2)
    assert cv.savings_hooks.feasible(s, 0, [1, 2]) == cv.savings_feasible(s, 0, [1, 2])


def test_helpers_reject_missing_problem_data_for_non_empty_fleets():
    with pytest.raises(AssertionError, match=r"vehicle_data_ptr\(0\) returned null"):
        cv.route_distance(cv.VrpSolution([[1, 2]], [None]), 0, 1, 2)


def test_route_helpers_replace_and_clone_routes():
    s = test_solution([[1, 2], [3]])
    cv.replace_route(s, 0, [2, 3])
    assert s.routes[0] == [2, 3] and cv.get_route(s, 0) == [2, 3]
    cv.replace_route(s, 1, [1])
    assert s.routes[1] == [1]


def test_route_feasibility_rejects_time_violations_while_savings_admits_them():
    s = test_solution([[1, 2], [3]])
    assert cv.route_feasible(s, 0, [1, 2])  # waits for customer 2 and still finishes in time
    assert not cv.route_feasible(s, 0, [2, 3]) and not cv.route_hooks.feasible(s, 0, [2, 3])
    assert cv.savings_feasible(s, 0, [2, 3]) and cv.savings_hooks.feasible(s, 0, [2, 3])


def test_route_feasibility_rejects_capacity_violations_while_savings_admits_them():
    one = base_problem_data()
    one.capacity = 4
    s = test_solution([[1, 2], [3]], [base_problem_data(), one])
    assert cv.route_feasible(s, 0, [1, 2]) and not cv.route_feasible(s, 1, [1, 2]) and not cv.route_hooks.feasible(s, 1, [1, 2])
    assert cv.savings_feasible(s, 1, [1, 2]) and cv.savings_hooks.feasible(s, 1, [1, 2])


def test_route_feasibility_rejects_structurally_invalid_routes():
    s = test_solution([[1, 2], [3]])
    null = cv.VrpSolution([[1, 2]], [None])
    for f in (cv.route_feasible, cv.savings_feasible, cv.route_hooks.feasible, cv.savings_hooks.feasible):
        assert not f(s, 0, [4])
    assert not cv.route_feasible(s, 2, [1]) and not cv.savings_feasible(s, 2, [1])
    assert not cv.route_feasible(null, 0, [1]) and not cv.savings_feasible(null, 0, [1])
    assert cv.route_feasible(null, 0, []) and cv.savings_feasible(null, 0, [])


def test_distance_meters_cover_invalid_positions_and_unreachable_legs():
    s = test_solution([[1, 2], [3]])
    assert cv.matrix_distance(s, 0, 0, 1, 0) == 6.0 and cv.matrix_intra_distance(s, 0, 0, 0, 1) == 4.0
    assert math.isinf(cv.matrix_distance(s, 0, 4, 1, 0)) and math.isinf(cv.matrix_intra_distance(s, 0, 0, 0, 4))
    d = base_problem_data()
    d.distance_matrix[1][2] = UNREACHABLE
    t = test_solution([[1, 2], [2]], [copy.deepcopy(d), d])
    assert math.isinf(cv.matrix_distance(t, 0, 0, 1, 0)) and math.isinf(cv.matrix_intra_distance(t, 0, 0, 0, 1))


def test_device_problem_keeps_what_the_device_consumes():
    d = base_problem_data()
    d.distance_matrix[0][1] = UNREACHABLE
    p = cv.to_device_problem(d)
    assert p["matrix"].shape == (4, 4) and p["matrix"][0, 1] == cv.MAX_SAFE_LEG_COST and p["matrix"][1, 2] == 4
    assert p["demands"].tolist() == [0, 2, 3, 4] and p["capacity"] == 10 and p["depot"] == 0
