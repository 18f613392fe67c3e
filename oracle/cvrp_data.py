"""ORACLE (test infrastructure; the product package never imports this file).  A cited CPU restatement of the reference's stock CVRP domain crate (crates/solverforge-cvrp/src): `ProblemData` with time windows, service
durations and travel times as DATA (problem_data.rs:1-48), the route-local helpers and hook bundles a `#[planning_list_variable(domain =
"cvrp")]` model gets (helpers.rs:1-218) and the two matrix distance meters (meters.rs:1-51).

What the device consumes of it: the distance matrix, demands, capacity and depot (`build_cvrp` / `sf_list_model_*`); its trial pricing and the
Clarke-Wright construction treat capacity as a score term, exactly like the reference's `savings_feasible`, which admits capacity AND
time-window violations ("remain scoreable during construction", helpers.rs:78-81).  `route_feasible` -- capacity + time windows + overflow-safe
accumulation -- is the gate of the route-improving list k-opt phase (manager/phase_factory/list_k_opt/kernel.rs:167); it runs here, on the host,
over a replica's downloaded routes (`ScoreDirector.working_lists`).  Arithmetic: i64 with checked additions like the reference; Python ints are
unbounded, so "overflow" is an explicit range test.

Names, argument order and results follow the crate; `tests/test_cvrp_data.py` restates its tests.rs one to one.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

UNREACHABLE = (1 << 63) - 1  # problem_data.rs:6 (i64::MAX; the sentinel of solverforge-maps road matrices)
MAX_SAFE_LEG_COST = UNREACHABLE // 4  # problem_data.rs:8
_I64_MAX = (1 << 63) - 1


@dataclass
class ProblemData:  # problem_data.rs:15-24
    capacity: int
    depot: int
    demands: List[int]
    distance_matrix: List[List[int]]
    time_windows: List[Tuple[int, int]]
    service_durations: List[int]
    travel_times: List[List[int]]
    vehicle_departure_time: int = 0

    @staticmethod
    def _finite(matrix, a: int, b: int) -> Optional[int]:  # problem_data.rs:43-46
        if not (0 <= a < len(matrix)) or not (0 <= b < len(matrix[a])):
            return None
        v = matrix[a][b]
        return v if (v >= 0 and v != UNREACHABLE) else None

    def distance_cost(self, a: int, b: int) -> int:  # problem_data.rs:28-31
        v = self._finite(self.distance_matrix, a, b)
        return MAX_SAFE_LEG_COST if v is None else v

    def finite_distance(self, a: int, b: int) -> Optional[int]:
        return self._finite(self.distance_matrix, a, b)

    def travel_time(self, a: int, b: int) -> Optional[int]:
        return self._finite(self.travel_times, a, b)


@dataclass
class VrpSolution:
    """The crate's `VrpSolution` trait as a value: per vehicle its route and its `ProblemData` (None = the trait's null pointer).  Vehicles
    that share one ProblemData OBJECT share a savings metric class (helpers.rs:42-58)."""

    routes: List[List[int]]
    data: List[Optional[ProblemData]] = field(default_factory=list)

    def vehicle_count(self) -> int:
        return len(self.routes)


def _problem_data_for_entity(plan: VrpSolution, e: int) -> Optional[ProblemData]:  # helpers.rs:4-19 (a null pointer for a non-empty fleet is an error)
    if e >= plan.vehicle_count():
        return None
    d = plan.data[e]
    if d is None:
        raise AssertionError("VrpSolution::vehicle_data_ptr(%d) returned null for a non-empty fleet" % e)
    return d


def _optional_problem_data_for_entity(plan: VrpSolution, e: int) -> Optional[ProblemData]:  # helpers.rs:21-37 (feasibility gates: null = not admissible)
    return None if e >= plan.vehicle_count() else plan.data[e]


def depot_for_entity(plan, e):  # helpers.rs:39-41
    d = _problem_data_for_entity(plan, e)
    return 0 if d is None else d.depot


def savings_metric_class(plan, e):  # helpers.rs:47-58
    if e >= plan.vehicle_count():
        return e
    d = plan.data[e]
    if d is None:
        raise AssertionError("VrpSolution::vehicle_data_ptr(%d) returned null for a non-empty fleet" % e)
    return id(d) + (1 << 40)  # (the pointer value in the reference: never collides with a small entity index)


def route_distance(plan, e, a, b):  # helpers.rs:91-93
    d = _problem_data_for_entity(plan, e)
    return 0 if d is None else d.distance_cost(a, b)


savings_depot_for_entity = depot_for_entity  # helpers.rs:61-63
savings_distance = route_distance  # helpers.rs:66-73


def get_route(plan, e):  # helpers.rs:104-106
    return list(plan.routes[e])


def replace_route(plan, e, route):  # helpers.rs:98-100
    plan.routes[e] = list(route)


def _route_is_structurally_valid(route: Sequence[int], d: ProblemData) -> bool:  # helpers.rs:140-166
    if not route:
        return True
    max_visit = max(route)
    if min(route) < 0:
        return False  # (usize in the reference)
    max_node = max(max_visit, d.depot)
    if (max_visit >= len(d.demands) or max_visit >= len(d.time_windows) or max_visit >= len(d.service_durations)
            or max_node >= len(d.distance_matrix) or max_node >= len(d.travel_times)):
        return False
    return all(len(d.distance_matrix[node]) > max_node and len(d.travel_times[node]) > max_node for node in list(route) + [d.depot])


def _route_is_capacity_feasible(route, d: ProblemData) -> bool:  # helpers.rs:168-178
    total = 0
    for v in route:
        total += d.demands[v]
        if total > _I64_MAX:
            return False
    return total <= d.capacity


def _route_is_time_feasible(route, d: ProblemData) -> bool:  # helpers.rs:180-218
    t = d.vehicle_departure_time
    prev = d.depot
    for v in route:
        tt = d.travel_time(prev, v)
        if tt is None:
            return False
        t += tt
        if t > _I64_MAX:
            return False
        lo, hi = d.time_windows[v]
        if t < lo:
            t = lo
        s = d.service_durations[v]
        if s < 0:
            return False
        t += s
        if t > _I64_MAX or t > hi:
            return False
        prev = v
    back = d.travel_time(prev, d.depot)
    return back is not None and t + back <= _I64_MAX


def route_feasible(plan, e, route) -> bool:  # helpers.rs:109-119
    if not route:
        return True
    d = _optional_problem_data_for_entity(plan, e)
    if d is None:
        return False
    return _route_is_structurally_valid(route, d) and _route_is_capacity_feasible(route, d) and _route_is_time_feasible(route, d)


def savings_feasible(plan, e, route) -> bool:  # helpers.rs:78-89: only routes that cannot be evaluated safely are rejected
    if not route:
        return True
    d = _optional_problem_data_for_entity(plan, e)
    return d is not None and _route_is_structurally_valid(route, d)


class route_hooks:  # helpers.rs:122-128
    depot = staticmethod(depot_for_entity)
    get = staticmethod(get_route)
    set = staticmethod(replace_route)
    distance = staticmethod(route_distance)
    feasible = staticmethod(route_feasible)


class savings_hooks:  # helpers.rs:134-138
    depot = staticmethod(savings_depot_for_entity)
    distance = staticmethod(savings_distance)
    feasible = staticmethod(savings_feasible)


def matrix_distance(plan, src_entity, src_pos, dst_entity, dst_pos) -> float:  # meters.rs:10-28 (MatrixDistanceMeter)
    a, b = plan.routes[src_entity], plan.routes[dst_entity]
    if src_pos >= len(a) or dst_pos >= len(b):
        return float("inf")
    d = _problem_data_for_entity(plan, src_entity)
    v = None if d is None else d.finite_distance(a[src_pos], b[dst_pos])
    return float("inf") if v is None else float(v)


def matrix_intra_distance(plan, src_entity, src_pos, _dst_entity, dst_pos) -> float:  # meters.rs:34-51 (MatrixIntraDistanceMeter)
    a = plan.routes[src_entity]
    if src_pos >= len(a) or dst_pos >= len(a):
        return float("inf")
    d = _problem_data_for_entity(plan, src_entity)
    v = None if d is None else d.finite_distance(a[src_pos], a[dst_pos])
    return float("inf") if v is None else float(v)


def to_device_problem(d: ProblemData):
    """The part of a ProblemData the device model is built from (`solverforge_amd.build_cvrp`): matrix, demands, capacity, depot.  Legs the
    reference treats as non-traversable get the finite construction cost (`distance_cost`), as the stock distance hooks do."""
    import numpy as np

    n = len(d.distance_matrix)
    m = np.array([[d.distance_cost(a, b) for b in range(n)] for a in range(n)], dtype=np.int64)
    return {"matrix": m, "demands": np.asarray(d.demands, dtype=np.int64), "capacity": int(d.capacity), "depot": int(d.depot)}
