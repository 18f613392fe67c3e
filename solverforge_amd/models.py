"""Model definitions: the device form of the reference examples' `define_constraints()`.

Each builder wires schema + constraint archetypes + default-policy selectors exactly as the
corresponding reference model does with ConstraintFactory streams.
"""
from .director import ConstraintKind, GpuScoreDirector, PairOp, SelectorKind

FACT_MATRIX, FACT_DEMAND, FACT_CUSTOMERS, FACT_ADJ, FACT_GROUP, FACT_COLUMN, FACT_AUX = 0, 1, 2, 3, 4, 5, 6


def build_cvrp(problem, n_replicas=1, device_id=0, max_nearby=20, leaves=("nearby_change", "nearby_swap"),
               sublist_sizes=(1, 3), kopt=(1, 20), ruin=(2, 5, 10), permute=(2, 5)):
    # leaves may also name the plain streams "list_change" / "list_swap" (generic N-leaf engine)
    """CVRP: HardSoftScore; all_customers_assigned (not-exists, 1 hard each —
    crates/solverforge/tests/list_clarke_wright_publication/domain/publication_plan.rs:51-65),
    vehicle_capacity (uni on routes, max(0, load-cap) hard), total_distance (uni on routes,
    depot->...->depot, soft); leaves = default list policy nearby change + nearby swap with
    MatrixDistanceMeter, max_nearby 20 (default_local_search/policy/list.rs:19,97-141)."""
    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    n_vehicles = len(problem["routes"])
    dim = problem["matrix"].shape[0]
    d.add_entity_class(0, n_vehicles)
    d.add_list_variable(0, problem["routes"], element_capacity=len(problem["customers"]), element_id_bound=dim)
    d.add_fact_matrix(FACT_MATRIX, problem["matrix"])
    d.add_fact_column_i32(FACT_DEMAND, problem["demands"])
    d.add_fact_column_u32(FACT_CUSTOMERS, problem["customers"])
    d.add_constraint(ConstraintKind.NOT_EXISTS_FLATTENED, 0, fact=FACT_CUSTOMERS, level=0, weight=1)
    d.add_constraint(ConstraintKind.ROUTE_CAPACITY, 0, fact=FACT_DEMAND, param=int(problem["capacity"]), level=0, weight=1)
    d.add_constraint(ConstraintKind.ROUTE_DISTANCE, 0, fact=FACT_MATRIX, param=int(problem["depot"]), level=1, weight=1)
    # selectors are declared in the order `leaves` names them: a configured root union (configure_union) schedules and weights
    # its children in declaration order; the default policy's union ignores it (policy/list.rs:24-33 order)
    for leaf in leaves:
        if leaf == "nearby_change":
            d.add_selector(SelectorKind.NEARBY_LIST_CHANGE, 0, max_nearby=max_nearby, fact_meter=FACT_MATRIX)
        elif leaf == "nearby_swap":
            d.add_selector(SelectorKind.NEARBY_LIST_SWAP, 0, max_nearby=max_nearby, fact_meter=FACT_MATRIX)
        elif leaf == "list_change":
            d.add_selector(SelectorKind.LIST_CHANGE, 0)
        elif leaf == "list_swap":
            d.add_selector(SelectorKind.LIST_SWAP, 0)
        elif leaf == "list_reverse":
            d.add_selector(SelectorKind.LIST_REVERSE, 0)
        elif leaf == "sublist_change":
            d.add_sublist_selector(SelectorKind.SUBLIST_CHANGE, 0, min_size=sublist_sizes[0], max_size=sublist_sizes[1])
        elif leaf == "sublist_swap":
            d.add_sublist_selector(SelectorKind.SUBLIST_SWAP, 0, min_size=sublist_sizes[0], max_size=sublist_sizes[1])
        elif leaf == "permute":  # permute = (min_window_size, max_window_size)
            d.add_permute_selector(0, min_window_size=permute[0], max_window_size=permute[1])
        elif leaf == "kopt":  # kopt = (min_segment_len, max_nearby); max_nearby 0 = full enumeration
            d.add_kopt_selector(0, min_segment_len=kopt[0], max_nearby=kopt[1])
        elif leaf == "ruin":  # ruin = (min_ruin_count, max_ruin_count, moves_per_step)
            d.add_ruin_selector(0, min_ruin_count=ruin[0], max_ruin_count=ruin[1], moves_per_step=ruin[2], variable_name="visits")
        else:
            raise ValueError(f"unknown leaf {leaf!r}")
    return d


def build_graph_coloring(problem, n_replicas=1, device_id=0, leaves=("change", "swap"), pair_ir=False):
    """Graph colouring: HardSoftScore; `Unassigned color` (uni, 1 hard each) and `Adjacent color
    conflict` (predicate cross-join left.id < right.id && neighbours && equal colour, 1 hard each —
    examples/scalar-graph-coloring/src/domain/graph_coloring.rs:21-44); leaves = scalar change +
    scalar swap on `color_idx` (value range 0..n_colors, allows_unassigned)."""
    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    n = problem["n"]
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, problem["n_colors"], True, problem["colors"])
    d.add_fact_csr(FACT_ADJ, problem["adj_off"], problem["adj"])
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    if pair_ir:  # the same join with its predicate written out as a program (sf_constraint_add_pair_join): neighbours && equal colour
        d.add_pair_join(0, [(PairOp.CSR_CONTAINS, 0, FACT_ADJ), (PairOp.VALUE_EQ, 1)], level=0, weight=1)
    else:
        d.add_constraint(ConstraintKind.CROSS_ADJACENT_EQUAL, 0, fact=FACT_ADJ, level=0, weight=1)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


def build_nqueens(rows, n_replicas=1, device_id=0, leaves=("change", "swap"), pair_ir=False):
    """N-queens: queen i sits in column i, planning variable row_idx in 0..n (allows_unassigned);
    `Unassigned` + row/diagonal conflicts (examples/nqueens/src/domain/board.rs:21-47)."""
    import numpy as np

    n = len(rows)
    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, n, True, rows)
    d.add_fact_column_i32(FACT_COLUMN, np.arange(n, dtype=np.int32))
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    if pair_ir:  # distinct columns && (same row || same diagonal), board.rs:30-44
        d.add_pair_join(0, [(PairOp.COL_NE, 0, FACT_COLUMN), (PairOp.VALUE_EQ, 1), (PairOp.VALUE_ABSDIFF_EQ_COL, 1, FACT_COLUMN)], level=0, weight=1)
    else:
        d.add_constraint(ConstraintKind.CROSS_QUEENS, 0, fact=FACT_COLUMN, level=0, weight=1)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


def build_jobshop(problem, n_replicas=1, device_id=0, bendable=True,
                  leaves=("list_change", "list_swap", "change", "swap"), makespan=False, ruin=(2, 5, 10), precedence_policy=False, pair_ir=False, owner_match_level=None):
    """Mixed job shop (examples/mixed-job-shop/src/domain/job_shop_plan.rs:28-69): class 0 =
    operations with the scalar `machine_idx` (0..n_machines, allows_unassigned), class 1 = machines
    with the list variable `sequence` of operation ids.  BendableScore<2,1> (BASELINE.json):
    level 0 unassigned machine, level 1 unscheduled operation (not-exists over the flattened
    sequences), level 2 same job on the same machine (predicate cross-join on operations)."""
    import numpy as np

    levels, hard = (3, 2) if bendable else (2, 1)
    lv = (0, 1, 2) if bendable else (0, 0, 1)
    d = GpuScoreDirector(score_levels=levels, hard_levels=hard, n_replicas=n_replicas, device_id=device_id)
    n_ops, n_m = problem["n_ops"], problem["n_machines"]
    d.add_entity_class(0, n_ops)
    d.add_scalar_variable(0, 0, n_m, True, problem["machine_idx"])
    d.add_entity_class(1, n_m)
    d.add_list_variable(1, problem["sequences"], element_capacity=n_ops, element_id_bound=n_ops)
    d.add_fact_column_i32(FACT_GROUP, np.asarray(problem["job"], dtype=np.int32))
    d.add_fact_column_u32(FACT_CUSTOMERS, np.arange(n_ops, dtype=np.uint32))
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=lv[0], weight=1)
    d.add_constraint(ConstraintKind.NOT_EXISTS_FLATTENED, 1, fact=FACT_CUSTOMERS, level=lv[1], weight=1)
    if pair_ir:  # same job && same machine, job_shop_plan.rs:50-62
        d.add_pair_join(0, [(PairOp.COL_EQ, 0, FACT_GROUP), (PairOp.VALUE_EQ, 1)], level=lv[2], weight=1)
    else:
        d.add_constraint(ConstraintKind.CROSS_GROUP_EQUAL, 0, fact=FACT_GROUP, level=lv[2], weight=1)
    if owner_match_level is not None:  # the join of the two planning classes: an operation on a machine that does not schedule it
        d.add_constraint(ConstraintKind.CROSS_OWNER_MATCH, 0, param=1, level=owner_match_level, weight=1)
    if makespan:  # the makespan objective (constraint/list_precedence.rs): job order = fixed successors, problem["durations"]
        job = np.asarray(problem["job"])
        succ = [[op + 1] if op + 1 < n_ops and job[op + 1] == job[op] else [] for op in range(n_ops)]
        d.add_list_precedence(1, problem["durations"], succ, None, hard_level=lv[1], makespan_level=lv[2])
    if "precedence" in leaves:  # the precedence pair leads the list policy (policy/list.rs:24-33)
        d.add_precedence_selector(1)
    if "permute" in leaves:
        d.add_permute_selector(1)
    if "ruin" in leaves:
        d.add_ruin_selector(1, min_ruin_count=ruin[0], max_ruin_count=ruin[1], moves_per_step=ruin[2], variable_name="sequence")
    if precedence_policy:
        d.set_precedence_policy(1)
    if "list_change" in leaves:
        d.add_selector(SelectorKind.LIST_CHANGE, 1)
    if "list_swap" in leaves:
        d.add_selector(SelectorKind.LIST_SWAP, 1)
    # the rest of the default policy of a list without a distance meter (policy/list.rs:24-33,161-171):
    # sublist change / swap (sizes 1..=3), reverse, unbounded 3-opt
    if "sublist_change" in leaves:
        d.add_sublist_selector(SelectorKind.SUBLIST_CHANGE, 1)
    if "sublist_swap" in leaves:
        d.add_sublist_selector(SelectorKind.SUBLIST_SWAP, 1)
    if "list_reverse" in leaves:
        d.add_selector(SelectorKind.LIST_REVERSE, 1)
    if "kopt" in leaves:
        d.add_kopt_selector(1, max_nearby=0)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


def build_precedence_shop(problem, n_replicas=1, device_id=0, leaves=("list_change", "list_swap"), levels=2, hard_levels=1,
                          hard_level=0, makespan_level=1, with_owner=True, ruin=(2, 5, 10), precedence_policy=False, element_capacity=None):
    """Job shop with the makespan objective: class 0 = machines with the list variable `sequence` of operation ids; the one
    constraint is the ListPrecedenceMakespanConstraint (crates/solverforge-scoring/src/constraint/list_precedence.rs:13-707) over
    the job order (fixed successors), the machine sequences and the expected machine of every operation."""
    d = GpuScoreDirector(score_levels=levels, hard_levels=hard_levels, n_replicas=n_replicas, device_id=device_id)
    n = len(problem["durations"])
    d.add_entity_class(0, len(problem["sequences"]))
    d.add_list_variable(0, problem["sequences"], element_capacity=n if element_capacity is None else element_capacity, element_id_bound=n)
    d.add_list_precedence(0, problem["durations"], problem["successors"], problem["expected_owner"] if with_owner else None,
                          hard_level=hard_level, makespan_level=makespan_level)
    if "precedence" in leaves:  # the precedence pair leads the list policy (policy/list.rs:24-33)
        d.add_precedence_selector(0)
    if "permute" in leaves:
        d.add_permute_selector(0)
    if "list_change" in leaves:
        d.add_selector(SelectorKind.LIST_CHANGE, 0)
    if "list_swap" in leaves:
        d.add_selector(SelectorKind.LIST_SWAP, 0)
    if "sublist_change" in leaves:
        d.add_sublist_selector(SelectorKind.SUBLIST_CHANGE, 0)
    if "sublist_swap" in leaves:
        d.add_sublist_selector(SelectorKind.SUBLIST_SWAP, 0)
    if "list_reverse" in leaves:
        d.add_selector(SelectorKind.LIST_REVERSE, 0)
    if "kopt" in leaves:
        d.add_kopt_selector(0, max_nearby=0)
    if "ruin" in leaves:  # ruin = (min_count, max_count, moves_per_step)
        d.add_ruin_selector(0, min_ruin_count=ruin[0], max_ruin_count=ruin[1], moves_per_step=ruin[2])
    if precedence_policy:
        d.set_precedence_policy(0)
    return d


def build_shift_schedule(nurse_idx, day, n_nurses, n_replicas=1, device_id=0, limit=2, w_streak=1, count_weight=0, target=-1,
                         leaves=("change", "swap"), required=None, presence=None):
    """examples/minimal-shift-scheduling/src/domain/schedule.rs:21-83: shifts choose a nurse.  Hard: unassigned shift; two shifts of
    one nurse on one day (predicate cross-join on the day column).  Soft: long work streaks -- group_by(nurse,
    consecutive_runs(day)).penalize(sum over runs of max(0, point_count - limit)) (stream/collector/runs.rs); count_weight > 0:
    target >= 0: balanced workload -- group_by(nurse, count()).complement(nurses, 0).penalize(|count - target|) (:61-74, the
    complemented grouped node); target < 0: count_weight * (shifts per nurse)^2 (plain grouped count)."""
    import numpy as np

    n = len(nurse_idx)
    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, n_nurses, True, nurse_idx)
    d.add_fact_column_i32(FACT_GROUP, np.asarray(day, dtype=np.int32))
    if required is None:
        d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    else:  # filter(required && unassigned): the filter / per-shift weight as a column (0 = not required)
        d.add_fact_column_i32(FACT_AUX, np.asarray(required, dtype=np.int32))
        d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, fact=FACT_AUX, level=0, weight=1)
    d.add_constraint(ConstraintKind.CROSS_GROUP_EQUAL, 0, fact=FACT_GROUP, level=0, weight=1)
    if presence is None:
        d.add_constraint(ConstraintKind.RUNS_VALUE, 0, fact=FACT_GROUP, param=limit, level=1, weight=w_streak)
    else:  # group_by(nurse, indexed_presence(day)).penalize(w_streak * min(count_in(lo..hi), cap)) instead of the streaks
        lo, hi, cap = presence[:3]
        mode = presence[3] if len(presence) > 3 else 0  # 1: excess of complement_runs(lo..hi) over cap
        d.add_constraint(ConstraintKind.PRESENCE_VALUE, 0, fact=FACT_GROUP, param=lo | (hi << 16) | (cap << 32) | (mode << 48), level=1, weight=w_streak)
    if count_weight > 0:
        d.add_fact_column_i32(FACT_COLUMN, np.ones(n, dtype=np.int32))
        if target >= 0:
            d.add_constraint(ConstraintKind.COMPLEMENTED_VALUE_SUM, 0, fact=FACT_COLUMN, param=target, level=1, weight=count_weight)
        else:
            d.add_constraint(ConstraintKind.GROUPED_VALUE_SUM, 0, fact=FACT_COLUMN, param=-1, level=1, weight=count_weight)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


def build_balance(bins, sizes, n_bins, n_replicas=1, device_id=0, w_pair=1, cap=-1, leaves=("change", "swap"), arity=2, balance_base=1000):
    """Bin balance: the keyed self-join (pairs of entities sharing a bin — IncrementalBiConstraint,
    constraint/nary_incremental/bi.rs:12-313) and the grouped sum (group_by(bin, sum(size)) with
    weight(sum) — constraint/grouped/{state,scorer}.rs) on one scalar variable.  HardSoftScore:
    hard = unassigned entities, soft = w_pair per same-bin pair + sum^2 (cap == -1) or excess over cap (cap >= 0);
    cap == -2 replaces the per-bin load by FAIRNESS: group_by(load_balance(bin, size)).penalize(unfairness)
    (stream/collector/load_balance.rs), sizes >= 1; cap == -3 replaces it by the BalanceConstraint (constraint/balance.rs):
    `balance_base` (1000) soft per unit of the standard deviation of the per-bin entity COUNTS."""
    import numpy as np

    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    n = len(bins)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, n_bins, True, bins)
    d.add_fact_column_i32(FACT_COLUMN, np.asarray(sizes, dtype=np.int32))
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    # arity 3 / 4 / 5: tri / quad / penta self-join (constraint/nary_incremental/higher_arity/shared.rs)
    d.add_constraint(ConstraintKind.SELFJOIN_VALUE_EQUAL, 0, level=1, weight=w_pair, param=0 if arity == 2 else arity)
    if cap == -2:
        d.add_constraint(ConstraintKind.LOAD_BALANCE_VALUE, 0, fact=FACT_COLUMN, level=1, weight=1)
    elif cap == -3:
        d.add_constraint(ConstraintKind.BALANCE_VALUE, 0, level=1, weight=balance_base)
    else:
        d.add_constraint(ConstraintKind.GROUPED_VALUE_SUM, 0, fact=FACT_COLUMN, param=cap, level=1, weight=1)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d


def build_assignment(values, cost, n_values, n_replicas=1, device_id=0, cost_weight=1, row_w=None, ex_mode=1, ex_level=1, ex_weight=1,
                     leaves=("change", "swap")):
    """Assignment: n entities choose one of n_values fact rows.  HardSoftScore: `Unassigned` (uni, 1 hard each); the keyed
    cross-join of the planning class with the fact class -- for_each(A).join(B, equal(A.value, B.id)).filter(cost != 0)
    .penalize(cost_weight * cost[a][b]) (cross_bi_incremental::Bi keyed by the planning value, the shape of
    constraint/tests/cross_bi_incr.rs:60-83) as the matrix fact `cost`; and for_each(B).if_exists / if_not_exists(A, equal(B.id,
    A.value)).penalize(ex_weight * row_w[b]) (IncrementalExistsConstraint, constraint/exists.rs:42-437) on `ex_level`
    (ex_level < 0: no exists node)."""
    import numpy as np

    n = len(values)
    d = GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=n_replicas, device_id=device_id)
    d.add_entity_class(0, n)
    d.add_scalar_variable(0, 0, n_values, True, values)
    d.add_fact_matrix(FACT_MATRIX, np.ascontiguousarray(cost, dtype=np.int64).reshape(n, n_values))
    d.add_constraint(ConstraintKind.UNI_UNASSIGNED, 0, level=0, weight=1)
    d.add_constraint(ConstraintKind.VALUE_COST, 0, fact=FACT_MATRIX, level=1, weight=cost_weight)
    if ex_level >= 0:
        fact = -1
        if row_w is not None:
            d.add_fact_column_i32(FACT_COLUMN, np.ascontiguousarray(row_w, dtype=np.int32))
            fact = FACT_COLUMN
        d.add_constraint(ConstraintKind.EXISTS_VALUE, 0, fact=fact, param=ex_mode, level=ex_level, weight=ex_weight)
    if "change" in leaves:
        d.add_selector(SelectorKind.SCALAR_CHANGE, 0)
    if "swap" in leaves:
        d.add_selector(SelectorKind.SCALAR_SWAP, 0)
    return d
