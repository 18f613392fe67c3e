/*
 * solverforge_amd.h — C ABI of the MI355X-native SolverForge hot path.
 *
 * The reference (SolverForge/solverforge, 100 % Rust) exposes no FFI; its seams are
 * monomorphised traits.  This header is the boundary a Rust `extern "C"` shim binds to
 * implement those traits on top of the HIP path (INTEGRATION.md shows the shim):
 *
 *   Director<S>            crates/solverforge-scoring/src/director/traits.rs:27-95
 *   ConstraintSet<S,Sc>    crates/solverforge-scoring/src/api/constraint_set/incremental.rs:152-212
 *   MoveSelector/MoveCursor crates/solverforge-solver/src/heuristic/selector/move_selector/iter.rs:239-279
 *   MoveCursorSource       crates/solverforge-solver/src/phase/localsearch/cursor_source.rs:23-48
 *   ValueSelector          crates/solverforge-solver/src/heuristic/selector/value_selector.rs:21-41
 *   ScalarCandidateProvider crates/solverforge-solver/src/planning/scalar/candidate.rs:190
 *   SolutionDescriptor     crates/solverforge-core/src/domain/descriptor/solution.rs:16-33
 *
 * Conventions: every entry point is extern "C", takes plain pointers and sizes, returns an
 * int32 status (0 = SF_OK, <0 = error; text via sf_last_error).  Input buffers are borrowed
 * for the duration of the call; the context owns all device memory.  One context = one HIP
 * stream; calls on one context are not re-entrant, distinct contexts are independent.
 * A context holds `n_replicas` independent searches of the same problem (a per-GPU batch of
 * the seed portfolio); replica r is a full Director + search state.
 * There is no CPU fallback: every call fails with SF_ERR_NO_DEVICE when no gfx950 device exists.
 */
#ifndef SOLVERFORGE_AMD_H
#define SOLVERFORGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sf_ctx sf_ctx;

enum {
    SF_OK = 0,
    SF_ERR_INVALID = -1,     /* bad argument / call order (the reference would panic!) */
    SF_ERR_NO_DEVICE = -2,   /* no HIP device: the product path never falls back to the CPU */
    SF_ERR_HIP = -3,         /* HIP runtime error, see sf_last_error */
    SF_ERR_UNSUPPORTED = -4, /* valid reference feature outside this build's scope */
    SF_ERR_CAPACITY = -5     /* output buffer too small */
};

#define SF_MAX_LEVELS 4
#define SF_NONE (-1) /* Option<usize>::None for scalar planning variables */

/* One candidate move.  A ScalarEdit{descriptor_index, entity_index, variable_name, to_value}
 * (crates/solverforge-solver/src/planning/scalar/candidate.rs:6-82) is kind = SF_MOVE_CHANGE
 * with a = entity_index, value = to_value; descriptor/variable come from the selector's slot. */
typedef struct sf_move_t {
    int32_t kind;  /* sf_move_kind */
    int32_t a;     /* Change: entity; Swap: left entity; List*: source / first entity */
    int32_t a_pos; /* List*: source / first position */
    int32_t b;     /* Swap: right entity; List*: destination / second entity */
    int32_t b_pos; /* List*: destination / second position (ListChange: pre-removal coordinates,
                      heuristic/move/list_kernel/change.rs:28-34) */
    int32_t value; /* Change: to_value (SF_NONE = unassign) */
} sf_move_t;

typedef enum sf_move_kind {
    SF_MOVE_CHANGE = 0,      /* heuristic/move/change.rs:118-221 */
    SF_MOVE_SWAP = 1,        /* heuristic/move/swap.rs:150-215 */
    SF_MOVE_LIST_CHANGE = 2, /* heuristic/move/list_kernel/change.rs:16-153 */
    SF_MOVE_LIST_SWAP = 3,   /* heuristic/move/list_kernel/swap.rs:17-110 */
    SF_MOVE_LIST_REVERSE = 4,/* heuristic/move/list_kernel/reverse.rs:15-57: reverse list `a` over [a_pos, b_pos) (b = a) */
    SF_MOVE_SUBLIST_CHANGE = 5,/* heuristic/move/list_kernel/sublist_change.rs:18-130: segment [a_pos, value) of list `a`
                                  -> list `b` at b_pos (post-removal coordinates when a == b) */
    SF_MOVE_SUBLIST_SWAP = 6,  /* heuristic/move/list_kernel/sublist_swap.rs:17-160: segment [a_pos, a_pos + (value & 0xFFFF)) of
                                  list `a` <-> segment [b_pos, b_pos + (value >> 16)) of list `b` */
    SF_MOVE_KOPT = 7,          /* heuristic/move/list_kernel/k_opt.rs:13-96 with k = 3: list `a` cut at a_pos < b < b_pos (`b`
                                  carries the MIDDLE CUT, not an entity) and reconnected by
                                  THREE_OPT_RECONNECTIONS[value] (move/k_opt_reconnection.rs:203-211), value in 0..6 */
    SF_MOVE_LIST_MULTI_SWAP = 10, /* heuristic/move/list_kernel/multi_swap.rs:13-128 (ListMultiSwapMove, emitted by the critical-path leaf): `a` swaps in
                                    pairwise different lists applied as one move; a_pos / b / b_pos = (list | first position << 16) of swap 0 / 1 / 2,
                                    value = (second - first) of each swap, one byte per swap.  Requires a score improvement
                                    (phase/localsearch/evaluation.rs:95-113).  sf_step_evaluate scores it (independent lists: the deltas add), sf_apply
                                    commits it */
    SF_MOVE_LIST_PERMUTE = 9,  /* heuristic/move/list_kernel/permute.rs:22-72 (ListPermuteMove): the window [a_pos, b_pos) of list `a`
                                  (b = a, 2..8 positions) reordered by the value-th permutation of its positions in lexicographic
                                  order (nth_permutation, selector/list_kernel/permute.rs:260-272; value >= 1, 0 would be the identity) */
    SF_MOVE_LIST_RUIN = 8      /* heuristic/move/list_kernel/ruin.rs:131-281 (one source list): list `a` loses the a_pos (1..6) elements
                                  at ascending positions packed 16 bits each into b (positions 0, 1), b_pos (2, 3), value (4, 5);
                                  every removed element is greedily re-inserted at its best (list, position).  Ruins of the critical-path
                                  leaf (at most five elements) use the top half of `value`: bit 31 = the move carries the precedence hooks
                                  (insertions that close a cycle are skipped, ruin.rs:186-220), bit 30 = two source lists
                                  (ListRuinMove::new_multi_source, list_ruin.rs:80-100): position 0 lies in list `a`, position 1 in the
                                  list held by bits 16..29 */
} sf_move_kind;

/* Declarative constraint archetypes (the reference's closure-typed ConstraintFactory streams
 * cannot run on a GPU).  Each is the device form of one incremental node. */
typedef enum sf_constraint_kind {
    /* for_each(E).unassigned().penalize(w) — IncrementalUniConstraint, constraint/incremental.rs:19-193.  On a scalar class
     * `fact_a` >= 0 names an i32 column w[e]: for_each(E).filter(f(e) && unassigned).penalize(weight * w(e)) with the filter and
     * the per-entity weight as data, 0 = filtered out (e.g. `shift.required && shift.nurse_idx.is_none()`,
     * examples/minimal-shift-scheduling/src/domain/schedule.rs:23-27); -1 = every entity, weight 1 */
    SF_C_UNI_UNASSIGNED = 1,
    /* predicate cross-join on one class: left.id<right.id && adjacent(left,right) &&
     * assigned && equal value — cross_bi_incremental::Bi with constant key
     * (examples/scalar-graph-coloring/src/domain/graph_coloring.rs:28-41); `fact_a` = CSR adjacency */
    SF_C_CROSS_ADJACENT_EQUAL = 2,
    /* predicate cross-join: left.id<right.id && group[left]==group[right] && assigned && equal value
     * (examples/mixed-job-shop/src/domain/job_shop_plan.rs:50-62); `fact_a` = i32 group column */
    SF_C_CROSS_GROUP_EQUAL = 3,
    /* N-queens row/diagonal conflicts (examples/nqueens/src/domain/board.rs:30-44); `fact_a` = column */
    SF_C_CROSS_QUEENS = 4,
    /* for_each(A).if_not_exists(for_each(owners).flattened(list), equal_bi(A.id, item)) —
     * IncrementalExistsConstraint, constraint/exists.rs:42-437; `fact_a` = u32 key column of A */
    SF_C_NOT_EXISTS_FLATTENED = 5,
    /* for_each(routes).penalize(max(0, sum(demand[visit]) - capacity)) — uni on list owners;
     * `fact_a` = i32 demand column, `param` = capacity */
    SF_C_ROUTE_CAPACITY = 6,
    /* for_each(routes).penalize(depot->...->depot matrix sum) — uni on list owners
     * (ProblemData::distance_cost, crates/solverforge-cvrp/src/problem_data.rs:28-31);
     * `fact_a` = i64 matrix, `param` = depot node */
    SF_C_ROUTE_DISTANCE = 7,
    /* for_each(E).join(equal(value)) on one scalar class, both assigned: `weight` per pair —
     * keyed self-join IncrementalBiConstraint, constraint/nary_incremental/bi.rs:12-313.  `param` = arity: 0 / 2 =
     * pairs, 3 / 4 / 5 = IncrementalTri / Quad / PentaConstraint (constraint/nary_incremental/higher_arity/shared.rs):
     * `weight` per index-sorted tuple of assigned entities sharing a value (C(members, arity) per value) */
    SF_C_SELFJOIN_VALUE_EQUAL = 8,
    /* for_each(E).filter(assigned).group_by(value, sum(fact_a)).penalize(weight * w(sum)) — grouped node +
     * sum collector, constraint/grouped/{state,scorer}.rs, stream/collector/sum.rs;
     * `fact_a` = i32 column summed per group, `param` < 0: w = sum^2, `param` >= 0: w = max(0, sum - param) */
    SF_C_GROUPED_VALUE_SUM = 9,
    /* for_each(E).filter(assigned).group_by(load_balance(value, fact_a)).penalize(weight * unfairness) — the grouped
     * node with the load_balance collector, stream/collector/load_balance.rs:100-226: unfairness =
     * round(sqrt(sum(load^2) - (sum load)^2 / keys)) in f64, the one floating-point step of the scoring path
     * (bit-identical: IEEE division, correctly rounded sqrt, round half away from zero).  `fact_a` = i32 metric
     * column, every metric >= 1 (SF_ERR_UNSUPPORTED otherwise: the reference skips zero metrics) */
    SF_C_LOAD_BALANCE_VALUE = 10,
    /* keyed cross-join of the planning class A with a FACT class B: for_each(A).join(for_each(B), equal(A.value, B.id))
     * .filter(f).penalize(w) -- cross_bi_incremental::Bi keyed by the planning value (constraint/cross_bi_incremental/state.rs:32-461,
     * incremental.rs:93-137; the shape of constraint/tests/cross_bi_incr.rs:60-83 "unavailable employee").  The closures become
     * data: `fact_a` = i64 matrix cost[n_rows][n_values], cost[a][b] = weight of the pair (a, b), 0 = the filter rejects it;
     * the level gets -weight * cost[a][value(a)] for every assigned a.  Also the uni form for_each(A).filter(f).penalize(w(a, value)). */
    SF_C_VALUE_COST = 11,
    /* for_each(B).if_exists / if_not_exists(for_each(A).filter(assigned), equal(B.id, A.value)).penalize(weight * w[B]) with B the
     * value-keyed fact rows -- IncrementalExistsConstraint in Exists (param 1) or NotExists (param 0) mode over a planning
     * class (constraint/exists.rs:42-437, key_existence_delta :218-231; constraint/tests/exists.rs:34-190): a row is scored while
     * some / no entity holds its value.  `fact_a` = i32 per-row weight column [n_values], or -1 for weight 1 */
    SF_C_EXISTS_VALUE = 12,
    /* BalanceConstraint (crates/solverforge-scoring/src/constraint/balance.rs:83-372): group the assigned entities by their value,
     * count each group, score -round(weight * population standard deviation of the counts) on `level` -- a GLOBAL statistic;
     * f64 in the reference's operation order, Score::multiply rounding (score/macros.rs:61-63).  `weight` = the base score of one
     * unit of standard deviation */
    SF_C_BALANCE_VALUE = 13,
    /* ListPrecedenceMakespanConstraint (crates/solverforge-scoring/src/constraint/list_precedence.rs:13-707); declared through
     * sf_constraint_add_list_precedence, listed here for its place in the sf_evaluate_each order */
    SF_C_LIST_PRECEDENCE_MAKESPAN = 14,
    /* for_each(E).filter(assigned).group_by(value, consecutive_runs(fact_a)).penalize(weight * sum over the runs of
     * max(0, run.point_count - param)) -- the grouped node with the consecutive-runs collector (stream/collector/runs.rs:11-229;
     * the "long work streaks" constraint of examples/minimal-shift-scheduling/src/domain/schedule.rs:45-59).  `fact_a` = i32 point
     * column (one point per entity, 0 <= point < 4096; duplicates count once per run), `param` = the run length that is still free.
     * The device keeps a [n_values][points] count table in LDS (<= 48 KiB); a trial reads the run lengths either side of the moved
     * point.  Scalar engine only; not chained in compound candidates */
    SF_C_RUNS_VALUE = 15,
    /* for_each(E).filter(assigned).group_by(value, sum(fact_a)).complement(B, |b| b.id, |_| 0).penalize(weight * |sum - param|)
     * with B the value-keyed fact rows -- the complemented grouped node (constraint/complemented/{state,helpers,incremental}.rs,
     * stream/grouped_stream/base.rs:142-200): EVERY value row is scored, a row nobody holds with the default result 0 (the
     * "balanced workload" constraint of examples/minimal-shift-scheduling/src/domain/schedule.rs:61-74 with a ones column).
     * `fact_a` = i32 column summed per group, `param` = the target.  Shares the grouped slot (one grouped constraint per class) */
    SF_C_COMPLEMENTED_VALUE_SUM = 16,
    /* for_each(E).filter(assigned).group_by(value, indexed_presence(fact_a)).penalize(weight * min(presence.count_in(lo..hi), cap))
     * -- the grouped node with the indexed-presence collector (stream/collector/indexed_presence.rs:1-147: contains / count /
     * count_in / any_in over the set of present points).  `fact_a` = i32 point column as for SF_C_RUNS_VALUE; `param` = lo |
     * hi << 16 | cap << 32 with 0 <= lo <= hi <= 4096: cap 0 = presence.count_in(lo..hi) (lo = 0, hi = 4096: presence.count(), the
     * distinct points of the group), cap 1 = presence.any_in(lo..hi) as 0 / 1.  param | 1 << 48: the row scores
     * sum over presence.complement_runs(lo..hi) of max(0, run.point_count - cap) instead -- the runs of ABSENT points inside the
     * horizon ("consecutive off bounds" of crates/solverforge-macros/tests/ui/pass/solverforge_constraints_indexed_presence.rs).  Shares the per-(value, point) count table and the
     * slot of SF_C_RUNS_VALUE (one of the two per class); scalar engine only; not chained in compound candidates */
    SF_C_PRESENCE_VALUE = 17,
    /* A join of the TWO planning classes of a mixed model, both sides moving (constraint/cross_bi_incremental/incremental.rs:93-137: an
     * insert / retract of either class runs its own side, state.rs:372-461):
     *   for_each(E).join(for_each(Owner), equal(e.value, owner.index)).filter(|e, o| !o.list.contains(e.id)).penalize(weight)
     * -- every entity of the scalar class `descriptor_index` whose assigned value names a list owner that does not hold it (a job-shop
     * operation assigned to a machine that does not schedule it).  `param` = descriptor of the list class; the scalar variable's values are
     * the owner indices (n_values == owners), the list elements are the scalar class's entity ids.  A scalar move changes the A side's key, a
     * list move the B side's filter: both are priced on the device (generic engine), the committed match table is an entity -> holding
     * list map in HBM.  Fused / traced search, sf_step_generate, sf_initialize / sf_evaluate_all / sf_evaluate_each, and (round 6) the
     * host-driven entry points sf_step_evaluate / sf_step_evaluate_compound / sf_apply / sf_apply_compound, which rebuild the map of the one
     * replica per call and price a record from its coordinates.  Still SF_ERR_UNSUPPORTED for a model that declares it: SF_MOVE_LIST_RUIN
     * records and unions with a ruin or critical-path leaf (the recreate does not price the join), the construction phases; sf_step_decide*
     * takes scalar-only models anyway */
    SF_C_CROSS_OWNER_MATCH = 18
} sf_constraint_kind;

/* ---- pair predicates as data (round 5) -------------------------------------------------------------------------------------
 * The reference composes the filter of a predicate join from closures (stream/join_target.rs:28-110: a join whose two sides extract
 * the same planning collection and whose joiner is a predicate takes a CONSTANT key -- every pair (left.id < right.id) is tested;
 * cross_bi_incremental/state.rs:260-296 add_match).  Here the predicate is a small program: a conjunction of clauses, each clause a
 * disjunction of terms over (left, right) = the pair in entity-index order, both assigned.  The three model-shaped kinds above are
 * presets of it:  SF_C_CROSS_ADJACENT_EQUAL = {CSR_CONTAINS(adj)} and {VALUE_EQ};  SF_C_CROSS_GROUP_EQUAL = {COL_EQ(group)} and {VALUE_EQ};
 * SF_C_CROSS_QUEENS = {COL_NE(column)} and {VALUE_EQ or VALUE_ABSDIFF_EQ_COL(column)}  (examples/scalar-graph-coloring/src/domain/
 * graph_coloring.rs:28-41, examples/mixed-job-shop/src/domain/job_shop_plan.rs:50-62, examples/nqueens/src/domain/board.rs:30-44).
 * On the device a clause that is one CSR_CONTAINS or one COL_EQ term becomes the partner index the trial walks (deg(e) tests instead of n);
 * the remaining clauses are interpreted per candidate pair from the kernel's argument block (wave-uniform control, per-lane data). */
typedef enum sf_pair_op {
    SF_PAIR_VALUE_EQ = 1,             /* left.value == right.value */
    SF_PAIR_VALUE_NE = 2,             /* left.value != right.value */
    SF_PAIR_VALUE_ABSDIFF_EQ_COL = 3, /* |left.value - right.value| == |col[left] - col[right]|     fact = i32 column */
    SF_PAIR_COL_EQ = 4,               /* col[left] == col[right]                                     fact = i32 column */
    SF_PAIR_COL_NE = 5,               /* col[left] != col[right] */
    SF_PAIR_COL_LT = 6,               /* col[left] < col[right]   (left = the lower entity index) */
    SF_PAIR_COL_ABSDIFF_EQ = 7,       /* |col[left] - col[right]| == param */
    SF_PAIR_COL_ABSDIFF_LE = 8,       /* |col[left] - col[right]| <= param */
    SF_PAIR_CSR_CONTAINS = 9,         /* right in csr[left] or left in csr[right]                    fact = CSR, one row per entity */
    SF_PAIR_TABLE_NONZERO = 10,       /* table[key[left]][key[right]] != 0      fact = i64 matrix, fact_b = i32 key column (row / column index) */
    SF_PAIR_VALUE_ABSDIFF_LE = 11     /* |left.value - right.value| <= param */
} sf_pair_op;
typedef struct sf_pair_term {
    int32_t op;      /* sf_pair_op */
    int32_t clause;  /* terms with the same clause id are OR-ed, the clauses are AND-ed; ids ascend along the array */
    int32_t fact;    /* column / CSR / matrix fact id (ops that take one), else -1 */
    int32_t fact_b;  /* SF_PAIR_TABLE_NONZERO: the key column, else -1 */
    int64_t param;
} sf_pair_term;
/* for_each(E).join(for_each(E), predicate).penalize(weight) on the scalar class `descriptor_index`: every pair left.id < right.id of
 * ASSIGNED entities for which the program holds costs `weight` on `level` (IncrementalBiConstraint over a predicate join:
 * constraint/cross_bi_incremental/{state,incremental}.rs).  One predicate join per scalar class; <= 8 terms, of which <= 6 remain after
 * the partner index took its clause.  SF_ERR_INVALID for malformed programs / facts, SF_ERR_UNSUPPORTED beyond the limits. */
int32_t sf_constraint_add_pair_join(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, const sf_pair_term* terms, int32_t n_terms,
                                    int32_t level, int64_t weight);

/* ---- uni filters / weights as data (round 6) ---------------------------------------------------------------------------------
 * for_each(A).filter(|a| pred(a, a.value)).penalize(w(a, a.value)) -- IncrementalUniConstraint with its two closures
 * (crates/solverforge-scoring/src/constraint/incremental.rs:19-160: on_insert / on_retract test the filter and score the weight of ONE entity).
 * Both closures see one entity and its assigned value only, so the program is COMPILED at sf_initialize: the host evaluates predicate and weight for
 * every (entity, value) pair into the cost[n_rows][n_values] matrix that SF_C_VALUE_COST prices on the device -- no interpreter in any kernel, an
 * unassigned entity never matches.  Programs (and at most one SF_C_VALUE_COST matrix) of one class fold into one matrix per score level, on at most
 * TWO levels (a hard filter beside soft weights); sf_evaluate_each still reports each of them on its own row (from per-constraint host copies).
 * Predicate: a conjunction of clauses, each a disjunction of its terms (as sf_pair_term); term = `lhs cmp param` with lhs one of: */
typedef enum sf_uni_lhs {
    SF_UNI_ROW_COL = 1,       /* fact = i32 column over the entities: col[a] */
    SF_UNI_VALUE = 2,         /* the assigned value itself */
    SF_UNI_VALUE_COL = 3,     /* fact = i32 column over the values: col[value] */
    SF_UNI_COL_DIFF = 4,      /* fact (entities), fact_b (values): col_a[a] - col_b[value] */
    SF_UNI_COL_ABSDIFF = 5,   /* |col_a[a] - col_b[value]| */
    SF_UNI_TABLE = 6          /* fact_c = i64 matrix, keyed by fact / fact_b: table[col_a[a]][col_b[value]]  (fact = -1: row key = a; fact_b = -1: column key = value) */
} sf_uni_lhs;
typedef enum sf_uni_cmp { SF_UNI_EQ = 0, SF_UNI_NE = 1, SF_UNI_LT = 2, SF_UNI_LE = 3, SF_UNI_GT = 4, SF_UNI_GE = 5 } sf_uni_cmp;
typedef struct sf_uni_term {
    int32_t lhs;     /* sf_uni_lhs */
    int32_t cmp;     /* sf_uni_cmp */
    int32_t clause;  /* ids ascend along the array; same id = OR, different ids = AND */
    int32_t fact, fact_b, fact_c; /* as the lhs says, else -1 */
    int64_t param;
} sf_uni_term;
/* the weight closure: scale * max(0, lhs-expression) of the same operand kinds (SF_UNI_* above; lhs 0 = the constant 1) */
typedef struct sf_uni_weight {
    int32_t lhs;     /* 0 = constant 1, or sf_uni_lhs */
    int32_t fact, fact_b, fact_c;
} sf_uni_weight;
/* n_terms == 0: no filter (every assigned entity matches).  <= 16 terms.  SF_ERR_INVALID for malformed programs / facts (checked at sf_initialize
 * where the facts are known), SF_ERR_UNSUPPORTED when the class's value-cost constraints sit on more than two levels. */
int32_t sf_constraint_add_uni_program(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, const sf_uni_term* terms, int32_t n_terms,
                                      const sf_uni_weight* weight, int32_t level, int64_t scale);

typedef enum sf_selector_kind {
    SF_SEL_SCALAR_CHANGE = 1,      /* selector/scalar_neighborhood/cursor/change.rs:27-121 */
    SF_SEL_SCALAR_SWAP = 2,        /* selector/scalar_neighborhood/cursor/swap.rs:22-160 */
    SF_SEL_LIST_CHANGE = 4,        /* selector/list_kernel/change.rs:25-241 */
    SF_SEL_LIST_SWAP = 8,          /* selector/list_kernel/swap.rs:25-270 */
    SF_SEL_NEARBY_LIST_CHANGE = 16,/* selector/list_kernel/nearby_change.rs:17-233 */
    SF_SEL_NEARBY_LIST_SWAP = 32,  /* selector/list_kernel/nearby_swap.rs:17-260 */
    SF_SEL_LIST_REVERSE = 64,      /* selector/list_kernel/reverse.rs:12-108 (intra-list 2-opt) */
    SF_SEL_SUBLIST_CHANGE = 128,   /* selector/list_kernel/sublist_change.rs:13-266 (Or-opt); sizes via sf_selector_add_sublist */
    SF_SEL_KOPT = 512,             /* 3-opt leaf, sf_selector_add_kopt: max_nearby > 0 = distance-pruned cursor
                                      (selector/list_kernel/k_opt/nearby.rs, nearby_state.rs; the default-policy leaf of lists
                                      with an intra-distance meter, policy/list.rs:144-160), 0 = full enumeration
                                      (selector/list_kernel/k_opt/full.rs) */
    SF_SEL_SUBLIST_SWAP = 256,     /* selector/list_kernel/sublist_swap.rs:13-330; sizes via sf_selector_add_sublist */
    SF_SEL_LIST_PERMUTE = 8192,   /* selector/list_kernel/permute.rs:22-205 (contiguous-window permutations); sf_selector_add_permute */
    SF_SEL_LIST_PRECEDENCE = 16384, /* selector/list_precedence.rs:121-210 over list_kernel/precedence/{analysis,coordinates,support,cursor,emission}.rs
                                      (the critical-path leaf); sf_selector_add_precedence.  First list leaf of the default policy
                                      for slots with precedence hooks (policy/list.rs:24-33,62-93) */
    SF_SEL_NEARBY_SCALAR_CHANGE = 2048, /* scalar_neighborhood/cursor/change.rs:123-392 (NearbyChangeCursor); sf_selector_add_nearby_scalar */
    SF_SEL_NEARBY_SCALAR_SWAP = 4096,   /* scalar_neighborhood/cursor/swap.rs:162-414 (NearbySwapCursor); sf_selector_add_nearby_scalar */
    SF_SEL_LIST_RUIN = 1024        /* selector/list_kernel/ruin.rs:38-144 + move/list_kernel/ruin.rs:131-281 (ruin and greedy recreate);
                                      sf_selector_add_ruin.  Last list leaf of the default policy (policy/list.rs:24-33,193-199) */
} sf_selector_kind;

typedef enum sf_selection_order { /* solverforge_config::SelectionOrder */
    SF_ORDER_ORIGINAL = 0, SF_ORDER_SORTED = 1, SF_ORDER_PROBABILISTIC = 2,
    SF_ORDER_RANDOM = 3, SF_ORDER_SHUFFLED = 4
} sf_selection_order;

typedef enum sf_acceptor_kind {
    SF_ACCEPT_HILL_CLIMBING = 0,  /* phase/localsearch/acceptor/hill_climbing.rs:33-41 */
    SF_ACCEPT_LATE_ACCEPTANCE = 1,/* phase/localsearch/acceptor/late_acceptance.rs:89-125 */
    /* 2 is reserved (internal "never accept" of the dry-run enumeration) */
    SF_ACCEPT_SIMULATED_ANNEALING = 3,/* phase/localsearch/acceptor/simulated_annealing.rs:11-430; the default of
                                       * scalar-only models (default_local_search/policy.rs:56-61).  Parameters:
                                       * sf_solver_configure_annealing; without it the reference defaults apply
                                       * (auto-calibrated, decay 0.999985, rng seed = random_seed). */
    SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE = 4 /* phase/localsearch/acceptor/diversified_late_acceptance.rs:40-176; the default
                                       * of grouped scalar-only models (default_local_search/policy.rs:52-55): late acceptance
                                       * over sf_solver_config::late_acceptance_size steps, or within `tolerance` of the best step
                                       * score of the phase (sf_solver_configure_diversified, default 0.01).  Wave, scalar and
                                       * generic engines (the block engine reports SF_ERR_UNSUPPORTED). */
} sf_acceptor_kind;

typedef enum sf_annealing_mode {
    SF_ANNEAL_SINGLE = 0,     /* SimulatedAnnealingAcceptor::with_seed: one temperature for every level (:115-133) */
    SF_ANNEAL_PER_LEVEL = 1,  /* with_level_temperatures_and_seed (:140-154) */
    SF_ANNEAL_CALIBRATED = 2  /* auto_calibrate_with_seed / with_calibration_and_seed (:180-220) */
} sf_annealing_mode;

/* SimulatedAnnealingConfig of the reference (builder/acceptor.rs:270-335) */
typedef struct sf_annealing_config {
    int32_t mode;                          /* sf_annealing_mode */
    int32_t never_accept_hard_regression;  /* HardRegressionPolicy (:18-21); levels < hard_levels are Hard */
    int32_t calibration_sample_size;       /* default 128 (:13) */
    int32_t reserved;
    double temperatures[4];                /* SINGLE: [0]; PER_LEVEL: one per score level */
    double decay_rate;                     /* (0, 1]; default 0.999985 (:11) */
    double hill_climbing_temperature;      /* default 1e-9 (:12) */
    double target_acceptance_probability;  /* (0, 1); default 0.80 (:14) */
    double fallback_temperature;           /* default 1.0 (:15) */
    uint64_t seed;                         /* SmallRng::seed_from_u64(seed + replica) */
} sf_annealing_config;

typedef enum sf_forager_kind {
    SF_FORAGER_ACCEPTED_COUNT = 0, /* phase/localsearch/forager.rs:157-250 */
    SF_FORAGER_FIRST_ACCEPTED = 1, /* phase/localsearch/forager.rs:252-337 */
    SF_FORAGER_BEST_SCORE = 2,     /* phase/localsearch/forager.rs:339-420 */
    /* phase/localsearch/forager/improving.rs:17-107: the step ends at the first accepted candidate that beats the best
     * score ever seen; otherwise the best accepted candidate of the whole neighbourhood */
    SF_FORAGER_FIRST_BEST_SCORE_IMPROVING = 3,
    /* improving.rs:113-227: the step ends at the first accepted candidate that beats the last step score, or at
     * accepted_count_limit accepted candidates (<= 0 = no limit).  The default forager of precedence / list models
     * (limit 256) and of grouped scalar models (no limit): default_local_search/policy.rs:63-71 */
    SF_FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING = 4
} sf_forager_kind;

/* Search engines of the fused local-search kernel (same results, different GPU mapping):
 * WAVE  = one wavefront per replica, presorted neighbour index (many small replicas);
 * BLOCK = one 1024-thread workgroup per replica, matrix-row scan (large problems);
 * AUTO  = WAVE when a replica's LDS slice allows >= 2 replicas per CU (<= 80 KiB), else BLOCK. */
typedef enum sf_engine_kind { SF_ENGINE_AUTO = 0, SF_ENGINE_BLOCK = 1, SF_ENGINE_WAVE = 2 } sf_engine_kind;

typedef struct sf_solver_config {
    int32_t acceptor;            /* sf_acceptor_kind */
    int32_t late_acceptance_size;/* default 400: runtime/compiler/default_local_search/policy.rs:18 */
    int32_t forager;             /* sf_forager_kind */
    int32_t accepted_count_limit;/* default 256: policy.rs:19 */
    int32_t random_ties;         /* ScoreTieBreak::Random = 1 (solverforge-config/src/forager.rs:5-9) */
    int32_t selection_order;     /* sf_selection_order; default policy = SF_ORDER_RANDOM */
    uint64_t random_seed;        /* replica r searches with random_seed + r */
} sf_solver_config;

/* SolverStats counters (crates/solverforge-solver/src/stats/solver.rs:23,112-119,246). */
typedef struct sf_stats {
    uint64_t step_count;
    uint64_t moves_generated;     /* candidates pulled from the cursor */
    uint64_t moves_evaluated;     /* consumed candidates, incl. not-doable (evaluation.rs:33-49) */
    uint64_t moves_accepted;
    uint64_t moves_applied;
    uint64_t score_calculations;  /* scored trials only (evaluation.rs:60) */
    uint64_t moves_not_doable;
    uint64_t candidates_scored;   /* device work incl. the speculative tail of each step */
    uint64_t sources_scanned;     /* generator calls that produced candidates (nearby sources scanned, batches of the precedence leaf):
                                     engine-specific generation work -- depends on the launch shape and the trials per wave, never
                                     compared with the oracle (the seven parity counters are the fields above) */
    uint64_t reserved;
} sf_stats;

/* ---- context ------------------------------------------------------------------------- */
int32_t sf_ctx_create(int32_t device_id, int32_t score_levels, int32_t hard_levels,
                      int32_t n_replicas, sf_ctx** out);
void sf_ctx_destroy(sf_ctx* ctx);
const char* sf_last_error(const sf_ctx* ctx); /* ctx may be NULL: last create error */
int32_t sf_device_count(void);
int32_t sf_sync(sf_ctx* ctx); /* hipStreamSynchronize on the context stream */

/* ---- schema: SolutionDescriptor mirror (descriptor_index / variable_index addressing) ---- */
int32_t sf_schema_add_entity_class(sf_ctx* ctx, int32_t descriptor_index, int32_t n_rows);
/* scalar planning variable, value range 0..n_values (ValueSource::CountableRange / SolutionCount);
 * `initial[n_rows]` int32, SF_NONE = unassigned; replicated to every replica */
int32_t sf_schema_add_scalar_variable(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index,
                                      int32_t n_values, int32_t allows_unassigned,
                                      const int32_t* initial);
/* ValueSource::EntitySlice (crates/solverforge-solver/src/builder/context/scalar/variable.rs:138-151; ValueSelector::iter,
 * heuristic/selector/value_selector.rs:21-41): the canonical value list of EVERY entity of a declared scalar variable as CSR --
 * offsets[n_rows + 1], values[offsets[n_rows]], each value in 0..n_values.  The change stream draws from the entity's list, a
 * swap needs each value in the other row's list (cursor/swap.rs:103-123), a compound edit's value must be in its entity's list.
 * Without this call every entity has the countable range 0..n_values. */
int32_t sf_schema_set_value_lists(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, const uint32_t* offsets,
                                  const int32_t* values);
/* list planning variable as CSR: offsets[n_rows+1], values[offsets[n_rows]];
 * `element_capacity` = max total elements (element ids < element_id_bound) */
int32_t sf_schema_add_list_variable(sf_ctx* ctx, int32_t descriptor_index,
                                    const uint32_t* offsets, const uint32_t* values,
                                    int32_t element_capacity, int32_t element_id_bound);
/* problem facts (immutable, shared by all replicas) */
int32_t sf_fact_matrix_i64(sf_ctx* ctx, int32_t fact_id, int32_t rows, int32_t cols, const int64_t* data);
int32_t sf_fact_column_i32(sf_ctx* ctx, int32_t fact_id, int32_t n, const int32_t* data);
int32_t sf_fact_column_u32(sf_ctx* ctx, int32_t fact_id, int32_t n, const uint32_t* data);
int32_t sf_fact_csr_u32(sf_ctx* ctx, int32_t fact_id, int32_t n_rows, const uint32_t* offsets,
                        const uint32_t* values);

/* ---- constraints / selectors ------------------------------------------------------------- */
/* weight is added to score level `level` as a penalty (ImpactType::Penalty). */
int32_t sf_constraint_add(sf_ctx* ctx, int32_t kind, int32_t descriptor_index, int32_t variable_index,
                          int32_t fact_a, int64_t param, int32_t level, int64_t weight);
/* ListPrecedenceMakespanConstraint::new(...).with_expected_owner(...) (crates/solverforge-scoring/src/constraint/list_precedence.rs:
 * 29-59) on the list variable of `descriptor_index`, its hooks as data: node_count nodes = list element ids, durations[node_count]
 * (node_duration), the fixed successor relation as CSR (fixed_successors; successors >= node_count count as invalid fixed edges),
 * expected_owner[node_count] (-1 = no expectation) or NULL for no hook.  Score: -(invalid fixed edges + wrong-owner items +
 * unassigned nodes + node_count if the graph of fixed + consecutive-list-item edges is cyclic) on `hard_level`, -(longest
 * duration-weighted path, 0 when cyclic) on `makespan_level` (the reference fixes HardSoftScore::of(-penalty, -makespan)).
 * Every trial is one full evaluation of the trial lists (Kahn over the whole graph by one wavefront); the search runs in the
 * generic N-leaf engine.  Limits: element ids < node_count, every element in at most one list position, sum of durations < 2^31.
 * The constraint's successors and durations double as the list slot's precedence hooks: the critical-path leaf
 * (sf_selector_add_precedence), the slot's precedence policy (sf_list_set_precedence_policy), the ruin leaf and ruin records
 * (recreated by this constraint alone: no distance / capacity constraint on the list class) and cheapest insertion
 * (sf_construct_list_cheapest) read them; round-robin construction scores no trial and takes the model as it is; Clarke-Wright and
 * ListKOpt (distance-matrix route hooks) refuse it. */
int32_t sf_constraint_add_list_precedence(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, int32_t node_count,
                                          const int32_t* durations, const uint32_t* succ_offsets, const uint32_t* succ_values,
                                          const int32_t* expected_owner, int32_t hard_level, int32_t makespan_level);
/* leaves are unioned in default-policy declaration order; >1 leaf => StratifiedRandom with equal
 * weights, 1 leaf => Sequential (default_local_search/policy.rs:104-108).
 * `fact_meter` = i64 matrix for MatrixDistanceMeter (crates/solverforge-cvrp/src/meters.rs:10-28). */
int32_t sf_selector_add(sf_ctx* ctx, int32_t kind, int32_t descriptor_index, int32_t variable_index,
                        int32_t max_nearby, int32_t fact_meter);
/* sublist leaves: segment sizes min_size..=max_size (default 1..=3, solverforge-config/src/move_selector.rs:713-715; <= 15) */
int32_t sf_selector_add_sublist(sf_ctx* ctx, int32_t kind, int32_t descriptor_index, int32_t variable_index,
                                int32_t min_size, int32_t max_size);
/* k-opt leaf (KOptMoveSelectorConfig, solverforge-config/src/move_selector.rs:521-550): k = 3 only; min_segment_len default 1;
 * max_nearby = 0 -> full enumeration, 1..64 -> distance-pruned by the list's matrix meter (the default policy passes 20) */
int32_t sf_selector_add_kopt(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, int32_t k,
                             int32_t min_segment_len, int32_t max_nearby);

/* ListPermuteMoveSelector (ListPermuteMoveSelectorConfig: min_window_size 2, max_window_size 5 by default; the default policy
 * declares it beside ListPrecedence for list slots with precedence hooks, default_local_search/policy/list.rs:62-93): every
 * non-identity permutation of every window of min..=max consecutive elements.  2 <= min <= max <= 8.  No owner restrictions and no
 * precedence-route-graph cycle filter (the reference drops permutations that would close a cycle through the route graph,
 * permute.rs:139-147: not restated).  Generic N-leaf engine. */
int32_t sf_selector_add_permute(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, int32_t min_window_size, int32_t max_window_size);

/* ListPrecedenceMoveSelector (heuristic/selector/list_precedence.rs:121-210; ListPrecedenceMoveConfig has no tunables): the critical-path
 * neighbourhood of a list class that carries the ListPrecedenceMakespanConstraint -- the constraint's fixed successors and durations are
 * the selector's `fixed_successors` / `node_duration` hooks.  Every step: earliest / latest starts, critical blocks per list
 * (list_kernel/precedence/analysis.rs:56-112), then the stream of list_kernel/precedence/cursor.rs:182-252: three-list multi-swaps
 * (critical, critical, support; they require a score improvement, emission.rs:280-294), two-block ruins, then per block the tiered
 * families change / swap / reverse / adjacent sublist swap / ruin window / sublist change / permutation.  Candidates whose lists would be
 * cyclic are pruned before they count (coordinates.rs:265-326, precedence_route.rs:257-304); the ruins recreate with the precedence hooks
 * (move/list_kernel/ruin.rs:186-220: insertions that close a cycle are skipped).  Moves come back as SF_MOVE_LIST_CHANGE .. SF_MOVE_LIST_PERMUTE,
 * SF_MOVE_LIST_RUIN (value bits 31 / 30 set, see sf_move_t) and SF_MOVE_LIST_MULTI_SWAP.  Generic N-leaf engine; the list class may
 * carry no distance / capacity constraint (the ruin's recreate is scored by the precedence constraint alone; a flattened not-exists is
 * fine), fixed successor lists without repeats, at most 2,048 nodes (the multi-swap stream is indexed in 32 bits); SF_ERR_UNSUPPORTED
 * otherwise.  One such leaf per union. */
int32_t sf_selector_add_precedence(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index);

/* The precedence policy of the compiled runtime list slot (ListVariableSlot::with_precedence_hooks; RuntimeListSlot::precedence_policy):
 * with the slot's successors declared, every runtime list leaf -- change, swap, nearby change / swap, sublist change / swap, reverse,
 * permute -- opens its cursor `with_precedence_route_graph` (list_leaf/cursor/slot.rs:191-404) and drops INTRA-list candidates whose new
 * route edges close a cycle (precedence_route.rs:171-255,313-317,419-451; the 3-opt cursor is not filtered), and the ruin leaf recreates
 * with the hooks (ruin_access.rs:195-217, move/list_kernel/ruin.rs:186-220).  enabled = 0 (default) = the public selectors, which do not
 * know the hooks.  Needs the precedence constraint on the list class (its fixed successors are the hooks); may be changed between
 * launches.  The ruin leaf on a precedence model -- with or without the policy -- needs a list class without distance / capacity /
 * not-exists constraints. */
int32_t sf_list_set_precedence_policy(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, int32_t enabled);

/* Nearby scalar leaves of a scalar slot (NearbyChangeMoveSelector / NearbySwapMoveSelector; the default policy declares them with
 * max_nearby 10 between the list rules and the ordinary change / swap pair, default_local_search/policy/scalar.rs:18-65).  The
 * slot's hooks arrive as data: offsets[n_rows + 1] / candidates = the nearby source row of every entity in SOURCE order --
 * nearby_value_candidates (values, kind SF_SEL_NEARBY_SCALAR_CHANGE) or nearby_entity_candidates (entity indices,
 * SF_SEL_NEARBY_SCALAR_SWAP); a slot without the hook passes what the reference falls back to (its ordinary candidate values /
 * every entity 0..n).  distances = the slot's nearby_value_distance / nearby_entity_distance per row entry, NULL = no meter (the
 * source order ranks; non-finite = dropped).  source_limit = value_candidate_limit (<= 0: none; change leaf only).  dynamic_slot
 * != 0: a DynamicScalarVariableSlot (the change leaf re-checks value legality, the swap leaf emits directional pairs right != left
 * instead of right > left).  Per row: skip the current value (change) / equal values and illegal exchanges (swap), stable top
 * max_nearby by (distance, source order, candidate), then apply_selection_order; the change leaf ends an assigned row with its
 * to-None candidate when the variable allows unassigned.  max_nearby 1..63.  The leaves run in the generic N-leaf engine. */
int32_t sf_selector_add_nearby_scalar(sf_ctx* ctx, int32_t kind, int32_t descriptor_index, int32_t variable_index, int32_t max_nearby,
                                      int64_t source_limit, const uint32_t* offsets, const int32_t* candidates, const double* distances,
                                      int32_t dynamic_slot);

/* Root union of the configured leaves (UnionMoveSelectorConfig: UnionSelectionOrder + UnionWeighting; scheduler
 * heuristic/selector/decorator/vec_union.rs:190-365).  selection_order: sf_union_order, -1 = the default policy's choice
 * (StratifiedRandom for more than one leaf, runtime/compiler/executor/local_search/lower.rs:285-293); weights[n_weights] = one
 * unsigned weight per leaf in union (declaration) order -- a configured union keeps the order of the sf_selector_add calls for
 * its children (weights, Sequential / RoundRobin child order), the default policy's union uses the policy's own declaration order
 * (default_local_search/policy/list.rs:24-33) whatever the call order --, NULL / 0 = equal (UnionWeighting::Equal); a zero weight disables the
 * leaf; weights other than 1 need SF_UNION_RANDOM or SF_UNION_STRATIFIED_RANDOM (vec_union.rs:215-222).  A non-default root
 * union runs in the generic N-leaf engine.  The weight count is checked against the leaf count at the next launch. */
typedef enum sf_union_order {
    SF_UNION_SEQUENTIAL = 0, SF_UNION_ROUND_ROBIN = 1, SF_UNION_ROTATING_ROUND_ROBIN = 2, SF_UNION_RANDOM = 3,
    SF_UNION_STRATIFIED_RANDOM = 4
} sf_union_order;
int32_t sf_union_configure(sf_ctx* ctx, int32_t selection_order, const int64_t* weights, int32_t n_weights);

/* list ruin leaf (ListRuinMoveSelectorConfig, solverforge-config/src/move_selector.rs:552-587; defaults 2, 5, 10, none, false):
 * per step `moves_per_step` (<= 16) candidates, each removing min..=max (<= 6) elements of one non-empty list (no longer than
 * max_source_list_len; 0 = no bound) and re-inserting them greedily.  `variable_name` = the list variable's name: the leaf's
 * per-solve random stream is SmallRng::seed_from_u64(scoped_seed(random_seed + replica, descriptor_index, variable_name,
 * "list_ruin_move_selector")) (heuristic/selector/seed.rs:3-17, list_leaf/cursor.rs:117-145), (re)seeded by sf_phase_start.
 * rand's xoshiro256++ / random_range are restated from the published algorithm: parity unpinned against the reference. */
int32_t sf_selector_add_ruin(sf_ctx* ctx, int32_t descriptor_index, int32_t variable_index, int32_t min_ruin_count,
                             int32_t max_ruin_count, int32_t moves_per_step, int32_t max_source_list_len,
                             int32_t skip_empty_destinations, const char* variable_name);

/* ---- Director surface -------------------------------------------------------------------- */
/* ≙ first Director::calculate_score (initialize_all): builds per-replica aggregates.
 * out_scores[n_replicas * score_levels] (may be NULL). */
int32_t sf_initialize(sf_ctx* ctx, int64_t* out_scores);
/* ≙ Director::fresh_score (evaluate_all from scratch; FullAssert check) */
int32_t sf_evaluate_all(sf_ctx* ctx, int64_t* out_scores);
/* committed (cached) score of every replica */
int32_t sf_get_scores(sf_ctx* ctx, int64_t* out_scores);
/* ≙ ConstraintSet::evaluate_each (api/constraint_set/incremental.rs:172,237-244): score and match count of every
 * declared constraint (declaration order = sf_constraint_add order) on `replica`'s working solution, by full
 * recomputation.  out_scores[n_constraints * score_levels], out_match_counts[n_constraints]. */
int32_t sf_evaluate_each(sf_ctx* ctx, int32_t replica, int64_t* out_scores, int64_t* out_match_counts);
/* ≙ n x evaluate_candidate (phase/localsearch/evaluation.rs:20-115) against replica `replica`:
 * one launch, state unchanged.  out_scores[n * score_levels], out_doable[n]. */
int32_t sf_step_evaluate(sf_ctx* ctx, int32_t replica, const sf_move_t* moves, int64_t n,
                         int64_t* out_scores, int32_t* out_doable);
/* ≙ committed Move::do_move + before/after_variable_changed + calculate_score */
int32_t sf_apply(sf_ctx* ctx, int32_t replica, const sf_move_t* move);
/* ScalarCandidateProvider surface (crates/solverforge-solver/src/planning/scalar/candidate.rs:85-190): a ScalarCandidate carries
 * several ScalarEdits and is scored / applied as ONE move (CompoundScalarMove, heuristic/move/compound_scalar.rs:207-330: retract
 * every affected entity, apply every edit in order, insert in reverse order; doable = at least one edit, every to_value legal,
 * some edit differs from its entity's current value).  Candidate i = edits[offsets[i] .. offsets[i + 1]), every edit a
 * SF_MOVE_CHANGE-shaped record (a = entity_index, value = to_value), at most 8 edits per candidate.  One launch, state unchanged.
 * out_scores[n * score_levels], out_doable[n].  SF_ERR_UNSUPPORTED on a load_balance model. */
int32_t sf_step_evaluate_compound(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, int64_t n,
                                  int64_t* out_scores, int32_t* out_doable);
/* One HOST-DRIVEN local-search step of replica `replica` whose cursor is a GroupedScalarMoveSelector over a candidate-backed
 * ScalarGroup (builder/selector/grouped_scalar.rs:82-176; declared by the default policy for every scalar group,
 * default_local_search/policy/scalar.rs:108-136): the ScalarCandidateProvider is a host closure over the working solution
 * (planning/scalar/candidate.rs:190), so the host calls it (sf_download_scalar gives the solution) and hands its output over --
 * candidate i = edits[offsets[i] .. offsets[i + 1]) as in sf_step_evaluate_compound.  The library restates the cursor's activation:
 * apply_selection_order with the salt 0xC0A1_E5CE_AAA0_0001 ^ group_name_len under the replica's step context (step index, step
 * seed, configured selection order); candidates without edits, repeats of a kept candidate, candidates with two edits on one
 * entity, with an illegal value or not doable are skipped; at most max_moves_per_step (<= 0: the default 256,
 * ScalarGroupLimits / grouped_scalar.rs:27-40) are kept.  The kept candidates are pulled in order through the configured acceptor
 * (HillClimbing / LateAcceptance / DiversifiedLateAcceptance) and forager exactly like a fused step (phase/candidates.rs:47-285), the
 * pick is committed and the step ends (acceptor history, best solution, counters, step index, step-seed draw).
 * out_kept[n] / *out_n_kept: provider indices of the kept candidates in pull order; out_scores[n * score_levels] trial scores and
 * out_flags[n] (bit0 doable, bit1 accepted, bit2 committed) of the *out_consumed pulled ones; *out_selected = ordinal of the
 * committed candidate in out_kept, -1 when the step applied nothing.  Candidate equality is equality of the edit lists (the
 * reference also compares the candidate's reason and construction keys, which do not cross this boundary).  Scalar-only models. */
int32_t sf_step_decide(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, int64_t n, int32_t group_name_len,
                       int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores, int32_t* out_flags,
                       int64_t* out_consumed, int64_t* out_selected);
/* The same step for candidates that carry evaluate_candidate's gates (phase/localsearch/evaluation.rs:75-113): gates[i] bit 0 =
 * Move::requires_hard_improvement (the candidate is rejected unless hard_score_delta(last step score, move score) is Improving:
 * the first differing HARD level is greater, phase/hard_delta.rs:11-35; conflict-repair candidates of the runtime provider cursor,
 * runtime/provider_cursor.rs), bit 1 = Move::requires_score_improvement (rejected unless move score > last step score).  A rejected
 * candidate is scored and counted (moves_evaluated, score_calculations) but never reaches the acceptor; its flags read doable,
 * not accepted, plus bit 3 (RejectedByHardImprovement) or bit 4 (RejectedByScoreImprovement), the candidate-trace dispositions
 * 4 / 5 of stats/candidate_trace.rs:530-543.  gates = NULL: no gate (sf_step_decide). */
int32_t sf_step_decide_gated(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                             int32_t group_name_len, int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores,
                             int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected);
/* The same step over the pull order of a cursor that has done its own activation -- the reference's RuntimeProviderCursor
 * (runtime/provider_cursor.rs:37-492; its leaf, runtime/compiler/executor/local_search/leaf.rs:362-402), which serves BOTH the
 * grouped-scalar leaf and the compound conflict-repair leaf of the default scalar policy (default_local_search/policy/scalar.rs:107-190):
 * the cursor has rotated (apply_selection_order with the binding's salts), normalised, deduplicated per provider scope, capped
 * (max_matches_per_step / max_repairs_per_match / max_moves_per_step) and pushed doable moves only (provider_cursor.rs:420-437); the
 * library restates none of that.  Candidate i is pull i: scored, gated (gates[i] as above; a conflict-repair leaf sets bit 0 when its
 * config says require_hard_improvement), shown to the acceptor and the forager, the pick committed and the step ended exactly like
 * sf_step_decide_gated.  A candidate that is not doable on the working solution is pulled and counted (moves_generated, moves_evaluated,
 * moves_not_doable; flags 0) like evaluate_candidate does (phase/localsearch/evaluation.rs:33-49).  Candidates without edits, with two
 * edits on one entity or with a value outside the entity's range are SF_ERR_INVALID (the cursor's normalisation removes them).
 * out_scores[n * score_levels], out_flags[n]; *out_selected = index of the committed candidate, -1 when the step applied nothing. */
int32_t sf_step_decide_cursor(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                              int64_t* out_scores, int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected);
/* committed do_move of one multi-edit candidate */
int32_t sf_apply_compound(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, int64_t n_edits);

/* ≙ ListCheapestInsertionPhase (crates/solverforge-solver/src/manager/phase_factory/list_construction/cheapest.rs; kernel
 * cheapest/kernel.rs:57-150, bookkeeping cheapest/live.rs:64-170) on EVERY replica's current lists: the elements of
 * `elements[n]` (source order; typically the A-side keys of the not-exists constraint) that are in no list yet are placed one by
 * one at the (list, position) whose trial score is strictly best (the first of equal scores stays).  Unrestricted owners, no
 * construction order key.  On a list class with the precedence constraint (and no distance / capacity constraint) every slot of an
 * element is priced from one forward + one backward pass over the graph; with the slot's precedence policy
 * (sf_list_set_precedence_policy) the phase has the hooks (with_precedence_hooks, cheapest.rs:112-120) and places the elements in
 * descending order of their downstream chain of fixed successors (precedence_downstream, cheapest/kernel.rs:162-229).  Counters: one generated + evaluated candidate and one score_calculation per trial (live.rs:118-127), one accepted + applied step per
 * placed element.  Commits the score of the constructed lists; out_scores[n_replicas * score_levels] may be NULL. */
int32_t sf_construct_list_cheapest(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, int64_t* out_scores);

/* ≙ ListRegretInsertionPhase (crates/solverforge-solver/src/manager/phase_factory/list_construction/regret.rs:223-260; the loop
 * regret/kernel/execute.rs:52-204, an element's best / second-best trial regret/kernel/evaluation.rs:120-230, the order of choices
 * regret/kernel/mod.rs:19-75) on EVERY replica's current lists: every round prices every (list, position) of every element of
 * `elements[n]` (source order, no repeated id) that is in no list yet; an element's regret is its best minus its second-best trial
 * score (Forced, above every finite regret, when it has a single slot); the element with the greatest regret goes to its best slot
 * (the first of equal scores) -- ties: the better best score, then the earlier element in (construction order key, source index)
 * order; order_keys[n] (parallel to `elements`, may be NULL) = element_order_key (regret.rs:103-106); owners[n] (may be NULL) =
 * the owner hook per element as in sf_construct_list_round_robin: -1 unrestricted, a list index = only that list's slots are
 * candidates (candidate_entities, regret/kernel/mod.rs:104-114), a value >= the list count = never placed.  The bounded fallbacks
 * the reference takes when the fixed-owner elements alone exceed its trial budget (sum len (len + 1) (len + 2) / 6 > 16,384 per
 * owner bucket, regret/kernel/fallback.rs:58-84) are not built: SF_ERR_UNSUPPORTED.  SF_ERR_UNSUPPORTED also on a list class with precedence hooks (regret/kernel/precedence.rs, fallback.rs are not built).  Counters:
 * one generated + evaluated candidate and one score_calculation per trial, one accepted + applied step per placed element.  Commits
 * the score of the constructed lists; out_scores[n_replicas * score_levels] may be NULL. */
int32_t sf_construct_list_regret(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, const int64_t* order_keys,
                                 const int32_t* owners, int64_t* out_scores);

/* ≙ ListKOptPhase (crates/solverforge-solver/src/manager/phase_factory/list_k_opt.rs; kernel list_k_opt/kernel.rs:57-220), the
 * route-local 2-opt polishing the default construction runs after Clarke-Wright, with the stock CVRP route hooks (the model's
 * depot, distance_cost legs of the attached matrix).  k: only 2 is implemented by the reference, every other value is a scored
 * no-op.  feasible_mode 0 = no feasibility hook, 1 = the capacity test of route_hooks::feasible (an over-capacity route takes no
 * reversal).  Every route with >= 4 visits of every replica is swept to its 2-opt local optimum in the reference's candidate
 * order (first improving reversal applied in place); max_sweeps >= 1 bounds the sweeps per route (the phase's termination policy:
 * on an asymmetric metric the 2-opt delta ignores the reversed inner legs and need not converge).  Counters: one generated + evaluated candidate per (i, j), one accepted move
 * per reversal, applied = the accepted reversals of a changed route, one step + score calculation per changed route.  Commits the
 * score of the resulting lists; out_scores[n_replicas * score_levels] may be NULL. */
int32_t sf_construct_list_k_opt(sf_ctx* ctx, int32_t descriptor_index, int32_t k, int32_t feasible_mode, int32_t max_sweeps, int64_t* out_scores);

/* ≙ ListConstructionPhase, the round-robin list construction (crates/solverforge-solver/src/manager/phase_factory/
 * list_construction/round_robin.rs; kernel round_robin/kernel.rs:71-175).  elements[n] = the declared elements in source order
 * (distinct); order_keys[n] (may be NULL) = construction_order_key per element; owners[n] (may be NULL) = the owner hook's value
 * per element, -1 = unrestricted (a value >= the owner count is OwnerRestriction::Invalid: the element is skipped,
 * list_placement.rs:54-69).  In every replica the elements that are in no list yet are taken in (order key, source index)
 * order: an unrestricted one is appended to the round-robin cursor's owner and advances it, a fixed-owner one is appended to its
 * owner.  Counters as the kernel records them (one generated + evaluated candidate, one accepted + applied step, one score
 * calculation per appended element).  Commits the score of the resulting lists; out_scores[n_replicas * score_levels] may be
 * NULL. */
int32_t sf_construct_list_round_robin(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, const int64_t* order_keys,
                                      const int32_t* owners, int64_t* out_scores);

/* ≙ ListClarkeWrightPhase (crates/solverforge-solver/src/manager/phase_factory/list_clarke_wright.rs:196-330; kernel
 * list_clarke_wright/kernel.rs:59-472, savings.rs:9-18, route_state.rs, owner_assignment.rs, completion.rs) with the hook bundle of
 * the stock CVRP domain (crates/solverforge-cvrp/src/helpers.rs:40-87): one savings metric class for the whole fleet, the model's
 * depot, distance_cost legs of the attached matrix.  elements[n] = the declared elements in source order (distinct); elements
 * already in a list of a replica are not routed there, the depot's own value is never routed.  feasible_mode 0 =
 * savings_hooks::feasible (structural only: capacity stays scoreable, the merge runs until one route per owner set remains),
 * 1 = the capacity test of route_hooks::feasible (route demand <= capacity).  Every replica builds the routes on its empty
 * owners: savings per pair, a stable descending sort (saving, left, right), merge passes, owner matching, completion by savings
 * insertion when the routes outnumber the empty owners; when the reference leaves the lists untouched (no empty owner, nothing to
 * route, unmatched routes that cannot be completed) so does the replica.  out_committed[n_replicas] (may be NULL): 1 = routes
 * committed.  Commits the score of the resulting lists; out_scores[n_replicas * score_levels] may be NULL.  Owner-restricted
 * elements (element_owner_fn), per-owner depots / metric classes / capacities and time windows are not modelled
 * (SF_ERR_UNSUPPORTED where detectable).  Solver counters are not advanced. */
int32_t sf_construct_list_clarke_wright(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, int32_t feasible_mode,
                                        int64_t* out_scores, int32_t* out_committed);

/* ---- MoveSelector / cursor surface ------------------------------------------------------- */
/* Opens the configured union cursor for MoveStreamContext(step_index, step_seed) with the given
 * selection order on replica `replica`, drains it, and returns every candidate in cursor order
 * together with its trial score (state unchanged).  out_scores/out_doable may be NULL. */
int32_t sf_step_generate(sf_ctx* ctx, int32_t replica, uint64_t step_index, uint64_t step_seed,
                         int32_t selection_order, sf_move_t* out_moves, int64_t* out_scores,
                         int32_t* out_doable, int64_t cap, int64_t* out_count);

/* ---- local search phase ------------------------------------------------------------------ */
int32_t sf_solver_configure(sf_ctx* ctx, const sf_solver_config* cfg);
/* compile_default_local_search_components (runtime/compiler/default_local_search/policy.rs:21-82) as a pure function of the five
 * model properties it reads: acceptor = LateAcceptance(400) with lists, DiversifiedLateAcceptance(400) for grouped scalar-only
 * models, SimulatedAnnealing otherwise; forager = FirstLastStepScoreImproving without a limit (accepted_count_limit 0) for grouped
 * scalar-only models, FirstLastStepScoreImproving(256) when a list slot supports precedence moves, else AcceptedCount(256 with
 * lists / nearby scalar leaves / conflict repairs, 1 otherwise).  Needs no device. */
int32_t sf_default_local_search_components(int32_t has_lists, int32_t has_groups, int32_t has_precedence, int32_t has_nearby_scalar,
                                           int32_t has_conflict_repairs, uint64_t random_seed, sf_solver_config* out);
/* sf_solver_configure with the components above, the model properties read from the context: has_lists = a list class is declared,
 * has_precedence = the list slot declares its precedence hooks (sf_list_set_precedence_policy) or carries the critical-path leaf
 * (list::supports_precedence_moves, policy/list.rs:287-299), has_nearby_scalar = a nearby scalar leaf is declared.  Scalar groups
 * and conflict repairs are host-side providers: the caller says whether the model has them.  out (may be NULL) = what was set. */
/* Host-side providers of the model, declared so that the default policy can be derived instead of passed in: a scalar group
 * (planning/scalar/group.rs: ScalarGroup with a candidate provider or an assignment rule) or a conflict repair
 * (planning/conflict_repair.rs:62-84: ConflictRepair::new(constraint_name, provider)).  The providers themselves stay on the host -- the
 * reference's RuntimeProviderCursor (runtime/provider_cursor.rs) pulls, normalises, caps and rotates their output and hands every
 * compound candidate to sf_step_decide_gated (gate bit 0 = its require_hard_improvement) -- the context only records that they exist. */
typedef enum sf_provider_kind { SF_PROVIDER_SCALAR_GROUP = 1, SF_PROVIDER_CONFLICT_REPAIR = 2 } sf_provider_kind;
int32_t sf_provider_declare(sf_ctx* ctx, int32_t kind, const char* name);
/* has_groups / has_conflict_repairs = -1: derived from sf_provider_declare (0 / 1: as passed, the behaviour of rounds 1-4) */
int32_t sf_solver_configure_default(sf_ctx* ctx, uint64_t random_seed, int32_t has_groups, int32_t has_conflict_repairs,
                                    sf_solver_config* out);
/* parameters of SF_ACCEPT_SIMULATED_ANNEALING (takes effect at the next sf_phase_start); validation follows
 * assert_simulated_annealing_parameters (simulated_annealing.rs:305-336) -> SF_ERR_INVALID instead of a panic */
int32_t sf_solver_configure_annealing(sf_ctx* ctx, const sf_annealing_config* cfg);
/* tolerance of SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE (DiversifiedLateAcceptanceAcceptor::new, diversified_late_acceptance.rs:86-98):
 * a candidate is also accepted when it is >= best - |best|.multiply(tolerance), each level rounded half away from zero */
int32_t sf_solver_configure_diversified(sf_ctx* ctx, double tolerance);
/* acceptor state of one replica: current temperatures [score_levels] and whether it is still calibrating */
int32_t sf_get_annealing_state(sf_ctx* ctx, int32_t replica, double* out_temperatures, int32_t* out_calibrating);
int32_t sf_solver_set_engine(sf_ctx* ctx, int32_t engine); /* sf_engine_kind; SF_ERR_UNSUPPORTED if it cannot run this model */
int32_t sf_solver_get_engine(sf_ctx* ctx, int32_t* out_engine); /* the engine launches resolve to (after sf_initialize) */
/* How the wave engine laid out its last fused launch (diagnostics; tests assert the path they mean to cover was taken):
 * out_mode = 0 general / 1 FAST / 2 FAST + 32-bit deltas / 3..7 the same on the COMPACT LDS slice (4, 5 = built for 5 / 6 waves per SIMD,
 * 6 = the node -> slot table of every replica in HBM instead of the slice: large models, more replicas per CU; 7 = the same built for
 * 8 waves per SIMD: small models, 32 replicas per CU),
 * -1 = no wave-engine launch yet; out_renumbered = 1 when that launch ran on the internal node numbering (the u16 matrix and the
 * neighbour index renumbered along a nearest-neighbour chain once they outgrow the L2: DESIGN 11.3; SF_AMD_RENUMBER=0 / 1 overrides).
 * The numbering is invisible at this boundary: every id that crosses it is the caller's. */
int32_t sf_list_wave_layout(sf_ctx* ctx, int32_t* out_mode, int32_t* out_renumbered);
/* explicit step seeds for parity runs (n_steps per replica, replica-major); NULL clears */
int32_t sf_solver_set_step_seeds(sf_ctx* ctx, const uint64_t* seeds, int64_t n_steps);
/* ≙ phase start: last_step_score = calculate_score, acceptor.phase_started, best = working.  Zeroes the replicas' counters (sf_get_stats
 * reports the phase's work; the counters of a construction call are read before it) and restarts the step-seed stream */
int32_t sf_phase_start(sf_ctx* ctx);
/* ≙ n_steps x execute_step (phase/localsearch/phase/step.rs:30-225) for EVERY replica, fused in
 * one persistent launch (generate -> trial-score -> accept -> forage -> apply). Asynchronous. */
int32_t sf_solve_steps(sf_ctx* ctx, int64_t n_steps);
/* Work-balanced launch for time-budgeted solves (termination by wall clock, not step count): every replica runs
 * whole steps until it has pulled >= move_budget candidates IN THIS LAUNCH or max_steps steps, whichever comes
 * first, so replicas whose steps are long (few accepted moves late in a search) do not hold the launch back while
 * the others sit idle.  Each replica's trajectory is still exactly the reference's for the steps it ran
 * (sf_get_stats(replica).step_count); replicas simply differ in how many steps they have completed. Asynchronous. */
int32_t sf_solve_moves(sf_ctx* ctx, int64_t max_steps, int64_t move_budget);
/* one traced step on every replica: per consumed candidate move/score/flags of replica `replica` (bit0 doable, bit1
 * accepted, bit2 = the forager's pick that the step committed, bit4 = rejected by the score-improvement gate before the
 * acceptor (ListMultiSwapMove of the critical-path leaf, evaluation.rs:95-113), bits 8..15 = MoveCursor::selector_index = the
 * leaf's position in the union); out_applied = 1 and *out_applied_move when a move was committed */
int32_t sf_solve_step_traced(sf_ctx* ctx, int32_t replica, sf_move_t* out_moves, int64_t* out_scores,
                             int32_t* out_flags, int64_t cap, int64_t* out_count,
                             int32_t* out_applied, sf_move_t* out_applied_move);
int32_t sf_get_stats(sf_ctx* ctx, int32_t replica, sf_stats* out);
int32_t sf_get_stats_sum(sf_ctx* ctx, sf_stats* out); /* counters summed over all replicas */
int32_t sf_get_best_scores(sf_ctx* ctx, int64_t* out_scores);
/* duration (ms, HIP events on the context stream) and launch count of sf_solve_steps launches
 * since the last call */
int32_t sf_profile_solve(sf_ctx* ctx, double* out_ms, int64_t* out_launches);

/* ---- state download ---------------------------------------------------------------------- */
int32_t sf_download_scalar(sf_ctx* ctx, int32_t replica, int32_t descriptor_index,
                           int32_t variable_index, int32_t* out, int32_t best);
int32_t sf_download_list(sf_ctx* ctx, int32_t replica, int32_t descriptor_index, uint32_t* out_offsets,
                         uint32_t* out_values, int32_t best);

/* ---- portfolio (multi-GPU): RCCL all-gather of (score levels, rank) + lexicographic max ---- */
int32_t sf_portfolio_unique_id(uint8_t* out_id128);          /* rank 0: ncclGetUniqueId */
int32_t sf_portfolio_init(sf_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world_size);
/* gathers every rank's best score over xGMI; all ranks get the same winner */
int32_t sf_portfolio_allgather_best(sf_ctx* ctx, int64_t* out_best_score, int32_t* out_winner_rank,
                                    int32_t* out_winner_replica);
/* ncclBroadcast of the winner's best list variable (offsets [n_owners + 1], values [offsets[n_owners]]) from
 * (winner_rank, winner_replica) to every rank; out buffers sized like sf_download_list's */
int32_t sf_portfolio_broadcast_best(sf_ctx* ctx, int32_t winner_rank, int32_t winner_replica, uint32_t* out_offsets,
                                    uint32_t* out_values);
int32_t sf_portfolio_destroy(sf_ctx* ctx);
/* EXTENSION (no reference counterpart; never called by parity runs): elite migration between the replicas of ONE context.  Replicas are
 * ranked by best score (descending, ties to the lower index); the n_replace last ones adopt the best solution of the n_elite first ones
 * (adopter i takes elite i % n_elite) as their working and best solution: per-route aggregates rebuilt, cached / last-step / best score =
 * the elite's best score, LateAcceptance history restarted at that score; counters, step index and step seeds keep running, so the copies
 * diverge.  An adopter whose best score already equals its elite's is left alone.  List-only models, HC / LA / DLA acceptors; call between
 * launches (after sf_phase_start).  After a migration a replica's trajectory is no longer the reference's single-chain trajectory.
 * out_adopted (may be NULL) = replicas that took a copy. */
int32_t sf_portfolio_migrate_local(sf_ctx* ctx, int32_t n_elite, int32_t n_replace, int32_t* out_adopted);


/* ---- candidate trace, wire format v3 (stats/candidate_trace.rs:15-60,718-812) ---------------
 * Host-side framing of what sf_solve_step_traced returned: every pull becomes the reference's canonical
 * CandidatePullTelemetry bytes and is folded into the two-lane CandidateTraceDigest, so a whole run's pull order,
 * identities and dispositions compare against a reference run's `prefix_digest` as one 128-bit value. */
typedef struct sf_trace_digest {
    uint64_t first;  /* FNV-1a 64 lane   (candidate_trace.rs:30-31,50-51) */
    uint64_t second; /* rotate-multiply lane (:32-33,53-58) */
} sf_trace_digest;
typedef struct sf_trace_scope {
    int32_t phase_index;         /* CandidatePullTelemetry::phase_index (0-based index of the phase in the run) */
    const char* phase_type;      /* "Local Search" for LocalSearchPhase (phase/localsearch/phase.rs:253) */
    int32_t list_descriptor;     /* descriptor_index + variable_name of List* / k-opt moves */
    const char* list_variable;
    int32_t scalar_descriptor;   /* descriptor_index + variable_name of Change / Swap moves */
    const char* scalar_variable;
} sf_trace_scope;
void sf_trace_digest_init(sf_trace_digest* d);                                     /* CandidateTraceDigest::empty */
void sf_trace_digest_update(sf_trace_digest* d, const void* bytes, size_t n);      /* CandidateTraceDigest::update */
/* Frames the n pulls of ONE traced step (moves/flags as sf_solve_step_traced wrote them; `first_ordinal` = pulls
 * recorded before this step; `step_index` = steps the phase had completed).  Writes the canonical bytes to `out`
 * (may be NULL: size query) and updates `digest` (may be NULL) pull by pull.  Every sf_move_t kind has its identity: the
 * scalar and list families, k_opt, and list_permute / list_ruin (sources merged per list) / list_multi_swap
 * (runtime/compiler/executor/list_leaf/move.rs:403-447, heuristic/move/list_kernel/permute.rs:170-188); flag bits 3 / 4
 * frame RejectedByHardImprovement / RejectedByScoreImprovement.  Returns the byte count, or
 * SF_ERR_INVALID (unknown move kind) / SF_ERR_CAPACITY (`out` too small). */
int64_t sf_trace_encode_step(const sf_trace_scope* scope, uint64_t first_ordinal, uint64_t step_index,
                             const sf_move_t* moves, const int32_t* flags, int64_t n, uint8_t* out, int64_t cap,
                             sf_trace_digest* digest);

#ifdef __cplusplus
}
#endif
#endif /* SOLVERFORGE_AMD_H */
