"""Host-side mirror of the reference seams on top of the C ABI.

GpuScoreDirector ≙ Director<S> (crates/solverforge-scoring/src/director/traits.rs:27-95) plus the
MoveCursorSource / local-search phase surface
(crates/solverforge-solver/src/phase/localsearch/cursor_source.rs:23-48, phase.rs:237-320).
Same names, argument meaning and error behaviour as the reference where a counterpart exists:
invalid use raises (the reference panics), there are no silent fallbacks.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import MOVE_DTYPE, AnnealingConfigStruct, SolverConfigStruct, SolverForgeError, StatsStruct, check, ptr


class MoveKind:
    CHANGE, SWAP, LIST_CHANGE, LIST_SWAP, LIST_REVERSE, SUBLIST_CHANGE, SUBLIST_SWAP = 0, 1, 2, 3, 4, 5, 6
    KOPT = 7  # a = list, a_pos / b / b_pos = the three cuts, value = reconnection pattern
    LIST_RUIN = 8  # a = list, a_pos = count, six 16-bit ascending positions in b / b_pos / value
    LIST_MULTI_SWAP = 10  # a = swaps; a_pos / b / b_pos = (list | first << 16) per swap; value = second - first, one byte per swap
    LIST_PERMUTE = 9  # a = b = list, [a_pos, b_pos) = the window, value = rank of the permutation (lexicographic, >= 1)


class UniLhs:  # sf_uni_lhs
    ONE, ROW_COL, VALUE, VALUE_COL, COL_DIFF, COL_ABSDIFF, TABLE = 0, 1, 2, 3, 4, 5, 6


class UniCmp:  # sf_uni_cmp
    EQ, NE, LT, LE, GT, GE = 0, 1, 2, 3, 4, 5


class SelectionOrder:  # solverforge_config::SelectionOrder
    ORIGINAL, SORTED, PROBABILISTIC, RANDOM, SHUFFLED = 0, 1, 2, 3, 4


class Acceptor:
    HILL_CLIMBING, LATE_ACCEPTANCE = 0, 1
    SIMULATED_ANNEALING = 3  # default of scalar-only models (default_local_search/policy.rs:56-61)
    DIVERSIFIED_LATE_ACCEPTANCE = 4  # default of grouped scalar-only models (default_local_search/policy.rs:52-55)


class AnnealingMode:
    SINGLE, PER_LEVEL, CALIBRATED = 0, 1, 2


class Forager:
    ACCEPTED_COUNT, FIRST_ACCEPTED, BEST_SCORE = 0, 1, 2
    # forager/improving.rs: quit at the first accepted candidate better than the best-ever / last-step score
    FIRST_BEST_SCORE_IMPROVING, FIRST_LAST_STEP_SCORE_IMPROVING = 3, 4


class Engine:  # sf_engine_kind
    AUTO, BLOCK, WAVE = 0, 1, 2


class PairOp:
    """sf_pair_op: terms of a pair-predicate program (add_pair_join)."""
    VALUE_EQ, VALUE_NE, VALUE_ABSDIFF_EQ_COL, COL_EQ, COL_NE, COL_LT, COL_ABSDIFF_EQ, COL_ABSDIFF_LE, CSR_CONTAINS, TABLE_NONZERO, VALUE_ABSDIFF_LE = range(1, 12)


class ConstraintKind:
    UNI_UNASSIGNED, CROSS_ADJACENT_EQUAL, CROSS_GROUP_EQUAL, CROSS_QUEENS = 1, 2, 3, 4
    NOT_EXISTS_FLATTENED, ROUTE_CAPACITY, ROUTE_DISTANCE = 5, 6, 7
    SELFJOIN_VALUE_EQUAL, GROUPED_VALUE_SUM, LOAD_BALANCE_VALUE = 8, 9, 10
    VALUE_COST, EXISTS_VALUE, BALANCE_VALUE = 11, 12, 13
    LIST_PRECEDENCE_MAKESPAN, RUNS_VALUE, COMPLEMENTED_VALUE_SUM, PRESENCE_VALUE = 14, 15, 16, 17
    CROSS_OWNER_MATCH = 18  # join of the two planning classes of a mixed model


class SelectorKind:
    SCALAR_CHANGE, SCALAR_SWAP, LIST_CHANGE, LIST_SWAP = 1, 2, 4, 8
    NEARBY_LIST_CHANGE, NEARBY_LIST_SWAP, LIST_REVERSE, SUBLIST_CHANGE, SUBLIST_SWAP = 16, 32, 64, 128, 256
    KOPT = 512
    LIST_RUIN = 1024
    NEARBY_SCALAR_CHANGE, NEARBY_SCALAR_SWAP = 2048, 4096
    LIST_PERMUTE = 8192
    LIST_PRECEDENCE = 16384


@dataclass
class SolverConfig:
    """Default policy of a list model: LateAcceptance(400) + AcceptedCount(256), every leaf
    SelectionOrder::Random (runtime/compiler/default_local_search/policy.rs:18-20,48-79,114-118)."""

    acceptor: int = Acceptor.LATE_ACCEPTANCE
    late_acceptance_size: int = 400
    forager: int = Forager.ACCEPTED_COUNT
    accepted_count_limit: int = 256
    random_ties: bool = True
    selection_order: int = SelectionOrder.RANDOM
    random_seed: int = 0

    @classmethod
    def default_components(cls, has_lists=False, has_groups=False, has_precedence=False, has_nearby_scalar=False,
                           has_conflict_repairs=False, random_seed=0):
        """compile_default_local_search_components (runtime/compiler/default_local_search/policy.rs:21-82) through the C ABI's
        sf_default_local_search_components: lists -> LateAcceptance(400); grouped scalar-only -> DiversifiedLateAcceptance(400) +
        FirstLastStepScoreImproving without a limit (accepted_count_limit 0); a list slot with precedence moves ->
        FirstLastStepScoreImproving(256); otherwise AcceptedCount(256 with lists / nearby scalar leaves / conflict repairs, else 1)
        and, without lists or groups, SimulatedAnnealing."""
        s = SolverConfigStruct()
        check(_lib.load().sf_default_local_search_components(int(has_lists), int(has_groups), int(has_precedence), int(has_nearby_scalar),
                                                            int(has_conflict_repairs), random_seed, C.byref(s)), None)
        return cls(acceptor=s.acceptor, late_acceptance_size=s.late_acceptance_size, forager=s.forager,
                   accepted_count_limit=s.accepted_count_limit, random_ties=bool(s.random_ties), selection_order=s.selection_order,
                   random_seed=s.random_seed)


class GpuScoreDirector:
    """One device context = `n_replicas` independent Director + search states of one problem."""

    def __init__(self, score_levels=2, hard_levels=1, n_replicas=1, device_id=0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.sf_ctx_create(device_id, score_levels, hard_levels, n_replicas, C.byref(h))
        if rc != 0:
            check(rc, None)
        self._h = h
        self.levels = score_levels
        self.hard_levels = hard_levels
        self.n_replicas = n_replicas
        self._entity_counts = {}
        self._list_capacity = {}
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.sf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- schema (SolutionDescriptor seam) ---------------------------------------------
    def add_entity_class(self, descriptor_index, n_rows):
        check(self._L.sf_schema_add_entity_class(self._h, descriptor_index, n_rows), self._h)
        self._entity_counts[descriptor_index] = n_rows

    def add_scalar_variable(self, descriptor_index, variable_index, n_values, allows_unassigned, initial):
        initial = np.ascontiguousarray(initial, dtype=np.int32)
        n_rows = self._entity_counts.get(descriptor_index)
        if n_rows is None or initial.shape != (n_rows,):  # the C side reads n_rows values
            raise SolverForgeError(f"SF_ERR_INVALID: initial must hold one value per row of class {descriptor_index} "
                                   f"({n_rows}), got shape {initial.shape}")
        check(self._L.sf_schema_add_scalar_variable(self._h, descriptor_index, variable_index, n_values,
                                                    int(allows_unassigned), ptr(initial)), self._h)

    def set_value_lists(self, descriptor_index, variable_index, lists):
        """ValueSource::EntitySlice: `lists[e]` = the canonical value list of entity e (values in 0..n_values)."""
        n_rows = self._entity_counts.get(descriptor_index)
        if n_rows is None or len(lists) != n_rows:  # the C side reads offsets[0..n_rows]
            raise SolverForgeError(f"SF_ERR_INVALID: value lists must hold one list per row of class {descriptor_index} "
                                   f"({n_rows}), got {len(lists)}")
        off = np.zeros(len(lists) + 1, dtype=np.uint32)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        vals = np.array([v for l in lists for v in l] or [0], dtype=np.int32)
        check(self._L.sf_schema_set_value_lists(self._h, descriptor_index, variable_index, ptr(off), ptr(vals)), self._h)

    def add_list_variable(self, descriptor_index, lists, element_capacity, element_id_bound):
        off = np.zeros(len(lists) + 1, dtype=np.uint32)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        vals = np.array([v for l in lists for v in l] or [0], dtype=np.uint32)
        check(self._L.sf_schema_add_list_variable(self._h, descriptor_index, ptr(off), ptr(vals),
                                                  element_capacity, element_id_bound), self._h)
        self._list_capacity[descriptor_index] = element_capacity

    def add_fact_matrix(self, fact_id, matrix):
        matrix = np.ascontiguousarray(matrix, dtype=np.int64)
        check(self._L.sf_fact_matrix_i64(self._h, fact_id, matrix.shape[0], matrix.shape[1], ptr(matrix)), self._h)

    def add_fact_column_i32(self, fact_id, column):
        column = np.ascontiguousarray(column, dtype=np.int32)
        check(self._L.sf_fact_column_i32(self._h, fact_id, len(column), ptr(column)), self._h)

    def add_fact_column_u32(self, fact_id, column):
        column = np.ascontiguousarray(column, dtype=np.uint32)
        check(self._L.sf_fact_column_u32(self._h, fact_id, len(column), ptr(column)), self._h)

    def add_fact_csr(self, fact_id, offsets, values):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        values = np.ascontiguousarray(values if len(values) else [0], dtype=np.uint32)
        check(self._L.sf_fact_csr_u32(self._h, fact_id, len(offsets) - 1, ptr(offsets), ptr(values)), self._h)

    def add_constraint(self, kind, descriptor_index, variable_index=0, fact=-1, param=0, level=0, weight=1):
        check(self._L.sf_constraint_add(self._h, kind, descriptor_index, variable_index, fact, param, level, weight), self._h)
        self._n_constraints = getattr(self, "_n_constraints", 0) + 1

    def add_pair_join(self, descriptor_index, terms, level=0, weight=1, variable_index=0):
        """Predicate join of a scalar class with itself, the predicate as data (sf_constraint_add_pair_join): `terms` = sequence of
        (op, clause, fact, fact_b, param) -- a conjunction of clauses, each a disjunction of its terms (PairOp names the ops)."""
        arr = np.zeros(len(terms), dtype=np.dtype([("op", np.int32), ("clause", np.int32), ("fact", np.int32), ("fact_b", np.int32), ("param", np.int64)]))
        for i, t in enumerate(terms):
            t = tuple(t) + (-1, -1, 0)[len(t) - 2:] if len(t) < 5 else tuple(t)
            arr[i] = t
        check(self._L.sf_constraint_add_pair_join(self._h, descriptor_index, variable_index, arr.ctypes.data_as(C.c_void_p), len(terms), level, weight), self._h)
        self._n_constraints = getattr(self, "_n_constraints", 0) + 1

    def add_uni_program(self, descriptor_index, terms, weight=(0, -1, -1, -1), level=0, scale=1, variable_index=0):
        """for_each(A).filter(pred).penalize(w) with both closures as data (sf_constraint_add_uni_program; compiled on the host into the value-cost
        matrix at initialize): `terms` = sequence of (lhs, cmp, clause, fact, fact_b, fact_c, param) -- UniLhs / UniCmp name the codes; a conjunction
        of clauses, each a disjunction of its terms; `weight` = (lhs, fact, fact_b, fact_c), lhs 0 = the constant 1; the entity costs scale * max(0, w)."""
        tdt = np.dtype([("lhs", np.int32), ("cmp", np.int32), ("clause", np.int32), ("fact", np.int32), ("fact_b", np.int32), ("fact_c", np.int32),
                        ("param", np.int64)])
        arr = np.zeros(max(len(terms), 1), dtype=tdt)
        for i, t in enumerate(terms):
            arr[i] = tuple(t)
        w = np.zeros(1, dtype=np.dtype([("lhs", np.int32), ("fact", np.int32), ("fact_b", np.int32), ("fact_c", np.int32)]))
        w[0] = tuple(weight)
        check(self._L.sf_constraint_add_uni_program(self._h, descriptor_index, variable_index, arr.ctypes.data_as(C.c_void_p), len(terms),
                                                    w.ctypes.data_as(C.c_void_p), level, scale), self._h)
        self._n_constraints = getattr(self, "_n_constraints", 0) + 1

    def add_list_precedence(self, descriptor_index, durations, successors, expected_owner=None, hard_level=0, makespan_level=1,
                            variable_index=0):
        """ListPrecedenceMakespanConstraint (constraint/list_precedence.rs): `successors[node]` = fixed successor ids."""
        dur = np.ascontiguousarray(durations, dtype=np.int32)
        off = np.zeros(len(successors) + 1, dtype=np.uint32)
        for i, l in enumerate(successors):
            off[i + 1] = off[i] + len(l)
        vals = np.array([v for l in successors for v in l] or [0], dtype=np.uint32)
        eo = None if expected_owner is None else np.ascontiguousarray(expected_owner, dtype=np.int32)
        check(self._L.sf_constraint_add_list_precedence(self._h, descriptor_index, variable_index, len(dur), ptr(dur), ptr(off), ptr(vals),
                                                        None if eo is None else ptr(eo), hard_level, makespan_level), self._h)
        self._n_constraints = getattr(self, "_n_constraints", 0) + 1

    def evaluate_each(self, replica=0):
        """ConstraintSet::evaluate_each: (scores [n_constraints, levels], match counts) in declaration order."""
        n = getattr(self, "_n_constraints", 0)
        sc = np.zeros((max(n, 1), self.levels), dtype=np.int64)
        cnt = np.zeros(max(n, 1), dtype=np.int64)
        check(self._L.sf_evaluate_each(self._h, replica, ptr(sc), ptr(cnt)), self._h)
        return sc[:n], cnt[:n]

    def add_selector(self, kind, descriptor_index, variable_index=0, max_nearby=0, fact_meter=-1):
        check(self._L.sf_selector_add(self._h, kind, descriptor_index, variable_index, max_nearby, fact_meter), self._h)

    def configure_union(self, selection_order=-1, weights=None):
        """Root union of the leaves: selection_order 0 Sequential, 1 RoundRobin, 2 RotatingRoundRobin, 3 Random,
        4 StratifiedRandom (-1 = default policy); weights = one unsigned weight per leaf in union order (None = equal)."""
        if weights is None:
            check(self._L.sf_union_configure(self._h, selection_order, None, 0), self._h)
        else:
            w = np.ascontiguousarray(weights, dtype=np.int64)
            check(self._L.sf_union_configure(self._h, selection_order, ptr(w), len(w)), self._h)

    def add_nearby_scalar_selector(self, kind, descriptor_index, rows, distances=None, variable_index=0, max_nearby=10, source_limit=0,
                                   dynamic=False):
        """Nearby scalar change / swap leaf (NearbyChangeMoveConfig / NearbySwapMoveConfig; default max_nearby 10,
        default_local_search/policy/scalar.rs:16).  rows[e] = the slot's nearby source row of entity e in source order (values for the
        change leaf, entity indices for the swap leaf; a slot without the hook passes its ordinary candidate values / range(n));
        distances[e][k] = the slot's distance meter for rows[e][k] (None: no meter, the source order ranks)."""
        n_rows = self._entity_counts.get(descriptor_index)
        if n_rows is None or len(rows) != n_rows:
            raise SolverForgeError(f"SF_ERR_INVALID: nearby source must hold one row per entity of class {descriptor_index} ({n_rows}), got {len(rows)}")
        off = np.zeros(len(rows) + 1, dtype=np.uint32)
        for i, r in enumerate(rows):
            off[i + 1] = off[i] + len(r)
        cand = np.array([v for r in rows for v in r] or [0], dtype=np.int32)
        dist = None
        if distances is not None:
            dist = np.array([v for r in distances for v in r] or [0.0], dtype=np.float64)
            if len(dist) != max(int(off[-1]), 1):
                raise SolverForgeError("SF_ERR_INVALID: distances must parallel the source rows")
        check(self._L.sf_selector_add_nearby_scalar(self._h, kind, descriptor_index, variable_index, max_nearby, source_limit, ptr(off), ptr(cand),
                                                    None if dist is None else ptr(dist), int(dynamic)), self._h)

    def add_ruin_selector(self, descriptor_index, variable_index=0, min_ruin_count=2, max_ruin_count=5, moves_per_step=10,
                          max_source_list_len=0, skip_empty_destinations=False, variable_name="visits"):
        """List ruin leaf (ListRuinMoveSelectorConfig defaults); max_source_list_len 0 = None."""
        check(self._L.sf_selector_add_ruin(self._h, descriptor_index, variable_index, min_ruin_count, max_ruin_count, moves_per_step,
                                           max_source_list_len, int(skip_empty_destinations), variable_name.encode()), self._h)

    def add_precedence_selector(self, descriptor_index, variable_index=0):
        """Critical-path precedence leaf (ListPrecedenceMoveSelector, heuristic/selector/list_precedence.rs:121-210) of a list class that
        carries the precedence constraint: multi-swaps, multi-block ruins, then the tiered move families of every critical block."""
        check(self._L.sf_selector_add_precedence(self._h, descriptor_index, variable_index), self._h)

    def set_precedence_policy(self, descriptor_index, enabled=True, variable_index=0):
        """The runtime list slot declares its precedence hooks to every list leaf (list_leaf/cursor/slot.rs:191-404): intra-list candidates
        that close a cycle through the route graph are dropped, the ruin leaf recreates with the hooks."""
        check(self._L.sf_list_set_precedence_policy(self._h, descriptor_index, variable_index, int(bool(enabled))), self._h)

    def add_permute_selector(self, descriptor_index, variable_index=0, min_window_size=2, max_window_size=5):
        """List permute leaf (ListPermuteMoveSelectorConfig defaults): every non-identity permutation of every window of
        min..=max consecutive elements."""
        check(self._L.sf_selector_add_permute(self._h, descriptor_index, variable_index, min_window_size, max_window_size), self._h)

    def add_kopt_selector(self, descriptor_index, variable_index=0, k=3, min_segment_len=1, max_nearby=20):
        """3-opt leaf (KOptMoveSelectorConfig); max_nearby = 0 enumerates every cut set, > 0 prunes by distance."""
        check(self._L.sf_selector_add_kopt(self._h, descriptor_index, variable_index, k, min_segment_len, max_nearby), self._h)

    def add_sublist_selector(self, kind, descriptor_index, variable_index=0, min_size=1, max_size=3):
        check(self._L.sf_selector_add_sublist(self._h, kind, descriptor_index, variable_index, min_size, max_size), self._h)

    # ---- Director surface ------------------------------------------------------------
    def _scores(self, fn):
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        check(fn(self._h, ptr(out)), self._h)
        return out

    def calculate_score(self):
        """First call ≙ initialize_all; later calls return the cached (committed) scores."""
        if not getattr(self, "_initialized", False):
            self._initialized = True
            return self._scores(self._L.sf_initialize)
        return self._scores(self._L.sf_get_scores)

    def fresh_score(self):
        """≙ Director::fresh_score: full evaluate_all on the device (FullAssert check)."""
        return self._scores(self._L.sf_evaluate_all)

    def entity_count(self, descriptor_index):
        return self._entity_counts.get(descriptor_index)

    def evaluate_moves(self, moves, replica=0):
        """≙ n x evaluate_candidate: one launch, state unchanged -> (scores[n, levels], doable[n])."""
        moves = np.ascontiguousarray(moves, dtype=MOVE_DTYPE)
        scores = np.zeros((len(moves), self.levels), dtype=np.int64)
        doable = np.zeros(len(moves), dtype=np.int32)
        check(self._L.sf_step_evaluate(self._h, replica, ptr(moves), len(moves), ptr(scores), ptr(doable)), self._h)
        return scores, doable

    @staticmethod
    def _compound_wire(candidates):
        offsets = np.zeros(len(candidates) + 1, dtype=np.int64)
        for i, c in enumerate(candidates):
            offsets[i + 1] = offsets[i] + len(c)
        edits = np.zeros(max(int(offsets[-1]), 1), dtype=MOVE_DTYPE)
        k = 0
        for c in candidates:
            for (entity, value) in c:
                edits[k] = (MoveKind.CHANGE, entity, 0, 0, 0, value)
                k += 1
        return edits, offsets

    def evaluate_candidates(self, candidates, replica=0):
        """ScalarCandidateProvider surface: `candidates` = list of lists of (entity_index, to_value) ScalarEdits (to_value -1 =
        None); each list is scored as ONE CompoundScalarMove -> (scores[n, levels], doable[n]).  State unchanged."""
        edits, offsets = self._compound_wire(candidates)
        scores = np.zeros((len(candidates), self.levels), dtype=np.int64)
        doable = np.zeros(len(candidates), dtype=np.int32)
        check(self._L.sf_step_evaluate_compound(self._h, replica, ptr(edits), ptr(offsets), len(candidates), ptr(scores), ptr(doable)), self._h)
        return scores, doable

    def step_decide(self, candidates, replica=0, group_name_len=0, max_moves_per_step=0, gates=None):
        """One host-driven local-search step over a ScalarCandidateProvider's output (GroupedScalarMoveSelector, sf_step_decide):
        returns (kept provider indices in pull order, trial scores [consumed, levels], flags [consumed], selected ordinal or -1).
        gates[i]: bit 0 = the candidate requires a hard improvement, bit 1 = a score improvement (sf_step_decide_gated)."""
        edits, offsets = self._compound_wire(candidates)
        n = len(candidates)
        if gates is not None:
            g = np.ascontiguousarray(gates, dtype=np.int32)
            assert len(g) == n
            kept = np.zeros(max(n, 1), dtype=np.int64)
            scores = np.zeros((max(n, 1), self.levels), dtype=np.int64)
            flags = np.zeros(max(n, 1), dtype=np.int32)
            nk, consumed, selected = C.c_int64(0), C.c_int64(0), C.c_int64(-1)
            check(self._L.sf_step_decide_gated(self._h, replica, ptr(edits), ptr(offsets), ptr(g), n, group_name_len, max_moves_per_step, ptr(kept),
                                               C.byref(nk), ptr(scores), ptr(flags), C.byref(consumed), C.byref(selected)), self._h)
            return kept[:nk.value], scores[:consumed.value], flags[:consumed.value], int(selected.value)
        kept = np.zeros(max(n, 1), dtype=np.int64)
        scores = np.zeros((max(n, 1), self.levels), dtype=np.int64)
        flags = np.zeros(max(n, 1), dtype=np.int32)
        nk, consumed, selected = C.c_int64(0), C.c_int64(0), C.c_int64(-1)
        check(self._L.sf_step_decide(self._h, replica, ptr(edits), ptr(offsets), n, group_name_len, max_moves_per_step, ptr(kept), C.byref(nk),
                                     ptr(scores), ptr(flags), C.byref(consumed), C.byref(selected)), self._h)
        return kept[:nk.value], scores[:consumed.value], flags[:consumed.value], int(selected.value)

    def step_decide_cursor(self, candidates, gates=None, replica=0):
        """One host-driven local-search step over a cursor's own pull order (the reference's RuntimeProviderCursor behind the grouped-scalar
        and conflict-repair leaves; sf_step_decide_cursor): nothing is re-ordered, filtered or capped here.  Returns (trial scores
        [consumed, levels], flags [consumed], committed index or -1)."""
        edits, offsets = self._compound_wire(candidates)
        n = len(candidates)
        g = None if gates is None else np.ascontiguousarray(gates, dtype=np.int32)
        assert g is None or len(g) == n
        scores = np.zeros((max(n, 1), self.levels), dtype=np.int64)
        flags = np.zeros(max(n, 1), dtype=np.int32)
        consumed, selected = C.c_int64(0), C.c_int64(-1)
        check(self._L.sf_step_decide_cursor(self._h, replica, ptr(edits), ptr(offsets), None if g is None else ptr(g), n, ptr(scores), ptr(flags),
                                            C.byref(consumed), C.byref(selected)), self._h)
        return scores[:consumed.value], flags[:consumed.value], int(selected.value)

    def apply_candidate(self, candidate, replica=0):
        """Committed do_move of one multi-edit ScalarCandidate."""
        edits, _ = self._compound_wire([candidate])
        check(self._L.sf_apply_compound(self._h, replica, ptr(edits), len(candidate)), self._h)

    def apply_move(self, move, replica=0):
        mv = np.zeros(1, dtype=MOVE_DTYPE)
        mv[0] = move
        check(self._L.sf_apply(self._h, replica, ptr(mv)), self._h)

    def construct_list_cheapest(self, descriptor_index, elements):
        """≙ ListCheapestInsertionPhase on every replica: places the elements of `elements` (source order) that are in no list
        yet, each at its best (list, position); returns the committed scores [n_replicas, levels]."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        check(self._L.sf_construct_list_cheapest(self._h, descriptor_index, ptr(el), len(el), ptr(out)), self._h)
        return out

    def construct_list_regret(self, descriptor_index, elements, order_keys=None, owners=None):
        """≙ ListRegretInsertionPhase on every replica: every round the unassigned element whose best and second-best insertion
        differ most goes to its best (list, position); order_keys = the construction order key per element (ties go to the
        smaller key, then the earlier element); owners = the owner hook's value per element (-1 unrestricted, a list index =
        only that list, a value >= the list count = never placed).  Returns the committed scores [n_replicas, levels]."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        ks = None if order_keys is None else np.ascontiguousarray(order_keys, dtype=np.int64)
        ow = None if owners is None else np.ascontiguousarray(owners, dtype=np.int32)
        if (ks is not None and len(ks) != len(el)) or (ow is not None and len(ow) != len(el)):
            raise SolverForgeError("order_keys / owners and elements differ in length")
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        check(self._L.sf_construct_list_regret(self._h, descriptor_index, ptr(el), len(el), None if ks is None else ptr(ks),
                                               None if ow is None else ptr(ow), ptr(out)), self._h)
        return out

    def construct_list_k_opt(self, descriptor_index, k=2, feasible_mode=1, max_sweeps=1000):
        """≙ ListKOptPhase on every replica: every route swept to its 2-opt local optimum (k = 2; other k: scored no-op), at
        most max_sweeps sweeps per route; feasible_mode 0 = no feasibility hook, 1 = capacity.  Returns the committed scores
        [n_replicas, levels]."""
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        check(self._L.sf_construct_list_k_opt(self._h, descriptor_index, int(k), int(feasible_mode), int(max_sweeps), ptr(out)), self._h)
        return out

    def construct_list_round_robin(self, descriptor_index, elements, order_keys=None, owners=None):
        """≙ ListConstructionPhase (round robin) on every replica: the elements of `elements` (source order) that are in no list
        yet, in (order key, source index) order, appended to the cursor's owner (owners[k] = -1) or to their fixed owner;
        returns the committed scores [n_replicas, levels]."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        ks = None if order_keys is None else np.ascontiguousarray(order_keys, dtype=np.int64)
        ow = None if owners is None else np.ascontiguousarray(owners, dtype=np.int32)
        if (ks is not None and len(ks) != len(el)) or (ow is not None and len(ow) != len(el)):
            raise ValueError("order_keys / owners: one value per element")
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        check(self._L.sf_construct_list_round_robin(self._h, descriptor_index, ptr(el), len(el), None if ks is None else ptr(ks),
                                                    None if ow is None else ptr(ow), ptr(out)), self._h)
        return out

    def construct_list_clarke_wright(self, descriptor_index, elements, feasible_mode=0):
        """≙ ListClarkeWrightPhase (stock CVRP hooks) on every replica: savings routes over the elements that are in no list yet,
        assigned to the empty owners; feasible_mode 0 = structural (savings_hooks), 1 = capacity (route_hooks).  Returns
        (committed scores [n_replicas, levels], committed flags [n_replicas])."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        out = np.zeros((self.n_replicas, self.levels), dtype=np.int64)
        flags = np.zeros(self.n_replicas, dtype=np.int32)
        check(self._L.sf_construct_list_clarke_wright(self._h, descriptor_index, ptr(el), len(el), int(feasible_mode), ptr(out), ptr(flags)), self._h)
        return out, flags

    # ---- MoveSelector / cursor surface ---------------------------------------------------
    def open_cursor(self, step_index, step_seed, selection_order=SelectionOrder.RANDOM, replica=0, cap=1 << 16):
        """Drains the configured union cursor for MoveStreamContext(step_index, step_seed):
        returns (moves, trial scores, doable) in cursor order; state unchanged."""
        moves = np.zeros(cap, dtype=MOVE_DTYPE)
        scores = np.zeros((cap, self.levels), dtype=np.int64)
        doable = np.zeros(cap, dtype=np.int32)
        n = C.c_int64(0)
        check(self._L.sf_step_generate(self._h, replica, step_index, step_seed, selection_order, ptr(moves),
                                       ptr(scores), ptr(doable), cap, C.byref(n)), self._h)
        k = n.value
        return moves[:k], scores[:k], doable[:k]

    # ---- local search phase ----------------------------------------------------------------
    def configure(self, cfg: SolverConfig):
        s = SolverConfigStruct(cfg.acceptor, cfg.late_acceptance_size, cfg.forager, cfg.accepted_count_limit,
                               int(cfg.random_ties), cfg.selection_order, cfg.random_seed)
        check(self._L.sf_solver_configure(self._h, C.byref(s)), self._h)

    def declare_provider(self, kind, name):
        """A host-side provider of the model (sf_provider_declare): kind 1 = scalar group, 2 = conflict repair; configure_default derives
        the default policy's has_groups / has_conflict_repairs from these when they are not passed."""
        check(self._L.sf_provider_declare(self._h, int(kind), name.encode()), self._h)

    def configure_default(self, random_seed=0, has_groups=None, has_conflict_repairs=None):
        """The reference's default acceptor + forager for THIS model (sf_solver_configure_default: lists / precedence hooks / nearby
        scalar leaves are read from the context); returns the SolverConfig that was set."""
        s = SolverConfigStruct()
        check(self._L.sf_solver_configure_default(self._h, random_seed, -1 if has_groups is None else int(has_groups),
                                                   -1 if has_conflict_repairs is None else int(has_conflict_repairs), C.byref(s)), self._h)
        return SolverConfig(acceptor=s.acceptor, late_acceptance_size=s.late_acceptance_size, forager=s.forager,
                            accepted_count_limit=s.accepted_count_limit, random_ties=bool(s.random_ties),
                            selection_order=s.selection_order, random_seed=s.random_seed)

    def configure_annealing(self, mode=AnnealingMode.CALIBRATED, temperatures=(), decay_rate=0.999985,
                            hill_climbing_temperature=1.0e-9, never_accept_hard_regression=False,
                            calibration_sample_size=128, target_acceptance_probability=0.80,
                            fallback_temperature=1.0, seed=0):
        """SimulatedAnnealingConfig (builder/acceptor.rs:270-335); replica r draws from SmallRng(seed + r)."""
        t = (C.c_double * 4)(*([float(x) for x in temperatures] + [0.0] * (4 - len(temperatures))))
        s = AnnealingConfigStruct(mode, int(never_accept_hard_regression), calibration_sample_size, 0, t, decay_rate,
                                  hill_climbing_temperature, target_acceptance_probability, fallback_temperature, seed)
        check(self._L.sf_solver_configure_annealing(self._h, C.byref(s)), self._h)

    def configure_diversified(self, tolerance=0.01):
        """Tolerance of AcceptorKind.DIVERSIFIED_LATE_ACCEPTANCE (DiversifiedLateAcceptanceAcceptor::new); the history size
        is SolverConfig.late_acceptance_size."""
        check(self._L.sf_solver_configure_diversified(self._h, float(tolerance)), self._h)

    def annealing_state(self, replica=0):
        """(current temperatures per score level, still calibrating?) of one replica's acceptor."""
        t = np.zeros(4, dtype=np.float64)
        c = C.c_int32(0)
        check(self._L.sf_get_annealing_state(self._h, replica, ptr(t), C.byref(c)), self._h)
        return t[:self.levels].copy(), bool(c.value)

    def set_engine(self, engine):
        """Pick the fused-kernel mapping (Engine.AUTO / BLOCK / WAVE); results are identical."""
        check(self._L.sf_solver_set_engine(self._h, engine), self._h)

    def set_step_seeds(self, seeds):
        if seeds is None:
            check(self._L.sf_solver_set_step_seeds(self._h, None, 0), self._h)
            return
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(self.n_replicas, -1)
        check(self._L.sf_solver_set_step_seeds(self._h, ptr(seeds), seeds.shape[1]), self._h)

    def phase_start(self):
        check(self._L.sf_phase_start(self._h), self._h)

    def solve_steps(self, n_steps, sync=True):
        check(self._L.sf_solve_steps(self._h, n_steps), self._h)
        if sync:
            check(self._L.sf_sync(self._h), self._h)

    def solve_moves(self, max_steps, move_budget, sync=True):
        """Work-balanced launch: every replica runs whole steps until it has pulled `move_budget` candidates in this
        launch (or `max_steps` steps).  `stats(r)["step_count"]` says how far replica r got."""
        check(self._L.sf_solve_moves(self._h, max_steps, move_budget), self._h)
        if sync:
            check(self._L.sf_sync(self._h), self._h)

    def sync(self):
        check(self._L.sf_sync(self._h), self._h)

    def solve_step_traced(self, replica=0, cap=1 << 16):
        moves = np.zeros(cap, dtype=MOVE_DTYPE)
        scores = np.zeros((cap, self.levels), dtype=np.int64)
        flags = np.zeros(cap, dtype=np.int32)
        n = C.c_int64(0)
        applied = C.c_int32(0)
        applied_move = np.zeros(1, dtype=MOVE_DTYPE)
        check(self._L.sf_solve_step_traced(self._h, replica, ptr(moves), ptr(scores), ptr(flags), cap,
                                           C.byref(n), C.byref(applied), ptr(applied_move)), self._h)
        k = n.value
        return moves[:k], scores[:k], flags[:k], bool(applied.value), applied_move[0]

    def stats(self, replica=0):
        st = StatsStruct()
        check(self._L.sf_get_stats(self._h, replica, C.byref(st)), self._h)
        return {name: int(getattr(st, name)) for name, _ in StatsStruct._fields_}

    def total_stats(self):
        st = StatsStruct()
        check(self._L.sf_get_stats_sum(self._h, C.byref(st)), self._h)
        return {name: int(getattr(st, name)) for name, _ in StatsStruct._fields_}

    def engine(self):
        """The engine launches resolve to: Engine.BLOCK or Engine.WAVE."""
        e = C.c_int32(0)
        check(self._L.sf_solver_get_engine(self._h, C.byref(e)), self._h)
        return e.value

    def wave_layout(self):
        """(launch mode, renumbered) of the wave engine's last fused launch (sf_list_wave_layout): diagnostics for tests."""
        m, n = C.c_int32(0), C.c_int32(0)
        check(self._L.sf_list_wave_layout(self._h, C.byref(m), C.byref(n)), self._h)
        return m.value, bool(n.value)

    def best_scores(self):
        return self._scores(self._L.sf_get_best_scores)

    def profile_solve(self):
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(self._L.sf_profile_solve(self._h, C.byref(ms), C.byref(n)), self._h)
        return ms.value, n.value

    # ---- state download --------------------------------------------------------------------
    def working_lists(self, descriptor_index=0, replica=0, best=False):
        n = self._entity_counts[descriptor_index]
        off = np.zeros(n + 1, dtype=np.uint32)
        vals = np.zeros(max(self._list_capacity[descriptor_index], 1), dtype=np.uint32)
        check(self._L.sf_download_list(self._h, replica, descriptor_index, ptr(off), ptr(vals), int(best)), self._h)
        return [list(map(int, vals[off[i]: off[i + 1]])) for i in range(n)]

    def working_values(self, descriptor_index=0, variable_index=0, replica=0, best=False):
        n = self._entity_counts[descriptor_index]
        out = np.zeros(n, dtype=np.int32)
        check(self._L.sf_download_scalar(self._h, replica, descriptor_index, variable_index, ptr(out), int(best)), self._h)
        return out

    # ---- portfolio -------------------------------------------------------------------------
    def portfolio_unique_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        check(self._L.sf_portfolio_unique_id(ptr(buf)), self._h)
        return buf

    def portfolio_init(self, unique_id, rank, world_size):
        unique_id = np.ascontiguousarray(unique_id, dtype=np.uint8)
        check(self._L.sf_portfolio_init(self._h, ptr(unique_id), rank, world_size), self._h)

    def portfolio_allgather_best(self):
        best = np.zeros(self.levels, dtype=np.int64)
        rank = C.c_int32(0)
        rep = C.c_int32(0)
        check(self._L.sf_portfolio_allgather_best(self._h, ptr(best), C.byref(rank), C.byref(rep)), self._h)
        return best, rank.value, rep.value

    def portfolio_broadcast_best(self, winner_rank, winner_replica, descriptor_index=0):
        """The winner's best lists on every rank (ncclBroadcast of the route CSR from the winning rank)."""
        n = self._entity_counts[descriptor_index]
        off = np.zeros(n + 1, dtype=np.uint32)
        vals = np.zeros(max(self._list_capacity[descriptor_index], 1), dtype=np.uint32)
        check(self._L.sf_portfolio_broadcast_best(self._h, winner_rank, winner_replica, ptr(off), ptr(vals)), self._h)
        return [list(map(int, vals[off[i]: off[i + 1]])) for i in range(n)]

    def migrate_local(self, n_elite, n_replace):
        """EXTENSION (sf_portfolio_migrate_local): the n_replace replicas with the worst best score adopt the best solution of the
        n_elite best ones (LateAcceptance history restarted at the adopted score); returns how many replicas took a copy."""
        n = C.c_int32(0)
        check(self._L.sf_portfolio_migrate_local(self._h, int(n_elite), int(n_replace), C.byref(n)), self._h)
        return n.value

    def portfolio_destroy(self):
        check(self._L.sf_portfolio_destroy(self._h), self._h)
