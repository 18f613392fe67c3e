"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/libsf_oracle.so (the C++ CPU restatement of the
reference algorithm).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; solverforge_amd/ never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "_build", "libsf_oracle.so")

MOVE_DTYPE = np.dtype(
    [("kind", "<i4"), ("a", "<i4"), ("a_pos", "<i4"), ("b", "<i4"), ("b_pos", "<i4"), ("value", "<i4")]
)
KIND_CHANGE, KIND_SWAP, KIND_LIST_CHANGE, KIND_LIST_SWAP, KIND_LIST_REVERSE, KIND_SUBLIST_CHANGE, KIND_SUBLIST_SWAP = 0, 1, 2, 3, 4, 5, 6
ORDER_ORIGINAL, ORDER_SORTED, ORDER_PROBABILISTIC, ORDER_RANDOM, ORDER_SHUFFLED = 0, 1, 2, 3, 4
LEAF_SCALAR_CHANGE, LEAF_SCALAR_SWAP, LEAF_LIST_CHANGE, LEAF_LIST_SWAP = 1, 2, 4, 8
LEAF_NEARBY_LIST_CHANGE, LEAF_NEARBY_LIST_SWAP = 16, 32
LEAF_LIST_REVERSE = 64
LEAF_SUBLIST_CHANGE = 128
LEAF_SUBLIST_SWAP = 256
LEAF_KOPT = 512
LEAF_LIST_RUIN = 1024
LEAF_NEARBY_SCALAR_CHANGE, LEAF_NEARBY_SCALAR_SWAP = 2048, 4096
LEAF_LIST_PERMUTE = 8192
LEAF_LIST_PRECEDENCE = 16384
KIND_KOPT, KIND_RUIN = 7, 8
ACCEPT_HILL_CLIMBING, ACCEPT_LATE_ACCEPTANCE = 0, 1
FORAGER_ACCEPTED_COUNT, FORAGER_FIRST_ACCEPTED, FORAGER_BEST_SCORE = 0, 1, 2
FORAGER_FIRST_BEST_SCORE_IMPROVING, FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING = 3, 4
UNION_SEQUENTIAL, UNION_ROUND_ROBIN, UNION_ROTATING, UNION_RANDOM, UNION_STRATIFIED = 0, 1, 2, 3, 4


def build(force=False):
    """Compile the oracle (g++).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(_DIR, f)) > os.path.getmtime(_LIB)
        for f in os.listdir(_DIR)
        if f.endswith((".hpp", ".cpp"))
    ):
        subprocess.check_call(["make", "-C", _DIR, "-j4"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u64, i64, i32, u32, vp = C.c_uint64, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p
        dbl = C.c_double
        sig = {
            "sfo_splitmix64": (u64, [u64]),
            "sfo_step_seed": (u64, [u64, u64]),
            "sfo_ctx_mixed_seed": (u64, [u64, u64, u64]),
            "sfo_ctx_random_index": (u64, [u64, u64, u64, u64]),
            "sfo_ctx_random_stride": (u64, [u64, u64, u64, u64]),
            "sfo_ctx_selection_index": (u64, [u64, u64, i32, u64, u64, u64]),
            "sfo_ctx_selection_index_wo": (u64, [u64, u64, i32, u64, u64, u64]),
            "sfo_reservoir_pick": (i32, [u64, u64]),
            "sfo_sort_and_limit": (i32, [vp, i32, i32, vp]),
            "sfo_nqueens_create": (vp, [i32, vp]),
            "sfo_graph_coloring_create": (vp, [i32, i32, vp, vp, vp]),
            "sfo_balance_create": (vp, [i32, i32, vp, vp, i64, i64]),
            "sfo_graph_coloring_create_indexed": (vp, [i32, i32, vp, vp, vp]),
            "sfo_jobshop_create_indexed": (vp, [i32, i32, vp, vp, vp, vp, i32]),
            "sfo_balance_create_nary": (vp, [i32, i32, vp, vp, i64, i64, i32]),
            "sfo_balance_create_base": (vp, [i32, i32, vp, vp, i64, i64]),
            "sfo_cvrp_create": (vp, [i32, i32, i64, i32, i32, vp, vp, vp, vp, vp]),
            "sfo_assignment_create": (vp, [i32, i32, vp, vp, i64, vp, i32, i32, i64]),
            "sfo_assignment_create2": (vp, [i32, i32, vp, vp, i64, vp, i32, i32, i64, vp, i32]),
            "sfo_list_toy_create": (vp, [i32, vp, vp, i32]),
            "sfo_shift_schedule_create": (vp, [i32, i32, vp, vp, i64, i64, i64, i64, vp]),
            "sfo_shift_schedule_create_presence": (vp, [i32, i32, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp]),
            "sfo_precedence_shop_create": (vp, [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
            "sfo_jobshop_create": (vp, [i32, i32, vp, vp, vp, vp, i32]),
            "sfo_jobshop_create_owner_match": (vp, [i32, i32, vp, vp, vp, vp, i32, i32]),
            "sfo_jobshop_create_makespan": (vp, [i32, i32, vp, vp, vp, vp, i32, i32, vp]),
            "sfo_model_destroy": (None, [vp]),
            "sfo_model_score": (None, [vp, vp]),
            "sfo_model_fresh_score": (None, [vp, vp]),
            "sfo_model_reset": (None, [vp]),
            "sfo_model_configure": (None, [vp, i32, i32, i32, i32, i32, i32, u32, i32, i32, u64, i32]),
            "sfo_model_set_kopt": (None, [vp, i32, i32]),
            "sfo_model_set_permute": (None, [vp, i32, i32]),
            "sfo_model_set_precedence_policy": (None, [vp, i32]),
            "sfo_model_evaluate_each": (i32, [vp, vp, vp, i32]),
            "sfo_model_configure_annealing": (None, [vp, i32, vp, i32, i32, dbl, dbl, i32, i32, dbl, dbl, u64]),
            "sfo_model_configure_diversified": (None, [vp, i32, dbl]),
            "sfo_model_set_nearby_scalar": (None, [vp, i32, i32, vp, vp, vp, i32, i32, i64]),
            "sfo_model_annealing_state": (None, [vp, vp, vp, vp]),
            "sfo_xoshiro256pp": (None, [vp, i32, vp]),
            "sfo_small_rng_seed": (None, [u64, vp]),
            "sfo_model_set_step_seeds": (None, [vp, vp, i32]),
            "sfo_model_set_union_weights": (None, [vp, vp, i32]),
            "sfo_model_set_value_lists": (None, [vp, vp, vp, i32]),
            "sfo_model_set_ruin": (None, [vp, i32, i32, i32, i32, i32, C.c_char_p]),
            "sfo_scoped_seed": (u64, [u64, u64, C.c_char_p, C.c_char_p]),
            "sfo_hash_str": (u64, [C.c_char_p]),
            "sfo_siphash": (u64, [i32, i32, u64, u64, vp, i64]),
            "sfo_random_range_stream": (None, [u64, u64, u64, i32, vp]),
            "sfo_model_set_sublist_sizes": (None, [vp, i32, i32]),
            "sfo_model_phase_start": (None, [vp]),
            "sfo_model_steps": (None, [vp, i64]),
            "sfo_model_steps_timed": (i64, [vp, dbl]),
            "sfo_model_step_traced": (i32, [vp, vp, vp, vp, i32, vp, vp]),
            "sfo_model_stats": (None, [vp, vp]),
            "sfo_model_last_step_score": (None, [vp, vp]),
            "sfo_model_best_score": (None, [vp, vp]),
            "sfo_model_enumerate": (i64, [vp, u32, u64, u64, i32, vp, i64]),
            "sfo_model_enumerate_count": (i64, [vp, u32, u64, u64, i32]),
            "sfo_model_evaluate_moves": (None, [vp, vp, i64, vp, vp]),
            "sfo_model_apply_move": (None, [vp, vp]),
            "sfo_model_evaluate_compound": (None, [vp, vp, vp, i64, vp, vp]),
            "sfo_model_apply_compound": (None, [vp, vp, i64]),
            "sfo_model_step_grouped": (i64, [vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp]),
            "sfo_model_step_grouped_gated": (i64, [vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp]),
            "sfo_model_construct_first_fit": (None, [vp]),
            "sfo_model_construct_list_cheapest": (None, [vp, vp, i32]),
            "sfo_model_construct_list_clarke_wright": (i32, [vp, vp, i32, i32, vp]),
            "sfo_model_construct_list_round_robin": (None, [vp, vp, i32, vp, vp]),
            "sfo_model_construct_list_k_opt": (None, [vp, i32, i32, i32, vp]),
            "sfo_model_set_time_windows": (None, [vp, vp, vp, vp, vp, i64]),
            "sfo_model_route_feasible": (i32, [vp, vp, i32]),
            "sfo_model_get_vars": (i32, [vp, i32, i32, vp]),
            "sfo_model_get_lists": (i32, [vp, i32, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.uint32)
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l)
    vals = np.array([v for l in lists for v in l], dtype=np.uint32)
    if vals.size == 0:
        vals = np.zeros(1, dtype=np.uint32)
    return off, vals


class Model:
    """Handle on one oracle model (ScoreDirector + slots + LocalSearch)."""

    def __init__(self, handle, n_entities_by_desc, keep=()):
        self.h = C.c_void_p(handle)
        self.n_by_desc = n_entities_by_desc
        self._keep = keep

    def __del__(self):
        try:
            lib().sfo_model_destroy(self.h)
        except Exception:
            pass

    # -- constructors -----------------------------------------------------------------
    @staticmethod
    def nqueens(rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        return Model(lib().sfo_nqueens_create(len(rows), _p(rows)), [len(rows)])

    @staticmethod
    def graph_coloring(n_colors, adj_off, adj, colors, indexed=False):
        """indexed=True: the INDEXED CPU baseline (partner-indexed join) instead of the reference's dense predicate join."""
        adj_off = np.ascontiguousarray(adj_off, dtype=np.uint32)
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        colors = np.ascontiguousarray(colors, dtype=np.int64)
        n = len(colors)
        fn = lib().sfo_graph_coloring_create_indexed if indexed else lib().sfo_graph_coloring_create
        return Model(fn(n, n_colors, _p(adj_off), _p(adj), _p(colors)), [n])

    @staticmethod
    def balance(n_bins, bins, sizes, w_pair=1, cap=-1, arity=2, balance_base=1000):
        bins = np.ascontiguousarray(bins, dtype=np.int64)
        sizes = np.ascontiguousarray(sizes, dtype=np.int64)
        if cap == -3 and balance_base != 1000:
            return Model(lib().sfo_balance_create_base(len(bins), n_bins, _p(bins), _p(sizes), w_pair, balance_base), [len(bins)])
        if arity != 2:
            return Model(lib().sfo_balance_create_nary(len(bins), n_bins, _p(bins), _p(sizes), w_pair, cap, arity), [len(bins)])
        return Model(lib().sfo_balance_create(len(bins), n_bins, _p(bins), _p(sizes), w_pair, cap), [len(bins)])

    @staticmethod
    def assignment(values, cost, n_values, cost_weight=1, row_w=None, ex_mode=1, ex_level=1, ex_weight=1, cost2=None, cost2_level=-1):
        """Keyed cross-join with a fact class (pair cost matrix) + exists / not-exists per fact row; ex_level < 0 = no exists node.
        cost2 / cost2_level: the same keyed join once more on another score level (weight 1)."""
        values = np.ascontiguousarray(values, dtype=np.int64)
        cost = np.ascontiguousarray(cost, dtype=np.int64)
        row_w = np.ones(n_values, dtype=np.int64) if row_w is None else np.ascontiguousarray(row_w, dtype=np.int64)
        if cost2 is not None and cost2_level >= 0:
            cost2 = np.ascontiguousarray(cost2, dtype=np.int64)
            h = lib().sfo_assignment_create2(len(values), n_values, _p(values), _p(cost), cost_weight, _p(row_w), ex_mode, ex_level, ex_weight, _p(cost2), cost2_level)
        else:
            h = lib().sfo_assignment_create(len(values), n_values, _p(values), _p(cost), cost_weight, _p(row_w), ex_mode, ex_level, ex_weight)
        return Model(h, [len(values)])

    @staticmethod
    def cvrp(capacity, depot, demands, matrix, customers, routes):
        demands = np.ascontiguousarray(demands, dtype=np.int32)
        matrix = np.ascontiguousarray(matrix, dtype=np.int64)
        customers = np.ascontiguousarray(customers, dtype=np.uint32)
        off, vals = csr(routes)
        dim = matrix.shape[0]
        h = lib().sfo_cvrp_create(
            len(customers), len(routes), int(capacity), int(depot), dim, _p(demands), _p(matrix), _p(customers),
            _p(off), _p(vals),
        )
        return Model(h, [len(routes)])

    @staticmethod
    def list_toy(routes, meter="equal"):
        off, vals = csr(routes)
        h = lib().sfo_list_toy_create(len(routes), _p(off), _p(vals), 0 if meter == "equal" else 1)
        return Model(h, [len(routes)])

    @staticmethod
    def shift_schedule(nurse_idx, day, n_nurses, limit=2, w_streak=1, count_weight=0, target=-1, required=None, presence=None):
        """examples/minimal-shift-scheduling: unassigned, one shift per nurse-day, long work streaks (consecutive_runs), workload.
        presence = (lo, hi, cap): the streak constraint becomes group_by(nurse, indexed_presence(day)) scored
        w_streak * min(count_in(lo..hi), cap) (cap 0 = uncapped, 1 = any_in)."""
        nurse_idx = np.ascontiguousarray(nurse_idx, dtype=np.int64)
        day = np.ascontiguousarray(day, dtype=np.int64)
        req = None if required is None else np.ascontiguousarray(required, dtype=np.int64)
        if presence is not None:
            lo, hi, cap = presence[:3]
            mode = presence[3] if len(presence) > 3 else 0  # 1: excess of complement_runs(lo..hi) over cap
            h = lib().sfo_shift_schedule_create_presence(len(nurse_idx), n_nurses, _p(nurse_idx), _p(day), lo, hi, cap, mode, w_streak,
                                                         count_weight, target, None if req is None else _p(req))
            return Model(h, [len(nurse_idx)])
        h = lib().sfo_shift_schedule_create(len(nurse_idx), n_nurses, _p(nurse_idx), _p(day), limit, w_streak, count_weight, target,
                                            None if req is None else _p(req))
        return Model(h, [len(nurse_idx)])

    @staticmethod
    def precedence_shop(duration, successors, lists, expected_owner=None, levels=2, hard_levels=1, hard_level=0, soft_level=1):
        """ListPrecedenceMakespanConstraint on a list-only model: `successors` = fixed successor lists per node."""
        duration = np.ascontiguousarray(duration, dtype=np.int64)
        soff, svals = csr(successors)
        off, vals = csr(lists)
        eo = None if expected_owner is None else np.ascontiguousarray(expected_owner, dtype=np.int64)
        h = lib().sfo_precedence_shop_create(len(duration), len(lists), _p(duration), _p(soff), _p(svals),
                                             None if eo is None else _p(eo), _p(off), _p(vals), levels, hard_levels, hard_level, soft_level)
        return Model(h, [len(lists)])

    @staticmethod
    def jobshop(job, machine_idx, sequences, bendable=True, indexed=False, durations=None, owner_match_level=None):
        """durations: adds the ListPrecedenceMakespanConstraint (job order + machine sequences) -- the makespan objective."""
        job = np.ascontiguousarray(job, dtype=np.int64)
        machine_idx = np.ascontiguousarray(machine_idx, dtype=np.int64)
        off, vals = csr(sequences)
        if owner_match_level is not None:  # + the join of the two planning classes (an operation on a machine that does not schedule it)
            h = lib().sfo_jobshop_create_owner_match(len(job), len(sequences), _p(job), _p(machine_idx), _p(off), _p(vals), int(bendable), int(owner_match_level))
            return Model(h, [len(job), len(sequences)])
        if durations is not None:
            dur = np.ascontiguousarray(durations, dtype=np.int64)
            h = lib().sfo_jobshop_create_makespan(len(job), len(sequences), _p(job), _p(machine_idx), _p(off), _p(vals), int(bendable),
                                                  int(indexed), _p(dur))
            return Model(h, [len(job), len(sequences)])
        fn = lib().sfo_jobshop_create_indexed if indexed else lib().sfo_jobshop_create
        h = fn(len(job), len(sequences), _p(job), _p(machine_idx), _p(off), _p(vals), int(bendable))
        return Model(h, [len(job), len(sequences)])

    # -- director ---------------------------------------------------------------------
    def score(self):
        out = np.zeros(4, dtype=np.int64)
        lib().sfo_model_score(self.h, _p(out))
        return out

    def fresh_score(self):
        out = np.zeros(4, dtype=np.int64)
        lib().sfo_model_fresh_score(self.h, _p(out))
        return out

    def reset(self):
        lib().sfo_model_reset(self.h)

    def evaluate_each(self, cap=16):
        """ConstraintSet::evaluate_each: (scores [n, 4], match counts [n]) in declaration order."""
        sc = np.zeros((cap, 4), dtype=np.int64)
        cnt = np.zeros(cap, dtype=np.int64)
        n = lib().sfo_model_evaluate_each(self.h, _p(sc), _p(cnt), cap)
        return sc[:n], cnt[:n]

    # -- search -----------------------------------------------------------------------
    def configure(self, acceptor=ACCEPT_LATE_ACCEPTANCE, la_size=400, forager=FORAGER_ACCEPTED_COUNT, limit=256,
                  random_ties=True, selection_order=ORDER_RANDOM, leaves=0, max_nearby=20, union_order=-1,
                  random_seed=0, public_entity_order=False):
        lib().sfo_model_configure(self.h, acceptor, la_size, forager, limit, int(random_ties), selection_order,
                                  leaves, max_nearby, union_order, random_seed, int(public_entity_order))

    def configure_annealing(self, mode=2, temperatures=(), levels=2, hard_levels=1, decay_rate=0.999985,
                            hill_climbing_temperature=1.0e-9, never_accept_hard=False, sample_size=128,
                            target_probability=0.80, fallback_temperature=1.0, seed=0):
        """Install a SimulatedAnnealingAcceptor (call after configure()). mode: 0 single, 1 per level, 2 calibrated."""
        t = np.zeros(4, dtype=np.float64)
        t[:len(temperatures)] = temperatures
        lib().sfo_model_configure_annealing(self.h, mode, _p(t), levels, hard_levels, decay_rate,
                                            hill_climbing_temperature, int(never_accept_hard), sample_size,
                                            target_probability, fallback_temperature, seed)

    def set_nearby_scalar(self, which, rows, distances=None, dynamic=False, max_nearby=10, source_limit=0):
        """Nearby sources of the scalar slot: which = 0 value candidates per entity, 1 entity candidates per left entity; rows in
        source order; distances = the meter's value per row entry (None: the source order ranks)."""
        off = np.zeros(len(rows) + 1, dtype=np.uint32)
        for i, r in enumerate(rows):
            off[i + 1] = off[i] + len(r)
        cand = np.array([v for r in rows for v in r] or [0], dtype=np.int64)
        dist = None if distances is None else np.array([v for r in distances for v in r] or [0.0], dtype=np.float64)
        lib().sfo_model_set_nearby_scalar(self.h, which, len(rows), _p(off), _p(cand), None if dist is None else _p(dist), int(dynamic),
                                          max_nearby, source_limit)

    def configure_diversified(self, la_size=400, tolerance=0.01):
        """Install a DiversifiedLateAcceptanceAcceptor (call after configure())."""
        lib().sfo_model_configure_diversified(self.h, la_size, tolerance)

    def annealing_state(self):
        t = np.zeros(4, dtype=np.float64)
        r = np.zeros(4, dtype=np.uint64)
        c = np.zeros(1, dtype=np.int32)
        lib().sfo_model_annealing_state(self.h, _p(t), _p(r), _p(c))
        return t, r, int(c[0])

    def set_sublist_sizes(self, min_size, max_size):
        lib().sfo_model_set_sublist_sizes(self.h, min_size, max_size)

    def set_precedence_policy(self, on=True):
        """The list slot declares its precedence hooks to every runtime list leaf (list_leaf/cursor/slot.rs:191-404): intra-list candidates
        that close a cycle through the route graph are dropped, ruins recreate with the hooks."""
        lib().sfo_model_set_precedence_policy(self.h, int(bool(on)))

    def set_permute(self, min_window_size=2, max_window_size=5):
        lib().sfo_model_set_permute(self.h, min_window_size, max_window_size)

    def set_kopt(self, min_segment_len=1, max_nearby=20):
        """3-opt leaf parameters; max_nearby = 0 selects the full-enumeration cursor."""
        lib().sfo_model_set_kopt(self.h, min_segment_len, max_nearby)

    def set_ruin(self, min_count=2, max_count=5, moves_per_step=10, max_source_list_len=0, skip_empty_destinations=False,
                 variable_name="visits"):
        """List ruin leaf parameters (ListRuinMoveSelectorConfig defaults); call after configure(): re-seeds the leaf's stream."""
        lib().sfo_model_set_ruin(self.h, min_count, max_count, moves_per_step, max_source_list_len, int(skip_empty_destinations),
                                 variable_name.encode())

    def set_value_lists(self, lists):
        """ValueSource::EntitySlice for the scalar slot."""
        off = np.zeros(len(lists) + 1, dtype=np.uint32)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        vals = np.array([v for l in lists for v in l] or [0], dtype=np.int64)
        lib().sfo_model_set_value_lists(self.h, _p(off), _p(vals), len(lists))

    def set_union_weights(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.uint64)
        lib().sfo_model_set_union_weights(self.h, _p(w), len(w))

    def set_step_seeds(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        lib().sfo_model_set_step_seeds(self.h, _p(seeds), len(seeds))

    def phase_start(self):
        lib().sfo_model_phase_start(self.h)

    def steps(self, n):
        lib().sfo_model_steps(self.h, n)

    def steps_timed(self, seconds):
        return lib().sfo_model_steps_timed(self.h, float(seconds))

    def step_traced(self, cap=1 << 20):
        moves = np.zeros(cap, dtype=MOVE_DTYPE)
        scores = np.zeros((cap, 4), dtype=np.int64)
        flags = np.zeros(cap, dtype=np.int32)
        applied = C.c_int32(0)
        applied_move = np.zeros(1, dtype=MOVE_DTYPE)
        n = lib().sfo_model_step_traced(self.h, _p(moves), _p(scores), _p(flags), cap, C.byref(applied), _p(applied_move))
        n = min(n, cap)
        return moves[:n], scores[:n], flags[:n], bool(applied.value), applied_move[0]

    def stats(self):
        out = np.zeros(8, dtype=np.uint64)
        lib().sfo_model_stats(self.h, _p(out))
        keys = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
                "score_calculations", "moves_not_doable"]
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def last_step_score(self):
        out = np.zeros(4, dtype=np.int64)
        lib().sfo_model_last_step_score(self.h, _p(out))
        return out

    def best_score(self):
        out = np.zeros(4, dtype=np.int64)
        lib().sfo_model_best_score(self.h, _p(out))
        return out

    def enumerate(self, leaf, step_index=0, step_seed=0, order=ORDER_ORIGINAL, cap=None):
        if cap is None:
            cap = lib().sfo_model_enumerate_count(self.h, leaf, step_index, step_seed, order)
        out = np.zeros(max(cap, 1), dtype=MOVE_DTYPE)
        n = lib().sfo_model_enumerate(self.h, leaf, step_index, step_seed, order, _p(out), cap)
        return out[: min(n, cap)]

    def enumerate_count(self, leaf, step_index=0, step_seed=0, order=ORDER_ORIGINAL):
        return lib().sfo_model_enumerate_count(self.h, leaf, step_index, step_seed, order)

    def evaluate_moves(self, moves):
        moves = np.ascontiguousarray(moves, dtype=MOVE_DTYPE)
        scores = np.zeros((len(moves), 4), dtype=np.int64)
        doable = np.zeros(len(moves), dtype=np.int32)
        lib().sfo_model_evaluate_moves(self.h, _p(moves), len(moves), _p(scores), _p(doable))
        return scores, doable

    def evaluate_compound(self, candidates):
        """candidates: list of lists of (entity, to_value) edits (to_value -1 = None), each scored as ONE CompoundScalarMove."""
        edits, offsets = compound_wire(candidates)
        scores = np.zeros((len(candidates), 4), dtype=np.int64)
        doable = np.zeros(len(candidates), dtype=np.int32)
        lib().sfo_model_evaluate_compound(self.h, _p(edits), _p(offsets), len(candidates), _p(scores), _p(doable))
        return scores, doable

    def step_grouped(self, candidates, group_name_len=0, max_moves_per_step=0, gates=None):
        """One local-search step of a GroupedScalarMoveSelector over `candidates` (the ScalarCandidateProvider's output for the working
        solution): (kept provider indices in pull order, scores [consumed, 4], flags [consumed], selected ordinal or -1)."""
        edits, offsets = compound_wire(candidates)
        n = len(candidates)
        kept = np.zeros(max(n, 1), dtype=np.int64)
        scores = np.zeros((max(n, 1), 4), dtype=np.int64)
        flags = np.zeros(max(n, 1), dtype=np.int32)
        consumed, selected = C.c_int64(0), C.c_int64(-1)
        if gates is not None:  # bit 0 requires_hard_improvement, bit 1 requires_score_improvement (evaluation.rs:75-113)
            g = np.ascontiguousarray(gates, dtype=np.int32)
            k = lib().sfo_model_step_grouped_gated(self.h, _p(edits), _p(offsets), _p(g), n, group_name_len, max_moves_per_step, _p(kept), _p(scores),
                                                   _p(flags), C.byref(consumed), C.byref(selected))
            return kept[:k], scores[:consumed.value], flags[:consumed.value], int(selected.value)
        k = lib().sfo_model_step_grouped(self.h, _p(edits), _p(offsets), n, group_name_len, max_moves_per_step, _p(kept), _p(scores), _p(flags),
                                         C.byref(consumed), C.byref(selected))
        return kept[:k], scores[:consumed.value], flags[:consumed.value], int(selected.value)

    def step_cursor(self, candidates, gates=None):
        """One local-search step over a cursor's own pull order (RuntimeProviderCursor: nothing re-ordered, filtered or capped):
        (scores [consumed, 4], flags [consumed], committed index or -1)."""
        g = np.zeros(len(candidates), dtype=np.int32) if gates is None else gates
        _, sc, fl, sel = self.step_grouped(candidates, group_name_len=-1, gates=g)
        return sc, fl, sel

    def apply_compound(self, candidate):
        edits, _ = compound_wire([candidate])
        lib().sfo_model_apply_compound(self.h, _p(edits), len(candidate))

    def apply_move(self, move):
        mv = np.zeros(1, dtype=MOVE_DTYPE)
        mv[0] = move
        lib().sfo_model_apply_move(self.h, _p(mv))

    def construct_list_cheapest(self, elements):
        """List cheapest-insertion construction of the unassigned `elements` (source order)."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        lib().sfo_model_construct_list_cheapest(self.h, _p(el), len(el))

    def construct_list_regret(self, elements, order_keys=None, owners=None):
        """List regret-insertion construction of the unassigned `elements` (source order; order_keys = element_order_key;
        owners = the owner hook's value per element, -1 unrestricted)."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        ks = None if order_keys is None else np.ascontiguousarray(order_keys, dtype=np.int64)
        ow = None if owners is None else np.ascontiguousarray(owners, dtype=np.int64)
        lib().sfo_model_construct_list_regret(self.h, _p(el), len(el), None if ks is None else _p(ks), None if ow is None else _p(ow))

    def construct_list_clarke_wright(self, elements, feasible_mode=0):
        """Clarke-Wright savings construction of the unassigned `elements` (source order) with the solverforge-cvrp hooks;
        feasible_mode 0 = structural (savings_hooks), 1 = capacity (route_hooks).  Returns (committed, stats[5])."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        st = np.zeros(5, dtype=np.uint64)
        rc = lib().sfo_model_construct_list_clarke_wright(self.h, _p(el), len(el), int(feasible_mode), _p(st))
        return bool(rc), st

    def construct_list_round_robin(self, elements, order_keys=None, owners=None):
        """Round-robin list construction of the unassigned `elements` (source order); order_keys / owners parallel to it
        (owners: -1 unrestricted, otherwise the owner hook's value)."""
        el = np.ascontiguousarray(elements, dtype=np.uint32)
        ks = None if order_keys is None else np.ascontiguousarray(order_keys, dtype=np.int64)
        ow = None if owners is None else np.ascontiguousarray(owners, dtype=np.int64)
        lib().sfo_model_construct_list_round_robin(self.h, _p(el), len(el), None if ks is None else _p(ks), None if ow is None else _p(ow))

    def construct_list_k_opt(self, k=2, feasible_mode=1, max_sweeps=1000):
        """Route-local 2-opt polishing (ListKOptPhase) with the stock CVRP route hooks; at most max_sweeps sweeps per route
        (0 = unlimited); returns stats[4] = candidates, accepted, applied, steps."""
        st = np.zeros(4, dtype=np.uint64)
        lib().sfo_model_construct_list_k_opt(self.h, int(k), int(feasible_mode), int(max_sweeps), _p(st))
        return st

    def set_time_windows(self, lo, hi, service, travel, departure=0):
        """Time windows / service durations / travel times of a CVRP model (solverforge-cvrp problem_data.rs:20-23): read by
        construct_list_k_opt(feasible_mode=2) and route_feasible; the scoring path does not use them."""
        a = [np.ascontiguousarray(x, dtype=np.int64) for x in (lo, hi, service, travel)]
        lib().sfo_model_set_time_windows(self.h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), int(departure))

    def route_feasible(self, route):
        r = np.ascontiguousarray(route, dtype=np.uint32)
        return bool(lib().sfo_model_route_feasible(self.h, _p(r), len(r)))

    def construct_first_fit(self):
        lib().sfo_model_construct_first_fit(self.h)

    def get_vars(self, desc=0, var=0):
        out = np.zeros(self.n_by_desc[desc], dtype=np.int64)
        lib().sfo_model_get_vars(self.h, desc, var, _p(out))
        return out

    def get_lists(self, desc=0, max_elements=1 << 20):
        n = self.n_by_desc[desc]
        off = np.zeros(n + 1, dtype=np.uint32)
        vals = np.zeros(max_elements, dtype=np.uint32)
        t = lib().sfo_model_get_lists(self.h, desc, _p(off), _p(vals))
        return [list(map(int, vals[off[i]: off[i + 1]])) for i in range(n)]


def compound_wire(candidates):
    """(edits as Change-shaped wire moves, offsets[n + 1]) of a list of multi-edit candidates."""
    offsets = np.zeros(len(candidates) + 1, dtype=np.int64)
    for i, c in enumerate(candidates):
        offsets[i + 1] = offsets[i] + len(c)
    edits = np.zeros(max(int(offsets[-1]), 1), dtype=MOVE_DTYPE)
    k = 0
    for c in candidates:
        for (entity, value) in c:
            edits[k] = (KIND_CHANGE, entity, 0, 0, 0, value)
            k += 1
    return edits, offsets


def scoped_seed(base_seed, descriptor_index, variable_name, selector_kind):
    return lib().sfo_scoped_seed(base_seed & 0xFFFFFFFFFFFFFFFF, descriptor_index, variable_name.encode(), selector_kind.encode())


def hash_str(s):
    return lib().sfo_hash_str(s.encode())


def random_range_stream(seed, low, high_inclusive, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().sfo_random_range_stream(seed, low, high_inclusive, n, _p(out))
    return out


def ruin_positions(move):
    """The ascending list positions of a KIND_RUIN wire move."""
    w = [int(move["b"]) & 0xFFFFFFFF, int(move["b_pos"]) & 0xFFFFFFFF, int(move["value"]) & 0xFFFFFFFF]
    return [(w[i // 2] >> (16 * (i & 1))) & 0xFFFF for i in range(int(move["a_pos"]))]


def splitmix64(v):
    return lib().sfo_splitmix64(v & 0xFFFFFFFFFFFFFFFF)
