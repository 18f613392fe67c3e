// Scalar-variable hot path (graph colouring / N-queens / machine assignment), gfx950 wave64.
//
// Reference semantics restated (paths under crates/solverforge-solver/src/ unless noted):
//   heuristic/selector/scalar_neighborhood/cursor/change.rs:27-121   scalar change stream
//   heuristic/selector/scalar_neighborhood/cursor/swap.rs:22-160     scalar swap stream
//   heuristic/selector/scalar_neighborhood/cursor.rs:371-378         slot_identity
//   heuristic/selector/scalar_neighborhood/move/apply.rs:15-35,219-245 doability
//   heuristic/move/change.rs:118-221, heuristic/move/swap.rs:150-215 move semantics
//   crates/solverforge-scoring/src/constraint/incremental.rs:19-193              uni (unassigned)
//   crates/solverforge-scoring/src/constraint/cross_bi_incremental/{state,incremental}.rs
//       predicate cross-join on one class (constant key, stream/join_target.rs:82-110)
//   examples/scalar-graph-coloring/src/domain/graph_coloring.rs:21-44, examples/nqueens/src/domain/board.rs:21-47
//
// GPU formulation: the reference re-tests all n rows of the predicate join on every insert
// (cross_bi_incremental/state.rs:392-397).  A predicate "partners with an equal value" only ever
// matches inside a declared partner set (graph adjacency, same group), so the device keeps a
// symmetric partner CSR and a trial delta is conflicts(e, new) - conflicts(e, old): deg(e) colour
// lookups, no state mutation, identical integer result.
//
// Engine: one wavefront = one search replica, values (i8 / i16) in the wave's LDS slice, rings of
// candidate coordinates per leaf, 64-wide replay in union cursor order (same step loop as
// sf_list_wave.hip; citations for acceptor / forager / union there and in sf_list_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_list_model.h"

namespace sf {

#ifndef SF_CONFLICT_W
#define SF_CONFLICT_W 16  // partner ids in flight per pass of scalar_conflict_delta
#endif
// SC_PARTNERS_EQUAL / SC_QUEENS are SPECIALISATIONS of the pair-predicate program (include/solverforge_amd.h: sf_pair_term) for the two
// programs the reference's examples use; SC_IR_PARTNERS / SC_IR_DENSE interpret any program (partner-index driven / every other entity).
enum ScalarCrossKind : int32_t { SC_NONE = 0, SC_PARTNERS_EQUAL = 1, SC_QUEENS = 2, SC_IR_PARTNERS = 3, SC_IR_DENSE = 4 };
constexpr int SF_IR_MAX = 6;  // residual terms interpreted per pair
// one residual term: op (sf_pair_op) | clause << 8, its column / CSR / table
struct PairTerm {
    int32_t op_clause;
    int32_t cols;          // SF_PAIR_TABLE_NONZERO: columns of the table
    int64_t param;
    const int32_t* col;    // i32 column (the key column for the table op)
    const uint32_t* coff;  // SF_PAIR_CSR_CONTAINS: row offsets / values
    const uint32_t* cval;
    const int64_t* table;
};

struct ScalarModel {
    int32_t n = 0;  // entities
    int32_t n_values = 0;
    int32_t allows_unassigned = 0;
    int32_t levels = 2;
    int32_t descriptor = 0, variable = 0;  // slot identity (cursor.rs:371-378)
    // constraints (level < 0 = absent)
    int32_t un_level = -1, cross_level = -1, cross_kind = SC_NONE;
    int64_t un_weight = 0, cross_weight = 0;
    const int32_t* un_w = nullptr;     // [n] optional per-entity weight of the unassigned constraint: for_each(E).filter(f(e) &&
                                       // unassigned).penalize(w(e)) with f / w as data (0 = filtered out); null = 1 for every entity
    const uint32_t* pn_off = nullptr;  // [n+1] symmetric partner CSR (SC_PARTNERS_EQUAL)
    const uint32_t* pn = nullptr;
    const int32_t* col = nullptr;      // [n] column fact (SC_QUEENS)
    int32_t ir_n = 0;                  // SC_IR_*: residual terms of the pair predicate, clause ids ascending
    const PairTerm* ir = nullptr;      // [ir_n] in device memory (wave-uniform reads: scalar loads on demand, nothing rides in the argument block)
    // value-keyed aggregates (per-value count / sum tables, maintained at apply):
    int32_t sj_level = -1, grp_level = -1;  // keyed self-join pairs; grouped sum
    int64_t sj_weight = 0, grp_weight = 0, grp_cap = -1;
    int32_t sj_arity = 2;  // keyed self-join arity: 2 pairs, 3 / 4 / 5 = tri / quad / penta tuples sharing a value
    int32_t grp_mode = 0;  // 0: sum of per-group weights (grouped node + sum collector); 1: load_balance collector (unfairness of
                           // the per-value metric sums); 2: BalanceConstraint (base x standard deviation of the per-value COUNTS)
    int64_t bal_base = 0;  // mode 2: the base score of one unit of standard deviation on grp_level (grp_weight is 1)
    // mode 0 variants: grp_shape 1 = |sum - grp_cap| (absolute deviation from a target); grp_complement = every value row is
    // scored, a row without members with the default result 0 (constraint/complemented/*.rs: `.complement(B, key, |_| 0)`)
    int32_t grp_shape = 0, grp_complement = 0;
    const int32_t* size = nullptr;     // [n] summed fact of the grouped constraint
    // keyed cross-join with a fact side: every assigned entity e matches the fact row its value names; filter + weight of the
    // pair are the data cost[e][value] (0 = filtered out)
    int32_t cost_level = -1;
    int64_t cost_weight = 0;
    const int64_t* cost = nullptr;     // [n][n_values]
    // a second (entity, value) cost matrix on another score level (round 6: the uni programs of a class may sit on two levels -- a hard filter beside
    // soft weights; weights and scales are folded into the entries)
    int32_t cost2_level = -1;
    const int64_t* cost2 = nullptr;    // [n][n_values]
    // exists / not-exists of planning entities per value-keyed fact row (uses the per-value count table)
    int32_t ex_level = -1, ex_mode = 1;  // 1: scored while some entity holds the value, 0: while none does
    int64_t ex_weight = 0;
    const int32_t* ex_w = nullptr;     // [n_values] per-row weight (null = 1)
    // ValueSource::EntitySlice (builder/context/scalar/variable.rs:138-151): per-entity value lists as CSR, null = the countable
    // range 0..n_values for every entity
    const uint32_t* vl_off = nullptr;  // [n + 1]
    const int32_t* vl = nullptr;       // values, each in 0..n_values
    // group_by(value, consecutive_runs(point)).penalize(weight * sum over the runs of max(0, point_count - run_limit))
    // (stream/collector/runs.rs): a per-(value, point) count table [n_values][run_P] of u16 follows the per-value tables
    int32_t run_level = -1, run_P = 0;
    int64_t run_weight = 0, run_limit = 0;
    // run_mode 1: the same table read as an indexed_presence result (stream/collector/indexed_presence.rs): a row scores
    // min(count_in(run_lo..run_hi), run_cap) -- run_cap 0 = uncapped (count / count_in), 1 = any_in.  run_mode 2: the row scores the
    // excess of its complement_runs(run_lo..run_hi) -- maximal runs of ABSENT points inside the horizon -- over run_limit; only a
    // row with members is a group (an empty row scores nothing), points >= run_P are never present
    int32_t run_mode = 0, run_lo = 0, run_hi = 0;
    int64_t run_cap = 0;
    const int32_t* run_point = nullptr;  // [n] point of every entity, in 0..run_P
    __host__ __device__ __forceinline__ bool tables() const { return sj_level >= 0 || grp_level >= 0 || ex_level >= 0 || run_level >= 0; }
    // per-replica committed state
    int32_t* vals = nullptr;        // [R][n]  (-1 = None)
    int64_t* score = nullptr;       // [R][4]
    int32_t* best_vals = nullptr;   // [R][n]
    int64_t* best_score = nullptr;  // [R][4]
};

// the canonical value list of entity e (ValueSelector::iter, heuristic/selector/value_selector.rs:21-41)
__device__ __forceinline__ uint32_t value_count(const ScalarModel& m, uint32_t e) {
    return m.vl_off ? m.vl_off[e + 1] - m.vl_off[e] : (uint32_t)m.n_values;
}
// value at stream offset `off` of entity e's list (selection_index over the list, cursor/change.rs:91-104)
__device__ __forceinline__ int32_t value_at(const ScalarModel& m, const StreamCtx& ctx, uint32_t e, uint32_t off, uint64_t salt, const FastMod& fm_vc) {
    if (!m.vl_off) return (int32_t)ctx.selection_index_fm(off, fm_vc, salt);
    const uint32_t b = m.vl_off[e];
    return m.vl[b + ctx.selection_index(off, m.vl_off[e + 1] - b, salt)];
}
// destination_is_legal (cursor/swap.rs:103-123): None needs allows_unassigned, a value must be in the row's list
__device__ __forceinline__ bool value_legal(const ScalarModel& m, uint32_t e, int32_t v) {
    if (v < 0) return m.allows_unassigned != 0;
    if (!m.vl_off) return true;
    for (uint32_t p = m.vl_off[e]; p < m.vl_off[e + 1]; ++p)
        if (m.vl[p] == v) return true;
    return false;
}

// The residual clauses of the pair predicate for the pair (e, o), o assigned with value vo, for TWO candidate values of e at once (a
// negative candidate = unassigned: never holds).  Returns bit 0 = holds with v0, bit 1 = holds with v1.  Wave-uniform control (the
// program is read with scalar loads, one term per trip of a loop that is NOT unrolled: inlined and unrolled at every pair of every
// partner loop it tripled the build time of the search kernels), per-lane data; the column facts are read once for both values.
// left = the lower entity index (the join's left.id < right.id).
// SF_SCALAR_PAIR_IR (a scalar-engine unit is built twice, csrc/Makefile): 0 = the interpreted joins compile OUT of scalar_conflict_delta -- the unit of
// the models whose program matched a specialised loop (graph colouring, queens, job-shop groups): their kernels do not carry the interpreter's live
// ranges (with it in: 9 -> 61 spilled vector registers and -4 % on the default SA policy of graph colouring, profiles/r06k_pair_ir_ab.txt); 1 = the
// unit the host launches for SC_IR_PARTNERS / SC_IR_DENSE models.  Every other unit (the generic engine, the C ABI) keeps both.  W = 4 everywhere (8 spills).
#ifndef SF_SCALAR_PAIR_IR
#define SF_SCALAR_PAIR_IR 1
#endif
#ifndef SF_PAIR_IR_W
#define SF_PAIR_IR_W 4  // partners of one entity evaluated side by side by the interpreted join (pair_program_holds2_w)
#endif
__device__ __forceinline__ uint32_t pair_program_holds2(const PairTerm* __restrict__ ir, int32_t ir_n, uint32_t e, uint32_t o, int32_t v0, int32_t v1, int32_t vo) {
    const bool swp = e > o;
    const uint32_t l = swp ? o : e, r = swp ? e : o;
    uint32_t all = 3u, any = 0u;
    int32_t clause = -1;
#pragma unroll 1
    for (int t = 0; t < ir_n; ++t) {
        const PairTerm& pt = ir[t];
        const int32_t op = pt.op_clause & 255, cl = pt.op_clause >> 8;
        if (cl != clause) {
            all &= clause < 0 ? 3u : any;
            any = 0u;
            clause = cl;
        }
        uint32_t h = 0u;
        if (op == 1 || op == 2 || op == 11) {  // value-only terms
            const int32_t d0 = v0 > vo ? v0 - vo : vo - v0, d1 = v1 > vo ? v1 - vo : vo - v1;
            const bool h0 = op == 1 ? v0 == vo : (op == 2 ? v0 != vo : (int64_t)d0 <= pt.param);
            const bool h1 = op == 1 ? v1 == vo : (op == 2 ? v1 != vo : (int64_t)d1 <= pt.param);
            h = (h0 ? 1u : 0u) | (h1 ? 2u : 0u);
        } else if (op == 9) {  // membership either way: a linear walk of both rows (residual use only: a CSR clause of its own drives the partner index)
            bool m = false;
            for (uint32_t q = pt.coff[l]; q < pt.coff[l + 1]; ++q) m = m || pt.cval[q] == r;
            for (uint32_t q = pt.coff[r]; q < pt.coff[r + 1]; ++q) m = m || pt.cval[q] == l;
            h = m ? 3u : 0u;
        } else {
            const int32_t cl_ = pt.col[l], cr_ = pt.col[r];
            const int32_t dc = cl_ > cr_ ? cl_ - cr_ : cr_ - cl_;
            if (op == 3) {
                const int32_t d0 = v0 > vo ? v0 - vo : vo - v0, d1 = v1 > vo ? v1 - vo : vo - v1;
                h = (d0 == dc ? 1u : 0u) | (d1 == dc ? 2u : 0u);
            } else {
                bool m = false;
                if (op == 4)
                    m = cl_ == cr_;
                else if (op == 5)
                    m = cl_ != cr_;
                else if (op == 6)
                    m = cl_ < cr_;
                else if (op == 7)
                    m = (int64_t)dc == pt.param;
                else if (op == 8)
                    m = (int64_t)dc <= pt.param;
                else if (op == 10)
                    m = pt.table[(size_t)cl_ * (size_t)pt.cols + (size_t)cr_] != 0;
                h = m ? 3u : 0u;
            }
        }
        any |= h;
    }
    all &= clause < 0 ? 3u : any;
    return all & ((v0 >= 0 ? 1u : 0u) | (v1 >= 0 ? 2u : 0u));
}

// The same program for W partners of e side by side (round 6).  The interpreted join used to walk its partner list one entity at a time -- a partner id
// from HBM / L2, its value from LDS, then per term the two column facts from HBM again, every load waiting for the one before: 2.9 x slower than the
// specialised loop, which keeps sixteen partner ids in flight (profiles/r05_pair_ir_graph.txt).  Here the term loop stays wave-uniform and NOT unrolled
// (one copy of each operator's code), and inside a term the W partners are independent: their fact loads issue together.  An inactive slot
// (`on[q]` false) carries o[q] = e: every index stays readable, its result is dropped.
template <int W>
__device__ __forceinline__ void pair_program_holds2_w(const PairTerm* __restrict__ ir, int32_t ir_n, uint32_t e, const uint32_t (&o)[W], const bool (&on)[W], int32_t v0,
                                                      int32_t v1, const int32_t (&vo)[W], uint32_t (&out)[W]) {
    uint32_t l[W], r[W], all[W], any[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const bool swp = e > o[q];
        l[q] = swp ? o[q] : e, r[q] = swp ? e : o[q];
        all[q] = 3u, any[q] = 0u;
    }
    int32_t clause = -1;
#pragma unroll 1
    for (int t = 0; t < ir_n; ++t) {
        const PairTerm& pt = ir[t];
        const int32_t op = pt.op_clause & 255, cl = pt.op_clause >> 8;
        if (cl != clause) {
#pragma unroll
            for (int q = 0; q < W; ++q) {
                all[q] &= clause < 0 ? 3u : any[q];
                any[q] = 0u;
            }
            clause = cl;
        }
        if (op == 1 || op == 2 || op == 11) {  // value-only terms
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int32_t d0 = v0 > vo[q] ? v0 - vo[q] : vo[q] - v0, d1 = v1 > vo[q] ? v1 - vo[q] : vo[q] - v1;
                const bool h0 = op == 1 ? v0 == vo[q] : (op == 2 ? v0 != vo[q] : (int64_t)d0 <= pt.param);
                const bool h1 = op == 1 ? v1 == vo[q] : (op == 2 ? v1 != vo[q] : (int64_t)d1 <= pt.param);
                any[q] |= (h0 ? 1u : 0u) | (h1 ? 2u : 0u);
            }
        } else if (op == 9) {  // membership either way: a linear walk of both rows (residual use only)
#pragma unroll 1
            for (int q = 0; q < W; ++q) {
                if (!on[q]) continue;
                bool m = false;
                for (uint32_t k = pt.coff[l[q]]; k < pt.coff[l[q] + 1]; ++k) m = m || pt.cval[k] == r[q];
                for (uint32_t k = pt.coff[r[q]]; k < pt.coff[r[q] + 1]; ++k) m = m || pt.cval[k] == l[q];
                any[q] |= m ? 3u : 0u;
            }
        } else {
            int32_t cl_[W], cr_[W];
#pragma unroll
            for (int q = 0; q < W; ++q) cl_[q] = pt.col[l[q]], cr_[q] = pt.col[r[q]];  // 2 W independent loads
            if (op == 10) {
                int64_t tb[W];
#pragma unroll
                for (int q = 0; q < W; ++q) tb[q] = pt.table[(size_t)cl_[q] * (size_t)pt.cols + (size_t)cr_[q]];
#pragma unroll
                for (int q = 0; q < W; ++q) any[q] |= tb[q] != 0 ? 3u : 0u;
            } else {
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const int32_t dc = cl_[q] > cr_[q] ? cl_[q] - cr_[q] : cr_[q] - cl_[q];
                    if (op == 3) {
                        const int32_t d0 = v0 > vo[q] ? v0 - vo[q] : vo[q] - v0, d1 = v1 > vo[q] ? v1 - vo[q] : vo[q] - v1;
                        any[q] |= (d0 == dc ? 1u : 0u) | (d1 == dc ? 2u : 0u);
                    } else {
                        const bool m = op == 4 ? cl_[q] == cr_[q] : (op == 5 ? cl_[q] != cr_[q] : (op == 6 ? cl_[q] < cr_[q] : (op == 7 ? (int64_t)dc == pt.param : (op == 8 ? (int64_t)dc <= pt.param : false))));
                        any[q] |= m ? 3u : 0u;
                    }
                }
            }
        }
    }
    const uint32_t vm = (v0 >= 0 ? 1u : 0u) | (v1 >= 0 ? 2u : 0u);
#pragma unroll
    for (int q = 0; q < W; ++q) out[q] = on[q] ? ((all[q] & (clause < 0 ? 3u : any[q])) & vm) : 0u;
}

// matches of entity e against every partner except `skip`, for two candidate values at once:
// returns conflicts(e, v_new) - conflicts(e, v_old) in ONE pass over the partner list, sixteen
// partner ids in flight per iteration (the list lives in HBM/L2, the values in LDS): an average
// graph-colouring row (degree 20) costs two memory round trips instead of twenty.
// `vals` is anything indexable by entity: the replica's value array (const VT*) or a view of it (ValsView)
template <class VA>
__device__ __forceinline__ int64_t scalar_conflict_delta(const ScalarModel& m, const VA& vals, uint32_t e, int32_t v_new,
                                                         int32_t v_old, uint32_t skip) {
    int32_t c = 0;
    if (m.cross_kind == SC_PARTNERS_EQUAL) {
        const uint32_t p1 = m.pn_off[e + 1];
        constexpr int W = SF_CONFLICT_W;
        for (uint32_t p = m.pn_off[e]; p < p1; p += W) {
            uint32_t o[W];
#pragma unroll
            for (int q = 0; q < W; ++q) o[q] = p + q < p1 ? m.pn[p + q] : skip;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                if (o[q] == skip || p + q >= p1) continue;
                const int32_t vo = (int32_t)vals[o[q]];
                c += (v_new >= 0 && vo == v_new) ? 1 : 0;
                c -= (v_old >= 0 && vo == v_old) ? 1 : 0;
            }
        }
#if SF_SCALAR_PAIR_IR
    } else if (m.cross_kind == SC_IR_PARTNERS) {  // the partner index names the pairs one clause admits, the program decides the rest
        const uint32_t p1 = m.pn_off[e + 1];
        constexpr int W = SF_PAIR_IR_W;
#pragma unroll 1
        for (uint32_t p = m.pn_off[e]; p < p1; p += W) {
            uint32_t o[W], h[W];
            bool on[W];
            int32_t vo[W];
#pragma unroll
            for (int q = 0; q < W; ++q) o[q] = p + q < p1 ? m.pn[p + q] : e;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                vo[q] = (int32_t)vals[o[q]];
                on[q] = p + q < p1 && o[q] != skip && o[q] != e && vo[q] >= 0;
            }
            pair_program_holds2_w<W>(m.ir, m.ir_n, e, o, on, v_new, v_old, vo, h);
#pragma unroll
            for (int q = 0; q < W; ++q) c += (int32_t)(h[q] & 1u) - (int32_t)(h[q] >> 1);
        }
    } else if (m.cross_kind == SC_IR_DENSE) {  // no clause to index by: every other assigned entity
        constexpr int W = SF_PAIR_IR_W;
#pragma unroll 1
        for (uint32_t b = 0; b < (uint32_t)m.n; b += W) {
            uint32_t o[W], h[W];
            bool on[W];
            int32_t vo[W];
#pragma unroll
            for (int q = 0; q < W; ++q) {
                o[q] = b + q < (uint32_t)m.n ? b + q : e;
                vo[q] = (int32_t)vals[o[q]];
                on[q] = b + q < (uint32_t)m.n && o[q] != e && o[q] != skip && vo[q] >= 0;
            }
            pair_program_holds2_w<W>(m.ir, m.ir_n, e, o, on, v_new, v_old, vo, h);
#pragma unroll
            for (int q = 0; q < W; ++q) c += (int32_t)(h[q] & 1u) - (int32_t)(h[q] >> 1);
        }
#endif
    } else if (m.cross_kind == SC_QUEENS) {  // board.rs:30-44: distinct columns, same row or same diagonal
        const int32_t ce = m.col[e];
        for (uint32_t o = 0; o < (uint32_t)m.n; ++o) {
            const int32_t vo = (int32_t)vals[o], co = m.col[o];
            if (o == e || o == skip || vo < 0 || co == ce) continue;
            const int32_t dc = co > ce ? co - ce : ce - co;
            if (v_new >= 0) {
                const int32_t dr = vo > v_new ? vo - v_new : v_new - vo;
                c += (vo == v_new || dr == dc) ? 1 : 0;
            }
            if (v_old >= 0) {
                const int32_t dr = vo > v_old ? vo - v_old : v_old - vo;
                c -= (vo == v_old || dr == dc) ? 1 : 0;
            }
        }
    }
    return (int64_t)c;
}

// ---- consecutive-runs collector (stream/collector/runs.rs:11-229) over the per-(value, point) count table ----------------
// The table sits behind the per-value count table (16-byte aligned): a pointer to the counts locates it.
__device__ __forceinline__ uint16_t* runs_table(const ScalarModel& m, const uint32_t* cnt) {
    return (uint16_t*)(((uintptr_t)(cnt + m.n_values) + 15) & ~(uintptr_t)15);
}
template <class T>
struct TableView;
template <class T>
__device__ __forceinline__ uint16_t* runs_table(const ScalarModel&, const TableView<T>&) { return nullptr; }  // compound candidates: not supported
__host__ __device__ inline size_t runs_table_bytes(int n_values, int run_P) { return ((size_t)n_values * run_P * 2 + 31) & ~(size_t)15; }
__device__ __forceinline__ int64_t run_excess(const ScalarModel& m, int64_t len) { return len > m.run_limit ? len - m.run_limit : 0; }
// lengths of the runs of present points directly left / right of point d in one value's row; point `ox` counts as absent
__device__ __forceinline__ void run_neighbours(const uint16_t* row, int P, int d, int ox, int64_t& left, int64_t& right) {
    left = right = 0;
    for (int x = d - 1; x >= 0 && x != ox && row[x] != 0; --x) ++left;
    for (int x = d + 1; x < P && x != ox && row[x] != 0; ++x) ++right;
}
// indexed_presence mode: present points of a row inside [run_lo, run_hi), point `ox` and point `skip` counted as absent
__device__ __forceinline__ int64_t presence_count(const ScalarModel& m, const uint16_t* row, int skip, int ox) {
    int64_t c = 0;
    for (int x = m.run_lo; x < m.run_hi; ++x) c += (x != skip && x != ox && row[x] != 0) ? 1 : 0;
    return c;
}
__device__ __forceinline__ int64_t presence_capped(const ScalarModel& m, int64_t c) { return (m.run_cap > 0 && c > m.run_cap) ? m.run_cap : c; }
// complement mode: lengths of the runs of absent points directly left / right of d inside the horizon; `ox` counts as absent
__device__ __forceinline__ bool presence_at(const ScalarModel& m, const uint16_t* row, int x, int ox) { return x < m.run_P && x != ox && row[x] != 0; }
__device__ __forceinline__ void gap_neighbours(const ScalarModel& m, const uint16_t* row, int d, int ox, int64_t& left, int64_t& right) {
    left = right = 0;
    for (int x = d - 1; x >= m.run_lo && !presence_at(m, row, x, ox); --x) ++left;
    for (int x = d + 1; x < m.run_hi && !presence_at(m, row, x, ox); ++x) ++right;
}
// complement mode: the score of a row whose only present point is d
__device__ __forceinline__ int64_t gap_single_point(const ScalarModel& m, int d) {
    if (d < m.run_lo || d >= m.run_hi) return run_excess(m, (int64_t)m.run_hi - m.run_lo);
    return run_excess(m, (int64_t)d - m.run_lo) + run_excess(m, (int64_t)m.run_hi - 1 - d);
}
// change of a row's summed run excess when point d becomes present (it was absent) / absent (it was the last item of the point)
__device__ __forceinline__ int64_t run_add_delta(const ScalarModel& m, const uint16_t* row, int d, int ox) {
    if (m.run_mode == 2) {  // the gap around d splits in two
        if (d < m.run_lo || d >= m.run_hi) return 0;
        int64_t l, r;
        gap_neighbours(m, row, d, ox, l, r);
        return run_excess(m, l) + run_excess(m, r) - run_excess(m, l + 1 + r);
    }
    if (m.run_mode == 1) {  // the row's presence score with d present minus without it
        if (d < m.run_lo || d >= m.run_hi) return 0;
        if (m.run_cap <= 0) return 1;
        const int64_t c = presence_count(m, row, d, ox);
        return presence_capped(m, c + 1) - presence_capped(m, c);
    }
    int64_t l, r;
    run_neighbours(row, m.run_P, d, ox, l, r);
    return run_excess(m, l + 1 + r) - run_excess(m, l) - run_excess(m, r);
}
// summed run excess of one row (full evaluation)
__device__ __forceinline__ int64_t run_row_excess(const ScalarModel& m, const uint16_t* row) {
    if (m.run_mode == 1) return presence_capped(m, presence_count(m, row, -1, -1));
    if (m.run_mode == 2) {  // callers gate on the row having members
        int64_t total = 0, len = 0;
        for (int x = m.run_lo; x < m.run_hi; ++x) {
            if (!presence_at(m, row, x, -1))
                ++len;
            else {
                total += run_excess(m, len);
                len = 0;
            }
        }
        return total + run_excess(m, len);
    }
    int64_t total = 0, len = 0;
    for (int x = 0; x < m.run_P; ++x) {
        if (row[x] != 0)
            ++len;
        else {
            total += run_excess(m, len);
            len = 0;
        }
    }
    return total + run_excess(m, len);
}

struct ScalarDelta {
    int64_t d_un;     // change of the number of unassigned entities
    int64_t d_cross;  // change of the number of matched pairs (predicate join)
    int64_t d_pairs;  // change of the number of same-value pairs (keyed self-join)
    int64_t d_grp;    // change of the summed group weights
    int64_t d_cost;   // change of the summed (entity, value) pair costs
    int64_t d_ex;     // change of the summed weights of the value rows whose existence test holds
    bool doable;
    int64_t d_run = 0;  // change of the summed run excess (consecutive-runs collector)
    int64_t d_cost2 = 0;  // change of the second cost matrix's sum (ScalarModel::cost2)
};

// grouped/scorer.rs:89-101: an empty group scores zero
__device__ __forceinline__ int64_t group_weight(const ScalarModel& m, int64_t sum, uint32_t count) {
    if (count == 0 && !m.grp_complement) return 0;
    if (count == 0) sum = 0;  // the complement's default result
    if (m.grp_shape == 1) {
        const int64_t dev = wsub(sum, m.grp_cap);
        return dev < 0 ? wsub(0, dev) : dev;
    }
    if (m.grp_cap < 0) return (int64_t)((uint64_t)sum * (uint64_t)sum);
    const int64_t over = wsub(sum, m.grp_cap);
    return over > 0 ? over : 0;
}

// tuples of `k` entities among `c` sharing a value: C(c, k), wrapping u64 (k <= 5; every prefix product is divisible)
__device__ __forceinline__ uint64_t choose_u64(uint64_t c, int k) {
    if (c < (uint64_t)k) return 0;
    uint64_t r = 1;
    for (int i = 1; i <= k; ++i) r = r * (c - (uint64_t)k + (uint64_t)i) / (uint64_t)i;
    return r;
}

// load_balance collector (stream/collector/load_balance.rs:167-184): unfairness = round(sqrt(fraction / n + integral)) with
// integral = sum of squared loads, fraction numerator = -(sum of loads)^2 (what the reference's incremental update
// maintains), n = keys holding at least one item; the same f64 operations in the same order (IEEE division, correctly
// rounded sqrt, round half away from zero); a NaN (negative radicand from rounding) casts to 0 like Rust's `as i64`.
__device__ __forceinline__ int64_t lb_unfairness(int64_t s1, int64_t s2, uint32_t nk) {
    if (nk == 0) return 0;
    const double frac = (double)(int64_t)(0ull - (uint64_t)s1 * (uint64_t)s1);
    const double tmp = nk == 1 ? frac + (double)s2 : frac / (double)nk + (double)s2;
    if (!(tmp >= 0.0)) return 0;
    return (int64_t)round(sqrt(tmp));
}
// BalanceConstraint (constraint/balance.rs:162-184): round(base * sqrt(sum_sq / n - (total / n)^2)), f64 in the reference's
// operation order; `Score::multiply` rounds half away from zero (score/macros.rs:61-63)
__device__ __forceinline__ int64_t balance_value(int64_t base, int64_t total, int64_t sum_sq, uint32_t nk) {
    if (nk == 0) return 0;
    const double n = (double)nk, mean = (double)total / n;
    const double variance = ((double)sum_sq / n) - (mean * mean);
    const double sd = variance <= 0.0 ? 0.0 : sqrt(variance);
    return (int64_t)round((double)base * sd);
}
// the global statistic of the value-keyed loads: mode 1 load_balance unfairness, mode 2 balance
__device__ __forceinline__ int64_t global_stat(const ScalarModel& m, int64_t s1, int64_t s2, uint32_t nk) {
    return m.grp_mode == 2 ? balance_value(m.bal_base, s1, s2, nk) : lb_unfairness(s1, s2, nk);
}
// one key's load x -> x + d inside (S1, S2)
__device__ __forceinline__ void lb_shift(int64_t& s1, int64_t& s2, int64_t x, int64_t d) {
    const int64_t y = wadd(x, d);
    s2 = wadd(wsub(s2, (int64_t)((uint64_t)x * (uint64_t)x)), (int64_t)((uint64_t)y * (uint64_t)y));
    s1 = wadd(s1, d);
}
// (S1, S2, keys, unfairness) of a per-value table pair, serially (cold paths: evaluate / apply kernels)
__device__ __forceinline__ void lb_from_tables(const ScalarModel& m, const uint32_t* cnt, const int64_t* sum, int64_t* lb) {
    int64_t s1 = 0, s2 = 0;
    uint32_t nk = 0;
    for (int v = 0; v < m.n_values; ++v)
        if (cnt[v]) {
            const int64_t x = m.grp_mode == 2 ? (int64_t)cnt[v] : sum[v];  // a key's load: its metric sum, or its entity count
            s1 = wadd(s1, x);
            s2 = wadd(s2, (int64_t)((uint64_t)x * (uint64_t)x));
            nk += 1;
        }
    lb[0] = s1, lb[1] = s2, lb[2] = (int64_t)nk, lb[3] = global_stat(m, s1, s2, nk);
}

// kind 0: Change(a -> value); kind 1: Swap(a, b)
// `cnt` / `sum`: per-value entity count and summed size of the step snapshot (null when the model has
// no value-keyed constraint)
// vals / cnt / sum: plain arrays, or views with operator[] (a compound candidate chains its edits through ValsView / TableView)
template <class VA, class CA, class SA>
__device__ __forceinline__ ScalarDelta eval_scalar_move_v(const ScalarModel& m, const VA& vals, int kind, uint32_t a,
                                                          uint32_t b, int32_t value, const CA& cnt, const SA& sum, const int64_t* lb) {
    ScalarDelta r{0, 0, 0, 0, 0, 0, false};
    if (kind == 0) {  // apply.rs:15-24,219-230
        if (a >= (uint32_t)m.n || value >= m.n_values || value < -1) return r;
        const int32_t old = (int32_t)vals[a];
        if (old == value) return r;
        if (value < 0 && !m.allows_unassigned) return r;
        r.doable = true;
        r.d_un = (int64_t)((value < 0 ? 1 : 0) - (old < 0 ? 1 : 0)) * (m.un_w ? (int64_t)m.un_w[a] : 1);
        if (m.cross_level >= 0)
            r.d_cross = scalar_conflict_delta(m, vals, a, value, old, 0xFFFFFFFFu);
        if (m.sj_level >= 0) {  // joining a value of c members adds C(c, k-1) tuples; leaving one of c removes C(c-1, k-1)
            if (m.sj_arity == 2)
                r.d_pairs = (value >= 0 ? (int64_t)cnt[value] : 0) - (old >= 0 ? (int64_t)cnt[old] - 1 : 0);
            else
                r.d_pairs = (int64_t)((value >= 0 ? choose_u64(cnt[value], m.sj_arity - 1) : 0ull) -
                                      (old >= 0 ? choose_u64(cnt[old] - 1u, m.sj_arity - 1) : 0ull));
        }
        if (m.cost_level >= 0)
            r.d_cost = wsub(value >= 0 ? m.cost[(size_t)a * m.n_values + value] : 0, old >= 0 ? m.cost[(size_t)a * m.n_values + old] : 0);
        if (m.cost2_level >= 0)
            r.d_cost2 = wsub(value >= 0 ? m.cost2[(size_t)a * m.n_values + value] : 0, old >= 0 ? m.cost2[(size_t)a * m.n_values + old] : 0);
        if (m.ex_level >= 0) {  // the row `value` starts to exist when it had no holder, the row `old` stops when a was its last
            const int64_t gain = (value >= 0 && cnt[value] == 0) ? (m.ex_w ? (int64_t)m.ex_w[value] : 1) : 0;
            const int64_t loss = (old >= 0 && cnt[old] == 1) ? (m.ex_w ? (int64_t)m.ex_w[old] : 1) : 0;
            r.d_ex = m.ex_mode ? wsub(gain, loss) : wsub(loss, gain);
        }
        if (m.run_level >= 0) {  // the point of `a` leaves the row of `old` / joins the row of `value` (distinct rows)
            const uint16_t* pt = runs_table(m, cnt);
            if (pt) {
                const int d = m.run_point[a];
                if (m.run_mode == 2) {  // an empty row is no group: the last member leaving drops the row's score, the first brings it
                    if (old >= 0) {
                        if (cnt[old] == 1) r.d_run -= gap_single_point(m, d);
                        else if (pt[(size_t)old * m.run_P + d] == 1) r.d_run -= run_add_delta(m, pt + (size_t)old * m.run_P, d, -1);
                    }
                    if (value >= 0) {
                        if (cnt[value] == 0) r.d_run += gap_single_point(m, d);
                        else if (pt[(size_t)value * m.run_P + d] == 0) r.d_run += run_add_delta(m, pt + (size_t)value * m.run_P, d, -1);
                    }
                } else {
                if (old >= 0 && pt[(size_t)old * m.run_P + d] == 1) r.d_run -= run_add_delta(m, pt + (size_t)old * m.run_P, d, -1);
                if (value >= 0 && pt[(size_t)value * m.run_P + d] == 0) r.d_run += run_add_delta(m, pt + (size_t)value * m.run_P, d, -1);
                }
            }
        }
        if (m.grp_level >= 0 && m.grp_mode >= 1) {  // global statistic (load balance: metrics are >= 1, validated at sf_constraint_add)
            const bool by_count = m.grp_mode == 2;
            const int64_t sz = by_count ? 1 : (int64_t)m.size[a];
            int64_t s1 = lb[0], s2 = lb[1];
            uint32_t nk = (uint32_t)lb[2];
            if (old >= 0) {
                lb_shift(s1, s2, by_count ? (int64_t)cnt[old] : (int64_t)sum[old], -sz);
                nk -= cnt[old] == 1 ? 1u : 0u;
            }
            if (value >= 0) {
                lb_shift(s1, s2, by_count ? (int64_t)cnt[value] : (int64_t)sum[value], sz);
                nk += cnt[value] == 0 ? 1u : 0u;
            }
            r.d_grp = wsub(global_stat(m, s1, s2, nk), lb[3]);
        } else if (m.grp_level >= 0) {
            const int64_t sz = (int64_t)m.size[a];
            int64_t d = 0;
            if (old >= 0) d = wsub(group_weight(m, wsub(sum[old], sz), cnt[old] - 1), group_weight(m, sum[old], cnt[old]));
            if (value >= 0) d = wadd(d, wsub(group_weight(m, wadd(sum[value], sz), cnt[value] + 1), group_weight(m, sum[value], cnt[value])));
            r.d_grp = d;
        }
    } else {  // apply.rs:25-35,232-245
        if (a >= (uint32_t)m.n || b >= (uint32_t)m.n || a == b) return r;
        const int32_t va = (int32_t)vals[a], vb = (int32_t)vals[b];
        if (va == vb) return r;
        r.doable = true;
        if (m.un_w)  // the None moves to the other entity: the count stays, the per-entity weights differ
            r.d_un = (int64_t)((vb < 0 ? 1 : 0) - (va < 0 ? 1 : 0)) * ((int64_t)m.un_w[a] - (int64_t)m.un_w[b]);
        if (m.cross_level >= 0)
            r.d_cross = scalar_conflict_delta(m, vals, a, vb, va, b) + scalar_conflict_delta(m, vals, b, va, vb, a);
        if (m.cost_level >= 0) {
            const size_t ra = (size_t)a * m.n_values, rb = (size_t)b * m.n_values;
            const int64_t after = wadd(vb >= 0 ? m.cost[ra + vb] : 0, va >= 0 ? m.cost[rb + va] : 0);
            const int64_t before = wadd(va >= 0 ? m.cost[ra + va] : 0, vb >= 0 ? m.cost[rb + vb] : 0);
            r.d_cost = wsub(after, before);
        }
        if (m.cost2_level >= 0) {
            const size_t ra = (size_t)a * m.n_values, rb = (size_t)b * m.n_values;
            const int64_t after = wadd(vb >= 0 ? m.cost2[ra + vb] : 0, va >= 0 ? m.cost2[rb + va] : 0);
            const int64_t before = wadd(va >= 0 ? m.cost2[ra + va] : 0, vb >= 0 ? m.cost2[rb + vb] : 0);
            r.d_cost2 = wsub(after, before);
        }
        if (m.run_level >= 0) {  // row va loses a's point and gains b's; row vb the other way round (nothing moves when the points agree)
            const uint16_t* pt = runs_table(m, cnt);
            const int da = m.run_point[a], db = m.run_point[b];
            if (pt && da != db) {
                if (va >= 0) {
                    const uint16_t* row = pt + (size_t)va * m.run_P;
                    const bool gone = row[da] == 1;
                    if (gone) r.d_run -= run_add_delta(m, row, da, -1);
                    if (row[db] == 0) r.d_run += run_add_delta(m, row, db, gone ? da : -1);
                }
                if (vb >= 0) {
                    const uint16_t* row = pt + (size_t)vb * m.run_P;
                    const bool gone = row[db] == 1;
                    if (gone) r.d_run -= run_add_delta(m, row, db, -1);
                    if (row[da] == 0) r.d_run += run_add_delta(m, row, da, gone ? db : -1);
                }
            }
        }
        // a swap exchanges two members: per-value counts (and so the same-value pairs and every row's existence) do not change
        if (m.grp_level >= 0 && m.grp_mode == 1) {  // a swap exchanges two members: key counts stay, loads shift
            const int64_t sa = (int64_t)m.size[a], sb = (int64_t)m.size[b];
            int64_t s1 = lb[0], s2 = lb[1];
            uint32_t nk = (uint32_t)lb[2];
            // entity b joins va (unless a was unassigned: then b becomes unassigned) and entity a joins vb
            if (va >= 0) lb_shift(s1, s2, sum[va], wsub(sb, sa));
            if (vb >= 0) lb_shift(s1, s2, sum[vb], wsub(sa, sb));
            r.d_grp = wsub(lb_unfairness(s1, s2, nk), lb[3]);
        } else if (m.grp_level >= 0 && m.grp_mode == 2) {
            r.d_grp = 0;  // the per-value counts do not change under a swap
        } else if (m.grp_level >= 0) {
            const int64_t sa = (int64_t)m.size[a], sb = (int64_t)m.size[b];
            int64_t d = 0;
            if (va >= 0) d = wsub(group_weight(m, wadd(wsub(sum[va], sa), sb), cnt[va]), group_weight(m, sum[va], cnt[va]));
            if (vb >= 0) d = wadd(d, wsub(group_weight(m, wadd(wsub(sum[vb], sb), sa), cnt[vb]), group_weight(m, sum[vb], cnt[vb])));
            r.d_grp = d;
        }
    }
    return r;
}

template <class VT>
__device__ __forceinline__ ScalarDelta eval_scalar_move(const ScalarModel& m, const VT* vals, int kind, uint32_t a,
                                                        uint32_t b, int32_t value, const uint32_t* cnt = nullptr,
                                                        const int64_t* sum = nullptr, const int64_t* lb = nullptr) {
    return eval_scalar_move_v(m, vals, kind, a, b, value, cnt, sum, lb);
}

// committed update of the per-value tables (one lane)
template <class VT>
__device__ __forceinline__ void scalar_tables_apply(const ScalarModel& m, const VT* vals, int kind, uint32_t a, uint32_t b,
                                                    int32_t value, uint32_t* cnt, int64_t* sum) {
    if (m.run_level >= 0) {  // the per-(value, point) counts
        uint16_t* pt = runs_table(m, cnt);
        if (kind == 0) {
            const int32_t old = (int32_t)vals[a];
            const int d = m.run_point[a];
            if (old >= 0) pt[(size_t)old * m.run_P + d] -= 1;
            if (value >= 0) pt[(size_t)value * m.run_P + d] += 1;
        } else {
            const int32_t va = (int32_t)vals[a], vb = (int32_t)vals[b];
            const int da = m.run_point[a], db = m.run_point[b];
            if (va >= 0) {
                pt[(size_t)va * m.run_P + da] -= 1;
                pt[(size_t)va * m.run_P + db] += 1;
            }
            if (vb >= 0) {
                pt[(size_t)vb * m.run_P + db] -= 1;
                pt[(size_t)vb * m.run_P + da] += 1;
            }
        }
    }
    if (kind == 0) {
        const int32_t old = (int32_t)vals[a];
        const int64_t sz = m.size ? (int64_t)m.size[a] : 0;
        if (old >= 0) {
            cnt[old] -= 1;
            sum[old] = wsub(sum[old], sz);
        }
        if (value >= 0) {
            cnt[value] += 1;
            sum[value] = wadd(sum[value], sz);
        }
    } else {
        const int32_t va = (int32_t)vals[a], vb = (int32_t)vals[b];
        const int64_t sa = m.size ? (int64_t)m.size[a] : 0, sb = m.size ? (int64_t)m.size[b] : 0;
        if (va >= 0) sum[va] = wadd(wsub(sum[va], sa), sb);
        if (vb >= 0) sum[vb] = wadd(wsub(sum[vb], sb), sa);
    }
}

// builds the per-value tables of `vals` with all threads of the block / wave (tables pre-zeroed)
template <class VT>
__device__ __forceinline__ void scalar_tables_accumulate(const ScalarModel& m, const VT* vals, uint32_t tid, uint32_t nthreads,
                                                         uint32_t* cnt, int64_t* sum) {
    uint32_t* pt32 = nullptr;
    if (m.run_level >= 0) {  // zero the per-(value, point) counts first (two u16 per word; every caller runs this with all its threads)
        pt32 = (uint32_t*)runs_table(m, cnt);
        const uint32_t words = ((uint32_t)m.n_values * (uint32_t)m.run_P + 1u) / 2u;
        for (uint32_t w = tid; w < words; w += nthreads) pt32[w] = 0;
        if (nthreads == 64u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else
            __syncthreads();
    }
    for (uint32_t e = tid; e < (uint32_t)m.n; e += nthreads) {
        const int32_t v = (int32_t)vals[e];
        if (v >= 0) {
            atomicAdd(&cnt[v], 1u);
            atomicAdd((unsigned long long*)&sum[v], (unsigned long long)(int64_t)(m.size ? m.size[e] : 0));
            if (pt32) {
                const uint32_t idx = (uint32_t)v * (uint32_t)m.run_P + (uint32_t)m.run_point[e];
                atomicAdd(&pt32[idx >> 1], 1u << (16u * (idx & 1u)));
            }
        }
    }
}

template <int L>
__device__ __forceinline__ ScoreV<L> apply_scalar_delta(const ScalarModel& m, const int64_t* cur, const ScalarDelta& d) {
    ScoreV<L> s;
#pragma unroll
    for (int k = 0; k < L; ++k) s.v[k] = cur[k];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (k == m.un_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.un_weight * (uint64_t)d.d_un));
        if (k == m.cross_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.cross_weight * (uint64_t)d.d_cross));
        if (k == m.sj_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.sj_weight * (uint64_t)d.d_pairs));
        if (k == m.grp_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.grp_weight * (uint64_t)d.d_grp));
        if (k == m.cost_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.cost_weight * (uint64_t)d.d_cost));
        if (k == m.cost2_level) s.v[k] = wsub(s.v[k], d.d_cost2);
        if (k == m.ex_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.ex_weight * (uint64_t)d.d_ex));
        if (k == m.run_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.run_weight * (uint64_t)d.d_run));
    }
    return s;
}

// ---- compound candidates (planning/scalar/candidate.rs:85-188 -> heuristic/move/compound_scalar.rs:207-330) --------------------
// A ScalarCandidate of several ScalarEdits is ONE move: every edit is applied, then the score is read.  Every score level is
// a sum of integer match weights, so the move's delta is the telescoping sum of its edits applied one after the other; edit k
// is priced against the snapshot as seen through the earlier edits of the same candidate (views below, a few words per lane).
constexpr int SF_COMPOUND_MAX = 8;  // edits per candidate on the device
struct ValsView {
    const int32_t* vals;
    uint32_t ent[SF_COMPOUND_MAX];
    int32_t val[SF_COMPOUND_MAX];
    int n;
    __device__ __forceinline__ int32_t operator[](uint32_t o) const {
        int32_t v = vals[o];
        for (int i = 0; i < n; ++i)
            if (ent[i] == o) v = val[i];
        return v;
    }
    __device__ __forceinline__ void set(uint32_t o, int32_t v) {
        for (int i = 0; i < n; ++i)
            if (ent[i] == o) {
                val[i] = v;
                return;
            }
        ent[n] = o;
        val[n] = v;
        n += 1;
    }
};
template <class T>
struct TableView {  // per-value count / sum table plus the candidate's own shifts
    const T* base;
    int32_t key[2 * SF_COMPOUND_MAX];
    T delta[2 * SF_COMPOUND_MAX];
    int n;
    __device__ __forceinline__ T operator[](int32_t v) const {
        T x = base ? base[v] : (T)0;
        for (int i = 0; i < n; ++i)
            if (key[i] == v) x = (T)(x + delta[i]);
        return x;
    }
    __device__ __forceinline__ void add(int32_t v, T d) {
        for (int i = 0; i < n; ++i)
            if (key[i] == v) {
                delta[i] = (T)(delta[i] + d);
                return;
            }
        key[n] = v;
        delta[n] = d;
        n += 1;
    }
};

// n x evaluate_candidate for multi-edit candidates: candidate t = edits[offsets[t] .. offsets[t + 1]) (Change-shaped moves).
// is_doable_on (compound_scalar.rs:254-270): at least one edit, every to_value legal, some edit differs from the CURRENT value.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_scalar_evaluate_compound(ScalarModel m, int replica, const int32_t* edits, const int64_t* offsets, int64_t n,
                                                                  int64_t* out_scores, int32_t* out_doable) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];
    const int32_t* vals = m.vals + (size_t)replica * m.n;
    const bool tables = m.tables();
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    if (tables) {
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
        __syncthreads();
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t b = offsets[t], e = offsets[t + 1];
    bool legal = e > b && e - b <= SF_COMPOUND_MAX, changes = false;
    for (int64_t k = b; k < e && legal; ++k) {
        const int32_t* mv = edits + k * 6;
        legal = mv[0] == 0 && mv[1] >= 0 && mv[1] < m.n && mv[5] >= -1 && mv[5] < m.n_values && value_legal(m, (uint32_t)mv[1], mv[5]);
        if (legal) changes = changes || vals[mv[1]] != mv[5];
    }
    const bool doable = legal && changes;
    ScoreV<4> s;
    for (int k = 0; k < 4; ++k) s.v[k] = m.score[(size_t)replica * 4 + k];
    if (doable) {
        ValsView vv{vals, {}, {}, 0};
        TableView<uint32_t> cv{tables ? t_cnt : nullptr, {}, {}, 0};
        TableView<int64_t> sv{tables ? t_sum : nullptr, {}, {}, 0};
        for (int64_t k = b; k < e; ++k) {
            const uint32_t a = (uint32_t)edits[k * 6 + 1];
            const int32_t to = edits[k * 6 + 5];
            const int32_t old = vv[a];
            const ScalarDelta d = eval_scalar_move_v(m, vv, 0, a, 0u, to, cv, sv, (const int64_t*)nullptr);
            if (!d.doable) continue;  // this edit repeats the value its entity already has at this point
            s = apply_scalar_delta<4>(m, s.v, d);
            if (tables) {
                const int64_t sz = m.size ? (int64_t)m.size[a] : 0;
                if (old >= 0) cv.add(old, (uint32_t)-1), sv.add(old, -sz);
                if (to >= 0) cv.add(to, 1u), sv.add(to, sz);
            }
            vv.set(a, to);
        }
    }
    out_doable[t] = doable ? 1 : 0;
    for (int k = 0; k < m.levels; ++k) out_scores[t * m.levels + k] = doable ? s.v[k] : 0;
}

// committed do_move of one compound candidate (compound_scalar.rs:291-308), one lane
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_scalar_apply_compound(ScalarModel m, int replica, const int32_t* edits, int n_edits, int32_t* out_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];
    int32_t* vals = m.vals + (size_t)replica * m.n;
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    const bool tables = m.tables();
    if (tables) {
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
        __syncthreads();
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    bool legal = n_edits > 0, changes = false;
    for (int k = 0; k < n_edits && legal; ++k) {
        const int32_t* mv = edits + k * 6;
        legal = mv[0] == 0 && mv[1] >= 0 && mv[1] < m.n && mv[5] >= -1 && mv[5] < m.n_values && value_legal(m, (uint32_t)mv[1], mv[5]);
        if (legal) changes = changes || vals[mv[1]] != mv[5];
    }
    if (!legal || !changes) {
        *out_ok = 0;
        return;
    }
    int64_t* cur = m.score + (size_t)replica * 4;
    for (int k = 0; k < n_edits; ++k) {
        const uint32_t a = (uint32_t)edits[k * 6 + 1];
        const int32_t to = edits[k * 6 + 5];
        const ScalarDelta d = eval_scalar_move(m, vals, 0, a, 0u, to, t_cnt, t_sum, (const int64_t*)nullptr);
        if (!d.doable) continue;
        const ScoreV<4> s = apply_scalar_delta<4>(m, cur, d);
        if (tables) scalar_tables_apply(m, vals, 0, a, 0u, to, t_cnt, t_sum);
        vals[a] = to;
        for (int q = 0; q < 4; ++q) cur[q] = s.v[q];
    }
    *out_ok = 1;
}

// One HOST-DRIVEN local-search step (sf_step_decide): the candidates were generated on the host (a ScalarCandidateProvider behind a
// GroupedScalarMoveSelector, builder/selector/grouped_scalar.rs) and scored by k_scalar_evaluate_compound; this wave pulls them in
// order through the configured acceptor (HillClimbing / LateAcceptance / DiversifiedLateAcceptance) and forager exactly like the
// fused engines (phase/candidates.rs:47-285), commits the pick (CompoundScalarMove::do_move) and ends the step (step.rs:122-221):
// acceptor.step_ended, best solution, counters, step index.  One wavefront, replica `replica`.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_scalar_step_decide(ScalarModel m, SearchParams p, int replica, const int32_t* edits, const int64_t* offsets, int64_t n,
                                                          const int64_t* scores, const int32_t* doable_in, int32_t* out_flags, int64_t* out_result,
                                                          const int32_t* gates, int hard_levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];
    constexpr int L = 4;  // scores padded with zero levels: the lexicographic order is unchanged
    const uint32_t lane = threadIdx.x & 63u;
    const int r = replica;
    int32_t* vals = m.vals + (size_t)r * m.n;
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    const bool tables = m.tables();
    if (tables) {
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
        __syncthreads();
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
    }
    ScoreV<L> curv, best_sol, late, dla_thr;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        curv.v[k] = m.score[(size_t)r * 4 + k];
        best_sol.v[k] = m.best_score[(size_t)r * 4 + k];
        late.v[k] = 0;
    }
    const int la_slot = p.la_idx[r];
    if (p.acceptor == 1 || p.acceptor == 4) {
#pragma unroll
        for (int k = 0; k < L; ++k) late.v[k] = p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + k];
    }
    dla_thr = late;
    if (p.acceptor == 4) {
        ScoreV<L> db;
#pragma unroll
        for (int k = 0; k < L; ++k) db.v[k] = p.dla_best[(size_t)r * 4 + k];
        dla_thr = dla_threshold<L>(db, p.dla_tolerance);
    }
    const uint64_t draw = p.seed_draws[r];
    const uint64_t sseed = (p.explicit_seeds && (int64_t)draw < p.n_explicit) ? p.explicit_seeds[(size_t)r * p.n_explicit + draw]
                                                                            : step_seed(p.random_seed + (uint64_t)r, draw);
    int has_best = 0;
    uint64_t equal_count = 0;
    uint32_t accepted = 0;
    ScoreV<L> best = curv;
    int64_t best_i = -1;
    uint64_t st_gen = 0, st_acc = 0, st_calc = 0;
    int64_t consumed_total = 0;
    for (int64_t base = 0; base < n; base += 64) {
        const uint32_t nvalid = (uint32_t)((n - base) < 64 ? (n - base) : 64);
        const bool valid = lane < nvalid;
        const int64_t ci = base + lane;
        ScoreV<L> sc = curv;
        bool doable = false;
        if (valid) {
            doable = doable_in[ci] != 0;
#pragma unroll
            for (int k = 0; k < L; ++k) sc.v[k] = k < m.levels ? scores[ci * m.levels + k] : 0;
        }
        // evaluate_candidate's gates (phase/localsearch/evaluation.rs:75-113): a move that requires a hard improvement (bit 0:
        // hard_score_delta != Improving, phase/hard_delta.rs:11-35) or a score improvement (bit 1: move score <= last step score) is
        // scored and counted but never reaches the acceptor
        bool consult = doable;
        int32_t gate_flag = 0;  // 8 RejectedByHardImprovement, 16 RejectedByScoreImprovement
        if (doable && gates) {
            const int32_t g = gates[ci];
            if (g & 1) {
                bool improving = false;
                for (int k = 0; k < hard_levels && k < L; ++k) {
                    if (sc.v[k] == curv.v[k]) continue;
                    improving = sc.v[k] > curv.v[k];
                    break;
                }
                consult = improving;
                if (!consult) gate_flag = 8;
            }
            if (consult && (g & 2) && score_cmp<L>(sc, curv) <= 0) consult = false, gate_flag = 16;
        }
        bool acc = false;
        if (consult) {
            if (p.acceptor == 0)
                acc = score_cmp<L>(sc, curv) > 0;
            else if (p.acceptor == 1)
                acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0;
            else if (p.acceptor == 4)
                acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0 || score_cmp<L>(sc, dla_thr) >= 0;
        }
        bool improving_pick = false;
        const ScoreV<L> forager_thr = p.forager == FORAGER_FIRST_BEST_IMPROVING ? best_sol : curv;
        const uint32_t nconsumed = forager_chunk_cut<L>(p.forager, (uint32_t)p.limit, accepted, acc, sc, forager_thr, nvalid, improving_pick);
        const bool consumed = lane < nconsumed;
        acc = acc && consumed;
        const uint64_t accmask = __ballot(acc);
        if (accmask) {
            if (improving_pick) {
                const int sel = (int)nconsumed - 1;
#pragma unroll
                for (int k = 0; k < L; ++k) best.v[k] = (int64_t)shfl_u64((uint64_t)sc.v[k], sel);
                best_i = base + sel;
                equal_count = 1;
                has_best = 1;
            } else if (p.forager == 1) {
                if (!has_best) {
                    const int sel = __ffsll((unsigned long long)accmask) - 1;
#pragma unroll
                    for (int k = 0; k < L; ++k) best.v[k] = (int64_t)shfl_u64((uint64_t)sc.v[k], sel);
                    best_i = base + sel;
                    has_best = 1;
                }
            } else if (!has_best || __ballot(acc && score_cmp<L>(sc, best) >= 0)) {
                const ScoreV<L> M = wave_max_score<L>(sc, acc);
                const int cm = has_best ? score_cmp<L>(M, best) : 1;
                if (cm >= 0) {
                    const bool newmax = cm > 0;
                    const uint64_t eq_base = newmax ? 0 : equal_count;
                    const bool in_eq = acc && score_cmp<L>(sc, M) == 0;
                    const uint64_t eq = __ballot(in_eq);
                    const uint32_t rank = mbcnt64(eq) + 1u;
                    const uint64_t cntq = eq_base + rank;
                    const bool pick = in_eq && ((newmax && rank == 1) || (p.random_ties && cntq > 1 && reservoir_pick(sseed, cntq)));
                    const uint64_t pm = __ballot(pick);
                    if (pm) best_i = base + (63 - __clzll((unsigned long long)pm));
                    best = M;
                    equal_count = eq_base + (uint64_t)__popcll(eq);
                    has_best = 1;
                }
            }
        }
        accepted += (uint32_t)__popcll(accmask);
        st_gen += nconsumed;
        st_acc += (uint64_t)__popcll(accmask);
        st_calc += (uint64_t)__popcll(__ballot(consumed && doable));
        if (consumed) out_flags[ci] = (doable ? 1 : 0) | (acc ? 2 : 0) | gate_flag;
        consumed_total += nconsumed;
        if (forager_quits(p.forager, (uint32_t)p.limit, accepted, has_best, improving_pick)) break;
    }
    __syncthreads();
    // commit the pick: CompoundScalarMove::do_move (one lane, as k_scalar_apply_compound)
    if (has_best && lane == 0) {
        out_flags[best_i] |= 4;
        int64_t* cur = m.score + (size_t)r * 4;
        for (int64_t k = offsets[best_i]; k < offsets[best_i + 1]; ++k) {
            const uint32_t a = (uint32_t)edits[k * 6 + 1];
            const int32_t to = edits[k * 6 + 5];
            const ScalarDelta d = eval_scalar_move(m, vals, 0, a, 0u, to, t_cnt, t_sum, (const int64_t*)nullptr);
            if (!d.doable) continue;
            const ScoreV<4> s2 = apply_scalar_delta<4>(m, cur, d);
            if (tables) scalar_tables_apply(m, vals, 0, a, 0u, to, t_cnt, t_sum);
            vals[a] = to;
            for (int q = 0; q < 4; ++q) cur[q] = s2.v[q];
        }
    }
    __syncthreads();
    if (has_best) curv = best;  // == the score the commit accumulated (asserted by the parity tests through sf_get_scores)
    const bool improved = has_best && score_cmp<L>(curv, best_sol) > 0;
    if (improved)  // update_best_solution (scope_progress.rs:89-107)
        for (uint32_t t = lane; t < (uint32_t)m.n; t += 64) m.best_vals[(size_t)r * m.n + t] = vals[t];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
            p.last_step_score[(size_t)r * 4 + k] = curv.v[k];
            if (improved) m.best_score[(size_t)r * 4 + k] = curv.v[k];
            if (p.acceptor == 1 || p.acceptor == 4) p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + k] = curv.v[k];
        }
        if (p.acceptor == 4) {
            ScoreV<L> db;
#pragma unroll
            for (int k = 0; k < L; ++k) db.v[k] = p.dla_best[(size_t)r * 4 + k];
            if (score_cmp<L>(curv, db) > 0) {
#pragma unroll
                for (int k = 0; k < L; ++k) p.dla_best[(size_t)r * 4 + k] = curv.v[k];
            }
        }
        p.la_idx[r] = la_slot + 1 >= p.la_size ? 0 : la_slot + 1;
        p.step_index[r] += 1;
        p.seed_draws[r] += 1;
        uint64_t* gs = p.stats + (size_t)r * SF_STATS_WORDS;
        gs[0] += 1;
        gs[1] += st_gen;
        gs[2] += st_gen;
        gs[3] += st_acc;
        gs[4] += has_best ? 1 : 0;
        gs[5] += st_calc;
        gs[6] += st_gen - st_calc;
        gs[7] += st_gen;
        out_result[0] = consumed_total;
        out_result[1] = has_best ? best_i : -1;
    }
}

// evaluate_all / initialize: full recomputation (fresh_score; FullAssert).  grid = R blocks.
// accumulate != 0: add this class's constraint scores to what the list class already wrote (mixed models)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_scalar_evaluate_all(ScalarModel m, int64_t* out_scores, int commit,
                                                             int accumulate, int64_t* out_parts = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];  // per-value tables (when used)
    __shared__ unsigned long long s_un, s_cross, s_pairs, s_grp, s_groups, s_cost, s_cost_n, s_ex, s_ex_n, s_run, s_run_groups, s_un_n, s_cost2;
    const int r = blockIdx.x;
    const int32_t* vals = m.vals + (size_t)r * m.n;
    const bool tables = m.tables();
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    if (threadIdx.x == 0) {
        s_un = 0;
        s_cross = 0;
        s_pairs = 0;
        s_grp = 0;
        s_groups = 0;
        s_cost = s_cost_n = s_ex = s_ex_n = 0;
        s_cost2 = 0;
        s_run = s_run_groups = 0;
        s_un_n = 0;
    }
    if (tables)
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
    __syncthreads();
    if (tables) {
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
        unsigned long long pairs = 0, grp = 0, groups = 0, ex = 0, ex_n = 0;
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            const unsigned long long c = t_cnt[v];
            pairs += m.sj_arity == 2 ? c * (c - (c ? 1 : 0)) / 2 : choose_u64(c, m.sj_arity);
            if (m.grp_mode == 0 && m.grp_level >= 0) grp += (unsigned long long)group_weight(m, t_sum[v], t_cnt[v]);
            groups += (c || m.grp_complement) ? 1 : 0;  // complemented: one match per value row (complemented/incremental.rs:54-57)
            if (m.ex_level >= 0 && ((c > 0) == (m.ex_mode != 0))) {
                ex += (unsigned long long)(int64_t)(m.ex_w ? m.ex_w[v] : 1);
                ex_n += 1;
            }
        }
        if (m.run_level >= 0) {  // one value row per thread: its runs
            unsigned long long run = 0, run_groups = 0;
            const uint16_t* pt = runs_table(m, t_cnt);
            for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
                if (m.run_mode != 2 || t_cnt[v]) run += (unsigned long long)run_row_excess(m, pt + (size_t)v * m.run_P);
                run_groups += t_cnt[v] ? 1 : 0;
            }
            atomicAdd(&s_run, run);
            atomicAdd(&s_run_groups, run_groups);
        }
        atomicAdd(&s_ex, ex);
        atomicAdd(&s_ex_n, ex_n);
        atomicAdd(&s_pairs, pairs);
        atomicAdd(&s_grp, grp);
        atomicAdd(&s_groups, groups);
        if (m.grp_level >= 0 && m.grp_mode >= 1) {
            __syncthreads();
            if (threadIdx.x == 0) {
                int64_t lb[4];
                lb_from_tables(m, t_cnt, t_sum, lb);
                s_grp = (unsigned long long)lb[3];
                s_groups = lb[2] ? 1 : 0;  // one group (the unit key) when any entity is in it
                if (m.grp_mode == 2) {  // match_count of the balance constraint: keys deviating from the mean by more than 0.5
                    unsigned long long dev = 0;
                    const double mean = lb[2] ? (double)lb[0] / (double)lb[2] : 0.0;
                    for (int v = 0; v < m.n_values; ++v)
                        if (t_cnt[v] && fabs((double)t_cnt[v] - mean) > 0.5) dev += 1;
                    s_groups = dev;
                }
            }
        }
    }
    unsigned long long un = 0, un_n = 0, cross = 0, cost = 0, cost_n = 0, cost2 = 0;
    for (uint32_t e = threadIdx.x; e < (uint32_t)m.n; e += blockDim.x) {
        const int32_t v = vals[e];
        if (v < 0) {
            un += (unsigned long long)(int64_t)(m.un_w ? m.un_w[e] : 1);
            un_n += (!m.un_w || m.un_w[e] != 0) ? 1 : 0;
        }
        if (m.cost_level >= 0 && v >= 0) {
            const int64_t c = m.cost[(size_t)e * m.n_values + v];
            cost += (unsigned long long)c;
            cost_n += c != 0 ? 1 : 0;
        }
        if (m.cost2_level >= 0 && v >= 0) cost2 += (unsigned long long)m.cost2[(size_t)e * m.n_values + v];
        if (m.cross_level >= 0 && v >= 0) {
            // every matched pair is seen from both sides: count it at its lower index
            if (m.cross_kind == SC_PARTNERS_EQUAL) {
                for (uint32_t p = m.pn_off[e]; p < m.pn_off[e + 1]; ++p) {
                    const uint32_t o = m.pn[p];
                    if (o > e && vals[o] == v) ++cross;
                }
            } else if (m.cross_kind == SC_IR_PARTNERS) {
                for (uint32_t p = m.pn_off[e]; p < m.pn_off[e + 1]; ++p) {
                    const uint32_t o = m.pn[p];
                    if (o > e && vals[o] >= 0 && (pair_program_holds2(m.ir, m.ir_n, e, o, v, -1, vals[o]) & 1u)) ++cross;
                }
            } else if (m.cross_kind == SC_IR_DENSE) {
                for (uint32_t o = e + 1; o < (uint32_t)m.n; ++o)
                    if (vals[o] >= 0 && (pair_program_holds2(m.ir, m.ir_n, e, o, v, -1, vals[o]) & 1u)) ++cross;
            } else if (m.cross_kind == SC_QUEENS) {
                const int32_t ce = m.col[e];
                for (uint32_t o = e + 1; o < (uint32_t)m.n; ++o) {
                    const int32_t vo = vals[o], co = m.col[o];
                    if (vo < 0 || co == ce) continue;
                    const int32_t dr = vo > v ? vo - v : v - vo, dc = co > ce ? co - ce : ce - co;
                    if (vo == v || dr == dc) ++cross;
                }
            }
        }
    }
    atomicAdd(&s_un, un);
    atomicAdd(&s_un_n, un_n);
    atomicAdd(&s_cross, cross);
    atomicAdd(&s_cost, cost);
    atomicAdd(&s_cost_n, cost_n);
    atomicAdd(&s_cost2, cost2);
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t sc[SF_MAX_LEVELS_CONST] = {0, 0, 0, 0};
        if (m.un_level >= 0) sc[m.un_level] = wsub(sc[m.un_level], (int64_t)((uint64_t)m.un_weight * s_un));
        if (m.cross_level >= 0) sc[m.cross_level] = wsub(sc[m.cross_level], (int64_t)((uint64_t)m.cross_weight * s_cross));
        if (m.sj_level >= 0) sc[m.sj_level] = wsub(sc[m.sj_level], (int64_t)((uint64_t)m.sj_weight * s_pairs));
        if (m.grp_level >= 0) sc[m.grp_level] = wsub(sc[m.grp_level], (int64_t)((uint64_t)m.grp_weight * s_grp));
        if (m.cost_level >= 0) sc[m.cost_level] = wsub(sc[m.cost_level], (int64_t)((uint64_t)m.cost_weight * s_cost));
        if (m.cost2_level >= 0) sc[m.cost2_level] = wsub(sc[m.cost2_level], (int64_t)s_cost2);
        if (m.ex_level >= 0) sc[m.ex_level] = wsub(sc[m.ex_level], (int64_t)((uint64_t)m.ex_weight * s_ex));
        if (m.run_level >= 0) sc[m.run_level] = wsub(sc[m.run_level], (int64_t)((uint64_t)m.run_weight * s_run));
        for (int k = 0; k < m.levels; ++k) {
            if (out_scores) out_scores[(size_t)r * m.levels + k] = accumulate ? wadd(out_scores[(size_t)r * m.levels + k], sc[k]) : sc[k];
            if (commit) m.score[(size_t)r * 4 + k] = accumulate ? wadd(m.score[(size_t)r * 4 + k], sc[k]) : sc[k];
        }
        if (out_parts) {
            int64_t* q = out_parts + (size_t)r * SF_EACH_WORDS;
            q[3] = (int64_t)s_un;
            q[4] = (int64_t)s_cross;
            q[5] = (int64_t)s_pairs;
            q[6] = (int64_t)s_grp;
            q[7] = (int64_t)s_groups;
            q[8] = (int64_t)s_cost;
            q[9] = (int64_t)s_cost_n;
            q[10] = (int64_t)s_ex;
            q[11] = (int64_t)s_ex_n;
            q[14] = (int64_t)s_run;
            q[15] = (int64_t)s_run_groups;
            q[16] = (int64_t)s_un_n;
        }
    }
}

// n x evaluate_candidate for host-provided moves (the ScalarCandidateProvider surface: a batch of
// ScalarEdit{entity, to_value} is kind = SF_MOVE_CHANGE): one thread per move, state unchanged.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_scalar_evaluate_moves(ScalarModel m, int replica, const int32_t* moves,
                                                               int64_t n, int64_t* out_scores, int32_t* out_doable,
                                                               int skip_foreign) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];
    const int32_t* vals = m.vals + (size_t)replica * m.n;
    const bool tables = m.tables();
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    if (tables) {  // every block rebuilds the per-value tables of the snapshot
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
        __syncthreads();
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
    }
    __shared__ int64_t s_lb[4];
    if (m.grp_level >= 0 && m.grp_mode >= 1) {
        if (threadIdx.x == 0) lb_from_tables(m, t_cnt, t_sum, s_lb);
        __syncthreads();
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (skip_foreign && moves[t * 6] != 0 && moves[t * 6] != 1) return;  // a list move of a mixed model
    const int64_t* cur = m.score + (size_t)replica * 4;
    const int32_t* mv = moves + t * 6;
    ScalarDelta d{0, 0, 0, 0, 0, 0, false};
    if (mv[0] == 0 && mv[1] >= 0)
        d = eval_scalar_move(m, vals, 0, (uint32_t)mv[1], 0u, mv[5], t_cnt, t_sum, s_lb);
    else if (mv[0] == 1 && mv[1] >= 0 && mv[3] >= 0)
        d = eval_scalar_move(m, vals, 1, (uint32_t)mv[1], (uint32_t)mv[3], 0, t_cnt, t_sum, s_lb);
    out_doable[t] = d.doable ? 1 : 0;
    const ScoreV<4> s = apply_scalar_delta<4>(m, cur, d);
    for (int k = 0; k < m.levels; ++k) out_scores[t * m.levels + k] = d.doable ? s.v[k] : 0;
}

// committed Change / Swap on global state (sf_apply)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_scalar_apply(ScalarModel m, int replica, int kind, int a, int b, int value,
                                                     int32_t* out_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab_mem[];
    int32_t* vals = m.vals + (size_t)replica * m.n;
    int64_t* t_sum = (int64_t*)tab_mem;
    uint32_t* t_cnt = (uint32_t*)(tab_mem + sizeof(int64_t) * (size_t)m.n_values);
    if (m.tables()) {
        for (int v = threadIdx.x; v < m.n_values; v += blockDim.x) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
        __syncthreads();
        scalar_tables_accumulate(m, vals, threadIdx.x, blockDim.x, t_cnt, t_sum);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    int64_t* cur = m.score + (size_t)replica * 4;
    int64_t lb[4] = {0, 0, 0, 0};
    if (m.grp_level >= 0 && m.grp_mode >= 1) lb_from_tables(m, t_cnt, t_sum, lb);
    const ScalarDelta d = eval_scalar_move(m, vals, kind, (uint32_t)a, (uint32_t)b, value, t_cnt, t_sum, lb);
    if (!d.doable) {
        *out_ok = 0;
        return;
    }
    const ScoreV<4> s = apply_scalar_delta<4>(m, cur, d);
    if (kind == 0)
        vals[a] = value;
    else {
        const int32_t t = vals[a];
        vals[a] = vals[b];
        vals[b] = t;
    }
    for (int k = 0; k < 4; ++k) cur[k] = s.v[k];
    *out_ok = 1;
}

SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_scalar_phase_start(ScalarModel m, SearchParams p) {
    const int r = blockIdx.x;
    const int64_t* cur = m.score + (size_t)r * 4;
    for (int k = threadIdx.x; k < 4; k += blockDim.x) {
        p.last_step_score[(size_t)r * 4 + k] = cur[k];
        m.best_score[(size_t)r * 4 + k] = cur[k];
        if (p.dla_best) p.dla_best[(size_t)r * 4 + k] = cur[k];  // DiversifiedLateAcceptance::phase_started
    }
    for (int h = threadIdx.x; h < p.la_size * 4; h += blockDim.x) p.la_hist[(size_t)r * p.la_size * 4 + h] = cur[h & 3];
    for (int t = threadIdx.x; t < m.n; t += blockDim.x) m.best_vals[(size_t)r * m.n + t] = m.vals[(size_t)r * m.n + t];
    if (threadIdx.x == 0) {
        p.la_idx[r] = 0;
        p.step_index[r] = 0;
        p.has_best[r] = 1;
    }
}

// ---------------------------------------------------------------------------------------
// Fused search kernel: one wavefront = one replica.
// ---------------------------------------------------------------------------------------
constexpr uint32_t SRC = 128;  // ring capacity per leaf (entries of 2 x u32)

template <class VT>
struct SCarve {
    size_t vals, ring, tsum, tcnt, tpt, total;
    // n_table = n_values when a value-keyed constraint exists, else 0; run_P = points of the consecutive-runs table, else 0
    __host__ __device__ __forceinline__ SCarve(int n, int n_table, int run_P = 0) {
        size_t o = 0;
        ring = o;
        o = align_up(o + sizeof(uint32_t) * 2 * SRC * 2, 16);
        tsum = o;
        o = align_up(o + sizeof(int64_t) * n_table, 16);
        tcnt = o;
        o = align_up(o + sizeof(uint32_t) * n_table, 16);
        tpt = o;  // == runs_table(cnt): the aligned end of the count table
        o = align_up(o + (run_P ? runs_table_bytes(n_table, run_P) : 0), 16);
        vals = o;
        o = align_up(o + sizeof(VT) * n, 16);
        total = o;
    }
};

// VT = int8_t (n_values <= 127) or int16_t: the replica's values in LDS
// IRK = SF_SCALAR_PAIR_IR of the unit that instantiates it: the two builds of one (L, TRACE, VT) are different kernels by name
template <int L, bool TRACE, class VT, bool IRK = (SF_SCALAR_PAIR_IR != 0)>
#ifndef SF_SCALAR_BLOCKS_PER_CU
#define SF_SCALAR_BLOCKS_PER_CU 3  // (round 5: 168 registers, 26-33 spilled values; 12 replicas per CU -- graph colouring 10k: LateAcceptance 6.85 -> 8.45 G moves/s, the default SimulatedAnnealing policy 82 -> 103 M, profiles/r05_graph_occupancy.txt)
#endif
__global__ __launch_bounds__(64 * 4, SF_SCALAR_BLOCKS_PER_CU) void k_scalar_search_wave(ScalarModel m, SearchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint64_t s_sa[4][SA_WORDS];  // SimulatedAnnealing acceptor state of the resident replicas
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_group = uni(threadIdx.x >> 6);  // wave-uniform by construction: per-replica base pointers in scalar registers
    const int rr = (int)(blockIdx.x * 4 + wave_in_group);
    if (rr >= p.n_launch) return;  // no workgroup barrier below
    const int r = rr + p.replica_base;
    uint64_t* saw = s_sa[wave_in_group];
    const bool annealing = p.acceptor == 3;
    if (annealing) sa_load(saw, p.sa, r, lane);
    const uint32_t n = (uint32_t)m.n;
    const bool tables = m.tables();
    const SCarve<VT> cv(m.n, tables ? m.n_values : 0, m.run_level >= 0 ? m.run_P : 0);
    unsigned char* mem = smem + (size_t)wave_in_group * cv.total;
    uint32_t* ring = (uint32_t*)(mem + cv.ring);  // [leaf][SRC][2]
    VT* s_vals = (VT*)(mem + cv.vals);
    int64_t* t_sum = (int64_t*)(mem + cv.tsum);  // per-value summed size / entity count of the working state
    uint32_t* t_cnt = (uint32_t*)(mem + cv.tcnt);
    int32_t* g_vals = m.vals + (size_t)r * n;
    int64_t* g_score = m.score + (size_t)r * 4;
    const bool tracing = TRACE && r == p.trace_replica;

    // leaves in default-policy declaration order: change, then swap (policy/scalar.rs:67-106)
    const int n_leaves = p.n_leaves;
    const bool chg0 = p.leaf[0].kind == 1, chg1 = n_leaves > 1 && p.leaf[1].kind == 1;
    const uint64_t identity = ((uint64_t)(uint32_t)m.descriptor << 32) ^ (uint64_t)(uint32_t)m.variable;
    const uint32_t vc = (uint32_t)m.n_values;  // ValueSource::CountableRange 0..n_values
    const FastMod fm_n = make_fastmod(n), fm_vc = make_fastmod(vc);  // fixed divisors of the two streams

    for (uint32_t t = lane; t < n; t += 64) s_vals[t] = (VT)g_vals[t];
    if (tables)
        for (uint32_t v = lane; v < (uint32_t)m.n_values; v += 64) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
    wave_sync();
    if (tables) {
        scalar_tables_accumulate(m, s_vals, lane, 64u, t_cnt, t_sum);
        wave_sync();
    }
    int64_t cur[L], best_sol[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        cur[k] = g_score[k];
        best_sol[k] = m.best_score[(size_t)r * 4 + k];
    }
    // per-launch counters in 32 bits (wave-uniform), folded into the 64-bit sf_stats words before they can wrap
    uint32_t st_steps = 0, st_gen = 0, st_acc = 0, st_applied = 0, st_calc = 0, st_scored = 0, st_sources = 0;
    uint64_t steps_run = 0;
    auto flush_stats = [&]() {
        if (lane == 0) {
            uint64_t* gs = p.stats + (size_t)r * SF_STATS_WORDS;
            gs[0] += st_steps;
            gs[1] += st_gen;
            gs[2] += st_gen;
            gs[3] += st_acc;
            gs[4] += st_applied;
            gs[5] += st_calc;
            gs[6] += st_gen - st_calc;
            gs[7] += st_scored;
            gs[8] += st_sources;
        }
        steps_run += st_steps;
        st_steps = st_gen = st_acc = st_applied = st_calc = st_scored = st_sources = 0;
    };
    uint64_t trace_n = 0;
    const uint64_t step_index0 = p.dry_run ? 0 : p.step_index[r];
    const uint64_t seed_draws0 = p.dry_run ? 0 : p.seed_draws[r];
    const int la_idx0 = p.dry_run ? 0 : p.la_idx[r];
    int la_cursor = la_idx0;  // (la_idx0 + step) % la_size, kept incrementally (no 64-bit division per step)
    // pulls per accepted candidate of the previous step: how far ahead of the forager's quota it pays to
    // generate and trial-score (scheduling only: the consumed prefix, hence every result, is unchanged)
    uint32_t prev_pulls = 64, prev_accepted = 1;
    // update_best_solution clones the working solution on every strict improvement (scope_progress.rs:89-107).  While
    // the search keeps improving, the working state IS the best state, so the clone is deferred: `best_pending` marks
    // "working == best, not yet written"; the snapshot is written right before a non-improving move leaves that state
    // (and at the end of the launch).  HBM sees one write per departure from a best state instead of one per improvement.
    bool best_pending = false;
    PH_DECL

    for (int64_t step = 0; step < p.n_steps; ++step) {
        // load-balance aggregates of the step snapshot (the tables change only at commit): every lane gets the totals
        int64_t lbv[4] = {0, 0, 0, 0};
        if (m.grp_level >= 0 && m.grp_mode >= 1) {
            int64_t s1 = 0, s2 = 0;
            uint32_t nk = 0;
            for (uint32_t v = lane; v < (uint32_t)m.n_values; v += 64)
                if (t_cnt[v]) {
                    const int64_t x = m.grp_mode == 2 ? (int64_t)t_cnt[v] : t_sum[v];
                    s1 = wadd(s1, x);
                    s2 = wadd(s2, (int64_t)((uint64_t)x * (uint64_t)x));
                    nk += 1;
                }
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                s1 = wadd(s1, (int64_t)shfl_u64((uint64_t)s1, (int)(lane ^ (uint32_t)o)));
                s2 = wadd(s2, (int64_t)shfl_u64((uint64_t)s2, (int)(lane ^ (uint32_t)o)));
                nk += (uint32_t)__shfl((int)nk, (int)(lane ^ (uint32_t)o));
            }
            lbv[0] = s1, lbv[1] = s2, lbv[2] = (int64_t)nk, lbv[3] = global_stat(m, s1, s2, nk);
        }
        uint64_t sidx, sseed;
        if (p.dry_run) {
            sidx = p.dry_step_index;
            sseed = p.dry_step_seed;
        } else {
            sidx = step_index0 + (uint64_t)step;
            const uint64_t draw = seed_draws0 + (uint64_t)step;
            if (p.explicit_seeds && (int64_t)draw < p.n_explicit)
                sseed = p.explicit_seeds[(size_t)r * p.n_explicit + draw];
            else
                sseed = step_seed(p.random_seed + (uint64_t)r, draw);
        }
        sidx = uni64(sidx);
        sseed = uni64(sseed);
        const StreamCtx ctx{sidx, sseed, p.order};
        ScoreV<L> late;
#pragma unroll
        for (int k = 0; k < L; ++k) late.v[k] = 0;
        const int la_slot = la_cursor;  // LateAcceptance history slot of this step
        if (p.acceptor == 1 || p.acceptor == 4) {
#pragma unroll
            for (int k = 0; k < L; ++k) late.v[k] = p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + k];
        }
        ScoreV<L> dla_thr = late;  // DiversifiedLateAcceptance: best step score of the phase minus its tolerance band
        if (p.acceptor == 4) {
            ScoreV<L> db;
#pragma unroll
            for (int k = 0; k < L; ++k) db.v[k] = (int64_t)uni64((uint64_t)p.dla_best[(size_t)r * 4 + k]);
            dla_thr = dla_threshold<L>(db, p.dla_tolerance);
        }
        int has_best = 0;
        uint64_t equal_count = 0;
        uint32_t accepted = 0, pulls = 0;
        ScoreV<L> best;
#pragma unroll
        for (int k = 0; k < L; ++k) best.v[k] = 0;
        uint32_t best_m0 = 0, best_m1 = 0;
        int best_leaf = 0;
        uint64_t best_ti = 0;  // trace ordinal (within the step) of the forager's current pick
        const uint32_t first_leaf = n_leaves > 1 ? ctx.random_index((uint32_t)n_leaves, SALT_UNION_OFFSET) : 0u;

        // entity permutations (selection_index_without_replacement)
        uint32_t cst = 0, csd = 1, lst = 0, lsd = 1;
        ctx.perm_params(n, SALT_SCALAR_CHANGE_ENTITY ^ identity, cst, csd);
        ctx.perm_params(n, (SALT_SCALAR_SWAP_LEFT ^ identity) ^ OFFSET_MIX, lst, lsd);
        cst = uni(cst), csd = uni(csd), lst = uni(lst), lsd = uni(lsd);

        // per-leaf cursor: (row, inner) = change: (entity offset, value offset incl. the to-None slot);
        // swap: (left offset, right offset)
        uint32_t head[2] = {0, 0}, tail[2] = {0, 0}, row[2] = {0, 0}, inner[2] = {0, 0};
        int gen_done[2] = {n == 0, n_leaves > 1 ? (n == 0) : 1}, ex[2] = {0, n_leaves > 1 ? 0 : 1};

        PH(0)
        int done = 0;
        while (!done) {
            // C1: fill the rings: as many pending candidates as the forager is expected to consume before it
            // quits (quota left x pulls per accept of the last step x 1.5), 8..64, split over the live leaves
            uint32_t spec = 64;
            if (p.forager <= FORAGER_FIRST_ACCEPTED) {
                const uint32_t want = p.forager == 0 ? (uint32_t)p.limit - accepted : 1u;
                const uint64_t est = ((uint64_t)want * prev_pulls * 3u) / (2u * prev_accepted);
                spec = est >= 64 ? 64u : est < 8 ? 8u : (uint32_t)est;
            }
            const uint32_t target = (!ex[0] && !ex[1]) ? (spec + 1) / 2 : spec;
            // A forager that ends the step at its first accepted candidate (AcceptedCount(1): the reference's default for scalar-only models,
            // 1.2 consumed candidates per step under SimulatedAnnealing; FirstAccepted) usually never pulls the second child of the union: the
            // first round fills only the ring of the child the scheduler pulls first, the other one when the replay gets to it.  Same
            // candidates in the same order; four of five steps skip one fill (the swap stream's is four hashed 64-wide chunks).
            const bool one_accept = p.forager == FORAGER_FIRST_ACCEPTED || (p.forager == 0 && (uint32_t)p.limit - accepted == 1u);
            const bool lazy_first = one_accept && n_leaves > 1 && pulls == 0 && !ex[0] && !ex[1];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                if (l >= n_leaves) continue;
                if (lazy_first && (uint32_t)l != first_leaf) continue;
                const bool is_change = l ? chg1 : chg0;
                uint32_t* rq = ring + (size_t)l * SRC * 2;
                while (!ex[l] && !gen_done[l] && tail[l] - head[l] < target) {
                    st_sources += 1;
                    if (is_change) {
                        // 64 consecutive (row, value-offset) slots of the change stream, one per lane
                        const uint32_t e_row = fastmod_u64((uint64_t)cst + (uint64_t)row[l] * csd, fm_n);
                        const bool has_none = m.allows_unassigned && (int32_t)s_vals[e_row] >= 0;
                        const uint32_t per0 = value_count(m, e_row) + (has_none ? 1u : 0u);  // candidates of the current row
                        // lanes walk rows starting at (row, inner): rows have vc or vc+1 candidates
                        uint32_t my_row = row[l], my_in = inner[l] + lane;
                        uint32_t per = per0;
                        bool valid = true;
                        for (;;) {  // skip whole rows (per-lane loop; <= 64/vc + 1 iterations)
                            if (my_row >= n) {
                                valid = false;
                                break;
                            }
                            if (my_in < per) break;
                            my_in -= per;
                            ++my_row;
                            if (my_row < n) {
                                const uint32_t e2 = fastmod_u64((uint64_t)cst + (uint64_t)my_row * csd, fm_n);
                                per = value_count(m, e2) + ((m.allows_unassigned && (int32_t)s_vals[e2] >= 0) ? 1u : 0u);
                            }
                        }
                        uint32_t e = 0;
                        int32_t v = -1;
                        if (valid) {
                            e = fastmod_u64((uint64_t)cst + (uint64_t)my_row * csd, fm_n);
                            if (my_in < value_count(m, e)) v = value_at(m, ctx, e, my_in, SALT_SCALAR_CHANGE_VALUE ^ (uint64_t)e ^ identity, fm_vc);
                        }
                        const uint64_t vm = __ballot(valid);
                        const uint32_t cnt = (uint32_t)__popcll(vm);  // valid lanes are a prefix
                        if (valid) {
                            const uint32_t qi = (tail[l] + lane) & (SRC - 1);
                            rq[qi * 2] = e;
                            rq[qi * 2 + 1] = (uint32_t)v;
                        }
                        tail[l] += cnt;
                        // advance the cursor past the last valid lane
                        const uint32_t last = cnt ? cnt - 1 : 0;
                        const uint32_t lr = uni(__shfl(my_row, (int)last)), li = uni(__shfl(my_in, (int)last));
                        if (cnt == 0) {
                            gen_done[l] = 1;
                        } else {
                            row[l] = lr;
                            inner[l] = li + 1;  // may equal the row's count: the next walk skips the row
                            if (cnt < 64) gen_done[l] = 1;
                        }
                    } else {
                        // 4 x 64 consecutive right offsets of the current left row, filtered
                        // (swap.rs:125-160).  The four chunks are independent draws (ILP hides the
                        // splitmix / remainder / LDS latency); they are appended in stream order and a
                        // chunk that would overfill the ring is left for the next call.
                        if (row[l] >= n) {
                            gen_done[l] = 1;
                            break;
                        }
                        const uint32_t left = n <= 1 ? 0u : fastmod_u64((uint64_t)lst + (uint64_t)row[l] * lsd, fm_n);
                        const int32_t lv = (int32_t)s_vals[left];
                        const uint64_t rsalt = (SALT_SCALAR_SWAP_RIGHT ^ (uint64_t)left ^ (uint64_t)(uint32_t)m.variable) ^ OFFSET_MIX;
                        bool keep[4];
                        uint32_t right[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t ro = inner[l] + 64u * q + lane;
                            keep[q] = false;
                            right[q] = 0;
                            if (ro < n) {
                                right[q] = n <= 1 ? 0u : ctx.selection_index_fm(ro, fm_n, rsalt);
                                if (left < right[q]) {
                                    const int32_t rv = (int32_t)s_vals[right[q]];
                                    keep[q] = lv != rv && value_legal(m, right[q], lv) && value_legal(m, left, rv);
                                }
                            }
                        }
                        uint32_t chunks_done = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint64_t km = __ballot(keep[q]);
                            const uint32_t cq = (uint32_t)__popcll(km);
                            const bool fits = chunks_done == (uint32_t)q && (q == 0 || tail[l] - head[l] + cq <= SRC) &&
                                              inner[l] + 64u * q < n;
                            if (fits) {
                                if (keep[q]) {
                                    const uint32_t qi = (tail[l] + mbcnt64(km)) & (SRC - 1);
                                    rq[qi * 2] = left;
                                    rq[qi * 2 + 1] = right[q];
                                }
                                tail[l] += cq;
                                chunks_done = q + 1;
                            }
                        }
                        inner[l] += 64u * chunks_done;
                        if (inner[l] >= n) {
                            inner[l] = 0;
                            row[l] += 1;
                            if (row[l] >= n) gen_done[l] = 1;
                        }
                    }
                }
            }
            wave_sync();

            PH(1)
            // C2: replay one batch in union cursor order
            {
                const bool live0 = !ex[0], live1 = !ex[1];
                if (!live0 && !live1) {
                    done = 1;
                    break;
                }
                uint32_t lf, idx;
                if (live0 && live1) {
                    const uint32_t l0 = (first_leaf + pulls) & 1u;
                    lf = (l0 + lane) & 1u;
                    idx = (lf ? head[1] : head[0]) + (lane >> 1);
                } else {
                    lf = live0 ? 0u : 1u;
                    idx = (lf ? head[1] : head[0]) + lane;
                }
                const bool avail = (int32_t)((lf ? tail[1] : tail[0]) - idx) > 0;
                const uint64_t availmask = __ballot(avail);
                const uint32_t nvalid = availmask == ~0ULL ? 64u : (uint32_t)(__ffsll((unsigned long long)~availmask) - 1);
                if (nvalid == 0) {
                    const uint32_t lf0 = uni(__shfl(lf, 0));
                    if (lf0 ? gen_done[1] : gen_done[0]) {
                        if (lf0)
                            ex[1] = 1;
                        else
                            ex[0] = 1;
                    }
                    continue;  // not exhausted: the fill above continues the leaf
                }
                const bool valid = lane < nvalid;
                uint32_t m0 = 0, m1 = 0;
                ScalarDelta dl{0, 0, 0, 0, false};
                const bool lane_change = lf ? chg1 : chg0;
                if (valid) {
                    const uint32_t qi = idx & (SRC - 1);
                    const uint32_t* rq = ring + ((size_t)lf * SRC + qi) * 2;
                    m0 = rq[0];
                    m1 = rq[1];
                    dl = lane_change ? eval_scalar_move(m, s_vals, 0, m0, 0u, (int32_t)m1, t_cnt, t_sum, lbv)
                                     : eval_scalar_move(m, s_vals, 1, m0, m1, 0, t_cnt, t_sum, lbv);
                }
                const ScoreV<L> sc = apply_scalar_delta<L>(m, cur, dl);
                ScoreV<L> curv;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) curv.v[kk] = cur[kk];
                const bool doable = valid && dl.doable;
                bool acc = false;
                if (doable) {
                    if (p.acceptor == 0)
                        acc = score_cmp<L>(sc, curv) > 0;
                    else if (p.acceptor == 1)
                        acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0;
                    else if (p.acceptor == 4)
                        acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0 || score_cmp<L>(sc, dla_thr) >= 0;
                }
                SaChunk sach;
                if (annealing) acc = sa_decide<L>(saw, p.sa, doable, sc, curv, lane, sach);
                uint64_t accmask = __ballot(acc);
                bool improving_pick = false;
                ScoreV<L> forager_thr = curv;  // FirstLastStepScoreImproving: the last step score
                if (p.forager == FORAGER_FIRST_BEST_IMPROVING) {  // the best score ever seen (step.rs:53-58)
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) forager_thr.v[kk] = best_sol[kk];
                }
                const uint32_t nconsumed = forager_chunk_cut<L>(p.forager, (uint32_t)p.limit, accepted, acc, sc, forager_thr, nvalid, improving_pick);
                const bool consumed = lane < nconsumed;
                if (annealing) sa_commit<L>(saw, p.sa, sach, nconsumed, lane);
                acc = acc && consumed;
                accmask = __ballot(acc);
                if (accmask) {
                    if (improving_pick) {  // BestCandidate::replace by the candidate that ends the step (improving.rs:92-95,205-208)
                        const int sel = (int)nconsumed - 1;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)uni64(shfl_u64((uint64_t)sc.v[kk], sel));
                        best_m0 = __shfl(m0, sel);
                        best_m1 = __shfl(m1, sel);
                        best_leaf = (int)__shfl(lf, sel);
                        if (TRACE) best_ti = trace_n + (uint64_t)sel;
                        equal_count = 1;
                        has_best = 1;
                    } else if (p.forager == 1) {
                        if (!has_best) {
                            const int sel = __ffsll((unsigned long long)accmask) - 1;
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)shfl_u64((uint64_t)sc.v[kk], sel);
                            best_m0 = __shfl(m0, sel);
                            best_m1 = __shfl(m1, sel);
                            best_leaf = (int)__shfl(lf, sel);
                            if (TRACE) best_ti = trace_n + (uint64_t)sel;
                            has_best = 1;
                        }
                    } else if (!has_best || __ballot(acc && score_cmp<L>(sc, best) >= 0)) {
                        const ScoreV<L> M = wave_max_score<L>(sc, acc);
                        const int cm = has_best ? score_cmp<L>(M, best) : 1;
                        if (cm >= 0) {
                            const bool newmax = cm > 0;
                            const uint64_t eq_base = newmax ? 0 : equal_count;
                            const bool in_eq = acc && score_cmp<L>(sc, M) == 0;
                            const uint64_t eq = __ballot(in_eq);
                            const uint32_t rank = mbcnt64(eq) + 1u;
                            const uint64_t cntq = eq_base + rank;
                            const bool pick = in_eq && ((newmax && rank == 1) ||
                                                        (p.random_ties && cntq > 1 && reservoir_pick(sseed, cntq)));
                            const uint64_t pm = __ballot(pick);
                            if (pm) {
                                const int sel = 63 - __clzll((unsigned long long)pm);
                                best_m0 = __shfl(m0, sel);
                                best_m1 = __shfl(m1, sel);
                                best_leaf = (int)__shfl(lf, sel);
                                if (TRACE) best_ti = trace_n + (uint64_t)sel;
                            }
                            best = M;
                            equal_count = eq_base + (uint64_t)__popcll(eq);
                            has_best = 1;
                        }
                    }
                }
                const uint32_t nacc = (uint32_t)__popcll(accmask);
                accepted += nacc;
                st_gen += nconsumed;
                st_acc += nacc;
                st_scored += nvalid;
                st_calc += (uint32_t)__popcll(__ballot(consumed && doable));
                if (tracing && consumed) {
                    const uint64_t ti = trace_n + lane;
                    if ((int64_t)ti < p.trace_cap) {
                        int32_t* tm = p.trace_moves + ti * 6;
                        tm[0] = lane_change ? 0 : 1;
                        tm[1] = (int32_t)m0;
                        tm[2] = 0;
                        tm[3] = lane_change ? 0 : (int32_t)m1;
                        tm[4] = 0;
                        tm[5] = lane_change ? (int32_t)m1 : -1;
                        for (int kk = 0; kk < L && kk < m.levels; ++kk) p.trace_scores[ti * m.levels + kk] = doable ? sc.v[kk] : 0;
                        p.trace_flags[ti] = (doable ? 1 : 0) | (acc ? 2 : 0) | ((int32_t)lf << 8);
                    }
                }
                if (tracing) trace_n += nconsumed;
                const uint32_t c1 = (uint32_t)__popcll(__ballot(consumed && lf == 1u));
                head[1] += c1;
                head[0] += nconsumed - c1;
                pulls += nconsumed;
                if (forager_quits(p.forager, (uint32_t)p.limit, accepted, has_best, improving_pick)) done = 1;
            }
        }

        PH(2)
        prev_pulls = pulls ? pulls : 1u;
        prev_accepted = accepted ? accepted : 1u;

        // ---- commit the forager's pick ----
        const bool applied = has_best && !p.dry_run;
        if (applied) {
            if (best_pending) {
                ScoreV<L> bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) bs.v[kk] = best_sol[kk];
                if (!(score_cmp<L>(best, bs) > 0)) {  // the pick does not improve on the best: snapshot before leaving it
                    for (uint32_t t = lane; t < n; t += 64) m.best_vals[(size_t)r * n + t] = (int32_t)s_vals[t];
                    best_pending = false;
                }
            }
            const bool pick_change = best_leaf ? chg1 : chg0;
            const uint32_t a = uni(best_m0), b = uni(best_m1);
            if (tracing && lane == 0) {
                p.trace_applied[0] = 1;
                if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                p.trace_applied[1] = pick_change ? 0 : 1;
                p.trace_applied[2] = (int32_t)a;
                p.trace_applied[3] = 0;
                p.trace_applied[4] = pick_change ? 0 : (int32_t)b;
                p.trace_applied[5] = 0;
                p.trace_applied[6] = pick_change ? (int32_t)b : -1;
            }
            if (lane == 0) {
                if (tables) scalar_tables_apply(m, s_vals, pick_change ? 0 : 1, a, b, (int32_t)b, t_cnt, t_sum);
                if (pick_change)
                    s_vals[a] = (VT)(int32_t)b;
                else {
                    const VT t = s_vals[a];
                    s_vals[a] = s_vals[b];
                    s_vals[b] = t;
                }
            }
            wave_sync();
#pragma unroll
            for (int kk = 0; kk < L; ++kk) cur[kk] = best.v[kk];
            st_applied += 1;
        } else if (tracing && lane == 0) {
            p.trace_applied[0] = 0;
        }
        if (!p.dry_run) {
            bool improved = false;
            if (applied) {
                ScoreV<L> cs, bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    cs.v[kk] = cur[kk];
                    bs.v[kk] = best_sol[kk];
                }
                improved = score_cmp<L>(cs, bs) > 0;
            }
            if (improved) {  // update_best_solution (scope_progress.rs:89-107): the clone is deferred (best_pending)
                best_pending = true;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) best_sol[kk] = cur[kk];
            }
            if ((p.acceptor == 1 || p.acceptor == 4) && lane == 0) {
#pragma unroll
                for (int kk = 0; kk < L; ++kk) p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + kk] = cur[kk];
            }
            if (p.acceptor == 4 && lane == 0) {  // step_ended: the phase's best step score (diversified_late_acceptance.rs:161-170)
                ScoreV<L> cs, db;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    cs.v[kk] = cur[kk];
                    db.v[kk] = p.dla_best[(size_t)r * 4 + kk];
                }
                if (score_cmp<L>(cs, db) > 0) {
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) p.dla_best[(size_t)r * 4 + kk] = cur[kk];
                }
            }
            if (annealing) sa_step_ended(saw, p.sa, lane);
            wave_sync();
            st_steps += 1;
            la_cursor = la_cursor + 1 >= p.la_size ? 0 : la_cursor + 1;
            if (p.move_budget > 0 && (int64_t)st_gen >= p.move_budget) break;  // work-balanced launch: see sf_solve_moves
            if (!p.dry_run && p.move_budget == 0 && st_scored >= 0x70000000u) flush_stats();
        }
        PH(3)
    }
    PH_DUMP

    if (!p.dry_run) {
        if (annealing) sa_store(saw, p.sa, r, lane);
        if (best_pending)
            for (uint32_t t = lane; t < n; t += 64) m.best_vals[(size_t)r * n + t] = (int32_t)s_vals[t];
        for (uint32_t t = lane; t < n; t += 64) g_vals[t] = (int32_t)s_vals[t];
        if (lane == 0) {
#pragma unroll
            for (int kk = 0; kk < L; ++kk) {
                g_score[kk] = cur[kk];
                p.last_step_score[(size_t)r * 4 + kk] = cur[kk];
                m.best_score[(size_t)r * 4 + kk] = best_sol[kk];
            }
            p.la_idx[r] = la_cursor;
            p.step_index[r] = step_index0 + steps_run + (uint64_t)st_steps;  // steps actually run (a move budget can end the launch early)
            p.seed_draws[r] = seed_draws0 + steps_run + (uint64_t)st_steps;
        }
        flush_stats();
    }
    if (tracing && lane == 0) *p.trace_count = (int64_t)trace_n;
}

}  // namespace sf
