// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the list-variable hot path:
// seeded nearby candidate generation, per-candidate delta scoring, acceptor/forager replay
// and move application, fused into ONE persistent kernel per launch (one workgroup = one
// search replica whose routes, loads and candidate rings live in LDS for the whole launch).
//
// Reference semantics restated (paths under crates/solverforge-solver/src/ unless noted):
//   heuristic/selector/list_kernel/nearby_change.rs:102-216   destination enumeration + top-k
//   heuristic/selector/list_kernel/nearby_swap.rs:104-216
//   heuristic/selector/nearby_list_support.rs:3-34            stable bounded top-k
//   runtime/compiler/executor/list_leaf/cursor/slot.rs:468-499 entity order w/o replacement
//   heuristic/selector/decorator/vec_union.rs:334-362          StratifiedRandom union (2 leaves)
//   phase/localsearch/phase/step.rs:30-225, phase/candidates.rs:47-285, evaluation.rs:20-115
//   phase/localsearch/forager.rs:70-250, acceptor/{hill_climbing,late_acceptance}.rs
//   heuristic/move/list_kernel/{change,swap}.rs               move semantics
//   crates/solverforge-cvrp/src/{problem_data,meters}.rs      distance_cost / MatrixDistanceMeter
//
// GPU formulation (not a translation): every candidate of a step is scored against the same
// immutable step snapshot, so the reference's do/score/undo (4x retract+insert per entity per
// constraint) becomes a stateless O(1) integer delta per candidate; integer addition is
// associative, so cur + delta == the reference's incremental score bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_list_model.h"

namespace sf {

// ---------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_le(uint32_t lane) {
    return lane == 63 ? ~0ULL : ((1ULL << (lane + 1)) - 1ULL);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int delta) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, delta);
    hi = __shfl_up(hi, delta);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((uint64_t)hi << 32) | lo;
}
// wave64 inclusive prefix sum on the DPP network (row shifts + row broadcasts): no LDS crossbar
// round trips.  Must be called with all 64 lanes active.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// ProblemData::distance_cost (problem_data.rs:28-31,44-47)
__device__ __forceinline__ int64_t dist_cost(const int64_t* __restrict__ mat, int32_t dim, uint32_t from,
                                             uint32_t to) {
    int64_t v = mat[(size_t)from * (size_t)dim + to];
    return (v >= 0 && v != UNREACHABLE) ? v : MAX_SAFE_LEG_COST;
}
__device__ __forceinline__ int64_t over_cap(int64_t load, int64_t cap) {
    int64_t o = wsub(load, cap);
    return o > 0 ? o : 0;
}

// Trial score deltas of one list move against the step snapshot.  d_cap / d_dist are the
// changes of the (positive) penalty sums of the capacity and distance constraints.
struct ListDelta {
    int64_t d_cap;
    int64_t d_dist;
    bool doable;
};

// ListChangeMove (a,i) -> (b,j), j in pre-removal coordinates (move/list_kernel/change.rs:28-34).
template <class VT>
__device__ __forceinline__ ListDelta eval_list_change(const ListModel& m, const VT* visits,
                                                      const uint32_t* off, const int64_t* load, uint32_t a,
                                                      uint32_t i, uint32_t b, uint32_t j) {
    ListDelta r{0, 0, true};
    uint32_t oa = off[a], la = off[a + 1] - oa;
    if (i >= la) {
        r.doable = false;
        return r;
    }
    uint32_t x = visits[oa + i];
    bool intra = a == b;
    uint32_t ob = off[b], lb = off[b + 1] - ob;
    if (j > (intra ? la : lb) || (intra && (j == i || j == i + 1))) {
        r.doable = false;
        return r;
    }
    const uint32_t depot = (uint32_t)m.depot;
    if (m.dist_level >= 0) {
        uint32_t pa = i > 0 ? visits[oa + i - 1] : depot;
        uint32_t na = i + 1 < la ? visits[oa + i + 1] : depot;
        int64_t rem;
        if (la == 1)
            rem = -wadd(dist_cost(m.mat, m.dim, depot, x), dist_cost(m.mat, m.dim, x, depot));
        else
            rem = wsub(dist_cost(m.mat, m.dim, pa, na),
                       wadd(dist_cost(m.mat, m.dim, pa, x), dist_cost(m.mat, m.dim, x, na)));
        uint32_t pl, nr;
        bool dst_empty;
        if (!intra) {
            pl = j > 0 ? visits[ob + j - 1] : depot;
            nr = j < lb ? visits[ob + j] : depot;
            dst_empty = lb == 0;
        } else {
            // sequence after removal s'[t] = t<i ? s[t] : s[t+1]; insert at jj
            uint32_t jj = j > i ? j - 1 : j;
            uint32_t l2 = la - 1;
            auto at = [&](uint32_t t) { return (uint32_t)visits[oa + (t < i ? t : t + 1)]; };
            pl = jj > 0 ? at(jj - 1) : depot;
            nr = jj < l2 ? at(jj) : depot;
            dst_empty = false;
        }
        int64_t ins = wadd(dist_cost(m.mat, m.dim, pl, x), dist_cost(m.mat, m.dim, x, nr));
        if (!dst_empty) ins = wsub(ins, dist_cost(m.mat, m.dim, pl, nr));
        r.d_dist = wadd(rem, ins);
    }
    if (m.cap_level >= 0 && !intra) {
        int64_t dx = (int64_t)m.demand[x];
        int64_t la0 = load[a], lb0 = load[b];
        int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        int64_t after = wadd(over_cap(wsub(la0, dx), m.capacity), over_cap(wadd(lb0, dx), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// ListSwapMove (a,i) <-> (b,j) (move/list_kernel/swap.rs:30-110).
template <class VT>
__device__ __forceinline__ ListDelta eval_list_swap(const ListModel& m, const VT* visits,
                                                    const uint32_t* off, const int64_t* load, uint32_t a,
                                                    uint32_t i, uint32_t b, uint32_t j) {
    ListDelta r{0, 0, true};
    uint32_t oa = off[a], la = off[a + 1] - oa;
    uint32_t ob = off[b], lb = off[b + 1] - ob;
    if (i >= la || j >= lb || (a == b && i == j)) {
        r.doable = false;
        return r;
    }
    uint32_t x = visits[oa + i], y = visits[ob + j];
    if (x == y) {
        r.doable = false;
        return r;
    }
    const uint32_t depot = (uint32_t)m.depot;
    if (m.dist_level >= 0) {
        if (a == b) {
            if (i > j) {  // normalise i < j (x stays the element at the lower position)
                uint32_t t = i;
                i = j;
                j = t;
                t = x;
                x = y;
                y = t;
            }
            uint32_t pa = i > 0 ? visits[oa + i - 1] : depot;
            uint32_t nb = j + 1 < la ? visits[oa + j + 1] : depot;
            if (j == i + 1) {
                int64_t after = wadd(wadd(dist_cost(m.mat, m.dim, pa, y), dist_cost(m.mat, m.dim, y, x)),
                                     dist_cost(m.mat, m.dim, x, nb));
                int64_t before = wadd(wadd(dist_cost(m.mat, m.dim, pa, x), dist_cost(m.mat, m.dim, x, y)),
                                      dist_cost(m.mat, m.dim, y, nb));
                r.d_dist = wsub(after, before);
            } else {
                uint32_t na = visits[oa + i + 1];
                uint32_t pb = visits[oa + j - 1];
                int64_t da = wsub(wadd(dist_cost(m.mat, m.dim, pa, y), dist_cost(m.mat, m.dim, y, na)),
                                  wadd(dist_cost(m.mat, m.dim, pa, x), dist_cost(m.mat, m.dim, x, na)));
                int64_t db = wsub(wadd(dist_cost(m.mat, m.dim, pb, x), dist_cost(m.mat, m.dim, x, nb)),
                                  wadd(dist_cost(m.mat, m.dim, pb, y), dist_cost(m.mat, m.dim, y, nb)));
                r.d_dist = wadd(da, db);
            }
        } else {
            uint32_t pa = i > 0 ? visits[oa + i - 1] : depot;
            uint32_t na = i + 1 < la ? visits[oa + i + 1] : depot;
            uint32_t pb = j > 0 ? visits[ob + j - 1] : depot;
            uint32_t nb = j + 1 < lb ? visits[ob + j + 1] : depot;
            int64_t da = wsub(wadd(dist_cost(m.mat, m.dim, pa, y), dist_cost(m.mat, m.dim, y, na)),
                              wadd(dist_cost(m.mat, m.dim, pa, x), dist_cost(m.mat, m.dim, x, na)));
            int64_t db = wsub(wadd(dist_cost(m.mat, m.dim, pb, x), dist_cost(m.mat, m.dim, x, nb)),
                              wadd(dist_cost(m.mat, m.dim, pb, y), dist_cost(m.mat, m.dim, y, nb)));
            r.d_dist = wadd(da, db);
        }
    }
    if (m.cap_level >= 0 && a != b) {
        int64_t dx = (int64_t)m.demand[x], dy = (int64_t)m.demand[y];
        int64_t la0 = load[a], lb0 = load[b];
        int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        int64_t after = wadd(over_cap(wadd(wsub(la0, dx), dy), m.capacity),
                             over_cap(wadd(wsub(lb0, dy), dx), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// Branch-light trial delta of a ListChange (is_change) or ListSwap move: the move is reduced to at
// most eight signed matrix legs whose gathers are all issued together (one memory round trip per
// 64-candidate batch, no divergence between change and swap lanes).  Same results as
// eval_list_change / eval_list_swap (wrapping i64 sums are order-independent).
template <class VT, bool M32, class LT = int64_t>
__device__ __forceinline__ ListDelta eval_list_move_legs(const ListModel& m, const VT* visits, const uint32_t* off,
                                                         const LT* load, bool is_change, uint32_t a,
                                                         uint32_t i, uint32_t b, uint32_t j) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    const uint32_t ob = off[b], lb = off[b + 1] - ob;
    const bool intra = a == b;
    if (is_change) {  // move/list_kernel/change.rs:44-71
        if (i >= la || j > lb || (intra && (j == i || j == i + 1))) return r;
    } else {  // move/list_kernel/swap.rs:30-56
        if (i >= la || j >= lb || (intra && i == j)) return r;
    }
    const uint32_t depot = (uint32_t)m.depot;
    if (!is_change && intra && i > j) {  // normalise i < j for the adjacent case (symmetric otherwise)
        const uint32_t t = i;
        i = j;
        j = t;
    }
    const uint32_t P = oa + i, Q = ob + j;
    const uint32_t x = visits[P];
    const uint32_t pa = i > 0 ? (uint32_t)visits[P - 1] : depot;
    const uint32_t na = i + 1 < la ? (uint32_t)visits[P + 1] : depot;
    const uint32_t vq = j < lb ? (uint32_t)visits[Q] : depot;  // change: right neighbour of the slot; swap: y
    const uint32_t pb = j > 0 ? (uint32_t)visits[Q - 1] : depot;
    const uint32_t nb = j + 1 < lb ? (uint32_t)visits[Q + 1] : depot;
    if (!is_change && x == vq) return r;
    r.doable = true;
    uint32_t f[8], t[8];
    int32_t sg[8];
    if (is_change) {
        // remove x from (pa, x, na); insert it into the slot (pb, vq) — pre-removal coordinates reduce
        // to the same neighbours for intra moves (j != i, i+1)
        const bool src_single = la == 1;
        const bool dst_empty = !intra && lb == 0;
        f[0] = pa, t[0] = x, sg[0] = -1;
        f[1] = x, t[1] = na, sg[1] = -1;
        f[2] = pa, t[2] = na, sg[2] = src_single ? 0 : 1;
        f[3] = pb, t[3] = x, sg[3] = 1;
        f[4] = x, t[4] = vq, sg[4] = 1;
        f[5] = pb, t[5] = vq, sg[5] = dst_empty ? 0 : -1;
        f[6] = 0, t[6] = 0, sg[6] = 0;
        f[7] = 0, t[7] = 0, sg[7] = 0;
    } else {
        const uint32_t y = vq;
        if (intra && j == i + 1) {
            f[0] = pa, t[0] = y, sg[0] = 1;
            f[1] = y, t[1] = x, sg[1] = 1;
            f[2] = x, t[2] = nb, sg[2] = 1;
            f[3] = pa, t[3] = x, sg[3] = -1;
            f[4] = x, t[4] = y, sg[4] = -1;
            f[5] = y, t[5] = nb, sg[5] = -1;
            f[6] = 0, t[6] = 0, sg[6] = 0;
            f[7] = 0, t[7] = 0, sg[7] = 0;
        } else {
            f[0] = pa, t[0] = y, sg[0] = 1;
            f[1] = y, t[1] = na, sg[1] = 1;
            f[2] = pa, t[2] = x, sg[2] = -1;
            f[3] = x, t[3] = na, sg[3] = -1;
            f[4] = pb, t[4] = x, sg[4] = 1;
            f[5] = x, t[5] = nb, sg[5] = 1;
            f[6] = pb, t[6] = y, sg[6] = -1;
            f[7] = y, t[7] = nb, sg[7] = -1;
        }
    }
    if (m.dist_level >= 0) {
        int64_t acc = 0;
        if (M32) {  // compact matrix copy: 4-byte gathers
            uint32_t v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = m.mat32[f[s] * (uint32_t)m.dim + t[s]];  // dim <= 16384: 32-bit offsets
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t c = v[s] != 0xFFFFFFFFu ? (int64_t)v[s] : MAX_SAFE_LEG_COST;
                acc = wadd(acc, (int64_t)((uint64_t)c * (uint64_t)(int64_t)sg[s]));
            }
        } else {
            int64_t v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = m.mat[(size_t)f[s] * (size_t)m.dim + t[s]];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t c = (v[s] >= 0 && v[s] != UNREACHABLE) ? v[s] : MAX_SAFE_LEG_COST;
                acc = wadd(acc, (int64_t)((uint64_t)c * (uint64_t)(int64_t)sg[s]));
            }
        }
        r.d_dist = acc;
    }
    if (m.cap_level >= 0 && !intra) {
        const int64_t dx = (int64_t)m.demand[x];
        const int64_t dy = is_change ? 0 : (int64_t)m.demand[vq];
        const int64_t la0 = load[a], lb0 = load[b];
        const int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        const int64_t after = wadd(over_cap(wadd(wsub(la0, dx), dy), m.capacity),
                                   over_cap(wadd(wsub(lb0, dy), dx), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// ListReverseMove: reverse list `a` over [start, end) (move/list_kernel/reverse.rs:22-57).  The
// matrix may be asymmetric, so every leg inside the range changes direction.
template <class VT>
__device__ __forceinline__ ListDelta eval_list_reverse(const ListModel& m, const VT* visits, const uint32_t* off,
                                                       uint32_t a, uint32_t start, uint32_t end) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    if (!(end > start + 1 && end <= la)) return r;
    r.doable = true;
    if (m.dist_level >= 0) {
        const uint32_t depot = (uint32_t)m.depot;
        const uint32_t prev = start > 0 ? (uint32_t)visits[oa + start - 1] : depot;
        const uint32_t next = end < la ? (uint32_t)visits[oa + end] : depot;
        const uint32_t first = visits[oa + start], last = visits[oa + end - 1];
        int64_t acc = wsub(wadd(dist_cost(m.mat, m.dim, prev, last), dist_cost(m.mat, m.dim, first, next)),
                           wadd(dist_cost(m.mat, m.dim, prev, first), dist_cost(m.mat, m.dim, last, next)));
        uint32_t u = first;
        for (uint32_t t = start + 1; t < end; ++t) {
            const uint32_t w = visits[oa + t];
            acc = wadd(acc, wsub(dist_cost(m.mat, m.dim, w, u), dist_cost(m.mat, m.dim, u, w)));
            u = w;
        }
        r.d_dist = acc;
    }
    return r;
}

// ListPermuteMove (move/list_kernel/permute.rs:22-72): the window [start, start + size) of list `a` reordered by the rank-th
// permutation of its positions in lexicographic order (nth_permutation, selector/list_kernel/permute.rs:260-272).  One list:
// only the distance aggregate can change; the matrix may be asymmetric, so every leg of the window is re-priced.
__device__ __forceinline__ uint32_t permute_factorial(uint32_t v) {
    uint32_t f = 1;
    for (uint32_t i = 2; i <= v; ++i) f *= i;
    return f;
}
// the permutation as nibbles: nibble k = the window position whose element moves to window position k (size <= 8)
__device__ __forceinline__ uint32_t nth_permutation_nibbles(uint32_t size, uint32_t rank) {
    uint32_t remaining = 0x76543210u, perm = 0;
    for (uint32_t position = 0; position < size; ++position) {
        const uint32_t step = permute_factorial(size - position - 1u);
        const uint32_t index = rank / step;
        rank -= index * step;
        perm |= ((remaining >> (4u * index)) & 15u) << (4u * position);
        const uint32_t low = index ? (remaining & ((1u << (4u * index)) - 1u)) : 0u;
        remaining = low | ((index < 7u ? (remaining >> (4u * (index + 1u))) : 0u) << (4u * index));
    }
    return perm;
}
template <class VT>
__device__ __forceinline__ ListDelta eval_list_permute(const ListModel& m, const VT* visits, const uint32_t* off, uint32_t a, uint32_t start,
                                                       uint32_t size, uint32_t rank) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    if (size < 2 || size > 8 || start + size > la || rank < 1 || rank >= permute_factorial(size)) return r;
    r.doable = true;
    if (m.dist_level >= 0) {
        const uint32_t depot = (uint32_t)m.depot;
        const uint32_t prev = start > 0 ? (uint32_t)visits[oa + start - 1] : depot;
        const uint32_t next = start + size < la ? (uint32_t)visits[oa + start + size] : depot;
        const uint32_t perm = nth_permutation_nibbles(size, rank);
        int64_t acc = 0;
        uint32_t uo = prev, un = prev;
        for (uint32_t k = 0; k < size; ++k) {
            const uint32_t wo = visits[oa + start + k], wn = visits[oa + start + ((perm >> (4u * k)) & 15u)];
            acc = wadd(acc, wsub(dist_cost(m.mat, m.dim, un, wn), dist_cost(m.mat, m.dim, uo, wo)));
            uo = wo;
            un = wn;
        }
        r.d_dist = wadd(acc, wsub(dist_cost(m.mat, m.dim, un, next), dist_cost(m.mat, m.dim, uo, next)));
    }
    return r;
}

// Reconnection patterns of a 3-opt move (heuristic/move/k_opt_reconnection.rs:203-211,
// THREE_OPT_RECONNECTIONS): patterns 0..2 keep the order A B C D, patterns 3..6 place C before B;
// bit 1 / bit 2 of the mask reverse segment B / C (flags follow the ORIGINAL segment index,
// move/list_kernel/k_opt.rs:73-80).
__device__ __forceinline__ bool kopt_swaps_segments(uint32_t pattern) { return pattern >= 3; }
__device__ __forceinline__ uint32_t kopt_reverse_mask(uint32_t pattern) {
    return pattern < 3 ? (pattern + 1u) << 1 : (pattern - 3u) << 1;  // {2,4,6} / {0,2,4,6}
}

// KOptMove with k = 3 on one list (move/list_kernel/k_opt.rs:13-96): cuts c1 < c2 < c3 split list a into
// A = [0,c1) B = [c1,c2) C = [c2,c3) D = [c3,len).  Only the distance aggregate can change; the matrix may
// be asymmetric, so a reversed segment is re-walked in the other direction.
template <class VT>
__device__ __forceinline__ ListDelta eval_kopt(const ListModel& m, const VT* visits, const uint32_t* off, uint32_t a,
                                               uint32_t c1, uint32_t c2, uint32_t c3, uint32_t pattern) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    if (!(pattern < 7 && c1 < c2 && c2 < c3 && c3 <= la)) return r;  // k_opt_is_doable
    r.doable = true;
    if (m.dist_level >= 0) {
        const uint32_t depot = (uint32_t)m.depot;
        const uint32_t prev = c1 > 0 ? (uint32_t)visits[oa + c1 - 1] : depot;
        const uint32_t next = c3 < la ? (uint32_t)visits[oa + c3] : depot;
        const uint32_t bf = visits[oa + c1], bl = visits[oa + c2 - 1], cf = visits[oa + c2], cl = visits[oa + c3 - 1];
        int64_t fB = 0, gB = 0, fC = 0, gC = 0;  // forward / backward internal leg sums
        uint32_t u = bf;
        for (uint32_t t = c1 + 1; t < c2; ++t) {
            const uint32_t w = visits[oa + t];
            fB = wadd(fB, dist_cost(m.mat, m.dim, u, w));
            gB = wadd(gB, dist_cost(m.mat, m.dim, w, u));
            u = w;
        }
        u = cf;
        for (uint32_t t = c2 + 1; t < c3; ++t) {
            const uint32_t w = visits[oa + t];
            fC = wadd(fC, dist_cost(m.mat, m.dim, u, w));
            gC = wadd(gC, dist_cost(m.mat, m.dim, w, u));
            u = w;
        }
        const int64_t old_cost = wadd(wadd(wadd(dist_cost(m.mat, m.dim, prev, bf), fB), wadd(dist_cost(m.mat, m.dim, bl, cf), fC)),
                                      dist_cost(m.mat, m.dim, cl, next));
        const uint32_t mask = kopt_reverse_mask(pattern);
        const bool rb = (mask >> 1) & 1u, rc = (mask >> 2) & 1u, sw = kopt_swaps_segments(pattern);
        const uint32_t Bf = rb ? bl : bf, Bl = rb ? bf : bl, Cf = rc ? cl : cf, Cl = rc ? cf : cl;
        const int64_t cB = rb ? gB : fB, cC = rc ? gC : fC;
        const uint32_t Xf = sw ? Cf : Bf, Xl = sw ? Cl : Bl, Yf = sw ? Bf : Cf, Yl = sw ? Bl : Cl;
        const int64_t new_cost = wadd(wadd(wadd(dist_cost(m.mat, m.dim, prev, Xf), cB), wadd(dist_cost(m.mat, m.dim, Xl, Yf), cC)),
                                      dist_cost(m.mat, m.dim, Yl, next));
        r.d_dist = wsub(new_cost, old_cost);
    }
    return r;
}

// SublistChangeMove: segment [s, e) of list a -> list b at position dp, dp in post-removal coordinates
// when a == b (move/list_kernel/sublist_change.rs:18-130).  The segment keeps its direction.
template <class VT>
__device__ __forceinline__ ListDelta eval_sublist_change(const ListModel& m, const VT* visits, const uint32_t* off,
                                                         const int64_t* load, uint32_t a, uint32_t s, uint32_t e, uint32_t b,
                                                         uint32_t dp) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    const uint32_t ob = off[b], lb = off[b + 1] - ob;
    const bool intra = a == b;
    if (!(s < e && e <= la)) return r;
    const uint32_t z = e - s;
    if (dp > (intra ? la - z : lb) || (intra && dp == s)) return r;
    r.doable = true;
    const uint32_t depot = (uint32_t)m.depot;
    if (m.dist_level >= 0) {
        const uint32_t first = visits[oa + s], last = visits[oa + e - 1];
        const uint32_t prev = s > 0 ? (uint32_t)visits[oa + s - 1] : depot;
        const uint32_t next = e < la ? (uint32_t)visits[oa + e] : depot;
        int64_t acc = -wadd(dist_cost(m.mat, m.dim, prev, first), dist_cost(m.mat, m.dim, last, next));
        if (la > z) acc = wadd(acc, dist_cost(m.mat, m.dim, prev, next));  // the source list is not emptied
        uint32_t pl, pr;
        bool dst_empty = false;
        if (!intra) {
            pl = dp > 0 ? (uint32_t)visits[ob + dp - 1] : depot;
            pr = dp < lb ? (uint32_t)visits[ob + dp] : depot;
            dst_empty = lb == 0;
        } else {
            const uint32_t l2 = la - z;
            auto at = [&](uint32_t t) { return (uint32_t)visits[oa + (t < s ? t : t + z)]; };
            pl = dp > 0 ? at(dp - 1) : depot;
            pr = dp < l2 ? at(dp) : depot;
        }
        acc = wadd(acc, wadd(dist_cost(m.mat, m.dim, pl, first), dist_cost(m.mat, m.dim, last, pr)));
        if (!dst_empty) acc = wsub(acc, dist_cost(m.mat, m.dim, pl, pr));
        r.d_dist = acc;
    }
    if (m.cap_level >= 0 && !intra) {
        int64_t dsum = 0;
        for (uint32_t t = s; t < e; ++t) dsum = wadd(dsum, (int64_t)m.demand[visits[oa + t]]);
        const int64_t la0 = load[a], lb0 = load[b];
        const int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        const int64_t after = wadd(over_cap(wsub(la0, dsum), m.capacity), over_cap(wadd(lb0, dsum), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// SublistSwapMove: segment [fs, fe) of list a <-> segment [ss, se) of list b, each keeping its order
// (move/list_kernel/sublist_swap.rs:17-160).
template <class VT>
__device__ __forceinline__ ListDelta eval_sublist_swap(const ListModel& m, const VT* visits, const uint32_t* off,
                                                       const int64_t* load, uint32_t a, uint32_t fs, uint32_t fe, uint32_t b,
                                                       uint32_t ss, uint32_t se) {
    ListDelta r{0, 0, false};
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    const uint32_t ob = off[b], lb = off[b + 1] - ob;
    const bool intra = a == b;
    if (!(fs < fe && ss < se && fe <= la && se <= lb)) return r;
    if (intra && fs < se && ss < fe) return r;  // overlapping ranges
    r.doable = true;
    if (intra && ss < fs) {  // the early segment first (the exchange is symmetric)
        uint32_t t = fs;
        fs = ss;
        ss = t;
        t = fe;
        fe = se;
        se = t;
    }
    const uint32_t depot = (uint32_t)m.depot;
    if (m.dist_level >= 0) {
        const uint32_t A0 = visits[oa + fs], Al = visits[oa + fe - 1], B0 = visits[ob + ss], Bl = visits[ob + se - 1];
        const uint32_t pa = fs > 0 ? (uint32_t)visits[oa + fs - 1] : depot;
        const uint32_t nb = se < lb ? (uint32_t)visits[ob + se] : depot;
        int64_t acc;
        if (intra && fe == ss) {  // adjacent: pa A B nb -> pa B A nb
            acc = wsub(wadd(wadd(dist_cost(m.mat, m.dim, pa, B0), dist_cost(m.mat, m.dim, Bl, A0)), dist_cost(m.mat, m.dim, Al, nb)),
                       wadd(wadd(dist_cost(m.mat, m.dim, pa, A0), dist_cost(m.mat, m.dim, Al, B0)), dist_cost(m.mat, m.dim, Bl, nb)));
        } else {
            const uint32_t na = fe < la ? (uint32_t)visits[oa + fe] : depot;
            const uint32_t pb = ss > 0 ? (uint32_t)visits[ob + ss - 1] : depot;
            acc = wsub(wadd(dist_cost(m.mat, m.dim, pa, B0), dist_cost(m.mat, m.dim, Bl, na)),
                       wadd(dist_cost(m.mat, m.dim, pa, A0), dist_cost(m.mat, m.dim, Al, na)));
            acc = wadd(acc, wsub(wadd(dist_cost(m.mat, m.dim, pb, A0), dist_cost(m.mat, m.dim, Al, nb)),
                                 wadd(dist_cost(m.mat, m.dim, pb, B0), dist_cost(m.mat, m.dim, Bl, nb))));
        }
        r.d_dist = acc;
    }
    if (m.cap_level >= 0 && !intra) {
        int64_t da = 0, db = 0;
        for (uint32_t t = fs; t < fe; ++t) da = wadd(da, (int64_t)m.demand[visits[oa + t]]);
        for (uint32_t t = ss; t < se; ++t) db = wadd(db, (int64_t)m.demand[visits[ob + t]]);
        const int64_t la0 = load[a], lb0 = load[b];
        const int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        const int64_t after = wadd(over_cap(wadd(wsub(la0, da), db), m.capacity), over_cap(wadd(wsub(lb0, db), da), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// One trial delta for every list move kind of the generic engine on a SYMMETRIC matrix (m.mat_symmetric, checked at
// upload): every kind reduces to at most eight signed boundary legs — the legs inside a reversed segment cost the same
// in both directions and cancel exactly in wrapping i64 arithmetic — so lanes of different kinds share ONE gather round
// trip instead of running the five per-kind evaluators one after the other.  Results are bit-identical to
// eval_list_move_legs / eval_list_reverse / eval_kopt / eval_sublist_change / eval_sublist_swap (parity tests run both).
// kind = selector kind of the lane (4/16 change, 8/32 swap, 64 reverse, 128 sublist change, 256 sublist swap, 512 3-opt);
// (m0, m1, mx) = the ring entry as the generic engine packs it.
template <class VT>
__device__ __forceinline__ ListDelta eval_list_unified(const ListModel& m, const VT* visits, const uint32_t* off,
                                                       const int64_t* load, int kind, uint32_t m0, uint32_t m1, uint32_t mx) {
    ListDelta r{0, 0, false};
    const uint32_t depot = (uint32_t)m.depot;
    const uint32_t a = m0 >> 16, i = m0 & 0xFFFFu;
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    uint32_t f[8], t[8];
    int32_t sg[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = 0, t[s] = 0, sg[s] = 0;
    uint32_t b = a, PA = 0, zA = 0, PB = 0, zB = 0;  // capacity: flat ranges of the elements leaving list a / list b
    auto at = [&](uint32_t flat) { return (uint32_t)visits[flat]; };
    if (kind == 64) {  // reverse [i, e)
        const uint32_t e = m1 & 0xFFFFu;
        if (!(e > i + 1 && e <= la)) return r;
        const uint32_t prev = i > 0 ? at(oa + i - 1) : depot, next = e < la ? at(oa + e) : depot;
        const uint32_t first = at(oa + i), last = at(oa + e - 1);
        f[0] = prev, t[0] = last, sg[0] = 1;
        f[1] = first, t[1] = next, sg[1] = 1;
        f[2] = prev, t[2] = first, sg[2] = -1;
        f[3] = last, t[3] = next, sg[3] = -1;
    } else if (kind == 512) {  // 3-opt: cuts c1 = i < c2 < c3, pattern mx
        const uint32_t c1 = i, c2 = m1 >> 16, c3 = m1 & 0xFFFFu;
        if (!(mx < 7 && c1 < c2 && c2 < c3 && c3 <= la)) return r;
        const uint32_t prev = c1 > 0 ? at(oa + c1 - 1) : depot, next = c3 < la ? at(oa + c3) : depot;
        const uint32_t bf = at(oa + c1), bl = at(oa + c2 - 1), cf = at(oa + c2), cl = at(oa + c3 - 1);
        const uint32_t mask = kopt_reverse_mask(mx);
        const bool rb = (mask >> 1) & 1u, rc = (mask >> 2) & 1u, sw = kopt_swaps_segments(mx);
        const uint32_t Bf = rb ? bl : bf, Bl = rb ? bf : bl, Cf = rc ? cl : cf, Cl = rc ? cf : cl;
        const uint32_t Xf = sw ? Cf : Bf, Xl = sw ? Cl : Bl, Yf = sw ? Bf : Cf, Yl = sw ? Bl : Cl;
        f[0] = prev, t[0] = Xf, sg[0] = 1;
        f[1] = Xl, t[1] = Yf, sg[1] = 1;
        f[2] = Yl, t[2] = next, sg[2] = 1;
        f[3] = prev, t[3] = bf, sg[3] = -1;
        f[4] = bl, t[4] = cf, sg[4] = -1;
        f[5] = cl, t[5] = next, sg[5] = -1;
    } else {
        b = m1 >> 16;
        const uint32_t j = m1 & 0xFFFFu;
        const uint32_t ob = off[b], lb = off[b + 1] - ob;
        const bool intra = a == b;
        if (kind == 128) {  // sublist change: [i, e) of a -> b at dp = j (post-removal coordinates when intra)
            const uint32_t e = i + mx;
            if (!(i < e && e <= la)) return r;
            const uint32_t z = e - i;
            if (j > (intra ? la - z : lb) || (intra && j == i)) return r;
            const uint32_t first = at(oa + i), last = at(oa + e - 1);
            const uint32_t prev = i > 0 ? at(oa + i - 1) : depot, next = e < la ? at(oa + e) : depot;
            uint32_t pl, pr;
            bool dst_empty = false;
            if (!intra) {
                pl = j > 0 ? at(ob + j - 1) : depot;
                pr = j < lb ? at(ob + j) : depot;
                dst_empty = lb == 0;
            } else {
                const uint32_t l2 = la - z;
                pl = j > 0 ? at(oa + (j - 1 < i ? j - 1 : j - 1 + z)) : depot;
                pr = j < l2 ? at(oa + (j < i ? j : j + z)) : depot;
            }
            f[0] = prev, t[0] = first, sg[0] = -1;
            f[1] = last, t[1] = next, sg[1] = -1;
            f[2] = prev, t[2] = next, sg[2] = la > z ? 1 : 0;
            f[3] = pl, t[3] = first, sg[3] = 1;
            f[4] = last, t[4] = pr, sg[4] = 1;
            f[5] = pl, t[5] = pr, sg[5] = dst_empty ? 0 : -1;
            PA = oa + i, zA = z;
        } else if (kind == 256) {  // sublist swap: [fs, fe) of a <-> [ss, se) of b
            uint32_t fs = i, fe = i + (mx & 15u), ss = j, se = j + (mx >> 4);
            if (!(fs < fe && ss < se && fe <= la && se <= lb)) return r;
            if (intra && fs < se && ss < fe) return r;
            if (intra && ss < fs) {
                uint32_t q = fs;
                fs = ss, ss = q;
                q = fe;
                fe = se, se = q;
            }
            const uint32_t A0 = at(oa + fs), Al = at(oa + fe - 1), B0 = at(ob + ss), Bl = at(ob + se - 1);
            const uint32_t pa = fs > 0 ? at(oa + fs - 1) : depot, nb = se < lb ? at(ob + se) : depot;
            if (intra && fe == ss) {
                f[0] = pa, t[0] = B0, sg[0] = 1;
                f[1] = Bl, t[1] = A0, sg[1] = 1;
                f[2] = Al, t[2] = nb, sg[2] = 1;
                f[3] = pa, t[3] = A0, sg[3] = -1;
                f[4] = Al, t[4] = B0, sg[4] = -1;
                f[5] = Bl, t[5] = nb, sg[5] = -1;
            } else {
                const uint32_t na = fe < la ? at(oa + fe) : depot, pb = ss > 0 ? at(ob + ss - 1) : depot;
                f[0] = pa, t[0] = B0, sg[0] = 1;
                f[1] = Bl, t[1] = na, sg[1] = 1;
                f[2] = pa, t[2] = A0, sg[2] = -1;
                f[3] = Al, t[3] = na, sg[3] = -1;
                f[4] = pb, t[4] = A0, sg[4] = 1;
                f[5] = Al, t[5] = nb, sg[5] = 1;
                f[6] = pb, t[6] = B0, sg[6] = -1;
                f[7] = Bl, t[7] = nb, sg[7] = -1;
            }
            PA = oa + fs, zA = fe - fs, PB = ob + ss, zB = se - ss;
        } else {  // list change / list swap (plain and nearby): same legs as eval_list_move_legs
            const bool is_change = kind == 4 || kind == 16;
            uint32_t ii = i, jj = j;
            if (is_change) {
                if (ii >= la || jj > lb || (intra && (jj == ii || jj == ii + 1))) return r;
            } else {
                if (ii >= la || jj >= lb || (intra && ii == jj)) return r;
                if (intra && ii > jj) {
                    const uint32_t q = ii;
                    ii = jj, jj = q;
                }
            }
            const uint32_t P = oa + ii, Q = ob + jj;
            const uint32_t x = at(P);
            const uint32_t pa = ii > 0 ? at(P - 1) : depot, na = ii + 1 < la ? at(P + 1) : depot;
            const uint32_t vq = jj < lb ? at(Q) : depot;
            const uint32_t pb = jj > 0 ? at(Q - 1) : depot, nb = jj + 1 < lb ? at(Q + 1) : depot;
            if (!is_change && x == vq) return r;
            if (is_change) {
                f[0] = pa, t[0] = x, sg[0] = -1;
                f[1] = x, t[1] = na, sg[1] = -1;
                f[2] = pa, t[2] = na, sg[2] = la == 1 ? 0 : 1;
                f[3] = pb, t[3] = x, sg[3] = 1;
                f[4] = x, t[4] = vq, sg[4] = 1;
                f[5] = pb, t[5] = vq, sg[5] = (!intra && lb == 0) ? 0 : -1;
                PA = P, zA = 1;
            } else if (intra && jj == ii + 1) {
                f[0] = pa, t[0] = vq, sg[0] = 1;
                f[1] = vq, t[1] = x, sg[1] = 1;
                f[2] = x, t[2] = nb, sg[2] = 1;
                f[3] = pa, t[3] = x, sg[3] = -1;
                f[4] = x, t[4] = vq, sg[4] = -1;
                f[5] = vq, t[5] = nb, sg[5] = -1;
            } else {
                f[0] = pa, t[0] = vq, sg[0] = 1;
                f[1] = vq, t[1] = na, sg[1] = 1;
                f[2] = pa, t[2] = x, sg[2] = -1;
                f[3] = x, t[3] = na, sg[3] = -1;
                f[4] = pb, t[4] = x, sg[4] = 1;
                f[5] = x, t[5] = nb, sg[5] = 1;
                f[6] = pb, t[6] = vq, sg[6] = -1;
                f[7] = vq, t[7] = nb, sg[7] = -1;
                PA = P, zA = 1, PB = Q, zB = 1;
            }
            if (!is_change && intra) zA = 0, zB = 0;
        }
    }
    r.doable = true;
    if (m.dist_level >= 0) {
        int64_t acc = 0;
        if (m.mat32) {
            uint32_t v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = m.mat32[f[s] * (uint32_t)m.dim + t[s]];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t c = v[s] != 0xFFFFFFFFu ? (int64_t)v[s] : MAX_SAFE_LEG_COST;
                acc = wadd(acc, (int64_t)((uint64_t)c * (uint64_t)(int64_t)sg[s]));
            }
        } else {
            int64_t v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = m.mat[(size_t)f[s] * (size_t)m.dim + t[s]];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t c = (v[s] >= 0 && v[s] != UNREACHABLE) ? v[s] : MAX_SAFE_LEG_COST;
                acc = wadd(acc, (int64_t)((uint64_t)c * (uint64_t)(int64_t)sg[s]));
            }
        }
        r.d_dist = acc;
    }
    if (m.cap_level >= 0 && a != b) {
        int64_t da = 0, db = 0;
        for (uint32_t q = 0; q < zA; ++q) da = wadd(da, (int64_t)m.demand[at(PA + q)]);
        for (uint32_t q = 0; q < zB; ++q) db = wadd(db, (int64_t)m.demand[at(PB + q)]);
        const int64_t la0 = load[a], lb0 = load[b];
        const int64_t before = wadd(over_cap(la0, m.capacity), over_cap(lb0, m.capacity));
        const int64_t after = wadd(over_cap(wadd(wsub(la0, da), db), m.capacity), over_cap(wadd(wsub(lb0, db), da), m.capacity));
        r.d_cap = wsub(after, before);
    }
    return r;
}

// Relocates the flat range [P, P+z) so that it starts where flat position Q was (Q <= P or Q >= P+z),
// shifting the elements in between; `sync` separates the read and write halves of every chunk
// (wave_sync for one wavefront on LDS, __syncthreads for a workgroup on global memory).
template <class VT, class SYNC>
__device__ __forceinline__ void relocate_flat_segment(VT* visits, uint32_t P, uint32_t z, uint32_t Q, uint32_t tid,
                                                      uint32_t nthreads, SYNC sync) {
    // the segment itself (z <= 15) is carried in registers by the first z threads
    const VT segv = tid < z ? visits[P + tid] : (VT)0;
    sync();
    if (Q >= P + z) {  // left shift of [P+z, Q) by z, ascending chunks
        const uint32_t n = Q - (P + z);
        for (uint32_t c0 = 0; c0 < n; c0 += nthreads) {
            const bool in = c0 + tid < n;
            const uint32_t t = P + c0 + tid;
            const VT nv = in ? visits[t + z] : (VT)0;
            sync();
            if (in) visits[t] = nv;
            sync();
        }
        if (tid < z) visits[Q - z + tid] = segv;
    } else {  // right shift of [Q, P) by z, descending chunks
        const uint32_t n = P - Q;
        for (uint32_t c0 = 0; c0 < n; c0 += nthreads) {
            const bool in = c0 + tid < n;
            const uint32_t t = P + z - 1 - (in ? c0 + tid : 0u);
            const VT nv = in ? visits[t - z] : (VT)0;
            sync();
            if (in) visits[t] = nv;
            sync();
        }
        if (tid < z) visits[Q + tid] = segv;
    }
    sync();
}

template <int L>
__device__ __forceinline__ ScoreV<L> apply_delta(const ListModel& m, const int64_t* cur, const ListDelta& d) {
    ScoreV<L> s;
#pragma unroll
    for (int k = 0; k < L; ++k) s.v[k] = cur[k];
    // penalties: score level -= weight * delta(penalty sum)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (k == m.cap_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.cap_weight * (uint64_t)d.d_cap));
        if (k == m.dist_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.dist_weight * (uint64_t)d.d_dist));
    }
    return s;
}

// compact u32 copy of the distance matrix (distance_cost semantics preserved: not-finite -> sentinel)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_mat_compress(const int64_t* __restrict__ mat, size_t n,
                                                      uint32_t* __restrict__ out) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int64_t v = mat[t];
        out[t] = (v >= 0 && v != UNREACHABLE) ? (uint32_t)v : 0xFFFFFFFFu;
    }
}

SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_mat_compress16(const int64_t* __restrict__ mat, size_t n, uint16_t* __restrict__ out) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int64_t v = mat[t];
        out[t] = (v >= 0 && v < 0xFFFF) ? (uint16_t)v : (uint16_t)0xFFFFu;
    }
}

// Join of the two planning classes of a mixed model (SF_C_CROSS_OWNER_MATCH): assigned scalar entities whose value is not the list that
// holds them = assigned - (list elements whose value is their owner).  grid = R; adds its level to the scores the other kernels wrote.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_cross_owner_evaluate_all(ListModel m, const int32_t* vals, int n_scalar, int level, int64_t weight, int64_t* out_scores,
                                                                  int commit, int64_t* out_parts) {
    __shared__ unsigned long long s_cnt[2];
    const int r = blockIdx.x;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* v = vals + (size_t)r * n_scalar;
    unsigned long long assigned = 0, matched = 0;
    for (int e = threadIdx.x; e < n_scalar; e += blockDim.x) assigned += v[e] >= 0 ? 1 : 0;
    const uint32_t* off = m.off + (size_t)r * (m.V + 1);
    const uint32_t* vis = m.visits + (size_t)r * m.n_cap;
    for (int o = 0; o < m.V; ++o)
        for (uint32_t q = off[o] + threadIdx.x; q < off[o + 1]; q += blockDim.x) {
            const uint32_t e = vis[q];
            matched += (e < (uint32_t)n_scalar && v[e] == o) ? 1 : 0;
        }
    atomicAdd(&s_cnt[0], assigned);
    atomicAdd(&s_cnt[1], matched);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t cnt = (int64_t)(s_cnt[0] - s_cnt[1]);
        const int64_t pen = (int64_t)((uint64_t)weight * (uint64_t)cnt);
        if (out_scores) out_scores[(size_t)r * m.levels + level] -= pen;
        if (commit) m.score[(size_t)r * 4 + level] -= pen;
        if (out_parts) out_parts[(size_t)r * SF_EACH_WORDS + 17] = cnt;
    }
}

// The same join for the host-driven entry points (sf_step_evaluate / sf_step_evaluate_compound / sf_apply / sf_apply_compound): the fused engines keep an
// entity -> holding list map per replica and price a move from its coordinates (sf_mixed_wave.hip: xown_scalar_delta / xown_list_delta); here the map of
// ONE replica is rebuilt from its lists per call (one block), then a thread per record adds the join's delta to the scores the other kernels wrote.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_cross_owner_holders(ListModel m, int replica, int n_scalar, uint16_t* tab_all) {
    uint16_t* tab = tab_all + (size_t)replica * n_scalar;
    for (int e = threadIdx.x; e < n_scalar; e += blockDim.x) tab[e] = 0xFFFFu;  // held by no list
    __syncthreads();
    const uint32_t* off = m.off + (size_t)replica * (m.V + 1);
    const uint32_t* vis = m.visits + (size_t)replica * m.n_cap;
    for (int o = 0; o < m.V; ++o)
        for (uint32_t q = off[o] + threadIdx.x; q < off[o + 1]; q += blockDim.x)
            if (vis[q] < (uint32_t)n_scalar) tab[vis[q]] = (uint16_t)o;
}
__device__ __forceinline__ int32_t cross_owner_pen(int32_t v, uint32_t owner) { return (v >= 0 && (uint32_t)v != owner) ? 1 : 0; }
// moves = sf_move_t records (six int32 each); out-of-range coordinates price nothing (such a record is not doable and its score is not read)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_cross_owner_evaluate_moves(ListModel m, const int32_t* vals_all, int n_scalar, const uint16_t* tab_all, int replica,
                                                                    const int32_t* __restrict__ moves, int64_t n, int level, int64_t weight, int64_t* sc,
                                                                    const int32_t* doable) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (doable && !doable[i])) return;
    const int32_t* mv = moves + i * 6;
    const int32_t kind = mv[0], a = mv[1], ap = mv[2], b = mv[3], bp = mv[4], val = mv[5];
    const int32_t* vals = vals_all + (size_t)replica * n_scalar;
    const uint16_t* tab = tab_all + (size_t)replica * n_scalar;
    const uint32_t* off = m.off + (size_t)replica * (m.V + 1);
    const uint32_t* vis = m.visits + (size_t)replica * m.n_cap;
    auto moved = [&](uint32_t e, uint32_t from, uint32_t to) -> int64_t {  // element e leaves list `from` for list `to`
        if (e >= (uint32_t)n_scalar) return 0;
        const int32_t v = vals[e];
        return (int64_t)cross_owner_pen(v, to) - (int64_t)cross_owner_pen(v, from);
    };
    auto len_of = [&](int32_t o) -> uint32_t { return off[o + 1] - off[o]; };
    int64_t d = 0;
    if (kind == 0) {  // SF_MOVE_CHANGE (the wire kinds of include/solverforge_amd.h, as in k_list_evaluate_moves)
        if (a >= 0 && a < n_scalar) {
            const uint32_t ow = tab[a];
            d = (int64_t)cross_owner_pen(val, ow) - (int64_t)cross_owner_pen(vals[a], ow);
        }
    } else if (kind == 1) {  // SF_MOVE_SWAP
        if (a >= 0 && a < n_scalar && b >= 0 && b < n_scalar) {
            const uint32_t o1 = tab[a], o2 = tab[b];
            const int32_t v1 = vals[a], v2 = vals[b];
            d = (int64_t)cross_owner_pen(v2, o1) - cross_owner_pen(v1, o1) + cross_owner_pen(v1, o2) - cross_owner_pen(v2, o2);
        }
    } else if (a >= 0 && a < m.V && b >= 0 && b < m.V && a != b && ap >= 0 && bp >= 0) {  // list kinds: only elements that change lists count
        if (kind == 2) {  // SF_MOVE_LIST_CHANGE
            if ((uint32_t)ap < len_of(a)) d = moved(vis[off[a] + ap], a, b);
        } else if (kind == 3) {  // SF_MOVE_LIST_SWAP
            if ((uint32_t)ap < len_of(a) && (uint32_t)bp < len_of(b)) d = moved(vis[off[a] + ap], a, b) + moved(vis[off[b] + bp], b, a);
        } else if (kind == 5) {  // SF_MOVE_SUBLIST_CHANGE: segment [a_pos, value) of list a -> list b
            if (val > ap && (uint32_t)val <= len_of(a))
                for (int32_t t = ap; t < val; ++t) d += moved(vis[off[a] + t], a, b);
        } else if (kind == 6) {  // SF_MOVE_SUBLIST_SWAP: [a_pos, a_pos + (value & 0xFFFF)) of a <-> [b_pos, b_pos + (value >> 16)) of b
            const uint32_t la = (uint32_t)val & 0xFFFFu, lb = (uint32_t)val >> 16;
            if ((uint32_t)ap + la <= len_of(a) && (uint32_t)bp + lb <= len_of(b)) {
                for (uint32_t t = 0; t < la; ++t) d += moved(vis[off[a] + ap + t], a, b);
                for (uint32_t t = 0; t < lb; ++t) d += moved(vis[off[b] + bp + t], b, a);
            }
        }
    }  // reverse / 3-opt / permute / multi-swap stay inside one list: nothing changes
    sc[i * m.levels + level] -= (int64_t)((uint64_t)weight * (uint64_t)d);
}
// compound scalar candidates: the edits of candidate i chain (a later edit of the same entity sees the earlier one's value)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_cross_owner_evaluate_compound(const int32_t* vals_all, int n_scalar, const uint16_t* tab_all, int replica, const int32_t* __restrict__ edits,
                                                                       const int64_t* __restrict__ offsets, int64_t n, int levels, int level, int64_t weight, int64_t* sc,
                                                                       const int32_t* doable) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (doable && !doable[i])) return;
    const int32_t* vals = vals_all + (size_t)replica * n_scalar;
    const uint16_t* tab = tab_all + (size_t)replica * n_scalar;
    int64_t d = 0;
    for (int64_t k = offsets[i]; k < offsets[i + 1]; ++k) {
        const int32_t e = edits[k * 6 + 1], nv = edits[k * 6 + 5];
        if (e < 0 || e >= n_scalar) continue;
        int32_t cur = vals[e];
        for (int64_t q = offsets[i]; q < k; ++q)
            if (edits[q * 6 + 1] == e) cur = edits[q * 6 + 5];
        const uint32_t ow = tab[e];
        d += (int64_t)cross_owner_pen(nv, ow) - (int64_t)cross_owner_pen(cur, ow);
    }
    sc[i * levels + level] -= (int64_t)((uint64_t)weight * (uint64_t)d);
}
// sf_apply / sf_apply_compound: the delta priced before the move is added to the committed score once the move went through
SF_PLAIN_KERNEL
__global__ void k_cross_owner_commit(int64_t* score_level, const int64_t* delta_level, const int32_t* ok) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *ok) *score_level += *delta_level;
}

// internal node numbering of the COMPACT wave kernel (ListModel::perm): the u16 matrix with rows and columns in internal order ...
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_mat16_renumber(const uint16_t* __restrict__ src, const uint16_t* __restrict__ inv, int dim, uint16_t* __restrict__ dst) {
    const uint16_t* row = src + (size_t)inv[blockIdx.x] * dim;
    for (int t = threadIdx.x; t < dim; t += blockDim.x) dst[(size_t)blockIdx.x * dim + t] = row[inv[t]];
}
// ... and the neighbour index: row i = the row of external node inv[i], same order (distance, then EXTERNAL id: the order the
// index was sorted in), entries renamed
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_nbr_renumber(const uint16_t* __restrict__ src, const uint16_t* __restrict__ inv, const uint16_t* __restrict__ perm, int dim,
                                                      uint16_t* __restrict__ dst) {
    const uint16_t* row = src + (size_t)inv[blockIdx.x] * dim;
    for (int t = threadIdx.x; t < dim; t += blockDim.x) {
        const uint32_t e = row[t];
        dst[(size_t)blockIdx.x * dim + t] = e == 0xFFFFu ? (uint16_t)0xFFFFu : (uint16_t)((e & 0x8000u) | perm[e & 0x7FFFu]);
    }
}
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_i32_renumber(const int32_t* __restrict__ src, const uint16_t* __restrict__ inv, int n, int32_t* __restrict__ dst) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) dst[t] = src[inv[t]];
}

// ---------------------------------------------------------------------------------------
// evaluate_all / initialize: full recomputation from scratch (fresh_score; FullAssert)
// grid = R blocks.  commit != 0 also (re)builds the per-route load aggregate + cached score.
// ---------------------------------------------------------------------------------------
// out_parts (optional, [R][SF_EACH_WORDS]): raw per-constraint aggregates for ConstraintSet::evaluate_each
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_list_evaluate_all(ListModel m, int64_t* out_scores, int commit, int64_t* out_parts = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* present = (uint32_t*)smem;  // bitmap over node ids
    __shared__ unsigned long long s_cap, s_dist, s_missing;
    const int r = blockIdx.x;
    const uint32_t* visits = m.visits + (size_t)r * m.n_cap;
    const uint32_t* off = m.off + (size_t)r * (m.V + 1);
    int64_t* load = m.load + (size_t)r * m.V;
    int words = (m.dim + 31) / 32;
    for (int w = threadIdx.x; w < words; w += blockDim.x) present[w] = 0;
    if (threadIdx.x == 0) {
        s_cap = 0;
        s_dist = 0;
        s_missing = 0;
    }
    __syncthreads();
    unsigned long long cap_sum = 0, dist_sum = 0;
    for (int v = threadIdx.x; v < m.V; v += blockDim.x) {
        uint32_t o = off[v], len = off[v + 1] - o;
        int64_t ld = 0, ds = 0;
        for (uint32_t p = 0; p < len; ++p) {
            uint32_t x = visits[o + p];
            atomicOr(&present[x >> 5], 1u << (x & 31));
            if (m.demand) ld = wadd(ld, (int64_t)m.demand[x]);
            if (m.dist_level >= 0) {
                uint32_t prev = p > 0 ? visits[o + p - 1] : (uint32_t)m.depot;
                ds = wadd(ds, dist_cost(m.mat, m.dim, prev, x));
                if (p + 1 == len) ds = wadd(ds, dist_cost(m.mat, m.dim, x, (uint32_t)m.depot));
            }
        }
        if (commit) load[v] = ld;
        if (m.cap_level >= 0) cap_sum += (unsigned long long)over_cap(ld, m.capacity);
        dist_sum += (unsigned long long)ds;
    }
    atomicAdd(&s_cap, cap_sum);
    atomicAdd(&s_dist, dist_sum);
    __syncthreads();
    unsigned long long missing = 0;
    if (m.ne_level >= 0)
        for (int k = threadIdx.x; k < m.ne_n; k += blockDim.x) {
            uint32_t key = m.ne_keys[k];
            bool here = key < (uint32_t)m.dim && ((present[key >> 5] >> (key & 31)) & 1u);
            missing += here ? 0 : 1;
        }
    atomicAdd(&s_missing, missing);
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t sc[SF_MAX_LEVELS_CONST] = {0, 0, 0, 0};
        if (m.ne_level >= 0) sc[m.ne_level] = wsub(sc[m.ne_level], (int64_t)((uint64_t)m.ne_weight * s_missing));
        if (m.cap_level >= 0) sc[m.cap_level] = wsub(sc[m.cap_level], (int64_t)((uint64_t)m.cap_weight * s_cap));
        if (m.dist_level >= 0) sc[m.dist_level] = wsub(sc[m.dist_level], (int64_t)((uint64_t)m.dist_weight * s_dist));
        for (int k = 0; k < m.levels; ++k) {
            if (out_scores) out_scores[(size_t)r * m.levels + k] = sc[k];
            if (commit) m.score[(size_t)r * 4 + k] = sc[k];
        }
        if (out_parts) {
            int64_t* q = out_parts + (size_t)r * SF_EACH_WORDS;
            q[0] = (int64_t)s_cap;      // sum over routes of max(0, load - capacity)
            q[1] = (int64_t)s_dist;     // sum over routes of the depot -> ... -> depot walk
            q[2] = (int64_t)s_missing;  // A rows without a flattened B match
        }
    }
}

// ---------------------------------------------------------------------------------------
// n x evaluate_candidate for host-provided moves (the ScalarCandidateProvider / MoveSelector
// plugin surface): one thread per move against replica `replica`, state unchanged.
// ---------------------------------------------------------------------------------------
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_list_evaluate_moves(ListModel m, int replica, const int32_t* moves,
                                                             int64_t n, int64_t* out_scores,
                                                             int32_t* out_doable, int skip_foreign) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (skip_foreign && (moves[t * 6] < 2 || (moves[t * 6] > 7 && moves[t * 6] != 9 && moves[t * 6] != 10))) return;  // a scalar move of a mixed model
    const uint32_t* visits = m.visits + (size_t)replica * m.n_cap;
    const uint32_t* off = m.off + (size_t)replica * (m.V + 1);
    const int64_t* load = m.load + (size_t)replica * m.V;
    const int64_t* cur = m.score + (size_t)replica * 4;
    const int32_t* mv = moves + t * 6;
    int32_t kind = mv[0];
    ListDelta d{0, 0, false};
    // (a 3-opt move carries its middle cut in `b`, not an entity)
    bool in_range = mv[1] >= 0 && mv[1] < m.V && mv[3] >= 0 && (kind == 7 || mv[3] < m.V) && mv[2] >= 0 && mv[4] >= 0;
    if (kind == 10) {  // ListMultiSwapMove: `a` swaps (list | first << 16) in pairwise different lists, value = (second - first) per swap, one byte each:
                       // independent lists, so the deltas add (multi_swap_is_doable, move/list_kernel/multi_swap.rs:30-60)
        in_range = false;
        const int cnt = mv[1];
        bool ok = cnt >= 1 && cnt <= 3;
        uint32_t le[3] = {0, 0, 0};
        ListDelta sum{0, 0, true};
        for (int q = 0; q < cnt && ok; ++q) {
            const uint32_t w = (uint32_t)mv[2 + q];
            const uint32_t e = w & 0xFFFFu, f = w >> 16;
            const int32_t dl = (int32_t)(int8_t)(((uint32_t)mv[5] >> (8 * q)) & 0xFFu);
            const int64_t g = (int64_t)f + dl;
            ok = e < (uint32_t)m.V && dl != 0 && g >= 0;
            for (int q2 = 0; q2 < q && ok; ++q2) ok = le[q2] != e;
            le[q] = e;
            if (!ok) break;
            const ListDelta one = eval_list_swap(m, visits, off, load, e, f, e, (uint32_t)g);
            ok = one.doable;
            sum.d_dist = wadd(sum.d_dist, one.d_dist);
            sum.d_cap = wadd(sum.d_cap, one.d_cap);
        }
        sum.doable = ok;
        if (ok) d = sum;
    }
    if (in_range) {
        if (kind == 2)
            d = eval_list_change(m, visits, off, load, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[3], (uint32_t)mv[4]);
        else if (kind == 3)
            d = eval_list_swap(m, visits, off, load, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[3], (uint32_t)mv[4]);
        else if (kind == 4 && mv[1] == mv[3])
            d = eval_list_reverse(m, visits, off, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[4]);
        else if (kind == 6 && mv[5] >= 0)
            d = eval_sublist_swap(m, visits, off, load, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[2] + ((uint32_t)mv[5] & 0xFFFFu),
                                  (uint32_t)mv[3], (uint32_t)mv[4], (uint32_t)mv[4] + ((uint32_t)mv[5] >> 16));
        else if (kind == 5 && mv[5] >= 0)
            d = eval_sublist_change(m, visits, off, load, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[5], (uint32_t)mv[3], (uint32_t)mv[4]);
        else if (kind == 7 && mv[5] >= 0 && mv[3] >= 0)  // (a, a_pos = cut 1, b = cut 2, b_pos = cut 3, value = pattern)
            d = eval_kopt(m, visits, off, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[3], (uint32_t)mv[4], (uint32_t)mv[5]);
        else if (kind == 9 && mv[1] == mv[3] && mv[4] > mv[2] && mv[5] >= 0)  // (a, a_pos = start, b = a, b_pos = end, value = permutation rank)
            d = eval_list_permute(m, visits, off, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)(mv[4] - mv[2]), (uint32_t)mv[5]);
    }
    out_doable[t] = d.doable ? 1 : 0;
    ScoreV<4> s = apply_delta<4>(m, cur, d);
    for (int k = 0; k < m.levels; ++k) out_scores[t * m.levels + k] = d.doable ? s.v[k] : 0;
}

// ---------------------------------------------------------------------------------------
// Committed move application on global state (sf_apply): one block per call.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void apply_list_move_block(const ListModel& m, uint32_t* visits, uint32_t* off,
                                                      int64_t* load, int kind, uint32_t a, uint32_t i,
                                                      uint32_t b, uint32_t j, uint32_t ext = 0) {
    // all threads of the block call this; contains __syncthreads
    const uint32_t total = off[m.V];
    if (kind == 2) {
        uint32_t P = off[a] + i, Q = off[b] + j;
        uint32_t x = visits[P];
        __syncthreads();
        // gather old values for the range this thread rewrites.  Chunk order matters once the flat
        // array is longer than the block: a left shift (P < Q) reads t+1, so chunks go upwards;
        // a right shift (P > Q) reads t-1, so chunks go downwards (reads stay ahead of writes).
        const uint32_t n_chunks = (total + blockDim.x - 1) / blockDim.x;
        for (uint32_t ci = 0; ci < n_chunks; ++ci) {
            const uint32_t base = (P > Q ? n_chunks - 1 - ci : ci) * blockDim.x;
            uint32_t t = base + threadIdx.x;
            uint32_t nv = 0;
            bool wr = false;
            if (t < total) {
                if (P < Q) {
                    if (t >= P && t + 1 < Q) {
                        nv = visits[t + 1];
                        wr = true;
                    } else if (t + 1 == Q) {
                        nv = x;
                        wr = true;
                    }
                } else if (P > Q) {
                    if (t > Q && t <= P) {
                        nv = visits[t - 1];
                        wr = true;
                    } else if (t == Q) {
                        nv = x;
                        wr = true;
                    }
                }
            }
            __syncthreads();
            if (wr) visits[t] = nv;
            __syncthreads();
        }
        if (a != b) {
            for (uint32_t rr = threadIdx.x; rr <= (uint32_t)m.V; rr += blockDim.x) {
                if (a < b && rr > a && rr <= b) off[rr] -= 1;
                if (a > b && rr > b && rr <= a) off[rr] += 1;
            }
            if (threadIdx.x == 0 && m.demand) {
                int64_t dx = (int64_t)m.demand[x];
                load[a] = wsub(load[a], dx);
                load[b] = wadd(load[b], dx);
            }
        }
    } else if (kind == 6) {  // sublist swap: [i, i + (ext & 0xFFFF)) of a <-> [j, j + (ext >> 16)) of b
        const uint32_t za = ext & 0xFFFFu, zb = ext >> 16;
        const uint32_t PA = off[a] + i, PB = off[b] + j;
        const bool a_first = PA < PB;  // X = the segment at the lower flat position
        const uint32_t PX = a_first ? PA : PB, zx = a_first ? za : zb, PY = a_first ? PB : PA, zy = a_first ? zb : za;
        const uint32_t ox = a_first ? a : b, oy = a_first ? b : a;
        int64_t da = 0, db = 0;
        if (threadIdx.x == 0 && a != b && m.demand) {
            for (uint32_t t = 0; t < za; ++t) da = wadd(da, (int64_t)m.demand[visits[PA + t]]);
            for (uint32_t t = 0; t < zb; ++t) db = wadd(db, (int64_t)m.demand[visits[PB + t]]);
        }
        __syncthreads();
        relocate_flat_segment(visits, PY, zy, PX, threadIdx.x, blockDim.x, [] { __syncthreads(); });            // Y X mid
        relocate_flat_segment(visits, PX + zy, zx, PY + zy, threadIdx.x, blockDim.x, [] { __syncthreads(); });  // Y mid X
        if (a != b) {
            for (uint32_t rr = threadIdx.x; rr <= (uint32_t)m.V; rr += blockDim.x)
                if (rr > ox && rr <= oy) off[rr] = off[rr] + zy - zx;
            if (threadIdx.x == 0 && m.demand) {
                load[a] = wadd(wsub(load[a], da), db);
                load[b] = wadd(wsub(load[b], db), da);
            }
        }
    } else if (kind == 5) {  // sublist change: segment [i, ext) of list a -> list b at j
        const uint32_t z = ext - i, P = off[a] + i;
        const uint32_t Q = a != b ? off[b] + j : (j <= i ? off[a] + j : off[a] + j + z);
        int64_t dsum = 0;
        if (threadIdx.x == 0 && a != b && m.demand)
            for (uint32_t t = 0; t < z; ++t) dsum = wadd(dsum, (int64_t)m.demand[visits[P + t]]);
        __syncthreads();
        relocate_flat_segment(visits, P, z, Q, threadIdx.x, blockDim.x, [] { __syncthreads(); });
        if (a != b) {
            for (uint32_t rr = threadIdx.x; rr <= (uint32_t)m.V; rr += blockDim.x) {
                if (a < b && rr > a && rr <= b) off[rr] -= z;
                if (a > b && rr > b && rr <= a) off[rr] += z;
            }
            if (threadIdx.x == 0 && m.demand) {
                load[a] = wsub(load[a], dsum);
                load[b] = wadd(load[b], dsum);
            }
        }
    } else if (kind == 4) {  // reverse [i, j) of list a: thread t exchanges the t-th pair from the ends
        const uint32_t lo = off[a] + i, hi = off[a] + j;  // hi exclusive
        for (uint32_t t = threadIdx.x; t < (hi - lo) / 2; t += blockDim.x) {
            const uint32_t x = visits[lo + t], y = visits[hi - 1 - t];
            visits[lo + t] = y;
            visits[hi - 1 - t] = x;
        }
    } else if (kind == 9) {  // permute the window [i, j) of list a by the ext-th permutation of its positions
        const uint32_t base = off[a] + i, size = j - i;
        const uint32_t perm = nth_permutation_nibbles(size, ext);
        uint32_t nv = 0;
        if (threadIdx.x < size) nv = visits[base + ((perm >> (4u * threadIdx.x)) & 15u)];
        __syncthreads();
        if (threadIdx.x < size) visits[base + threadIdx.x] = nv;
    } else if (kind == 7) {  // 3-opt: cuts i < b < j of list a, pattern = ext; at most three range reversals
        const uint32_t base = off[a], c1 = i, c2 = b, c3 = j;
        const uint32_t mask = kopt_reverse_mask(ext);
        const bool rb = (mask >> 1) & 1u, rc = (mask >> 2) & 1u;
        auto reverse_range = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t t = threadIdx.x; t < (hi - lo) / 2; t += blockDim.x) {
                const uint32_t x = visits[base + lo + t], y = visits[base + hi - 1 - t];
                visits[base + lo + t] = y;
                visits[base + hi - 1 - t] = x;
            }
            __syncthreads();
        };
        if (!kopt_swaps_segments(ext)) {
            if (rb) reverse_range(c1, c2);
            if (rc) reverse_range(c2, c3);
        } else {  // B C -> reverse all = C^r B^r, then undo the reversal of whichever segment keeps its direction
            const uint32_t zc = c3 - c2;
            reverse_range(c1, c3);
            if (!rc) reverse_range(c1, c1 + zc);
            if (!rb) reverse_range(c1 + zc, c3);
        }
    } else if (kind == 3) {
        if (threadIdx.x == 0) {
            uint32_t pa = off[a] + i, pb = off[b] + j;
            uint32_t x = visits[pa], y = visits[pb];
            visits[pa] = y;
            visits[pb] = x;
            if (a != b && m.demand) {
                int64_t dx = (int64_t)m.demand[x], dy = (int64_t)m.demand[y];
                load[a] = wadd(wsub(load[a], dx), dy);
                load[b] = wadd(wsub(load[b], dy), dx);
            }
        }
    }
    __syncthreads();
}

SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_list_apply(ListModel m, int replica, int kind, uint32_t a, uint32_t i,
                                                    uint32_t b, uint32_t j, uint32_t ext, int32_t* out_ok) {
    uint32_t* visits = m.visits + (size_t)replica * m.n_cap;
    uint32_t* off = m.off + (size_t)replica * (m.V + 1);
    int64_t* load = m.load + (size_t)replica * m.V;
    int64_t* cur = m.score + (size_t)replica * 4;
    __shared__ ListDelta s_d;
    if (threadIdx.x == 0) {
        s_d = kind == 2   ? eval_list_change(m, visits, off, load, a, i, b, j)
              : kind == 3 ? eval_list_swap(m, visits, off, load, a, i, b, j)
              : kind == 4 ? eval_list_reverse(m, visits, off, a, i, j)
              : kind == 7 ? eval_kopt(m, visits, off, a, i, b, j, ext)
              : kind == 9 ? ((a == b && j > i) ? eval_list_permute(m, visits, off, a, i, j - i, ext) : ListDelta{0, 0, false})
              : kind == 5 ? eval_sublist_change(m, visits, off, load, a, i, ext, b, j)
                          : eval_sublist_swap(m, visits, off, load, a, i, i + (ext & 0xFFFFu), b, j, j + (ext >> 16));
    }
    __syncthreads();
    ListDelta d = s_d;
    if (!d.doable) {
        if (threadIdx.x == 0) *out_ok = 0;
        return;
    }
    apply_list_move_block(m, visits, off, load, kind, a, i, b, j, ext);
    if (threadIdx.x == 0) {
        ScoreV<4> s = apply_delta<4>(m, cur, d);
        for (int k = 0; k < 4; ++k) cur[k] = s.v[k];
        *out_ok = 1;
    }
}

// phase start: last_step_score = score; LA history filled; best = working (phase.rs:250-261)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_list_phase_start(ListModel m, SearchParams p) {
    const int r = blockIdx.x;
    const int64_t* cur = m.score + (size_t)r * 4;
    for (int k = threadIdx.x; k < 4; k += blockDim.x) {
        p.last_step_score[(size_t)r * 4 + k] = cur[k];
        m.best_score[(size_t)r * 4 + k] = cur[k];
        if (p.dla_best) p.dla_best[(size_t)r * 4 + k] = cur[k];  // DiversifiedLateAcceptance::phase_started
    }
    for (int h = threadIdx.x; h < p.la_size * 4; h += blockDim.x)
        p.la_hist[(size_t)r * p.la_size * 4 + h] = cur[h & 3];
    for (int t = threadIdx.x; t < m.n_cap; t += blockDim.x)
        m.best_visits[(size_t)r * m.n_cap + t] = m.visits[(size_t)r * m.n_cap + t];
    for (int t = threadIdx.x; t <= m.V; t += blockDim.x)
        m.best_off[(size_t)r * (m.V + 1) + t] = m.off[(size_t)r * (m.V + 1) + t];
    if (threadIdx.x == 0) {
        p.la_idx[r] = 0;
        p.step_index[r] = 0;
        p.has_best[r] = 1;
    }
}

// Elite migration inside one context (sf_portfolio_migrate_local; an extension, not in the reference): replica r with
// src[r] != r adopts the BEST solution of replica src[r] as its working and best solution.  Its per-route loads are rebuilt
// from the adopted lists, its cached / last-step / best score is the elite's best score and its LateAcceptance history
// restarts at that score (phase_started semantics, late_acceptance.rs:89-101); counters, step index and seed draws keep
// running, so the copies diverge from the elite and from each other.  Elites (src[r] == r) are never written.  grid = R.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_list_migrate(ListModel m, SearchParams p, const int32_t* __restrict__ src) {
    const int r = blockIdx.x;
    const int s = src[r];
    if (s == r) return;
    const uint32_t* so = m.best_off + (size_t)s * (m.V + 1);
    const uint32_t* sv = m.best_visits + (size_t)s * m.n_cap;
    const uint32_t tot = so[m.V];
    for (uint32_t t = threadIdx.x; t <= (uint32_t)m.V; t += blockDim.x) {
        m.off[(size_t)r * (m.V + 1) + t] = so[t];
        m.best_off[(size_t)r * (m.V + 1) + t] = so[t];
    }
    for (uint32_t t = threadIdx.x; t < tot; t += blockDim.x) {
        m.visits[(size_t)r * m.n_cap + t] = sv[t];
        m.best_visits[(size_t)r * m.n_cap + t] = sv[t];
    }
    for (uint32_t v = threadIdx.x; v < (uint32_t)m.V; v += blockDim.x) {
        int64_t ld = 0;
        if (m.demand)
            for (uint32_t q = so[v]; q < so[v + 1]; ++q) ld = wadd(ld, (int64_t)m.demand[sv[q]]);
        m.load[(size_t)r * m.V + v] = ld;
    }
    const int64_t* bs = m.best_score + (size_t)s * 4;
    for (int k = threadIdx.x; k < 4; k += blockDim.x) {
        m.score[(size_t)r * 4 + k] = bs[k];
        m.best_score[(size_t)r * 4 + k] = bs[k];
        p.last_step_score[(size_t)r * 4 + k] = bs[k];
        if (p.dla_best) p.dla_best[(size_t)r * 4 + k] = bs[k];
    }
    for (int h = threadIdx.x; h < p.la_size * 4; h += blockDim.x) p.la_hist[(size_t)r * p.la_size * 4 + h] = bs[h & 3];
    if (threadIdx.x == 0) p.la_idx[r] = 0;
}

// ---------------------------------------------------------------------------------------
// The fused persistent search kernel.
// ---------------------------------------------------------------------------------------
constexpr int MAXW = 16;  // waves per workgroup (1024 threads)

template <int L>
struct Ctl {
    uint64_t step_index, step_seed;
    int64_t cur[L], late[L], best[L], best_sol[L];
    uint32_t best_m0, best_m1;
    int32_t best_kind, has_best;
    uint64_t equal_count;
    uint32_t accepted;
    uint32_t head[MAX_LEAVES], tail[MAX_LEAVES], src_next[MAX_LEAVES], src_total[MAX_LEAVES];
    uint32_t perm_start[MAX_LEAVES], perm_stride[MAX_LEAVES];
    int32_t gen_done[MAX_LEAVES], exhausted[MAX_LEAVES];
    uint32_t pulls;
    int32_t first_leaf;
    int32_t done;
    int32_t wave_leaf[MAXW];
    uint32_t wave_src[MAXW], wave_cnt[MAXW];
    uint64_t st[SF_STATS_WORDS];
    uint64_t trace_n;
    uint64_t best_ti;  // trace ordinal (within the step) of the forager's current pick
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// LDS carve (bytes); mirrored by list_search_lds_bytes() on the host.
template <int L>
struct Carve {
    size_t ctl, load, qscore, visits, off, node, qmove, slotbase, routeat, rankof, total;
    __host__ __device__ Carve(int V, int n_cap, int dim) {
        size_t o = 0;
        ctl = o;
        o = align_up(o + sizeof(Ctl<L>), 16);
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        qscore = o;
        o = align_up(o + sizeof(int64_t) * L * QCAP * MAX_LEAVES, 16);
        visits = o;
        o = align_up(o + sizeof(uint32_t) * n_cap, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        node = o;
        o = align_up(o + sizeof(uint32_t) * dim, 16);
        qmove = o;
        o = align_up(o + sizeof(uint32_t) * 2 * QCAP * MAX_LEAVES, 16);
        slotbase = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1) * MAX_LEAVES, 16);
        routeat = o;
        o = align_up(o + sizeof(uint32_t) * V * MAX_LEAVES, 16);
        rankof = o;
        o = align_up(o + sizeof(uint32_t) * V * MAX_LEAVES, 16);
        total = o;
    }
};

template <int L>
__device__ __forceinline__ ScoreV<L> wave_max_score(ScoreV<L> s, bool valid) {
    // lexicographic max across the wave; invalid lanes contribute -inf
    if (!valid) {
#pragma unroll
        for (int k = 0; k < L; ++k) s.v[k] = INT64_MIN;
    }
#pragma unroll
    for (int mlane = 32; mlane >= 1; mlane >>= 1) {
        ScoreV<L> o;
#pragma unroll
        for (int k = 0; k < L; ++k) o.v[k] = (int64_t)shfl_xor_u64((uint64_t)s.v[k], mlane);
        if (score_cmp<L>(o, s) > 0) s = o;
    }
    return s;
}

// Sorted top-K list held one entry per lane (lane i = i-th smallest key).
struct TopK {
    uint64_t key;
    uint32_t pay;
    uint64_t kth;  // wave-uniform current K-th key (threshold)
};

__device__ __forceinline__ void topk_insert(TopK& t, uint32_t K, uint64_t bk, uint32_t bp) {
    const uint32_t lane = lane_id();
    uint32_t pos = (uint32_t)__popcll(__ballot(lane < K && t.key < bk));
    if (pos < K) {
        uint64_t upk = shfl_up_u64(t.key, 1);
        uint32_t upp = __shfl_up(t.pay, 1);
        if (lane > pos && lane < K) {
            t.key = upk;
            t.pay = upp;
        }
        if (lane == pos) {
            t.key = bk;
            t.pay = bp;
        }
        t.kth = shfl_u64(t.key, (int)K - 1);
    }
}

__device__ __forceinline__ void topk_offer(TopK& t, uint32_t K, uint64_t key, uint32_t pay) {
    uint64_t mask = __ballot(key < t.kth);
    while (mask) {
        int j = __ffsll((unsigned long long)mask) - 1;
        mask &= mask - 1;
        uint64_t bk = shfl_u64(key, j);
        uint32_t bp = __shfl(pay, j);
        if (bk < t.kth) topk_insert(t, K, bk, bp);
    }
}

template <int L, bool TRACE>
__global__ __launch_bounds__(1024) void k_list_search(ListModel m, SearchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Carve<L> cv(m.V, m.n_cap, m.dim);
    Ctl<L>& c = *(Ctl<L>*)(smem + cv.ctl);
    int64_t* s_load = (int64_t*)(smem + cv.load);
    int64_t* q_score = (int64_t*)(smem + cv.qscore);
    uint32_t* s_visits = (uint32_t*)(smem + cv.visits);
    uint32_t* s_off = (uint32_t*)(smem + cv.off);
    uint32_t* node_slot = (uint32_t*)(smem + cv.node);
    uint32_t* q_move = (uint32_t*)(smem + cv.qmove);
    uint32_t* slot_base = (uint32_t*)(smem + cv.slotbase);
    uint32_t* route_at = (uint32_t*)(smem + cv.routeat);
    uint32_t* rank_of = (uint32_t*)(smem + cv.rankof);

    const int r = blockIdx.x + p.replica_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t NW = blockDim.x >> 6;
    const int V = m.V;
    uint32_t* g_visits = m.visits + (size_t)r * m.n_cap;
    uint32_t* g_off = m.off + (size_t)r * (V + 1);
    int64_t* g_load = m.load + (size_t)r * V;
    int64_t* g_score = m.score + (size_t)r * 4;
    const bool tracing = TRACE && r == p.trace_replica;
    __shared__ uint64_t s_sa[SA_WORDS];  // SimulatedAnnealing acceptor state (touched by wave 0 only)
    const bool annealing = p.acceptor == 3;
    if (annealing && wave == 0) sa_load(s_sa, p.sa, r, lane);

    // ---- load replica state into LDS ----
    for (uint32_t t = tid; t <= (uint32_t)V; t += blockDim.x) s_off[t] = g_off[t];
    for (uint32_t t = tid; t < (uint32_t)V; t += blockDim.x) s_load[t] = g_load[t];
    for (uint32_t t = tid; t < (uint32_t)m.dim; t += blockDim.x) node_slot[t] = NODE_NONE;
    __syncthreads();
    const uint32_t total0 = s_off[V];
    for (uint32_t t = tid; t < total0; t += blockDim.x) s_visits[t] = g_visits[t];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
            c.cur[k] = g_score[k];
            c.best_sol[k] = m.best_score[(size_t)r * 4 + k];
        }
        for (int k = 0; k < SF_STATS_WORDS; ++k) c.st[k] = 0;
        c.trace_n = 0;
    }
    __syncthreads();
    for (uint32_t v = tid; v < (uint32_t)V; v += blockDim.x) {
        uint32_t o = s_off[v], len = s_off[v + 1] - o;
        for (uint32_t q = 0; q < len; ++q) node_slot[s_visits[o + q]] = (v << 16) | q;
    }
    __syncthreads();

    for (int64_t step = 0; step < p.n_steps; ++step) {
        // ---- (A) step start ------------------------------------------------------------
        if (tid == 0) {
            uint64_t sidx, sseed;
            if (p.dry_run) {
                sidx = p.dry_step_index;
                sseed = p.dry_step_seed;
            } else {
                sidx = p.step_index[r] + (uint64_t)step;
                uint64_t draw = p.seed_draws[r] + (uint64_t)step;
                if (p.explicit_seeds && (int64_t)draw < p.n_explicit)
                    sseed = p.explicit_seeds[(size_t)r * p.n_explicit + draw];
                else
                    sseed = step_seed(p.random_seed + (uint64_t)r, draw);
            }
            c.step_index = sidx;
            c.step_seed = sseed;
            StreamCtx ctx{sidx, sseed, p.order};
            if (p.acceptor == 1) {
                int idx = p.dry_run ? 0 : (int)((p.la_idx[r] + step) % p.la_size);
#pragma unroll
                for (int k = 0; k < L; ++k) c.late[k] = p.la_hist[((size_t)r * p.la_size + idx) * 4 + k];
            }
            c.has_best = 0;
            c.equal_count = 0;
            c.accepted = 0;
            c.pulls = 0;
            c.done = 0;
            for (int l = 0; l < p.n_leaves; ++l) {
                uint64_t ent_salt = (p.leaf[l].kind == 16 ? SALT_NEARBY_CHANGE_ENTITY : SALT_NEARBY_SWAP_ENTITY) ^
                                    (uint64_t)p.leaf[l].descriptor;
                uint32_t st, sd;
                ctx.perm_params((uint32_t)V, ent_salt, st, sd);
                c.perm_start[l] = st;
                c.perm_stride[l] = sd;
                c.head[l] = c.tail[l] = 0;
                c.src_next[l] = 0;
                c.src_total[l] = s_off[V];
                c.gen_done[l] = s_off[V] == 0;
                c.exhausted[l] = 0;
            }
            // union: >1 leaf => StratifiedRandom, equal weights (vec_union.rs:229-245); with two
            // children the stride is always 1, so the order is first, other, first, ...
            c.first_leaf = p.n_leaves > 1 ? (int32_t)ctx.random_index((uint32_t)p.n_leaves, SALT_UNION_OFFSET) : 0;
        }
        __syncthreads();
        // ---- (B) per-leaf entity order tables -------------------------------------------
        for (int l = 0; l < p.n_leaves; ++l) {
            for (uint32_t k = tid; k < (uint32_t)V; k += blockDim.x) {
                uint32_t e = (uint32_t)(((uint64_t)c.perm_start[l] + (uint64_t)k * c.perm_stride[l]) % (uint32_t)V);
                route_at[l * V + k] = e;
                rank_of[l * V + e] = k;
            }
        }
        __syncthreads();
        if (wave < (uint32_t)p.n_leaves) {  // slot_base[k] = sum_{k'<k} (len(route_at[k'])+1)
            const int l = wave;
            uint32_t carry = 0;
            for (uint32_t base = 0; base < (uint32_t)V; base += 64) {
                uint32_t k = base + lane;
                uint32_t v = 0;
                if (k < (uint32_t)V) {
                    uint32_t e = route_at[l * V + k];
                    v = s_off[e + 1] - s_off[e] + 1;
                }
                uint32_t inc = wave_incl_scan(v);
                if (k < (uint32_t)V) slot_base[l * (V + 1) + k] = carry + inc - v;
                carry += __shfl(inc, 63);
            }
            if (lane == 0) slot_base[l * (V + 1) + V] = carry;
        }
        __syncthreads();

        // ---- (C) candidate rounds ------------------------------------------------------
        for (;;) {
            // C1: wave -> (leaf, source) assignment by one lane
            if (tid == 0) {
                uint32_t assigned[MAX_LEAVES] = {0, 0};
                for (uint32_t w = 0; w < NW; ++w) {
                    int pick = -1;
                    uint32_t pick_pending = 0xFFFFFFFFu;
                    for (int l = 0; l < p.n_leaves; ++l) {
                        if (c.gen_done[l]) continue;
                        uint32_t K = (uint32_t)p.leaf[l].max_nearby;
                        uint32_t pending = (c.tail[l] - c.head[l]) + assigned[l] * K;
                        if (pending + K > QCAP) continue;
                        if (pending < pick_pending) {
                            pick = l;
                            pick_pending = pending;
                        }
                    }
                    c.wave_leaf[w] = pick;
                    c.wave_cnt[w] = 0;
                    if (pick >= 0) {
                        c.wave_src[w] = c.src_next[pick]++;
                        assigned[pick]++;
                        if (c.src_next[pick] >= c.src_total[pick]) c.gen_done[pick] = 1;
                    }
                }
            }
            __syncthreads();
            // C2: per-wave source scan + stable top-k
            const int wl = c.wave_leaf[wave];
            TopK tk{~0ULL, 0u, ~0ULL};
            uint32_t K = 0, cnt = 0, se = 0, sp = 0, sx = 0;
            bool is_change = false;
            if (wl >= 0) {
                const int l = wl;
                K = (uint32_t)p.leaf[l].max_nearby;
                is_change = p.leaf[l].kind == 16;
                const uint32_t s = c.wave_src[wave];
                const uint32_t* sb = slot_base + l * (V + 1);
                // rank k with src_base[k] <= s < src_base[k+1], src_base[k] = slot_base[k] - k
                uint32_t lo = 0, hi = (uint32_t)V;  // invariant: src_base[lo] <= s < src_base[hi]
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (sb[mid] - mid <= s)
                        lo = mid;
                    else
                        hi = mid;
                }
                const uint32_t k = lo;
                se = route_at[l * V + k];
                const uint32_t len = s_off[se + 1] - s_off[se];
                const uint32_t o = s - (sb[k] - k);
                StreamCtx ctx{c.step_index, c.step_seed, p.order};
                const uint64_t src_salt = (is_change ? SALT_NEARBY_CHANGE_SOURCE : SALT_NEARBY_SWAP_SOURCE) ^
                                          (uint64_t)se ^ (uint64_t)p.leaf[l].descriptor;
                sp = ctx.selection_index(o, len, src_salt);
                sx = s_visits[s_off[se] + sp];
                const int64_t* row = m.mat + (size_t)sx * (size_t)m.dim;
                for (uint32_t base = 0; base < (uint32_t)m.dim; base += 64) {
                    const uint32_t y = base + lane;
                    uint64_t key0 = ~0ULL, key1 = ~0ULL;
                    uint32_t pay0 = 0, pay1 = 0;
                    if (y < (uint32_t)m.dim) {
                        const uint32_t slot = node_slot[y];
                        if (slot != NODE_NONE) {
                            const uint32_t r2 = slot >> 16, dp = slot & 0xFFFFu;
                            const int64_t v = row[y];
                            if (v >= 0 && v != UNREACHABLE) {  // finite_distance (problem_data.rs:44-47)
                                const uint64_t hi_key = (uint64_t)v << 24;
                                if (is_change) {
                                    if (r2 == se) {
                                        if (dp != sp && dp != sp + 1) {
                                            key0 = hi_key | dp;
                                            pay0 = slot;
                                        }
                                        if (dp + 1 == len && len != sp + 1) {  // end slot dp2 = len
                                            key1 = hi_key | len;
                                            pay1 = (r2 << 16) | len;
                                        }
                                    } else {
                                        const uint32_t len2 = s_off[r2 + 1] - s_off[r2];
                                        const uint32_t ord = ORD_INTER_BASE + sb[rank_of[l * V + r2]] + dp;
                                        key0 = hi_key | ord;
                                        pay0 = slot;
                                        if (dp + 1 == len2) {
                                            key1 = hi_key | (ord + 1);
                                            pay1 = (r2 << 16) | len2;
                                        }
                                    }
                                } else {
                                    if (r2 == se) {
                                        if (dp > sp) {
                                            key0 = hi_key | dp;
                                            pay0 = slot;
                                        }
                                    } else if (rank_of[l * V + r2] > k) {
                                        key0 = hi_key | (ORD_INTER_BASE + sb[rank_of[l * V + r2]] + dp);
                                        pay0 = slot;
                                    }
                                }
                            }
                        }
                    }
                    topk_offer(tk, K, key0, pay0);
                    if (is_change) topk_offer(tk, K, key1, pay1);
                }
                cnt = (uint32_t)__popcll(__ballot(lane < K && tk.key != ~0ULL));
                if (lane == 0) c.wave_cnt[wave] = cnt;
            }
            __syncthreads();
            // C3: trial-score the kept candidates and append them to the leaf ring in source order
            if (wl >= 0 && cnt > 0) {
                uint32_t offq = c.tail[wl];
                for (uint32_t w2 = 0; w2 < wave; ++w2)
                    if (c.wave_leaf[w2] == wl) offq += c.wave_cnt[w2];
                if (lane < cnt) {
                    const uint32_t r2 = tk.pay >> 16, dp = tk.pay & 0xFFFFu;
                    ListDelta d = is_change ? eval_list_change(m, s_visits, s_off, s_load, se, sp, r2, dp)
                                            : eval_list_swap(m, s_visits, s_off, s_load, se, sp, r2, dp);
                    ScoreV<L> sc = apply_delta<L>(m, c.cur, d);
                    const uint32_t qi = (offq + lane) & (QCAP - 1);
                    uint32_t* qm = q_move + ((size_t)wl * QCAP + qi) * 2;
                    qm[0] = (se << 16) | sp | (d.doable ? 0u : 0x80000000u);
                    qm[1] = tk.pay;
                    int64_t* qs = q_score + ((size_t)wl * QCAP + qi) * L;
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) qs[kk] = sc.v[kk];
                }
            }
            __syncthreads();
            // C4: union replay by wave 0 (acceptor + forager in cursor order)
            if (wave == 0) {
                if (lane == 0) {
                    uint64_t scored = 0, nsrc = 0;
                    for (uint32_t w2 = 0; w2 < NW; ++w2)
                        if (c.wave_leaf[w2] >= 0) {
                            c.tail[c.wave_leaf[w2]] += c.wave_cnt[w2];
                            scored += c.wave_cnt[w2];
                            ++nsrc;
                        }
                    c.st[7] += scored;
                    c.st[8] += nsrc;
                }
                __builtin_amdgcn_wave_barrier();
                // wave-uniform replay state kept in registers
                uint32_t head0 = c.head[0], head1 = c.head[1];
                const uint32_t tail0 = c.tail[0], tail1 = c.tail[1];
                int ex0 = c.exhausted[0], ex1 = p.n_leaves > 1 ? c.exhausted[1] : 1;
                const int gd0 = c.gen_done[0], gd1 = p.n_leaves > 1 ? c.gen_done[1] : 1;
                uint32_t pulls = c.pulls, accepted = c.accepted;
                int has_best = c.has_best;
                uint64_t equal_count = c.equal_count;
                ScoreV<L> best, cur, late, forager_best;  // forager_best: best score ever seen at step start (step.rs:53-58)
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    best.v[kk] = c.best[kk];
                    cur.v[kk] = c.cur[kk];
                    late.v[kk] = c.late[kk];
                    forager_best.v[kk] = c.best_sol[kk];
                }
                uint32_t best_m0 = c.best_m0, best_m1 = c.best_m1;
                int best_kind = c.best_kind;
                uint64_t st_gen = 0, st_acc = 0, st_calc = 0, st_nd = 0;
                int done = 0;
                for (;;) {
                    const bool live0 = !ex0, live1 = !ex1;
                    if (!live0 && !live1) {
                        done = 1;
                        break;
                    }
                    uint32_t lf, idx;
                    if (live0 && live1) {
                        const uint32_t l0 = ((uint32_t)c.first_leaf + pulls) & 1u;
                        lf = (l0 + lane) & 1u;
                        idx = (lf ? head1 : head0) + (lane >> 1);
                    } else {
                        lf = live0 ? 0u : 1u;
                        idx = (lf ? head1 : head0) + lane;
                    }
                    const bool avail = idx < (lf ? tail1 : tail0);
                    const uint64_t availmask = __ballot(avail);
                    const uint32_t nvalid = availmask == ~0ULL ? 64u : (uint32_t)(__ffsll((unsigned long long)~availmask) - 1);
                    if (nvalid == 0) {
                        const uint32_t lf0 = __shfl(lf, 0);
                        const int gd = lf0 ? gd1 : gd0;
                        if (gd) {  // the scheduler discovers the exhausted child at this pull
                            if (lf0)
                                ex1 = 1;
                            else
                                ex0 = 1;
                            continue;
                        }
                        break;  // need another generation round
                    }
                    const bool valid = lane < nvalid;
                    uint32_t m0 = 0, m1 = 0;
                    ScoreV<L> sc;
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) sc.v[kk] = 0;
                    if (valid) {
                        const uint32_t qi = idx & (QCAP - 1);
                        const uint32_t* qm = q_move + ((size_t)lf * QCAP + qi) * 2;
                        m0 = qm[0];
                        m1 = qm[1];
                        const int64_t* qs = q_score + ((size_t)lf * QCAP + qi) * L;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) sc.v[kk] = qs[kk];
                    }
                    const bool doable = valid && !(m0 & 0x80000000u);
                    bool acc = false;
                    if (doable) {
                        if (p.acceptor == 0)
                            acc = score_cmp<L>(sc, cur) > 0;
                        else if (p.acceptor == 1)
                            acc = score_cmp<L>(sc, cur) >= 0 || score_cmp<L>(sc, late) >= 0;
                    }
                    SaChunk sach;
                    if (annealing) acc = sa_decide<L>(s_sa, p.sa, doable, sc, cur, lane, sach);
                    uint64_t accmask = __ballot(acc);
                    bool improving_pick = false;
                    const uint32_t nconsumed = forager_chunk_cut<L>(p.forager, (uint32_t)p.limit, accepted, acc, sc,
                                                                    p.forager == FORAGER_FIRST_BEST_IMPROVING ? forager_best : cur, nvalid, improving_pick);
                    const bool consumed = lane < nconsumed;
                    if (annealing) sa_commit<L>(s_sa, p.sa, sach, nconsumed, lane);
                    acc = acc && consumed;
                    accmask = __ballot(acc);
                    if (accmask) {
                        if (improving_pick) {  // BestCandidate::replace by the candidate that ends the step
                            const int sel = (int)nconsumed - 1;
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)shfl_u64((uint64_t)sc.v[kk], sel);
                            best_m0 = __shfl(m0, sel);
                            best_m1 = __shfl(m1, sel);
                            best_kind = (int)__shfl(lf, sel);
                            if (TRACE && lane == 0) c.best_ti = c.trace_n + (uint64_t)sel;
                            equal_count = 1;
                            has_best = 1;
                        } else if (p.forager == 1) {
                            if (!has_best) {
                                const int sel = __ffsll((unsigned long long)accmask) - 1;
#pragma unroll
                                for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)shfl_u64((uint64_t)sc.v[kk], sel);
                                best_m0 = __shfl(m0, sel);
                                best_m1 = __shfl(m1, sel);
                                best_kind = (int)__shfl(lf, sel);
                                if (TRACE && lane == 0) c.best_ti = c.trace_n + (uint64_t)sel;
                                has_best = 1;
                            }
                        } else {
                            const ScoreV<L> M = wave_max_score<L>(sc, acc);
                            const int cm = has_best ? score_cmp<L>(M, best) : 1;
                            if (cm >= 0) {
                                const bool newmax = cm > 0;
                                const uint64_t base = newmax ? 0 : equal_count;
                                const bool in_eq = acc && score_cmp<L>(sc, M) == 0;
                                const uint64_t eq = __ballot(in_eq);
                                const uint32_t rank = (uint32_t)__popcll(eq & lanemask_le(lane));
                                const uint64_t cntq = base + rank;
                                const bool pick = in_eq && ((newmax && rank == 1) ||
                                                            (p.random_ties && cntq > 1 && reservoir_pick(c.step_seed, cntq)));
                                const uint64_t pm = __ballot(pick);
                                if (pm) {
                                    const int sel = 63 - __clzll((unsigned long long)pm);
                                    best_m0 = __shfl(m0, sel);
                                    best_m1 = __shfl(m1, sel);
                                    best_kind = (int)__shfl(lf, sel);
                                    if (TRACE && lane == 0) c.best_ti = c.trace_n + (uint64_t)sel;
                                }
                                best = M;
                                equal_count = base + (uint64_t)__popcll(eq);
                                has_best = 1;
                            }
                        }
                    }
                    const uint32_t nacc = (uint32_t)__popcll(accmask);
                    accepted += nacc;
                    st_gen += nconsumed;
                    st_acc += nacc;
                    const uint32_t ndo = (uint32_t)__popcll(__ballot(consumed && doable));
                    st_calc += ndo;
                    st_nd += nconsumed - ndo;
                    if (tracing && consumed) {
                        const uint64_t ti = c.trace_n + lane;
                        if ((int64_t)ti < p.trace_cap) {
                            int32_t* tm = p.trace_moves + ti * 6;
                            tm[0] = p.leaf[lf].kind == 16 ? 2 : 3;
                            tm[1] = (int32_t)((m0 & 0x7FFFFFFFu) >> 16);
                            tm[2] = (int32_t)(m0 & 0xFFFFu);
                            tm[3] = (int32_t)(m1 >> 16);
                            tm[4] = (int32_t)(m1 & 0xFFFFu);
                            tm[5] = -1;
                            for (int kk = 0; kk < L && kk < m.levels; ++kk) p.trace_scores[ti * m.levels + kk] = doable ? sc.v[kk] : 0;
                            p.trace_flags[ti] = (doable ? 1 : 0) | (acc ? 2 : 0) | ((int32_t)lf << 8);
                        }
                    }
                    if (tracing) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) c.trace_n += nconsumed;
                        __builtin_amdgcn_wave_barrier();
                    }
                    const uint32_t c1 = (uint32_t)__popcll(__ballot(consumed && lf == 1u));
                    head1 += c1;
                    head0 += nconsumed - c1;
                    pulls += nconsumed;
                    if (forager_quits(p.forager, (uint32_t)p.limit, accepted, has_best, improving_pick)) {
                        done = 1;
                        break;
                    }
                }
                if (lane == 0) {
                    c.head[0] = head0;
                    c.head[1] = head1;
                    c.exhausted[0] = ex0;
                    if (p.n_leaves > 1) c.exhausted[1] = ex1;
                    c.pulls = pulls;
                    c.accepted = accepted;
                    c.has_best = has_best;
                    c.equal_count = equal_count;
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) c.best[kk] = best.v[kk];
                    c.best_m0 = best_m0;
                    c.best_m1 = best_m1;
                    c.best_kind = best_kind;
                    c.st[1] += st_gen;
                    c.st[2] += st_gen;
                    c.st[3] += st_acc;
                    c.st[5] += st_calc;
                    c.st[6] += st_nd;
                    c.done = done;
                }
            }
            __syncthreads();
            if (c.done) break;
        }

        // ---- (D) commit the forager's pick ----------------------------------------------
        const bool applied = c.has_best && !p.dry_run;
        if (applied) {
            const int kind = p.leaf[c.best_kind].kind == 16 ? 2 : 3;
            const uint32_t a = (c.best_m0 & 0x7FFFFFFFu) >> 16, i = c.best_m0 & 0xFFFFu;
            const uint32_t b = c.best_m1 >> 16, j = c.best_m1 & 0xFFFFu;
            if (tracing && tid == 0) {
                p.trace_applied[0] = 1;
                if ((int64_t)c.best_ti < p.trace_cap) p.trace_flags[c.best_ti] |= 4;  // Selected + Applied
                p.trace_applied[1] = kind;
                p.trace_applied[2] = (int32_t)a;
                p.trace_applied[3] = (int32_t)i;
                p.trace_applied[4] = (int32_t)b;
                p.trace_applied[5] = (int32_t)j;
                p.trace_applied[6] = -1;
            }
            apply_list_move_block(m, s_visits, s_off, s_load, kind, a, i, b, j);
            // refresh node -> (route, position) for the two touched routes
            {
                const uint32_t oa = s_off[a], la = s_off[a + 1] - oa;
                const uint32_t ob = s_off[b], lb = s_off[b + 1] - ob;
                for (uint32_t t = tid; t < la + (a != b ? lb : 0u); t += blockDim.x) {
                    if (t < la)
                        node_slot[s_visits[oa + t]] = (a << 16) | t;
                    else
                        node_slot[s_visits[ob + (t - la)]] = (b << 16) | (t - la);
                }
            }
            if (tid == 0) {
#pragma unroll
                for (int kk = 0; kk < L; ++kk) c.cur[kk] = c.best[kk];
                c.st[4] += 1;
            }
            __syncthreads();
        } else if (tracing && tid == 0) {
            p.trace_applied[0] = 0;
        }
        if (!p.dry_run) {
            // update_best_solution (scope_progress.rs:89-107): clone on strict improvement
            bool improved = false;
            if (applied) {
                ScoreV<L> cs, bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    cs.v[kk] = c.cur[kk];
                    bs.v[kk] = c.best_sol[kk];
                }
                improved = score_cmp<L>(cs, bs) > 0;
            }
            __syncthreads();
            if (improved) {
                const uint32_t total = s_off[V];
                for (uint32_t t = tid; t < total; t += blockDim.x) m.best_visits[(size_t)r * m.n_cap + t] = s_visits[t];
                for (uint32_t t = tid; t <= (uint32_t)V; t += blockDim.x) m.best_off[(size_t)r * (V + 1) + t] = s_off[t];
            }
            __syncthreads();
            if (improved && tid == 0) {
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    c.best_sol[kk] = c.cur[kk];
                    m.best_score[(size_t)r * 4 + kk] = c.cur[kk];
                }
            }
            if (tid == 0) {
                // acceptor.step_ended(last_step_score) always (step.rs:216-221)
                if (p.acceptor == 1) {
                    const int idx = (int)((p.la_idx[r] + step) % p.la_size);
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) p.la_hist[((size_t)r * p.la_size + idx) * 4 + kk] = c.cur[kk];
                }
                c.st[0] += 1;
            }
            if (annealing && wave == 0) sa_step_ended(s_sa, p.sa, lane);
            __syncthreads();
            if (p.move_budget > 0 && (int64_t)c.st[1] >= p.move_budget) break;  // uniform: c.st lives in LDS, read after the barrier
        }
    }

    // ---- write back ------------------------------------------------------------------
    if (!p.dry_run) {
        if (annealing && wave == 0) sa_store(s_sa, p.sa, r, lane);
        const uint32_t total = s_off[V];
        for (uint32_t t = tid; t < total; t += blockDim.x) g_visits[t] = s_visits[t];
        for (uint32_t t = tid; t <= (uint32_t)V; t += blockDim.x) g_off[t] = s_off[t];
        for (uint32_t t = tid; t < (uint32_t)V; t += blockDim.x) g_load[t] = s_load[t];
        if (tid == 0) {
#pragma unroll
            for (int kk = 0; kk < L; ++kk) {
                g_score[kk] = c.cur[kk];
                p.last_step_score[(size_t)r * 4 + kk] = c.cur[kk];
            }
            const int64_t steps_run = (int64_t)c.st[0];  // a move budget can end the launch before n_steps
            p.la_idx[r] = (int32_t)((p.la_idx[r] + steps_run) % p.la_size);
            p.step_index[r] += (uint64_t)steps_run;
            p.seed_draws[r] += (uint64_t)steps_run;
        }
    }
    if (tid == 0) {
        if (!p.dry_run)
            for (int k = 0; k < SF_STATS_WORDS; ++k) p.stats[(size_t)r * SF_STATS_WORDS + k] += c.st[k];
        if (tracing) *p.trace_count = (int64_t)c.trace_n;
    }
}

}  // namespace sf
