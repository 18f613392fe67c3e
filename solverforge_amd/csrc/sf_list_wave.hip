// Wave-per-replica search engine (gfx950 / CDNA4, wave64) for the list-variable hot path.
//
// One 64-lane wavefront owns one search replica for the whole launch: its routes (flat CSR),
// per-route loads, node -> (route, position) table, per-leaf entity-order tables and the
// per-leaf candidate rings live in that wave's private slice of LDS.  Everything inside a
// replica is wave-synchronous (ballot / shuffle / prefix-scan, LDS in program order), so the
// kernel contains NO workgroup barrier; the CU hides a wave's memory latency behind the other
// replicas resident on it.  Same reference semantics as the block engine in
// sf_list_kernels.hip (citations there); what differs is the GPU formulation:
//
//  * generation reads a PRESORTED neighbour index (k_nbr_presort: every matrix row sorted by
//    (distance, node) once at sf_initialize, the matrix is an immutable problem fact) instead
//    of scanning the whole matrix row per source: the stable bounded top-k of
//    nearby_list_support.rs:3-34 is then "walk the row in distance order until max_nearby
//    valid destinations are complete", usually one 64-entry chunk;
//  * the (distance, enumeration ordinal) order inside a chunk is recovered by counting
//    inversions inside equal-distance groups (lanes are already distance-sorted), not by
//    serial insertion;
//  * trial scoring happens at replay time with all 64 lanes busy (lane i = i-th candidate of
//    the union cursor order), so the rings only hold move coordinates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sf_list_model.h"

namespace sf {

constexpr uint32_t RC_MAX = 128;   // ring capacity per leaf when max_nearby > 32 (>= 64 + 64 - 1)
constexpr uint32_t RC_SMALL = 64;  // ring capacity per leaf when max_nearby <= 32 (>= 32 + 32)
constexpr uint32_t NBR_NODE_MASK = 0x7FFFu;  // node id (dim <= 16384 in the wave engine)
constexpr uint32_t NBR_SAME_FLAG = 0x8000u;  // same distance as the previous entry of the row
constexpr uint32_t NBR_END = 0xFFFFu;        // past the finite entries of the row
#ifndef SF_WPB
#define SF_WPB 4
#endif
constexpr int WPB = SF_WPB;  // waves (= replicas) per workgroup


// wavefront-scope ordering of LDS/global traffic between lanes of one wave (no instruction cost:
// a wave executes in lockstep and its DS/VMEM queues are in order; this pins the compiler).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | (uint64_t)uni((uint32_t)v);
}

// ---------------------------------------------------------------------------------------
// Neighbour index: bitonic sort of every matrix row in LDS.  grid = dim rows.
// ---------------------------------------------------------------------------------------
SF_PLAIN_KERNEL
__global__ __launch_bounds__(256) void k_nbr_presort(const int64_t* __restrict__ mat, int dim, int P,
                                                     uint16_t* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* sk = (uint64_t*)smem;
    const int row = blockIdx.x;
    const int64_t* rp = mat + (size_t)row * dim;
    for (int t = threadIdx.x; t < P; t += blockDim.x) {
        uint64_t k = ~0ULL;
        if (t < dim) {
            const int64_t v = rp[t];
            if (v >= 0 && v != UNREACHABLE) k = ((uint64_t)v << 24) | (uint64_t)(uint32_t)t;  // finite_distance (problem_data.rs:44-47)
        }
        sk[t] = k;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int lo = (t / stride) * 2 * stride + (t % stride);
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = sk[lo], b = sk[hi];
                if ((a > b) == up) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
            __syncthreads();
        }
    for (int t = threadIdx.x; t < dim; t += blockDim.x) {
        const uint64_t k = sk[t];
        uint32_t e = NBR_END;
        if (k != ~0ULL) {
            e = (uint32_t)k & NBR_NODE_MASK;
            if (t > 0 && (sk[t - 1] >> 24) == (k >> 24)) e |= NBR_SAME_FLAG;
        }
        keys[(size_t)row * dim + t] = (uint16_t)e;
    }
}

// LDS carve of ONE replica (bytes); mirrored on the host.
struct WCarve {
    size_t load, off, node, ring, visits, rtab, routeat, total;
    uint32_t rc;  // ring capacity per leaf
    // compact: the COMPACT instantiation's layout -- per-list loads as int32 (MODE 2 guarantees the range) and the node -> slot
    // table as 16 bits per node (NodeSlotT<true>): CVRP-5000 / 500 drops from 43 KB to 31 KB per replica (3 -> 5 replicas per CU)
    // node_global: the node -> slot table lives in HBM (ListModel::node_tab), not in the slice
    __host__ __device__ WCarve(int V, int n_cap, int dim, int max_k, bool compact = false, bool node_global = false) {
        rc = max_k <= 32 ? RC_SMALL : RC_MAX;
        size_t o = 0;
        load = o;
        o = align_up(o + (compact ? sizeof(int32_t) : sizeof(int64_t)) * V, 16);
        off = o;
        o = align_up(o + (node_global ? sizeof(uint16_t) : sizeof(uint32_t)) * (V + 1), 16);  // (node_global layout: 16-bit offsets, n_cap <= 65535)
        node = o;
        o = align_up(o + (node_global ? 0 : (compact ? sizeof(uint16_t) : sizeof(uint32_t)) * dim), 16);
        ring = o;
        o = align_up(o + sizeof(uint32_t) * 2 * rc * MAX_LEAVES, 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        rtab = o;  // [leaf][route] u32: rank of the route in the leaf's entity order | first destination slot ordinal << 16
        o = align_up(o + (node_global ? 0 : sizeof(uint32_t) * V * MAX_LEAVES), 16);  // (the node_global layout computes ranks instead: RouteArith)
        routeat = o;  // [leaf][rank] -> route; the compact layout recomputes it from the leaf's permutation parameters instead
        o = align_up(o + (compact ? 0 : sizeof(uint16_t) * V * MAX_LEAVES), 16);
        total = o;
    }
};

// node -> (route << 16 | position) of the replica's current lists.  Wide: one u32 per node.  Compact: one u16 per node,
// route in the top `rb` bits and the position below; a position that does not fit the field is stored as the field's maximum
// and found by scanning the route from there (routes that long are rare: the field holds 127 positions at 500 routes, 511 at
// 100); the all-ones pattern means "in no list".  get() returns the wide form either way.
template <bool COMPACT>
struct NodeSlotT;
template <>
struct NodeSlotT<false> {
    uint32_t* p;
    __device__ __forceinline__ NodeSlotT(unsigned char* base, int, const uint16_t*, const void*, bool = false) : p((uint32_t*)base) {}
    __device__ __forceinline__ void clear(uint32_t node) const { p[node] = NODE_NONE; }
    __device__ __forceinline__ void set(uint32_t node, uint32_t route, uint32_t pos) const { p[node] = (route << 16) | pos; }
    __device__ __forceinline__ uint32_t get(uint32_t node) const { return p[node]; }
};
template <>
struct NodeSlotT<true> {
    uint16_t* p;
    const uint16_t* visits;
    const void* off;  // uint32_t[V + 1], or uint16_t[V + 1] when off16
    bool off16;
    uint32_t pb, pmax;  // position bits, the saturated position
    __device__ __forceinline__ uint32_t off_at(uint32_t i) const { return off16 ? (uint32_t)((const uint16_t*)off)[i] : ((const uint32_t*)off)[i]; }
    __device__ __forceinline__ NodeSlotT(unsigned char* base, int V, const uint16_t* visits_, const void* off_, bool off16_ = false)
        : p((uint16_t*)base), visits(visits_), off(off_), off16(off16_) {
        uint32_t rb = 1;
        while ((1u << rb) - 1u < (uint32_t)V) ++rb;  // route ids 0 .. V - 1, the all-ones route is "none"
        pb = 16u - rb;
        pmax = (1u << pb) - 1u;
    }
    __device__ __forceinline__ void clear(uint32_t node) const { p[node] = 0xFFFFu; }
    __device__ __forceinline__ void set(uint32_t node, uint32_t route, uint32_t pos) const {
        p[node] = (uint16_t)((route << pb) | (pos < pmax ? pos : pmax));
    }
    __device__ __forceinline__ uint32_t get(uint32_t node) const {
        const uint32_t s = p[node];
        const bool none = s == 0xFFFFu;
        const uint32_t route = s >> pb;
        uint32_t pos = s & pmax;
        // saturated: the node sits at position >= pmax of its route.  Rare (routes that long), so the test is ONE wave-uniform branch
        // on a ballot and the common path has no exec-mask region at all
        if (__ballot(!none && pos == pmax) != 0ull) {
            if (!none && pos == pmax) {
                const uint32_t o = off_at(route), len = off_at(route + 1) - o;
                while (pos + 1 < len && (uint32_t)visits[o + pos] != node) ++pos;
            }
        }
        return none ? NODE_NONE : ((route << 16) | pos);
    }
};
// does the compact layout carry this model?  route ids need a bit pattern below all-ones, positions at least 6 bits
__host__ __device__ inline bool node_slot_compact_ok(int V) { return V >= 1 && V <= 1022; }

// One neighbour-row entry seen from source (se, sp) of a leaf: up to two consecutive keys
// (destination slot dp, and the end slot of its route when the node is the last element).
struct NearbyItem {
    uint32_t w;     // number of valid keys (0, 1, 2)
    uint32_t ord;   // enumeration ordinal of the first valid key (the second is ord + 1)
    uint32_t pay0;  // (route << 16 | position) of the first valid key
    uint32_t pay1;  // second key
};

__device__ __forceinline__ NearbyItem nearby_item(bool is_change, uint32_t slot, uint32_t se, uint32_t sp,
                                                  uint32_t len, uint32_t k, const uint32_t* s_off,
                                                  const uint16_t* sb, const uint16_t* ro) {
    // branch-free: every lane reads its route's length / rank / slot base (route 0 for unassigned nodes)
    const bool some = slot != NODE_NONE;
    const uint32_t r2 = some ? slot >> 16 : 0u, dp = slot & 0xFFFFu;
    const uint32_t len2 = s_off[r2 + 1] - s_off[r2];
    const uint32_t rk = ro[r2];
    const uint32_t inter_ord = ORD_INTER_BASE + (uint32_t)sb[rk] + dp;
    const bool intra = r2 == se;
    const uint32_t end_pay = (r2 << 16) | len2;
    NearbyItem it;
    if (is_change) {  // nearby_change.rs:133-195
        const bool v0 = !intra || (dp != sp && dp != sp + 1);
        const bool v1 = dp + 1 == len2 && (!intra || len != sp + 1);  // end slot `len` probes element len-1
        it.w = some ? (uint32_t)v0 + (uint32_t)v1 : 0u;
        it.ord = intra ? (v0 ? dp : len) : inter_ord;
        it.pay0 = v0 ? slot : end_pay;
        it.pay1 = end_pay;
    } else {  // nearby_swap.rs: intra partners after the source, inter only higher-ranked entities
        const bool v = intra ? dp > sp : rk > k;
        it.w = (some && v) ? 1u : 0u;
        it.ord = intra ? dp : inter_ord;
        it.pay0 = slot;
        it.pay1 = 0;
    }
    return it;
}

// Same item with the per-route facts read from ONE packed table (wave engine): rt[route] = rank of the route in the
// leaf's entity order | ordinal of the route's first destination slot << 16 — one LDS gather instead of the dependent
// pair rank_of[route] -> slot_base[rank].
// The per-route facts of a leaf's entity order as a table (RouteTab: the word above) or as arithmetic (RouteArith, the wave kernel whose tables live in HBM):
// the entity order is the permutation rank -> (start + rank x stride) mod V, so the rank of a route is ((route - start) x stride^-1) mod V
// -- a handful of vector instructions instead of a table that is rebuilt every step (and, in the layout whose tables live in HBM, instead
// of a dependent global gather per row entry).  The enumeration ordinal of an inter-list slot only ever breaks ties inside one
// equal-distance group, i.e. it is compared, never added up: (rank, position) compares exactly like first-slot-ordinal-of-rank + position,
// so the arithmetic form needs no prefix sum over the lists either.
struct RouteTab {
    const uint32_t* rt;
    static constexpr int KEY_SHIFT = 24;  // serial top-k key = distance << KEY_SHIFT | ordinal
    __device__ __forceinline__ uint32_t word(uint32_t route) const { return rt[route]; }
    __device__ __forceinline__ static uint32_t rank(uint32_t w) { return w & 0xFFFFu; }
    __device__ __forceinline__ static uint32_t inter_ord(uint32_t w, uint32_t dp) { return ORD_INTER_BASE + (w >> 16) + dp; }
};
constexpr uint32_t ORD_ARITH_BASE = 1u << 26;  // above every intra-list ordinal (a position, < 2^16); rank << 16 | position stays below 2^27
struct RouteArith {
    uint32_t st, inv, V, recip;  // permutation start, stride^-1 mod V, V, floor(2^32 / V) (V <= 1022: every product below stays under 2^21)
    static constexpr int KEY_SHIFT = 28;  // (the legs of a COMPACT kernel are < 2^26)
    __device__ __forceinline__ uint32_t word(uint32_t route) const {
        uint32_t x = route + V - st;
        x = x >= V ? x - V : x;
        const uint32_t y = x * inv;
        uint32_t rk = y - __umulhi(y, recip) * V;  // 32-bit Barrett step: the quotient estimate is exact or one short
        return rk >= V ? rk - V : rk;
    }
    __device__ __forceinline__ static uint32_t rank(uint32_t w) { return w; }
    __device__ __forceinline__ static uint32_t inter_ord(uint32_t w, uint32_t dp) { return ORD_ARITH_BASE + (w << 16) + dp; }
};
template <class OT, class RT>
__device__ __forceinline__ NearbyItem nearby_item_rt(bool is_change, uint32_t slot, uint32_t se, uint32_t sp,
                                                     uint32_t len, uint32_t k, const OT* s_off, const RT& rt) {
    const bool some = slot != NODE_NONE;
    const uint32_t r2 = some ? slot >> 16 : 0u, dp = slot & 0xFFFFu;
    const uint32_t len2 = s_off[r2 + 1] - s_off[r2];
    const uint32_t t = rt.word(r2);
    const uint32_t rk = RT::rank(t);
    const uint32_t inter_ord = RT::inter_ord(t, dp);
    const bool intra = r2 == se;
    const uint32_t end_pay = (r2 << 16) | len2;
    NearbyItem it;
    if (is_change) {  // nearby_change.rs:133-195
        const bool v0 = !intra || (dp != sp && dp != sp + 1);
        const bool v1 = dp + 1 == len2 && (!intra || len != sp + 1);  // end slot `len` probes element len-1
        it.w = some ? (uint32_t)v0 + (uint32_t)v1 : 0u;
        it.ord = intra ? (v0 ? dp : len) : inter_ord;
        it.pay0 = v0 ? slot : end_pay;
        it.pay1 = end_pay;
    } else {  // nearby_swap.rs: intra partners after the source, inter only higher-ranked entities
        const bool v = intra ? dp > sp : rk > k;
        it.w = (some && v) ? 1u : 0u;
        it.ord = intra ? dp : inter_ord;
        it.pay0 = slot;
        it.pay1 = 0;
    }
    return it;
}

// number of set bits of `mask` below this lane (v_mbcnt: no LDS crossbar involved)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Committed move application on the wave's LDS state (ListChange / ListSwap do_move).
template <class LT, class OT>  // LT = the replica's per-list load type in LDS: int64_t, or int32_t in the COMPACT wave layout; OT = offsets, uint32_t or uint16_t
__device__ __forceinline__ void apply_list_move_wave(const ListModel& m, uint16_t* visits, OT* off,
                                                     LT* load, int kind, uint32_t a, uint32_t i, uint32_t b,
                                                     uint32_t j, uint32_t ext = 0) {
    const uint32_t lane = threadIdx.x & 63u;
    if (kind == 2) {
        const uint32_t P = off[a] + i, Q = off[b] + j;
        const uint32_t x = visits[P];
        wave_sync();
        if (P < Q) {  // [P, Q-2] <- t+1 ; Q-1 <- x   (ascending chunks: reads run ahead of writes)
            for (uint32_t t0 = P; t0 < Q; t0 += 64) {
                const uint32_t t = t0 + lane;
                uint32_t nv = 0;
                if (t < Q) nv = (t + 1 < Q) ? (uint32_t)visits[t + 1] : x;
                wave_sync();
                if (t < Q) visits[t] = (uint16_t)nv;
                wave_sync();
            }
        } else if (P > Q) {  // (Q, P] <- t-1 ; Q <- x   (descending chunks)
            for (uint32_t c0 = 0; c0 <= P - Q; c0 += 64) {
                const uint32_t dd = c0 + lane;
                const bool in = dd <= P - Q;
                const uint32_t t = P - (in ? dd : 0u);
                uint32_t nv = 0;
                if (in) nv = t > Q ? (uint32_t)visits[t - 1] : x;
                wave_sync();
                if (in) visits[t] = (uint16_t)nv;
                wave_sync();
            }
        }
        if (a != b) {
            for (uint32_t rr = lane; rr <= (uint32_t)m.V; rr += 64) {
                if (a < b && rr > a && rr <= b) off[rr] -= 1;
                if (a > b && rr > b && rr <= a) off[rr] += 1;
            }
            if (lane == 0 && m.demand) {
                const int64_t dx = (int64_t)m.demand[x];
                load[a] = (LT)wsub(load[a], dx);
                load[b] = (LT)wadd(load[b], dx);
            }
        }
    } else if (kind == 6) {  // sublist swap: [i, i + (ext & 0xFFFF)) of a <-> [j, j + (ext >> 16)) of b
        const uint32_t za = ext & 0xFFFFu, zb = ext >> 16;
        const uint32_t PA = off[a] + i, PB = off[b] + j;
        const bool a_first = PA < PB;
        const uint32_t PX = a_first ? PA : PB, zx = a_first ? za : zb, PY = a_first ? PB : PA, zy = a_first ? zb : za;
        const uint32_t ox = a_first ? a : b, oy = a_first ? b : a;
        int64_t da = 0, db = 0;
        if (a != b && m.demand) {
            for (uint32_t t = 0; t < za; ++t) da = wadd(da, (int64_t)m.demand[visits[PA + t]]);
            for (uint32_t t = 0; t < zb; ++t) db = wadd(db, (int64_t)m.demand[visits[PB + t]]);
        }
        wave_sync();
        relocate_flat_segment(visits, PY, zy, PX, lane, 64u, [] { wave_sync(); });            // Y X mid
        relocate_flat_segment(visits, PX + zy, zx, PY + zy, lane, 64u, [] { wave_sync(); });  // Y mid X
        if (a != b) {
            for (uint32_t rr = lane; rr <= (uint32_t)m.V; rr += 64)
                if (rr > ox && rr <= oy) off[rr] = off[rr] + zy - zx;
            if (lane == 0 && m.demand) {
                load[a] = (LT)wadd(wsub(load[a], da), db);
                load[b] = (LT)wadd(wsub(load[b], db), da);
            }
        }
    } else if (kind == 5) {  // sublist change: segment [i, ext) of list a -> list b at j
        const uint32_t z = ext - i, P = off[a] + i;
        const uint32_t Q = a != b ? off[b] + j : (j <= i ? off[a] + j : off[a] + j + z);
        int64_t dsum = 0;
        if (a != b && m.demand)
            for (uint32_t t = 0; t < z; ++t) dsum = wadd(dsum, (int64_t)m.demand[visits[P + t]]);
        wave_sync();
        relocate_flat_segment(visits, P, z, Q, lane, 64u, [] { wave_sync(); });
        if (a != b) {
            for (uint32_t rr = lane; rr <= (uint32_t)m.V; rr += 64) {
                if (a < b && rr > a && rr <= b) off[rr] -= z;
                if (a > b && rr > b && rr <= a) off[rr] += z;
            }
            if (lane == 0 && m.demand) {
                load[a] = (LT)wsub(load[a], dsum);
                load[b] = (LT)wadd(load[b], dsum);
            }
        }
    } else if (kind == 4) {  // reverse [i, j) of list a
        const uint32_t lo = off[a] + i, hi = off[a] + j;
        for (uint32_t t = lane; t < (hi - lo) / 2; t += 64) {
            const uint16_t x = visits[lo + t], y = visits[hi - 1 - t];
            visits[lo + t] = y;
            visits[hi - 1 - t] = x;
        }
    } else if (kind == 9) {  // permute the window [i, j) of list a by the ext-th permutation of its positions
        const uint32_t base = off[a] + i, size = j - i;
        const uint32_t perm = nth_permutation_nibbles(size, ext);
        uint32_t nv = 0;
        if (lane < size) nv = visits[base + ((perm >> (4u * lane)) & 15u)];
        wave_sync();
        if (lane < size) visits[base + lane] = (uint16_t)nv;
    } else if (kind == 7) {  // 3-opt: cuts i < b < j of list a (b carries the middle cut), pattern = ext
        const uint32_t base = off[a], c1 = i, c2 = b, c3 = j;
        const uint32_t mask = kopt_reverse_mask(ext);
        const bool rb = (mask >> 1) & 1u, rc = (mask >> 2) & 1u;
        auto reverse_range = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t t = lane; t < (hi - lo) / 2; t += 64) {
                const uint16_t x = visits[base + lo + t], y = visits[base + hi - 1 - t];
                visits[base + lo + t] = y;
                visits[base + hi - 1 - t] = x;
            }
            wave_sync();
        };
        if (!kopt_swaps_segments(ext)) {
            if (rb) reverse_range(c1, c2);
            if (rc) reverse_range(c2, c3);
        } else {
            const uint32_t zc = c3 - c2;
            reverse_range(c1, c3);
            if (!rc) reverse_range(c1, c1 + zc);
            if (!rb) reverse_range(c1 + zc, c3);
        }
    } else if (kind == 3) {
        if (lane == 0) {
            const uint32_t pa = off[a] + i, pb = off[b] + j;
            const uint16_t x = visits[pa], y = visits[pb];
            visits[pa] = y;
            visits[pb] = x;
            if (a != b && m.demand) {
                const int64_t dx = (int64_t)m.demand[x], dy = (int64_t)m.demand[y];
                load[a] = (LT)wadd(wsub(load[a], dx), dy);
                load[b] = (LT)wadd(wsub(load[b], dy), dx);
            }
        }
    }
    wave_sync();
}

// One nearby source into a leaf ring (free-function form of the wave engine's generation loop, used
// by the generic N-leaf engine): walks the presorted neighbour row of element `sx` from offset
// `base`, `key` = the 64 row entries at base; appends at ring position tl + emitted; returns the
// number of candidates appended so far.
__device__ __forceinline__ uint32_t nearby_source_to_ring(const ListModel& m, const NbrIndex& nb, bool is_change, uint32_t se,
                                                          uint32_t sp, uint32_t len, uint32_t k, uint32_t sx,
                                                          const uint32_t* node_slot, const uint32_t* s_off, const uint16_t* sb,
                                                          const uint16_t* ro, uint32_t* rq, uint32_t RCM, uint32_t tl,
                                                          uint32_t key, uint32_t base, uint32_t need, uint32_t emitted) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanebit = 1ULL << lane;
    const uint32_t dim = (uint32_t)m.dim;
    const uint16_t* rowk = nb.keys + (size_t)sx * dim;
    const uint32_t mv0 = (se << 16) | sp;
    for (;;) {
        const bool have = key != NBR_END;
        const uint64_t havemask = __ballot(have);
        if (havemask == 0) break;  // finite entries of the row exhausted
        NearbyItem it{0u, 0u, 0u, 0u};
        if (have) it = nearby_item(is_change, node_slot[key & NBR_NODE_MASK], se, sp, len, k, s_off, sb, ro);
        // equal-distance groups are contiguous lane ranges; the index marks entries that
        // continue the previous entry's distance
        const uint64_t startmask = __ballot(have && (lane == 0 || !(key & NBR_SAME_FLAG)));
        const bool more = havemask == ~0ULL && base + 64 < dim;
        const uint32_t fo = 63u - (uint32_t)__clzll((unsigned long long)startmask);  // first lane of the last group
        if (more && fo == 0) {
            // one distance group wider than a chunk (degenerate ties): exact serial
            // insertion top-k over the rest of the row.
            TopK tk{~0ULL, 0u, ~0ULL};
            for (uint32_t b2 = base; b2 < dim; b2 += 64) {
                const uint32_t j2 = b2 + lane;
                const uint32_t ky = j2 < dim ? (uint32_t)rowk[j2] : NBR_END;
                NearbyItem i2{0u, 0u, 0u, 0u};
                uint64_t hk = 0;
                if (ky != NBR_END) {
                    const uint32_t y2 = ky & NBR_NODE_MASK;
                    i2 = nearby_item(is_change, node_slot[y2], se, sp, len, k, s_off, sb, ro);
                    hk = (uint64_t)m.mat[(size_t)sx * dim + y2] << 24;  // finite by construction of the index
                }
                if (!__ballot(ky != NBR_END)) break;
                topk_offer(tk, need, i2.w >= 1 ? (hk | i2.ord) : ~0ULL, i2.pay0);
                topk_offer(tk, need, i2.w == 2 ? (hk | (i2.ord + 1)) : ~0ULL, i2.pay1);
            }
            const uint32_t cnt = (uint32_t)__popcll(__ballot(lane < need && tk.key != ~0ULL));
            if (lane < cnt) {
                const uint32_t qi = (tl + emitted + lane) & RCM;
                rq[qi * 2] = mv0;
                rq[qi * 2 + 1] = tk.pay;
            }
            emitted += cnt;
            break;
        }
        const uint32_t nclosed = more ? fo : (uint32_t)__popcll(havemask);  // lanes [0, nclosed): complete groups
        const uint64_t closedmask = nclosed >= 64 ? ~0ULL : ((1ULL << nclosed) - 1ULL);
        const bool closed = (closedmask & lanebit) != 0;
        const uint32_t wc = closed ? it.w : 0u;
        const uint64_t w1 = __ballot(wc == 1), w2 = __ballot(wc == 2);
        const uint32_t Wc = (uint32_t)__popcll(w1) + 2u * (uint32_t)__popcll(w2);
        if (Wc > 0) {
            // sorted position = weight before me, corrected by the inversions inside my group
            int32_t pos = (int32_t)(mbcnt64(w1) + 2u * mbcnt64(w2));
            const uint64_t nonstart = ~startmask & closedmask;  // lane continues the group of lane-1
            if (nonstart) {
                const uint32_t packed = (it.ord << 2) | wc;
                uint64_t run = nonstart;  // bit t: lanes t-d .. t are one group
                for (uint32_t d = 1; run != 0; ++d) {
                    const uint32_t p_dn = __shfl_down(packed, d), p_up = __shfl_up(packed, d);
                    const bool same_up = (run & lanebit) != 0;
                    const bool same_dn = ((run >> d) & lanebit) != 0;
                    if (same_dn && (p_dn >> 2) < it.ord) pos += (int32_t)(p_dn & 3u);
                    if (same_up && (p_up >> 2) > it.ord) pos -= (int32_t)(p_up & 3u);
                    run &= nonstart << d;
                }
            }
            if (wc >= 1 && (uint32_t)pos < need) {
                const uint32_t qi = (tl + emitted + (uint32_t)pos) & RCM;
                rq[qi * 2] = mv0;
                rq[qi * 2 + 1] = it.pay0;
            }
            if (wc == 2 && (uint32_t)pos + 1 < need) {
                const uint32_t qi = (tl + emitted + (uint32_t)pos + 1) & RCM;
                rq[qi * 2] = mv0;
                rq[qi * 2 + 1] = it.pay1;
            }
            const uint32_t ne = Wc < need ? Wc : need;
            emitted += ne;
            need -= ne;
        }
        if (need == 0 || !more) break;
        base += nclosed;
        const uint32_t jj = base + lane;
        key = jj < dim ? (uint32_t)rowk[jj] : NBR_END;
    }
    return emitted;
}

#if defined(SF_PHASE_PROFILE) || defined(SF_GEN_COUNT)
__device__ unsigned long long g_phase[8];
#endif
// analysis builds (-DSF_ISA_MARK): comments in the generated assembly that delimit the hot regions, for static instruction counts per region
#ifdef SF_ISA_MARK
#define ISA_MARK(name) asm volatile("; SF_MARK " name)
#else
#define ISA_MARK(name)
#endif
#if defined(SF_PHASE_PROFILE) && !defined(SF_PHASE_PGRP)  // (SF_PHASE_PGRP: the slots belong to the stages of prec_eval_grouped, sf_prec_group.h)
#define PH_DECL uint64_t ph_t = clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i)                          \
    {                                  \
        const uint64_t _t = clock64(); \
        ph_acc[i] += _t - ph_t;        \
        ph_t = _t;                     \
    }
#define PH_DUMP \
    if (lane == 0) for (int _k = 0; _k < 8; ++_k) atomicAdd(&g_phase[_k], (unsigned long long)ph_acc[_k]);
#elif defined(SF_GEN_COUNT)  // event counts of the generation instead of clocks (scripts/gen_count.py): 0 paired passes, 1 / 2 single-source passes of leaf 0 / 1,
// 3 / 4 gen_rest calls out of a paired pass for leaf 0 / 1, 5 / 6 gen_rest chunk iterations of leaf 0 / 1, 7 iterations that found no key
#define PH_DECL uint64_t ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i)
#define GC(i) ph_acc[i] += 1;
#define PH_DUMP \
    if (lane == 0) for (int _k = 0; _k < 8; ++_k) atomicAdd(&g_phase[_k], (unsigned long long)ph_acc[_k]);
#else
#define PH_DECL
#define PH(i)
#define PH_DUMP
#endif
#ifndef GC
#define GC(i)
#endif

// ---- SMALL mode helpers (MODE 2) --------------------------------------------------------------------------------
// wave64 maximum of an int32 over the DPP network (row shifts + row broadcasts, like wave_incl_scan); all lanes active.
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
    const int lo = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
template <int L>
__device__ __forceinline__ bool small_ge(const int32_t (&a)[L], const int32_t (&b)[L]) {  // lexicographic a >= b
    bool ge = true;
#pragma unroll
    for (int k = L - 1; k >= 0; --k) ge = a[k] > b[k] || (a[k] == b[k] && ge);
    return ge;
}
template <int L>
__device__ __forceinline__ bool small_ge0(const int32_t (&a)[L]) {  // lexicographic a >= 0
    bool ge = true;
#pragma unroll
    for (int k = L - 1; k >= 0; --k) ge = a[k] > 0 || (a[k] == 0 && ge);
    return ge;
}
__device__ __forceinline__ int32_t clamp_i64_to_i32(int64_t v) {
    return v > 0x7FFFFFFFll ? 0x7FFFFFFF : v < -0x7FFFFFFFll - 1 ? (int32_t)0x80000000 : (int32_t)v;
}

// Trial delta of a ListChange / ListSwap candidate in 32-bit arithmetic: the same at-most-eight matrix legs as
// eval_list_move_legs, laid out as four "plus" and four "minus" legs so no lane multiplies by a sign, gathered from
// the compact u32 matrix through 32-bit byte offsets; every leg finite (host-checked), sums < 2^30.
// dv[k] = change of score level k.  Returns doable.
template <int L, class LT, bool M16 = false, class OT = uint32_t>
__device__ __forceinline__ bool eval_list_move_small(const ListModel& m, const uint16_t* visits, const OT* off, const LT* load,
                                                     bool chg, uint32_t a, uint32_t i, uint32_t b, uint32_t j, int32_t (&dv)[L]) {
    // Branch-free on purpose: every neighbour is read from a position that always exists (the source position when the real one does
    // not) and replaced by the depot with a select afterwards, the doability tests are one predicate at the end.  Conditional LDS reads
    // cost an exec-mask save / restore pair on the scalar unit each, and the scalar unit -- shared by the 20 waves of a CU -- is the
    // busiest pipe of this kernel.
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    const uint32_t ob = off[b], lb = off[b + 1] - ob;
    const bool intra = a == b;
    // move/list_kernel/change.rs:44-71, swap.rs:30-56
    bool ok = i < la && (chg ? j <= lb : j < lb) && !(intra && (j == i || (chg && j == i + 1)));
    const bool flip = !chg && intra && i > j;
    // A lane whose candidate is not doable prices the empty pattern: positions and lengths 0, so every neighbour predicate is false,
    // every LDS read goes to its list's first slot and every matrix row is the depot's -- nothing is gathered through a value that was
    // not validated (the selects are vector ops; the scalar unit sees none of this).
    const uint32_t i2 = ok ? (flip ? j : i) : 0u, j2 = ok ? (flip ? i : j) : 0u;
    const uint32_t la_ = ok ? la : 0u, lb_ = ok ? lb : 0u;
    const uint32_t depot = (uint32_t)m.depot;
    const uint32_t P = oa + i2, Q = ob + j2;
    const bool has_pa = i2 > 0, has_na = i2 + 1 < la_, has_q = j2 < lb_, has_pb = j2 > 0, has_nb = j2 + 1 < lb_;
    const uint32_t r_x = visits[P];
    const uint32_t x = ok ? r_x : depot;
    const uint32_t r_pa = visits[has_pa ? P - 1 : P], r_na = visits[has_na ? P + 1 : P];
    const uint32_t r_q = visits[has_q ? Q : P], r_pb = visits[has_pb ? Q - 1 : P], r_nb = visits[has_nb ? Q + 1 : P];
    const uint32_t pa = has_pa ? r_pa : depot, na = has_na ? r_na : depot;
    const uint32_t vq = has_q ? r_q : depot;  // change: right neighbour of the slot; swap: y
    const uint32_t pb = has_pb ? r_pb : depot, nb = has_nb ? r_nb : depot;
    ok = ok && (chg || x != vq);
    const bool adj = !chg && intra && j2 == i2 + 1;
    const bool ca = chg || adj;
    const bool src_single = la == 1, dst_empty = !intra && lb == 0;
    // plus legs P0..P3, minus legs M0..M3 (change | swap | adjacent swap):
    //   P0 (pa,na)* | (pa,y) | (pa,y)      M0 (pa,x)  | (pa,x) | (pa,x)
    //   P1 (pb,x)   | (y,na) | (y,x)       M1 (x,na)  | (x,na) | (x,y)
    //   P2 (x,vq)   | (pb,x) | (x,nb)      M2 (pb,vq)*| (pb,y) | (y,nb)
    //   P3   -      | (x,nb) |   -         M3   -     | (y,nb) |   -          (* absent for a single-element source /
    //                                                                            an empty destination route)
    const uint32_t dim4 = (uint32_t)m.dim * (M16 ? 2u : 4u);
    auto leg = [&](uint32_t f, uint32_t t) -> uint32_t {
        const uint32_t byte_off = f * dim4 + t * (M16 ? 2u : 4u);  // dim <= 16384: < 2^30
        if (M16) return *(const uint16_t*)((const char*)m.mat16 + byte_off);  // every leg finite (MODE 2) and < 65535
        return *(const uint32_t*)((const char*)m.mat32 + byte_off);
    };
    const uint32_t p0 = leg(pa, chg ? na : vq);
    const uint32_t p1 = leg(chg ? pb : vq, ca ? x : na);
    const uint32_t p2 = leg(ca ? x : pb, chg ? vq : (adj ? nb : x));
    const uint32_t p3 = leg(x, nb);
    const uint32_t m0 = leg(pa, x);
    const uint32_t m1 = leg(x, adj ? vq : na);
    const uint32_t m3 = leg(vq, nb);
    const uint32_t m2 = leg(adj ? vq : pb, adj ? nb : vq);
    int32_t d_cap = 0;
    if (m.cap_level >= 0) {
        const int32_t dx = m.demand[x];
        const int32_t dyq = m.demand[vq];  // read for every lane, used by the swaps
        const int32_t dy = chg ? 0 : dyq;
        const int32_t cap = (int32_t)m.capacity;
        const int32_t la0 = (int32_t)load[a], lb0 = (int32_t)load[b];
        const int32_t la1 = la0 - dx + dy, lb1 = lb0 - dy + dx;
        d_cap = max(la1 - cap, 0) + max(lb1 - cap, 0) - max(la0 - cap, 0) - max(lb0 - cap, 0);
        d_cap = intra ? 0 : d_cap;
    }
    const uint32_t plus = ((chg && src_single) ? 0u : p0) + p1 + p2 + (ca ? 0u : p3);
    const uint32_t minus = m0 + m1 + ((chg && dst_empty) ? 0u : m2) + (ca ? 0u : m3);
    const int32_t d_dist = (int32_t)(plus - minus);
    const int32_t cw = (int32_t)m.cap_weight, dw = (int32_t)m.dist_weight;
#pragma unroll
    for (int k = 0; k < L; ++k) {  // penalties: score level -= weight * delta(penalty sum)
        int32_t v = 0;
        if (k == m.cap_level) v -= cw * d_cap;
        if (k == m.dist_level) v -= dw * d_dist;
        dv[k] = ok ? v : 0;
    }
    return ok;
}

// The same trial delta for a candidate the nearby generators of THIS step emitted (the FAST kernels' rings hold nothing else): such a candidate
// is structurally doable by construction -- nearby_change.rs:133-195 / nearby_swap.rs enumerate destinations of the committed lists only, skip the
// source slot and its successor, and pair an intra swap with a LATER position -- so move/list_kernel/change.rs:44-71 / swap.rs:30-56 hold and
// nothing is tested here.  Written for the scalar unit's sake (the kernel's bound): a per-lane predicate is a v_cmp into an SGPR pair, every
// `&&` / `||` of two predicates one scalar instruction, every compare-then-select a wait state on gfx950; so neighbours are addressed with
// min() arithmetic, and `chg` arrives as an integer (1 = ListChange, 0 = ListSwap).
template <int L, class LT, bool M16 = false, class OT = uint32_t>
__device__ __forceinline__ void eval_generated_small(const ListModel& m, const uint16_t* visits, const OT* off, const LT* load, uint32_t c, uint32_t a,
                                                     uint32_t i, uint32_t b, uint32_t j, int32_t (&dv)[L]) {
    const uint32_t oa = off[a], la = off[a + 1] - oa;
    const uint32_t ob = off[b], lb = off[b + 1] - ob;
    const uint32_t depot = (uint32_t)m.depot;
    const uint32_t P = oa + i, Q = ob + j;
    // 1 when the neighbour exists (its position is then one step away; otherwise the read goes to the slot itself and the depot is selected)
    const uint32_t h_pa = min(i, 1u), h_na = min(la - 1u - i, 1u);      // i > 0 ; i + 1 < la   (i < la)
    const uint32_t h_q = min(lb - j, 1u), h_pb = min(j, 1u);            // j < lb ; j > 0       (j <= lb)
    const uint32_t h_nb = min(max(lb, j + 1u) - (j + 1u), 1u);          // j + 1 < lb
    const uint32_t x = visits[P];
    const uint32_t r_pa = visits[P - h_pa], r_na = visits[P + h_na];
    const uint32_t r_q = visits[Q - (1u - h_q)], r_pb = visits[Q - h_pb], r_nb = visits[Q + h_nb];  // (an end slot reads its left neighbour; unused)
    const uint32_t pa = h_pa ? r_pa : depot, na = h_na ? r_na : depot;
    const uint32_t vq = h_q ? r_q : depot;  // change: right neighbour of the slot; swap: y
    const uint32_t pb = h_pb ? r_pb : depot, nb = h_nb ? r_nb : depot;
    const bool chg = c != 0;
    const bool intra = a == b;
    const bool adj = !chg && intra && j == i + 1;
    const bool ca = chg || adj;
    const uint32_t dim4 = (uint32_t)m.dim * (M16 ? 2u : 4u);
    auto leg = [&](uint32_t f, uint32_t t) -> uint32_t {
        const uint32_t byte_off = __umul24(f, dim4) + t * (M16 ? 2u : 4u);  // node ids and the row pitch are below 2^24: one full-rate v_mad_u32_u24 (a 32-bit v_mul_lo is quarter rate); dim <= 16384: < 2^30
        if (M16) return *(const uint16_t*)((const char*)m.mat16 + byte_off);
        return *(const uint32_t*)((const char*)m.mat32 + byte_off);
    };
    // plus legs P0..P3, minus legs M0..M3: the table of eval_list_move_small
    const uint32_t p0 = leg(pa, chg ? na : vq);
    const uint32_t p1 = leg(chg ? pb : vq, ca ? x : na);
    const uint32_t p2 = leg(ca ? x : pb, chg ? vq : (adj ? nb : x));
    const uint32_t p3 = leg(x, nb);
    const uint32_t m0 = leg(pa, x);
    const uint32_t m1 = leg(x, adj ? vq : na);
    const uint32_t m3 = leg(vq, nb);
    const uint32_t m2 = leg(adj ? vq : pb, adj ? nb : vq);
    int32_t d_cap = 0;
    if (m.cap_level >= 0) {
        const int32_t dx = m.demand[x];
        const int32_t dyq = m.demand[vq];
        const int32_t dy = chg ? 0 : dyq;
        const int32_t cap = (int32_t)m.capacity;
        const int32_t la0 = (int32_t)load[a] - cap, lb0 = (int32_t)load[b] - cap;
        const int32_t sh = dx - dy;
        d_cap = max(la0 - sh, 0) + max(lb0 + sh, 0) - max(la0, 0) - max(lb0, 0);
        d_cap = intra ? 0 : d_cap;
    }
    const uint32_t plus = ((chg && la == 1) ? 0u : p0) + p1 + p2 + (ca ? 0u : p3);
    const uint32_t minus = m0 + m1 + ((chg && !intra && lb == 0) ? 0u : m2) + (ca ? 0u : m3);
    const int32_t d_dist = (int32_t)(plus - minus);
    // penalties: score level -= weight * delta(penalty sum); a level's share is picked with a 32-bit all-ones / zero word per (level, constraint) --
    // two wave-uniform words instead of a lane mask in a scalar register pair and a select each
    const int32_t tc = -((int32_t)m.cap_weight * d_cap), td = -((int32_t)m.dist_weight * d_dist);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t wc = k == m.cap_level ? 0xFFFFFFFFu : 0u, wd = k == m.dist_level ? 0xFFFFFFFFu : 0u;
        dv[k] = (int32_t)(((uint32_t)tc & wc) + ((uint32_t)td & wd));
    }
}

// Per-leaf cursor state of one step.  The NEXT source of the leaf is always resolved ahead of use
// and the first 64-entry chunk of its neighbour row is already in flight (`pk`), so the global
// load latency hides behind the other leaf's generation and the replay batches.
struct LeafCursor {
    uint32_t head, tail;  // candidate ring (monotonic counters)
    uint32_t left;        // sources not yet generated
    uint32_t k, o;        // next source: entity rank / offset inside the entity's list
    uint32_t se, len, sp, sx;  // next source resolved: entity, its list length, position, element
    uint32_t vk, vbase;        // entity rank / offset base the leaf's spvec holds (0xFFFFFFFF = none)
    int ex;               // exhausted (union scheduler)
    uint32_t pk;          // per lane: prefetched entry of chunk 0 of the next source's neighbour row
    uint32_t pv;          // per lane: (source position | source element << 16) of offset vbase + lane of entity rank vk
    uint32_t cend;        // offsets o < cend of the current entity are served by `pv` as it stands (0 = nothing cached): resolve()'s one-compare fast path
};

// MODE 1 (FAST): compile-time specialisation for the default list policy (nearby change + nearby swap union,
// LateAcceptance + AcceptedCount, committed steps) — fewer live scalars and branches in the hot loops.
// MODE 2 (FAST + SMALL): additionally every quantity of a trial delta fits 32 bits (host-checked: all matrix legs finite
// and < 2^26, demands / capacity / weights bounded so that |level delta| < 2^30): the replay scores, accepts and forages
// in DELTA space — a candidate is the int32 change of each score level against the step's (wave-uniform) current score,
// `score >= late` becomes `delta >= late - current` with the right-hand side clamped once per step — so no lane carries a
// 64-bit score vector; the committed score is advanced by the winner's delta.  Same decisions bit for bit.
#ifndef SF_WAVES_PER_EU
#define SF_WAVES_PER_EU 4
#endif
// COMPACT (with MODE 2 only): the replica's LDS slice in the compact layout of WCarve, chosen by the host when it lets more
// replicas share a CU (CVRP-5000: 5 instead of 3).
// WPE = waves per SIMD the kernel is compiled for (512 / WPE VGPRs): 4, or 5 / 6 for the COMPACT slice of a model small enough for 20 /
// 24 replicas per CU.  Round 4: with the replica index declared wave-uniform the per-replica base pointers live in scalar registers and
// the MODE 2 kernels need 77 VGPRs and no scratch (they were 96 VGPRs + 120 B: 64-bit pointer pairs spilled in the prologue).
// NODEG (with COMPACT only): the node -> slot table in HBM (ListModel::node_tab) instead of the slice.  A wave's speed does not depend on
// the model size (CVRP-1000 and CVRP-5000 both run 7.4 M moves/s per resident wave: the wave's own chain of dependent accesses is the
// bound), the throughput is the number of resident waves -- and at CVRP-5000 the 29 KB slice holds that at 5 per CU.
template <int L, bool TRACE, int MODE, bool COMPACT = false, int WPE = SF_WAVES_PER_EU, bool NODEG = false>
__global__ __launch_bounds__(64 * WPB, WPE) void k_list_search_wave(ListModel m, SearchParams p, NbrIndex nb) {
    constexpr bool FAST = MODE >= 1, SMALL = MODE == 2;
    static_assert(!COMPACT || SMALL, "COMPACT stores loads in 32 bits: MODE 2 only");
    static_assert(!NODEG || COMPACT, "NODEG: the 16-bit table of the COMPACT layout");
    using LT = typename std::conditional<COMPACT, int32_t, int64_t>::type;
    using OT = typename std::conditional<NODEG, uint16_t, uint32_t>::type;  // list offsets in the slice (NODEG: 16 bits, n_cap <= 65535 in this engine)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // wave-uniform by construction; saying so keeps every per-replica base pointer in scalar registers (they were 64-bit VGPR pairs
    // spilled to scratch in the prologue and reloaded for the write-back)
    const uint32_t wave_in_group = uni(threadIdx.x >> 6);
    const int rr = (int)(blockIdx.x * (blockDim.x >> 6) + wave_in_group);  // 1..WPB replicas per workgroup
    if (rr >= p.n_launch) return;  // no workgroup barrier anywhere below
    const int r = rr + p.replica_base;
    const int V = m.V;
    const uint32_t dim = (uint32_t)m.dim;

    // leaf constants (no dynamic indexing of the kernarg block)
    const int n_leaves = FAST ? 2 : p.n_leaves;
    const uint32_t K0 = (uint32_t)p.leaf[0].max_nearby, K1 = n_leaves > 1 ? (uint32_t)p.leaf[1].max_nearby : 1u;
    const bool chg0 = FAST ? true : p.leaf[0].kind == 16, chg1 = FAST ? false : (n_leaves > 1 && p.leaf[1].kind == 16);
    const int acceptor = FAST ? 1 : p.acceptor, forager = FAST ? 0 : p.forager;
    const bool dry_run = FAST ? false : p.dry_run != 0;
    __shared__ uint64_t s_sa[FAST ? 1 : WPB][FAST ? 1 : SA_WORDS];  // SimulatedAnnealing acceptor state of the resident replicas (FAST: LateAcceptance only, no static LDS -- the host's occupancy plan counts on that)
    uint64_t* saw = s_sa[FAST ? 0 : wave_in_group];
    const bool annealing = !FAST && acceptor == 3;
    if constexpr (!FAST)
        if (annealing) sa_load(saw, p.sa, r, lane);
    const uint64_t desc0 = (uint64_t)p.leaf[0].descriptor, desc1 = n_leaves > 1 ? (uint64_t)p.leaf[1].descriptor : 0;

    const WCarve cv(V, m.n_cap, m.dim, (int)(K0 > K1 ? K0 : K1), COMPACT, NODEG);
    const uint32_t RCM = cv.rc - 1;
    unsigned char* mem = smem + (size_t)wave_in_group * cv.total;
    LT* s_load = (LT*)(mem + cv.load);
    OT* s_off = (OT*)(mem + cv.off);
    uint32_t* ring = (uint32_t*)(mem + cv.ring);  // [leaf][rc][2]
    uint16_t* s_visits = (uint16_t*)(mem + cv.visits);
    const NodeSlotT<COMPACT> node_slot(NODEG ? (unsigned char*)(m.node_tab + (size_t)r * dim) : mem + cv.node, V, s_visits, (const void*)s_off, NODEG);
    // table writes -> table reads of other lanes: LDS in program order (wave_sync); HBM through the CU's write-through L1 (ring_sync's fences)
    auto node_sync = [&]() {
        if constexpr (NODEG) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            wave_sync();
        }
    };
    // [leaf][route] rank | first slot ordinal << 16 (RouteTab).  The NODEG layout computes the rank instead (RouteArith) and carries no table: in
    // round 5 its table lived in HBM, a dependent global gather per row entry behind the node -> slot one -- CVRP-5000 15.9 -> 17.9 G moves/s and
    // 322 -> 227 B of memory-side traffic per candidate without it.  With the table in LDS (CVRP-1000) the lookup is cheaper than the arithmetic
    // (50.4 -> 48.8 G when every COMPACT kernel computed ranks, profiles/r06d_route_arith_ab.txt), so those keep it.
    constexpr bool ARANK = NODEG;
    using RT = typename std::conditional<ARANK, RouteArith, RouteTab>::type;
    uint32_t* rtab = (uint32_t*)(mem + cv.rtab);
    uint16_t* route_at = (uint16_t*)(mem + cv.routeat);  // [leaf][rank] -> route

    uint32_t* g_visits = m.visits + (size_t)r * m.n_cap;
    uint32_t* g_off = m.off + (size_t)r * (V + 1);
    int64_t* g_load = m.load + (size_t)r * V;
    int64_t* g_score = m.score + (size_t)r * 4;
    const bool tracing = TRACE && r == p.trace_replica;

    // ---- load replica state into LDS ----
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) s_off[t] = (OT)g_off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) s_load[t] = (LT)g_load[t];
    for (uint32_t t = lane; t < dim; t += 64) node_slot.clear(t);
    node_sync();
    const uint32_t total0 = uni(s_off[V]);
    if (COMPACT && m.perm) {  // internal node numbering (ListModel::perm): the lists are renamed on the way in and on every way out
        for (uint32_t t = lane; t < total0; t += 64) s_visits[t] = m.perm[g_visits[t]];
    } else {
        for (uint32_t t = lane; t < total0; t += 64) s_visits[t] = (uint16_t)g_visits[t];
    }
    wave_sync();
    for (uint32_t v = lane; v < (uint32_t)V; v += 64) {
        const uint32_t o = s_off[v], len = s_off[v + 1] - o;
        for (uint32_t q = 0; q < len; ++q) node_slot.set(s_visits[o + q], v, q);
    }
    node_sync();

    auto ext_id = [&](uint32_t x) -> uint32_t { return (COMPACT && m.inv) ? (uint32_t)m.inv[x] : x; };  // an LDS element under the caller's numbering
    int64_t cur[L], best_sol[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        cur[k] = (int64_t)uni64((uint64_t)g_score[k]);  // wave-uniform: keep in scalar registers
        best_sol[k] = (int64_t)uni64((uint64_t)m.best_score[(size_t)r * 4 + k]);
    }
    // per-launch counters in 32 bits (wave-uniform: scalar registers), folded into the replica's 64-bit sf_stats words
    // before they can wrap (flush_stats): a long fixed-step launch never loses counts
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
    const uint64_t step_index0 = dry_run ? 0 : p.step_index[r];
    const uint64_t seed_draws0 = dry_run ? 0 : p.seed_draws[r];
    const int la_idx0 = dry_run ? 0 : p.la_idx[r];
    int la_cursor = la_idx0;  // (la_idx0 + step) % la_size, kept incrementally (no 64-bit division per step)
    const uint64_t lanebit = 1ULL << lane;
    PH_DECL

    bool best_pending = false;  // working == best, snapshot not yet written (see sf_scalar_kernels.hip: deferred clone)
    const FastMod fm_V = make_fastmod(V > 0 ? (uint32_t)V : 1u);
    const FastMod fm_V1 = make_fastmod(V > 1 ? (uint32_t)V - 1u : 1u);
    const uint32_t v_recip32 = V > 1 ? (uint32_t)(0x100000000ull / (uint32_t)V) : 0u;  // floor(2^32 / V): resolve()'s 32-bit remainder (V = 1: every rank maps to list 0)
    // coprimality of every candidate permutation stride of the V list owners, once per launch (lane s tests s and s + 64)
    const bool use_cm = V >= 2 && V <= 128;
    const uint64_t cm_lo = use_cm ? __ballot(lane >= 1 && lane < (uint32_t)V && gcd_u32(lane, (uint32_t)V) == 1) : 0ull;
    const uint64_t cm_hi = use_cm ? __ballot(lane + 64 < (uint32_t)V && gcd_u32(lane + 64, (uint32_t)V) == 1) : 0ull;
    for (int64_t step = 0; step < p.n_steps; ++step) {
        PH(7)
        // ---- (A) step start (step.rs:60-74) -------------------------------------------------
        uint64_t sidx, sseed;
        if (dry_run) {
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
        const StreamCtx ctx{sidx, sseed, FAST ? 3 : p.order};  // FAST: SelectionOrder::Random, the default policy's (host-checked)
        ScoreV<L> late;
#pragma unroll
        for (int k = 0; k < L; ++k) late.v[k] = 0;
        const int la_slot = la_cursor;  // LateAcceptance history slot of this step
        if (acceptor == 1 || acceptor == 4) {
#pragma unroll
            for (int k = 0; k < L; ++k) late.v[k] = (int64_t)uni64((uint64_t)p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + k]);
        }
        ScoreV<L> dla_thr = late;  // DiversifiedLateAcceptance: best step score of the phase minus its tolerance band
        if (acceptor == 4) {
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
        // MODE 2: the LateAcceptance threshold and the forager's best as level deltas against `cur` (wave-uniform);
        // |candidate delta| < 2^30, so clamping the threshold to int32 keeps every comparison exact
        int32_t late_d[L], best_d[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            late_d[k] = SMALL ? clamp_i64_to_i32(wsub(late.v[k], cur[k])) : 0;
            best_d[k] = 0;
        }
        // (L == 2) the same thresholds as one signed 64-bit key each: hard x 2^32 + soft
        int64_t acc_thr = 0, best_key = 0;
        if constexpr (SMALL && L == 2) {
            // (|candidate level delta| < 2^30: a hard threshold at or below -2^30 accepts every candidate whatever it is, so the clamp is exact and the
            // key cannot leave 64 bits)
            const int64_t lh = late_d[0] < -(1 << 30) ? -(int64_t)(1 << 30) : (int64_t)late_d[0];
            const int64_t late_key = lh * 4294967296ll + (int64_t)late_d[1];
            acc_thr = late_key < 0 ? late_key : 0;  // delta >= 0  ||  delta >= late - current
        }
        const uint32_t total = uni(s_off[V]);
        // union: >1 leaf => StratifiedRandom, equal weights (vec_union.rs:229-245); with two children
        // the stride is always 1, so the order is first, other, first, ...
        // The five hashes a FAST step starts with (the union's first child; start and stride of the two leaves' entity permutations) are
        // computed SIDE BY SIDE, one per lane, on the vector unit: as wave-uniform values they were ~400 scalar instructions per step (64-bit
        // multiplies and Barrett remainders expand to a dozen s_mul each), a tenth of the kernel's scalar work, and the scalar unit -- shared by
        // the 24 waves of a CU -- is its bound (profiles/r05_salu_fit.json: 1,042 scalar instructions per step before this).
        uint32_t hashed = 0;  // lane l < 5: remainder of mixed_seed(salt_l) by its divisor
        const bool lane_hash = FAST && use_cm;
        if (lane_hash) {
            const uint64_t es0 = SALT_NEARBY_CHANGE_ENTITY ^ desc0, es1 = SALT_NEARBY_SWAP_ENTITY ^ desc1;  // FAST: leaf 0 = nearby change, leaf 1 = nearby swap
            const uint64_t salt = lane == 0 ? SALT_UNION_OFFSET : lane == 1 ? es0 : lane == 2 ? (es0 ^ STRIDE_SALT_MIX) : lane == 3 ? es1 : (es1 ^ STRIDE_SALT_MIX);
            const FastMod fm2 = make_fastmod(2u);
            FastMod f;
            f.M = lane == 0 ? fm2.M : ((lane & 1u) ? fm_V.M : fm_V1.M);
            f.n = lane == 0 ? fm2.n : ((lane & 1u) ? fm_V.n : fm_V1.n);
            hashed = fastmod_u64(ctx.mixed_seed(salt), f);
        }
        const uint32_t first_leaf = lane_hash ? (uint32_t)__builtin_amdgcn_readlane((int)hashed, 0)
                                              : (n_leaves > 1 ? ctx.random_index((uint32_t)n_leaves, SALT_UNION_OFFSET) : 0u);

        uint32_t perm_st0 = 0, perm_sd0 = 1, perm_st1 = 0, perm_sd1 = 1;  // entity permutations of the two leaves (set in (B))
        uint32_t perm_inv0 = 0, perm_inv1 = 0;                              // ARANK: stride^-1 mod V of each
        auto route_facts = [&](int l) -> RT {
            if constexpr (ARANK)
                return RouteArith{l ? perm_st1 : perm_st0, l ? perm_inv1 : perm_inv0, (uint32_t)V, v_recip32};
            else
                return RouteTab{rtab + l * V};
        };
        // resolve the source at cursor (k, o) of leaf l and put its first key chunk in flight
        auto resolve = [&](LeafCursor& c, int l) {
            const uint16_t* ra = route_at + l * V;
            uint32_t k = c.k, o = c.o, se = c.se, len = c.len;
            // the common case -- the next offset of the same entity inside the 64 offsets `pv` already holds -- is ONE compare: everything below
            // this test is the per-entity / per-chunk path (it used to be entered through two compound tests, ~20 scalar instructions per source)
            if (o >= c.cend) {
            if (k != c.vk || o >= len) {  // a new entity: its list owner and length (skip empty routes; left > 0
                                          // guarantees a source exists)
                for (;;) {
                    if constexpr (COMPACT) {  // no rank -> route table in the compact slice: the permutation itself (a few entities per step)
                        // start + k x stride < V^2 + V < 2^21 (V <= 1022 in this layout): a 32-bit Barrett step -- the quotient estimate is exact or one
                        // short -- instead of the 64 x 64 high multiply of fastmod_u64 (two dozen scalar instructions)
                        const uint32_t xx = (l ? perm_st1 : perm_st0) + k * (l ? perm_sd1 : perm_sd0);
                        uint32_t rr_ = xx - __umulhi(xx, v_recip32) * (uint32_t)V;
                        rr_ = rr_ >= (uint32_t)V ? rr_ - (uint32_t)V : rr_;
                        se = uni(rr_);
                    } else
                        se = uni((uint32_t)ra[k]);
                    len = uni(s_off[se + 1] - s_off[se]);
                    if (o < len) break;
                    ++k;
                    o = 0;
                }
            }
            if (k != c.vk || (o & ~63u) != c.vbase) {
                // 64 consecutive offsets of this entity, one per lane: source position and source element, kept in a
                // register (`pv` = position | element << 16); a source is then one v_readlane away, no LDS round trip
                const uint64_t src_salt = ((l ? chg1 : chg0) ? SALT_NEARBY_CHANGE_SOURCE : SALT_NEARBY_SWAP_SOURCE) ^
                                          (uint64_t)se ^ (l ? desc1 : desc0);
                const uint32_t oo = (o & ~63u) + lane;
                const uint32_t spl = oo < len ? ctx.selection_index(oo, len, src_salt) : 0u;
                const uint32_t sxl = oo < len ? (uint32_t)s_visits[s_off[se] + spl] : 0u;
                c.pv = spl | (sxl << 16);
                c.vk = k;
                c.vbase = o & ~63u;
            }
            c.cend = len < (o & ~63u) + 64u ? len : (o & ~63u) + 64u;
            c.k = k;
            c.se = se;
            c.len = len;
            }
            const uint32_t pvv = (uint32_t)__builtin_amdgcn_readlane((int)c.pv, (int)(o & 63u));
            const uint32_t sp = pvv & 0xFFFFu, sx = pvv >> 16;
            c.o = o;
            c.sp = sp;
            c.sx = sx;
            const uint32_t ent = l ? ((lane + 32u) & 63u) : lane;  // leaf 1: lanes 32-63 hold entries 0-31
            // (the index has dim^2 <= 2^28 entries: a 32-bit byte offset per lane on the scalar base pointer, not a 64-bit scalar row address)
            // unconditional load (a lane past a row shorter than 64 entries reads the row's last entry and is then blanked): no exec-mask region
            const uint32_t entc = ent < dim ? ent : dim - 1u;
            const uint32_t pk_any = (uint32_t) * (const uint16_t*)((const char*)nb.keys + ((sx * dim + entc) << 1));
            c.pk = ent < dim ? pk_any : NBR_END;
        };

        PH(0)
        // ---- (B) per-leaf entity order tables (slot.rs:468-499) --------------------------------
        LeafCursor C0{0, 0, total, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0, 0, NBR_END, 0, 0};
        LeafCursor C1{0, 0, n_leaves > 1 ? total : 0u, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0, n_leaves > 1 ? 0 : 1, NBR_END, 0, 0};
        for (int l = 0; l < n_leaves; ++l) {
            const uint64_t ent_salt = ((l ? chg1 : chg0) ? SALT_NEARBY_CHANGE_ENTITY : SALT_NEARBY_SWAP_ENTITY) ^ (l ? desc1 : desc0);
            uint32_t pst, psd;
            if (lane_hash) {  // perm_params_fm with the two hashes already taken (above)
                pst = (uint32_t)__builtin_amdgcn_readlane((int)hashed, l ? 3 : 1);
                psd = StreamCtx::first_coprime_from((uint32_t)__builtin_amdgcn_readlane((int)hashed, l ? 4 : 2) + 1u, cm_lo, cm_hi);
            } else if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, ent_salt, pst, psd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, ent_salt, pst, psd);
            pst = uni(pst);
            psd = uni(psd);
            if (l)
                perm_st1 = pst, perm_sd1 = psd;
            else
                perm_st0 = pst, perm_sd0 = psd;
            if constexpr (ARANK) {
                // stride^-1 mod V (the stride is coprime to V): lane c tests c, c + 64, ... -- one or two rounds at 100 lists, eight at 500
                uint32_t inv = 0;
                for (uint32_t base = 0; base < (uint32_t)V; base += 64) {
                    const uint32_t c = base + lane;
                    const uint32_t xx = psd * c;  // < V^2 + 64 V < 2^21
                    uint32_t rm = xx - __umulhi(xx, v_recip32) * (uint32_t)V;
                    rm = rm >= (uint32_t)V ? rm - (uint32_t)V : rm;
                    const uint64_t hit = __ballot(c < (uint32_t)V && rm == 1u);
                    if (hit) {
                        inv = base + (uint32_t)__ffsll((unsigned long long)hit) - 1u;
                        break;
                    }
                }
                if (l)
                    perm_inv1 = inv;
                else
                    perm_inv0 = inv;
                continue;  // no tables: resolve() and RouteArith work from (start, stride, stride^-1)
            }
            uint16_t* ra = route_at + l * V;
            uint32_t* rt = rtab + l * V;
            uint32_t carry = 0;  // first slot ordinal of rank k = sum_{k'<k} (len(route_at[k']) + 1)
            for (uint32_t base = 0; base < (uint32_t)V; base += 64) {
                const uint32_t k = base + lane;
                uint32_t v = 0, e = 0;
                if (k < (uint32_t)V) {
                    e = fastmod_u64((uint64_t)pst + (uint64_t)k * psd, fm_V);
                    if constexpr (!COMPACT) ra[k] = (uint16_t)e;
                    v = s_off[e + 1] - s_off[e] + 1;
                }
                const uint32_t inc = wave_incl_scan(v);
                if (k < (uint32_t)V) rt[e] = k | ((carry + inc - v) << 16);
                carry += __shfl(inc, 63);
            }
        }
        wave_sync();
        if (total > 0) {
            resolve(C0, 0);
            if (n_leaves > 1) resolve(C1, 1);
        }

        PH(1)
        // ---- (C) candidate rounds: fill the rings, replay one 64-wide batch, repeat -----------
        int done = 0;
        while (!done) {
            PH(4)
            // C1: generation.  Keep >= 32 candidates pending per live leaf (>= min(64, rc - K) when
            // only one leaf is live) so that the replay batch below finds every lane a candidate.
            //
            // gen_rest: one source of leaf l from neighbour-row offset `base` on, `key` = the 64 row
            // entries at base (lane i = entry base + i); returns the number of candidates appended.
            auto gen_rest = [&](int l, uint32_t se, uint32_t sp, uint32_t len, uint32_t k, uint32_t sx, uint32_t tl,
                                uint32_t key, uint32_t base, uint32_t need, uint32_t emitted) -> uint32_t {
                const bool is_change = l ? chg1 : chg0;
                const RT rt = route_facts(l);
                uint32_t* rq = ring + (size_t)l * cv.rc * 2;
                const uint16_t* rowk = nb.keys + (size_t)sx * dim;
                const uint32_t mv0 = (se << 16) | sp;
                for (;;) {
                    const bool have = key != NBR_END;
                    const uint64_t havemask = __ballot(have);
                    if (havemask == 0) break;  // finite entries of the row exhausted
                    NearbyItem it{0u, 0u, 0u, 0u};
                    if (have) it = nearby_item_rt(is_change, node_slot.get(key & NBR_NODE_MASK), se, sp, len, k, s_off, rt);
                    // equal-distance groups are contiguous lane ranges; the index marks entries that
                    // continue the previous entry's distance
                    const uint64_t startmask = __ballot(have && (lane == 0 || !(key & NBR_SAME_FLAG)));
                    const bool more = havemask == ~0ULL && base + 64 < dim;
                    const uint32_t fo = 63u - (uint32_t)__clzll((unsigned long long)startmask);  // first lane of the last group
                    if (more && fo == 0) {
                        // one distance group wider than a chunk (degenerate ties): exact serial
                        // insertion top-k over the rest of the row.
                        TopK tk{~0ULL, 0u, ~0ULL};
                        for (uint32_t b2 = base; b2 < dim; b2 += 64) {
                            const uint32_t j2 = b2 + lane;
                            const uint32_t ky = j2 < dim ? (uint32_t)rowk[j2] : NBR_END;
                            NearbyItem i2{0u, 0u, 0u, 0u};
                            uint64_t hk = 0;
                            if (ky != NBR_END) {
                                const uint32_t y2 = ky & NBR_NODE_MASK;
                                i2 = nearby_item_rt(is_change, node_slot.get(y2), se, sp, len, k, s_off, rt);
                                hk = (uint64_t)m.mat[(size_t)ext_id(sx) * dim + ext_id(y2)] << RT::KEY_SHIFT;  // finite by construction of the index
                            }
                            if (!__ballot(ky != NBR_END)) break;
                            topk_offer(tk, need, i2.w >= 1 ? (hk | i2.ord) : ~0ULL, i2.pay0);
                            topk_offer(tk, need, i2.w == 2 ? (hk | (i2.ord + 1)) : ~0ULL, i2.pay1);
                        }
                        const uint32_t cnt = (uint32_t)__popcll(__ballot(lane < need && tk.key != ~0ULL));
                        if (lane < cnt) {
                            const uint32_t qi = (tl + emitted + lane) & RCM;
                            rq[qi * 2] = mv0;
                            rq[qi * 2 + 1] = tk.pay;
                        }
                        emitted += cnt;
                        break;
                    }
                    const uint32_t nclosed = more ? fo : (uint32_t)__popcll(havemask);  // lanes [0, nclosed): complete groups
                    const uint64_t closedmask = nclosed >= 64 ? ~0ULL : ((1ULL << nclosed) - 1ULL);
                    const bool closed = (closedmask & lanebit) != 0;
                    const uint32_t wc = closed ? it.w : 0u;
                    const uint64_t w1 = __ballot(wc == 1), w2 = __ballot(wc == 2);
                    const uint32_t Wc = (uint32_t)__popcll(w1) + 2u * (uint32_t)__popcll(w2);
                    GC(l ? 6 : 5)
                    if (Wc == 0) { GC(7) }
                    if (Wc > 0) {
                        // sorted position = weight before me, corrected by the inversions inside my group
                        int32_t pos = (int32_t)(mbcnt64(w1) + 2u * mbcnt64(w2));
                        const uint64_t nonstart = ~startmask & closedmask;  // lane continues the group of lane-1
                        if (nonstart) {
                            const uint32_t packed = (it.ord << 2) | wc;
                            uint64_t run = nonstart;  // bit t: lanes t-d .. t are one group
                            uint32_t p_dn = packed, p_up = packed;
                            for (uint32_t d = 1; run != 0; ++d) {
                                // lane i <- lane i +- d: one whole-wave DPP shift per iteration (no LDS crossbar)
                                p_dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p_dn, 0x130, 0xf, 0xf, false);  // wave_shl:1
                                p_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p_up, 0x138, 0xf, 0xf, false);  // wave_shr:1
                                const bool same_up = (run & lanebit) != 0;
                                const bool same_dn = ((run >> d) & lanebit) != 0;
                                if (same_dn && (p_dn >> 2) < it.ord) pos += (int32_t)(p_dn & 3u);
                                if (same_up && (p_up >> 2) > it.ord) pos -= (int32_t)(p_up & 3u);
                                run &= nonstart << d;
                            }
                        }
                        if (wc >= 1 && (uint32_t)pos < need) {
                            const uint32_t qi = (tl + emitted + (uint32_t)pos) & RCM;
                            rq[qi * 2] = mv0;
                            rq[qi * 2 + 1] = it.pay0;
                        }
                        if (wc == 2 && (uint32_t)pos + 1 < need) {
                            const uint32_t qi = (tl + emitted + (uint32_t)pos + 1) & RCM;
                            rq[qi * 2] = mv0;
                            rq[qi * 2 + 1] = it.pay1;
                        }
                        const uint32_t ne = Wc < need ? Wc : need;
                        emitted += ne;
                        need -= ne;
                    }
                    if (need == 0 || !more) break;
                    base += nclosed;
                    const uint32_t jj = base + lane;
                    key = jj < dim ? (uint32_t)rowk[jj] : NBR_END;
                }
                return emitted;
            };
            const uint32_t single0 = cv.rc - K0 < 64u ? cv.rc - K0 : 64u, single1 = cv.rc - K1 < 64u ? cv.rc - K1 : 64u;
            for (;;) {
                const bool both_live = !C0.ex && !C1.ex;
                const bool need0 = !C0.ex && C0.left > 0 && C0.tail - C0.head < (both_live ? 32u : single0);
                const bool need1 = n_leaves > 1 && !C1.ex && C1.left > 0 && C1.tail - C1.head < (both_live ? 32u : single1);
                if (!need0 && !need1) break;
                if (need0 && need1) {
                    ISA_MARK("pair_begin");
                    GC(0)
                    // ---- paired pass: lanes 0-31 = the first 32 row entries of leaf 0's source, lanes
                    // 32-63 = the first 32 of leaf 1's (its prefetch is stored rotated by 32 lanes) ----
                    const bool hi = lane >= 32;
                    const uint32_t seA = C0.se, spA = C0.sp, lenA = C0.len, kA = C0.k, sxA = C0.sx, tlA = C0.tail;
                    const uint32_t seB = C1.se, spB = C1.sp, lenB = C1.len, kB = C1.k, sxB = C1.sx, tlB = C1.tail;
                    const uint32_t key = hi ? C1.pk : C0.pk;
                    C0.left -= 1;
                    C0.o += 1;
                    if (C0.left > 0) resolve(C0, 0);
                    C1.left -= 1;
                    C1.o += 1;
                    if (C1.left > 0) resolve(C1, 1);
                    st_sources += 2;
                    const uint32_t se = hi ? seB : seA, sp = hi ? spB : spA, len = hi ? lenB : lenA, kk = hi ? kB : kA;
                    const uint32_t Kh = hi ? K1 : K0;
                    const bool have = key != NBR_END;
                    NearbyItem it{0u, 0u, 0u, 0u};
                    {
                        const uint32_t slot_any = node_slot.get(have ? (key & NBR_NODE_MASK) : 0u);  // unconditional read (node 0 for the lanes past the row)
                        const uint32_t slot = have ? slot_any : NODE_NONE;
                        RT rth;
                        if constexpr (ARANK)
                            rth = RouteArith{hi ? perm_st1 : perm_st0, hi ? perm_inv1 : perm_inv0, (uint32_t)V, v_recip32};
                        else
                            rth = RouteTab{rtab + (hi ? V : 0)};
                        if constexpr (FAST) {
                            // leaf 0 = nearby change (lanes 0-31), leaf 1 = nearby swap (lanes 32-63): the two items of nearby_item_rt written as ONE, the
                            // facts they share read once and the half a lane belongs to folded into the predicates (nearby_change.rs:133-195, nearby_swap.rs)
                            const bool some = slot != NODE_NONE;
                            const uint32_t r2 = some ? slot >> 16 : 0u, dp = slot & 0xFFFFu;
                            const uint32_t len2 = s_off[r2 + 1] - s_off[r2];
                            const uint32_t t = rth.word(r2);
                            const bool intra = r2 == se;
                            const uint32_t end_pay = (r2 << 16) | len2;
                            // change: the slot itself unless it is the source's own or the one behind it (dp - sp is 0 or 1, unsigned); swap: a later position of
                            // the source's list, or a list ranked after it
                            const bool v0 = hi ? (intra ? dp > sp : RT::rank(t) > kk) : (!intra || dp - sp >= 2u);
                            const bool v1 = !hi && dp + 1 == len2 && (!intra || len != sp + 1);  // change only: the end slot `len` probes element len - 1
                            it.w = some ? (uint32_t)v0 + (uint32_t)v1 : 0u;
                            it.ord = intra ? ((hi || v0) ? dp : len) : RT::inter_ord(t, dp);
                            it.pay0 = (hi || v0) ? slot : end_pay;
                            it.pay1 = end_pay;
                        } else {
                            const NearbyItem ic = nearby_item_rt(true, slot, se, sp, len, kk, s_off, rth);
                            const NearbyItem is = nearby_item_rt(false, slot, se, sp, len, kk, s_off, rth);
                            const bool lane_change = hi ? chg1 : chg0;
                            it.w = lane_change ? ic.w : is.w;
                            it.ord = lane_change ? ic.ord : is.ord;
                            it.pay0 = lane_change ? ic.pay0 : is.pay0;
                            it.pay1 = lane_change ? ic.pay1 : is.pay1;
                        }
                    }
                    const uint64_t havemask = __ballot(have);
                    const uint64_t startmask = __ballot(have && ((lane & 31u) == 0 || !(key & NBR_SAME_FLAG)));
                    const uint32_t hvA = (uint32_t)havemask, hvB = (uint32_t)(havemask >> 32);
                    const uint32_t stA = (uint32_t)startmask, stB = (uint32_t)(startmask >> 32);
                    const bool moreA = hvA == 0xFFFFFFFFu && 32u < dim, moreB = hvB == 0xFFFFFFFFu && 32u < dim;
                    // complete groups of each half (a half whose only group is still open contributes nothing)
                    const uint32_t ncA = moreA ? 31u - (uint32_t)__clz(stA) : (uint32_t)__popc(hvA);
                    const uint32_t ncB = moreB ? 31u - (uint32_t)__clz(stB) : (uint32_t)__popc(hvB);
                    const uint64_t closedmask = (uint64_t)(ncA >= 32 ? 0xFFFFFFFFu : ((1u << ncA) - 1u)) |
                                                ((uint64_t)(ncB >= 32 ? 0xFFFFFFFFu : ((1u << ncB) - 1u)) << 32);
                    const bool closed = (closedmask & lanebit) != 0;
                    const uint32_t wc = closed ? it.w : 0u;
                    const uint64_t w1 = __ballot(wc == 1), w2 = __ballot(wc == 2);
                    const uint32_t WcA = (uint32_t)__popc((uint32_t)w1) + 2u * (uint32_t)__popc((uint32_t)w2);
                    const uint32_t WcB = (uint32_t)__popc((uint32_t)(w1 >> 32)) + 2u * (uint32_t)__popc((uint32_t)(w2 >> 32));
                    if (WcA + WcB > 0) {
                        const uint32_t pa = __builtin_amdgcn_mbcnt_lo((uint32_t)w1, 0u) + 2u * __builtin_amdgcn_mbcnt_lo((uint32_t)w2, 0u);
                        const uint32_t pb = __builtin_amdgcn_mbcnt_hi((uint32_t)(w1 >> 32), 0u) + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(w2 >> 32), 0u);
                        int32_t pos = (int32_t)(hi ? pb : pa);
                        const uint64_t nonstart = ~startmask & closedmask;  // never crosses lane 32 (forced start)
                        if (nonstart) {
                            const uint32_t packed = (it.ord << 2) | wc;
                            uint64_t run = nonstart;
                            uint32_t p_dn = packed, p_up = packed;
                            for (uint32_t d = 1; run != 0; ++d) {
                                p_dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p_dn, 0x130, 0xf, 0xf, false);  // wave_shl:1
                                p_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p_up, 0x138, 0xf, 0xf, false);  // wave_shr:1
                                const bool same_up = (run & lanebit) != 0;
                                const bool same_dn = ((run >> d) & lanebit) != 0;
                                if (same_dn && (p_dn >> 2) < it.ord) pos += (int32_t)(p_dn & 3u);
                                if (same_up && (p_up >> 2) > it.ord) pos -= (int32_t)(p_up & 3u);
                                run &= nonstart << d;
                            }
                        }
                        uint32_t* rq = ring + (hi ? (size_t)cv.rc * 2 : 0);
                        const uint32_t tl = hi ? tlB : tlA;
                        const uint32_t mv0 = (se << 16) | sp;
                        if (wc >= 1 && (uint32_t)pos < Kh) {
                            const uint32_t qi = (tl + (uint32_t)pos) & RCM;
                            rq[qi * 2] = mv0;
                            rq[qi * 2 + 1] = it.pay0;
                        }
                        if (wc == 2 && (uint32_t)pos + 1 < Kh) {
                            const uint32_t qi = (tl + (uint32_t)pos + 1) & RCM;
                            rq[qi * 2] = mv0;
                            rq[qi * 2 + 1] = it.pay1;
                        }
                    }
                    uint32_t emA = WcA < K0 ? WcA : K0, emB = WcB < K1 ? WcB : K1;
                    if (emA < K0 && moreA) {  // rare: leaf 0's source needs entries beyond its half
                        GC(3)
                        const uint32_t jj = ncA + lane;
                        emA = gen_rest(0, seA, spA, lenA, kA, sxA, tlA, jj < dim ? (uint32_t)nb.keys[(size_t)sxA * dim + jj] : NBR_END, ncA, K0 - emA, emA);
                    }
                    if (emB < K1 && moreB) {
                        GC(4)
                        const uint32_t jj = ncB + lane;
                        emB = gen_rest(1, seB, spB, lenB, kB, sxB, tlB, jj < dim ? (uint32_t)nb.keys[(size_t)sxB * dim + jj] : NBR_END, ncB, K1 - emB, emB);
                    }
                    C0.tail = tlA + emA;
                    C1.tail = tlB + emB;
                    ISA_MARK("pair_end");
                } else {
                    // ---- single source of the one leaf that needs candidates ----
                    const int l = need0 ? 0 : 1;
                    GC(l ? 2 : 1)
                    LeafCursor c = l ? C1 : C0;
                    const uint32_t se = c.se, sp = c.sp, len = c.len, k = c.k, sx = c.sx;
                    // leaf 1 keeps its prefetched entries rotated by 32 lanes for the paired pass
                    const uint32_t key = l ? __shfl(c.pk, (int)((lane + 32u) & 63u)) : c.pk;
                    c.left -= 1;
                    c.o += 1;
                    if (c.left > 0) resolve(c, l);
                    st_sources += 1;
                    c.tail += gen_rest(l, se, sp, len, k, sx, c.tail, key, 0u, l ? K1 : K0, 0u);
                    if (l)
                        C1 = c;
                    else
                        C0 = c;
                }
            }
            wave_sync();
            PH(2)

            // C2: replay one batch in union cursor order: trial score, acceptor, forager
            ISA_MARK("replay_begin");
            {
                const bool live0 = !C0.ex, live1 = !C1.ex;
                if (!live0 && !live1) {
                    done = 1;
                    break;
                }
                uint32_t lf, idx;
                if (live0 && live1) {
                    const uint32_t l0 = (first_leaf + pulls) & 1u;
                    lf = (l0 + lane) & 1u;
                    idx = (lf ? C1.head : C0.head) + (lane >> 1);
                } else {
                    lf = live0 ? 0u : 1u;
                    idx = (lf ? C1.head : C0.head) + lane;
                }
                const bool avail = (int32_t)((lf ? C1.tail : C0.tail) - idx) > 0;
                const uint64_t availmask = __ballot(avail);
                const uint32_t nvalid = availmask == ~0ULL ? 64u : (uint32_t)(__ffsll((unsigned long long)~availmask) - 1);
                if (nvalid == 0) {
                    // the scheduler discovers an exhausted child at this pull (vec_union.rs:334-362);
                    // a leaf with sources left was refilled above, so an empty ring means exhausted.
                    const uint32_t lf0 = uni(__shfl(lf, 0));
                    if (lf0)
                        C1.ex = 1;
                    else
                        C0.ex = 1;
                    continue;
                }
                const bool valid = lane < nvalid;
                uint32_t m0 = 0, m1 = 0;
                ScoreV<L> sc;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) sc.v[kk] = 0;
                bool doable = false, acc = false, consumed = false, improving_pick = false;
                uint64_t accmask = 0;
                uint32_t nconsumed = nvalid;
                if constexpr (SMALL) {
                    // ---- delta-space replay (MODE 2): int32 level deltas against the step's current score ----
                    // Every ring entry of a FAST kernel is a nearby candidate generated from the committed lists of THIS step: structurally doable
                    // (eval_generated_small), so there is no doability test, and no exec-mask region either -- a lane past the valid ones re-prices
                    // the batch's first candidate and is masked out of the decisions.
                    int32_t dv[L];
                    {
                        const uint32_t lf0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lf), idx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
                        const uint32_t lfq = valid ? lf : lf0;
                        const uint32_t qi = (valid ? idx : idx0) & RCM;
                        const uint32_t* rq = ring + ((size_t)lfq * cv.rc + qi) * 2;
                        m0 = rq[0];
                        m1 = rq[1];
                        const uint32_t a = m0 >> 16, i = m0 & 0xFFFFu, b = m1 >> 16, j = m1 & 0xFFFFu;
                        eval_generated_small<L, LT, COMPACT, OT>(m, s_visits, s_off, s_load, 1u - lfq, a, i, b, j, dv);  // FAST: leaf 0 = change, leaf 1 = swap; COMPACT reads the u16 matrix
                    }
                    doable = valid;
                    // LateAcceptance: score >= last step score || score >= late score (late_acceptance.rs:89-125).  Two levels: the pair of int32 deltas
                    // is ONE signed 64-bit key (hard x 2^32 + soft, |soft| < 2^31 keeps the order lexicographic), and the two tests are one compare
                    // against the wave-uniform min(0, late - current)
                    int64_t key = 0;
                    if constexpr (L == 2) {
                        key = (int64_t)(((uint64_t)(uint32_t)dv[0] << 32) + (uint64_t)(int64_t)dv[1]);
                        acc = valid && key >= acc_thr;
                    } else {
                        acc = valid && (small_ge0<L>(dv) || small_ge<L>(dv, late_d));
                    }
                    accmask = __ballot(acc);
                    {  // AcceptedCount quota (forager.rs:232-239)
                        const uint32_t remaining = (uint32_t)p.limit - accepted;
                        const uint32_t pre = mbcnt64(accmask) + (acc ? 1u : 0u);
                        const uint64_t cutmask = __ballot(acc && pre == remaining);
                        nconsumed = cutmask ? (uint32_t)__ffsll((unsigned long long)cutmask) : nvalid;
                    }
                    consumed = lane < nconsumed;
                    acc = acc && consumed;
                    accmask = __ballot(acc);
                    bool challenger;
                    if constexpr (L == 2)
                        challenger = acc && key >= best_key;
                    else
                        challenger = acc && small_ge<L>(dv, best_d);
                    if (accmask && (!has_best || __ballot(challenger))) {
                        // lexicographic maximum of the accepted lanes, level by level
                        int32_t M[L];
                        bool in_max = acc;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) {
                            M[kk] = wave_max_i32(in_max ? dv[kk] : (int32_t)0x80000000);
                            in_max = in_max && dv[kk] == M[kk];
                        }
                        const bool ge = !has_best || small_ge<L>(M, best_d);
                        if (ge) {
                            const bool newmax = !has_best || !small_ge<L>(best_d, M);
                            const uint64_t eq_base = newmax ? 0 : equal_count;
                            const uint64_t eq = __ballot(in_max);
                            const uint32_t rank = mbcnt64(eq) + 1u;
                            const uint64_t cntq = eq_base + rank;
                            const bool pick = in_max && ((newmax && rank == 1) ||
                                                         (p.random_ties && cntq > 1 && reservoir_pick(sseed, cntq)));
                            const uint64_t pm = __ballot(pick);
                            if (pm) {
                                const int sel = 63 - __clzll((unsigned long long)pm);
                                best_m0 = __shfl(m0, sel);
                                best_m1 = __shfl(m1, sel);
                                best_leaf = (int)__shfl(lf, sel);
                            }
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) best_d[kk] = M[kk];
                            if constexpr (L == 2) best_key = (int64_t)(((uint64_t)(uint32_t)M[0] << 32) + (uint64_t)(int64_t)M[1]);
                            equal_count = eq_base + (uint64_t)__popcll(eq);
                            has_best = 1;
                        }
                    }
                } else {
                    ListDelta dl{0, 0, false};
                    if (valid) {
                        const uint32_t qi = idx & RCM;
                        const uint32_t* rq = ring + ((size_t)lf * cv.rc + qi) * 2;
                        m0 = rq[0];
                        m1 = rq[1];
                        const uint32_t a = m0 >> 16, i = m0 & 0xFFFFu, b = m1 >> 16, j = m1 & 0xFFFFu;
                        dl = eval_list_move_legs<uint16_t, FAST>(m, s_visits, s_off, s_load, lf ? chg1 : chg0, a, i, b, j);
                    }
                    sc = apply_delta<L>(m, cur, dl);
                    ScoreV<L> curv;
    #pragma unroll
                    for (int kk = 0; kk < L; ++kk) curv.v[kk] = cur[kk];
                    doable = valid && dl.doable;
                    if (doable) {
                        if (acceptor == 0)
                            acc = score_cmp<L>(sc, curv) > 0;
                        else if (acceptor == 1)
                            acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0;
                        else if (acceptor == 4)
                            acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0 || score_cmp<L>(sc, dla_thr) >= 0;
                    }
                    SaChunk sach;
                    if constexpr (!FAST)
                        if (annealing) acc = sa_decide<L>(saw, p.sa, doable, sc, curv, lane, sach);
                    accmask = __ballot(acc);
                    ScoreV<L> forager_thr = curv;  // FirstLastStepScoreImproving: the last step score
                    if (forager == FORAGER_FIRST_BEST_IMPROVING) {  // the best score ever seen (step.rs:53-58)
    #pragma unroll
                        for (int kk = 0; kk < L; ++kk) forager_thr.v[kk] = best_sol[kk];
                    }
                    nconsumed = forager_chunk_cut<L>(forager, (uint32_t)p.limit, accepted, acc, sc, forager_thr, nvalid, improving_pick);
                    consumed = lane < nconsumed;
                    if constexpr (!FAST)
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
                        } else if (forager == 1) {
                            if (!has_best) {
                                const int sel = __ffsll((unsigned long long)accmask) - 1;
    #pragma unroll
                                for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)uni64(shfl_u64((uint64_t)sc.v[kk], sel));
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
    #pragma unroll
                                for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)uni64((uint64_t)M.v[kk]);
                                equal_count = eq_base + (uint64_t)__popcll(eq);
                                has_best = 1;
                            }
                        }
                    }
                }
                const uint32_t nacc = (uint32_t)__popcll(accmask);
                accepted += nacc;
                st_scored += nvalid;
                if constexpr (!SMALL) {  // (SMALL: moves_generated / accepted are added once per step from `pulls` / `accepted`; every candidate is doable)
                    st_gen += nconsumed;
                    st_acc += nacc;
                    const uint32_t ndo = (uint32_t)__popcll(__ballot(consumed && doable));
                    st_calc += ndo;
                }
                if (tracing && consumed) {
                    const uint64_t ti = trace_n + lane;
                    if ((int64_t)ti < p.trace_cap) {
                        int32_t* tm = p.trace_moves + ti * 6;
                        tm[0] = (lf ? chg1 : chg0) ? 2 : 3;
                        tm[1] = (int32_t)(m0 >> 16);
                        tm[2] = (int32_t)(m0 & 0xFFFFu);
                        tm[3] = (int32_t)(m1 >> 16);
                        tm[4] = (int32_t)(m1 & 0xFFFFu);
                        tm[5] = -1;
                        for (int kk = 0; kk < L && kk < m.levels; ++kk) p.trace_scores[ti * m.levels + kk] = doable ? sc.v[kk] : 0;
                        p.trace_flags[ti] = (doable ? 1 : 0) | (acc ? 2 : 0) | ((int32_t)lf << 8);
                    }
                }
                if (tracing) trace_n += nconsumed;
                uint32_t c1;
                if constexpr (SMALL) {  // the lanes alternate the two leaves from the batch's first pull on: no ballot needed
                    const uint32_t lfirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lf);
                    c1 = (live0 && live1) ? ((nconsumed + lfirst) >> 1) : (lfirst ? nconsumed : 0u);
                } else {
                    c1 = (uint32_t)__popcll(__ballot(consumed && lf == 1u));
                }
                C1.head += c1;
                C0.head += nconsumed - c1;
                pulls += nconsumed;
                if (forager_quits(forager, (uint32_t)p.limit, accepted, has_best, improving_pick)) done = 1;
            }
            ISA_MARK("replay_end");
            PH(3)
        }
        PH(4)

        // ---- (D) commit the forager's pick (step.rs:122-221) ----------------------------------
        if constexpr (SMALL) {
#pragma unroll
            for (int kk = 0; kk < L; ++kk) best.v[kk] = wadd(cur[kk], (int64_t)best_d[kk]);
        }
        const bool applied = has_best && !dry_run;
        if (applied) {
            if (best_pending) {
                ScoreV<L> bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) bs.v[kk] = best_sol[kk];
                if (!(score_cmp<L>(best, bs) > 0)) {  // leaving the best state: write its snapshot first
                    const uint32_t tot = uni(s_off[V]);
                    for (uint32_t t = lane; t < tot; t += 64) m.best_visits[(size_t)r * m.n_cap + t] = ext_id(s_visits[t]);
                    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) m.best_off[(size_t)r * (V + 1) + t] = s_off[t];
                    best_pending = false;
                }
            }
            const int kind = (best_leaf ? chg1 : chg0) ? 2 : 3;
            const uint32_t a = uni(best_m0 >> 16), i = uni(best_m0 & 0xFFFFu);
            const uint32_t b = uni(best_m1 >> 16), j = uni(best_m1 & 0xFFFFu);
            if (tracing && lane == 0) {
                p.trace_applied[0] = 1;
                if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                p.trace_applied[1] = kind;
                p.trace_applied[2] = (int32_t)a;
                p.trace_applied[3] = (int32_t)i;
                p.trace_applied[4] = (int32_t)b;
                p.trace_applied[5] = (int32_t)j;
                p.trace_applied[6] = -1;
            }
            apply_list_move_wave(m, s_visits, s_off, s_load, kind, a, i, b, j);
            {  // refresh node -> (route, position) for the two touched routes
                const uint32_t oa = s_off[a], la = s_off[a + 1] - oa;
                const uint32_t ob = s_off[b], lb = s_off[b + 1] - ob;
                for (uint32_t t = lane; t < la + (a != b ? lb : 0u); t += 64) {
                    if (t < la)
                        node_slot.set(s_visits[oa + t], a, t);
                    else
                        node_slot.set(s_visits[ob + (t - la)], b, t - la);
                }
            }
            node_sync();
#pragma unroll
            for (int kk = 0; kk < L; ++kk) cur[kk] = best.v[kk];
            st_applied += 1;
        } else if (tracing && lane == 0) {
            p.trace_applied[0] = 0;
        }
        if (!dry_run) {
            // update_best_solution (scope_progress.rs:89-107): clone on strict improvement
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
            if (improved) {  // the clone is deferred until the search leaves this state (best_pending)
                best_pending = true;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) best_sol[kk] = cur[kk];
            }
            // acceptor.step_ended(last_step_score) always (step.rs:216-221)
            if ((acceptor == 1 || acceptor == 4) && lane == 0) {
#pragma unroll
                for (int kk = 0; kk < L; ++kk) p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + kk] = cur[kk];
            }
            if (acceptor == 4 && lane == 0) {  // step_ended: the phase's best step score (diversified_late_acceptance.rs:161-170)
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
            if constexpr (!FAST)
                if (annealing) sa_step_ended(saw, p.sa, lane);
            wave_sync();
            st_steps += 1;
            if constexpr (SMALL) {
                st_gen += pulls;
                st_calc += pulls;
                st_acc += accepted;
            }
            la_cursor = la_cursor + 1 >= p.la_size ? 0 : la_cursor + 1;
            if (p.move_budget > 0 && (int64_t)st_gen >= p.move_budget) break;  // work-balanced launch: see sf_solve_moves (budget < 2^31)
            if (p.move_budget == 0 && st_scored >= 0x70000000u) flush_stats();
        }
        PH(5)
    }
    PH_DUMP

    // ---- write back ----------------------------------------------------------------------
    if (!dry_run) {
        if constexpr (!FAST)
            if (annealing) sa_store(saw, p.sa, r, lane);
        const uint32_t tot = uni(s_off[V]);
        if (best_pending) {  // the launch ends in a best state: its deferred snapshot
            for (uint32_t t = lane; t < tot; t += 64) m.best_visits[(size_t)r * m.n_cap + t] = ext_id(s_visits[t]);
            for (uint32_t t = lane; t <= (uint32_t)V; t += 64) m.best_off[(size_t)r * (V + 1) + t] = s_off[t];
        }
        for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = ext_id(s_visits[t]);
        for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = s_off[t];
        for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = (int64_t)s_load[t];
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
