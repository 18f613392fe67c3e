// 3-opt candidate streams of the generic N-leaf engine (one wavefront = one replica).
//
// Reference semantics restated (paths under crates/solverforge-solver/src/heuristic/selector/):
//   list_kernel/k_opt/full.rs:12-101      KOptCursor: entity order without replacement, move offset ->
//                                         selection_index over (cut combinations x patterns)
//   k_opt/iterators.rs:98-178             count_cut_combinations, cut_combination_at (lexicographic unranking)
//   list_kernel/k_opt/nearby.rs:16-148    NearbyKOptCursor: entities, lazy cut sets, 7 patterns per cut set
//   list_kernel/k_opt/nearby_state.rs:22-241  NearbyCutState: distance-pruned depth-first cut generation
//
// The full-enumeration stream is lane-parallel (lane = one move offset, unranked independently).  The
// distance-pruned stream is a depth-first state machine with data-dependent backtracking, so its CONTROL
// runs wave-uniform (every lane follows the same path; state in registers during a call, in LDS between
// calls) while each level build -- distances from the origin, stable rank by (distance, position), range
// filter, selection order -- is done across the 64 lanes.
#pragma once
#include <stdint.h>

#include "sf_list_model.h"

namespace sf {

constexpr uint64_t SALT_KF_ENTITY = 0x4B0F7E1171000001ULL, SALT_KF_MOVE = 0x4B0F7E1171000002ULL;  // full.rs:47,73
constexpr uint64_t SALT_KN_FIRST = 0x4B0F7E1172EA0001ULL, SALT_KN_LEVEL = 0x4B0F7E1172EA0002ULL;  // nearby_state.rs:96,137
constexpr uint64_t SALT_KN_ENTITY = 0x4B0F7E1172EA0003ULL, SALT_KN_STATE = 0x4B0F7E1172EA0004ULL;  // nearby.rs:62,96
constexpr uint64_t SALT_KN_PATTERN = 0x4B0F7E1172EA0005ULL;                                         // nearby.rs:117
constexpr uint64_t KOPT_POS_MIX = 0xBF58476D1CE4E5B9ULL;

constexpr uint32_t KOPT_LDS_KEYS = 128;   // routes up to this length rank their distances in LDS, longer ones in HBM scratch
constexpr uint32_t KOPT_MAX_NEARBY = 64;  // one wave pass per level
constexpr uint32_t KOPT_TRIPLES = 9;      // cut sets per fill call: 9 x 7 patterns = 63 lanes
constexpr uint32_t KOPT_FAST_LEN = 16;    // routes up to this length rank ALL origins once when the entity is opened

// LDS working set of the distance-pruned stream (one per resident replica)
struct KoptLds {
    static constexpr size_t bytes = KOPT_LDS_KEYS * 8 + 64 * 2 + 64 * 2 + 3 * 64 * 2 + KOPT_TRIPLES * 4 * 2 + 8 + 24 * 4 + KOPT_FAST_LEN * KOPT_FAST_LEN;
    uint64_t* keys;    // [KOPT_LDS_KEYS] f64 bit patterns of the distances origin -> position
    uint16_t* sorted;  // [64] positions by rank
    uint16_t* vl;      // [64] range-filtered positions, rank order
    uint16_t* cache;   // [3][64] nearby_cache of the stack levels (level 0 is always empty)
    uint16_t* trip;    // [KOPT_TRIPLES][4] (entity, cut 1, cut 2, cut 3) of this fill call
    uint32_t* st;      // [24] machine state between calls
    uint32_t* dist;    // [KOPT_FAST_LEN][KOPT_FAST_LEN] short routes: compact-matrix legs between the route's positions (aliases `keys`)
    uint8_t* near;     // [KOPT_FAST_LEN][KOPT_FAST_LEN] short routes: positions by (distance, position) rank, per origin
    __device__ explicit KoptLds(unsigned char* base) {
        keys = (uint64_t*)base;
        sorted = (uint16_t*)(base + KOPT_LDS_KEYS * 8);
        vl = sorted + 64;
        cache = vl + 64;
        trip = cache + 3 * 64;
        st = (uint32_t*)(base + KOPT_LDS_KEYS * 8 + 64 * 2 + 64 * 2 + 3 * 64 * 2 + KOPT_TRIPLES * 4 * 2 + 8);
        dist = (uint32_t*)base;
        near = (uint8_t*)(st + 24);
    }
};

#if defined(__HIPCC__)

// ---- 64-bit stream helpers (the move count of one long list exceeds 32 bits: C(len - 1, 3) * 7) ------------------
__device__ __forceinline__ uint64_t kopt_gcd64(uint64_t a, uint64_t b) {
    while (b != 0) {
        const uint64_t r = a % b;
        a = b;
        b = r;
    }
    return a;
}
// MoveStreamContext::selection_index over a 64-bit length (iter.rs:112-128)
__device__ __forceinline__ uint64_t kopt_selection_index64(const StreamCtx& ctx, uint64_t offset, uint64_t len, uint64_t salt) {
    if (ctx.order <= 2) return offset;
    if (len <= 1) return 0;
    if (ctx.order == 3) return ctx.mixed_seed(salt ^ (offset * OFFSET_MIX)) % len;
    const uint64_t start = ctx.mixed_seed(salt) % len;
    uint64_t s = ctx.mixed_seed(salt ^ STRIDE_SALT_MIX) % (len - 1) + 1;
    while (kopt_gcd64(s, len) != 1) s = (s == len - 1) ? 1 : s + 1;
    return (start + offset * s) % len;
}
// count_cut_combinations(3, len, min_seg) (iterators.rs:98-106): C(len - 4 * min_seg + 3, 3)
__device__ __forceinline__ uint64_t kopt_cut_count(uint32_t len, uint32_t mseg) {
    if (len < 4u * mseg) return 0;
    const uint64_t n = (uint64_t)len - 4ull * mseg + 3ull;
    return n < 3 ? 0 : n * (n - 1) / 2 * (n - 2) / 3;  // exact: n(n-1)/2 is an integer, one of three consecutive factors has a 3
}
// cut_combination_at(3, len, min_seg, rank) (iterators.rs:108-161)
__device__ __forceinline__ void kopt_unrank(uint32_t len, uint32_t mseg, uint64_t rank, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
    const uint64_t n = (uint64_t)len - 4ull * mseg + 3ull;  // choice_count
    uint64_t sel0 = 0, sel1 = 0, sel2 = 0;
    for (uint64_t cand = 0; cand + 3 <= n; ++cand) {  // position 0: suffix = C(n - cand - 1, 2)
        const uint64_t m = n - cand - 1;
        const uint64_t suffix = m * (m - 1) / 2;
        if (rank < suffix) {
            sel0 = cand;
            break;
        }
        rank -= suffix;
    }
    for (uint64_t cand = sel0 + 1; cand + 2 <= n; ++cand) {  // position 1: suffix = n - cand - 1
        const uint64_t suffix = n - cand - 1;
        if (rank < suffix) {
            sel1 = cand;
            break;
        }
        rank -= suffix;
    }
    sel2 = sel1 + 1 + rank;  // position 2: suffix = 1 per candidate
    c1 = (uint32_t)(sel0 + mseg);
    c2 = (uint32_t)(sel1 + mseg + (mseg - 1));
    c3 = (uint32_t)(sel2 + mseg + 2ull * (mseg - 1));
}
// the 7-pattern order of one cut set (nearby.rs:115-128)
__device__ __forceinline__ uint64_t kopt_pattern_salt(uint64_t desc, uint32_t entity, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint64_t salt = SALT_KN_PATTERN ^ desc;
    salt ^= ((uint64_t)entity * GOLDEN) ^ ((uint64_t)c1 * KOPT_POS_MIX);
    salt ^= ((uint64_t)entity * GOLDEN) ^ ((uint64_t)c2 * KOPT_POS_MIX);
    salt ^= ((uint64_t)entity * GOLDEN) ^ ((uint64_t)c3 * KOPT_POS_MIX);
    return salt;
}

// ---- distance-pruned cut state machine ---------------------------------------------------------------------------
struct KoptS {
    uint32_t depth, p0, p1, p2, i0, i1, i2, n0, n1, n2;
    uint32_t first_offset, fst, fsd, fcount, done, active, entity, len, fast;
    __device__ __forceinline__ uint32_t pos(uint32_t t) const { return t == 0 ? p0 : (t == 1 ? p1 : p2); }
    __device__ __forceinline__ uint32_t idx(uint32_t t) const { return t == 0 ? i0 : (t == 1 ? i1 : i2); }
    __device__ __forceinline__ uint32_t cnt(uint32_t t) const { return t == 0 ? n0 : (t == 1 ? n1 : n2); }
    __device__ __forceinline__ void set(uint32_t t, uint32_t p, uint32_t i) {
        if (t == 0) p0 = p, i0 = i;
        if (t == 1) p1 = p, i1 = i;
        if (t == 2) p2 = p, i2 = i;
    }
    __device__ __forceinline__ void set_cnt(uint32_t t, uint32_t n) {
        if (t == 0) n0 = n;
        if (t == 1) n1 = n;
        if (t == 2) n2 = n;
    }
};
__device__ __forceinline__ uint32_t kopt_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void kopt_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void kopt_load_state(const uint32_t* st, KoptS& s) {
    s.depth = kopt_uni(st[0]), s.p0 = kopt_uni(st[1]), s.p1 = kopt_uni(st[2]), s.p2 = kopt_uni(st[3]);
    s.i0 = kopt_uni(st[4]), s.i1 = kopt_uni(st[5]), s.i2 = kopt_uni(st[6]);
    s.n0 = kopt_uni(st[7]), s.n1 = kopt_uni(st[8]), s.n2 = kopt_uni(st[9]);
    s.first_offset = kopt_uni(st[10]), s.fst = kopt_uni(st[11]), s.fsd = kopt_uni(st[12]), s.fcount = kopt_uni(st[13]);
    s.done = kopt_uni(st[14]), s.active = kopt_uni(st[15]), s.entity = kopt_uni(st[16]), s.len = kopt_uni(st[17]);
    s.fast = kopt_uni(st[18]);
}
__device__ __forceinline__ void kopt_store_state(uint32_t* st, const KoptS& s, uint32_t lane) {
    if (lane == 0) {
        st[0] = s.depth, st[1] = s.p0, st[2] = s.p1, st[3] = s.p2, st[4] = s.i0, st[5] = s.i1, st[6] = s.i2;
        st[7] = s.n0, st[8] = s.n1, st[9] = s.n2, st[10] = s.first_offset, st[11] = s.fst, st[12] = s.fsd, st[13] = s.fcount;
        st[14] = s.done, st[15] = s.active, st[16] = s.entity, st[17] = s.len, st[18] = s.fast;
    }
}

struct KoptEnv {
    const ListModel* lm;
    const uint16_t* visits;  // the replica's flat lists (LDS)
    const uint32_t* off;
    KoptLds mem;
    uint64_t* gkeys;  // HBM scratch [n_cap] of this replica (routes longer than KOPT_LDS_KEYS)
    StreamCtx ctx;
    uint64_t desc;
    uint32_t mseg, max_nearby, lane;
};

// first_positions[i] (nearby_state.rs:90-98): min_seg + the i-th element of the permutation without replacement
__device__ __forceinline__ uint32_t kopt_first_position(const KoptEnv& e, const KoptS& s, uint32_t i) {
    return e.mseg + (uint32_t)(((uint64_t)s.fst + (uint64_t)i * s.fsd) % s.fcount);
}

// nearby_positions(origin) -> range filter -> apply_selection_order (nearby_state.rs:22-52,121-140) into cache[level];
// returns the number of valid positions.  All 64 lanes call.
__device__ __forceinline__ uint32_t kopt_build_level(const KoptEnv& e, const KoptS& s, uint32_t level, uint32_t origin,
                                                     uint32_t minp, uint32_t maxp) {
    const uint32_t base = e.off[s.entity], len = s.len, lane = e.lane;
    if (s.fast) {  // the entity's rank table is already in LDS (kopt_rank_all_origins): no memory round trip per level
        const uint32_t cnt = len - 1 < e.max_nearby ? len - 1 : e.max_nearby;
        uint32_t mine = 0;
        bool ok = false;
        if (lane < cnt) {
            mine = e.mem.near[origin * KOPT_FAST_LEN + lane];
            ok = mine >= minp && mine <= maxp;
        }
        const uint64_t okm = __ballot(ok);
        const uint32_t nv = (uint32_t)__popcll(okm);
        if (ok) e.mem.vl[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = (uint16_t)mine;
        kopt_sync();
        if (lane < nv) {
            const uint64_t salt2 = (SALT_KN_STATE ^ e.desc ^ (uint64_t)s.entity) ^ SALT_KN_LEVEL ^ ((uint64_t)origin * GOLDEN) ^ (uint64_t)level;
            e.mem.cache[level * 64 + lane] = e.mem.vl[e.ctx.selection_index(lane, nv, salt2)];
        }
        kopt_sync();
        return nv;
    }
    // routes that do not fit the LDS key buffer rank through the replica's HBM scratch: device-scope
    // accesses, so the lanes of this wave see each other's keys through L2
    const bool glob = len > KOPT_LDS_KEYS;
    auto put = [&](uint32_t p, uint64_t k) {
        if (glob)
            __hip_atomic_store(&e.gkeys[p], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            e.mem.keys[p] = k;
    };
    auto get = [&](uint32_t p) -> uint64_t {
        return glob ? __hip_atomic_load(&e.gkeys[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e.mem.keys[p];
    };
    const uint32_t vo = e.visits[base + origin];
    const int64_t* row = e.lm->mat + (size_t)vo * (uint32_t)e.lm->dim;
    for (uint32_t p = lane; p < len; p += 64) {
        const int64_t d = row[e.visits[base + p]];
        // finite_distance (problem_data.rs:44-47) as f64, else INFINITY (meters.rs:45-52); the bit pattern of a
        // non-negative f64 orders like the number
        const double x = (d >= 0 && d != UNREACHABLE) ? (double)d : __longlong_as_double(0x7FF0000000000000LL);
        put(p, (uint64_t)__double_as_longlong(x));
    }
    if (glob) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    kopt_sync();
    // stable sort by distance + truncate(max_nearby) = rank by (distance, position) and keep rank < max_nearby
    for (uint32_t p = lane; p < len; p += 64) {
        if (p == origin) continue;
        const uint64_t kp = get(p);
        uint32_t rank = 0;
        for (uint32_t q = 0; q < len; ++q) {
            const uint64_t kq = get(q);
            rank += (q != origin && (kq < kp || (kq == kp && q < p))) ? 1u : 0u;
        }
        if (rank < e.max_nearby) e.mem.sorted[rank] = (uint16_t)p;
    }
    kopt_sync();
    const uint32_t cnt = len - 1 < e.max_nearby ? len - 1 : e.max_nearby;
    uint32_t mine = 0;
    bool ok = false;
    if (lane < cnt) {
        mine = e.mem.sorted[lane];
        ok = mine >= minp && mine <= maxp;
    }
    const uint64_t okm = __ballot(ok);
    const uint32_t nv = (uint32_t)__popcll(okm);
    if (ok) e.mem.vl[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = (uint16_t)mine;
    kopt_sync();
    if (lane < nv) {
        const uint64_t salt2 = (SALT_KN_STATE ^ e.desc ^ (uint64_t)s.entity) ^ SALT_KN_LEVEL ^ ((uint64_t)origin * GOLDEN) ^ (uint64_t)level;
        e.mem.cache[level * 64 + lane] = e.mem.vl[e.ctx.selection_index(lane, nv, salt2)];
    }
    kopt_sync();
    return nv;
}

// NearbyCutState::backtrack (nearby_state.rs:159-193)
__device__ __forceinline__ bool kopt_backtrack(const KoptEnv& e, KoptS& s) {
    while (s.depth > 0) {
        s.depth -= 1;  // stack.pop(), nearby_cache.pop()
        if (s.depth > 0) {
            const uint32_t t = s.depth - 1;  // the new top and its cache
            const uint32_t next_index = s.idx(t) + 1;
            if (next_index < s.cnt(t)) {
                const uint32_t next_position = kopt_uni((uint32_t)e.mem.cache[t * 64 + next_index]);
                const uint32_t position = s.pos(t);
                s.set(t, position, next_index);
                if (next_position > position) {
                    s.set(t, next_position, next_index);
                    return true;
                }
            }
        } else {
            s.first_offset += 1;
            if (s.first_offset < s.fcount) {
                s.set(0, kopt_first_position(e, s, s.first_offset), 0);
                s.n0 = 0;
                s.depth = 1;
                return true;
            }
        }
    }
    return false;
}
// extend_stack (nearby_state.rs:113-157)
__device__ __forceinline__ void kopt_extend(const KoptEnv& e, KoptS& s) {
    while (s.depth < 3 && !s.done) {
        const uint32_t last = s.pos(s.depth - 1), level = s.depth;
        const uint32_t remaining = 3 - s.depth;
        const uint32_t minp = last + e.mseg, maxp = s.len - e.mseg * remaining;
        const uint32_t nv = kopt_build_level(e, s, level, last, minp, maxp);
        if (nv == 0) {
            if (!kopt_backtrack(e, s)) {
                s.done = 1;
                return;
            }
        } else {
            s.set_cnt(level, nv);
            s.set(level, kopt_uni((uint32_t)e.mem.cache[level * 64]), 0);
            s.depth += 1;
        }
    }
}
// advance (nearby_state.rs:195-219)
__device__ __forceinline__ void kopt_advance(const KoptEnv& e, KoptS& s) {
    if (s.done || s.depth == 0) {
        s.done = 1;
        return;
    }
    const uint32_t t = s.depth - 1;
    const uint32_t next_index = s.idx(t) + 1;
    if (next_index < s.cnt(t)) {
        s.set(t, kopt_uni((uint32_t)e.mem.cache[t * 64 + next_index]), next_index);
        return;
    }
    if (kopt_backtrack(e, s))
        kopt_extend(e, s);
    else
        s.done = 1;
}
// Short routes on a compact matrix: nearby_positions(origin) of EVERY origin of the entity in one pass -- one batched
// gather of the len x len legs, then the stable (distance, position) rank of each (origin, position) pair -- so the
// level builds of the depth-first generator read LDS only.  0xFFFFFFFF (not finite) orders last and ties by position,
// exactly like the f64 INFINITY keys of the general path; finite legs < 2^32 compare like their f64 values.
__device__ __forceinline__ void kopt_rank_all_origins(const KoptEnv& e, uint32_t entity, uint32_t len) {
    const uint32_t base = e.off[entity], lane = e.lane;
    const uint32_t dim = (uint32_t)e.lm->dim;
    for (uint32_t idx = lane; idx < len * KOPT_FAST_LEN; idx += 64) {
        const uint32_t o = idx / KOPT_FAST_LEN, p = idx % KOPT_FAST_LEN;
        if (p < len) e.mem.dist[idx] = e.lm->mat32[(uint32_t)e.visits[base + o] * dim + (uint32_t)e.visits[base + p]];
    }
    kopt_sync();
    for (uint32_t idx = lane; idx < len * KOPT_FAST_LEN; idx += 64) {
        const uint32_t o = idx / KOPT_FAST_LEN, p = idx % KOPT_FAST_LEN;
        if (p >= len || p == o) continue;
        const uint32_t kp = e.mem.dist[idx];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < len; ++q) {
            const uint32_t kq = e.mem.dist[o * KOPT_FAST_LEN + q];
            rank += (q != o && (kq < kp || (kq == kp && q < p))) ? 1u : 0u;
        }
        if (rank < KOPT_FAST_LEN) e.mem.near[o * KOPT_FAST_LEN + rank] = (uint8_t)p;
    }
    kopt_sync();
}

// NearbyCutState::new (nearby_state.rs:70-111)
__device__ __forceinline__ void kopt_open_entity(const KoptEnv& e, KoptS& s, uint32_t entity, uint32_t len) {
    s.entity = entity;
    s.len = len;
    s.active = 1;
    s.done = 0;
    s.depth = 0;
    s.first_offset = 0;
    s.n0 = s.n1 = s.n2 = 0;
    s.i0 = s.i1 = s.i2 = 0;
    s.p0 = s.p1 = s.p2 = 0;
    s.fast = 0;
    if (len < 4u * e.mseg) {
        s.done = 1;
        return;
    }
    s.fcount = len - 3u * e.mseg - e.mseg + 1u;  // positions min_seg ..= len - 3 * min_seg
    uint32_t st = 0, sd = 1;
    e.ctx.perm_params(s.fcount, (SALT_KN_STATE ^ e.desc ^ (uint64_t)entity) ^ SALT_KN_FIRST, st, sd);
    s.fst = kopt_uni(st);
    s.fsd = kopt_uni(sd);
    s.set(0, kopt_first_position(e, s, 0), 0);
    s.depth = 1;
    s.fast = (len <= KOPT_FAST_LEN && e.lm->mat32 != nullptr) ? 1u : 0u;
    if (s.fast) kopt_rank_all_origins(e, entity, len);
}
// next_cuts (nearby_state.rs:221-240)
__device__ __forceinline__ bool kopt_next_cuts(const KoptEnv& e, KoptS& s, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
    kopt_extend(e, s);
    if (s.done || s.depth != 3) return false;
    c1 = s.p0, c2 = s.p1, c3 = s.p2;
    kopt_advance(e, s);
    return true;
}

#endif  // __HIPCC__

}  // namespace sf
