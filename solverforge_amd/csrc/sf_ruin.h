// List ruin leaf of the generic N-leaf engine (one wavefront = one replica): the seventh leaf of the reference's
// default list policy (runtime/compiler/default_local_search/policy/list.rs:24-33,193-199).
//
// Reference semantics restated (paths under crates/solverforge-solver/src/):
//   heuristic/selector/list_kernel/ruin.rs:38-144          RuinCursor: source list, ruin count and a partial Fisher-Yates of the
//                                                          positions, all drawn from the cursor's SmallRng
//   runtime/compiler/executor/list_leaf/cursor.rs:58-72,112-145   per-solve stream state: one SmallRng seeded from
//                                                          scoped_seed(random_seed, descriptor, variable, "list_ruin_move_selector"),
//                                                          one u64 drawn per cursor open, XOR context.offset_seed(salt)
//   runtime/compiler/executor/list_leaf/cursor/probe.rs:218-238   unrestricted source pool: non-empty lists (<= max_source_list_len)
//   heuristic/move/list_kernel/ruin.rs:131-281             ruin_do_move: remove, then greedy recreate -- every remaining element x
//                                                          every list x every position is trial-inserted and fully scored, the
//                                                          strictly best (first of equals) is placed, until nothing remains
//   phase/localsearch/evaluation.rs:52-60                  the candidate's score is the score after the recreate; one
//                                                          score_calculation per candidate
// rand's xoshiro256++ / random_range are not in the reference tree (Cargo.lock:314-338): restated from the published
// algorithm, PARITY UNPINNED against the reference, pinned between the oracle (oracle/sfo_core.hpp) and this file.
//
// GPU formulation.  The reference does O(removed^2 x (elements + lists)) do/score/undo round trips per candidate; here the
// whole wave scores one round of the recreate at once: the insertion slots of up to 64 consecutive lists are laid onto the
// lanes (DPP scan + shuffle search, map_slots_to_groups), each lane prices its slot for every remaining element (three
// matrix legs + the capacity overshoot of the destination) and keeps one running lexicographic best of (score, -(element,
// list, position)); a wave max + min picks the placement.  The removed elements are PARKED at the end of their source list
// (a stable partition inside that one list), so a placement is an ordinary list-change commit whose shift is bounded by the
// distance between the two lists, and a trial evaluation undoes itself with the inverse changes: the replica's LDS state is
// bit-identical before and after.  A committed ruin simply skips the undo.
#pragma once
#include <stdint.h>

#include "sf_list_model.h"

namespace sf {

constexpr uint64_t SALT_RUIN_SEED = 0x71578011C0DE0001ULL;  // list_leaf/cursor.rs:67
constexpr uint32_t RUIN_MAX_COUNT = 6;                      // elements per ruin (reference default 2..=5; wire format: six 16-bit positions)
constexpr uint32_t RUIN_MAX_MOVES = 16;                     // moves_per_step (reference default 10)

// Address-space qualified pointers for the out-of-line recreate: behind a call boundary the compiler cannot tell that a plain
// pointer is LDS and emits FLAT accesses for all of them (measured: 550 flat instructions, no ds_read, and a recreate five
// times slower than its instruction count explains).  With these types the same code compiles to ds_read / global_load.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) int64_t lds_i64;
typedef __attribute__((address_space(1))) const uint32_t glb_cu32;
typedef __attribute__((address_space(1))) const int32_t glb_ci32;
typedef __attribute__((address_space(1))) const int64_t glb_ci64;

// what the recreate reads of the list model, by value (uniform registers), pointers typed as global memory
struct RuinModel {
    int32_t V, dim, depot;
    int32_t cap_level, dist_level;
    int64_t capacity, cap_weight, dist_weight;
    glb_cu32* mat32;
    glb_ci64* mat;
    glb_ci32* demand;
};
__device__ __forceinline__ RuinModel ruin_model(const ListModel& m) {
    return RuinModel{m.V, m.dim, m.depot, m.cap_level, m.dist_level, m.capacity, m.cap_weight, m.dist_weight,
                     (glb_cu32*)m.mat32, (glb_ci64*)m.mat, (glb_ci32*)m.demand};
}
template <int L, class M>
__device__ __forceinline__ ScoreV<L> ruin_apply_delta(const M& m, const ScoreV<L>& cur, const ListDelta& d) {  // == apply_delta
    ScoreV<L> s = cur;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (k == m.cap_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.cap_weight * (uint64_t)d.d_cap));
        if (k == m.dist_level) s.v[k] = wsub(s.v[k], (int64_t)((uint64_t)m.dist_weight * (uint64_t)d.d_dist));
    }
    return s;
}
template <class M>
__device__ __forceinline__ int64_t ruin_dist_cost(const M& m, uint32_t from, uint32_t to) {  // == dist_cost (problem_data.rs:28-31)
    const int64_t v = m.mat[(size_t)from * (size_t)m.dim + to];
    return (v >= 0 && v != UNREACHABLE) ? v : MAX_SAFE_LEG_COST;
}

// Per-replica LDS block of the leaf.
struct RuinLds {
    static constexpr size_t CAND_WORDS = 8;  // u16: list, count, positions[6]
    static constexpr size_t bytes = 8 * 8 + RUIN_MAX_MOVES * (CAND_WORDS * 2 + 4 * 8) + 128;
    static_assert(bytes == RUIN_LDS_BYTES, "GCarve reserves RUIN_LDS_BYTES");
    uint64_t* prng;   // [4] per-solve stream (loaded at launch start, stored at launch end)
    uint64_t* crng;   // [4] cursor stream of this step
    int64_t* score;   // [RUIN_MAX_MOVES][4] trial score of every generated candidate
    uint16_t* cand;   // [RUIN_MAX_MOVES][CAND_WORDS]
    uint16_t* work;   // [64]: removed nodes [8], placements (list, position, next element, old edge) [8][4], remaining nodes [8]
    __device__ explicit RuinLds(unsigned char* base) {
        prng = (uint64_t*)base;
        crng = prng + 4;
        score = (int64_t*)(crng + 4);
        cand = (uint16_t*)(score + RUIN_MAX_MOVES * 4);
        work = cand + RUIN_MAX_MOVES * CAND_WORDS;
    }
};

// rand `UniformInt<u32>::sample_single_inclusive` (Canon's method, one extra step); `next_u32` of xoshiro256++ is the upper
// half of next_u64.  low <= high < 2^32 - 1.
__device__ __forceinline__ uint32_t ruin_random_range(SaRng& g, uint32_t low, uint32_t high) {
    const uint32_t range = high - low + 1u;
    const uint64_t m = (uint64_t)(uint32_t)(g.next() >> 32) * range;
    uint32_t result = (uint32_t)(m >> 32);
    const uint32_t lo_order = (uint32_t)m;
    if (lo_order > 0u - range) {
        const uint32_t new_hi = (uint32_t)(((uint64_t)(uint32_t)(g.next() >> 32) * range) >> 32);
        if ((uint64_t)lo_order + new_hi > 0xFFFFFFFFull) result += 1;
    }
    return low + result;
}

__device__ __forceinline__ bool ruin_eligible(const RuinParams& rp, uint32_t len) {
    return len > 0 && (rp.max_source_len <= 0 || len <= (uint32_t)rp.max_source_len);
}

// Cursor open (once per step): one draw of the per-solve stream seeds the cursor stream; returns the pool size.
__device__ __forceinline__ uint32_t ruin_open_cursor(const RuinParams& rp, const RuinLds& rl, const StreamCtx& ctx, const uint32_t* off, int V,
                                                     bool advance, uint32_t lane) {
    SaRng pr{rl.prng[0], rl.prng[1], rl.prng[2], rl.prng[3]};
    const uint64_t draw = uni64(pr.next());
    uint64_t state = draw ^ (ctx.canonical() ? 0ull : ctx.mixed_seed(SALT_RUIN_SEED));
    uint64_t s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // SmallRng::seed_from_u64: splitmix64 expansion
        state += 0x9E3779B97F4A7C15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        s[i] = z ^ (z >> 31);
    }
    uint32_t pool = 0;
    for (uint32_t base = 0; base < (uint32_t)V; base += 64) {
        const uint32_t e = base + lane;
        const bool el = e < (uint32_t)V && ruin_eligible(rp, off[e + 1] - off[e]);
        pool += (uint32_t)__popcll(__ballot(el));
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rl.crng[i] = s[i];
        if (advance) {  // a dry run (sf_step_generate) peeks at the next draw without consuming it
            rl.prng[0] = pr.s0, rl.prng[1] = pr.s1, rl.prng[2] = pr.s2, rl.prng[3] = pr.s3;
        }
    }
    wave_sync();
    return uni(pool);
}

// next_unrestricted_move (selector/list_kernel/ruin.rs:88-103): candidate `c` of this step into the table.
__device__ __forceinline__ void ruin_next_candidate(const RuinParams& rp, const RuinLds& rl, const uint32_t* off, int V, uint32_t pool,
                                                    uint32_t c, uint32_t lane) {
    SaRng g{uni64(rl.crng[0]), uni64(rl.crng[1]), uni64(rl.crng[2]), uni64(rl.crng[3])};
    uint32_t pick = uni(ruin_random_range(g, 0u, pool - 1u));
    uint32_t ent = 0, len = 0;
    for (uint32_t base = 0; base < (uint32_t)V; base += 64) {  // the pick-th eligible list in list order
        const uint32_t e = base + lane;
        const uint32_t l = e < (uint32_t)V ? off[e + 1] - off[e] : 0u;
        const bool el = e < (uint32_t)V && ruin_eligible(rp, l);
        const uint64_t m = __ballot(el);
        const uint32_t n = (uint32_t)__popcll(m);
        if (pick < n) {
            const uint64_t hit = __ballot(el && mbcnt64(m) == pick);
            const int src = __ffsll((unsigned long long)hit) - 1;
            ent = uni((uint32_t)__shfl((int)e, src));
            len = uni((uint32_t)__shfl((int)l, src));
            break;
        }
        pick -= n;
    }
    const uint32_t mn = (uint32_t)rp.min_count < len ? (uint32_t)rp.min_count : len;
    const uint32_t mx = (uint32_t)rp.max_count < len ? (uint32_t)rp.max_count : len;
    const uint32_t cnt = mn == mx ? mn : uni(ruin_random_range(g, mn, mx));  // choose_ruin_count (:78-86)
    // partial Fisher-Yates of (0..len) kept sparse: lane t < n_ov holds one displaced entry (position, value); position i < the
    // current index is never read again, so only the partner side of every swap is recorded
    uint32_t ov_pos = 0, ov_val = 0, n_ov = 0, mine = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t j = uni(ruin_random_range(g, i, len - 1u));
        const uint64_t mi = __ballot(lane < n_ov && ov_pos == i);
        const uint32_t vi = mi ? uni((uint32_t)__shfl((int)ov_val, __ffsll((unsigned long long)mi) - 1)) : i;
        const uint64_t mj = __ballot(lane < n_ov && ov_pos == j);
        const uint32_t vj = mj ? uni((uint32_t)__shfl((int)ov_val, __ffsll((unsigned long long)mj) - 1)) : j;
        const uint32_t slot = mj ? (uint32_t)(__ffsll((unsigned long long)mj) - 1) : n_ov;
        if (lane == slot) ov_pos = j, ov_val = vi;
        if (!mj) n_ov += 1;
        if (lane == i) mine = vj;  // indices[i] after the swap
    }
    // single_ruin_source (move/list_kernel/ruin.rs:28-32): ascending positions (distinct values: rank by counting)
    uint32_t rank = 0;
    for (uint32_t s = 0; s < cnt; ++s) rank += (uint32_t)__shfl((int)mine, (int)s) < mine ? 1u : 0u;
    uint16_t* cd = rl.cand + (size_t)c * RuinLds::CAND_WORDS;
    if (lane < cnt) cd[2 + rank] = (uint16_t)mine;
    if (lane >= cnt && lane < RUIN_MAX_COUNT) cd[2 + lane] = 0;
    if (lane == 0) {
        cd[0] = (uint16_t)ent;
        cd[1] = (uint16_t)cnt;
        rl.crng[0] = g.s0, rl.crng[1] = g.s1, rl.crng[2] = g.s2, rl.crng[3] = g.s3;
    }
    wave_sync();
}

template <class M>
__device__ __forceinline__ int64_t ruin_leg(const M& m, uint32_t from, uint32_t to) {
    if (m.mat32) {
        const uint32_t v = m.mat32[from * (uint32_t)m.dim + to];
        return v != 0xFFFFFFFFu ? (int64_t)v : MAX_SAFE_LEG_COST;
    }
    return ruin_dist_cost(m, from, to);
}

__device__ __forceinline__ int64_t ruin_wave_sum(int64_t v) {
#pragma unroll
    for (int mlane = 32; mlane >= 1; mlane >>= 1) v = wadd(v, (int64_t)shfl_xor_u64((uint64_t)v, mlane));
    return v;
}
__device__ __forceinline__ uint64_t ruin_wave_min_u64(uint64_t v) {
#pragma unroll
    for (int mlane = 32; mlane >= 1; mlane >>= 1) {
        const uint64_t o = shfl_xor_u64(v, mlane);
        v = o < v ? o : v;
    }
    return v;
}

// depot -> first `len` elements of list e -> depot (an empty list costs nothing, like the route-distance uni constraint)
template <class M, class P16>
__device__ __forceinline__ int64_t ruin_route_distance(const M& m, P16 visits, uint32_t o, uint32_t len, uint32_t lane) {
    int64_t acc = 0;
    if (len == 0) return 0;
    const uint32_t depot = (uint32_t)m.depot;
    for (uint32_t q = lane; q <= len; q += 64) {
        const uint32_t from = q > 0 ? (uint32_t)visits[o + q - 1] : depot;
        const uint32_t to = q < len ? (uint32_t)visits[o + q] : depot;
        acc = wadd(acc, ruin_leg(m, from, to));
    }
    return ruin_wave_sum(acc);
}

#ifdef SF_PHASE_PROFILE  // diagnostic build: shader clocks per part of the recreate (scripts/phase_probe_generic.py)
__device__ unsigned long long g_rphase[8];
#define RPH_DECL uint64_t rph_t = clock64(), rph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RPH(i)                           \
    {                                    \
        const uint64_t _t = clock64();   \
        rph_acc[i] += _t - rph_t;        \
        rph_t = _t;                      \
    }
#define RPH_DUMP \
    if (lane == 0) for (int _k = 0; _k < 8; ++_k) atomicAdd(&g_rphase[_k], (unsigned long long)rph_acc[_k]);
#else
#define RPH_DECL
#define RPH(i)
#define RPH_DUMP
#endif

// Inlined since round 5 (it was an out-of-line call): with the candidates scored at the start of the step the recreate no longer sits in
// the fill / replay loop nest, and without the call boundary the seven-leaf step of CVRP-1000 went from 6.6 M to 3.4 M shader clocks
// (profiles/r05_phase7_variants.txt).  -DSF_RUIN_NOINLINE restores the call (A/B).
#ifdef SF_RUIN_NOINLINE
#define SF_RUIN_ATTR __attribute__((noinline))
#else
#define SF_RUIN_ATTR __forceinline__
#endif

template <bool M32, class M>
__device__ __forceinline__ int64_t ruin_leg_t(const M& m, uint32_t from, uint32_t to) {
    if (M32) {
        const uint32_t v = m.mat32[from * (uint32_t)m.dim + to];
        return v != 0xFFFFFFFFu ? (int64_t)v : MAX_SAFE_LEG_COST;
    }
    return ruin_dist_cost(m, from, to);
}

// ---- LDS fast path (symmetric matrix whose finite legs fit 16 bits) ---------------------------------------------------------
// A round of the recreate prices ~ (elements + lists) slots x remaining elements; as matrix gathers that is ~20 K scattered
// 4-byte reads per candidate and wave, each pulling a 128-byte line through L2 (measured: 4.7 ms per local-search step at
// CVRP-1000, 85 % of it here).  The fast path keeps everything a slot needs in LDS: `edge[n]` = the leg entering element n in
// the replica's current lists (edge_end[e] = last element of list e -> depot), built once per step and patched by the
// candidate's own removals / placements, and `row[]` = the matrix row of the element being placed (one coalesced 2-4 KB
// read per element and round).  A slot's delta is then row[prev] + row[next] - edge: three LDS reads, no global traffic.
template <class P16>
struct RuinFastT {
    P16 edge;      // [dim]  0xFFFF = not finite (-> MAX_SAFE_LEG_COST)
    P16 edge_end;  // [V]    0 for an empty list
    P16 row;       // [dim]
    P16 slot;      // [n_cap + V] list of every insertion slot of the current round (slots numbered list by list)
};
typedef RuinFastT<uint16_t*> RuinFast;         // as the kernel holds it
typedef RuinFastT<lds_u16*> RuinFastLds;       // as the out-of-line recreate takes it
__device__ __forceinline__ int64_t ruin_leg16(uint32_t v) { return v != 0xFFFFu ? (int64_t)v : MAX_SAFE_LEG_COST; }
template <class M>
__device__ __forceinline__ uint32_t ruin_raw16(const M& m, uint32_t from, uint32_t to) {
    const uint32_t v = m.mat32[from * (uint32_t)m.dim + to];
    return v >= 0xFFFFu ? 0xFFFFu : v;
}

// slot prefix of the current lists (the source list counts without its `parked` tail)
template <class M, class P32>
__device__ __forceinline__ void ruin_slot_prefix(const M& lm, P32 off, P32 sbase, uint32_t ent, uint32_t parked, int skip_empty) {
    const uint32_t lane = threadIdx.x & 63u, V = (uint32_t)lm.V;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < V; base += 64) {
        const uint32_t rk = base + lane;
        uint32_t c = 0;
        if (rk < V) {
            const uint32_t l = off[rk + 1] - off[rk] - (rk == ent ? parked : 0u);
            c = (skip_empty && l == 0) ? 0u : l + 1u;
        }
        const uint32_t inc = wave_incl_scan(c);
        if (rk < V) sbase[rk] = carry + inc - c;
        carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) sbase[V] = carry;
    wave_sync();
}

// edges of every list, once per step (the lists do not change inside a step except under a candidate's own trial)
__device__ __forceinline__ void ruin_build_edges(const ListModel& lm, uint16_t* visits, uint32_t* off, uint32_t* sbase, const RuinFast& rf) {
    constexpr int U = 4;
    const uint32_t lane = threadIdx.x & 63u, V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    ruin_slot_prefix(lm, off, sbase, 0xFFFFFFFFu, 0u, 0);
    const uint32_t total = uni(sbase[V]);
    uint32_t top = 1;
    while (top < V) top <<= 1;
    for (uint32_t t0 = 0; t0 < total; t0 += 64u * U) {
        uint32_t e[U], nx[U], val[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + (uint32_t)u * 64u + lane;
            valid[u] = t < total;
            const uint32_t tt = valid[u] ? t : 0u;
            uint32_t lo = 0;
            for (uint32_t stepw = top >> 1; stepw; stepw >>= 1) {
                const uint32_t cand = lo + stepw;
                if (cand < V && sbase[cand] <= tt) lo = cand;
            }
            e[u] = lo;
            const uint32_t b0 = sbase[lo], le = sbase[lo + 1] - b0 - 1u, o = tt - b0, ob = off[lo];
            const uint32_t prev = o > 0 ? (uint32_t)visits[ob + o - 1] : depot;
            nx[u] = o < le ? (uint32_t)visits[ob + o] : 0xFFFFFFFFu;
            val[u] = le == 0 ? 0u : ruin_raw16(lm, prev, nx[u] != 0xFFFFFFFFu ? nx[u] : depot);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (valid[u]) {
                if (nx[u] != 0xFFFFFFFFu)
                    rf.edge[nx[u]] = (uint16_t)val[u];
                else
                    rf.edge_end[e[u]] = (uint16_t)val[u];
            }
    }
    wave_sync();
}

// edges of the first `len` elements of list e (one gather pass); returns the list's distance
template <class M, class P16>
__device__ __forceinline__ int64_t ruin_rebuild_list_edges(const M& lm, P16 visits, uint32_t o, uint32_t len, uint32_t e, const RuinFastT<P16>& rf) {
    const uint32_t lane = threadIdx.x & 63u, depot = (uint32_t)lm.depot;
    int64_t acc = 0;
    if (len == 0) {
        if (lane == 0) rf.edge_end[e] = 0;
        wave_sync();
        return 0;
    }
    for (uint32_t q = lane; q <= len; q += 64) {
        const uint32_t from = q > 0 ? (uint32_t)visits[o + q - 1] : depot;
        const uint32_t to = q < len ? (uint32_t)visits[o + q] : depot;
        const uint32_t v = ruin_raw16(lm, from, to);
        if (q < len)
            rf.edge[to] = (uint16_t)v;
        else
            rf.edge_end[e] = (uint16_t)v;
        acc = wadd(acc, ruin_leg16(v));
    }
    wave_sync();
    return ruin_wave_sum(acc);
}

// the distance of list e from the edge table (no matrix access)
template <class P16>
__device__ __forceinline__ int64_t ruin_list_distance_from_edges(P16 visits, uint32_t o, uint32_t len, uint32_t e, const RuinFastT<P16>& rf) {
    const uint32_t lane = threadIdx.x & 63u;
    int64_t acc = 0;
    if (len == 0) return 0;
    for (uint32_t q = lane; q <= len; q += 64) acc = wadd(acc, ruin_leg16(q < len ? (uint32_t)rf.edge[visits[o + q]] : (uint32_t)rf.edge_end[e]));
    return ruin_wave_sum(acc);
}

// what a lane remembers of its best slot besides the score: the three raw legs and the element after the slot
struct RuinPick {
    uint32_t da, db, d0, next;  // next = 0xFFFFFFFF: the slot is a list's end
};

// list of every slot, once per round (lane = one list, a list of len elements owns len + 1 consecutive slots)
__device__ __forceinline__ void ruin_build_slot_lists(uint32_t V, const lds_u32* sbase, const RuinFastLds& rf) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t e = lane; e < V; e += 64) {
        const uint32_t b0 = sbase[e], b1 = sbase[e + 1];
        for (uint32_t t = b0; t < b1; ++t) rf.slot[t] = (uint16_t)e;
    }
    wave_sync();
}

// One element of one round on the fast path: `rf.row` holds the row of element x.  Every LDS read of a slot depends on the one
// before it (slot -> list -> bounds -> neighbours -> legs: four round trips), so U slots per lane are resolved side by side.
template <int L>
__device__ __forceinline__ void ruin_scan_element_fast(const RuinModel& lm, const lds_u16* visits, const lds_u32* off, const lds_i64* load,
                                                       const lds_u32* sbase, const RuinFastLds& rf, uint32_t ri, int64_t dx, uint32_t ent,
                                                       int64_t parked_dem, const ScoreV<L>& s, ScoreV<L>& bs, uint64_t& bkey, bool& has, RuinPick& pick) {
    constexpr int U = 8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    const bool has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    const uint32_t total = uni(sbase[V]);
    for (uint32_t t0 = 0; t0 < total; t0 += 64u * U) {
        uint32_t tt[U], e[U], b0[U], b1[U], ob[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + (uint32_t)u * 64u + lane;
            valid[u] = t < total;
            tt[u] = valid[u] ? t : 0u;
            e[u] = rf.slot[tt[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b0[u] = sbase[e[u]];
            b1[u] = sbase[e[u] + 1];
            ob[u] = off[e[u]];
        }
        uint32_t o[U], pv[U], nx[U];
        int64_t ld[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = tt[u] - b0[u];
            const uint32_t le = b1[u] - b0[u] - 1u;
            pv[u] = o[u] > 0 ? (uint32_t)visits[ob[u] + o[u] - 1] : depot;
            nx[u] = o[u] < le ? (uint32_t)visits[ob[u] + o[u]] : 0xFFFFFFFFu;
            ld[u] = has_cap ? load[e[u]] : 0;
        }
        uint32_t da[U], db[U], d0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            da[u] = rf.row[pv[u]];
            db[u] = rf.row[nx[u] != 0xFFFFFFFFu ? nx[u] : depot];
            d0[u] = nx[u] != 0xFFFFFFFFu ? (uint32_t)rf.edge[nx[u]] : (uint32_t)rf.edge_end[e[u]];  // an empty list's edge_end is 0
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ListDelta d{0, 0, true};
            d.d_dist = wsub(wadd(ruin_leg16(da[u]), ruin_leg16(db[u])), ruin_leg16(d0[u]));
            if (has_cap) {
                const int64_t l0 = e[u] == ent ? wsub(ld[u], parked_dem) : ld[u];
                d.d_cap = wsub(over_cap(wadd(l0, dx), lm.capacity), over_cap(l0, lm.capacity));
            }
            const ScoreV<L> sc = ruin_apply_delta<L>(lm, s, d);
            const uint64_t key = ((uint64_t)ri << 32) | ((uint64_t)e[u] << 16) | (uint64_t)o[u];
            const int cmp = has ? score_cmp<L>(sc, bs) : 1;
            if (valid[u] && (cmp > 0 || (cmp == 0 && key < bkey))) {
                bs = sc;
                bkey = key;
                has = true;
                pick = RuinPick{da[u], db[u], d0[u], nx[u]};
            }
        }
    }
}

#ifndef SF_RUIN_SMALL_U
#define SF_RUIN_SMALL_U 4  // slots per lane in flight in the 32-bit scan
#endif
// The same scan in 32-bit arithmetic, for models whose every trial delta provably fits (ListModel::small32: all legs finite and
// < 2^26, small weights / loads -- the bounds of the wave engine's delta-space replay): a lane keeps its best as per-level
// DELTAS against the round's base score, which order exactly like the full scores do.  Branch-free: the PMC profile of the first
// version showed ~137 VALU instructions per slot, most of them control flow around the per-level compare and the update.
// A lane visits its slots in increasing (element, slot number) order and slot numbers order like (list, position), so inside a
// lane a tie never replaces the running best: the update is "strictly better" only, and the key (element << 32 | slot number)
// is needed by the cross-lane reduction alone.  The winner's (list, position) is decoded from its slot number afterwards.
template <int L>
__device__ __forceinline__ void ruin_scan_element_small(const RuinModel& lm, const lds_u16* visits, const lds_u32* off, const lds_i64* load,
                                                        const lds_u32* sbase, const RuinFastLds& rf, uint32_t ri, int32_t dx, uint32_t ent,
                                                        int32_t parked_dem, int32_t (&bdv)[L], uint64_t& bkey, bool& has, RuinPick& pick) {
    constexpr int U = SF_RUIN_SMALL_U;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    const bool has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    const int32_t cap32 = (int32_t)lm.capacity;
    int32_t ca[L], cb[L];  // per-level coefficients: delta[k] = ca[k] * (capacity overshoot delta) + cb[k] * (distance delta)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        ca[k] = (has_cap && k == lm.cap_level) ? -(int32_t)lm.cap_weight : 0;
        cb[k] = k == lm.dist_level ? -(int32_t)lm.dist_weight : 0;
    }
    const lds_u32* load32 = (const lds_u32*)load;  // low words of the i64 loads (small32: loads < 2^28)
    const uint32_t total = uni(sbase[V]);
    for (uint32_t t0 = 0; t0 < total; t0 += 64u * U) {
        uint32_t tt[U], e[U], b0[U], b1[U], ob[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + (uint32_t)u * 64u + lane;
            valid[u] = t < total;
            tt[u] = valid[u] ? t : 0u;
            e[u] = rf.slot[tt[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b0[u] = sbase[e[u]];
            b1[u] = sbase[e[u] + 1];
            ob[u] = off[e[u]];
        }
        uint32_t pvr[U], nxr[U], o[U];
        int32_t ld[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = tt[u] - b0[u];
            const uint32_t ip = ob[u] + o[u];
            pvr[u] = visits[ip - (o[u] > 0 ? 1u : 0u)];  // read unconditionally (a stale word past a list is discarded below)
            nxr[u] = visits[ip];
            ld[u] = has_cap ? (int32_t)load32[2u * e[u]] : 0;
        }
        uint32_t da[U], db[U], d0[U], nxe[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool at_end = o[u] + 1u >= b1[u] - b0[u];  // o == logical length
            const uint32_t pv = o[u] > 0 ? pvr[u] : depot;
            nxe[u] = at_end ? 0xFFFFu : nxr[u];
            da[u] = rf.row[pv];
            db[u] = rf.row[at_end ? depot : nxr[u]];
            const lds_u16* dp = at_end ? rf.edge_end + e[u] : rf.edge + nxr[u];  // an empty list's edge_end is 0
            d0[u] = *dp;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t dd = (int32_t)da[u] + (int32_t)db[u] - (int32_t)d0[u];
            const int32_t l0 = ld[u] - (e[u] == ent ? parked_dem : 0);
            const int32_t over1 = l0 + dx - cap32, over0 = l0 - cap32;
            const int32_t dc = (over1 > 0 ? over1 : 0) - (over0 > 0 ? over0 : 0);
            int32_t dv[L];
            bool gt = false, eq = true;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                dv[k] = ca[k] * dc + cb[k] * dd;
                gt = gt || (eq && dv[k] > bdv[k]);
                eq = eq && dv[k] == bdv[k];
            }
            const bool take = valid[u] && (!has || gt);
#pragma unroll
            for (int k = 0; k < L; ++k) bdv[k] = take ? dv[k] : bdv[k];
            bkey = take ? (((uint64_t)ri << 32) | (uint64_t)tt[u]) : bkey;
            pick.da = take ? da[u] : pick.da;
            pick.db = take ? db[u] : pick.db;
            pick.d0 = take ? d0[u] : pick.d0;
            pick.next = take ? (nxe[u] == 0xFFFFu ? 0xFFFFFFFFu : nxe[u]) : pick.next;
            has = has || take;
        }
    }
}

// One round of the recreate (general path): every lane prices insertion slots for the NR remaining elements and keeps its running best.
// Slots are numbered list by list (slot_base[e] = slots of the lists before e, a list of len elements has len + 1 slots, a
// skipped empty list none); a lane finds its slot's list with a fixed-depth binary search, so the U chunks of one group are
// independent of each other: their LDS reads and their 1 + 2 * NR matrix gathers per slot are all in flight together (the
// round is latency-bound: one memory round trip per group instead of one per chunk and element).
template <int L, int NR, bool M32>
__device__ __forceinline__ void ruin_scan_round(const RuinModel& lm, const lds_u16* visits, const lds_u32* off, const lds_i64* load,
                                                const lds_u16* rem, const lds_u32* sbase, uint32_t n_rem, uint32_t ent, int64_t parked_dem,
                                                const ScoreV<L>& s, ScoreV<L>& bs, uint64_t& bkey, bool& has) {
    constexpr int U = NR <= 2 ? 6 : (NR <= 4 ? 4 : 3);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    const bool has_dist = lm.dist_level >= 0, has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    const uint32_t total = uni(sbase[V]);
    uint32_t top = 1;
    while (top < V) top <<= 1;  // search steps top/2, top/4, .. 1
    uint32_t x[NR];
    int64_t dx[NR];
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        x[ri] = uni((uint32_t)rem[(uint32_t)ri < n_rem ? ri : 0]);
        dx[ri] = has_cap ? (int64_t)lm.demand[x[ri]] : 0;
    }
    for (uint32_t t0 = 0; t0 < total; t0 += 64u * U) {
        uint32_t e[U], o[U], prev[U], next[U];
        int64_t ld[U];
        bool valid[U], empty[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + (uint32_t)u * 64u + lane;
            valid[u] = t < total;
            const uint32_t tt = valid[u] ? t : 0u;
            uint32_t lo = 0;
            for (uint32_t stepw = top >> 1; stepw; stepw >>= 1) {  // last list whose first slot is <= tt
                const uint32_t cand = lo + stepw;
                if (cand < V && sbase[cand] <= tt) lo = cand;
            }
            e[u] = lo;
            const uint32_t b0 = sbase[lo], le = sbase[lo + 1] - b0 - 1u;
            o[u] = tt - b0;
            const uint32_t ob = off[lo];
            prev[u] = o[u] > 0 ? (uint32_t)visits[ob + o[u] - 1] : depot;
            next[u] = o[u] < le ? (uint32_t)visits[ob + o[u]] : depot;
            empty[u] = le == 0;
            ld[u] = has_cap ? (lo == ent ? wsub(load[lo], parked_dem) : load[lo]) : 0;
        }
        int64_t d0[U], da[U][NR], db[U][NR];
        if (has_dist) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d0[u] = ruin_leg_t<M32>(lm, prev[u], next[u]);
#pragma unroll
                for (int ri = 0; ri < NR; ++ri) {
                    da[u][ri] = ruin_leg_t<M32>(lm, prev[u], x[ri]);
                    db[u][ri] = ruin_leg_t<M32>(lm, x[ri], next[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int ri = 0; ri < NR; ++ri) {
                ListDelta d{0, 0, true};
                if (has_dist) d.d_dist = wsub(wadd(da[u][ri], db[u][ri]), empty[u] ? 0 : d0[u]);
                if (has_cap) d.d_cap = wsub(over_cap(wadd(ld[u], dx[ri]), lm.capacity), over_cap(ld[u], lm.capacity));
                const ScoreV<L> sc = ruin_apply_delta<L>(lm, s, d);
                const uint64_t key = ((uint64_t)ri << 32) | ((uint64_t)e[u] << 16) | (uint64_t)o[u];
                const int cmp = has ? score_cmp<L>(sc, bs) : 1;
                if (valid[u] && (uint32_t)ri < n_rem && (cmp > 0 || (cmp == 0 && key < bkey))) {  // strictly better, or the earlier of equals
                    bs = sc;
                    bkey = key;
                    has = true;
                }
            }
        }
    }
}

// ListChange commit (a, i) -> (b, j) on the typed LDS state: the kind == 2 branch of apply_list_move_wave
__device__ __forceinline__ void ruin_list_change(const RuinModel& m, lds_u16* visits, lds_u32* off, lds_i64* load, uint32_t a, uint32_t i, uint32_t b,
                                                 uint32_t j) {
    const uint32_t lane = threadIdx.x & 63u;
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
            load[a] = wsub(load[a], dx);
            load[b] = wadd(load[b], dx);
        }
    }
    wave_sync();
}

// ruin_do_move on the replica's LDS state for the candidate `cd` = (list, count, ascending positions).  Writes the score
// after the recreate to out_score[0..L); when `commit` is false the state is restored before returning.  `sbase` = V + 1
// words of LDS scratch (slot prefix of the current round); `rf.edge` != nullptr selects the LDS fast path (its edge table is
// valid for the current lists on entry and again on return of a trial).
template <int L>
__device__ SF_RUIN_ATTR void ruin_recreate_lds(const RuinModel lm_in, lds_u16* visits, lds_u32* off, lds_i64* load, const lds_u16* cd, lds_u16* work,
                                               lds_u32* sbase, const RuinFastLds rf, int fast, int skip_empty, bool commit, const ScoreV<L> cur,
                                               lds_i64* out_score) {  // fast: 0 general path, 1 LDS tables, 2 LDS tables + 32-bit deltas
    const uint32_t lane = threadIdx.x & 63u;
    // by-value arguments of an out-of-line function arrive in vector registers: make the model wave-uniform again so that
    // loops over lists / chunks are scalar loops and not per-lane waterfalls
    RuinModel lm;
    lm.V = (int32_t)uni((uint32_t)lm_in.V), lm.dim = (int32_t)uni((uint32_t)lm_in.dim), lm.depot = (int32_t)uni((uint32_t)lm_in.depot);
    lm.cap_level = (int32_t)uni((uint32_t)lm_in.cap_level), lm.dist_level = (int32_t)uni((uint32_t)lm_in.dist_level);
    lm.capacity = (int64_t)uni64((uint64_t)lm_in.capacity), lm.cap_weight = (int64_t)uni64((uint64_t)lm_in.cap_weight);
    lm.dist_weight = (int64_t)uni64((uint64_t)lm_in.dist_weight);
    lm.mat32 = (glb_cu32*)uni64((uint64_t)lm_in.mat32), lm.mat = (glb_ci64*)uni64((uint64_t)lm_in.mat), lm.demand = (glb_ci32*)uni64((uint64_t)lm_in.demand);
    skip_empty = (int)uni((uint32_t)skip_empty);
    fast = (int)uni((uint32_t)fast);
    commit = uni(commit ? 1u : 0u) != 0;
    const uint32_t ent = uni((uint32_t)cd[0]), cnt = uni((uint32_t)cd[1]);
    const bool has_dist = lm.dist_level >= 0, has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    lds_u16* rem0 = work;          // [8] removed nodes in removal (= ascending position) order
    lds_u16* place = work + 8;     // [8][4] (list, position, element after the slot or 0xFFFF, old raw edge) of every placement
    lds_u16* rem = work + 40;      // [8] remaining nodes, order preserved
    RPH_DECL
    ScoreV<L> s = cur;
    const uint32_t oe = uni(off[ent]), plen = uni(off[ent + 1]) - oe;  // physical length of the source list (never changes below
                                                                      // until an element leaves for another list)
    // ---- remove: stable partition of the source list into [kept .. | removed ..] ----
    const int64_t dist_before = !has_dist ? 0 : fast ? ruin_list_distance_from_edges(visits, oe, plen, ent, rf) : ruin_route_distance(lm, visits, oe, plen, lane);
    const int64_t load_before = has_cap ? load[ent] : 0;
    if (lane < cnt) {
        const uint32_t x = visits[oe + cd[2 + lane]];
        rem0[lane] = (uint16_t)x;
        rem[lane] = (uint16_t)x;
    }
    wave_sync();
    for (uint32_t c0 = 0; c0 < plen; c0 += 64) {  // ascending chunks: reads never trail the writes of an earlier chunk
        const uint32_t q = c0 + lane;  // new position
        uint32_t nv = 0;
        if (q < plen - cnt) {
            uint32_t src = q;  // q-th kept element = old position q + #removed positions <= it
            for (uint32_t j = 0; j < cnt; ++j) src += (uint32_t)cd[2 + j] <= src ? 1u : 0u;
            nv = visits[oe + src];
        } else if (q < plen) {
            nv = rem0[q - (plen - cnt)];
        }
        wave_sync();
        if (q < plen) visits[oe + q] = (uint16_t)nv;
        wave_sync();
    }
    int64_t parked_dem = 0;  // demand of the parked tail: the source list's logical load excludes it
    if (has_cap)
        for (uint32_t j = 0; j < cnt; ++j) parked_dem = wadd(parked_dem, (int64_t)lm.demand[rem0[j]]);
    {
        ListDelta d{0, 0, true};
        if (has_dist) {
            const int64_t dist_after = fast ? ruin_rebuild_list_edges(lm, visits, oe, plen - cnt, ent, rf) : ruin_route_distance(lm, visits, oe, plen - cnt, lane);
            d.d_dist = wsub(dist_after, dist_before);
        }
        if (has_cap) d.d_cap = wsub(over_cap(wsub(load_before, parked_dem), lm.capacity), over_cap(load_before, lm.capacity));
        s = ruin_apply_delta<L>(lm, s, d);
    }
    RPH(0)
    // ---- recreate: one wave-wide round per remaining element ----
    uint32_t n_rem = cnt, n_pl = 0;
    bool rolled_back = false;
    // fast path: the matrix row of the next element to scan is fetched (16 coalesced loads per lane in flight) while the
    // current element is scanned, and the first row of a round while the previous placement is applied
    uint32_t pre[16];
    auto fetch_row = [&](uint32_t xn) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t c = (uint32_t)k * 64u + lane;
            pre[k] = c < (uint32_t)lm.dim ? lm.mat32[xn * (uint32_t)lm.dim + c] : 0u;
        }
    };
    if (fast) fetch_row(uni((uint32_t)rem[0]));
    while (n_rem > 0) {
        ruin_slot_prefix(lm, off, sbase, ent, n_rem, skip_empty);
        RPH(1)
        ScoreV<L> bs;
#pragma unroll
        for (int k = 0; k < L; ++k) bs.v[k] = INT64_MIN;
        uint64_t bkey = ~0ull;
        bool has = false;
        RuinPick pick{0, 0, 0, 0};
        int32_t bdv[L];
#pragma unroll
        for (int k = 0; k < L; ++k) bdv[k] = INT32_MIN;
        if (fast) {
            ruin_build_slot_lists((uint32_t)lm.V, sbase, rf);
            for (uint32_t ri = 0; ri < n_rem; ++ri) {
                const uint32_t x = uni((uint32_t)rem[ri]);
                wave_sync();  // the previous element's scan is done with the row
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t c = (uint32_t)k * 64u + lane;
                    if (c < (uint32_t)lm.dim) rf.row[c] = (uint16_t)(pre[k] >= 0xFFFFu ? 0xFFFFu : pre[k]);
                }
                for (uint32_t c = 1024u + lane; c < (uint32_t)lm.dim; c += 64) {  // rows longer than the prefetch window
                    const uint32_t v = lm.mat32[x * (uint32_t)lm.dim + c];
                    rf.row[c] = (uint16_t)(v >= 0xFFFFu ? 0xFFFFu : v);
                }
                wave_sync();
                if (ri + 1 < n_rem) fetch_row(uni((uint32_t)rem[ri + 1]));
                RPH(6)
                if (fast == 2)
                    ruin_scan_element_small<L>(lm, visits, off, load, sbase, rf, ri, has_cap ? (int32_t)lm.demand[x] : 0, ent, (int32_t)parked_dem, bdv,
                                               bkey, has, pick);
                else
                    ruin_scan_element_fast<L>(lm, visits, off, load, sbase, rf, ri, has_cap ? (int64_t)lm.demand[x] : 0, ent, parked_dem, s, bs, bkey, has,
                                              pick);
                RPH(2)
            }
            if (fast == 2 && has) {
#pragma unroll
                for (int k = 0; k < L; ++k) bs.v[k] = wadd(s.v[k], (int64_t)bdv[k]);
            }
        } else if (lm.mat32 || !has_dist) {  // without a distance constraint no leg is read: any instantiation will do
            switch (n_rem) {
                case 1: ruin_scan_round<L, 1, true>(lm, visits, off, load, rem, sbase, n_rem, ent, parked_dem, s, bs, bkey, has); break;
                case 2: ruin_scan_round<L, 2, true>(lm, visits, off, load, rem, sbase, n_rem, ent, parked_dem, s, bs, bkey, has); break;
                case 3: ruin_scan_round<L, 3, true>(lm, visits, off, load, rem, sbase, n_rem, ent, parked_dem, s, bs, bkey, has); break;
                default: ruin_scan_round<L, 6, true>(lm, visits, off, load, rem, sbase, n_rem, ent, parked_dem, s, bs, bkey, has); break;
            }
        } else {
            ruin_scan_round<L, 6, false>(lm, visits, off, load, rem, sbase, n_rem, ent, parked_dem, s, bs, bkey, has);
        }
        RPH(2)
        if (__ballot(has) == 0ull) {  // no destination at all: restore_removed_elements (:250-253)
            rolled_back = true;
            break;
        }
        const ScoreV<L> M = wave_max_score<L>(bs, has);
        const bool at_max = has && score_cmp<L>(bs, M) == 0;
        const uint64_t kmin = uni64(ruin_wave_min_u64(at_max ? bkey : ~0ull));
        const uint32_t ri = (uint32_t)(kmin >> 32);
        uint32_t be = (uint32_t)(kmin >> 16) & 0xFFFFu, bp = (uint32_t)kmin & 0xFFFFu;
        if (fast == 2) {  // the 32-bit scan keys its slots by slot number: decode (list, position)
            const uint32_t tw = (uint32_t)kmin;
            be = uni((uint32_t)rf.slot[tw]);
            bp = tw - uni(sbase[be]);
        }
        const uint32_t x = uni((uint32_t)rem[ri]);
        const int win = __ffsll((unsigned long long)__ballot(at_max && bkey == kmin)) - 1;
        const uint32_t w_da = uni((uint32_t)__shfl((int)pick.da, win)), w_db = uni((uint32_t)__shfl((int)pick.db, win));
        const uint32_t w_d0 = uni((uint32_t)__shfl((int)pick.d0, win)), w_next = uni((uint32_t)__shfl((int)pick.next, win));
        // the parked element ri sits at logical end + ri of the source list: an ordinary list change (pre-removal destination)
        const uint32_t src_pos = uni(off[ent + 1]) - uni(off[ent]) - n_rem + ri;
        if (fast && n_rem > 1) fetch_row(uni((uint32_t)rem[ri == 0 ? 1 : 0]));  // first row of the next round, in flight during the placement
        RPH(3)
        ruin_list_change(lm, visits, off, load, ent, src_pos, be, bp);
        RPH(4)
        if (has_cap) parked_dem = wsub(parked_dem, (int64_t)lm.demand[x]);
        const uint32_t moved = (lane >= ri && lane + 1 < n_rem) ? (uint32_t)rem[lane + 1] : 0u;
        wave_sync();
        if (lane >= ri && lane + 1 < n_rem) rem[lane] = (uint16_t)moved;
        if (lane == 0) {
            place[n_pl * 4] = (uint16_t)be;
            place[n_pl * 4 + 1] = (uint16_t)bp;
            place[n_pl * 4 + 2] = (uint16_t)(w_next == 0xFFFFFFFFu ? 0xFFFFu : w_next);
            place[n_pl * 4 + 3] = (uint16_t)w_d0;
            if (fast) {  // the two legs the placement created (the third, prev -> next, is what w_d0 remembers)
                rf.edge[x] = (uint16_t)w_da;
                if (w_next == 0xFFFFFFFFu)
                    rf.edge_end[be] = (uint16_t)w_db;
                else
                    rf.edge[w_next] = (uint16_t)w_db;
            }
        }
        wave_sync();
        n_pl += 1;
        n_rem -= 1;
#pragma unroll
        for (int k = 0; k < L; ++k) s.v[k] = (int64_t)uni64((uint64_t)M.v[k]);
        RPH(3)
    }
    if (rolled_back) s = cur;  // the reference puts everything back and the move scores like the untouched solution
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) out_score[k] = s.v[k];
    }
    if (!commit || rolled_back) {
        // undo: every placement back to the parked tail (reverse order: each recorded position is valid again), then the
        // source list back into its original order
        for (uint32_t i = n_pl; i-- > 0;) {
            const uint32_t be = uni((uint32_t)place[i * 4]), bp = uni((uint32_t)place[i * 4 + 1]);
            const uint32_t elen = uni(off[ent + 1]) - uni(off[ent]);
            // destination = physical end of the source list (pre-removal coordinates: intra moves name the slot after the last element)
            ruin_list_change(lm, visits, off, load, be, bp, ent, elen);
            if (fast && lane == 0) {
                const uint32_t nx = place[i * 4 + 2];
                if (nx == 0xFFFFu)
                    rf.edge_end[be] = place[i * 4 + 3];
                else
                    rf.edge[nx] = place[i * 4 + 3];
            }
        }
        const uint32_t oe2 = uni(off[ent]);
        for (uint32_t c0 = 0; c0 < plen; c0 += 64) {  // descending chunks: position q reads the kept element at q - #removed < q
            const uint32_t q = plen - 1u - (c0 + lane);
            const bool in = c0 + lane < plen;
            uint32_t nv = 0;
            if (in) {
                uint32_t before = 0, hit = 0xFFFFFFFFu;
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t pj = cd[2 + j];
                    before += pj < q ? 1u : 0u;
                    if (pj == q) hit = j;
                }
                nv = hit != 0xFFFFFFFFu ? (uint32_t)rem0[hit] : (uint32_t)visits[oe2 + q - before];
            }
            wave_sync();
            if (in) visits[oe2 + q] = (uint16_t)nv;
            wave_sync();
        }
        if (has_cap && lane == 0) load[ent] = load_before;
        wave_sync();
        if (fast && has_dist) (void)ruin_rebuild_list_edges(lm, visits, oe2, plen, ent, rf);
    }
    RPH(5)
    RPH_DUMP
}

// the kernel's entry: plain pointers into the replica's LDS slice -> typed pointers
template <int L>
__device__ __forceinline__ void ruin_recreate(const ListModel& lm, uint16_t* visits, uint32_t* off, int64_t* load, const uint16_t* cd, uint16_t* work,
                                              uint32_t* sbase, const RuinFast& rf, int skip_empty, bool commit, const int64_t* cur, int64_t* out_score) {
    ScoreV<L> c;
#pragma unroll
    for (int k = 0; k < L; ++k) c.v[k] = cur[k];
    ruin_recreate_lds<L>(ruin_model(lm), (lds_u16*)visits, (lds_u32*)off, (lds_i64*)load, (const lds_u16*)cd, (lds_u16*)work, (lds_u32*)sbase,
                         RuinFastLds{(lds_u16*)rf.edge, (lds_u16*)rf.edge_end, (lds_u16*)rf.row, (lds_u16*)rf.slot}, rf.edge != nullptr ? (lm.small32 ? 2 : 1) : 0, skip_empty, commit, c,
                         (lds_i64*)out_score);  // `fast` travels as a flag: the LDS null pointer is not the generic one
}

}  // namespace sf
