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

// Per-replica LDS block of the leaf.
struct RuinLds {
    static constexpr size_t CAND_WORDS = 8;  // u16: list, count, positions[6]
    static constexpr size_t bytes = 8 * 8 + RUIN_MAX_MOVES * (CAND_WORDS * 2 + 4 * 8) + 64;
    static_assert(bytes == RUIN_LDS_BYTES, "GCarve reserves RUIN_LDS_BYTES");
    uint64_t* prng;   // [4] per-solve stream (loaded at launch start, stored at launch end)
    uint64_t* crng;   // [4] cursor stream of this step
    int64_t* score;   // [RUIN_MAX_MOVES][4] trial score of every generated candidate
    uint16_t* cand;   // [RUIN_MAX_MOVES][CAND_WORDS]
    uint16_t* work;   // [32]: removed nodes [8], placements (list, position) [8][2], remaining nodes [8]
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

__device__ __forceinline__ int64_t ruin_leg(const ListModel& m, uint32_t from, uint32_t to) {
    if (m.mat32) {
        const uint32_t v = m.mat32[from * (uint32_t)m.dim + to];
        return v != 0xFFFFFFFFu ? (int64_t)v : MAX_SAFE_LEG_COST;
    }
    return dist_cost(m.mat, m.dim, from, to);
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
__device__ __forceinline__ int64_t ruin_route_distance(const ListModel& m, const uint16_t* visits, uint32_t o, uint32_t len, uint32_t lane) {
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

#ifndef SF_RUIN_INLINE
#define SF_RUIN_ATTR __attribute__((noinline))
#else
#define SF_RUIN_ATTR __forceinline__
#endif

// ruin_do_move on the replica's LDS state for the candidate `cd` = (list, count, ascending positions).  Writes the score
// after the recreate to out_score[0..L); when `commit` is false the state is restored before returning.
template <int L>
__device__ SF_RUIN_ATTR void ruin_recreate(const ListModel& lm, uint16_t* visits, uint32_t* off, int64_t* load, const uint16_t* cd, uint16_t* work,
                                           int skip_empty, bool commit, const int64_t* cur, int64_t* out_score) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    const uint32_t ent = uni((uint32_t)cd[0]), cnt = uni((uint32_t)cd[1]);
    const bool has_dist = lm.dist_level >= 0, has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    uint16_t* rem0 = work;          // [8] removed nodes in removal (= ascending position) order
    uint16_t* place = work + 8;     // [8][2] (list, position) of every placement, in placement order
    uint16_t* rem = work + 24;      // [8] remaining nodes, order preserved
    ScoreV<L> s;
#pragma unroll
    for (int k = 0; k < L; ++k) s.v[k] = cur[k];
    const uint32_t oe = uni(off[ent]), plen = uni(off[ent + 1]) - oe;  // physical length of the source list (never changes below
                                                                      // until an element leaves for another list)
    // ---- remove: stable partition of the source list into [kept .. | removed ..] ----
    const int64_t dist_before = has_dist ? ruin_route_distance(lm, visits, oe, plen, lane) : 0;
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
        if (has_dist) d.d_dist = wsub(ruin_route_distance(lm, visits, oe, plen - cnt, lane), dist_before);
        if (has_cap) d.d_cap = wsub(over_cap(wsub(load_before, parked_dem), lm.capacity), over_cap(load_before, lm.capacity));
        s = apply_delta<L>(lm, s.v, d);
    }
    // ---- recreate: one wave-wide round per remaining element ----
    uint32_t n_rem = cnt, n_pl = 0;
    bool rolled_back = false;
    while (n_rem > 0) {
        ScoreV<L> bs;
#pragma unroll
        for (int k = 0; k < L; ++k) bs.v[k] = INT64_MIN;
        uint64_t bkey = ~0ull;
        bool has = false;
        uint32_t rd = 0, so = 0;
        while (rd < V) {
            const uint32_t rk = rd + lane;
            uint32_t cnt_k = 0;
            if (rk < V) {
                const uint32_t l = off[rk + 1] - off[rk] - (rk == ent ? n_rem : 0u);
                cnt_k = (skip_empty && l == 0) ? 0u : l + 1u;
            }
            const uint32_t full_k = cnt_k;
            if (lane == 0) cnt_k = cnt_k > so ? cnt_k - so : 0u;
            uint32_t grp, o, total;
            map_slots_to_groups(cnt_k, lane, grp, o, total);
            const uint32_t slots = (uint32_t)__shfl((int)full_k, (int)grp);
            if (grp == 0) o += so;
            const uint32_t e = rd + grp;
            if (lane < total) {
                const uint32_t ob = off[e], le = slots - 1u;
                const uint32_t prev = o > 0 ? (uint32_t)visits[ob + o - 1] : depot;
                const uint32_t next = o < le ? (uint32_t)visits[ob + o] : depot;
                const int64_t d0 = (has_dist && le != 0) ? ruin_leg(lm, prev, next) : 0;
                const int64_t ld = has_cap ? (e == ent ? wsub(load[e], parked_dem) : load[e]) : 0;
                for (uint32_t ri = 0; ri < n_rem; ++ri) {
                    const uint32_t x = rem[ri];
                    ListDelta d{0, 0, true};
                    if (has_dist) d.d_dist = wsub(wadd(ruin_leg(lm, prev, x), ruin_leg(lm, x, next)), d0);
                    if (has_cap) d.d_cap = wsub(over_cap(wadd(ld, (int64_t)lm.demand[x]), lm.capacity), over_cap(ld, lm.capacity));
                    const ScoreV<L> sc = apply_delta<L>(lm, s.v, d);
                    const uint64_t key = ((uint64_t)ri << 32) | ((uint64_t)e << 16) | (uint64_t)o;
                    const int cmp = has ? score_cmp<L>(sc, bs) : 1;
                    if (cmp > 0 || (cmp == 0 && key < bkey)) {  // strictly better, or the earlier of equals (:228-237)
                        bs = sc;
                        bkey = key;
                        has = true;
                    }
                }
            }
            if (total <= 64) {
                rd += 64;
                so = 0;
            } else {  // resume after lane 63's slot
                const uint32_t lg = uni((uint32_t)__shfl((int)grp, 63)), lo_ = uni((uint32_t)__shfl((int)o, 63));
                const uint32_t ls = uni((uint32_t)__shfl((int)slots, 63));
                if (lo_ + 1 >= ls) {
                    rd += lg + 1;
                    so = 0;
                } else {
                    rd += lg;
                    so = lo_ + 1;
                }
            }
        }
        if (__ballot(has) == 0ull) {  // no destination at all: restore_removed_elements (:250-253)
            rolled_back = true;
            break;
        }
        const ScoreV<L> M = wave_max_score<L>(bs, has);
        const uint64_t kmin = uni64(ruin_wave_min_u64((has && score_cmp<L>(bs, M) == 0) ? bkey : ~0ull));
        const uint32_t ri = (uint32_t)(kmin >> 32), be = (uint32_t)(kmin >> 16) & 0xFFFFu, bp = (uint32_t)kmin & 0xFFFFu;
        const uint32_t x = uni((uint32_t)rem[ri]);
        // the parked element ri sits at logical end + ri of the source list: an ordinary list change (pre-removal destination)
        const uint32_t src_pos = uni(off[ent + 1]) - uni(off[ent]) - n_rem + ri;
        apply_list_move_wave(lm, visits, off, load, 2, ent, src_pos, be, bp);
        if (has_cap) parked_dem = wsub(parked_dem, (int64_t)lm.demand[x]);
        const uint32_t moved = (lane >= ri && lane + 1 < n_rem) ? (uint32_t)rem[lane + 1] : 0u;
        wave_sync();
        if (lane >= ri && lane + 1 < n_rem) rem[lane] = (uint16_t)moved;
        if (lane == 0) {
            place[n_pl * 2] = (uint16_t)be;
            place[n_pl * 2 + 1] = (uint16_t)bp;
        }
        wave_sync();
        n_pl += 1;
        n_rem -= 1;
#pragma unroll
        for (int k = 0; k < L; ++k) s.v[k] = (int64_t)uni64((uint64_t)M.v[k]);
    }
    if (rolled_back) {  // the reference puts everything back and the move scores like the untouched solution
#pragma unroll
        for (int k = 0; k < L; ++k) s.v[k] = cur[k];
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) out_score[k] = s.v[k];
    }
    if (!commit || rolled_back) {
        // undo: every placement back to the parked tail (reverse order: each recorded position is valid again), then the
        // source list back into its original order
        for (uint32_t i = n_pl; i-- > 0;) {
            const uint32_t be = uni((uint32_t)place[i * 2]), bp = uni((uint32_t)place[i * 2 + 1]);
            const uint32_t elen = uni(off[ent + 1]) - uni(off[ent]);
            // destination = physical end of the source list (pre-removal coordinates: intra moves name the slot after the last element)
            apply_list_move_wave(lm, visits, off, load, 2, be, bp, ent, elen);
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
    }
}

}  // namespace sf
