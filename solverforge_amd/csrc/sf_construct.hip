// List cheapest-insertion construction on the device.
//
// Reference semantics restated (paths under crates/solverforge-solver/src/manager/phase_factory/list_construction/):
//   cheapest/kernel.rs:57-150   run_cheapest: the unassigned elements in (construction order key, source index) order; for every
//                               element every (list, position) is trial-inserted and fully scored, the strictly best score wins
//                               (the first of equals stays), the insertion is committed
//   cheapest/live.rs:64-170     one score_calculation per trial; one accepted + applied step per committed element
// (unrestricted owners, no order key, no precedence hooks: the list models this build runs).
//
// One wavefront = one replica, its lists in LDS.  An element's round is one pass of the ruin leaf's recreate scan with a single
// remaining element (sf_ruin.h: slot prefix, fixed-depth slot -> list search, U slots per lane in flight), a wave max + min
// picks the placement, the flat CSR shifts right by one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_ruin.h"

namespace sf {

struct ConstructCarve {
    size_t visits, off, load, sbase, rem, present, total;
    __host__ __device__ ConstructCarve(int V, int n_cap, int dim) {
        size_t o = 0;
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        sbase = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        rem = o;
        o = align_up(o + 16, 16);
        present = o;
        o = align_up(o + sizeof(uint32_t) * (((size_t)dim + 31) / 32), 16);
        total = o;
    }
};

// inserts x at (e, p): the flat CSR shifts right by one from the slot on (descending chunks: reads stay ahead of writes)
__device__ __forceinline__ void construct_list_insert(const RuinModel& m, lds_u16* visits, lds_u32* off, lds_i64* load, uint32_t e, uint32_t p, uint32_t x) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tot = uni(off[m.V]), Q = uni(off[e]) + p;
    for (uint32_t c0 = 0; c0 < tot - Q; c0 += 64) {
        const uint32_t dd = c0 + lane;
        const bool in = dd < tot - Q;
        const uint32_t t = tot - (in ? dd : 0u);
        const uint32_t nv = in ? (uint32_t)visits[t - 1] : 0u;
        wave_sync();
        if (in) visits[t] = (uint16_t)nv;
        wave_sync();
    }
    if (lane == 0) visits[Q] = (uint16_t)x;
    for (uint32_t rr = lane; rr <= (uint32_t)m.V; rr += 64)
        if (rr > e) off[rr] += 1;
    if (lane == 0 && m.demand) load[e] = wadd(load[e], (int64_t)m.demand[x]);
    wave_sync();
}

template <int L>
__global__ __launch_bounds__(64) void k_list_construct_cheapest(ListModel lm, const uint32_t* __restrict__ elements, int n_el, uint64_t* stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int r = blockIdx.x;
    const int V = lm.V;
    const ConstructCarve cv(V, lm.n_cap, lm.dim);
    lds_u16* visits = (lds_u16*)(smem + cv.visits);
    lds_u32* off = (lds_u32*)(smem + cv.off);
    lds_i64* load = (lds_i64*)(smem + cv.load);
    lds_u32* sbase = (lds_u32*)(smem + cv.sbase);
    lds_u16* rem = (lds_u16*)(smem + cv.rem);
    lds_u32* present = (lds_u32*)(smem + cv.present);
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    int64_t* g_load = lm.load + (size_t)r * V;
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) off[t] = g_off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) load[t] = g_load[t];
    for (uint32_t t = lane; t < ((uint32_t)lm.dim + 31u) / 32u; t += 64) present[t] = 0u;
    wave_sync();
    const uint32_t tot0 = uni(off[V]);
    for (uint32_t t = lane; t < tot0; t += 64) {
        const uint32_t x = g_visits[t];
        visits[t] = (uint16_t)x;
        atomicOr((uint32_t*)&present[x >> 5], 1u << (x & 31u));
    }
    wave_sync();
    RuinModel rm = ruin_model(lm);
    const bool has_dist = lm.dist_level >= 0;
    ScoreV<L> s;
#pragma unroll
    for (int k = 0; k < L; ++k) s.v[k] = lm.score[(size_t)r * 4 + k];
    uint64_t trials = 0, placed = 0;
    for (int k = 0; k < n_el; ++k) {
        const uint32_t x = elements[k];
        if (x >= (uint32_t)lm.dim || ((uni(present[x >> 5]) >> (x & 31u)) & 1u)) continue;  // already in a list
        if (uni(off[V]) >= (uint32_t)lm.n_cap) break;                                        // element capacity reached
        ruin_slot_prefix(rm, off, sbase, 0xFFFFFFFFu, 0u, 0);
        if (lane == 0) rem[0] = (uint16_t)x;
        wave_sync();
        ScoreV<L> bs;
#pragma unroll
        for (int q = 0; q < L; ++q) bs.v[q] = INT64_MIN;
        uint64_t bkey = ~0ull;
        bool has = false;
        if (lm.mat32 || !has_dist)
            ruin_scan_round<L, 1, true>(rm, visits, off, load, rem, sbase, 1u, 0xFFFFFFFFu, 0, s, bs, bkey, has);
        else
            ruin_scan_round<L, 1, false>(rm, visits, off, load, rem, sbase, 1u, 0xFFFFFFFFu, 0, s, bs, bkey, has);
        trials += uni(sbase[V]);
        if (__ballot(has) == 0ull) continue;  // no list at all
        const ScoreV<L> M = wave_max_score<L>(bs, has);
        const uint64_t kmin = uni64(ruin_wave_min_u64((has && score_cmp<L>(bs, M) == 0) ? bkey : ~0ull));
        construct_list_insert(rm, visits, off, load, (uint32_t)(kmin >> 16) & 0xFFFFu, (uint32_t)kmin & 0xFFFFu, x);
        if (lane == 0) present[x >> 5] |= 1u << (x & 31u);
        wave_sync();
#pragma unroll
        for (int q = 0; q < L; ++q) s.v[q] = (int64_t)uni64((uint64_t)M.v[q]);
        placed += 1;
    }
    const uint32_t tot = uni(off[V]);
    for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = visits[t];
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = load[t];
    if (stats && lane == 0) {  // live.rs: one score calculation per trial, one accepted + applied step per placed element
        uint64_t* gs = stats + (size_t)r * SF_STATS_WORDS;
        gs[0] += placed;
        gs[1] += trials;  // record_construction_candidate (live.rs:123-127): a generated + evaluated candidate per trial
        gs[2] += trials;
        gs[3] += placed;
        gs[4] += placed;
        gs[5] += trials;
        gs[7] += trials;
    }
}

// Regret-insertion list construction (manager/phase_factory/list_construction/regret/kernel/execute.rs:52-204, the element's
// best / second-best trial kernel/evaluation.rs:120-230, the ordering of choices kernel/mod.rs:19-75; unrestricted owners, no order
// key, no precedence hooks).  One wavefront = one replica, lists in LDS like the cheapest-insertion kernel above.  A round prices
// every slot of every still-unassigned element: the slots go over the lanes, NR elements share one pass (one binary search, one
// pair of neighbours and one removed leg per slot; the 2 * NR legs of a slot's elements are gathers in flight together), every lane
// keeps each element's best (first of equals) and second-best score of ITS slots; per element a wave max + min finds the best slot
// and the second best is the max over the other lanes' bests and the winning lane's runner-up.  The choice among the elements
// (greatest regret, Forced above Finite; then the better score; then the earlier element) is wave-uniform scalar work.  An element
// with a fixed owner (the owner hook, list_placement.rs:54-69) takes part in the same pass with the other lists' slots masked.
struct RegretCarve {
    size_t visits, off, load, sbase, present, un, own, total;
    __host__ __device__ RegretCarve(int V, int n_cap, int dim, int n_el) {
        size_t o = 0;
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        sbase = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        present = o;
        o = align_up(o + sizeof(uint32_t) * (((size_t)dim + 31) / 32), 16);
        un = o;
        o = align_up(o + sizeof(uint16_t) * (n_el > 0 ? n_el : 1), 16);
        own = o;
        o = align_up(o + sizeof(int16_t) * (n_el > 0 ? n_el : 1), 16);
        total = o;
    }
};

template <int L, int NR, bool M32>
__device__ __forceinline__ void regret_scan(const RuinModel& lm, const lds_u16* visits, const lds_u32* off, const lds_i64* load, const lds_u32* sbase,
                                            const uint32_t (&x)[NR], const int32_t (&ow)[NR], uint32_t n_x, const ScoreV<L>& s, ScoreV<L> (&bs)[NR],
                                            uint32_t (&bkey)[NR], bool (&has)[NR], ScoreV<L> (&s2)[NR], bool (&has2)[NR]) {
    constexpr int U = 2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot;
    const bool has_dist = lm.dist_level >= 0, has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    const uint32_t total = uni(sbase[V]);
    uint32_t top = 1;
    while (top < V) top <<= 1;
    int64_t dx[NR];
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) dx[ri] = has_cap ? (int64_t)lm.demand[x[ri]] : 0;
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
            ld[u] = has_cap ? load[lo] : 0;
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
        // a lane meets its slots in (list, position) order: a tie never replaces its running best.  The updates are selects, not
        // branches: the branchy form (if (better) { bs = sc; bkey = key; }) lost the key update of a lane's second slot on gfx950
        // (ROCm 7.2 hipcc -O3; the scores moved, the key did not) -- caught by the oracle on the last element of a 60-customer run
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int ri = 0; ri < NR; ++ri) {
                ListDelta d{0, 0, true};
                if (has_dist) d.d_dist = wsub(wadd(da[u][ri], db[u][ri]), empty[u] ? 0 : d0[u]);
                if (has_cap) d.d_cap = wsub(over_cap(wadd(ld[u], dx[ri]), lm.capacity), over_cap(ld[u], lm.capacity));
                const ScoreV<L> sc = ruin_apply_delta<L>(lm, s, d);
                const uint32_t key = (e[u] << 16) | o[u];
                const bool live = valid[u] && (uint32_t)ri < n_x && (ow[ri] < 0 || e[u] == (uint32_t)ow[ri]);  // candidate_entities (mod.rs:104-114)
                const bool had = has[ri];
                const bool take = live && (!had || score_cmp<L>(sc, bs[ri]) > 0);
                ScoreV<L> runner;  // what this slot leaves for the second place: the displaced best, or the slot itself
#pragma unroll
                for (int q = 0; q < L; ++q) {
                    runner.v[q] = take ? bs[ri].v[q] : sc.v[q];
                    bs[ri].v[q] = take ? sc.v[q] : bs[ri].v[q];
                }
                bkey[ri] = take ? key : bkey[ri];
                has[ri] = had || take;
                const bool take2 = live && had && (!has2[ri] || score_cmp<L>(runner, s2[ri]) > 0);
#pragma unroll
                for (int q = 0; q < L; ++q) s2[ri].v[q] = take2 ? runner.v[q] : s2[ri].v[q];
                has2[ri] = has2[ri] || take2;
            }
        }
    }
}

template <int L>
__global__ __launch_bounds__(64) void k_list_construct_regret(ListModel lm, const uint32_t* __restrict__ elements, const int32_t* __restrict__ owners,
                                                              int n_el, uint64_t* stats) {
    constexpr int NR = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int r = blockIdx.x;
    const int V = lm.V;
    const RegretCarve cv(V, lm.n_cap, lm.dim, n_el);
    lds_u16* visits = (lds_u16*)(smem + cv.visits);
    lds_u32* off = (lds_u32*)(smem + cv.off);
    lds_i64* load = (lds_i64*)(smem + cv.load);
    lds_u32* sbase = (lds_u32*)(smem + cv.sbase);
    lds_u16* un = (lds_u16*)(smem + cv.un);
    __attribute__((address_space(3))) int16_t* own = (__attribute__((address_space(3))) int16_t*)(smem + cv.own);  // owner hook per unassigned element, -1 = unrestricted
    lds_u32* present = (lds_u32*)(smem + cv.present);
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    int64_t* g_load = lm.load + (size_t)r * V;
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) off[t] = g_off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) load[t] = g_load[t];
    for (uint32_t t = lane; t < ((uint32_t)lm.dim + 31u) / 32u; t += 64) present[t] = 0u;
    wave_sync();
    const uint32_t tot0 = uni(off[V]);
    for (uint32_t t = lane; t < tot0; t += 64) {
        const uint32_t x = g_visits[t];
        visits[t] = (uint16_t)x;
        atomicOr((uint32_t*)&present[x >> 5], 1u << (x & 31u));
    }
    wave_sync();
    // the unassigned elements of this replica, source order kept (the host refuses a repeated id)
    uint32_t n_un = 0;
    for (int k0 = 0; k0 < n_el; k0 += 64) {
        const int k = k0 + (int)lane;
        const uint32_t x = k < n_el ? elements[k] : 0xFFFFFFFFu;
        const bool take = k < n_el && x < (uint32_t)lm.dim && !((present[x >> 5] >> (x & 31u)) & 1u);
        const uint64_t mask = __ballot(take);
        if (take) {
            const uint32_t at = n_un + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            un[at] = (uint16_t)x;
            own[at] = (int16_t)(owners ? owners[k] : -1);
        }
        n_un += (uint32_t)__popcll(mask);
    }
    wave_sync();
    RuinModel rm = ruin_model(lm);
    const bool has_dist = lm.dist_level >= 0;
    const bool m32 = lm.mat32 || !has_dist;
    ScoreV<L> s;
#pragma unroll
    for (int k = 0; k < L; ++k) s.v[k] = lm.score[(size_t)r * 4 + k];
    uint64_t trials = 0, placed = 0;
    while (n_un > 0 && V > 0) {
        if (uni(off[V]) >= (uint32_t)lm.n_cap) break;  // element capacity reached
        ruin_slot_prefix(rm, off, sbase, 0xFFFFFFFFu, 0u, 0);
        const uint32_t total = uni(sbase[V]);
        bool c_have = false, c_forced = false;
        ScoreV<L> c_regret, c_score;
        uint32_t c_li = 0, c_key = 0;
#pragma unroll
        for (int q = 0; q < L; ++q) c_regret.v[q] = 0, c_score.v[q] = 0;
        for (uint32_t li0 = 0; li0 < n_un; li0 += NR) {
            const uint32_t n_x = n_un - li0 < (uint32_t)NR ? n_un - li0 : (uint32_t)NR;
            uint32_t x[NR];
            int32_t ow[NR];
            ScoreV<L> bs[NR], s2[NR];
            uint32_t bkey[NR];
            bool has[NR], has2[NR];
#pragma unroll
            for (int ri = 0; ri < NR; ++ri) {
                x[ri] = uni((uint32_t)un[li0 + ((uint32_t)ri < n_x ? (uint32_t)ri : 0u)]);
                ow[ri] = (int32_t)uni((uint32_t)(int32_t)own[li0 + ((uint32_t)ri < n_x ? (uint32_t)ri : 0u)]);
                if ((uint32_t)ri < n_x) trials += ow[ri] < 0 ? (uint64_t)total : (uint64_t)(uni(sbase[ow[ri] + 1]) - uni(sbase[ow[ri]]));
                has[ri] = has2[ri] = false;
                bkey[ri] = 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < L; ++q) bs[ri].v[q] = s2[ri].v[q] = INT64_MIN;
            }
            if (m32)
                regret_scan<L, NR, true>(rm, visits, off, load, sbase, x, ow, n_x, s, bs, bkey, has, s2, has2);
            else
                regret_scan<L, NR, false>(rm, visits, off, load, sbase, x, ow, n_x, s, bs, bkey, has, s2, has2);
#pragma unroll
            for (int ri = 0; ri < NR; ++ri) {
                if ((uint32_t)ri >= n_x || __ballot(has[ri]) == 0ull) continue;
                const ScoreV<L> M = wave_max_score<L>(bs[ri], has[ri]);
                const bool at_max = has[ri] && score_cmp<L>(bs[ri], M) == 0;
                const uint32_t kmin = uni((uint32_t)ruin_wave_min_u64(at_max ? (uint64_t)bkey[ri] : ~0ull));
                const bool winner = at_max && bkey[ri] == kmin;
                const bool other = winner ? has2[ri] : has[ri];
                const bool forced = __ballot(other) == 0ull;  // RegretValue::Forced: the element has one slot only
                const ScoreV<L> S2 = wave_max_score<L>(winner ? s2[ri] : bs[ri], other);
                ScoreV<L> best, regret;
#pragma unroll
                for (int q = 0; q < L; ++q) {
                    best.v[q] = (int64_t)uni64((uint64_t)M.v[q]);
                    regret.v[q] = forced ? 0 : wsub(best.v[q], (int64_t)uni64((uint64_t)S2.v[q]));
                }
                bool better = !c_have;
                if (c_have) {
                    const int rc = forced != c_forced ? (forced ? 1 : -1) : (forced ? 0 : score_cmp<L>(regret, c_regret));
                    better = rc > 0 || (rc == 0 && score_cmp<L>(best, c_score) > 0);
                }
                if (better) {
                    c_have = true, c_forced = forced;
                    c_regret = regret, c_score = best;
                    c_li = li0 + (uint32_t)ri, c_key = kmin;
                }
            }
        }
        if (!c_have) break;
        const uint32_t x = uni((uint32_t)un[c_li]);
        construct_list_insert(rm, visits, off, load, c_key >> 16, c_key & 0xFFFFu, x);
        for (uint32_t t0 = c_li; t0 + 1 < n_un; t0 += 64) {  // unassigned.remove(list_index): the tail moves up by one (ascending chunks)
            const uint32_t t = t0 + lane;
            const uint32_t nv = t + 1 < n_un ? (uint32_t)un[t + 1] : 0u;
            const int16_t no = t + 1 < n_un ? own[t + 1] : (int16_t)-1;
            wave_sync();
            if (t + 1 < n_un) un[t] = (uint16_t)nv, own[t] = no;
            wave_sync();
        }
        n_un -= 1;
        s = c_score;
        placed += 1;
    }
    const uint32_t tot = uni(off[V]);
    for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = visits[t];
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = load[t];
    if (stats && lane == 0) {  // evaluation.rs:60-66: one generated + evaluated candidate and one score calculation per trial
        uint64_t* gs = stats + (size_t)r * SF_STATS_WORDS;
        gs[0] += placed;
        gs[1] += trials;
        gs[2] += trials;
        gs[3] += placed;
        gs[4] += placed;
        gs[5] += trials;
        gs[7] += trials;
    }
}

// Round-robin list construction (manager/phase_factory/list_construction/round_robin/kernel.rs:71-175).  The host hands over the
// declared elements sorted by (construction order key, source index) with the owner hook's value per element (-1 = unrestricted;
// elements whose hook names no valid owner are dropped on the host: they are skipped in every replica).  One wavefront per
// replica: an element already in one of the replica's lists is not a candidate; the k-th unrestricted candidate goes to owner
// k mod V (the cursor advances only for unrestricted elements), a fixed-owner candidate to its owner; every candidate is appended
// behind what its list holds so far.  64 elements per round: ranks by ballot prefix, the slot inside a list by counting the
// earlier lanes with the same target.  Counters: one generated + evaluated candidate, one accepted + applied step, one score
// calculation per appended element.
struct RoundRobinCarve {
    size_t load, off, cnt, visits, tgt, pos, present, total;
    __host__ __device__ RoundRobinCarve(int V, int n_cap, int dim, int n) {
        size_t o = 0;
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        cnt = o;
        o = align_up(o + sizeof(uint32_t) * V, 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        tgt = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        pos = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        present = o;
        o = align_up(o + sizeof(uint32_t) * (((size_t)dim + 31) / 32), 16);
        total = o;
    }
};

__global__ __launch_bounds__(64) void k_list_construct_round_robin(ListModel lm, const uint32_t* __restrict__ elements, const int32_t* __restrict__ owners,
                                                                   int n_el, uint64_t* stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int r = blockIdx.x;
    const uint32_t V = (uint32_t)lm.V;
    const RoundRobinCarve cv(lm.V, lm.n_cap, lm.dim, n_el);
    lds_i64* load = (lds_i64*)(smem + cv.load);
    lds_u32* off = (lds_u32*)(smem + cv.off);
    lds_u32* cnt = (lds_u32*)(smem + cv.cnt);
    lds_u16* visits = (lds_u16*)(smem + cv.visits);
    lds_u16* tgt = (lds_u16*)(smem + cv.tgt);
    lds_u16* pos = (lds_u16*)(smem + cv.pos);
    lds_u32* present = (lds_u32*)(smem + cv.present);
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    int64_t* g_load = lm.load + (size_t)r * V;
    for (uint32_t t = lane; t < V; t += 64) cnt[t] = g_off[t + 1] - g_off[t];
    for (uint32_t t = lane; t < ((uint32_t)lm.dim + 31u) / 32u; t += 64) present[t] = 0u;
    wave_sync();
    const uint32_t tot0 = g_off[V];
    for (uint32_t t = lane; t < tot0; t += 64) {
        const uint32_t x = g_visits[t];
        atomicOr((uint32_t*)&present[x >> 5], 1u << (x & 31u));
    }
    wave_sync();
    const uint32_t room = (uint32_t)lm.n_cap - tot0;
    uint32_t placed = 0, cursor = 0;  // candidates so far, unrestricted candidates so far
    for (uint32_t k0 = 0; k0 < (uint32_t)n_el; k0 += 64) {
        const uint32_t k = k0 + lane;
        bool act = false, unres = false;
        int32_t ow = -1;
        if (k < (uint32_t)n_el) {
            const uint32_t x = elements[k];
            ow = owners ? owners[k] : -1;
            act = !((present[x >> 5] >> (x & 31u)) & 1u);
        }
        const uint64_t am = __ballot(act);
        const uint32_t arank = placed + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        act = act && arank < room;  // element capacity of the flat lists: construction stops when it is reached
        const uint64_t am2 = __ballot(act);
        unres = act && ow < 0;
        const uint64_t um = __ballot(unres);
        uint32_t target = 0xFFFFFFFFu;
        if (act) target = unres ? (cursor + (uint32_t)__popcll(um & ((1ull << lane) - 1ull))) % V : (uint32_t)ow;
        uint32_t intra = 0;
        for (uint32_t j = 0; j < 64; ++j) {
            const uint32_t tj = (uint32_t)__shfl((int)target, (int)j, 64);
            if (j < lane && tj == target) ++intra;
        }
        uint32_t at = 0;
        if (act) at = cnt[target] + intra;
        wave_sync();
        if (act) {
            atomicAdd((uint32_t*)&cnt[target], 1u);
            tgt[k] = (uint16_t)target;
            pos[k] = (uint16_t)at;
        } else if (k < (uint32_t)n_el) {
            tgt[k] = 0xFFFFu;
        }
        wave_sync();
        placed += (uint32_t)__popcll(am2);
        cursor += (uint32_t)__popcll(um);
    }
    // new offsets, old contents, appended elements, loads
    uint32_t acc = 0;
    for (uint32_t e0 = 0; e0 < V; e0 += 64) {
        const uint32_t e = e0 + lane;
        const uint32_t len = e < V ? (uint32_t)cnt[e] : 0u;
        uint32_t inc = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= (uint32_t)d) inc += o;
        }
        if (e < V) off[e] = acc + inc - len;
        acc += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (lane == 0) off[V] = acc;
    wave_sync();
    for (uint32_t e = 0; e < V; ++e) {
        const uint32_t go = g_off[e], len = g_off[e + 1] - go, o = off[e];
        for (uint32_t t = lane; t < len; t += 64) visits[o + t] = (uint16_t)g_visits[go + t];
    }
    for (uint32_t k = lane; k < (uint32_t)n_el; k += 64)
        if (tgt[k] != 0xFFFFu) visits[off[tgt[k]] + pos[k]] = (uint16_t)elements[k];
    wave_sync();
    for (uint32_t e = lane; e < V; e += 64) {
        int64_t l = 0;
        if (lm.demand)
            for (uint32_t t = off[e]; t < off[e + 1]; ++t) l = wadd(l, (int64_t)lm.demand[visits[t]]);
        load[e] = l;
    }
    wave_sync();
    for (uint32_t t = lane; t < acc; t += 64) g_visits[t] = visits[t];
    for (uint32_t t = lane; t <= V; t += 64) g_off[t] = off[t];
    for (uint32_t t = lane; t < V; t += 64) g_load[t] = load[t];
    if (stats && lane == 0) {
        uint64_t* gs = stats + (size_t)r * SF_STATS_WORDS;
        gs[0] += placed, gs[1] += placed, gs[2] += placed, gs[3] += placed, gs[4] += placed, gs[5] += placed, gs[7] += placed;
    }
}

// Host-provided list ruin moves (SF_MOVE_LIST_RUIN through sf_step_evaluate / sf_apply): one wavefront per move, the replica's
// lists copied into LDS, the same recreate the fused step runs (general matrix-gather path).  move t of the batch = moves[idx[t]].
// doable = ruin_is_doable without an owner binding (move/list_kernel/ruin.rs:97-113) with the wire format's ascending, distinct
// positions.  commit: the recreate is kept, lists + load + committed score written back (grid = 1).
struct RuinMoveCarve {
    size_t visits, off, load, sbase, cand, work, score, total;
    __host__ __device__ RuinMoveCarve(int V, int n_cap) {
        size_t o = 0;
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        score = o;
        o = align_up(o + sizeof(int64_t) * 4, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        sbase = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        cand = o;
        o = align_up(o + 16, 16);
        work = o;
        o = align_up(o + 128, 16);
        total = o;
    }
};
template <int L>
__global__ __launch_bounds__(64) void k_list_ruin_moves(ListModel lm, int replica, const int32_t* __restrict__ moves, const int32_t* __restrict__ idx,
                                                        int64_t* out_scores, int32_t* out_doable, int skip_empty, int commit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int V = lm.V;
    const RuinMoveCarve cv(V, lm.n_cap);
    uint16_t* visits = (uint16_t*)(smem + cv.visits);
    uint32_t* off = (uint32_t*)(smem + cv.off);
    int64_t* load = (int64_t*)(smem + cv.load);
    uint32_t* sbase = (uint32_t*)(smem + cv.sbase);
    uint16_t* cand = (uint16_t*)(smem + cv.cand);
    uint16_t* work = (uint16_t*)(smem + cv.work);
    int64_t* sc = (int64_t*)(smem + cv.score);
    uint32_t* g_visits = lm.visits + (size_t)replica * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)replica * (V + 1);
    int64_t* g_load = lm.load + (size_t)replica * V;
    const int64_t t = idx[blockIdx.x];
    const int32_t* mv = moves + t * 6;
    for (uint32_t q = lane; q <= (uint32_t)V; q += 64) off[q] = g_off[q];
    for (uint32_t q = lane; q < (uint32_t)V; q += 64) load[q] = g_load[q];
    wave_sync();
    const uint32_t tot = uni(off[V]);
    for (uint32_t q = lane; q < tot; q += 64) visits[q] = (uint16_t)g_visits[q];
    const int32_t a = mv[1], cnt = mv[2];
    bool ok = a >= 0 && a < V && cnt >= 1 && cnt <= (int32_t)RUIN_MAX_COUNT;
    const uint32_t w3[3] = {(uint32_t)mv[3], (uint32_t)mv[4], (uint32_t)mv[5]};
    uint32_t pos[RUIN_MAX_COUNT];
#pragma unroll
    for (int i = 0; i < (int)RUIN_MAX_COUNT; ++i) pos[i] = (w3[i / 2] >> (16 * (i & 1))) & 0xFFFFu;
    if (ok) {
        const uint32_t len = uni(off[a + 1]) - uni(off[a]);
#pragma unroll
        for (int i = 0; i < (int)RUIN_MAX_COUNT; ++i)
            if (i < cnt) ok = ok && pos[i] < len && (i == 0 || pos[i] > pos[i - 1]);
    }
    if (lane == 0) {
        cand[0] = (uint16_t)a;
        cand[1] = (uint16_t)cnt;
#pragma unroll
        for (int i = 0; i < (int)RUIN_MAX_COUNT; ++i) cand[2 + i] = (uint16_t)pos[i];
    }
    wave_sync();
    if (!ok) {
        if (lane == 0) {
            out_doable[t] = 0;
            for (int k = 0; k < lm.levels; ++k) out_scores[t * lm.levels + k] = 0;
        }
        return;
    }
    int64_t cur[L];
#pragma unroll
    for (int k = 0; k < L; ++k) cur[k] = lm.score[(size_t)replica * 4 + k];
    ruin_recreate<L>(lm, visits, off, load, cand, work, sbase, RuinFast{nullptr, nullptr, nullptr, nullptr}, skip_empty, commit != 0, cur, sc);
    wave_sync();
    if (lane == 0) {
        out_doable[t] = 1;
        for (int k = 0; k < lm.levels; ++k) out_scores[t * lm.levels + k] = k < L ? sc[k] : 0;
    }
    if (commit) {
        const uint32_t tot2 = uni(off[V]);
        for (uint32_t q = lane; q < tot2; q += 64) g_visits[q] = visits[q];
        for (uint32_t q = lane; q <= (uint32_t)V; q += 64) g_off[q] = off[q];
        for (uint32_t q = lane; q < (uint32_t)V; q += 64) g_load[q] = load[q];
        if (lane == 0)
            for (int k = 0; k < L; ++k) lm.score[(size_t)replica * 4 + k] = sc[k];
    }
}

}  // namespace sf
