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
        gs[3] += placed;
        gs[4] += placed;
        gs[5] += trials;
        gs[7] += trials;
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
