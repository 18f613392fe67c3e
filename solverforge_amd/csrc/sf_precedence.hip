// Director-surface kernels of the ListPrecedenceMakespanConstraint (sf_precedence.h): full evaluation of every replica
// (initialize / fresh_score / evaluate_each), the committed refresh after sf_apply, and n x evaluate_candidate for host-provided
// list moves (one wavefront per record: the replica's lists staged in LDS, the move applied there, one full evaluation).
#pragma once
#include "sf_precedence.h"
#include "sf_prec_leaf.h"

namespace sf {

// ≙ ListCheapestInsertionPhase on a list class scored by the precedence constraint (manager/phase_factory/list_construction/cheapest/
// kernel.rs:57-150; the host orders the elements, incl. precedence_downstream :162-229).  One wavefront per replica, lists staged in
// LDS.  Per element: one forward evaluation of the lists, one backward pass, two reachability searches and one sweep over every
// insertion slot (plf_best_slot, hooks = false: a slot that closes a cycle is priced as the constraint prices a cycle) instead of
// one full evaluation per slot; when the lists are already cyclic the element slides through every slot, one evaluation each.
// order: 0 hard penalty before makespan, 1 makespan first, 2 both on one level.
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_prec_construct_cheapest(ListModel lm, PrecModel pm, PlfModel pl, const uint32_t* __restrict__ elements, int n_el, int order,
                                                                uint64_t* stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_info[4];
    const uint32_t lane = threadIdx.x & 63u;
    const int r = blockIdx.x;
    const int V = lm.V;
    const size_t pn = (size_t)pm.n, pc = (size_t)lm.n_cap;
    uint32_t* off = (uint32_t*)smem;
    uint32_t* present = off + (((size_t)V + 1 + 3) & ~(size_t)3);
    uint16_t* visits = (uint16_t*)(present + ((((size_t)lm.dim + 31) / 32 + 3) & ~(size_t)3));
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    PlfRep t{};
    t.latest = pl.latest + (size_t)r * pn, t.posn = pl.posn + (size_t)r * pn, t.flag = pl.flag + (size_t)r * pc, t.roff = pl.roff + (size_t)r * (pn + 2);
    t.first = pl.first + (size_t)r * pc, t.cnl = pl.cnl + (size_t)r * pn, t.visit = pl.visit + (size_t)r * pn;
    int32_t* E = pm.earliest + (size_t)r * pn;
    int32_t* D = pm.indeg + (size_t)r * pn;
    uint32_t* Q = pm.queue + (size_t)r * pn;
    uint32_t* S = pm.lsucc + (size_t)r * pn;
    for (uint32_t i = lane; i <= (uint32_t)V; i += 64) off[i] = g_off[i];
    for (uint32_t i = lane; i < ((uint32_t)lm.dim + 31u) / 32u; i += 64) present[i] = 0u;
    plf_sync();
    const uint32_t tot0 = plf_uni(off[V]);
    for (uint32_t i = lane; i < tot0; i += 64) {
        const uint32_t x = g_visits[i];
        visits[i] = (uint16_t)x;
        atomicOr(&present[x >> 5], 1u << (x & 31u));
    }
    plf_sync();
    uint64_t trials = 0, placed = 0;
    for (int k = 0; k < n_el; ++k) {
        const uint32_t x = elements[k];
        if (x >= (uint32_t)pm.n || ((plf_uni(present[x >> 5]) >> (x & 31u)) & 1u)) continue;  // already in a list
        const uint32_t total = plf_uni(off[V]);
        if (total >= (uint32_t)lm.n_cap) break;
        trials += total + (uint32_t)V;
        const PrecResult base = prec_eval<uint16_t, PrecMemGlobal>(pm, visits, off, V, E, D, Q, S, t.first, s_info, t.roff);
        plf_sync();
        const bool base_cyc = plf_uni(s_info[1]) != 0u;
        uint32_t be = 0, bk = 0;
        if (!base_cyc) {
            plf_tails<PrecMemGlobal>(pm, t, Q, S, plf_uni(s_info[2]));
            PlfSlotPick pk{0, 0, 0, 0, 0};
            plf_best_slot<PrecMemGlobal, uint16_t>(pk, pm, t, visits, off, V, E, S, base.penalty, (int32_t)base.makespan, x, false, false, order);
            be = pk.e, bk = pk.k;
        } else {  // the element slides through every slot (ascending (list, position)), one evaluation each
            plf_list_insert(visits, off, V, 0, 0, x);
            uint64_t b1 = ~0ull, b2 = ~0ull;
            uint32_t e = 0, pos = 0, g = 0;
            for (;;) {
                const PrecResult pr = prec_eval<uint16_t, PrecMemGlobal>(pm, visits, off, V, E, D, Q, S);
                const uint64_t k1 = (uint64_t)(order == 0 ? pr.penalty : (order == 1 ? pr.makespan : pr.penalty + pr.makespan));
                const uint64_t k2 = (uint64_t)(order == 0 ? pr.makespan : (order == 1 ? pr.penalty : 0));
                if (k1 < b1 || (k1 == b1 && k2 < b2)) b1 = k1, b2 = k2, be = e, bk = pos;
                const uint32_t others = plf_uni(off[e + 1] - off[e]) - 1u;
                if (pos < others) {
                    if (lane == 0) {
                        const uint16_t y = visits[g + 1];
                        visits[g + 1] = (uint16_t)x;
                        visits[g] = y;
                    }
                    g += 1, pos += 1;
                } else if (e + 1 < (uint32_t)V) {
                    if (lane == 0) off[e + 1] -= 1;
                    e += 1, pos = 0;
                } else
                    break;
                plf_sync();
            }
            if (lane == 0) off[V] -= 1;
            plf_sync();
        }
        plf_list_insert(visits, off, V, be, bk, x);
        if (lane == 0) present[x >> 5] |= 1u << (x & 31u);
        plf_sync();
        placed += 1;
    }
    const uint32_t tot = plf_uni(off[V]);
    for (uint32_t i = lane; i < tot; i += 64) g_visits[i] = visits[i];
    for (uint32_t i = lane; i <= (uint32_t)V; i += 64) g_off[i] = off[i];
    if (stats && lane == 0) {  // live.rs: one score calculation per trial, one accepted + applied step per placed element
        uint64_t* gs = stats + (size_t)r * SF_STATS_WORDS;
        gs[0] += placed;
        gs[3] += placed;
        gs[4] += placed;
        gs[5] += trials;
        gs[7] += trials;
    }
}

// one wavefront per replica; adds the constraint's two levels to out_scores ([R][levels], already holding the other constraints)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_prec_evaluate_all(ListModel m, PrecModel pm, int64_t* out_scores, int commit, int64_t* out_parts) {
    const int r = blockIdx.x;
    const size_t n = (size_t)pm.n;
    const PrecResult pr = prec_eval<uint32_t>(pm, m.visits + (size_t)r * m.n_cap, m.off + (size_t)r * (m.V + 1), m.V, pm.earliest + r * n,
                                              pm.indeg + r * n, pm.queue + r * n, pm.lsucc + r * n);
    if ((threadIdx.x & 63u) == 0) {
        if (out_scores) {
            out_scores[(size_t)r * m.levels + pm.hard_level] -= pr.penalty;
            out_scores[(size_t)r * m.levels + pm.mk_level] -= pr.makespan;
        }
        if (commit) {
            m.score[(size_t)r * 4 + pm.hard_level] -= pr.penalty;
            m.score[(size_t)r * 4 + pm.mk_level] -= pr.makespan;
            pm.state[(size_t)r * 2] = pr.penalty;
            pm.state[(size_t)r * 2 + 1] = pr.makespan;
        }
        if (out_parts) {
            out_parts[(size_t)r * SF_EACH_WORDS + 12] = pr.penalty;
            out_parts[(size_t)r * SF_EACH_WORDS + 13] = pr.makespan;
        }
    }
}

// after a committed list move of `replica` (k_list_apply updated the lists and the other constraints' score)
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_prec_after_apply(ListModel m, PrecModel pm, int replica) {
    const int r = replica;
    const size_t n = (size_t)pm.n;
    const PrecResult pr = prec_eval<uint32_t>(pm, m.visits + (size_t)r * m.n_cap, m.off + (size_t)r * (m.V + 1), m.V, pm.earliest + r * n,
                                              pm.indeg + r * n, pm.queue + r * n, pm.lsucc + r * n);
    if ((threadIdx.x & 63u) == 0) {
        m.score[(size_t)r * 4 + pm.hard_level] -= pr.penalty - pm.state[(size_t)r * 2];
        m.score[(size_t)r * 4 + pm.mk_level] -= pr.makespan - pm.state[(size_t)r * 2 + 1];
        pm.state[(size_t)r * 2] = pr.penalty;
        pm.state[(size_t)r * 2 + 1] = pr.makespan;
    }
}

struct PrecMoveCarve {
    size_t load, off, visits, total;
    __host__ __device__ PrecMoveCarve(int V, int n_cap) {
        size_t o = 0;
        load = o, o += (size_t)V * 8;
        off = o, o += ((size_t)V + 1) * 4;
        visits = o, o += (size_t)n_cap * 2;
        total = (o + 15) & ~(size_t)15;
    }
};

// records [base, base + gridDim.x) of `moves` ([n][6] sf_move_t words) against replica `replica`: adds the constraint's delta to
// the trial scores the list kernel wrote.  Scratch slot = blockIdx.x (< R).
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_prec_evaluate_moves(ListModel m, PrecModel pm, int replica, const int32_t* moves, int64_t base,
                                                            int64_t* out_scores, const int32_t* doable) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t i = base + blockIdx.x;
    const int32_t* mv = moves + i * 6;
    const int kind = mv[0];
    if (kind < 2 || (kind > 7 && kind != 9 && kind != 10) || !doable[i]) return;
    const uint32_t lane = threadIdx.x & 63u;
    const PrecMoveCarve cv(m.V, m.n_cap);
    int64_t* s_load = (int64_t*)(smem + cv.load);
    uint32_t* s_off = (uint32_t*)(smem + cv.off);
    uint16_t* s_visits = (uint16_t*)(smem + cv.visits);
    const uint32_t* g_visits = m.visits + (size_t)replica * m.n_cap;
    const uint32_t* g_off = m.off + (size_t)replica * (m.V + 1);
    const int64_t* g_load = m.load + (size_t)replica * m.V;
    for (uint32_t t = lane; t <= (uint32_t)m.V; t += 64) s_off[t] = g_off[t];
    for (uint32_t t = lane; t < (uint32_t)m.V; t += 64) s_load[t] = g_load[t];
    wave_sync();
    const uint32_t tot = uni(s_off[m.V]);
    for (uint32_t t = lane; t < tot; t += 64) s_visits[t] = (uint16_t)g_visits[t];
    wave_sync();
    if (kind == 10) {  // multi-swap: one lane per swap (pairwise different lists)
        if (lane < (uint32_t)mv[1]) {
            const uint32_t w = (uint32_t)mv[2 + lane];
            const uint32_t base_ = s_off[w & 0xFFFFu], f = w >> 16;
            const uint32_t g = (uint32_t)((int32_t)f + (int32_t)(int8_t)(((uint32_t)mv[5] >> (8 * lane)) & 0xFFu));
            const uint16_t x = s_visits[base_ + f], y = s_visits[base_ + g];
            s_visits[base_ + f] = y, s_visits[base_ + g] = x;
        }
        wave_sync();
    } else
        apply_list_move_wave(m, s_visits, s_off, s_load, kind, (uint32_t)mv[1], (uint32_t)mv[2], (uint32_t)mv[3], (uint32_t)mv[4],
                             (uint32_t)(mv[5] > 0 ? mv[5] : 0));
    const size_t n = (size_t)pm.n, slot = blockIdx.x;
    const PrecResult pr = prec_eval<uint16_t>(pm, s_visits, s_off, m.V, pm.earliest + slot * n, pm.indeg + slot * n, pm.queue + slot * n,
                                              pm.lsucc + slot * n);
    if (lane == 0) {
        out_scores[i * m.levels + pm.hard_level] -= pr.penalty - pm.state[(size_t)replica * 2];
        out_scores[i * m.levels + pm.mk_level] -= pr.makespan - pm.state[(size_t)replica * 2 + 1];
    }
}

}  // namespace sf
