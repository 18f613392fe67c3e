// Director-surface kernels of the ListPrecedenceMakespanConstraint (sf_precedence.h): full evaluation of every replica
// (initialize / fresh_score / evaluate_each), the committed refresh after sf_apply, and n x evaluate_candidate for host-provided
// list moves (one wavefront per record: the replica's lists staged in LDS, the move applied there, one full evaluation).
#pragma once
#include "sf_precedence.h"

namespace sf {

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
    if (kind < 2 || (kind > 7 && kind != 9) || !doable[i]) return;
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
