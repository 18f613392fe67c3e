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
    const size_t pn = (size_t)pm.n, pc = (size_t)pl.pc;
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
        gs[1] += trials;  // record_construction_candidate (live.rs:123-127): a generated + evaluated candidate per trial
        gs[2] += trials;
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


// SF_MOVE_LIST_RUIN records of a host batch on a precedence model (sf_step_evaluate / sf_apply): record which[blockIdx.x] of `moves`
// against replica `replica`, one wavefront each, scratch slot = blockIdx.x (< R).  The recreate of sf_mixed_wave.hip's plf_ruin,
// restated over HBM scratch: per round one forward evaluation + one backward pass, per remaining element plf_best_slot (exact on
// acyclic lists) or the one-evaluation-per-slot slide; `hooks`: insertions that close a cycle are skipped (the record's flag bit 31,
// or the slot's precedence policy).  Writes the trial score (commit = 0) or commits the lists of the replica (commit = 1; the caller
// refreshes the committed scores).
SF_PLAIN_KERNEL
__global__ __launch_bounds__(64) void k_prec_ruin_moves(ListModel lm, PrecModel pm, PlfModel pl, int replica, const int32_t* __restrict__ moves,
                                                        const int32_t* __restrict__ which, int64_t* out_scores, int32_t* out_doable, int commit, int order,
                                                        int policy, int leaf_skip_empty) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_info[4];
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t i = which[blockIdx.x];
    const int32_t* mv = moves + i * 6;
    const int V = lm.V;
    const size_t pn = (size_t)pm.n, pc = (size_t)pl.pc, slot = blockIdx.x;
    uint32_t* off = (uint32_t*)smem;
    uint16_t* visits = (uint16_t*)(off + (((size_t)V + 1 + 3) & ~(size_t)3));
    uint32_t* g_visits = lm.visits + (size_t)replica * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)replica * (V + 1);
    PlfRep t{};
    t.latest = pl.latest + slot * pn, t.flag = pl.flag + slot * pc, t.roff = pl.roff + slot * (pn + 2), t.first = pl.first + slot * pc;
    t.cnl = pl.cnl + slot * pn, t.visit = pl.visit + slot * pn;
    int32_t* E = pm.earliest + slot * pn;
    int32_t* D = pm.indeg + slot * pn;
    uint32_t* Q = pm.queue + slot * pn;
    uint32_t* S = pm.lsucc + slot * pn;
    for (uint32_t q = lane; q <= (uint32_t)V; q += 64) off[q] = g_off[q];
    plf_sync();
    const uint32_t tot0 = plf_uni(off[V]);
    for (uint32_t q = lane; q < tot0; q += 64) visits[q] = (uint16_t)g_visits[q];
    plf_sync();
    // ---- the record (include/solverforge_amd.h: SF_MOVE_LIST_RUIN) ----
    const uint32_t cnt = (uint32_t)mv[2];
    const bool flagged = cnt <= 5 && ((uint32_t)mv[5] & 0x80000000u) != 0u;
    const bool multi = flagged && ((uint32_t)mv[5] & 0x40000000u) != 0u;
    const uint32_t second = ((uint32_t)mv[5] >> 16) & 0x3FFFu;
    const bool hooks = flagged || policy != 0;
    const bool skip_empty = (flagged && (!policy || multi)) ? false : leaf_skip_empty != 0;
    uint32_t el[PLF_EL_MAX], vals[PLF_EL_MAX];
    bool ok = cnt >= 1 && cnt <= PLF_EL_MAX && mv[1] >= 0 && mv[1] < V && (!multi || (cnt == 2 && second < (uint32_t)V));
#pragma unroll
    for (uint32_t k = 0; k < PLF_EL_MAX; ++k) {
        const uint32_t w3 = k < 2 ? (uint32_t)mv[3] : (k < 4 ? (uint32_t)mv[4] : (uint32_t)mv[5]);
        uint32_t pos = (w3 >> (16u * (k & 1u))) & 0xFFFFu;
        if (flagged && k == 5) pos = 0;
        const uint32_t list = (multi && k == cnt - 1) ? second : (uint32_t)mv[1];
        el[k] = (list << 16) | pos;
        vals[k] = 0;
        if (ok && k < cnt) {
            ok = pos < off[list + 1] - off[list];
            if (k > 0 && el[k] <= el[k - 1]) ok = false;  // ascending (list, position)
        }
    }
    ok = plf_uni(ok ? 1u : 0u) != 0u;
    if (lane == 0) out_doable[i] = ok ? 1 : 0;
    if (!ok) return;
    const int64_t pen0 = pm.state[(size_t)replica * 2], mk0 = pm.state[(size_t)replica * 2 + 1];
    auto key_of = [&](int64_t pen, int64_t mk, uint64_t& k1, uint64_t& k2) {
        k1 = (uint64_t)(order == 0 ? pen : (order == 1 ? mk : pen + mk));
        k2 = (uint64_t)(order == 0 ? mk : (order == 1 ? pen : 0));
    };
    for (uint32_t k = cnt; k-- > 0;) {
        const uint32_t x = plf_list_remove(visits, off, V, el[k] >> 16, el[k] & 0xFFFFu);
#pragma unroll
        for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
            if (q == k) vals[q] = x;
    }
    uint32_t remaining = (1u << cnt) - 1u;
    int64_t last_pen = pen0, last_mk = mk0;
    bool rolled = false;
    for (uint32_t round = 0; round < cnt && !rolled; ++round) {
        bool have = false;
        uint64_t b1 = ~0ull, b2 = ~0ull;
        int64_t b_pen = 0, b_mk = 0;
        uint32_t b_ri = 0, b_e = 0, b_pos = 0;
        const PrecResult base = prec_eval<uint16_t, PrecMemGlobal>(pm, visits, off, V, E, D, Q, S, t.first, s_info, t.roff);
        plf_sync();
        const bool base_cyc = plf_uni(s_info[1]) != 0u;
        if (!base_cyc) plf_tails<PrecMemGlobal>(pm, t, Q, S, plf_uni(s_info[2]));
        for (uint32_t ri = 0; ri < cnt; ++ri) {
            if (!((remaining >> ri) & 1u)) continue;
            uint32_t x = 0;
#pragma unroll
            for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
                if (q == ri) x = vals[q];
            x = plf_uni(x);
            if (!base_cyc) {
                PlfSlotPick pk{0, 0, 0, 0, 0};
                plf_best_slot<PrecMemGlobal, uint16_t>(pk, pm, t, visits, off, V, E, S, base.penalty, (int32_t)base.makespan, x, hooks, skip_empty, order);
                if (pk.found) {
                    uint64_t k1, k2;
                    key_of(pk.pen, pk.mk, k1, k2);
                    if (!have || k1 < b1 || (k1 == b1 && k2 < b2)) have = true, b1 = k1, b2 = k2, b_pen = pk.pen, b_mk = pk.mk, b_ri = ri, b_e = pk.e, b_pos = pk.k;
                }
                continue;
            }
            plf_list_insert(visits, off, V, 0, 0, x);
            uint32_t e = 0, pos = 0, g = 0;
            for (;;) {
                const uint32_t others = plf_uni(off[e + 1] - off[e]) - 1u;
                if (!(skip_empty && others == 0u)) {
                    const PrecResult pr = prec_eval<uint16_t, PrecMemGlobal>(pm, visits, off, V, E, D, Q, S, nullptr, s_info, nullptr);
                    plf_sync();
                    const bool cyc = plf_uni(s_info[1]) != 0u;
                    if (!(cyc && hooks)) {
                        uint64_t k1, k2;
                        key_of(pr.penalty, pr.makespan, k1, k2);
                        if (!have || k1 < b1 || (k1 == b1 && k2 < b2)) have = true, b1 = k1, b2 = k2, b_pen = pr.penalty, b_mk = pr.makespan, b_ri = ri, b_e = e, b_pos = pos;
                    }
                }
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
        if (!have) {
            rolled = true;
            break;
        }
        b_ri = plf_uni(b_ri), b_e = plf_uni(b_e), b_pos = plf_uni(b_pos);
        uint32_t bx = 0;
#pragma unroll
        for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
            if (q == b_ri) bx = vals[q];
        plf_list_insert(visits, off, V, b_e, b_pos, plf_uni(bx));
        remaining &= ~(1u << b_ri);
        last_pen = b_pen, last_mk = b_mk;
    }
    if (rolled) last_pen = pen0, last_mk = mk0;  // restore_removed_elements: the move leaves the lists as they were
    if (lane == 0) {
        for (int k = 0; k < lm.levels; ++k) out_scores[i * lm.levels + k] = lm.score[(size_t)replica * 4 + k];
        out_scores[i * lm.levels + pm.hard_level] -= last_pen - pen0;
        out_scores[i * lm.levels + pm.mk_level] -= last_mk - mk0;
    }
    if (commit && !rolled) {
        const uint32_t tot = plf_uni(off[V]);
        for (uint32_t q = lane; q < tot; q += 64) g_visits[q] = visits[q];
        for (uint32_t q = lane; q <= (uint32_t)V; q += 64) g_off[q] = off[q];
    }
}

}  // namespace sf
