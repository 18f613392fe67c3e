// Critical-path precedence leaf on the device (generic engine, PREC instantiations): the reference's
// ListPrecedenceMoveSelector / RuntimeListNeighborhoodSpec::Precedence stream, restated for one wavefront per replica.
//
// Reference semantics restated (paths under crates/solverforge-solver/src/heuristic/selector/):
//   list_kernel/precedence/analysis.rs:56-192     earliest / latest starts -> critical nodes and arcs -> critical blocks
//   list_kernel/precedence/coordinates.rs:15-326  the seven move families of a block, the tiered order, cycle pruning
//   list_kernel/precedence/support.rs:22-161      critical / support adjacent swaps, multi-swap triples, multi-block ruins
//   list_kernel/precedence/cursor.rs:182-252      stream order: multi-swaps, multi-ruins, blocks
//   precedence_route.rs:171-304                   the cycle tests (= "the lists after the move are cyclic", see below)
//
// Analysis (once per step, plf_analyse): the committed evaluation of the precedence constraint already leaves the earliest
// starts, Kahn's pop order and its rounds; the latest starts are one sweep over the rounds in reverse (the nodes of a round
// are mutually independent), every list position gets (critical node, critical arc to the next position) flags, and blocks /
// critical swaps / critical nodes are ballot compactions of those flags in list order (= the reference's entity-major scan).
// The support swaps keep the reference's first-occurrence order through an atomic min of a sequence number per swap slot.
//
// Cycle pruning: the reference tests whether the edges a move adds close a cycle over (fixed edges + route edges - removed
// edges) (added_edges_introduce_cycle / route_delta_has_cycle).  The current graph is acyclic whenever the leaf has blocks, so
// that is exactly "the graph of the lists after the move is cyclic", which the trial evaluation of the constraint (prec_eval)
// reports anyway: a pruned candidate is one whose trial comes back cyclic, and it never reaches the ring.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_precedence.h"

namespace sf {

constexpr uint64_t SALT_PL_BLOCK = 0xC9171EAF5EED0001ULL, SALT_PL_MOVE = 0xC9171EAF5EED0002ULL;  // cursor.rs:196-246
constexpr uint64_t SALT_PL_MULTI_RUIN = 0xC9171EAF5EED0003ULL, SALT_PL_MULTI_SWAP = 0xC9171EAF5EED0004ULL;
constexpr uint64_t SALT_PL_ADJACENT = 0xAD1ACE1700000001ULL, SALT_PL_BOUNDARY = 0xAD1ACE1700000002ULL, SALT_PL_REST = 0xAD1ACE1700000003ULL;  // coordinates.rs:91-130
constexpr uint32_t PLF_RUIN_MAX = 5, PLF_SUBLIST_MAX = 3, PLF_PERMUTE_MAX = 5;  // coordinates.rs:11-13
constexpr uint32_t PLF_EL_MAX = 6;  // elements of one ruin move (the ordinary ruin leaf takes up to six, sf_ruin.h)

// The leaf's scratch: per replica in HBM (L2-resident at the sizes the leaf is used at), carved on the host.
struct PlfModel {
    int32_t on;
    int32_t leaf;       // the union has the critical-path leaf (kind 16384): the blocks are analysed every step
    int32_t policy;     // the slot declares its precedence hooks to the runtime leaves (list_leaf/cursor/slot.rs:191-404): the other list leaves
                        // drop intra-list candidates that close a cycle through a new route edge, the ruin leaf recreates with the hooks
    int32_t slow;       // diagnostics / parity tests (SF_AMD_PLF_SLOW): the recreate slides every element through every slot, one evaluation each
    int32_t force64;    // diagnostics / parity tests (SF_AMD_PLF_FORCE64): the multi-swap stream takes its 64-bit index path whatever its length
    int32_t dmax;       // max (fixed successors + fixed predecessors) of a node: spacing of the support-swap sequence numbers
    int32_t pc;         // row stride of `flag` / `first`: max(node_count, element_capacity) -- they are indexed by list position in the leaf's
                        // analysis and by node id (predecessor / reachability tables) in the recreate and the construction
    int32_t* latest;    // [R][n]
    uint32_t* posn;     // [R][n]      node -> (list << 16 | position), PREC_NONE = in no list
    uint32_t* flag;     // [R][pc]     per list position: bit 0 critical node, bit 1 critical arc to the next position, bit 2 first of its list
    uint32_t* roff;     // [R][n + 2]  Kahn rounds of the committed evaluation
    uint32_t* blk;      // [R][n][2]   (list << 16 | start, len << 16 | route_len)
    uint32_t* csw;      // [R][n]      critical adjacent swaps (list << 16 | position)
    uint32_t* ssw;      // [R][n]      support adjacent swaps, first-occurrence order
    uint32_t* first;    // [R][pc]     smallest sequence number that named the swap slot
    uint32_t* cnl;      // [R][n]      list positions of the critical nodes, in list order
    uint64_t* msrow;    // [R][n + 1]  multi-swap candidates before the rows of critical swap i (64 bits: C^2 * S / 2 passes 2^32 near 2,000 nodes)
    uint32_t* mrrow;    // [R][n + 1]  multi-ruin candidates before the rows of block i
    uint32_t* sE;       // [R][V]      support swaps per list
    int64_t* score;     // [R][GRC][4] trial scores of the ring entries
    uint32_t* visit;    // [R][n]      visited marks of the reachability search (cyclic working state only)
    int64_t* cache;     // [R][GL][GRC][2] (hard penalty, makespan) of the ring entries the route-graph filter evaluated; penalty INT64_MIN = none
};
struct PlfRep {  // one replica's slices + the counts of this step (wave-uniform)
    int32_t* latest;
    uint32_t *posn, *flag, *roff, *blk, *csw, *ssw, *first, *cnl, *mrrow, *sE, *visit;
    uint64_t* msrow;
    uint32_t nb, C, S, mr_count;
    uint64_t ms_count;
};

// Hand-off through HBM between lanes of one wavefront.  The workgroup-scope fences of prec_sync() compile to nothing on gfx950 (one
// L1 per CU), which leaves a plain store and a later L2 atomic (or sc1 load) of ANOTHER lane to the same word unordered: wait for
// the wave's outstanding vector memory operations explicitly.
__device__ __forceinline__ void plf_gsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint32_t plf_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t plf_ald(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t plf_aldi(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One step's analysis.  E / Q / S = earliest starts, pop order, list successor of the committed evaluation that just ran
// (rounds in t.roff); `cyclic` = that evaluation left nodes unprocessed: no blocks at all (analysis.rs:61-70).
template <class MEM>
__device__ __noinline__ void plf_analyse(const PrecModel& pm_ref, const PlfModel& pl_ref, PlfRep& t, const uint16_t* visits, const uint32_t* off, int V,
                                         typename MEM::I32 E, typename MEM::U32 Q, typename MEM::U32 S, uint32_t rounds, int32_t mk, bool cyclic) {
    const PrecModel pm = pm_ref;  // (private copies: see prec_eval)
    const PlfModel pl = pl_ref;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    t.nb = t.C = t.S = t.ms_count = t.mr_count = 0;
    const uint32_t total = plf_uni(off[V]);
    // ---- node -> (list, position) of the committed lists (also what the route-graph filter of the other leaves reads) ----
    for (uint32_t i = lane; i < n; i += 64) t.posn[i] = PREC_NONE;
    plf_gsync();
    for (uint32_t p0 = 0; p0 < total; p0 += 64) {
        const uint32_t p = p0 + lane;
        if (p < total) {
            uint32_t lo = 0, hi = (uint32_t)V;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (off[mid] <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            t.posn[visits[p]] = (lo << 16) | (p - off[lo]);
        }
    }
    plf_gsync();
    if (cyclic || !pl.leaf) return;
    // ---- latest starts: the rounds in reverse ----
    for (uint32_t rd = rounds; rd-- > 0;) {
        const uint32_t lo = plf_uni(plf_ald(t.roff + rd)), hi = plf_uni(plf_ald(t.roff + rd + 1));
        const uint32_t i = lo + lane;
        if (i < hi) {
            const uint32_t w = MEM::ld(Q + i);
            const int32_t d = pm.dur[w];
            int32_t best = INT32_MAX;
            for (uint32_t k = pm.succ_off[w]; k < pm.succ_off[w + 1]; ++k) {
                const int32_t c = t.latest[pm.succ[k]] - d;
                best = c < best ? c : best;
            }
            const uint32_t ls = MEM::ld(S + w);
            if (ls != PREC_NONE) {
                const int32_t c = t.latest[ls] - d;
                best = c < best ? c : best;
            }
            t.latest[w] = best == INT32_MAX ? mk - d : best;
        }
        plf_gsync();
    }
    // ---- critical flags per list position ----
    for (uint32_t v = lane; v < (uint32_t)V; v += 64) t.sE[v] = 0;
    plf_gsync();
    for (uint32_t p0 = 0; p0 < total; p0 += 64) {
        const uint32_t p = p0 + lane;
        if (p < total) {
            uint32_t lo = 0, hi = (uint32_t)V;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (off[mid] <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            const uint32_t x = visits[p];
            const int32_t ex = MEM::ld(E + x);
            const bool cn = ex == t.latest[x];
            bool arc = false;
            if (cn && p + 1 < off[lo + 1]) {
                const uint32_t y = visits[p + 1];
                const int32_t ey = MEM::ld(E + y);
                arc = ey == t.latest[y] && ex + pm.dur[x] == ey;
            }
            t.flag[p] = (cn ? 1u : 0u) | (arc ? 2u : 0u) | (p == off[lo] ? 4u : 0u);
            t.first[p] = 0xFFFFFFFFu;
        }
    }
    plf_gsync();
    // ---- blocks, critical swaps, critical nodes: compactions in list order ----
    uint32_t nb = 0, C = 0, NC = 0;
    for (uint32_t p0 = 0; p0 < total; p0 += 64) {
        const uint32_t p = p0 + lane;
        bool cn = false, arc = false, start = false;
        uint32_t where = 0;
        if (p < total) {
            const uint32_t fl = t.flag[p];
            cn = fl & 1u, arc = (fl >> 1) & 1u;
            const bool prevarc = !(fl & 4u) && ((t.flag[p - 1] >> 1) & 1u);
            start = cn && !prevarc;
            where = t.posn[visits[p]];
        }
        const uint64_t mc = __ballot(cn), ma = __ballot(arc), ms = __ballot(start);
        if (cn) t.cnl[NC + prec_mbcnt(mc)] = p;
        if (arc) t.csw[C + prec_mbcnt(ma)] = where;
        if (start) {
            uint32_t q = p;
            while ((t.flag[q] >> 1) & 1u) ++q;
            const uint32_t e = where >> 16;
            const uint32_t bi = nb + prec_mbcnt(ms);
            t.blk[2 * bi] = where;
            t.blk[2 * bi + 1] = ((q - p + 1) << 16) | (off[e + 1] - off[e]);
        }
        NC += (uint32_t)__popcll(mc), C += (uint32_t)__popcll(ma), nb += (uint32_t)__popcll(ms);
    }
    plf_gsync();
    // ---- support swaps (support.rs:38-62,163-188): the swaps around the fixed successors / predecessors of every critical node, first
    // occurrence order.  Item k of critical node t has the sequence number t * 2 * dmax + 2 * k (+ 1 for the swap after the node). ----
    const uint32_t span = 2u * (uint32_t)pl.dmax;
    auto for_items = [&](uint32_t tnode, auto&& fn) {  // fn(seq, swap slot (global position), list, position)
        const uint32_t p = t.cnl[tnode];
        const uint32_t x = visits[p];
        const uint32_t so = pm.succ_off[x], ds = pm.succ_off[x + 1] - so, po = pm.pred_off[x], dp = pm.pred_off[x + 1] - po;
        for (uint32_t k = 0; k < ds + dp; ++k) {
            const uint32_t y = k < ds ? pm.succ[so + k] : pm.pred[po + k - ds];
            const uint32_t w = t.posn[y];
            if (w == PREC_NONE) continue;
            const uint32_t e = w >> 16, pos = w & 0xFFFFu, base = off[e], len = off[e + 1] - base;
            const uint32_t seq = tnode * span + 2u * k;
            if (pos > 0) fn(seq, base + pos - 1, e, pos - 1);
            if (pos + 1 < len) fn(seq + 1, base + pos, e, pos);
        }
    };
    for (uint32_t t0 = 0; t0 < NC; t0 += 64)
        if (t0 + lane < NC)
            for_items(t0 + lane, [&](uint32_t seq, uint32_t slot, uint32_t, uint32_t) {
                __hip_atomic_fetch_min(t.first + slot, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
    plf_gsync();
    uint32_t Sn = 0;
    for (uint32_t t0 = 0; t0 < NC; t0 += 64) {
        uint32_t mine = 0;
        if (t0 + lane < NC)
            for_items(t0 + lane, [&](uint32_t seq, uint32_t slot, uint32_t, uint32_t) { mine += plf_ald(t.first + slot) == seq ? 1u : 0u; });
        const uint32_t incl = wave_incl_scan(mine);
        uint32_t at = Sn + incl - mine;
        if (t0 + lane < NC)
            for_items(t0 + lane, [&](uint32_t seq, uint32_t slot, uint32_t e, uint32_t pos) {
                if (plf_ald(t.first + slot) == seq) {
                    t.ssw[at++] = (e << 16) | pos;
                    __hip_atomic_fetch_add(t.sE + e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            });
        Sn += (uint32_t)__shfl((int)incl, 63);
    }
    plf_gsync();
    // ---- multi-swap rows (support.rs:64-84): row i = the triples whose first critical swap is i ----
    uint64_t ms_total = 0;  // a row holds < C * S <= n^2 < 2^32 triples (n <= 65,535), the stream C^2 * S / 2 of them: 64-bit prefixes
    for (uint32_t i0 = 0; i0 < C; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t cnt = 0;
        if (i < C) {
            const uint32_t ei = t.csw[i] >> 16;
            const uint32_t si = plf_ald(t.sE + ei);
            for (uint32_t j = i + 1; j < C; ++j) {
                const uint32_t ej = t.csw[j] >> 16;
                if (ej != ei) cnt += Sn - si - plf_ald(t.sE + ej);
            }
        }
        // 64-bit inclusive scan of the 64 row sizes (their sum may pass 2^32): low and high halves scanned apart
        uint64_t incl = (uint64_t)cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o);
            if ((int)lane >= o) incl += ((uint64_t)hi << 32) | lo;
        }
        if (i < C) t.msrow[i] = ms_total + incl - cnt;
        ms_total += ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)incl, 63);
    }
    if (lane == 0) t.msrow[C] = ms_total;
    // ---- multi-ruin rows (support.rs:124-161): row i = len_i * (the lengths of the blocks after i) ----
    uint32_t len_total = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += 64) {
        uint32_t l = i0 + lane < nb ? t.blk[2 * (i0 + lane) + 1] >> 16 : 0u;
#pragma unroll
        for (int o = 32; o; o >>= 1) l += (uint32_t)__shfl_xor((int)l, o);
        len_total += l;
    }
    uint32_t len_before = 0, mr_total = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t l = i < nb ? t.blk[2 * i + 1] >> 16 : 0u;
        const uint32_t li = wave_incl_scan(l);
        const uint32_t row = l * (len_total - (len_before + li));
        const uint32_t ri = wave_incl_scan(row);
        if (i < nb) t.mrrow[i] = mr_total + ri - row;
        len_before += (uint32_t)__shfl((int)li, 63);
        mr_total += (uint32_t)__shfl((int)ri, 63);
    }
    if (lane == 0) t.mrrow[nb] = mr_total;
    plf_gsync();
    t.nb = nb, t.C = C, t.S = Sn, t.ms_count = ms_total, t.mr_count = mr_total;
}

// Does `from` reach `target` over (fixed successors + the list successors S of the lists just evaluated)?  One wavefront, breadth
// first, 64 frontier nodes per round; visit / queue = n words each.  The slow half of the route-graph filter: only a cyclic working
// state needs it (reaches_with_route_delta, precedence_route.rs:519-556, over the graph after the move).
template <class MEM>
__device__ __noinline__ bool plf_reaches(const PrecModel& pm_ref, uint32_t* visit, uint32_t* queue, typename MEM::U32 S, uint32_t from, uint32_t target) {
    const PrecModel pm = pm_ref;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    for (uint32_t i = lane; i < n; i += 64) visit[i] = 0;
    plf_gsync();
    if (lane == 0) {
        __hip_atomic_store(visit + from, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(queue, from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    plf_gsync();
    uint32_t head = 0, tail = 1;
    bool found = from == target;
    while (head < tail && !found) {
        const uint32_t cnt = tail - head < 64u ? tail - head : 64u;
        const bool act = lane < cnt;
        uint32_t so = 0, deg = 0, ls = PREC_NONE;
        if (act) {
            const uint32_t w = plf_ald(queue + head + lane);
            so = pm.succ_off[w];
            deg = pm.succ_off[w + 1] - so;
            ls = MEM::ld(S + w);
        }
        const uint32_t degt = deg + ((act && ls != PREC_NONE) ? 1u : 0u);
        bool hit = false;
        for (uint32_t k = 0;; ++k) {
            const bool has = k < degt;
            if (!__ballot(has)) break;
            bool fresh = false;
            uint32_t s = 0;
            if (has) {
                s = k < deg ? pm.succ[so + k] : ls;
                hit = hit || s == target;
                fresh = __hip_atomic_exchange(visit + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
            }
            const uint64_t m = __ballot(fresh);
            if (fresh) __hip_atomic_store(queue + tail + prec_mbcnt(m), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail += (uint32_t)__popcll(m);
        }
        found = __ballot(hit) != 0;
        head += cnt;
        plf_gsync();
    }
    return found;
}


// ---- one element of one recreate round, every insertion slot at once (acyclic lists) ----------------------------------------------
// The lists hold everything but the element x.  One forward evaluation (earliest starts E, pop order Q with its rounds, list
// successors S, list predecessors LP) and one backward pass (TAIL[v] = longest path from the start of v to the end) price every
// slot without touching the lists: a slot (list e, position k) puts x between p = L[k-1] and q = L[k]; the longest path through x is
//   max(finish(p), finish(fixed predecessors of x)) + dur(x) + max(TAIL(q), TAIL(fixed successors of x))
// and every other path of the new graph is a path of the old one (the edge p -> q only moves onto p -> x -> q), so the makespan is
// the max of that and the old makespan.  The new graph is cyclic iff a successor side of x reaches a predecessor side: q in A,
// p in B or a fixed predecessor in B, with A = the ancestors of x's fixed predecessors (themselves included) and B = the descendants
// of x's fixed successors (themselves included) -- two breadth-first searches.  Result: the best slot by (hard penalty, makespan) in
// the score's level order, the first in (list, position) order among equals; `hooks`: cyclic slots are skipped, else they are priced
// as the constraint prices a cycle (+ node_count hard, makespan 0).
struct PlfSlotPick {
    uint32_t found, e, k;
    int64_t pen, mk;
};
// the backward pass of a recreate round (once per round: the lists are the same for every remaining element)
template <class MEM>
__device__ __noinline__ void plf_tails(const PrecModel& pm_ref, const PlfRep& t, typename MEM::U32 Q, typename MEM::U32 S, uint32_t rounds) {
    const PrecModel pm = pm_ref;
    const uint32_t lane = threadIdx.x & 63u;
    int32_t* const TAIL = t.latest;
    plf_gsync();
    for (uint32_t rd = rounds; rd-- > 0;) {
        const uint32_t lo = plf_uni(plf_ald(t.roff + rd)), hi = plf_uni(plf_ald(t.roff + rd + 1));
        const uint32_t i = lo + lane;
        if (i < hi) {
            const uint32_t w = MEM::ld(Q + i);
            const uint32_t r0 = pm.nd[2 * (size_t)w], r1 = pm.nd[2 * (size_t)w + 1];  // duration; out-degree << 24 | first fixed successor
            const uint32_t ls = MEM::ld(S + w), s1 = r1 & 0xFFFFFFu;
            const int32_t c1 = s1 != 0xFFFFFFu ? plf_aldi(TAIL + s1) : 0, c2 = ls != PREC_NONE ? plf_aldi(TAIL + ls) : 0;  // both tails in flight together
            int32_t best = c1 > c2 ? c1 : c2;
            if ((r1 >> 24) > 1u)  // further fixed successors (none in a job shop)
                for (uint32_t k = pm.succ_off[w] + 1; k < pm.succ_off[w + 1]; ++k) {
                    const int32_t c = plf_aldi(TAIL + pm.succ[k]);
                    best = c > best ? c : best;
                }
            TAIL[w] = best + (int32_t)r0;
        }
        plf_gsync();
    }
}
template <class MEM, class VT = uint16_t>
__device__ __noinline__ void plf_best_slot(PlfSlotPick& r, const PrecModel& pm_ref, const PlfRep& t, const VT* visits, const uint32_t* off, int V, typename MEM::I32 E,
                                                  typename MEM::U32 S, int64_t base_pen, int32_t base_mk, uint32_t x, bool hooks, bool skip_empty,
                                                  int order /* 0: penalty first, 1: makespan first, 2: one level */) {
    const PrecModel pm = pm_ref;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    int32_t* const TAIL = t.latest;
    uint32_t* const LP = t.first;   // list predecessors (written by the evaluation)
    uint32_t* const INA = t.visit;  // ancestors of x's fixed predecessors
    uint32_t* const INB = t.flag;   // descendants of x's fixed successors
    uint32_t* const QUE = t.cnl;
    // ---- A and B ----
    for (uint32_t i = lane; i < n; i += 64) INA[i] = 0, INB[i] = 0;
    plf_gsync();
    for (int side = 0; side < 2; ++side) {
        uint32_t* const mark = side == 0 ? INA : INB;
        const uint32_t so = side == 0 ? pm.pred_off[x] : pm.succ_off[x], sn = (side == 0 ? pm.pred_off[x + 1] : pm.succ_off[x + 1]) - so;
        uint32_t tail = 0;
        for (uint32_t k0 = 0; k0 < sn; k0 += 64) {  // seeds
            const bool has = k0 + lane < sn;
            uint32_t y = 0;
            bool fresh = false;
            if (has) {
                y = side == 0 ? pm.pred[so + k0 + lane] : pm.succ[so + k0 + lane];
                fresh = __hip_atomic_exchange(mark + y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
            }
            const uint64_t m = __ballot(fresh);
            if (fresh) __hip_atomic_store(QUE + tail + prec_mbcnt(m), y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail += (uint32_t)__popcll(m);
        }
        plf_gsync();
        uint32_t head = 0;
        while (head < tail) {
            const uint32_t cnt = tail - head < 64u ? tail - head : 64u;
            const bool act = lane < cnt;
            uint32_t fo = 0, deg = 0, y1 = PREC_NONE, ln = PREC_NONE;
            if (act) {  // the first fixed neighbour on this side (descendants: from the node record) and the list neighbour
                const uint32_t w = plf_ald(QUE + head + lane);
                if (side == 0) {
                    fo = pm.pred_off[w];
                    deg = pm.pred_off[w + 1] - fo;
                    if (deg) y1 = pm.pred[fo];
                    ln = plf_ald(LP + w);
                } else {
                    const uint32_t rec = pm.nd[2 * (size_t)w + 1];
                    deg = rec >> 24;
                    y1 = (rec & 0xFFFFFFu) == 0xFFFFFFu ? PREC_NONE : (rec & 0xFFFFFFu);
                    if (deg > 1u) {
                        fo = pm.succ_off[w];
                        deg = pm.succ_off[w + 1] - fo;  // (the record saturates at 255)
                    }
                    ln = MEM::ld(S + w);
                }
            }
            // both marks in flight together
            const bool h1 = y1 != PREC_NONE, h2 = ln != PREC_NONE;
            uint32_t o1 = 1u, o2 = 1u;
            if (h1) o1 = __hip_atomic_exchange(mark + y1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h2) o2 = __hip_atomic_exchange(mark + ln, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool f1 = h1 && o1 == 0u, f2 = h2 && o2 == 0u;  // (the same node twice: the first exchange took it)
            const uint64_t m1 = __ballot(f1), m2 = __ballot(f2);
            if (f1) __hip_atomic_store(QUE + tail + prec_mbcnt(m1), y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail += (uint32_t)__popcll(m1);
            if (f2) __hip_atomic_store(QUE + tail + prec_mbcnt(m2), ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail += (uint32_t)__popcll(m2);
            if (__ballot(deg > 1u)) {  // further fixed neighbours (none in a job shop)
                for (uint32_t k = 1;; ++k) {
                    const bool has = k < deg;
                    if (!__ballot(has)) break;
                    bool fresh = false;
                    uint32_t y = 0;
                    if (has) {
                        y = side == 0 ? pm.pred[fo + k] : pm.succ[fo + k];
                        fresh = __hip_atomic_exchange(mark + y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
                    }
                    const uint64_t m = __ballot(fresh);
                    if (fresh) __hip_atomic_store(QUE + tail + prec_mbcnt(m), y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tail += (uint32_t)__popcll(m);
                }
            }
            head += cnt;
            plf_gsync();
        }
    }
    // ---- what x brings by itself: its fixed predecessors' latest finish, its fixed successors' longest tail, a cycle through them alone ----
    int32_t hf = 0, ts = 0;
    bool always = false;
    for (uint32_t k = pm.pred_off[x]; k < pm.pred_off[x + 1]; ++k) {
        const uint32_t f = pm.pred[k];
        const int32_t c = MEM::ld(E + f) + pm.dur[f];
        hf = c > hf ? c : hf;
        always = always || plf_ald(INB + f) != 0u;
    }
    for (uint32_t k = pm.succ_off[x]; k < pm.succ_off[x + 1]; ++k) {
        const int32_t c = plf_aldi(TAIL + pm.succ[k]);
        ts = c > ts ? c : ts;
    }
    const int32_t dx = pm.dur[x];
    const int32_t ox = pm.owner ? pm.owner[x] : -1;
    // ---- every slot: lane s handles the slots s, s + 64, .. in (list, position) order ----
    const uint32_t total = plf_uni(off[V]);
    const uint32_t slots = total + (uint32_t)V;
    uint64_t bk1 = ~0ull, bk2 = ~0ull;
    uint32_t bslot = 0xFFFFFFFFu, be = 0, bkk = 0;
    int64_t bpen = 0, bmk = 0;
    for (uint32_t s0 = 0; s0 < slots; s0 += 64) {
        const uint32_t sl = s0 + lane;
        if (sl >= slots) continue;
        // slot sl of list e at position k: slots of list e start at off[e] + e
        uint32_t lo = 0, hi = (uint32_t)V;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] + mid <= sl)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t e = lo, k = sl - (off[e] + e), len = off[e + 1] - off[e];
        if (skip_empty && len == 0) continue;
        const uint32_t p = k > 0 ? (uint32_t)visits[off[e] + k - 1] : PREC_NONE, q = k < len ? (uint32_t)visits[off[e] + k] : PREC_NONE;
        const bool cyc = always || (q != PREC_NONE && plf_ald(INA + q) != 0u) || (p != PREC_NONE && plf_ald(INB + p) != 0u);
        if (cyc && hooks) continue;
        int64_t pen = base_pen - 1 + ((ox >= 0 && (uint32_t)ox != e) ? 1 : 0), mk;
        if (cyc) {
            pen += (int64_t)n;
            mk = 0;
        } else {
            const int32_t hp = p != PREC_NONE ? MEM::ld(E + p) + pm.dur[p] : 0;
            const int32_t tq = q != PREC_NONE ? plf_aldi(TAIL + q) : 0;
            const int32_t through = (hp > hf ? hp : hf) + dx + (tq > ts ? tq : ts);
            mk = through > base_mk ? through : base_mk;
        }
        const uint64_t k1 = (uint64_t)(order == 0 ? pen : (order == 1 ? mk : pen + mk)), k2 = (uint64_t)(order == 0 ? mk : (order == 1 ? pen : 0));
        if (k1 < bk1 || (k1 == bk1 && k2 < bk2)) {  // ascending slots per lane: strict improvement keeps the first of equals
            bk1 = k1, bk2 = k2, bslot = sl, be = e, bkk = k, bpen = pen, bmk = mk;
        }
    }
    // wave argmin of (k1, k2, slot)
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const uint64_t o1 = shfl_xor_u64(bk1, o), o2 = shfl_xor_u64(bk2, o);
        const uint32_t os = (uint32_t)__shfl_xor((int)bslot, o), oe = (uint32_t)__shfl_xor((int)be, o), ok = (uint32_t)__shfl_xor((int)bkk, o);
        const int64_t op = (int64_t)shfl_xor_u64((uint64_t)bpen, o), om = (int64_t)shfl_xor_u64((uint64_t)bmk, o);
        const bool take = os != 0xFFFFFFFFu && (bslot == 0xFFFFFFFFu || o1 < bk1 || (o1 == bk1 && (o2 < bk2 || (o2 == bk2 && os < bslot))));
        if (take) bk1 = o1, bk2 = o2, bslot = os, be = oe, bkk = ok, bpen = op, bmk = om;
    }
    r.found = plf_uni(bslot != 0xFFFFFFFFu ? 1u : 0u);
    r.e = plf_uni(be), r.k = plf_uni(bkk);
    r.pen = (int64_t)(((uint64_t)plf_uni((uint32_t)((uint64_t)bpen >> 32)) << 32) | plf_uni((uint32_t)bpen));
    r.mk = (int64_t)(((uint64_t)plf_uni((uint32_t)((uint64_t)bmk >> 32)) << 32) | plf_uni((uint32_t)bmk));
}

// ---- decoding (wave-uniform arguments and results) ----
struct PlfMove {
    int32_t kind;               // sf_move_kind: 2 change, 3 swap, 4 reverse, 5 sublist change, 6 sublist swap, 8 ruin, 9 permute, 10 multi-swap
    uint32_t a, ap, b, bp, ext;  // the arguments of apply_list_move_wave for the single-list kinds
    uint32_t n;                  // ruin: elements (list << 16 | position), sorted by (list, position); multi-swap: (list << 16 | first position)
    uint32_t el[PLF_EL_MAX];
};

// largest i in [0, rows) with prefix[i] <= idx (prefix exclusive, prefix[rows] = total > idx)
__device__ __forceinline__ uint32_t plf_find_row(const uint32_t* prefix, uint32_t rows, uint32_t idx) {
    uint32_t lo = 0, hi = rows;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (plf_uni(prefix[mid]) <= idx)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// first j in [from, to) whose running sum of v(j) exceeds `offset`; offset becomes the remainder inside j.  v is evaluated one
// index per lane.
template <class F>
__device__ __forceinline__ uint32_t plf_walk(uint32_t from, uint32_t to, uint32_t& offset, F&& v) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t j0 = from; j0 < to; j0 += 64) {
        const uint32_t j = j0 + lane;
        const uint32_t c = j < to ? v(j) : 0u;
        const uint32_t incl = wave_incl_scan(c);
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
        if (offset < tot) {
            const uint64_t m = __ballot(incl > offset);
            const int hit = __ffsll((unsigned long long)m) - 1;
            offset -= (uint32_t)__shfl((int)(incl - c), hit);
            return j0 + (uint32_t)hit;
        }
        offset -= tot;
    }
    return to;  // not reached for a valid offset
}

// multi_support_swaps (support.rs:86-109)
__device__ __noinline__ void plf_decode_multi_swap(const PlfRep& t, uint64_t idx, PlfMove& m) {
    uint32_t i = 0;
    {  // largest row whose 64-bit prefix is <= idx
        uint32_t lo = 0, hi = t.C;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t pm_ = t.msrow[mid];
            const uint64_t pu = ((uint64_t)plf_uni((uint32_t)(pm_ >> 32)) << 32) | plf_uni((uint32_t)pm_);
            if (pu <= idx)
                lo = mid;
            else
                hi = mid;
        }
        i = lo;
    }
    const uint64_t row0 = t.msrow[i];
    uint32_t off = (uint32_t)(idx - (((uint64_t)plf_uni((uint32_t)(row0 >> 32)) << 32) | plf_uni((uint32_t)row0)));  // inside a row: < 2^32
    const uint32_t ci = plf_uni(t.csw[i]), ei = ci >> 16;
    const uint32_t si = plf_uni(plf_ald(t.sE + ei));
    const uint32_t j = plf_walk(i + 1, t.C, off, [&](uint32_t jj) {
        const uint32_t ej = t.csw[jj] >> 16;
        return ej != ei ? t.S - si - plf_ald(t.sE + ej) : 0u;
    });
    const uint32_t cj = plf_uni(t.csw[j]), ej = cj >> 16;
    const uint32_t s = plf_walk(0, t.S, off, [&](uint32_t ss) {
        const uint32_t es = t.ssw[ss] >> 16;
        return (es != ei && es != ej) ? 1u : 0u;
    });
    m.kind = 10;
    m.n = 3;
    m.el[0] = ci, m.el[1] = cj, m.el[2] = plf_uni(t.ssw[s]);
    m.a = ei, m.ap = ci & 0xFFFFu, m.b = ej, m.bp = cj & 0xFFFFu, m.ext = 0;
}
// multi_critical_ruin_sources (support.rs:139-161) + merged_ruin_sources (move/list_kernel/ruin.rs:34-54)
__device__ __noinline__ void plf_decode_multi_ruin(const PlfRep& t, uint32_t idx, PlfMove& m) {
    const uint32_t i = plf_find_row(t.mrrow, t.nb, idx);
    uint32_t off = idx - plf_uni(t.mrrow[i]);
    const uint32_t li = plf_uni(t.blk[2 * i + 1]) >> 16;
    const uint32_t j = plf_walk(i + 1, t.nb, off, [&](uint32_t jj) { return li * (t.blk[2 * jj + 1] >> 16); });
    const uint32_t lj = plf_uni(t.blk[2 * j + 1]) >> 16;
    uint32_t x = plf_uni(t.blk[2 * i]) + off / lj, y = plf_uni(t.blk[2 * j]) + off % lj;  // (list << 16 | start) + offset inside the block
    if (x > y) {
        const uint32_t s = x;
        x = y, y = s;
    }
    m.kind = 8;
    m.n = 2;
    m.el[0] = x, m.el[1] = y;
    m.a = x >> 16, m.ap = 2, m.b = y >> 16, m.bp = 0, m.ext = 0;
}

struct PlfBlock {
    uint32_t e, start, len, rl;
    __device__ __forceinline__ uint32_t adjacent() const { return len - 1; }
    __device__ __forceinline__ uint32_t change() const { return len * (rl - 1); }
    __device__ __forceinline__ uint32_t boundary() const { return len == 1 ? rl - 1 : 2 * rl - 3; }  // count_boundary_change_moves (:328-340)
    __device__ __forceinline__ uint32_t pairs() const { return len * (len - 1) / 2; }
    __device__ __forceinline__ uint32_t sublist_swap() const {  // count_adjacent_sublist_swap_moves_for_len (:405-427), closed form for max size 3
        if (len < 3) return 0;
        auto pos = [](uint32_t a, uint32_t b) { return a > b ? a - b : 0u; };
        return 2 * pos(len, 2) + 3 * pos(len, 3) + 2 * pos(len, 4) + pos(len, 5);
    }
    __device__ __forceinline__ uint32_t ruin() const { return len < 2 ? 0 : len - (len < PLF_RUIN_MAX ? len : PLF_RUIN_MAX) + 1; }
    __device__ __forceinline__ uint32_t sublist_change() const {  // (:444-456)
        if (len < 2 || rl < 2) return 0;
        uint32_t mx = PLF_SUBLIST_MAX < len ? PLF_SUBLIST_MAX : len, c = 0;
        mx = mx < rl ? mx : rl;
        for (uint32_t z = 2; z <= mx; ++z) c += (len - z + 1) * (rl - z);
        return c;
    }
    __device__ __forceinline__ uint32_t permute() const {  // (:429-442)
        if (len < 2) return 0;
        const uint32_t mw = PLF_PERMUTE_MAX < len ? PLF_PERMUTE_MAX : len;
        uint32_t c = 0;
        for (uint32_t s = 0; s < len; ++s) {
            const uint32_t mv = mw < len - s ? mw : len - s;
            uint32_t f = 1;
            for (uint32_t z = 2; z <= mv; ++z) {
                f *= z;
                c += f - 1;
            }
        }
        return c;
    }
    __device__ __forceinline__ uint32_t moves() const { return change() + 2 * pairs() + sublist_swap() + ruin() + sublist_change() + permute(); }
};
__device__ __forceinline__ PlfBlock plf_block(const PlfRep& t, uint32_t bi) {
    const uint32_t w0 = plf_uni(t.blk[2 * bi]), w1 = plf_uni(t.blk[2 * bi + 1]);
    return PlfBlock{w0 >> 16, w0 & 0xFFFFu, w1 >> 16, w1 & 0xFFFFu};
}
// the d-th destination in 0..=rl that is none of source, source + 1 and (when the source is not the block's last element) source + 2
__device__ __forceinline__ uint32_t plf_nth_dest(uint32_t d, uint32_t source, bool skip_two) {
    uint32_t dest = d;
    if (dest >= source) ++dest;
    if (dest >= source + 1) ++dest;
    if (skip_two && dest >= source + 2) ++dest;
    return dest;
}
// block-local move index -> move (cursor.rs:83-179 over coordinates.rs:132-252)
__device__ __noinline__ void plf_decode_block(const PlfBlock& bl_ref, uint32_t idx, PlfMove& m) {
    const PlfBlock bl = bl_ref;
    m.a = m.b = bl.e;
    m.ext = 0;
    m.n = 0;
    const uint32_t adjacent = bl.adjacent();
    if (idx < adjacent) {
        m.kind = 2, m.ap = bl.start + idx, m.bp = bl.start + idx + 2;
        return;
    }
    if (idx < bl.change()) {  // non_adjacent_change (:132-143): the boundary sources, then the interior ones
        uint32_t o = idx - adjacent;
        const uint32_t boundary = bl.boundary();
        uint32_t so, d;
        if (o < boundary) {
            const uint32_t first = bl.len == 1 ? bl.rl - 1 : bl.rl - 2;
            so = o < first ? 0 : bl.len - 1;
            d = o < first ? o : o - first;
        } else {
            o -= boundary;
            so = 1 + o / (bl.rl - 2);
            d = o % (bl.rl - 2);
        }
        m.kind = 2, m.ap = bl.start + so, m.bp = plf_nth_dest(d, bl.start + so, so + 1 < bl.len);
        return;
    }
    uint32_t o = idx - bl.change();
    if (o < 2 * bl.pairs()) {  // critical_swap / critical_reverse (:145-169): pairs first < second
        const bool rev = o >= bl.pairs();
        if (rev) o -= bl.pairs();
        uint32_t f = 0;
        while (o >= bl.len - 1 - f) {
            o -= bl.len - 1 - f;
            ++f;
        }
        m.kind = rev ? 4 : 3, m.ap = bl.start + f, m.bp = bl.start + f + 1 + o + (rev ? 1u : 0u);
        return;
    }
    o -= 2 * bl.pairs();
    if (o < bl.sublist_swap()) {  // critical_adjacent_sublist_swap (:171-204)
        const uint32_t mx = PLF_SUBLIST_MAX < bl.len ? PLF_SUBLIST_MAX : bl.len;
        for (uint32_t s = 0; s < bl.len; ++s)
            for (uint32_t fs = 1; fs <= mx && s + fs < bl.len; ++fs)
                for (uint32_t ss = 1; ss <= mx; ++ss) {
                    if ((fs == 1 && ss == 1) || s + fs + ss > bl.len) continue;
                    if (o == 0) {
                        m.kind = 6, m.ap = bl.start + s, m.bp = bl.start + s + fs, m.ext = fs | (ss << 16);
                        return;
                    }
                    --o;
                }
    }
    o -= bl.sublist_swap();
    if (o < bl.ruin()) {  // critical_ruin_indices (:190-204)
        const uint32_t w = bl.len < PLF_RUIN_MAX ? bl.len : PLF_RUIN_MAX;
        m.kind = 8, m.n = w, m.ap = w, m.bp = 0;
        for (uint32_t k = 0; k < PLF_RUIN_MAX; ++k) m.el[k] = (bl.e << 16) | (bl.start + o + k);
        return;
    }
    o -= bl.ruin();
    if (o < bl.sublist_change()) {  // critical_sublist_change (:206-228): destinations in post-removal coordinates, never the segment's own start
        uint32_t mx = PLF_SUBLIST_MAX < bl.len ? PLF_SUBLIST_MAX : bl.len;
        mx = mx < bl.rl ? mx : bl.rl;
        for (uint32_t z = 2; z <= mx; ++z) {
            const uint32_t cnt = (bl.len - z + 1) * (bl.rl - z);
            if (o >= cnt) {
                o -= cnt;
                continue;
            }
            const uint32_t ss = o / (bl.rl - z), d = o % (bl.rl - z);
            m.kind = 5, m.ap = bl.start + ss, m.ext = bl.start + ss + z, m.bp = d < bl.start + ss ? d : d + 1;
            return;
        }
    }
    o -= bl.sublist_change();
    {  // critical_permutation (:230-252): (start, size, rank = offset + 1)
        const uint32_t mw = PLF_PERMUTE_MAX < bl.len ? PLF_PERMUTE_MAX : bl.len;
        for (uint32_t s = 0; s < bl.len; ++s) {
            const uint32_t mv = mw < bl.len - s ? mw : bl.len - s;
            uint32_t f = 1;
            for (uint32_t z = 2; z <= mv; ++z) {
                f *= z;
                if (o < f - 1) {
                    m.kind = 9, m.ap = bl.start + s, m.bp = bl.start + s + z, m.ext = o + 1;
                    return;
                }
                o -= f - 1;
            }
        }
    }
    m.kind = 0;  // not reached for idx < moves()
}
// tiered_precedence_move_index (coordinates.rs:91-130)
__device__ __forceinline__ uint32_t plf_tiered_index(const StreamCtx& ctx, const PlfBlock& bl, uint32_t offset, uint64_t salt) {
    const uint32_t adjacent = bl.adjacent();
    if (offset < adjacent) return ctx.selection_index(offset, adjacent, salt ^ SALT_PL_ADJACENT);
    const uint32_t boundary = bl.boundary();
    if (offset < adjacent + boundary) return adjacent + ctx.selection_index(offset - adjacent, boundary, salt ^ SALT_PL_BOUNDARY);
    return adjacent + boundary + ctx.selection_index(offset - adjacent - boundary, bl.moves() - adjacent - boundary, salt ^ SALT_PL_REST);
}
// wire format of a decoded move (sf_move_t: kind, a, a_pos, b, b_pos, value), include/solverforge_amd.h
__device__ __forceinline__ void plf_wire(const PlfMove& m, int32_t* w) {
    w[0] = m.kind;
    if (m.kind == 10) {  // a = swaps, (list | first << 16) x 3, value = (second - first) per swap, one byte each
        w[1] = 3;
        w[2] = (int32_t)((m.el[0] >> 16) | ((m.el[0] & 0xFFFFu) << 16));
        w[3] = (int32_t)((m.el[1] >> 16) | ((m.el[1] & 0xFFFFu) << 16));
        w[4] = (int32_t)((m.el[2] >> 16) | ((m.el[2] & 0xFFFFu) << 16));
        w[5] = 0x010101;
    } else if (m.kind == 8) {  // a = first list, a_pos = count, five 16-bit positions; value bits 16..31: 0x8000 precedence hooks, 0x4000 second list in bits 16..29
        const uint32_t second = m.el[m.n - 1] >> 16;
        const bool multi = second != (m.el[0] >> 16);
        uint32_t ps[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < m.n; ++k) ps[k] = m.el[k] & 0xFFFFu;
        w[1] = (int32_t)(m.el[0] >> 16);
        w[2] = (int32_t)m.n;
        w[3] = (int32_t)(ps[0] | (ps[1] << 16));
        w[4] = (int32_t)(ps[2] | (ps[3] << 16));
        w[5] = (int32_t)(ps[4] | ((0x8000u | (multi ? (0x4000u | second) : 0u)) << 16));
    } else {
        w[1] = (int32_t)m.a, w[2] = (int32_t)m.ap, w[3] = (int32_t)m.b, w[4] = (int32_t)m.bp;
        w[5] = m.kind == 5 ? (int32_t)m.ext : (m.kind == 6 ? (int32_t)m.ext : (m.kind == 9 ? (int32_t)m.ext : -1));
    }
}

// ---- list edits of the precedence-aware recreate (one wavefront, lists in LDS) ----
__device__ __forceinline__ void plf_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// remove the element at (list e, position pos); returns it
__device__ __forceinline__ uint32_t plf_list_remove(uint16_t* visits, uint32_t* off, int V, uint32_t e, uint32_t pos) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = plf_uni(off[e]) + pos, total = plf_uni(off[V]);
    const uint32_t x = plf_uni((uint32_t)visits[g]);
    for (uint32_t t0 = g; t0 + 1 < total; t0 += 64) {  // ascending chunks: the reads run ahead of the writes
        const uint32_t t = t0 + lane;
        const uint32_t nv = t + 1 < total ? (uint32_t)visits[t + 1] : 0u;
        plf_sync();
        if (t + 1 < total) visits[t] = (uint16_t)nv;
        plf_sync();
    }
    for (uint32_t r = e + 1 + lane; r <= (uint32_t)V; r += 64) off[r] -= 1;
    plf_sync();
    return x;
}
__device__ __forceinline__ void plf_list_insert(uint16_t* visits, uint32_t* off, int V, uint32_t e, uint32_t pos, uint32_t x) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = plf_uni(off[e]) + pos, total = plf_uni(off[V]);
    for (uint32_t c0 = 0; c0 < total - g; c0 += 64) {  // descending chunks: [g, total) -> [g + 1, total]
        const uint32_t dd = c0 + lane;
        const bool in = dd < total - g;
        const uint32_t t = total - (in ? dd : 0u);
        const uint32_t nv = in ? (uint32_t)visits[t - 1] : 0u;
        plf_sync();
        if (in) visits[t] = (uint16_t)nv;
        plf_sync();
    }
    if (lane == 0) visits[g] = (uint16_t)x;
    for (uint32_t r = e + 1 + lane; r <= (uint32_t)V; r += 64) off[r] += 1;
    plf_sync();
}

}  // namespace sf
