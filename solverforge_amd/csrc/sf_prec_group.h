// Grouped trial evaluation of the ListPrecedenceMakespanConstraint (round 4): T candidates per wavefront, G = 64 / T lanes each.
//
// A job-shop precedence graph is narrow -- a Kahn round of prec_eval (sf_precedence.h) pops a handful of ready nodes onto 64 lanes, and the
// rounds are a chain of dependent LDS round trips -- so one wave-wide evaluation per candidate leaves the wave mostly idle.  Here T trials
// walk their rounds side by side, each on its own lane group with private scratch (earliest start, in-degree, queue, list successor: 12 bytes
// per node and trial, in the replica's LDS slice).  Nothing is applied to the replica's lists: a trial's list edges are written from the
// committed lists through the candidate's POSITION MAP -- old_index(list, new position) -> position in the committed lists -- for the at most
// three lists the candidate touches; every other node keeps the committed list successor and in-degree (copied from two shared arrays built
// once per step).  The evaluation itself is Kahn's algorithm exactly as in prec_eval, so cyclic trials are detected the same way (nodes left
// unprocessed) and a cyclic COMMITTED state needs no special case.
//
// Result per trial: the constraint's (hard penalty, makespan) of the lists after the move -- what prec_eval returns for the applied state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_precedence.h"

namespace sf {

typedef __attribute__((address_space(3))) uint16_t pg_lds_u16;
typedef __attribute__((address_space(3))) uint32_t pg_lds_u32;
typedef __attribute__((address_space(3))) int32_t pg_lds_i32;

constexpr uint32_t PG_NONE16 = 0xFFFFu;

// LDS of the grouped evaluator inside one replica's slice: committed list successor + in-degree (shared by the trials), then per trial
// E (i32), D (i32), S (u16), Q (u16).  E and D carry one extra word per lane of the group: the "no successor" target of a relaxation, so that
// both relaxations of a Kahn round are issued without a branch (a lane's own dummy: no same-address serialisation).
__host__ __device__ inline size_t pgrp_a4(int n, int trials) { return ((size_t)(n + 64 / (trials > 0 ? trials : 1)) * 4 + 15) / 16 * 16; }
// shared by the trials of a replica: committed list successor, committed in-degree, the nodes whose committed in-degree is 0, the fixed
// in-degree of the element at every flat list position (u16 x n each) and the wrong-owner items of every committed list (u32 x V)
__host__ __device__ inline size_t pgrp_shared_bytes(int n, int V) { return 4 * (((size_t)n * 2 + 15) / 16 * 16) + ((size_t)V * 4 + 15) / 16 * 16; }
__host__ __device__ inline size_t pgrp_bytes(int n, int trials, int V) {
    if (trials <= 0) return 0;
    const size_t a2 = ((size_t)n * 2 + 15) / 16 * 16;
    return pgrp_shared_bytes(n, V) + (2 * pgrp_a4(n, trials) + 2 * a2) * (size_t)trials;
}
struct PgrpLds {
    pg_lds_u16* Sc;  // [n] committed list successor (PG_NONE16 = none)
    pg_lds_u16* Dc;  // [n] committed in-degree (fixed + list predecessor)
    pg_lds_u16* Rc;  // the nodes with Dc == 0 (a trial's ready set is these, re-checked, plus the heads of the lists it touches)
    pg_lds_u16* Iv;  // [flat position] fixed in-degree of the element there (a trial reads it beside the element, not behind it)
    pg_lds_u32* Vl;  // [V] wrong-owner items per committed list
    pg_lds_i32* E;   // this lane's trial
    pg_lds_i32* D;
    pg_lds_u16* S;
    pg_lds_u16* Q;
    __device__ PgrpLds(unsigned char* base, int n, int V, uint32_t trial, int trials) {
        const size_t a4 = pgrp_a4(n, trials), a2 = ((size_t)n * 2 + 15) / 16 * 16;
        Sc = (pg_lds_u16*)base;
        Dc = (pg_lds_u16*)(base + a2);
        Rc = (pg_lds_u16*)(base + 2 * a2);
        Iv = (pg_lds_u16*)(base + 3 * a2);
        Vl = (pg_lds_u32*)(base + 4 * a2);
        unsigned char* t = base + pgrp_shared_bytes(n, V) + (size_t)trial * (2 * a4 + 2 * a2);
        E = (pg_lds_i32*)t;
        D = (pg_lds_i32*)(t + a4);
        S = (pg_lds_u16*)(t + 2 * a4);
        Q = (pg_lds_u16*)(t + 2 * a4 + a2);
    }
};

// the shared arrays, once per step: one wavefront.  Returns the wrong-owner items of the lists and the number of committed-ready nodes (wave-uniform).
template <class VT>
__device__ __forceinline__ void pgrp_build_committed(const PrecModel& pm, const PREC_L VT* visits, const PREC_L uint32_t* off, int V, const PgrpLds& L, uint32_t& out_viol,
                                                     uint32_t& out_ready) {
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)pm.n;
    for (uint32_t i = lane; i < n; i += 64) {
        L.Sc[i] = (uint16_t)PG_NONE16;
        L.Dc[i] = (uint16_t)pm.indeg0[i];
    }
    for (uint32_t e = lane; e < (uint32_t)V; e += 64) L.Vl[e] = 0u;
    prec_sync();
    uint32_t viol = 0;
    for (uint32_t e = 0; e < (uint32_t)V; ++e) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[e]), len = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[e + 1]) - o;
        uint32_t mine = 0;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t x = (uint32_t)visits[o + k];
            if (k + 1 < len) L.Sc[x] = (uint16_t)visits[o + k + 1];
            const uint32_t i0 = (uint32_t)pm.indeg0[x];
            L.Iv[o + k] = (uint16_t)i0;
            if (k > 0) L.Dc[x] = (uint16_t)(i0 + 1);
            if (pm.owner) {
                const int32_t ow = pm.owner[x];
                mine += (ow >= 0 && (uint32_t)ow != e) ? 1u : 0u;
            }
        }
        if (mine) __hip_atomic_fetch_add(L.Vl + e, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        viol += mine;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) viol += (uint32_t)__shfl_xor((int)viol, o);
    prec_sync();
    uint32_t cnt = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
        const uint32_t v = b0 + lane;
        const bool ready = v < n && L.Dc[v] == 0;
        const uint64_t m = __ballot(ready);
        if (ready) L.Rc[cnt + prec_mbcnt(m)] = (uint16_t)v;
        cnt += (uint32_t)__popcll(m);
    }
    prec_sync();
    out_viol = viol;
    out_ready = cnt;
}

// One candidate as the generic engine's ring holds it, decoded to the arguments of apply_list_move_wave
struct PgrpMove {
    uint32_t kind;  // sf_move_kind: 2 change, 3 swap, 4 reverse, 5 sublist change, 6 sublist swap, 7 3-opt, 9 permute, 10 multi-swap; 0 = idle group
    uint32_t a, i, b, j, ext;
    uint32_t el2;   // multi-swap: the third swap (list << 16 | first position); a / i and b / j are the first two
};

// Position map of a candidate, list by list: the list e AFTER the move is a concatenation of at most five runs of the committed flat
// array -- [end[s-1], end[s]) of the new list reads src[s] + offset (src[s] - offset when bit s of `rev` is set; the permuted window of a
// list permute, always run 1, reads src[1] + its nibble).  Built once per (trial, list); a position costs five compares and selects.
struct PgrpSegs {
    uint32_t end[5], src[5];
    uint32_t rev, perm, is_perm;
    uint32_t len;  // length of list e after the move
};
// la / lb = committed lengths of lists a / b.  e is one of the lists the move touches.
template <class OffP>
__device__ __forceinline__ void pgrp_segments(const PgrpMove& m, OffP off, uint32_t e, uint32_t la, uint32_t lb, PgrpSegs& g) {
    const bool one_list = m.kind == 4 || m.kind == 7 || m.kind == 9;  // b / j carry cut positions, not a second list
    const uint32_t oa = off[m.a], ob = one_list ? oa : off[m.b];
    const uint32_t a = m.a, b = m.b, i = m.i, j = m.j;
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, s0 = oa, s1 = 0, s2 = 0, s3 = 0, s4 = 0, len = la, rev = 0, perm = 0, is_perm = 0;
    bool four = false;  // the fifth run is in use (else it is empty)
    switch (m.kind) {
        case 2:  // (a, i) -> (b, j), j in pre-removal coordinates
            if (a != b) {
                if (e == a) {
                    len = la - 1;
                    e0 = i, e1 = len, s1 = oa + i + 1, e2 = e3 = len;
                } else {
                    len = lb + 1;
                    s0 = ob, e0 = j, e1 = j + 1, s1 = oa + i, e2 = len, s2 = ob + j, e3 = len;
                }
            } else {
                const uint32_t f = j > i ? j - 1 : j;  // final position of the element
                if (f >= i)
                    e0 = i, e1 = f, s1 = oa + i + 1, e2 = f + 1, s2 = oa + i, e3 = la, s3 = oa + f + 1;
                else
                    e0 = f, e1 = f + 1, s1 = oa + i, e2 = i + 1, s2 = oa + f, e3 = la, s3 = oa + i + 1;
            }
            break;
        case 3:  // (a, i) <-> (b, j)
            if (a != b) {
                if (e == a)
                    e0 = i, e1 = i + 1, s1 = ob + j, e2 = e3 = la, s2 = oa + i + 1;
                else
                    len = lb, s0 = ob, e0 = j, e1 = j + 1, s1 = oa + i, e2 = e3 = lb, s2 = ob + j + 1;
            } else {
                const uint32_t p = i < j ? i : j, q = i < j ? j : i;
                e0 = p, e1 = p + 1, s1 = oa + q, e2 = q, s2 = oa + p + 1, e3 = q + 1, s3 = oa + p, s4 = oa + q + 1, four = true;
            }
            break;
        case 4:  // reverse [i, j) of a
            e0 = i, e1 = j, s1 = oa + j - 1, rev = 2u, e2 = e3 = la, s2 = oa + j;
            break;
        case 5: {  // [i, ext) of a -> b at j (intra: j in post-removal coordinates)
            const uint32_t z = m.ext - i;
            if (a != b) {
                if (e == a) {
                    len = la - z;
                    e0 = i, e1 = e2 = e3 = len, s1 = oa + i + z;
                } else {
                    len = lb + z;
                    s0 = ob, e0 = j, e1 = j + z, s1 = oa + i, e2 = e3 = len, s2 = ob + j;
                }
            } else if (j <= i)
                e0 = j, e1 = j + z, s1 = oa + i, e2 = i + z, s2 = oa + j, e3 = la, s3 = oa + i + z;
            else
                e0 = i, e1 = j, s1 = oa + i + z, e2 = j + z, s2 = oa + i, e3 = la, s3 = oa + j + z;
            break;
        }
        case 6: {  // [i, i + za) of a <-> [j, j + zb) of b
            const uint32_t za = m.ext & 0xFFFFu, zb = m.ext >> 16;
            if (a != b) {
                if (e == a) {
                    len = la - za + zb;
                    e0 = i, e1 = i + zb, s1 = ob + j, e2 = e3 = len, s2 = oa + i + za;
                } else {
                    len = lb - zb + za;
                    s0 = ob, e0 = j, e1 = j + za, s1 = oa + i, e2 = e3 = len, s2 = ob + j + zb;
                }
            } else {  // one list: X = the earlier segment, Y = the later one; new layout from x0: Y, the elements between them, X
                const bool a_first = i < j;
                const uint32_t x0 = a_first ? i : j, zx = a_first ? za : zb, y0 = a_first ? j : i, zy = a_first ? zb : za;
                const uint32_t mid = y0 - (x0 + zx);
                e0 = x0, e1 = x0 + zy, s1 = oa + y0, e2 = x0 + zy + mid, s2 = oa + x0 + zx, e3 = y0 + zy, s3 = oa + x0, s4 = oa + y0 + zy, four = true;
            }
            break;
        }
        case 7: {  // 3-opt of list a: cuts c1 = i < c2 = b < c3 = j, pattern ext
            const uint32_t c1 = i, c2 = m.b, c3 = j;
            const uint32_t mask = kopt_reverse_mask(m.ext);
            const bool rb = (mask >> 1) & 1u, rc = (mask >> 2) & 1u;
            e0 = c1, e2 = c3, e3 = la, s3 = oa + c3;
            if (!kopt_swaps_segments(m.ext)) {
                e1 = c2;
                s1 = rb ? oa + c2 - 1 : oa + c1, s2 = rc ? oa + c3 - 1 : oa + c2;
                rev = (rb ? 2u : 0u) | (rc ? 4u : 0u);
            } else {  // the second segment comes first
                e1 = c1 + (c3 - c2);
                s1 = rc ? oa + c3 - 1 : oa + c2, s2 = rb ? oa + c2 - 1 : oa + c1;
                rev = (rc ? 2u : 0u) | (rb ? 4u : 0u);
            }
            break;
        }
        case 9:  // window [i, j) of a permuted by the ext-th permutation
            e0 = i, e1 = j, s1 = oa + i, perm = nth_permutation_nibbles(j - i, m.ext), is_perm = 1u, e2 = e3 = la, s2 = oa + j;
            break;
        default: {  // 10: three adjacent swaps in three different lists
            const uint32_t p = e == a ? i : (e == b ? j : (m.el2 & 0xFFFFu));
            const uint32_t oe = off[e];
            len = off[e + 1] - oe;
            s0 = oe, e0 = p, e1 = p + 1, s1 = oe + p + 1, e2 = p + 2, s2 = oe + p, e3 = len, s3 = oe + p + 2;
            break;
        }
    }
    g.end[0] = e0, g.end[1] = e1, g.end[2] = e2, g.end[3] = e3, g.end[4] = four ? len : e3;
    g.src[0] = s0, g.src[1] = s1, g.src[2] = s2, g.src[3] = s3, g.src[4] = s4;
    g.rev = rev, g.perm = perm, g.is_perm = is_perm, g.len = four ? len : e3;
}
// flat index (into the committed `visits`) of the element at position k of the list after the move
__device__ __forceinline__ uint32_t pgrp_seg_src(const PgrpSegs& g, uint32_t k) {
    uint32_t start = 0, res = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const uint32_t o = k - start;
        uint32_t v = ((g.rev >> s) & 1u) ? g.src[s] - o : g.src[s] + o;
        if (s == 1) v = g.is_perm ? g.src[1] + ((g.perm >> (4u * (o & 7u))) & 15u) : v;
        res = (k >= start && k < g.end[s]) ? v : res;
        start = g.end[s];
    }
    return res;
}

// The constraint's static graph as the grouped evaluator reads it: typed LDS pointers into the workgroup-shared copy (sf_mixed_wave.hip).
struct PgrpStatic {
    const pg_lds_u32* nd;        // [n][2]: duration; min(fixed out-degree, 255) << 24 | first fixed successor (0xFFFFFF = none) -- PrecModel::nd
    const pg_lds_u32* succ_off;  // the further fixed successors of a node with more than one
    const pg_lds_u32* succ;
    const pg_lds_i32* indeg0;
    const pg_lds_i32* owner;
    uint32_t has_owner;          // expected-owner hook (an explicit flag: the null value of an LDS pointer is not 0)
};

#ifdef SF_PHASE_PGRP  // diagnostics (-DSF_PHASE_PROFILE -DSF_PHASE_PGRP): shader clocks per stage of a pass in g_phase[0..4], passes in [7]
extern __device__ unsigned long long g_phase[8];
#define PGT(i)                                                                     \
    {                                                                              \
        const uint64_t _t = clock64();                                             \
        if (lane == 0) atomicAdd(&g_phase[i], (unsigned long long)(_t - pg_t));    \
        pg_t = _t;                                                                 \
    }
#else
#define PGT(i)
#endif

// T trials side by side.  `gshift` = log2(lanes per group); lane's group g = lane >> gshift, its index inside the group lg.  `mv` is uniform
// inside a group (kind 0 = the group idles).  Returns, uniform inside each group, the (penalty, makespan, cycle flag) of the trial.
//   fixed_pen = const_penalty + unassigned nodes (unchanged by a list move that keeps every element), viol_c = wrong-owner items of the
//   committed lists.
// A Kahn round is four dependent LDS round trips: the popped node; its record, earliest start and list successor; the relaxations of the
// first fixed successor and of the list successor (issued together); the queue writes.  Pop order is free (the result does not depend on it).
template <class VT>
__device__ __noinline__ void prec_eval_grouped(const PgrpStatic& ps_ref, uint32_t n, int V, const PREC_L VT* visits, const PREC_L uint32_t* off, unsigned char* lds_base, uint32_t gshift,
                                               const PgrpMove mv, int64_t fixed_pen, uint32_t viol_c, uint32_t n_ready, int64_t& out_pen, int64_t& out_mk, bool& out_cyclic) {
    const PgrpStatic ps = ps_ref;  // (a private copy the optimizer can keep in registers: through the reference every field is re-read after each store)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t G = 1u << gshift, lg = lane & (G - 1u), g = lane >> gshift;
    const uint64_t gmask = (G >= 64u ? ~0ull : ((1ull << G) - 1ull)) << (g << gshift);  // this group's lanes
    const uint64_t below = (1ull << lane) - 1ull;
    const PgrpLds L(lds_base, (int)n, V, g, (int)(64u >> gshift));
    const bool active = mv.kind != 0;
#ifdef SF_PHASE_PGRP
    uint64_t pg_t = clock64();
    if (lane == 0) atomicAdd(&g_phase[7], 1ull);
#endif
    // ---- the trial's list edges and in-degrees: the committed ones, then the lists the move touches through its position map ----
    if (active) {  // two nodes per lane and iteration (the u16 arrays as 32-bit words), iterations unrolled so that their reads overlap
        const pg_lds_u32* const Sc2 = (const pg_lds_u32*)L.Sc;
        const pg_lds_u32* const Dc2 = (const pg_lds_u32*)L.Dc;
        pg_lds_u32* const S2 = (pg_lds_u32*)L.S;
        const uint32_t words = (n + 1u) >> 1;  // (the arrays are padded to 16 bytes: the odd tail word exists)
#pragma unroll 4
        for (uint32_t w = lg; w < words; w += G) {
            const uint32_t sc = Sc2[w], dc = Dc2[w];
            S2[w] = sc;
            L.D[2 * w] = (int32_t)(dc & 0xFFFFu);
            L.E[2 * w] = 0;
            if (2 * w + 1 < n) {
                L.D[2 * w + 1] = (int32_t)(dc >> 16);
                L.E[2 * w + 1] = 0;
            }
        }
    }
    prec_sync();
    PGT(0)
    int32_t dviol = 0;
    uint32_t nl = 0, len0 = 0, len1 = 0, len2 = 0;  // the lists the move touches and their lengths afterwards
    uint32_t hd0 = 0, hd1 = 0, hd2 = 0;             // their first elements (on the group's first lane)
    if (active) {  // the touched lists: list successor and in-degree of their elements in the new order
        const bool one_list = mv.kind == 4 || mv.kind == 7 || mv.kind == 9;
        const uint32_t la = off[mv.a + 1] - off[mv.a], lb = one_list ? la : off[mv.b + 1] - off[mv.b];
        const uint32_t e3 = mv.el2 >> 16;
        nl = mv.kind == 10 ? 3u : ((mv.kind == 2 || mv.kind == 3 || mv.kind == 5 || mv.kind == 6) && mv.a != mv.b ? 2u : 1u);
        for (uint32_t li = 0; li < nl; ++li) {
            const uint32_t e = li == 0 ? mv.a : (li == 1 ? mv.b : e3);
            PgrpSegs sg;
            pgrp_segments(mv, off, e, la, lb, sg);
            // G - 1 elements per iteration: the group's last lane only supplies the successor of the one before it (the element itself
            // belongs to the next iteration's first lane), so the position map is evaluated once per lane and the successor is a lane shift
            for (uint32_t k0 = 0; k0 < sg.len; k0 += G - 1u) {
                const uint32_t k = k0 + lg;
                const bool in = k < sg.len;
                const uint32_t src = in ? pgrp_seg_src(sg, k) : 0u;
                const uint32_t x = in ? (uint32_t)visits[src] : 0u, i0 = in ? (uint32_t)L.Iv[src] : 0u;
                const uint32_t nx = (uint32_t)__shfl_down((int)x, 1);  // (inside the group for lg < G - 1)
                if (in && lg < G - 1u) {
                    L.S[x] = (uint16_t)(k + 1 < sg.len ? nx : PG_NONE16);
                    L.D[x] = (int32_t)(i0 + (k > 0 ? 1u : 0u));
                    if (k == 0) hd0 = li == 0 ? x : hd0, hd1 = li == 1 ? x : hd1, hd2 = li == 2 ? x : hd2;  // (lane lg == 0)
                    if (ps.has_owner) {
                        const int32_t o = ps.owner[x];
                        dviol += (o >= 0 && (uint32_t)o != e) ? 1 : 0;
                    }
                }
            }
            if (ps.has_owner && lg == 0) dviol -= (int32_t)L.Vl[e];  // minus what the committed list e contributed
            len0 = li == 0 ? sg.len : len0, len1 = li == 1 ? sg.len : len1, len2 = li == 2 ? sg.len : len2;
        }
    }
    if (ps.has_owner) {  // group sum of dviol (xor butterfly stays inside the group for offsets < G)
        for (uint32_t o = G >> 1; o; o >>= 1) dviol += __shfl_xor(dviol, (int)o);
    }
    prec_sync();
    PGT(1)
    // ---- Kahn: ready nodes, then rounds of up to G pops per group ----
    // ready set: the committed-ready nodes that still have no predecessor, plus the heads of the touched lists that lost theirs (a node
    // that is not a list head keeps a list predecessor; the committed-ready ones are covered by the first loop)
    uint32_t head = 0, tail = 0;
    for (uint32_t b0 = 0; b0 < n_ready; b0 += G) {  // (n_ready is wave-uniform: every group scans the same chunks)
        const uint32_t idx = b0 + lg;
        const uint32_t v = idx < n_ready ? (uint32_t)L.Rc[idx] : 0u;
        const bool ready = active && idx < n_ready && L.D[v] == 0;
        const uint64_t m = __ballot(ready) & gmask;
        if (ready) L.Q[tail + (uint32_t)__popcll(m & below)] = (uint16_t)v;
        tail += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (uint32_t li = 0; li < 3; ++li) {
        const uint32_t h = li == 0 ? hd0 : (li == 1 ? hd1 : hd2), len = li == 0 ? len0 : (li == 1 ? len1 : len2);
        const bool ready = active && lg == 0 && li < nl && len > 0 && L.D[h] == 0 && L.Dc[h] != 0;
        const uint64_t m = __ballot(ready) & gmask;
        if (ready) L.Q[tail] = (uint16_t)h;
        tail += (uint32_t)__popcll(m);
    }
    prec_sync();
    PGT(2)
    int32_t mk = 0;
    uint32_t pg_rounds = 0;
    while (__ballot(head < tail) != 0ull) {
        pg_rounds += 1;
        const uint32_t cnt = tail - head < G ? tail - head : G;
        const bool act = lg < cnt;
        int32_t fin = 0;
        uint32_t node = 0, deg = 0, s1 = PG_NONE16, s2 = PG_NONE16;
        if (act) {
            node = (uint32_t)L.Q[head + lg];
            const uint32_t r0 = ps.nd[2 * node], r1 = ps.nd[2 * node + 1];
            fin = L.E[node] + (int32_t)r0;
            s2 = (uint32_t)L.S[node];
            mk = fin > mk ? fin : mk;
            deg = r1 >> 24;
            s1 = (r1 & 0xFFFFFFu) == 0xFFFFFFu ? PG_NONE16 : (r1 & 0xFFFFFFu);
        }
        // both relaxations in flight before the first wait: a missing successor relaxes the lane's dummy word behind the arrays
        const uint32_t t1 = s1 != PG_NONE16 ? s1 : n + lg, t2 = s2 != PG_NONE16 ? s2 : n + lg;
        __hip_atomic_fetch_max(L.E + t1, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int32_t o1 = __hip_atomic_fetch_add(L.D + t1, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(L.E + t2, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int32_t o2 = __hip_atomic_fetch_add(L.D + t2, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool new1 = s1 != PG_NONE16 && o1 == 1, new2 = s2 != PG_NONE16 && o2 == 1;
        const uint64_t m1 = __ballot(new1) & gmask, m2 = __ballot(new2) & gmask;
        uint32_t ntail = tail;
        if (new1) L.Q[ntail + (uint32_t)__popcll(m1 & below)] = (uint16_t)s1;
        ntail += (uint32_t)__popcll(m1);
        if (new2) L.Q[ntail + (uint32_t)__popcll(m2 & below)] = (uint16_t)s2;
        ntail += (uint32_t)__popcll(m2);
        if (__ballot(deg > 1u)) {  // further fixed successors (none in a job shop)
            uint32_t so = 0;
            if (deg > 1u) {
                so = ps.succ_off[node];
                deg = ps.succ_off[node + 1] - so;  // (the record saturates at 255)
            }
            for (uint32_t k = 1;; ++k) {
                const bool has = k < deg;
                if (!__ballot(has)) break;
                bool newly = false;
                uint32_t s = 0;
                if (has) {
                    s = ps.succ[so + k];
                    __hip_atomic_fetch_max(L.E + s, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    newly = __hip_atomic_fetch_add(L.D + s, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 1;
                }
                const uint64_t m = __ballot(newly) & gmask;
                if (newly) L.Q[ntail + (uint32_t)__popcll(m & below)] = (uint16_t)s;
                ntail += (uint32_t)__popcll(m);
            }
        }
        head += cnt;
        tail = ntail;
        prec_sync();
    }
    PGT(3)
#ifdef SF_PHASE_PGRP
    if (lane == 0) atomicAdd(&g_phase[6], (unsigned long long)pg_rounds);
#endif
    for (uint32_t o = G >> 1; o; o >>= 1) {  // group maximum of the finish times
        const int32_t other = __shfl_xor(mk, (int)o);
        mk = other > mk ? other : mk;
    }
    const bool cyclic = head < n;  // Kahn left nodes unprocessed (rebuild_graph_summary :584-588)
    out_pen = fixed_pen + (int64_t)((int32_t)viol_c + dviol) + (cyclic ? (int64_t)n : 0);
    out_mk = cyclic ? 0 : (int64_t)mk;
    out_cyclic = cyclic;
}

}  // namespace sf
