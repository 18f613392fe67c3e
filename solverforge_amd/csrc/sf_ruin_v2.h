// Trial evaluation of a list ruin candidate WITHOUT touching the lists (round 5): the recreate of sf_ruin.h restated for the case the
// default policy runs all day -- a symmetric u16 matrix (RuinFast), 32-bit deltas (ListModel::small32), <= 128 lists.
//
// Same reference semantics as ruin_recreate_lds (heuristic/move/list_kernel/ruin.rs:131-281: remove, then greedy recreate -- every
// remaining element x every list x every position, the strictly best placed, first of equals in (element, list, position) order), same
// result bit for bit; sf_ruin.h stays the committed path and the fallback.  What differs is the formulation:
//
//  * NOTHING IS MOVED.  sf_ruin.h parks the removed elements at the end of their list and applies every placement as a list change on
//    the flat CSR (a shift of everything between the two lists, then the inverse for the undo).  Here the committed lists are read-only:
//    a list the candidate has changed (the source list without the removed elements, every list that took a placement) is a short copy
//    in a scratch arena, named by a per-list (base, length, in-arena) word that every scan reads instead of `off`.
//  * ONE FULL SCAN PER ELEMENT, not one per element and round.  A placement changes the insertion prices of ONE list (its slots and its
//    load).  The first scan of an element therefore keeps its best slot in each of its `cnt` best lists (cnt = elements removed); in
//    every later round only the lists changed so far are priced again (a dozen slots each) and compared with the best kept entry whose
//    list is still untouched.  That is exact: at most cnt - 1 lists change before the last round, so at least one of the cnt kept lists
//    is untouched, and every list outside the kept ones is untouched too and was no better than any of them.
//  * A LANE OWNS A LIST in the full scan (lists lane and lane + 64): the walk carries the previous element in a register, the capacity
//    term is per list, and a slot costs one list read, the row and edge reads and a compare -- a fifth of the instructions of the
//    slot-parallel scan, which re-derives list, bounds and neighbours of every slot.
//
// The edge table (RuinFast::edge: the leg entering every element) is patched in place for the seams the candidate creates and restored
// from a log at the end, exactly as sf_ruin.h does.  profiles/r05_phase7_*.txt: one recreate of sf_ruin.h costs ~250 K shader clocks of a
// CVRP-1000 replica (ten per step: 2.5 M of the 3.4 M clocks of a seven-leaf step).
#pragma once
#ifndef SF_RV2_U
#define SF_RV2_U 4
#endif
#include <stdint.h>

#include "sf_ruin.h"

namespace sf {

#ifdef SF_PHASE_PROFILE  // shader clocks per part of the v2 trial: 0 removal, 1 first scans, 2 best-lists extraction, 3 re-pricing, 4 pick + placement, 5 restore
__device__ unsigned long long g_rphase2[8];
#endif
#ifdef SF_RUIN_V2_CHECK  // diagnostic build: every trial is scored by both paths (scripts/ruin_v2_check.py)
__device__ unsigned long long g_rv2_check[8];  // 0 candidates, 1 mismatches, 2 fallbacks, 3.. first mismatch: replica, candidate, count, v2 soft, old soft
#endif

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {  // all lanes active; DPP row shifts + row broadcasts (cf. wave_max_i32)
    const int hi = (int)0xFFFFFFFFu;
    auto mn = [](uint32_t a, int b) { return a < (uint32_t)b ? a : (uint32_t)b; };
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x111, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x112, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x114, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x118, 0xf, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x142, 0xa, 0xf, false));
    v = mn(v, __builtin_amdgcn_update_dpp(hi, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}

// per-list word of the v2 scans: where the list's current elements are and how many
constexpr uint32_t RV2_ARENA = 0x80000000u;
__device__ __forceinline__ uint32_t rv2_word(uint32_t base, uint32_t len, bool arena) { return base | (len << 16) | (arena ? RV2_ARENA : 0u); }
// does the model fit the packing (16-bit bases, 15-bit lengths)?
__device__ __forceinline__ bool rv2_model_ok(const ListModel& lm) {
    return lm.V <= 128 && lm.n_cap <= 32767 && lm.dim <= 32767 && lm.small32 != 0 && lm.mat16 != nullptr;
}

// the table for the committed lists, once per step
__device__ __forceinline__ void rv2_build_words(const uint32_t* off, uint32_t V, uint32_t* words) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t e = lane; e < V; e += 64) words[e] = rv2_word(off[e], off[e + 1] - off[e], false);
    wave_sync();
}

// lexicographic a > b / a == b over L int32 levels (level 0 most significant)
template <int L>
__device__ __forceinline__ bool rv2_gt(const int32_t (&a)[L], const int32_t (&b)[L]) {
    bool gt = false, eq = true;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        gt = gt || (eq && a[k] > b[k]);
        eq = eq && a[k] == b[k];
    }
    return gt;
}

// Wave-wide pick among lane-held offers: the lexicographic maximum of dv, then the smallest key1, then the smallest key2.  Returns false
// when no lane offers.  M / k1 / k2 are wave-uniform on return.
template <int L>
__device__ __forceinline__ bool rv2_pick(bool offer, const int32_t (&dv)[L], uint32_t key1, uint32_t key2, int32_t (&M)[L], uint32_t& k1, uint32_t& k2) {
    if (__ballot(offer) == 0ull) return false;
    bool in_max = offer;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        M[k] = wave_max_i32(in_max ? dv[k] : (int32_t)0x80000000);
        in_max = in_max && dv[k] == M[k];
    }
    k1 = wave_min_u32(in_max ? key1 : 0xFFFFFFFFu);
    in_max = in_max && key1 == k1;
    k2 = wave_min_u32(in_max ? key2 : 0xFFFFFFFFu);
    return true;
}

// Trial score of candidate `cd` = (list, count, ascending positions) on the committed lists.  words = the per-list table of this step
// (rv2_build_words; restored on return), arena = `arena_cap` u16 of scratch, work = RuinLds::work.  Returns false -- with the edge table
// and the words as it found them -- when the arena cannot hold the lists the candidate changes; the caller then takes sf_ruin.h's path.
// `keep` (the committed move, wave-uniform): the recreated lists replace the committed ones -- the flat CSR is rebuilt from the per-list
// words through `gscratch` (n_cap words of HBM nobody reads inside a launch: the replica's state of the previous launch), offsets and the
// loads of the changed lists follow; the edge table and the words are left stale (both are rebuilt at the start of the next step).
template <int L>
__device__ __forceinline__ bool ruin_trial_v2(const ListModel& lm, uint16_t* visits, uint32_t* off, int64_t* load, const uint16_t* cd, uint16_t* work,
                                              uint32_t* words, const RuinFast& rf, uint16_t* arena, uint32_t arena_cap, int skip_empty, const int64_t* cur,
                                              int64_t* out_score, bool keep = false, uint32_t* gscratch = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t V = (uint32_t)lm.V, depot = (uint32_t)lm.depot, dim = (uint32_t)lm.dim;
    const bool has_cap = lm.cap_level >= 0 && lm.demand != nullptr;
    const int32_t cap32 = (int32_t)lm.capacity;
    int32_t ca[L], cb[L];  // delta[k] = ca[k] * (capacity overshoot delta) + cb[k] * (distance delta)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        ca[k] = (has_cap && k == lm.cap_level) ? -(int32_t)lm.cap_weight : 0;
        cb[k] = k == lm.dist_level ? -(int32_t)lm.dist_weight : 0;
    }
    const uint32_t* load32 = (const uint32_t*)load;  // low words (small32: loads < 2^28)
    const uint16_t* mat16 = lm.mat16;
    const uint32_t ent = uni((uint32_t)cd[0]), cnt = uni((uint32_t)cd[1]);
    const uint32_t oe = uni(off[ent]), plen = uni(off[ent + 1]) - oe, klen = plen - cnt;
    RPH_DECL
    uint16_t* elog = work;  // [<= 24][2] (edge index | 0x8000 for edge_end, old value), restored in reverse
    uint32_t nlog = 0;
    auto log_edge = [&](uint32_t idx, bool is_end) {  // wave-uniform: remember the old value of edge[idx] / edge_end[idx]
        if (lane == 0) {
            elog[nlog * 2] = (uint16_t)(is_end ? (0x8000u | idx) : idx);
            elog[nlog * 2 + 1] = is_end ? rf.edge_end[idx] : rf.edge[idx];
        }
        nlog += 1;
    };
    // modified lists (lane m < nmod holds one): list id, its current load, whether a placement went into it
    uint32_t mod_list = 0xFFFFFFFFu, nmod = 0, dirty = 0;
    int32_t mod_load = 0;
    uint32_t abump = 0;
    bool ok = true;

    // ---- removed elements: lane j < cnt ----
    const uint32_t pj = lane < cnt ? (uint32_t)cd[2 + lane] : 0xFFFFu;
    const uint32_t xj = lane < cnt ? (uint32_t)visits[oe + pj] : 0u;
    const int32_t dxj = (has_cap && lane < cnt) ? lm.demand[xj] : 0;
    const int32_t dem_removed = has_cap ? wave_sum_i32(dxj) : 0;
    const int32_t load_ent = has_cap ? (int32_t)load32[2u * ent] : 0;

    // ---- the source list without them: a copy in the arena (room for every element to come back) ----
    if (plen > arena_cap) return false;
    const uint32_t a_ent = abump;
    abump += plen;
    for (uint32_t c0 = 0; c0 < klen; c0 += 64) {
        const uint32_t q = c0 + lane;
        if (q < klen) {
            uint32_t src = q;  // q-th kept element = old position q + #removed positions <= it
            for (uint32_t j = 0; j < cnt; ++j) src += (uint32_t)cd[2 + j] <= src ? 1u : 0u;
            arena[a_ent + q] = visits[oe + src];
        }
    }
    if (lane == 0) words[ent] = rv2_word(a_ent, klen, true);
    if (lane == nmod) mod_list = ent, mod_load = load_ent - dem_removed;
    nmod += 1;

    // ---- score after the removal; the seams of the removed runs in the edge table ----
    int64_t s[L];
#pragma unroll
    for (int k = 0; k < L; ++k) s[k] = cur[k];
    {
        const uint32_t pprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pj, 0x138, 0xf, 0xf, false);  // wave_shr:1 -> position of removed element j - 1
        const bool run_start = lane < cnt && (lane == 0 || pj != pprev + 1u);
        int32_t contrib = lane < cnt ? -(int32_t)(uint32_t)rf.edge[xj] : 0;
        uint32_t seam_idx = 0, seam_new = 0;
        bool seam_end = false;
        if (run_start) {
            uint32_t pb = pj;
            for (uint32_t b = lane + 1; b < cnt && (uint32_t)cd[2 + b] == pb + 1u; ++b) pb += 1;
            const uint32_t prev = pj > 0 ? (uint32_t)visits[oe + pj - 1] : depot;
            const bool has_next = pb + 1 < plen;
            const uint32_t nxt = has_next ? (uint32_t)visits[oe + pb + 1] : depot;
            const uint32_t after = has_next ? (uint32_t)rf.edge[nxt] : (uint32_t)rf.edge_end[ent];
            seam_new = klen == 0 ? 0u : (uint32_t)mat16[prev * dim + nxt];  // an empty list costs nothing
            contrib += (int32_t)seam_new - (int32_t)after;
            seam_idx = has_next ? nxt : ent;
            seam_end = !has_next;
        }
        const int32_t d_dist = wave_sum_i32(contrib);
        wave_sync();  // every read of the old edges is done
        uint64_t rs = __ballot(run_start);
        while (rs) {  // (wave-uniform loop: the log is ordered)
            const int j = __ffsll((unsigned long long)rs) - 1;
            rs &= rs - 1;
            const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)seam_idx, j), nv = (uint32_t)__builtin_amdgcn_readlane((int)seam_new, j);
            const bool is_end = __builtin_amdgcn_readlane((int)seam_end, j) != 0;
            log_edge(idx, is_end);
            if (lane == 0) {
                if (is_end)
                    rf.edge_end[idx] = (uint16_t)nv;
                else
                    rf.edge[idx] = (uint16_t)nv;
            }
        }
        const int32_t over1 = load_ent - dem_removed - cap32, over0 = load_ent - cap32;
        const int32_t d_cap = has_cap ? (over1 > 0 ? over1 : 0) - (over0 > 0 ? over0 : 0) : 0;
#pragma unroll
        for (int k = 0; k < L; ++k) s[k] += (int64_t)(ca[k] * d_cap + cb[k] * d_dist);
    }
    wave_sync();

    RPH(0)
    // current load of list e (lane-private e): the committed load unless the candidate changed the list
    auto load_of = [&](uint32_t e) -> int32_t {
        int32_t l0 = has_cap ? (int32_t)load32[2u * e] : 0;
        for (uint32_t m = 0; m < nmod; ++m) {
            const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)mod_list, (int)m);
            const int32_t mv = __builtin_amdgcn_readlane(mod_load, (int)m);
            l0 = e == ml ? mv : l0;
        }
        return l0;
    };

    // ---- first scan: element i against every list, a lane per list; its cnt best lists kept in lanes i * 8 + m ----
    int32_t T_dv[L];
    uint32_t T_key = 0;  // list << 16 | position
    bool T_ok = false;
#pragma unroll
    for (int k = 0; k < L; ++k) T_dv[k] = 0;
    const uint32_t n_pass = (V + 63u) / 64u;  // <= 2
    // ONE walk over every list prices its slots for ALL removed elements: the list read, the end test and the edge are the same for each of
    // them; an element adds its row entry x -> next (x -> prev is the previous slot's) and a compare.  The legs come straight from the
    // matrix rows (2 KB each, 16 lines: L1 / L2 hits); sf_ruin.h stages one row at a time in LDS for its slot-parallel scan.
    constexpr uint32_t NE = RUIN_MAX_COUNT;
    const uint16_t* xrow[NE];
    int32_t dxs[NE];
    uint32_t rw_depot[NE];
#pragma unroll
    for (uint32_t i = 0; i < NE; ++i) {
        const uint32_t x = i < cnt ? (uint32_t)__builtin_amdgcn_readlane((int)xj, (int)i) : depot;
        dxs[i] = __builtin_amdgcn_readlane(dxj, (int)i);
        xrow[i] = mat16 + x * dim;
        rw_depot[i] = uni((uint32_t)xrow[i][depot]);
    }
    int32_t b_dv[2][NE][L];
    uint32_t b_pos[2][NE];
    uint32_t b_has[2] = {0u, 0u};  // bit i: this lane's list has a slot for element i
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (uint32_t i = 0; i < NE; ++i) {
            b_pos[ps][i] = 0;
#pragma unroll
            for (int k = 0; k < L; ++k) b_dv[ps][i][k] = (int32_t)0x80000000;
        }
        if ((uint32_t)ps >= n_pass) continue;
        const uint32_t e = (uint32_t)ps * 64u + lane;
        const bool in_v = e < V;
        const uint32_t ec = in_v ? e : 0u;
        const uint32_t w = words[ec];
        const uint32_t len = (w >> 16) & 0x7FFFu;
        const uint16_t* lp = ((w & RV2_ARENA) ? arena : visits) + (w & 0xFFFFu);
        const bool act = in_v && !(skip_empty && len == 0);
        const int32_t l0 = load_of(ec);
        int32_t dcs[NE];
#pragma unroll
        for (uint32_t i = 0; i < NE; ++i) {
            const int32_t o1 = l0 + dxs[i] - cap32, o0 = l0 - cap32;
            dcs[i] = has_cap ? (o1 > 0 ? o1 : 0) - (o0 > 0 ? o0 : 0) : 0;
        }
        const uint32_t nslots = uni((uint32_t)wave_max_i32(act ? (int32_t)len + 1 : 0));
        uint32_t rw_prev[NE];
#pragma unroll
        for (uint32_t i = 0; i < NE; ++i) rw_prev[i] = rw_depot[i];
        constexpr uint32_t U = SF_RV2_U;  // slots of the walk in flight per lane (a variant build sweeps it: scripts/build_variant.sh ... "-DSF_RV2_U=8")
        for (uint32_t q0 = 0; q0 < nslots; q0 += U) {
            uint32_t nxr[U], d0[U];
            bool at_end[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t q = q0 + u;
                nxr[u] = lp[q < len ? q : 0u];
                at_end[u] = q >= len;
            }
            uint32_t rw[NE][U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t nx = at_end[u] ? depot : nxr[u];
#pragma unroll
                for (uint32_t i = 0; i < NE; ++i)
                    if (i < cnt) rw[i][u] = xrow[i][nx];
                const uint16_t* dp = at_end[u] ? rf.edge_end + ec : rf.edge + nxr[u];  // an empty list's edge_end is 0
                d0[u] = *dp;
            }
#pragma unroll
            for (uint32_t i = 0; i < NE; ++i) {
                if (i >= cnt) continue;
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    const uint32_t q = q0 + u;
                    const uint32_t da = u == 0 ? rw_prev[i] : rw[i][u - 1];
                    const int32_t dd = (int32_t)da + (int32_t)rw[i][u] - (int32_t)d0[u];
                    int32_t dv[L];
#pragma unroll
                    for (int k = 0; k < L; ++k) dv[k] = ca[k] * dcs[i] + cb[k] * dd;
                    const bool has = (b_has[ps] >> i) & 1u;
                    const bool take = act && q <= len && (!has || rv2_gt<L>(dv, b_dv[ps][i]));  // strictly better: the first of equals stays
#pragma unroll
                    for (int k = 0; k < L; ++k) b_dv[ps][i][k] = take ? dv[k] : b_dv[ps][i][k];
                    b_pos[ps][i] = take ? q : b_pos[ps][i];
                    b_has[ps] |= take ? (1u << i) : 0u;
                }
                rw_prev[i] = rw[i][U - 1];
            }
        }
    }
    RPH(1)
    // the cnt best lists of every element in (score, list) order -> lanes i * 8 + m
#pragma unroll
    for (uint32_t i = 0; i < NE; ++i) {
        if (i >= cnt) continue;
        bool h0 = (b_has[0] >> i) & 1u, h1 = (b_has[1] >> i) & 1u;
        for (uint32_t m = 0; m < cnt; ++m) {
            // a lane offers the better of its two lists (the first one on a tie: it is the lower list)
            const bool second = h1 && (!h0 || rv2_gt<L>(b_dv[1][i], b_dv[0][i]));
            int32_t o_dv[L];
#pragma unroll
            for (int k = 0; k < L; ++k) o_dv[k] = second ? b_dv[1][i][k] : b_dv[0][i][k];
            const bool offer = h0 || h1;
            const uint32_t o_key = (((second ? 64u : 0u) + lane) << 16) | (second ? b_pos[1][i] : b_pos[0][i]);
            int32_t M[L];
            uint32_t k1 = 0, k2 = 0;
            if (!rv2_pick<L>(offer, o_dv, o_key, 0u, M, k1, k2)) break;
            if (offer && o_key == k1) {  // the owner retires that list
                if (second)
                    h1 = false;
                else
                    h0 = false;
            }
            if (lane == i * 8u + m) {
#pragma unroll
                for (int k = 0; k < L; ++k) T_dv[k] = M[k];
                T_key = k1;
                T_ok = true;
            }
        }
    }
    RPH(2)
    // ---- rounds: place the best (element, list, position) until nothing remains ----
    uint32_t remmask = (1u << cnt) - 1u;
    bool rolled_back = false;
    while (remmask) {
        // Every lane keeps ONE running offer (delta, element, list << 16 | position), ordered like the reference's scan: better score, then
        // the earlier element, then the earlier (list, position).  It starts from the lane's kept entry (if that list is untouched) ...
        const uint32_t my_i = lane >> 3;
        bool o_has = T_ok && my_i < cnt && ((remmask >> my_i) & 1u) != 0;
        for (uint32_t d = 0; d < nmod; ++d) {
            const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)mod_list, (int)d);
            if ((dirty >> d) & 1u) o_has = o_has && (T_key >> 16) != le;
        }
        int32_t o_dv[L];
#pragma unroll
        for (int k = 0; k < L; ++k) o_dv[k] = T_dv[k];
        uint32_t o_i = my_i, o_key = T_key;
        // ... and takes in the lists changed so far, priced again for every remaining element (lane = slot): no reduction per list
        for (uint32_t d = 0; d < nmod; ++d) {
            if (!((dirty >> d) & 1u)) continue;
            const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)mod_list, (int)d);
            const int32_t l0 = __builtin_amdgcn_readlane(mod_load, (int)d);
            const uint32_t w = uni(words[le]);
            const uint32_t len = (w >> 16) & 0x7FFFu;
            const uint16_t* lp = arena + (w & 0xFFFFu);
            for (uint32_t c0 = 0; c0 <= len; c0 += 64) {
                const uint32_t q = c0 + lane;
                const bool in = q <= len;
                const bool at_end = q >= len;
                const uint32_t pv = (in && q > 0) ? (uint32_t)lp[q - 1] : depot;
                const uint32_t nxr = (in && !at_end) ? (uint32_t)lp[q] : 0u;
                const uint32_t nx = at_end ? depot : nxr;
                const uint16_t* dp = at_end ? rf.edge_end + le : rf.edge + nxr;
                const int32_t d0 = (int32_t)(uint32_t)*dp;
                uint32_t rm = remmask;
                while (rm) {
                    const uint32_t i = (uint32_t)__ffs((int)rm) - 1u;
                    rm &= rm - 1u;
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xj, (int)i);
                    const int32_t dx = __builtin_amdgcn_readlane(dxj, (int)i);
                    const int32_t o1 = l0 + dx - cap32, o0 = l0 - cap32;
                    const int32_t dc = has_cap ? (o1 > 0 ? o1 : 0) - (o0 > 0 ? o0 : 0) : 0;
                    const int32_t da = (int32_t)(uint32_t)mat16[x * dim + pv], db = (int32_t)(uint32_t)mat16[x * dim + nx];
                    const int32_t dd = da + db - d0;
                    int32_t dv[L];
#pragma unroll
                    for (int k = 0; k < L; ++k) dv[k] = ca[k] * dc + cb[k] * dd;
                    const uint32_t key = (le << 16) | q;
                    const bool gt = rv2_gt<L>(dv, o_dv), lt = rv2_gt<L>(o_dv, dv);
                    const bool take = in && (!o_has || gt || (!lt && (i < o_i || (i == o_i && key < o_key))));
#pragma unroll
                    for (int k = 0; k < L; ++k) o_dv[k] = take ? dv[k] : o_dv[k];
                    o_i = take ? i : o_i;
                    o_key = take ? key : o_key;
                    o_has = o_has || take;
                }
            }
        }
        RPH(3)
        int32_t M[L];
        uint32_t wi = 0, wkey = 0;
        if (!rv2_pick<L>(o_has, o_dv, o_i, o_key, M, wi, wkey)) {  // no destination at all: restore_removed_elements (:250-253)
            rolled_back = true;
            break;
        }
        const uint32_t be = wkey >> 16, bp = wkey & 0xFFFFu;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xj, (int)wi);
        const int32_t dx = __builtin_amdgcn_readlane(dxj, (int)wi);
#pragma unroll
        for (int k = 0; k < L; ++k) s[k] += (int64_t)M[k];
        remmask &= ~(1u << wi);
        if (!remmask && !keep) break;  // (a trial: the last placement changes nothing anybody reads)
        // ---- the placement: list `be` gets x at position bp (a copy in the arena; the edge table's two seams) ----
        uint32_t md = 0xFFFFFFFFu;
        for (uint32_t d = 0; d < nmod; ++d)
            if ((uint32_t)__builtin_amdgcn_readlane((int)mod_list, (int)d) == be) md = d;
        uint32_t w = uni(words[be]);
        uint32_t len = (w >> 16) & 0x7FFFu;
        if (md == 0xFFFFFFFFu) {  // first change of this list: copy it (room for the elements still to place)
            const uint32_t need = len + (uint32_t)__popc(remmask) + 1u;
            if (abump + need > arena_cap) {
                ok = false;
                break;
            }
            const uint32_t ab = abump;
            abump += need;
            const uint16_t* src = visits + (w & 0xFFFFu);
            for (uint32_t q = lane; q < len; q += 64) arena[ab + q] = src[q];
            w = rv2_word(ab, len, true);
            md = nmod;
            if (lane == md) mod_list = be, mod_load = has_cap ? (int32_t)load32[2u * be] : 0;
            nmod += 1;
            wave_sync();
        }
        uint16_t* lp = arena + (w & 0xFFFFu);
        const uint32_t prev = bp > 0 ? uni((uint32_t)lp[bp - 1]) : depot;
        const bool has_next = bp < len;
        const uint32_t nxt = has_next ? uni((uint32_t)lp[bp]) : depot;
        for (uint32_t c0 = 0; c0 < len - bp; c0 += 64) {  // descending chunks: (bp, len] <- t - 1
            const uint32_t dd = c0 + lane;
            const bool in = dd < len - bp;
            const uint32_t t = len - (in ? dd : 0u);
            uint32_t nv = 0;
            if (in) nv = lp[t - 1];
            wave_sync();
            if (in) lp[t] = (uint16_t)nv;
            wave_sync();
        }
        const uint32_t w_da = uni((uint32_t)mat16[x * dim + prev]), w_db = uni((uint32_t)mat16[x * dim + nxt]);
        log_edge(x, false);
        log_edge(has_next ? nxt : be, !has_next);
        if (lane == 0) {
            lp[bp] = (uint16_t)x;
            rf.edge[x] = (uint16_t)w_da;
            if (has_next)
                rf.edge[nxt] = (uint16_t)w_db;
            else
                rf.edge_end[be] = (uint16_t)w_db;
            words[be] = rv2_word(w & 0xFFFFu, len + 1u, true);
        }
        if (lane == md) mod_load += dx;
        dirty |= 1u << md;
        wave_sync();
        RPH(4)
    }
    RPH(4)

    if (keep && ok && !rolled_back) {
        // ---- the committed move: new offsets (prefix over the lists' current lengths), every list copied to its new place ----
        uint32_t carry = 0;
        for (uint32_t base = 0; base < V; base += 64) {
            const uint32_t e = base + lane;
            const uint32_t w = e < V ? words[e] : 0u;
            const uint32_t len = (w >> 16) & 0x7FFFu;
            const uint32_t inc = wave_incl_scan(len);
            const uint32_t no = carry + inc - len;
            if (e < V) {
                const uint16_t* src = ((w & RV2_ARENA) ? arena : visits) + (w & 0xFFFFu);
                for (uint32_t q = 0; q < len; ++q) gscratch[no + q] = src[q];
            }
            wave_sync();  // (the reads of `off` by nobody: the sources are named by the words)
            if (e < V) off[e] = no;
            carry += (uint32_t)__shfl((int)inc, 63);
        }
        if (lane == 0) off[V] = carry;
        if (has_cap && lane < nmod) load[mod_list] = (int64_t)mod_load;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the copies, through the CU's write-through L1, before their reload
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t t = lane; t < carry; t += 64) visits[t] = (uint16_t)gscratch[t];
        wave_sync();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < L; ++k) out_score[k] = s[k];
        }
        return true;
    }
    // ---- leave everything as it was found ----
    for (uint32_t t = nlog; t-- > 0;) {
        if (lane == 0) {
            const uint32_t idx = elog[t * 2], val = elog[t * 2 + 1];
            if (idx & 0x8000u)
                rf.edge_end[idx & 0x7FFFu] = (uint16_t)val;
            else
                rf.edge[idx] = (uint16_t)val;
        }
    }
    if (lane < nmod) words[mod_list] = rv2_word(off[mod_list], off[mod_list + 1] - off[mod_list], false);
    wave_sync();
    RPH(5)
#ifdef SF_PHASE_PROFILE
    if (lane == 0) for (int _k = 0; _k < 8; ++_k) atomicAdd(&g_rphase2[_k], (unsigned long long)rph_acc[_k]);
#endif
    if (!ok) return false;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) out_score[k] = rolled_back ? cur[k] : s[k];
    }
    return true;
}

}  // namespace sf
