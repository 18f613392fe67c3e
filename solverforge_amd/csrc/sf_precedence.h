// ListPrecedenceMakespanConstraint on the device (crates/solverforge-scoring/src/constraint/list_precedence.rs:13-707).
//
// Reference semantics restated: nodes = list elements with a duration; edges = the fixed successor relation plus every pair
// of consecutive elements of every owner's list; score = HardSoft(-(invalid fixed edges + invalid items + wrong-owner items +
// assignment penalty + cycle penalty), -makespan), makespan = the longest duration-weighted path, a cyclic graph costs
// node_count hard and has makespan 0.  The reference keeps hash-indexed adjacency and refreshes the earliest starts of the
// descendants of changed edges (refresh_graph_after_route_change :494-551), with a full Kahn rebuild when the cached state is
// cyclic (rebuild_graph_summary :553-589).  Every path ends in the same pure function of the lists, which is what this file
// computes: one wavefront runs Kahn's algorithm over the whole graph with a frontier queue -- the 64 lanes pop 64 ready nodes
// per round, push their finish time into the successors with atomic max, and the lane that takes a successor's in-degree to
// zero appends it to the queue (ballot + mbcnt compaction, queue tail wave-uniform).  Rounds >= the depth of the graph, so the
// evaluation is latency bound by construction; the scratch arrays (earliest start, in-degree, queue, list successor) sit in the
// replica's LDS slice when the node count allows and in HBM / L2 otherwise (accessed with agent-scope atomics: the vector L1
// is not coherent with the L2 atomics that update the same words).
//
// Preconditions checked on the host (sf_api.hip): element ids < node_count, every element in at most one list (the device's
// list moves keep it so), sum of durations < 2^31.  With them owner_invalid_total = 0 and the assignment penalty is the number
// of unassigned nodes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace sf {

constexpr uint32_t PREC_NONE = 0xFFFFFFFFu;

struct PrecModel {
    int32_t on;  // 0 = the model has no precedence constraint
    int32_t hard_level, mk_level;
    int32_t n;                 // node_count
    int32_t n_edges;           // valid fixed edges (entries of succ / pred)
    const int32_t* dur;        // [n]
    const uint32_t* succ_off;  // [n + 1] fixed successors (valid ones only)
    const uint32_t* succ;
    const int32_t* indeg0;     // [n] fixed in-degree
    const int32_t* owner;      // [n] expected owner, -1 = none; nullptr = no expected-owner hook
    const uint32_t* nd;        // [n][2] node records: duration; min(out-degree, 255) << 24 | first fixed successor (0xFFFFFF = none)
    int64_t const_penalty;     // invalid fixed edges (successor >= node_count)
    // per-replica scratch in HBM, [R][n] each
    int32_t* earliest;
    int32_t* indeg;
    uint32_t* queue;
    uint32_t* lsucc;
    int64_t* state;  // [R][2] (hard penalty, makespan) of the committed lists
    // incremental trial refresh (prec_trial_inc; generic engine, scratch in HBM): fixed predecessors, and per replica the list
    // predecessor of every node, trial earliest starts with their stamps, queue stamps, the nodes a trial changed, a second frontier
    const uint32_t* pred_off;  // [n + 1] fixed predecessors (valid ones only)
    const uint32_t* pred;
    uint32_t* lpred;     // [R][n]
    uint32_t* stamp_e;   // [R][n] trial id whose earliest start sits in `indeg` (reused as the trial value array)
    uint32_t* stamp_q;   // [R][n] round stamp: queued for the next frontier / visited by a cycle search
    uint32_t* changed;   // [R][n] nodes whose earliest start a trial replaced
    uint32_t* queue2;    // [R][n] second frontier
    // lane-per-trial sweep (prec_trial_sweep64): the committed topological order lives in `queue` (Kahn's pop order), plus
    uint32_t* pos;       // [R][n]      position of every node in that order
    uint32_t* roff;      // [R][n + 1]  start of every Kahn round in the order (the nodes of one round are mutually independent)
    uint32_t* rnd;       // [R][n]      Kahn round of every node
    uint32_t* rec;       // [R][n][16]  one 64-byte sweep record per order position (PrecRec)
    int32_t* pmax;       // [R][n + 1]  prefix maximum of the finish times along the order
    int32_t* elane;      // [R][n][64]  earliest starts of the 64 trials in flight, node-major (one coalesced 256-byte row per node)
    int32_t has_zero_duration;  // a zero-duration node could hide a cycle from the sweep: the full evaluation is used instead
};

// LDS bytes of the Kahn scratch of one replica (earliest start i32, in-degree i32, queue u16, list successor u16: PrecMemLds)
__host__ __device__ inline size_t prec_lds_scratch_bytes(int n) { return (size_t)n * 12; }
// the slim workgroup-shared copy of the static graph: node records, fixed in-degrees, owners -- what every wave-wide evaluation reads per
// node (models whose full copy -- plus durations and the CSR arrays -- does not fit)
__host__ __device__ inline size_t prec_static_slim_bytes(int n, bool has_owner) { return 4 * (size_t)n * (has_owner ? 4 : 3) + 16; }

struct PrecResult {
    int64_t penalty, makespan;
};

// Memory policies of the Kahn scratch.  HBM / L2: agent-scope atomic loads / stores (the vector L1 is not coherent with the L2
// atomics that update the same words).  LDS: address-space typed pointers, so the accesses are ds_read / ds_write / ds_max_rtn /
// ds_add_rtn instead of FLAT instructions.
struct PrecMemGlobal {
    typedef int32_t* I32;
    typedef uint32_t* U32;
    static __device__ __forceinline__ int32_t ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ uint32_t ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void fmax(int32_t* p, int32_t v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ int32_t fadd(int32_t* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    template <class T>
    static __device__ __forceinline__ const T* lists(const T* p) { return p; }  // the lists may live anywhere (generic pointer)
    static constexpr uint32_t IDX_NONE = 0xFFFFFFFFu;  // "none" as ld_idx returns it
    static __device__ __forceinline__ uint32_t ld_idx(const uint32_t* p) { return ld(p); }
};
typedef __attribute__((address_space(3))) int32_t prec_lds_i32;
typedef __attribute__((address_space(3))) uint32_t prec_lds_u32;
typedef __attribute__((address_space(3))) uint16_t prec_lds_u16;
// The index arrays (queue, list successor) are 16 bits wide in LDS -- the scratch lives there for a few thousand nodes at most -- with
// 0xFFFF standing for PREC_NONE: 12 bytes per node instead of 16, which is what leaves room for the workgroup's copy of the node records.
struct PrecMemLds {
    typedef prec_lds_i32* I32;
    typedef prec_lds_u16* U32;
    static __device__ __forceinline__ int32_t ld(const prec_lds_i32* p) { return *p; }
    static __device__ __forceinline__ uint32_t ld(const prec_lds_u16* p) {
        const uint32_t v = *p;
        return v == 0xFFFFu ? 0xFFFFFFFFu : v;
    }
    static __device__ __forceinline__ void st(prec_lds_i32* p, int32_t v) { *p = v; }
    static __device__ __forceinline__ void st(prec_lds_u16* p, uint32_t v) { *p = (uint16_t)v; }
    static __device__ __forceinline__ void fmax(prec_lds_i32* p, int32_t v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static __device__ __forceinline__ int32_t fadd(prec_lds_i32* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static constexpr uint32_t IDX_NONE = 0xFFFFu;  // "none" as ld_idx returns it: the stored 16 bits, not widened to PREC_NONE
    static __device__ __forceinline__ uint32_t ld_idx(const prec_lds_u16* p) { return *p; }
    template <class T>  // scratch in LDS <=> the caller's lists are the replica's LDS copy: ds_read instead of FLAT
    static __device__ __forceinline__ const __attribute__((address_space(3))) T* lists(const T* p) { return (const __attribute__((address_space(3))) T*)p; }
};
// cross-lane hand-off through memory inside one wavefront: every outstanding memory operation of the wave has completed
__device__ __forceinline__ void prec_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint32_t prec_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

#ifdef SF_PHASE_PEVAL  // diagnostics (-DSF_PHASE_PROFILE -DSF_PHASE_PEVAL): shader clocks of the stages of prec_eval in g_rphase2 (sf_ruin_v2.h defines it):
// 0 evaluations, 1 init, 2 list pass, 3 ready scan, 4 Kahn rounds, 5 number of rounds, 6 nodes
extern __device__ unsigned long long g_rphase2[8];
#define PEV(i)                                                                       \
    {                                                                                \
        const uint64_t _t = clock64();                                               \
        if (pev_on) atomicAdd(&g_rphase2[i], (unsigned long long)(_t - pev_t));      \
        pev_t = _t;                                                                  \
    }
#else
#define PEV(i)
#endif

// Full evaluation of the lists `visits` / `off` (V owners) by one wavefront; E / D / Q / S = the four scratch arrays of pm.n
// words.  Wave-uniform result.  lint: small-pod-return (PrecResult = two int64: returned in four registers; never called inside a
// conditional expression -- DESIGN 8.15 item 2 was a 32-byte struct through `?:`; scripts/lint_device_patterns.py checks both)
// STATIC_LDS: pm.nd / pm.indeg0 / pm.owner point into the workgroup's LDS copy of the static graph (sf_mixed_wave.hip) -- read with ds_read
// instead of FLAT instructions (a FLAT access that lands in LDS still takes the vector-memory path first).
// ORDERED = false (trial and commit evaluations whose pop order nobody reads): when every node is in a list, the ready set is collected by the
// list pass itself (the heads of the lists without a fixed predecessor) instead of a scan over the in-degrees -- the queue then starts in list
// order, not in ascending node order.  The result does not depend on the pop order.
// (lint: small-pod-return -- see above)
template <class VT, class MEM = PrecMemGlobal, bool ORDERED = true, bool STATIC_LDS = false>
__device__ __noinline__ PrecResult prec_eval(const PrecModel& pm_ref, const VT* visits, const uint32_t* off, int V, typename MEM::I32 E, typename MEM::I32 D,
                                             typename MEM::U32 Q, typename MEM::U32 S, uint32_t* LP = nullptr, uint32_t* out_info = nullptr, uint32_t* ROFF = nullptr) {
    // the parameter block arrives by reference (one copy per kernel instead of one 240-byte stack copy per inlined call site); the stage works on a
    // private copy of it -- not escaping, so the optimizer splits it into the fields this stage uses and keeps them in registers; read through the
    // reference, every field would have to be re-read after each store to the scratch arrays (they may alias)
    const PrecModel pm = pm_ref;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    const auto vis = MEM::lists(visits);
    const auto offs = MEM::lists(off);
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    typedef __attribute__((address_space(3))) const int32_t lds_ci32;
    auto nd_at = [&](uint32_t i) -> uint32_t {
        if constexpr (STATIC_LDS) return ((lds_cu32*)pm.nd)[i];
        else return pm.nd[i];
    };
    auto indeg0_at = [&](uint32_t i) -> int32_t {
        if constexpr (STATIC_LDS) return ((lds_ci32*)pm.indeg0)[i];
        else return pm.indeg0[i];
    };
#ifdef SF_PHASE_PEVAL
    const bool pev_on = lane == 0 && (blockIdx.x & 31u) == 0u;  // (one workgroup in 32 reports: same-address atomics of every wave would be the bottleneck)
    uint64_t pev_t = clock64();
    if (pev_on) atomicAdd(&g_rphase2[0], 1ull), atomicAdd(&g_rphase2[6], (unsigned long long)n);
#endif
    // ---- set-up.  Every loop keeps four independent iterations in flight: the evaluation is a chain of dependent round trips from start to end,
    // and at 1,000 nodes the set-up used to be as long a chain as Kahn's rounds (one binary search over `off` per 64 items, every load waited for).
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)offs[V]);
    const bool all_listed = total == n && LP == nullptr;  // the list pass writes every node's three words itself: no init pass
    for (uint32_t i0 = 0; i0 < n && !all_listed; i0 += 256) {
        int32_t d0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * u + lane;
            d0[u] = indeg0_at(i < n ? i : 0u);  // (unconditional: see the list pass)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * u + lane;
            if (i < n) {
                MEM::st(E + i, 0);
                MEM::st(D + i, d0[u]);
                MEM::st(S + i, PREC_NONE);
                if (LP) LP[i] = PREC_NONE;
            }
        }
    }
    prec_sync();
    PEV(1)
    uint32_t viol = 0;
    uint32_t head = 0, tail = 0;
    const bool ready_from_lists = !ORDERED && total == n;  // every node is listed: only list heads can be ready
    {   // list by list, 64 positions per chunk, four chunks in flight: list successor, in-degree and owner check of every item.  The
        // list offsets sit one per lane (V < 64; otherwise they are read where needed), so walking the chunks costs no memory access.
        const bool has_owner = pm.owner != nullptr;
        const int32_t* const own = has_owner ? pm.owner : pm.indeg0;  // (always a readable table)
        const uint32_t my_off = V < 64 ? (uint32_t)offs[lane <= (uint32_t)V ? lane : (uint32_t)V] : 0u;
        const uint32_t my_off1 = V < 64 ? (uint32_t)offs[lane + 1 <= (uint32_t)V ? lane + 1 : (uint32_t)V] : 0u;  // (end of list `lane`)
        auto list_pass = [&](auto held_c) {  // (two copies of the loop: the offset source is a compile-time choice inside each)
            constexpr bool held = decltype(held_c)::value;
            auto off_at = [&](uint32_t e) -> uint32_t {
                if constexpr (held) return (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)e);
                else return (uint32_t)__builtin_amdgcn_readfirstlane((int)offs[e]);
            };
            // chunk -> (list, start): with the offsets held per lane, lane e also holds the number of chunks of list e and their running total
            // (one wave scan per evaluation), and a chunk finds its list with one ballot -- no loop over the lists, nothing divergent
            uint32_t cstart = 0, nch = 0, n_chunks = 0;
            if constexpr (held) {
                const uint32_t len = lane < (uint32_t)V ? my_off1 - my_off : 0u;
                nch = (len + 63u) >> 6;
                uint32_t incl = nch;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                    incl += lane >= (uint32_t)o ? up : 0u;
                }
                cstart = incl - nch;
                n_chunks = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            uint32_t e = 0, k0 = 0, eo = (!held && V > 0) ? off_at(0) : 0u, en = (!held && V > 0) ? off_at(1) : 0u;  // (walker of the offset-table path)
            for (uint32_t c0 = 0;; c0 += 4) {
                uint32_t ce[4], cb[4], cl[4];  // chunk: list, flat index of its first position, positions left in the list from there
                bool cf[4];                    // the chunk starts its list
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (held) {
                        const uint32_t c = c0 + (uint32_t)u;
                        const uint64_t hit = __ballot(c >= cstart && c < cstart + nch);  // (exactly one lane for c < n_chunks)
                        const bool ok = c < n_chunks;
                        const uint32_t le = ok ? (uint32_t)__builtin_ctzll(hit) : 0u;
                        const uint32_t ko = (c - (uint32_t)__builtin_amdgcn_readlane((int)cstart, (int)le)) << 6;
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)le), hi = (uint32_t)__builtin_amdgcn_readlane((int)my_off1, (int)le);
                        ce[u] = le, cb[u] = ok ? lo + ko : 0u, cl[u] = ok ? hi - (lo + ko) : 0u, cf[u] = ko == 0u;
                        any = any || ok;
                        continue;
                    }
                    while (e < (uint32_t)V && eo + k0 >= en) {  // next list with items left
                        e += 1, k0 = 0, eo = en;
                        en = e < (uint32_t)V ? off_at(e + 1) : en;
                    }
                    const bool ok = e < (uint32_t)V;
                    ce[u] = e, cb[u] = eo + k0, cl[u] = ok ? en - (eo + k0) : 0u, cf[u] = k0 == 0u;
                    any = any || ok;
                    k0 += 64;
                }
                if (!any) break;
                // every load below is unconditional (an idle lane reads position 0 / node vis[0]): a load inside a lane-masked branch is waited
                // for at the end of its branch, which serialised the eight table reads of an iteration
                bool in[4];
                uint32_t x[4], nx[4], px[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    in[u] = lane < cl[u];
                    const uint32_t t = in[u] ? cb[u] + lane : 0u;
                    const bool more = in[u] && lane + 1 < cl[u];
                    x[u] = (uint32_t)vis[t];
                    nx[u] = (uint32_t)vis[more ? t + 1 : 0u];
                    nx[u] = more ? nx[u] : PREC_NONE;
                    px[u] = PREC_NONE;
                    if (LP) px[u] = (in[u] && (lane > 0 || !cf[u])) ? (uint32_t)vis[t - 1] : PREC_NONE;
                }
                int32_t i0v[4], ow[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    i0v[u] = indeg0_at(x[u]);
                    if constexpr (STATIC_LDS) ow[u] = ((lds_ci32*)own)[x[u]];
                    else ow[u] = own[x[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (in[u]) {
                        const bool first = lane == 0 && cf[u];
                        MEM::st(S + x[u], nx[u]);
                        MEM::st(E + x[u], 0);
                        if (LP) LP[x[u]] = px[u];
                        MEM::st(D + x[u], i0v[u] + (first ? 0 : 1));
                        viol += (has_owner && ow[u] >= 0 && (uint32_t)ow[u] != ce[u]) ? 1u : 0u;
                    }
                    if (!ORDERED && ready_from_lists && cf[u]) {  // (uniform: the chunk starts a list; at most lane 0 is ready)
                        const bool ready = in[u] && lane == 0 && i0v[u] == 0;
                        const uint64_t m = __ballot(ready);
                        if (ready) MEM::st(Q + tail, x[u]);
                        tail += (uint32_t)__popcll(m);
                    }
                }
            }
        };
        if (V < 64) list_pass(std::true_type{});
        else list_pass(std::false_type{});
    }
    prec_sync();
    PEV(2)
    for (uint32_t b = 0; b < n && !ready_from_lists; b += 256) {  // the ready nodes in ascending order (four reads in flight)
        int32_t dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = b + 64u * u + lane;
            dv[u] = i < n ? MEM::ld(D + i) : 1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = b + 64u * u + lane;
            const bool ready = dv[u] == 0;
            const uint64_t m = __ballot(ready);
            if (ready) MEM::st(Q + tail + prec_mbcnt(m), i);
            tail += (uint32_t)__popcll(m);
        }
    }
    prec_sync();
    PEV(3)
    int32_t mk = 0;
    uint32_t rounds = 0;
    // A round.  Both relaxations are issued before the first wait: a missing successor relaxes the node itself with operands that change
    // nothing (max with INT_MIN, add 0) instead of sitting in a branch of its own.  (Same target twice -- a fixed successor that is also the
    // list successor: LDS executes a wave's instructions in order, so the two decrements return consecutive values; with the scratch in HBM
    // the second relaxation of such a lane waits for the first.)
    // further fixed successors of the nodes of this round (none in a job shop): the record saturates at 255, the CSR has the rest
    auto more_successors = [&](uint32_t node, uint32_t deg, int32_t fin) {
        if (!__ballot(deg > 1u)) return;
        uint32_t so = 0;
        if (deg > 1u) {
            so = pm.succ_off[node];
            deg = pm.succ_off[node + 1] - so;
        }
        for (uint32_t k = 1;; ++k) {
            const bool has = k < deg;
            if (!__ballot(has)) break;
            bool newly = false;
            uint32_t s = 0;
            if (has) {
                s = pm.succ[so + k];
                MEM::fmax(E + s, fin);
                newly = MEM::fadd(D + s, -1) == 1;
            }
            const uint64_t m = __ballot(newly);
            if (newly) MEM::st(Q + tail + prec_mbcnt(m), s);
            tail += (uint32_t)__popcll(m);
        }
    };
    // A round: the popped node; its record (duration, out-degree, first fixed successor), earliest start and list successor; the relaxations;
    // the queue writes.  The round has NO lane-masked region -- a wave runs about one instruction per 4 - 8 clocks whatever the lanes do, so the
    // mask bookkeeping of five small branches was a quarter of a round: a lane without a popped node re-reads the round's first one, a missing
    // successor (and every idle lane) relaxes a per-lane dummy node with operands that change nothing (max with INT_MIN, add 0), and a lane
    // without a push rewrites a popped slot with the value it already holds.  (Same target twice -- a fixed successor that is also the list
    // successor: LDS executes a wave's instructions in order, so the two decrements return consecutive values; with the scratch in HBM the
    // second relaxation of such a lane waits for the first.)
    constexpr bool in_lds = std::is_same<MEM, PrecMemLds>::value;
    const uint32_t dummy = lane < n ? lane : 0u;
    while (head < tail) {
        const uint32_t cnt = tail - head < 64u ? tail - head : 64u;
        if (ORDERED && ROFF && lane == 0) ROFF[rounds] = head;
        rounds += 1;
        const bool act = lane < cnt;
        const auto slot = Q + head + (act ? lane : 0u);
        const uint32_t node = MEM::ld_idx(slot);
        const uint32_t r0 = nd_at(2 * node), r1 = nd_at(2 * node + 1);  // (two loads: the LDS copy of the records is only 4-byte aligned)
        const int32_t fin = MEM::ld(E + node) + (int32_t)r0;
        const uint32_t ls = MEM::ld_idx(S + node), s1 = r1 & 0xFFFFFFu;
        mk = fin > mk ? fin : mk;
        const uint32_t deg = act ? r1 >> 24 : 0u;
        const bool h1 = act && s1 != 0xFFFFFFu, same = !in_lds && h1 && s1 == ls, h2 = act && ls != MEM::IDX_NONE && !same;
        const uint32_t t1 = h1 ? s1 : dummy, t2 = h2 ? ls : dummy;
        int32_t o1 = 0, o2 = 0;
        if (in_lds || h1) {  // (HBM scratch: no dummy traffic to L2 -- there a round waits on memory, not on the instruction stream)
            MEM::fmax(E + t1, h1 ? fin : INT32_MIN);
            o1 = MEM::fadd(D + t1, h1 ? -1 : 0);
        }
        if (in_lds || h2) {
            MEM::fmax(E + t2, h2 ? fin : INT32_MIN);
            o2 = MEM::fadd(D + t2, h2 ? -1 : 0);
        }
        const bool new1 = h1 && o1 == 1;
        bool new2 = h2 && o2 == 1;
        if (!in_lds && __ballot(same)) {
            if (same) {
                MEM::fmax(E + ls, fin);
                new2 = MEM::fadd(D + ls, -1) == 1;
            }
        }
        const uint64_t m1 = __ballot(new1), m2 = __ballot(new2);
        const uint32_t c1 = (uint32_t)__popcll(m1);
        MEM::st(new1 ? Q + tail + prec_mbcnt(m1) : slot, new1 ? s1 : node);
        MEM::st(new2 ? Q + tail + c1 + prec_mbcnt(m2) : slot, new2 ? ls : node);
        tail += c1 + (uint32_t)__popcll(m2);
        more_successors(node, deg, fin);
        head += cnt;
        prec_sync();
    }
    PEV(4)
#ifdef SF_PHASE_PEVAL
    if (pev_on) atomicAdd(&g_rphase2[5], (unsigned long long)rounds);
#endif
    const bool cyclic = head < n;  // Kahn left nodes unprocessed (rebuild_graph_summary :584-588)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t other = __shfl_xor(mk, o);
        mk = other > mk ? other : mk;
        viol += (uint32_t)__shfl_xor((int)viol, o);
    }
    if (ROFF && lane == 0) ROFF[rounds] = head;
    if (out_info) {  // wave-uniform: wrong-owner items, cyclic flag, Kahn rounds
        out_info[0] = viol;
        out_info[1] = cyclic ? 1u : 0u;
        out_info[2] = rounds;
    }
    PrecResult r;
    r.penalty = pm.const_penalty + (int64_t)viol + (int64_t)(n - total) + (cyclic ? (int64_t)n : 0);
    r.makespan = cyclic ? 0 : (int64_t)mk;
    return r;
}

// ---- incremental trial refresh (HBM scratch) -------------------------------------------------------------------------------
// The reference refreshes the earliest starts of the descendants of the changed edges (refresh_graph_after_route_change,
// list_precedence.rs:494-551) after testing every added edge for a cycle (added_edges_introduce_cycle :638-651).  Same result
// here without touching the replica's lists: a list change / list swap candidate changes the list neighbours of at most six
// nodes (an OVERLAY held one entry per lane), the cycle test is a wave-wide breadth-first search from the head of every added
// edge that prunes by the committed earliest starts (a node that finishes later than every tail of an added edge cannot lead
// back to one: the committed graph is acyclic, so its earliest starts are a potential), and the refresh relaxes frontiers of
// nodes 64 at a time against the view "trial value if stamped by this trial, committed value otherwise".  The committed arrays
// are never written, so there is nothing to undo.  The makespan is the committed one unless every node that attains it moved
// (then one scan of the view).
// MEASURED (profiles/r03f_precedence.txt, MI355X): parity-green, but 4x SLOWER than one full wave-wide Kahn pass per trial at
// 50 x 20 and 100 x 20 (5.6 M vs 22.1 M moves/s with the full pass in LDS) and 9x slower at C4 (500 x 20, 20.8 K vs 187.6 K
// moves/s): a move early in a machine's list shifts most of the downstream schedule, so the refresh runs as many dependent
// frontier rounds as the full pass while every round costs several L2 round trips (stamped view, atomic exchange per push) and
// the search from each added edge walks the whole time window up to the latest tail.  The path stays OPT-IN
// (SF_AMD_PREC_INC=1; the parity tests run it).  What the numbers say is needed instead: one trial per LANE with private
// scratch in HBM (64 independent dependent chains in flight per wave instead of one), see DESIGN.md §8.
struct PrecInc {
    int32_t* E;        // committed earliest starts (prec_eval)
    uint32_t* LS;      // committed list successor
    uint32_t* LP;      // committed list predecessor
    int32_t* ET;       // trial earliest starts (valid where stamp_e == trial)
    uint32_t* SE;
    uint32_t* SQ;
    uint32_t* CH;
    uint32_t* Q1;
    uint32_t* Q2;
    uint32_t trial;    // stamp counters of this launch (wave-uniform)
    uint32_t round;
    int64_t pen_fixed; // const_penalty + unassigned nodes of the committed lists
    uint32_t viol;     // wrong-owner items of the committed lists
    int32_t mk;        // committed makespan (acyclic state)
    uint32_t mk_count; // nodes whose finish equals it
    int32_t ok;        // the committed state is acyclic: its earliest starts are valid
};
__device__ __forceinline__ uint32_t prec_writelane(uint32_t v, int k, uint32_t old) { return (int)(threadIdx.x & 63u) == k ? v : old; }
struct PrecOverlay {   // lane k < n holds one node's new list neighbours
    uint32_t key, pr, su;
    uint32_t n;
    __device__ __forceinline__ int find(uint32_t node) const {
        const uint64_t m = __ballot((threadIdx.x & 63u) < n && key == node);
        return m ? __ffsll((unsigned long long)m) - 1 : -1;
    }
    __device__ __forceinline__ int slot(uint32_t node, const PrecInc& st) {  // find or add (fields start as the committed neighbours)
        int k = find(node);
        if (k >= 0) return k;
        k = (int)n;
        const uint32_t p0 = PrecMemGlobal::ld(st.LP + node), s0 = PrecMemGlobal::ld(st.LS + node);
        key = prec_writelane(node, k, key);
        pr = prec_writelane(p0, k, pr);
        su = prec_writelane(s0, k, su);
        n += 1;
        return k;
    }
    __device__ __forceinline__ uint32_t pred_of(uint32_t node, const PrecInc& st) const {
        const int k = find(node);
        return k >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)pr, k) : PrecMemGlobal::ld(st.LP + node);
    }
    __device__ __forceinline__ uint32_t succ_of(uint32_t node, const PrecInc& st) const {
        const int k = find(node);
        return k >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)su, k) : PrecMemGlobal::ld(st.LS + node);
    }
    __device__ __forceinline__ void set_pred(uint32_t node, uint32_t v, const PrecInc& st) {
        if (node == PREC_NONE) return;
        const int k = slot(node, st);
        pr = prec_writelane(v, k, pr);
    }
    __device__ __forceinline__ void set_succ(uint32_t node, uint32_t v, const PrecInc& st) {
        if (node == PREC_NONE) return;
        const int k = slot(node, st);
        su = prec_writelane(v, k, su);
    }
    // per-lane lookups (divergent node): the overlay entries are read through readlane, <= 8 of them
    __device__ __forceinline__ uint32_t lane_pred(uint32_t node, const PrecInc& st) const {
        uint32_t v = PrecMemGlobal::ld(st.LP + node);
        for (uint32_t k = 0; k < n; ++k)
            if ((uint32_t)__builtin_amdgcn_readlane((int)key, (int)k) == node) v = (uint32_t)__builtin_amdgcn_readlane((int)pr, (int)k);
        return v;
    }
    __device__ __forceinline__ uint32_t lane_succ(uint32_t node, const PrecInc& st) const {
        uint32_t v = PrecMemGlobal::ld(st.LS + node);
        for (uint32_t k = 0; k < n; ++k)
            if ((uint32_t)__builtin_amdgcn_readlane((int)key, (int)k) == node) v = (uint32_t)__builtin_amdgcn_readlane((int)su, (int)k);
        return v;
    }
};

// Committed-state summary after a full evaluation: how many nodes attain the makespan.
__device__ __forceinline__ uint32_t prec_count_makespan(const PrecModel& pm, const int32_t* E, int32_t mk) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t c = 0;
    for (uint32_t i = lane; i < (uint32_t)pm.n; i += 64) c += (PrecMemGlobal::ld(E + i) + pm.dur[i] == mk) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    return c;
}

// One trial: ck = 2 list change (a, i) -> (b, j) (j in pre-removal coordinates), ck = 3 list swap (a, i) <-> (b, j), against
// the committed lists `visits` / `off`.  Returns false when the incremental path does not apply (the caller evaluates fully).
template <class VT>
__device__ __noinline__ bool prec_trial_inc(const PrecModel& pm_ref, PrecInc& st, const VT* visits, const uint32_t* off, int ck, uint32_t a, uint32_t i,
                                            uint32_t b, uint32_t j, PrecResult& out) {
    const PrecModel pm = pm_ref;
    typedef PrecMemGlobal M;
    if (!st.ok || (ck != 2 && ck != 3)) return false;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    const uint32_t oa = off[a], la = off[a + 1] - oa, ob = off[b], lb = off[b + 1] - ob;
    if (i >= la) return false;
    const uint32_t x = (uint32_t)visits[oa + i];
    PrecOverlay ov{PREC_NONE, PREC_NONE, PREC_NONE, 0u};
    int32_t dviol = 0;
    if (ck == 2) {
        if (j > lb) return false;
        if (a == b && (j == i || j == i + 1)) {  // the element stays where it is
            out.penalty = st.pen_fixed + (int64_t)st.viol;
            out.makespan = st.mk;
            return true;
        }
        const uint32_t s = j < lb ? (uint32_t)visits[ob + j] : PREC_NONE;  // x goes in front of s (the end of the list when none)
        const uint32_t p = M::ld(st.LP + x), q = M::ld(st.LS + x);
        ov.set_succ(p, q, st);
        ov.set_pred(q, p, st);
        uint32_t r;  // the element in front of the slot once x is out
        if (s != PREC_NONE)
            r = ov.pred_of(s, st);
        else {
            r = lb ? (uint32_t)visits[ob + lb - 1] : PREC_NONE;
            if (r == x) r = p;
        }
        ov.set_succ(r, x, st);
        ov.set_pred(x, r, st);
        ov.set_succ(x, s, st);
        ov.set_pred(s, x, st);
        if (pm.owner && a != b) {
            const int32_t ow = pm.owner[x];
            dviol = (ow >= 0 && (uint32_t)ow != b ? 1 : 0) - (ow >= 0 && (uint32_t)ow != a ? 1 : 0);
        }
    } else {
        if (j >= lb) return false;
        const uint32_t y = (uint32_t)visits[ob + j];
        if (x == y) {
            out.penalty = st.pen_fixed + (int64_t)st.viol;
            out.makespan = st.mk;
            return true;
        }
        const uint32_t px = M::ld(st.LP + x), qx = M::ld(st.LS + x), py = M::ld(st.LP + y), qy = M::ld(st.LS + y);
        if (qx == y) {  // px x y qy -> px y x qy
            ov.set_succ(px, y, st), ov.set_pred(y, px, st), ov.set_succ(y, x, st), ov.set_pred(x, y, st), ov.set_succ(x, qy, st), ov.set_pred(qy, x, st);
        } else if (qy == x) {  // py y x qx -> py x y qx
            ov.set_succ(py, x, st), ov.set_pred(x, py, st), ov.set_succ(x, y, st), ov.set_pred(y, x, st), ov.set_succ(y, qx, st), ov.set_pred(qx, y, st);
        } else {
            ov.set_succ(px, y, st), ov.set_pred(y, px, st), ov.set_succ(y, qx, st), ov.set_pred(qx, y, st);
            ov.set_succ(py, x, st), ov.set_pred(x, py, st), ov.set_succ(x, qy, st), ov.set_pred(qy, x, st);
        }
        if (pm.owner && a != b) {
            const int32_t ox = pm.owner[x], oy = pm.owner[y];
            dviol = (ox >= 0 && (uint32_t)ox != b ? 1 : 0) - (ox >= 0 && (uint32_t)ox != a ? 1 : 0) + (oy >= 0 && (uint32_t)oy != a ? 1 : 0) -
                    (oy >= 0 && (uint32_t)oy != b ? 1 : 0);
        }
    }
    const int64_t pen_ok = st.pen_fixed + (int64_t)((int32_t)st.viol + dviol);
    // added edges (u -> su'[u] where it differs from the committed successor) and the pruning bound: the latest tail
    const bool mine = lane < ov.n;
    const uint32_t old_su = mine ? M::ld(st.LS + ov.key) : PREC_NONE;
    const bool added = mine && ov.su != PREC_NONE && ov.su != old_su;
    const uint64_t added_m = __ballot(added);
    int32_t tb = added ? M::ld(st.E + ov.key) : INT32_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t other = __shfl_xor(tb, o);
        tb = other > tb ? other : tb;
    }
    const uint64_t tails_m = added_m;
    // ---- cycle test: from the head of every added edge, is its tail reachable in the new graph? ----
    bool cyclic = false;
    for (uint64_t em = added_m; em && !cyclic; em &= em - 1) {
        const int k = __ffsll((unsigned long long)em) - 1;
        const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)ov.key, k), head = (uint32_t)__builtin_amdgcn_readlane((int)ov.su, k);
        st.round += 1;
        const uint32_t mark = st.round;
        uint32_t* cur = st.Q1;
        uint32_t* nxt = st.Q2;
        uint32_t ncur = 1;
        if (lane == 0) {
            M::st(cur, head);
            M::st(st.SQ + head, mark);
        }
        if (head == tail) cyclic = true;
        prec_sync();
        while (ncur && !cyclic) {
            uint32_t nnext = 0;
            for (uint32_t base = 0; base < ncur && !cyclic; base += 64) {
                const bool act = base + lane < ncur;
                uint32_t w = 0, so = 0, deg = 0, ls = PREC_NONE;
                if (act) {
                    w = M::ld(cur + base + lane);
                    so = pm.succ_off[w];
                    deg = pm.succ_off[w + 1] - so;
                    ls = ov.lane_succ(w, st);
                }
                const uint32_t degt = deg + ((act && ls != PREC_NONE) ? 1u : 0u);
                for (uint32_t t = 0;; ++t) {
                    const bool has = t < degt;
                    if (!__ballot(has)) break;
                    bool push = false, hit = false;
                    uint32_t sx = 0;
                    if (has) {
                        sx = t < deg ? pm.succ[so + t] : ls;
                        hit = sx == tail;
                        if (!hit) {
                            // a node that is no tail and finishes after every tail started cannot reach one
                            bool is_tail = false;
                            for (uint64_t tm = tails_m; tm; tm &= tm - 1)
                                is_tail = is_tail || (uint32_t)__builtin_amdgcn_readlane((int)ov.key, __ffsll((unsigned long long)tm) - 1) == sx;
                            const bool prune = !is_tail && M::ld(st.E + sx) + pm.dur[sx] > tb;
                            if (!prune) push = __hip_atomic_exchange(st.SQ + sx, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mark;
                        }
                    }
                    if (__ballot(hit)) cyclic = true;
                    const uint64_t pm_ = __ballot(push);
                    if (push) M::st(nxt + nnext + prec_mbcnt(pm_), sx);
                    nnext += (uint32_t)__popcll(pm_);
                }
            }
            prec_sync();
            uint32_t* t2 = cur;
            cur = nxt;
            nxt = t2;
            ncur = nnext;
        }
    }
    if (cyclic) {
        out.penalty = pen_ok + (int64_t)n;
        out.makespan = 0;
        return true;
    }
    // ---- refresh: the nodes whose list predecessor changed, then whatever their new earliest start reaches ----
    st.trial += 1;
    const uint32_t tr = st.trial;
    uint32_t* cur = st.Q1;
    uint32_t* nxt = st.Q2;
    const bool seed = mine && ov.pr != M::ld(st.LP + ov.key);
    const uint64_t seed_m = __ballot(seed);
    if (seed) M::st(cur + prec_mbcnt(seed_m), ov.key);
    uint32_t ncur = (uint32_t)__popcll(seed_m), nch = 0, lost = 0, rounds = 0;
    prec_sync();
    while (ncur) {
        if (++rounds > n + 2u) {  // cannot happen on an acyclic graph (a longest path has at most n nodes): never spin on the device
            out.penalty = pen_ok + (int64_t)n;
            out.makespan = 0;
            return true;
        }
        st.round += 1;
        const uint32_t mark = st.round;
        uint32_t nnext = 0;
        for (uint32_t base = 0; base < ncur; base += 64) {
            const bool act = base + lane < ncur;
            uint32_t w = 0;
            int32_t ne = 0, oldv = 0;
            bool changed = false, first = false;
            if (act) {
                w = M::ld(cur + base + lane);
                const uint32_t lp = ov.lane_pred(w, st);
                if (lp != PREC_NONE) ne = (M::ld(st.SE + lp) == tr ? M::ld(st.ET + lp) : M::ld(st.E + lp)) + pm.dur[lp];
                for (uint32_t t = pm.pred_off[w]; t < pm.pred_off[w + 1]; ++t) {
                    const uint32_t pp = pm.pred[t];
                    const int32_t f = (M::ld(st.SE + pp) == tr ? M::ld(st.ET + pp) : M::ld(st.E + pp)) + pm.dur[pp];
                    ne = f > ne ? f : ne;
                }
                first = M::ld(st.SE + w) != tr;
                oldv = first ? M::ld(st.E + w) : M::ld(st.ET + w);
                changed = ne != oldv;
                if (changed) {
                    M::st(st.ET + w, ne);
                    M::st(st.SE + w, tr);
                }
            }
            const uint64_t first_m = __ballot(changed && first);
            if (changed && first) {
                M::st(st.CH + nch + prec_mbcnt(first_m), w);
                lost += (oldv + pm.dur[w] == st.mk) ? 1u : 0u;
            }
            nch += (uint32_t)__popcll(first_m);
            uint32_t so = 0, deg = 0, ls = PREC_NONE;
            if (changed) {
                so = pm.succ_off[w];
                deg = pm.succ_off[w + 1] - so;
                ls = ov.lane_succ(w, st);
            }
            const uint32_t degt = deg + ((changed && ls != PREC_NONE) ? 1u : 0u);
            for (uint32_t t = 0;; ++t) {
                const bool has = t < degt;
                if (!__ballot(has)) break;
                bool push = false;
                uint32_t sx = 0;
                if (has) {
                    sx = t < deg ? pm.succ[so + t] : ls;
                    push = __hip_atomic_exchange(st.SQ + sx, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mark;
                }
                const uint64_t pm_ = __ballot(push);
                if (push) M::st(nxt + nnext + prec_mbcnt(pm_), sx);
                nnext += (uint32_t)__popcll(pm_);
            }
        }
        prec_sync();
        uint32_t* t2 = cur;
        cur = nxt;
        nxt = t2;
        ncur = nnext;
    }
    // ---- makespan of the view ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lost += (uint32_t)__shfl_xor((int)lost, o);
    int32_t mk = INT32_MIN;
    for (uint32_t t = lane; t < nch; t += 64) {
        const uint32_t w = M::ld(st.CH + t);
        const int32_t f = M::ld(st.ET + w) + pm.dur[w];
        mk = f > mk ? f : mk;
    }
    if (lost < st.mk_count) {  // some node still attains the committed makespan
        mk = st.mk > mk ? st.mk : mk;
    } else {  // every node that attained it moved: scan the view
        for (uint32_t w = lane; w < n; w += 64) {
            const int32_t f = (M::ld(st.SE + w) == tr ? M::ld(st.ET + w) : M::ld(st.E + w)) + pm.dur[w];
            mk = f > mk ? f : mk;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t other = __shfl_xor(mk, o);
        mk = other > mk ? other : mk;
    }
    out.penalty = pen_ok;
    out.makespan = mk < 0 ? 0 : (int64_t)mk;
    return true;
}

// ---- lane-per-trial sweep (HBM scratch) ------------------------------------------------------------------------------------
// What the incremental numbers above asked for: 64 trials in flight per wavefront instead of one.  Every list change / list swap
// candidate of a replay chunk gets a LANE; all lanes walk the COMMITTED topological order (Kahn's pop order of the last full
// evaluation) together, each recomputing the earliest start of the current node from its own view of the graph -- the fixed
// predecessors (wave-uniform), plus the list predecessor, which differs from the committed one only for the at most six nodes
// of the lane's overlay.  The per-trial earliest starts live node-major in HBM ([node][lane]: one coalesced 256-byte row per
// access); the nodes of one Kahn round are mutually independent, so a round's loads pipeline and the wave synchronises once per
// round.  The sweep starts at the earliest position any lane's change can reach (everything before keeps its committed value,
// the prefix maximum of the finish times gives its share of the makespan).
// Convergence: old edges lead to a LATER Kahn round; only a lane's added edges can lead to the same or an earlier round
// ("backward": the nodes of one round are priced from the values of the rounds before it).  A path with b backward edges is
// exact after b + 1 sweeps (Gauss-Seidel over the rounds), so a lane without backward edges is exact -- and acyclic -- after one
// sweep; a lane with b > 0 backward edges sweeps until nothing changes, and if sweep b + 2 still changes a value the
// added edges close a cycle (every duration is positive, checked on the host, so a cycle never stabilises).
// One order position as the sweep reads it: the node, its duration, and per predecessor (two fixed ones and the committed list
// predecessor; PREC_NONE = absent) the node, its order position, its duration and its committed finish time.  Built at every
// commit, so a batch of nodes costs one record fetch and one round of earliest-start loads instead of a chain of five.
// Address-space typed pointers for the out-of-line sweep: behind a call boundary plain pointers compile to FLAT accesses.
#define PREC_G __attribute__((address_space(1)))
#define PREC_L __attribute__((address_space(3)))
struct PrecRec {
    uint32_t w, dur_w, np, _pad;
    uint32_t p[3], pos[3], dur[3], cfin[3];
};
static_assert(sizeof(PrecRec) == 64, "one cache line per order position");
struct PrecSweep {
    const PREC_G PrecRec* REC;
    const PREC_G int32_t* E;      // committed earliest starts
    const PREC_G uint32_t* LP;    // committed list predecessor / successor
    const PREC_G uint32_t* LS;
    const PREC_G uint32_t* TOPO;  // committed topological order
    const PREC_G uint32_t* POS;
    const PREC_G uint32_t* ROFF;
    const PREC_G uint32_t* RND;
    const PREC_G int32_t* PMAX;
    PREC_G int32_t* EL;
    uint32_t rounds;
    int64_t pen_fixed;
    uint32_t viol;
    int32_t mk;
    int32_t ok;            // committed state acyclic and every duration positive
};
__device__ __forceinline__ uint32_t prec_gld(const PREC_G uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t prec_gld(const PREC_G int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct LaneOverlay {  // one trial's changed list neighbours, in this lane's registers (no dynamic indexing)
    uint32_t key[6], pr[6], su[6];
    uint32_t ppos[6], pdur[6], pcfin[6];  // of pr[k]: order position, duration, committed finish (filled by finish())
    uint32_t n;
    __device__ __forceinline__ void finish(const PrecModel& pm, const PrecSweep& st) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ppos[k] = pdur[k] = pcfin[k] = 0;
            if ((uint32_t)k < n && pr[k] != PREC_NONE) {
                const int32_t dk = ((const PREC_G int32_t*)pm.dur)[pr[k]];
                ppos[k] = prec_gld(st.POS + pr[k]);
                pdur[k] = (uint32_t)dk;
                pcfin[k] = (uint32_t)(prec_gld(st.E + pr[k]) + dk);
            }
        }
    }
    __device__ __forceinline__ void init() {
        n = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) key[k] = pr[k] = su[k] = PREC_NONE;
    }
    __device__ __forceinline__ void touch(uint32_t node, const PrecSweep& st, bool set_p, uint32_t p, bool set_s, uint32_t sv) {
        if (node == PREC_NONE) return;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (!found && (uint32_t)k < n && key[k] == node) {
                found = true;
                if (set_p) pr[k] = p;
                if (set_s) su[k] = sv;
            }
        if (found) return;
        const uint32_t p0 = prec_gld(st.LP + node), s0 = prec_gld(st.LS + node);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if ((uint32_t)k == n) {
                key[k] = node;
                pr[k] = set_p ? p : p0;
                su[k] = set_s ? sv : s0;
            }
        n += 1;
    }
    __device__ __forceinline__ void set_pred(uint32_t node, uint32_t v, const PrecSweep& st) { touch(node, st, true, v, false, 0u); }
    __device__ __forceinline__ void set_succ(uint32_t node, uint32_t v, const PrecSweep& st) { touch(node, st, false, 0u, true, v); }
    __device__ __forceinline__ uint32_t pred_of(uint32_t node, uint32_t committed) const {
        uint32_t v = committed;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if ((uint32_t)k < n && key[k] == node) v = pr[k];
        return v;
    }
};

// Lanes with `cand` hold one candidate each (ck 2 = list change (a, i) -> (b, j), 3 = list swap); the others idle along.
// Returns per lane the constraint's (penalty, makespan) of the trial state.
template <class VT>
__device__ __noinline__ void prec_trial_sweep64(const PrecModel& pm_ref, const PrecSweep& st_ref, const PREC_L VT* visits, const PREC_L uint32_t* off, bool cand, int ck, uint32_t a,
                                                uint32_t i, uint32_t b, uint32_t j, int64_t& out_pen, int64_t& out_mk) {
    const PrecModel pm = pm_ref;
    const PrecSweep st = st_ref;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    LaneOverlay ov;
    ov.init();
    int32_t dviol = 0;
    bool same = !cand;  // the candidate leaves the lists as they are
    if (cand) {
        const uint32_t oa = off[a], la = off[a + 1] - oa, ob = off[b], lb = off[b + 1] - ob;
        const uint32_t x = (uint32_t)visits[oa + i];
        if (ck == 2) {
            if (a == b && (j == i || j == i + 1)) {
                same = true;
            } else {
                const uint32_t s = j < lb ? (uint32_t)visits[ob + j] : PREC_NONE;  // x goes in front of s (the end of the list when none)
                const uint32_t p = prec_gld(st.LP + x), q = prec_gld(st.LS + x);
                ov.set_succ(p, q, st);
                ov.set_pred(q, p, st);
                uint32_t r;  // the element in front of the slot once x is out
                if (s != PREC_NONE)
                    r = ov.pred_of(s, prec_gld(st.LP + s));
                else {
                    r = lb ? (uint32_t)visits[ob + lb - 1] : PREC_NONE;
                    if (r == x) r = p;
                }
                ov.set_succ(r, x, st);
                ov.set_pred(x, r, st);
                ov.set_succ(x, s, st);
                ov.set_pred(s, x, st);
                if (pm.owner && a != b) {
                    const int32_t ow = ((const PREC_G int32_t*)pm.owner)[x];
                    dviol = (ow >= 0 && (uint32_t)ow != b ? 1 : 0) - (ow >= 0 && (uint32_t)ow != a ? 1 : 0);
                }
            }
        } else {
            const uint32_t y = (uint32_t)visits[ob + j];
            if (x == y) {
                same = true;
            } else {
                const uint32_t px = prec_gld(st.LP + x), qx = prec_gld(st.LS + x), py = prec_gld(st.LP + y), qy = prec_gld(st.LS + y);
                if (qx == y) {  // px x y qy -> px y x qy
                    ov.set_succ(px, y, st), ov.set_pred(y, px, st), ov.set_succ(y, x, st), ov.set_pred(x, y, st), ov.set_succ(x, qy, st), ov.set_pred(qy, x, st);
                } else if (qy == x) {  // py y x qx -> py x y qx
                    ov.set_succ(py, x, st), ov.set_pred(x, py, st), ov.set_succ(x, y, st), ov.set_pred(y, x, st), ov.set_succ(y, qx, st), ov.set_pred(qx, y, st);
                } else {
                    ov.set_succ(px, y, st), ov.set_pred(y, px, st), ov.set_succ(y, qx, st), ov.set_pred(qx, y, st);
                    ov.set_succ(py, x, st), ov.set_pred(x, py, st), ov.set_succ(x, qy, st), ov.set_pred(qy, x, st);
                }
                if (pm.owner && a != b) {
                    const int32_t ox = pm.owner[x], oy = pm.owner[y];
                    dviol = (ox >= 0 && (uint32_t)ox != b ? 1 : 0) - (ox >= 0 && (uint32_t)ox != a ? 1 : 0) + (oy >= 0 && (uint32_t)oy != a ? 1 : 0) -
                            (oy >= 0 && (uint32_t)oy != b ? 1 : 0);
                }
            }
        }
    }
    // where the lane's change starts in the committed order, and how many of its added edges point backward
    uint32_t start_pos = n, back = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if ((uint32_t)k < ov.n) {
            const uint32_t node = ov.key[k];
            const uint32_t pk = prec_gld(st.POS + node);
            if (ov.pr[k] != prec_gld(st.LP + node)) start_pos = pk < start_pos ? pk : start_pos;
            if (ov.su[k] != PREC_NONE && ov.su[k] != prec_gld(st.LS + node) &&
                prec_gld(st.RND + ov.su[k]) <= prec_gld(st.RND + node))  // not into a later round: needs a sweep of its own
                back += 1;
        }
    const bool sweeping = !same && start_pos < n;
    uint32_t wstart = sweeping ? start_pos : n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)wstart, o);
        wstart = other < wstart ? other : wstart;
    }
    wstart = (uint32_t)__builtin_amdgcn_readfirstlane((int)wstart);
    const int64_t pen_ok = st.pen_fixed + (int64_t)((int32_t)st.viol + dviol);
    out_pen = pen_ok;
    out_mk = st.mk;
    if (wstart >= n) return;  // no lane changes an earliest start
    // the round that holds position wstart
    uint32_t k0 = 0;
    {
        uint32_t lo = 0, hi = st.rounds;  // ROFF[lo] <= wstart < ROFF[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (prec_gld(st.ROFF + mid) <= wstart)
                lo = mid;
            else
                hi = mid;
        }
        k0 = lo;
    }
    // the committed arrays were written by this wave earlier in the launch: coherent (never scalar-cached) loads (prec_gld)
    PREC_G int32_t* const el = st.EL + lane;  // this lane's column: only this lane ever touches it, program order is all the ordering it needs
    bool done = !sweeping;      // this lane's values are final
    bool cyclic = false;
    int32_t mk = 0;
    ov.finish(pm, st);
    const PREC_G uint32_t* const g_pred = (const PREC_G uint32_t*)pm.pred;
    const PREC_G uint32_t* const g_pred_off = (const PREC_G uint32_t*)pm.pred_off;
    const PREC_G int32_t* const g_dur = (const PREC_G int32_t*)pm.dur;
    constexpr int U = 8;  // nodes of one round priced together: one record fetch, one round of earliest-start loads, then the stores
    for (uint32_t sweep = 1; __ballot(!done); ++sweep) {
        const bool live = !done;
        bool changed = false;
        int32_t smk = prec_gld(st.PMAX + wstart);  // finish times before the sweep window
        for (uint32_t k = k0; k < st.rounds; ++k) {
            const uint32_t q0 = prec_gld(st.ROFF + k), r1 = prec_gld(st.ROFF + k + 1);
            const uint32_t r0 = q0 < wstart ? wstart : q0;
            for (uint32_t t = r0; t < r1; t += U) {
                int32_t ne[U], old[U];
                uint32_t wn[U], wd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {  // the batch's tail repeats the round's last node (same value stored twice)
                    const uint32_t tt = t + (uint32_t)u < r1 ? t + (uint32_t)u : r1 - 1u;
                    const PrecRec rc = *(const PrecRec*)(st.REC + tt);  // wave-uniform 64-byte record (the cast keeps the host pass happy; the device infers global)
                    wn[u] = rc.w;
                    wd[u] = rc.dur_w;
                    int32_t v = 0;
#pragma unroll
                    for (int q = 0; q < 2; ++q)  // fixed predecessors
                        if (rc.p[q] != PREC_NONE) {
                            const int32_t f = rc.pos[q] >= wstart ? el[(size_t)rc.p[q] * 64] + (int32_t)rc.dur[q] : (int32_t)rc.cfin[q];
                            v = f > v ? f : v;
                        }
                    for (uint32_t q = 2; q < rc.np; ++q) {  // more than two fixed predecessors: the general path
                        const uint32_t pp = g_pred[g_pred_off[rc.w] + q];
                        const int32_t f = (prec_gld(st.POS + pp) >= wstart ? el[(size_t)pp * 64] : prec_gld(st.E + pp)) + g_dur[pp];
                        v = f > v ? f : v;
                    }
                    // the list predecessor: the committed one unless this lane's overlay names the node
                    uint32_t lp = rc.p[2], lpos = rc.pos[2], ldur = rc.dur[2], lcf = rc.cfin[2];
#pragma unroll
                    for (int o = 0; o < 6; ++o)
                        if ((uint32_t)o < ov.n && ov.key[o] == rc.w) lp = ov.pr[o], lpos = ov.ppos[o], ldur = ov.pdur[o], lcf = ov.pcfin[o];
                    if (lp != PREC_NONE) {
                        const int32_t f = lpos >= wstart ? el[(size_t)lp * 64] + (int32_t)ldur : (int32_t)lcf;
                        v = f > v ? f : v;
                    }
                    ne[u] = v;
                    old[u] = sweep > 1 ? el[(size_t)rc.w * 64] : 0;
                }
                if (live) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (sweep > 1) changed = changed || old[u] != ne[u];
                        el[(size_t)wn[u] * 64] = ne[u];
                        const int32_t f = ne[u] + (int32_t)wd[u];
                        smk = f > smk ? f : smk;
                    }
                }
            }
        }
        if (live) {
            mk = smk;
            if (back == 0 || (sweep > 1 && !changed)) {
                done = true;
            } else if (sweep >= back + 2u) {  // still moving after every backward edge had its sweep: a cycle
                done = true;
                cyclic = true;
            }
        }
    }
    if (sweeping) {
        out_pen = cyclic ? pen_ok + (int64_t)n : pen_ok;
        out_mk = cyclic ? 0 : (int64_t)mk;
    }
}

}  // namespace sf
