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

namespace sf {

constexpr uint32_t PREC_NONE = 0xFFFFFFFFu;

struct PrecModel {
    int32_t on;  // 0 = the model has no precedence constraint
    int32_t hard_level, mk_level;
    int32_t n;                 // node_count
    const int32_t* dur;        // [n]
    const uint32_t* succ_off;  // [n + 1] fixed successors (valid ones only)
    const uint32_t* succ;
    const int32_t* indeg0;     // [n] fixed in-degree
    const int32_t* owner;      // [n] expected owner, -1 = none; nullptr = no expected-owner hook
    int64_t const_penalty;     // invalid fixed edges (successor >= node_count)
    // per-replica scratch in HBM, [R][n] each
    int32_t* earliest;
    int32_t* indeg;
    uint32_t* queue;
    uint32_t* lsucc;
    int64_t* state;  // [R][2] (hard penalty, makespan) of the committed lists
};

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
};
typedef __attribute__((address_space(3))) int32_t prec_lds_i32;
typedef __attribute__((address_space(3))) uint32_t prec_lds_u32;
struct PrecMemLds {
    typedef prec_lds_i32* I32;
    typedef prec_lds_u32* U32;
    static __device__ __forceinline__ int32_t ld(const prec_lds_i32* p) { return *p; }
    static __device__ __forceinline__ uint32_t ld(const prec_lds_u32* p) { return *p; }
    static __device__ __forceinline__ void st(prec_lds_i32* p, int32_t v) { *p = v; }
    static __device__ __forceinline__ void st(prec_lds_u32* p, uint32_t v) { *p = v; }
    static __device__ __forceinline__ void fmax(prec_lds_i32* p, int32_t v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static __device__ __forceinline__ int32_t fadd(prec_lds_i32* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
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

// Full evaluation of the lists `visits` / `off` (V owners) by one wavefront; E / D / Q / S = the four scratch arrays of pm.n
// words.  Wave-uniform result.
template <class VT, class MEM = PrecMemGlobal>
__device__ __noinline__ PrecResult prec_eval(const PrecModel pm, const VT* visits, const uint32_t* off, int V, typename MEM::I32 E, typename MEM::I32 D,
                                             typename MEM::U32 Q, typename MEM::U32 S) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)pm.n;
    for (uint32_t i = lane; i < n; i += 64) {
        MEM::st(E + i, 0);
        MEM::st(D + i, pm.indeg0[i]);
        MEM::st(S + i, PREC_NONE);
    }
    prec_sync();
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[V]);
    uint32_t viol = 0;
    for (uint32_t t0 = 0; t0 < total; t0 += 64) {  // one list item per lane: its list successor, its in-degree, its owner check
        const uint32_t t = t0 + lane;
        if (t < total) {
            uint32_t lo = 0, hi = (uint32_t)V;  // the owner v with off[v] <= t < off[v + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (off[mid] <= t)
                    lo = mid;
                else
                    hi = mid;
            }
            const uint32_t x = (uint32_t)visits[t];
            if (t + 1 != off[lo + 1]) MEM::st(S + x, (uint32_t)visits[t + 1]);
            if (t != off[lo]) MEM::st(D + x, pm.indeg0[x] + 1);
            if (pm.owner) {
                const int32_t o = pm.owner[x];
                viol += (o >= 0 && (uint32_t)o != lo) ? 1u : 0u;
            }
        }
    }
    prec_sync();
    uint32_t head = 0, tail = 0;
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t i = b + lane;
        const bool ready = i < n && MEM::ld(D + i) == 0;
        const uint64_t m = __ballot(ready);
        if (ready) MEM::st(Q + tail + prec_mbcnt(m), i);
        tail += (uint32_t)__popcll(m);
    }
    prec_sync();
    int32_t mk = 0;
    while (head < tail) {
        const uint32_t cnt = tail - head < 64u ? tail - head : 64u;
        const bool act = lane < cnt;
        int32_t fin = 0;
        uint32_t so = 0, deg = 0, ls = PREC_NONE;
        if (act) {
            const uint32_t node = MEM::ld(Q + head + lane);
            fin = MEM::ld(E + node) + pm.dur[node];
            mk = fin > mk ? fin : mk;
            so = pm.succ_off[node];
            deg = pm.succ_off[node + 1] - so;
            ls = MEM::ld(S + node);
        }
        const uint32_t degt = deg + ((act && ls != PREC_NONE) ? 1u : 0u);
        for (uint32_t k = 0;; ++k) {
            const bool has = k < degt;
            if (!__ballot(has)) break;
            bool newly = false;
            uint32_t s = 0;
            if (has) {
                s = k < deg ? pm.succ[so + k] : ls;
                MEM::fmax(E + s, fin);
                newly = MEM::fadd(D + s, -1) == 1;
            }
            const uint64_t m = __ballot(newly);
            if (newly) MEM::st(Q + tail + prec_mbcnt(m), s);
            tail += (uint32_t)__popcll(m);
        }
        head += cnt;
        prec_sync();
    }
    const bool cyclic = head < n;  // Kahn left nodes unprocessed (rebuild_graph_summary :584-588)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t other = __shfl_xor(mk, o);
        mk = other > mk ? other : mk;
        viol += (uint32_t)__shfl_xor((int)viol, o);
    }
    PrecResult r;
    r.penalty = pm.const_penalty + (int64_t)viol + (int64_t)(n - total) + (cyclic ? (int64_t)n : 0);
    r.makespan = cyclic ? 0 : (int64_t)mk;
    return r;
}

}  // namespace sf
