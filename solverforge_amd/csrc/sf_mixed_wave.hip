// Generic N-leaf search engine (gfx950 wave64): one wavefront = one replica of a model with an
// optional scalar class and an optional list class, and a union of up to twelve leaves -- scalar
// change / swap (plain and nearby), list change / swap, nearby list change / swap, sublist change / swap, list
// reverse (2-opt), 3-opt (full or distance-pruned), list ruin, list permute, the critical-path precedence
// leaf (sf_prec_leaf.h) -- scheduled by the reference's union scheduler (mixed job shop; CVRP under the
// default list policy; job shops under the nine-leaf policy of a slot with precedence hooks).  The two-leaf
// nearby union of the headline bench has its own engines (sf_list_wave.hip / sf_list_kernels.hip).
//
// Reference semantics restated (paths under crates/solverforge-solver/src/):
//   heuristic/selector/list_kernel/change.rs:25-241   list change stream (salts :25-30)
//   heuristic/selector/list_kernel/swap.rs:25-270     list swap stream   (salts :25-31)
//   runtime/compiler/executor/list_leaf/cursor/slot.rs:468-499  entity order without replacement
//   heuristic/selector/scalar_neighborhood/cursor/{change,swap}.rs  scalar streams
//   heuristic/selector/decorator/vec_union.rs:190-365 UnionScheduler::StratifiedRandom, equal weights
//   phase/localsearch/{phase/step.rs,phase/candidates.rs,forager.rs,acceptor/*}  step loop
//
// Every leaf generator walks 64 consecutive OFFSETS of its innermost loop per call (one per
// lane), applies the stream's skip rules and compacts the survivors into the leaf's ring with a
// ballot / mbcnt prefix, so the ring order is the cursor order.  The replay simulates the union
// scheduler pull by pull (wave-uniform scalar code) to lay 64 candidates onto the lanes, scores
// them against the step snapshot and replays acceptor + forager exactly like the other engines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_kopt.h"
#include "sf_list_model.h"
#include "sf_precedence.h"
#include "sf_prec_leaf.h"
#include "sf_prec_group.h"

namespace sf {

constexpr int GL = 12;         // max leaves of the generic union (a slot with precedence hooks declares nine list leaves; + scalar pair)
constexpr uint32_t GRC = 128;  // ring capacity per leaf
#ifndef SF_MIXED_RING_LDS
#define SF_MIXED_RING_LDS 0  // diagnostics: 1 keeps the candidate rings in the replica's LDS slice (the round-2 layout)
#endif
constexpr size_t RUIN_LDS_BYTES = 8 * 8 + 16 * (8 * 2 + 4 * 8) + 128;  // == RuinLds::bytes (sf_ruin.h)

constexpr uint64_t SALT_LC_ENTITY = 0x1157C4A46E000001ULL, SALT_LC_SOURCE = 0x1157C4A46E000002ULL;
constexpr uint64_t SALT_LC_INTRA = 0x1157C4A46E000003ULL, SALT_LC_INTER = 0x1157C4A46E000004ULL;
constexpr uint64_t SALT_LS_ENTITY = 0x11575A0900000001ULL, SALT_LS_FIRST = 0x11575A0900000002ULL;
constexpr uint64_t SALT_LS_SECOND = 0x11575A0900000003ULL, SALT_LS_IFIRST = 0x11575A0900000004ULL;
constexpr uint64_t SALT_LS_ISECOND = 0x11575A0900000005ULL;
constexpr uint64_t SALT_LR_ENTITY = 0x11572A0700000001ULL, SALT_LR_START = 0x11572A0700000002ULL;  // reverse.rs:12-14
constexpr uint64_t SALT_LR_END = 0x11572A0700000003ULL;
constexpr uint64_t SALT_SC_ENTITY = 0x5B157C4A46E00001ULL, SALT_SC_START = 0x5B157C4A46E00002ULL;  // sublist_change.rs:22-28
constexpr uint64_t SALT_SC_SIZE = 0x5B157C4A46E00003ULL, SALT_SC_INTRA = 0x5B157C4A46E00004ULL;
constexpr uint64_t SALT_SC_INTER = 0x5B157C4A46E00005ULL;
constexpr uint64_t SALT_SS_ENTITY = 0x5B1575A090000001ULL, SALT_SS_START = 0x5B1575A090000002ULL;  // sublist_swap.rs:74,89,104
constexpr uint64_t SALT_SS_SIZE = 0x5B1575A090000003ULL;

// nearby scalar leaves (scalar_neighborhood/cursor/change.rs:15-17, cursor/swap.rs:12-14)
constexpr uint64_t SALT_NSC_START = 0xC4A46E00AAAA0001ULL, SALT_NSC_STRIDE = 0xC4A46E00AAAA0002ULL, SALT_NSC_VALUE = 0xC4A46E00AAAA0003ULL;
constexpr uint64_t SALT_NSW_START = 0x5A095CA1AAAA0001ULL, SALT_NSW_STRIDE = 0x5A095CA1AAAA0002ULL, SALT_NSW_TARGET = 0x5A095CA1AAAA0003ULL;

// list permute leaf (selector/list_kernel/permute.rs:16-18; entity order: list_leaf/cursor/slot.rs:281)
constexpr uint64_t SALT_PM_ENTITY = 0x91D79E8A00000001ULL, SALT_PM_START = 0x91D79E8A00000002ULL;
constexpr uint64_t SALT_PM_SIZE = 0x91D79E8A00000003ULL, SALT_PM_ORDER = 0x91D79E8A00000004ULL;
// selector kind of a list leaf -> sf_move_kind of the moves it emits
__device__ __forceinline__ int list_move_kind_of(int k) {
    return (k == 4 || k == 16) ? 2 : ((k == 8 || k == 32) ? 3 : (k == 64 ? 4 : (k == 128 ? 5 : (k == 512 ? 7 : (k == 8192 ? 9 : 6)))));
}
// `ext` argument of apply_list_move_wave for the ring entry (m0, m1, mx) of a leaf of selector kind k
__device__ __forceinline__ uint32_t list_move_ext_of(int k, uint32_t m0, uint32_t m1, uint32_t mx) {
    return k == 512 ? mx : (k == 256 ? ((mx & 15u) | ((mx >> 4) << 16)) : (k == 8192 ? (m1 & 0xFFFFu) : (m0 & 0xFFFFu) + mx));
}

struct RuinParams {  // list ruin leaf (sf_ruin.h)
    int32_t min_count, max_count, moves_per_step, max_source_len;  // max_source_len 0 = None
    int32_t skip_empty;
    uint64_t* rng;  // [R][4] per-solve SmallRng state of the leaf
};

struct GLeaves {
    int32_t n;
    int32_t kind[GL];   // sf_selector_kind: 1 scalar change, 2 scalar swap, 4 list change, 8 list swap, 64 list reverse,
                        // 16 / 32 nearby change / swap, 128 sublist change, 256 sublist swap
    int32_t list_desc;  // descriptor_index of the list class (stream salts)
    int32_t max_nearby[GL];  // nearby leaves (kinds 16 / 32)
    int32_t has_nearby;
    int32_t levels;  // score levels of the model (the kernel is instantiated for 2 or 4)
    int32_t min_size[GL], max_size[GL];  // sublist leaves; k-opt leaf: min_size = min_segment_len
    int32_t kopt_nearby;     // the union has a distance-pruned 3-opt leaf (kind 512, max_nearby > 0)
    uint64_t* kopt_scratch;  // [R][n_cap] distance keys of routes longer than KOPT_LDS_KEYS
    // UnionSelectionOrder + UnionWeighting of the root union (vec_union.rs:190-365): order 0 Sequential, 1 RoundRobin,
    // 2 RotatingRoundRobin, 3 Random, 4 StratifiedRandom; union_custom = 0: the default policy (StratifiedRandom, equal weights)
    int32_t union_order, union_custom;
    int32_t weight[GL];
    int32_t has_ruin;        // the union has a list ruin leaf (kind 1024); parameters + per-solve stream in `ruin`
    RuinParams ruin;
    // nearby scalar leaves (kinds 2048 / 4096): the slot's nearby value / entity sources, one row per entity, already ranked on
    // the host by (distance, source order, candidate) -- the meters are facts, only the state-dependent filter runs here
    const uint32_t* ns_off[2];  // [n + 1]  0 = value candidates (nearby change), 1 = entity candidates (nearby swap)
    const int32_t* ns_val[2];
    int32_t ns_dynamic;         // DynamicScalarVariableSlot: legality re-check (change), directional pairs (swap)
    // candidate rings [R][GL][GRC] of (m0, m1) + one side byte, in HBM (L2-resident): written by the generators with coalesced
    // stores, read once per replay round -- the replica's LDS slice keeps only what is touched with dependent latency
    uint32_t* ring;
    uint8_t* ringx;
    // [R][GL][GRC][2] int32 (capacity-overshoot delta, distance delta; the first word INT32_MIN = not doable): the trial deltas of the list
    // candidates, written by the FAST kernels' scoring stage -- one leaf at a time, so every lane runs the SAME move kind -- and read by the replay
    int32_t* ringd;
    PrecModel prec;          // ListPrecedenceMakespanConstraint of the list class (prec.on; PREC instantiations, sf_precedence.h)
    int32_t prec_lds;        // its scratch arrays are carved from the replica's LDS slice (small node counts)
    int32_t prec_inc;        // HBM scratch: list change / swap trials take the incremental refresh (prec_trial_inc; opt-in, see sf_precedence.h)
    int32_t prec_sweep;      // HBM scratch: the list change / swap trials of a replay chunk are scored 64 at a time (prec_trial_sweep64)
    int32_t prec_static;     // bytes of the workgroup-shared LDS copy of the constraint's static graph (0 = read it from HBM)
    int32_t prec_static_slim;  // that copy holds the node records, fixed in-degrees and owners only (the CSR arrays and durations stay in HBM)
    int32_t prec_groups;     // LDS scratch: trials per wavefront of the grouped evaluator (prec_eval_grouped: 8, 4 or 2; 0 = off)
    PlfModel plf;            // critical-path precedence leaf (kind 16384; PREC instantiations, sf_prec_leaf.h)
    // Join of the two planning classes (SF_C_CROSS_OWNER_MATCH; cross_bi_incremental/incremental.rs:93-137 with A = the scalar class keyed
    // by its value, B = the list owners keyed by their index, filter = the owner's list does not contain the entity): xown_level < 0 = absent.
    // xown_tab [R][n_scalar] u16 in HBM: the list that holds every scalar entity's id (0xFFFF = none) -- what a scalar move's trial reads.
    int32_t xown_level;
    int64_t xown_weight;
    uint16_t* xown_tab;
    uint32_t* node_tab;      // [R][dim] node -> (list << 16 | position) in HBM: the FAST + ruin instantiation keeps it out of the LDS slice (12 replicas per CU)
};

// u16 entries of RuinFast::slot: the slot table of sf_ruin.h (n_cap + V) and the arena of sf_ruin_v2.h, which never overflows at n_cap + 21
// (the lists a candidate changes hold at most n_cap elements together, every newly changed list reserves one entry per element still to place)
__host__ __device__ inline size_t ruin_arena_cap(int n_cap, int V) { return (size_t)n_cap + (size_t)(V > 24 ? V : 24); }
template <class VT>
struct GCarve {
    size_t ring, ringx, load, off, visits, vals, tsum, tcnt, tpt, nstmp, node, slotbase, routeat, rankof, spvec, kopt, ruin, ruin_fast, prec, pgrp, leaftab, total;
    // dim_nearby = node-id bound when the union has nearby leaves (node -> slot table + two leaves'
    // entity-order tables), else 0
    // n_leaves rings only: the LDS slice decides how many replicas a CU holds
    // has_ruin: 0 no ruin leaf, 1 general path (slot prefix only), 2 LDS fast path (+ edge table, list-end edges, matrix row; sf_ruin.h)
    // prec_words: node count of the precedence constraint when its four scratch arrays live in LDS (sf_precedence.h), else 0
    // n_table / run_P: per-value tables of the scalar class's value-keyed constraints (n_values entries; the consecutive-runs /
    // presence table behind the count table, as SCarve lays them out), 0 when the class has none
    // has_ruin 3: the list-preserving recreate only (sf_ruin_v2.h; the FAST instantiation): edge table + list-end edges + arena, no matrix row
    // node_global: the node -> slot table lives in HBM (GLeaves::node_tab)
    __host__ __device__ __forceinline__ GCarve(int n_scalar, int V, int n_cap, int dim_nearby, int kopt_nearby = 0, int n_leaves = GL, int has_ruin = 0, int dim = 0,
                               int prec_words = 0, int n_table = 0, int run_P = 0, int prec_groups = 0, bool node_global = false) {
        size_t o = 0;
        ring = o;  // the candidate rings live in HBM (GLeaves::ring / ringx) unless SF_MIXED_RING_LDS
        o = align_up(o + (SF_MIXED_RING_LDS ? sizeof(uint32_t) * 2 * GRC * n_leaves : 0), 16);
        ringx = o;  // one extra byte per ring entry (segment size of the sublist leaves)
        o = align_up(o + (SF_MIXED_RING_LDS ? GRC * n_leaves : 0), 16);
        node = o;
        o = align_up(o + (node_global ? 0 : sizeof(uint32_t) * dim_nearby), 16);
        slotbase = o;
        o = align_up(o + (dim_nearby ? sizeof(uint16_t) * (V + 1) * 2 : 0), 16);
        routeat = o;
        o = align_up(o + (dim_nearby ? sizeof(uint16_t) * V * 2 : 0), 16);
        rankof = o;
        o = align_up(o + (dim_nearby ? sizeof(uint16_t) * V * 2 : 0), 16);
        spvec = o;
        o = align_up(o + (dim_nearby ? sizeof(uint16_t) * 64 * 2 : 0), 16);
        load = o;
        o = align_up(o + sizeof(int64_t) * V, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        vals = o;
        o = align_up(o + sizeof(VT) * n_scalar, 16);
        tsum = o;
        o = align_up(o + sizeof(int64_t) * n_table, 16);
        tcnt = o;
        o = align_up(o + sizeof(uint32_t) * n_table, 16);
        tpt = o;  // == runs_table(cnt): the aligned end of the count table
        o = align_up(o + (run_P ? runs_table_bytes(n_table, run_P) : 0), 16);
        nstmp = o;  // nearby scalar leaves: the ranked survivors of the row being emitted
        o = align_up(o + (n_scalar ? sizeof(uint16_t) * 64 : 0), 16);
        kopt = o;  // working set of the distance-pruned 3-opt stream
        o = align_up(o + (kopt_nearby ? KoptLds::bytes : 0), 16);
        ruin = o;  // list ruin leaf: streams, candidate table, recreate work area (RuinLds)
        o = align_up(o + (has_ruin ? RUIN_LDS_BYTES + sizeof(uint32_t) * (V + 1) : 0), 16);  // + the slot prefix of a recreate round
        ruin_fast = o;  // edge[dim], row[dim], edge_end[V], slot[n_cap + V]
        o = align_up(o + (has_ruin >= 2 ? sizeof(uint16_t) * ((has_ruin == 2 ? 2 : 1) * (size_t)dim + (size_t)V + ruin_arena_cap(n_cap, V)) : 0), 16);
        prec = o;  // earliest start, in-degree, queue, list successor of the precedence constraint's Kahn pass
        o = align_up(o + prec_lds_scratch_bytes(prec_words), 16);  // (i32, i32, u16, u16)
        pgrp = o;  // grouped trial evaluator (sf_prec_group.h): committed successor / in-degree + per-trial scratch
        o = align_up(o + pgrp_bytes(prec_words, prec_groups, V), 16);
        leaftab = o;  // per-leaf generator / ring / scheduler state (LeafTab)
        o = align_up(o + sizeof(uint32_t) * 16 * GL, 16);
        total = o;
    }
};

// Generator state of one leaf.  Field meaning per kind:
//  scalar change: a = row offset, b = inner offset (value offset incl. the to-None slot)
//  scalar swap:   a = left offset, b = right offset
//  list change:   a = source entity rank, b = source position offset, c = stage (0 intra, 1 inter),
//                 d = destination entity rank (inter), e = destination position offset
//  list swap:     a = entity rank, c = stage, b = first offset, e = second offset, d = destination rank
//  list reverse:  a = entity rank, b = start offset, e = end offset
//  sublist change: a = source rank, b = segment start offset, f = segment size offset, c = stage,
//                 d = destination rank, e = destination position offset
//  sublist swap:  a = first entity rank, b / f = first segment start / size offset, d = second entity rank,
//                 e = second segment start offset window
//  nearby change / swap: a = entity rank, b = offset in the entity's list, c / d = rank / offset base the
//                 leaf's position vector holds, e = sources left
//  3-opt (full):  a = entity rank, b / c = low / high word of the move offset
//  3-opt (distance-pruned): a = entity rank of the NEXT entity to open (the cut state machine lives in LDS)
#ifdef SF_PHASE_PREC
#define PHS(i) PH(0)
#define PHR(i) PH(i)
#else
#define PHS(i) PH(i)
#define PHR(i)
#endif
// workgroup-shared LDS copy of a precedence model's static graph: dur, indeg0, [owner], succ_off, succ, pred_off, pred (32-bit words)
__host__ __device__ inline size_t prec_static_bytes(int n, int n_edges, bool has_owner) {
    return 4 * ((size_t)n * (has_owner ? 3 : 2) + 2 * ((size_t)n + 1) + 2 * (size_t)n_edges + 2 * (size_t)n) + 16;  // + the grouped evaluator's node records
}

struct GGen {
    uint32_t a, b, c, d, e, f;
    int done;
};

// Per-leaf state of one replica, in LDS (16 words per leaf) instead of 8-way unrolled register arrays: the
// kernel stays small enough for several waves per SIMD and the leaf loops index it dynamically.  Every lane
// issues the same (wave-uniform) access, so a value stored by all lanes is what each lane reads back: no
// lane-0 predicate, no fence.
struct LeafTab {
    enum : int { GEN = 0, DONE = 6, HEAD = 7, TAIL = 8, EX = 9, TAKEN = 10, WCUR = 11, KIND = 12, MAXNB = 13, MINSZ = 14, MAXSZ = 15 };
    uint32_t* w;
    __device__ __forceinline__ uint32_t get(int l, int f) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)w[l * 16 + f]); }
    __device__ __forceinline__ int32_t geti(int l, int f) const { return __builtin_amdgcn_readfirstlane((int)w[l * 16 + f]); }
    __device__ __forceinline__ void set(int l, int f, uint32_t v) const { w[l * 16 + f] = v; }
    __device__ __forceinline__ GGen gen(int l) const {
        return GGen{get(l, 0), get(l, 1), get(l, 2), get(l, 3), get(l, 4), get(l, 5), geti(l, DONE)};
    }
    __device__ __forceinline__ void put_gen(int l, const GGen& g) const {
        set(l, 0, g.a), set(l, 1, g.b), set(l, 2, g.c), set(l, 3, g.d), set(l, 4, g.e), set(l, 5, g.f), set(l, DONE, (uint32_t)g.done);
    }
};

// The generators' ring stores (HBM, through the CU's write-through L1) are visible to every lane's loads of the replay:
// workgroup scope = wait for the stores, no cache maintenance (one CU, one L1).
__device__ __forceinline__ void ring_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Lays consecutive groups onto the 64 lanes: lane k holds the size `cnt` of group k (groups in stream order); slot q = lane
// belongs to the last group whose exclusive prefix is <= q.  Returns that group and the offset inside it; `total` = all
// slots of the 64 groups (slots >= min(64, total) are not valid).  One DPP scan + a 6-step shuffle search: a generator
// call emits up to 64 candidates across several short lists instead of one list's handful.
__device__ __forceinline__ void map_slots_to_groups(uint32_t cnt, uint32_t slot, uint32_t& group, uint32_t& offset, uint32_t& total) {
    const uint32_t lane = slot;  // the caller's slot index (usually its lane id; lane / S when a slot spans S lanes)
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t pre = incl - cnt;
    total = (uint32_t)__shfl((int)incl, 63);
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t stepw = 32; stepw; stepw >>= 1) {
        const uint32_t cand = lo + stepw;  // <= 63
        const uint32_t pc = (uint32_t)__shfl((int)pre, (int)cand);
        if (pc <= lane) lo = cand;
    }
    group = lo;
    offset = lane - (uint32_t)__shfl((int)pre, (int)lo);
}

// ---- join of the two planning classes: trial deltas of the match count (an assigned entity whose value is not the list holding it) ----
__device__ __forceinline__ int32_t xown_pen(int32_t v, uint32_t owner) { return (v >= 0 && (uint32_t)v != owner) ? 1 : 0; }
// a scalar change (kind 1: entity m0 takes value m1) or swap (kind 2: entities m0 and m1 exchange their values)
template <class VT>
__device__ __forceinline__ int32_t xown_scalar_delta(int kind, uint32_t m0, uint32_t m1, const VT* vals, const uint16_t* owner_tab) {
    if (kind == 1) {
        const uint32_t ow = owner_tab[m0];
        return xown_pen((int32_t)m1, ow) - xown_pen((int32_t)vals[m0], ow);
    }
    const uint32_t o1 = owner_tab[m0], o2 = owner_tab[m1];
    const int32_t v1 = (int32_t)vals[m0], v2 = (int32_t)vals[m1];
    return xown_pen(v2, o1) - xown_pen(v1, o1) + xown_pen(v1, o2) - xown_pen(v2, o2);
}
// a list move in ring coordinates: the elements that change lists (intra-list moves change nothing)
template <class VT>
__device__ __forceinline__ int32_t xown_list_delta(int kind, uint32_t m0, uint32_t m1, uint32_t mx, const uint16_t* visits, const uint32_t* off, const VT* vals) {
    const uint32_t a = m0 >> 16, i = m0 & 0xFFFFu, b = m1 >> 16, j = m1 & 0xFFFFu;
    if (a == b || kind == 64 || kind == 512 || kind == 8192) return 0;
    auto moved = [&](uint32_t e, uint32_t from, uint32_t to) { const int32_t v = (int32_t)vals[e]; return xown_pen(v, to) - xown_pen(v, from); };
    if (kind == 4 || kind == 16) return moved(visits[off[a] + i], a, b);
    if (kind == 8 || kind == 32) return moved(visits[off[a] + i], a, b) + moved(visits[off[b] + j], b, a);
    int32_t d = 0;
    if (kind == 128) {  // segment [i, i + mx) of a -> b
        for (uint32_t t = 0; t < mx; ++t) d += moved(visits[off[a] + i + t], a, b);
    } else if (kind == 256) {  // [i, i + (mx & 15)) of a <-> [j, j + (mx >> 4)) of b
        for (uint32_t t = 0; t < (mx & 15u); ++t) d += moved(visits[off[a] + i + t], a, b);
        for (uint32_t t = 0; t < (mx >> 4); ++t) d += moved(visits[off[b] + j + t], b, a);
    }
    return d;
}

}  // namespace sf
#include "sf_ruin.h"
#include "sf_ruin_v2.h"
namespace sf {

// workgroups of 4 waves: 2 resident workgroups per CU = 2 waves per SIMD (<= 256 VGPRs)
#ifndef SF_MIXED_BLOCKS_PER_CU
#define SF_MIXED_BLOCKS_PER_CU 2
#endif
// RUIN = the union has a list ruin leaf: its own instantiation, so unions without one keep their register allocation
// PREC = the list class carries a ListPrecedenceMakespanConstraint: every doable list candidate of a chunk is applied to the LDS
// lists in turn, scored by one full wave-wide evaluation (prec_eval) and undone from the committed copy in HBM
// MODE 1 (FAST): compile-time specialisation for the reference's default LIST policy -- a list-only model, the default root
// union (StratifiedRandom, equal weights), LateAcceptance + AcceptedCount, committed untraced steps with generated step seeds,
// the unified trial delta (symmetric matrix).  The scalar leaves, the other acceptors / foragers / union orders, the dry run and
// the trace compile out: fewer wave-uniform values stay live across the step loop (the general instantiation spills ~900 SGPRs),
// so the kernel fits more waves per SIMD (SF_MIXED_FAST_BLOCKS_PER_CU; the RUIN instantiation keeps 2: its recreate spills 1.3 KB of
// scratch per lane at 168 VGPRs and ran 1.6x slower, profiles/r03e).  Same decisions bit for bit.
// diagnostics (register-pressure bisection): -DSF_DBG_KINDS=<mask> compiles the generators of the masked leaf kinds out
// MODE 1 additionally compiles out the generators / evaluators no default list policy of a slot with a distance meter declares (round 4):
// plain list change / swap (4, 8), the full 3-opt enumeration (the `1` bit below), list permute (8192) -- the host takes FAST only for
// unions of nearby change / swap, sublist change / swap, reverse, distance-pruned 3-opt and ruin (launch_mixed_t).
#ifdef SF_DBG_KINDS
#define DBGK(k) ((((SF_DBG_KINDS) & (k)) == 0) && (!FAST || ((k) & (16 | 32 | 64 | 128 | 256 | 512)) != 0))
#else
#define DBGK(k) (!FAST || ((k) & (16 | 32 | 64 | 128 | 256 | 512)) != 0)
#endif
#ifndef SF_MIXED_FAST_BLOCKS_PER_CU
#define SF_MIXED_FAST_BLOCKS_PER_CU 4
#endif
// MODE 2 (PREC instantiations, untraced): the same code built for four 4-wave workgroups per CU (128 registers per lane).  A precedence
// trial is one replica's serial chain of Kahn rounds, so a CU that can hold more than eight small replicas (LDS slice permitting) hides
// that latency with more of them: nine-leaf policy on a 20 x 10 job shop 38.3 M moves/s at 2,048 replicas -> 54.7 M at 4,096, 10 x 5
// 102.7 M (profiles/r03w_prec_occupancy.txt).  LDS-bound sizes (50 x 20: eight replicas per CU either way) keep MODE 0: -4 % with the
// smaller register budget.  The host picks per launch (launch_mixed_t).
#ifndef SF_MIXED_PREC_BLOCKS_PER_CU
#define SF_MIXED_PREC_BLOCKS_PER_CU 4
#endif
template <int L, bool TRACE, class VT, bool RUIN = false, bool PREC = false, int MODE = 0>
// the FAST kernels without a ruin leaf keep their node -> slot table in HBM too (1) or in the LDS slice (0)
#ifndef SF_MIXED_FAST_NODEG
#define SF_MIXED_FAST_NODEG 1  // (round 5: with the 128-register build 16 replicas share a CU: six-leaf CVRP-1000 9.1 -> 10.0 G moves/s sustained)
#endif
#ifndef SF_MIXED_FAST_RUIN_BLOCKS_PER_CU
#define SF_MIXED_FAST_RUIN_BLOCKS_PER_CU 3  // (round 5: 168 registers, 91 spilled values -- like the kernel without the leaf; LDS slice 13.4 KB at CVRP-1000)
#endif
__global__ __launch_bounds__(256, MODE == 1 ? (RUIN ? SF_MIXED_FAST_RUIN_BLOCKS_PER_CU : SF_MIXED_FAST_BLOCKS_PER_CU) : (MODE == 2 ? SF_MIXED_PREC_BLOCKS_PER_CU : SF_MIXED_BLOCKS_PER_CU)) void k_mixed_search_wave(
    ListModel lm, ScalarModel sm, GLeaves gl, SearchParams p, int has_list_arg, int has_scalar_arg, NbrIndex nb) {
    constexpr bool FAST = MODE == 1;
    static_assert(!FAST || (!TRACE && !PREC), "FAST: untraced, no precedence constraint");
    const int has_list = FAST ? 1 : has_list_arg, has_scalar = FAST ? 0 : has_scalar_arg;
    const int acceptor = FAST ? 1 : p.acceptor, forager = FAST ? 0 : p.forager, dry_run = FAST ? 0 : p.dry_run;
    const int union_custom = FAST ? 0 : gl.union_custom;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // wave-uniform by construction; saying so keeps the per-replica base pointers (and everything derived from the replica index) in
    // scalar registers instead of 64-bit VGPR pairs (sf_list_wave.hip: 96 -> 77 VGPRs, scratch 120 B -> 0)
    const uint32_t wave_in_group = uni(threadIdx.x >> 6);
    const int rr = (int)(blockIdx.x * (blockDim.x >> 6) + wave_in_group);
    if (rr >= p.n_launch) return;  // no workgroup barrier below
    const int r = rr + p.replica_base;
    __shared__ uint64_t s_sa[FAST ? 1 : 4][FAST ? 1 : SA_WORDS];  // SimulatedAnnealing acceptor state of the resident replicas (FAST: LateAcceptance, no static LDS)
    uint64_t* saw = s_sa[FAST ? 0 : wave_in_group];
    const bool annealing = acceptor == 3;
    if (annealing) sa_load(saw, p.sa, r, lane);
    const uint32_t ns = has_scalar ? (uint32_t)sm.n : 0u;
    const int V = has_list ? lm.V : 0;
    const bool has_nearby = gl.has_nearby != 0;
    const bool unified_eval = FAST || (has_list && (lm.mat_symmetric != 0 || lm.dist_level < 0) && !p.legacy_eval);
    const bool tables = !FAST && has_scalar && sm.tables();  // value-keyed constraints of the scalar class: per-value tables in LDS
    // FAST + ruin: only the list-preserving recreate (no matrix row in LDS) and the node -> slot table in HBM: 19.4 -> 13.4 KB per replica at
    // CVRP-1000, twelve replicas per CU with the 168-register build instead of eight
    constexpr bool NODEG = FAST && (RUIN || SF_MIXED_FAST_NODEG != 0);
    const GCarve<VT> cv((int)ns, V, has_list ? lm.n_cap : 0, has_nearby ? lm.dim : 0, gl.kopt_nearby, gl.n, RUIN ? (FAST ? 3 : (lm.leg16 ? 2 : 1)) : 0, lm.dim,
                        PREC && gl.prec_lds ? gl.prec.n : 0, tables ? sm.n_values : 0, tables && sm.run_level >= 0 ? sm.run_P : 0,
                        PREC && gl.prec_lds ? gl.prec_groups : 0, NODEG);
    unsigned char* mem = smem + (size_t)wave_in_group * cv.total;
    PH_DECL
    PgrpStatic pgs{};  // the same arrays behind typed LDS pointers
    if (PREC && gl.prec_static) {  // every wave writes the same words (no workgroup barrier: a wave may have returned above)
        uint32_t* sh = (uint32_t*)(smem + (size_t)(blockDim.x >> 6) * cv.total);
        const uint32_t n = (uint32_t)gl.prec.n, m = (uint32_t)gl.prec.n_edges;
        auto take = [&](const void* src, uint32_t words) -> const void* {
            uint32_t* dst = sh;
            for (uint32_t t = lane; t < words; t += 64) dst[t] = ((const uint32_t*)src)[t];
            sh += words;
            return dst;
        };
        const bool slim = gl.prec_static_slim != 0;
        if (!slim) gl.prec.dur = (const int32_t*)take(gl.prec.dur, n);
        gl.prec.indeg0 = (const int32_t*)take(gl.prec.indeg0, n);
        if (gl.prec.owner) gl.prec.owner = (const int32_t*)take(gl.prec.owner, n);
        if (!slim) {
            gl.prec.succ_off = (const uint32_t*)take(gl.prec.succ_off, n + 1);
            gl.prec.succ = (const uint32_t*)take(gl.prec.succ, m);
            gl.prec.pred_off = (const uint32_t*)take(gl.prec.pred_off, n + 1);
            gl.prec.pred = (const uint32_t*)take(gl.prec.pred, m);
        }
        gl.prec.nd = (const uint32_t*)take(gl.prec.nd, 2 * n);
        pgs.nd = (const pg_lds_u32*)gl.prec.nd;
        pgs.succ_off = (const pg_lds_u32*)gl.prec.succ_off, pgs.succ = (const pg_lds_u32*)gl.prec.succ;
        pgs.indeg0 = (const pg_lds_i32*)gl.prec.indeg0, pgs.owner = (const pg_lds_i32*)gl.prec.owner, pgs.has_owner = gl.prec.owner != nullptr ? 1u : 0u;
        wave_sync();
    }
    // ONE private copy of the constraint's parameter blocks for the out-of-line stages (prec_eval, plf_*, prec_trial_*): they take them by
    // reference.  By value, each of their ~60 inlined call sites built its own 240-byte copy on the stack -- round 5's 5.1 - 5.9 KB of scratch
    // per lane were mostly those copies, not spilled registers.
    const PrecModel precm = gl.prec;
    const PlfModel plfm = gl.plf;
    uint32_t* ring = SF_MIXED_RING_LDS ? (uint32_t*)(mem + cv.ring) : gl.ring + (size_t)r * GL * GRC * 2;  // [leaf][GRC][2]
    uint8_t* ringx = SF_MIXED_RING_LDS ? (uint8_t*)(mem + cv.ringx) : gl.ringx + (size_t)r * GL * GRC;   // [leaf][GRC]
    // Scoring stage of the FAST kernels (round 6): the trial deltas of the list candidates are computed right after a fill round, ONE LEAF AT A
    // TIME -- 64 lanes of the same move kind -- and parked beside the ring; the replay, whose 64 pulls interleave six or seven kinds (every kind's
    // code ran there with a sixth of the lanes), only reads them.  Needs 32-bit deltas (ListModel::small32; the host passes ringd only then).
    int32_t* const ringd = (FAST && gl.ringd) ? gl.ringd + (size_t)r * GL * GRC * 2 : nullptr;
    const bool pre_eval = FAST && ringd != nullptr;
    uint32_t prev_pulls = 0;  // pulls of the replica's previous step in this launch: long steps fill (and score) their rings in lumps of ~64 per leaf
    int64_t* s_load = (int64_t*)(mem + cv.load);
    uint32_t* s_off = (uint32_t*)(mem + cv.off);
    uint16_t* s_visits = (uint16_t*)(mem + cv.visits);
    VT* s_vals = (VT*)(mem + cv.vals);
    uint16_t* ns_tmp = (uint16_t*)(mem + cv.nstmp);
    int64_t* t_sum = (int64_t*)(mem + cv.tsum);  // per-value summed size / entity count of the working state
    uint32_t* t_cnt = (uint32_t*)(mem + cv.tcnt);
    uint32_t* node_slot = NODEG ? gl.node_tab + (size_t)r * lm.dim : (uint32_t*)(mem + cv.node);
    uint16_t* nb_slot_base = (uint16_t*)(mem + cv.slotbase);  // [nearby leaf 0/1][V+1]
    uint16_t* nb_route_at = (uint16_t*)(mem + cv.routeat);
    uint16_t* nb_rank_of = (uint16_t*)(mem + cv.rankof);
    uint16_t* nb_spvec = (uint16_t*)(mem + cv.spvec);
    const bool tracing = TRACE && r == p.trace_replica;
    const int nl = gl.n;
    const LeafTab lt{(uint32_t*)(mem + cv.leaftab)};
#pragma unroll
    for (int l = 0; l < GL; ++l) {  // constants of the leaves (compile-time l: no dynamic indexing of the kernarg block)
        lt.set(l, LeafTab::KIND, (uint32_t)gl.kind[l]);
        lt.set(l, LeafTab::MAXNB, (uint32_t)gl.max_nearby[l]);
        lt.set(l, LeafTab::MINSZ, (uint32_t)gl.min_size[l]);
        lt.set(l, LeafTab::MAXSZ, (uint32_t)gl.max_size[l]);
    }
    const uint64_t identity = ((uint64_t)(uint32_t)sm.descriptor << 32) ^ (uint64_t)(uint32_t)sm.variable;
    const uint32_t vc = (uint32_t)sm.n_values;
    const FastMod fm_n = make_fastmod(ns), fm_vc = make_fastmod(vc);  // fixed divisors of the scalar streams
    const uint64_t ldesc = (uint64_t)(uint32_t)gl.list_desc;

    uint32_t* g_visits = has_list ? lm.visits + (size_t)r * lm.n_cap : nullptr;
    uint32_t* g_off = has_list ? lm.off + (size_t)r * (V + 1) : nullptr;
    int64_t* g_load = has_list ? lm.load + (size_t)r * V : nullptr;
    int32_t* g_vals = has_scalar ? sm.vals + (size_t)r * ns : nullptr;
    // the committed score lives with the list model when there is one
    int64_t* g_score = (has_list ? lm.score : sm.score) + (size_t)r * 4;
    int64_t* g_best_score = (has_list ? lm.best_score : sm.best_score) + (size_t)r * 4;

    if (has_list) {
        for (uint32_t t = lane; t <= (uint32_t)V; t += 64) s_off[t] = g_off[t];
        for (uint32_t t = lane; t < (uint32_t)V; t += 64) s_load[t] = g_load[t];
        wave_sync();
        const uint32_t tot = uni(s_off[V]);
        for (uint32_t t = lane; t < tot; t += 64) s_visits[t] = (uint16_t)g_visits[t];
    }
    for (uint32_t t = lane; t < ns; t += 64) s_vals[t] = (VT)g_vals[t];
    if (tables)
        for (uint32_t v = lane; v < (uint32_t)sm.n_values; v += 64) {
            t_sum[v] = 0;
            t_cnt[v] = 0;
        }
    wave_sync();
    if (tables) {
        scalar_tables_accumulate(sm, s_vals, lane, 64u, t_cnt, t_sum);
        wave_sync();
    }
    if (has_nearby) {  // node -> (route << 16 | position)
        for (uint32_t t = lane; t < (uint32_t)lm.dim; t += 64) node_slot[t] = NODE_NONE;
        wave_sync();
        for (uint32_t v = lane; v < (uint32_t)V; v += 64) {
            const uint32_t o = s_off[v], len = s_off[v + 1] - o;
            for (uint32_t q = 0; q < len; ++q) node_slot[s_visits[o + q]] = (v << 16) | q;
        }
        wave_sync();
        if (NODEG) ring_sync();  // (HBM table: through the CU's write-through L1)
    }

    // join of the two planning classes: entity -> the list that holds it (HBM, L2-resident; through the CU's write-through L1)
    const bool xown_on = !FAST && has_list && has_scalar && gl.xown_level >= 0;
    uint16_t* const xown = xown_on ? gl.xown_tab + (size_t)r * ns : nullptr;
    if (xown_on) {
        for (uint32_t t = lane; t < ns; t += 64) xown[t] = (uint16_t)0xFFFFu;
        ring_sync();
        for (uint32_t v = lane; v < (uint32_t)V; v += 64) {
            const uint32_t o = s_off[v], len = s_off[v + 1] - o;
            for (uint32_t q = 0; q < len; ++q)
                if ((uint32_t)s_visits[o + q] < ns) xown[s_visits[o + q]] = (uint16_t)v;
        }
        ring_sync();
    }

    const RuinLds rl(mem + cv.ruin);
    uint32_t* ruin_sbase = (uint32_t*)(mem + cv.ruin + RUIN_LDS_BYTES);
    RuinFast rfast{nullptr, nullptr, nullptr, nullptr};
    if (RUIN && (FAST || lm.leg16)) {
        rfast.edge = (uint16_t*)(mem + cv.ruin_fast);
        rfast.row = FAST ? nullptr : rfast.edge + lm.dim;  // (FAST: sf_ruin_v2.h reads the legs from the matrix rows)
        rfast.edge_end = rfast.edge + (FAST ? 1 : 2) * lm.dim;
        rfast.slot = rfast.edge_end + V;
    }
    if (RUIN) {  // the leaf's per-solve stream lives in LDS for the launch
        if (lane < 4) rl.prng[lane] = gl.ruin.rng[(size_t)r * 4 + lane];
        wave_sync();
    }

    int64_t cur[L], best_sol[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        cur[k] = g_score[k];
        best_sol[k] = g_best_score[k];
    }
    // committed (hard penalty, makespan) of the precedence constraint; the trial deltas are taken against it
    int64_t prec_pen = 0, prec_mk = 0;
    const bool prec_in_lds = PREC && gl.prec_lds != 0;
    int32_t* const prec_E = !PREC ? nullptr : (prec_in_lds ? (int32_t*)(mem + cv.prec) : gl.prec.earliest + (size_t)r * gl.prec.n);
    int32_t* const prec_D = !PREC ? nullptr : (prec_in_lds ? prec_E + gl.prec.n : gl.prec.indeg + (size_t)r * gl.prec.n);
    // queue and list successor: 16-bit arrays behind the in-degrees when the scratch lives in LDS (PrecMemLds), 32-bit rows in HBM otherwise
    uint16_t* const prec_Q16 = (PREC && prec_in_lds) ? (uint16_t*)(prec_D + gl.prec.n) : nullptr;
    uint16_t* const prec_S16 = (PREC && prec_in_lds) ? prec_Q16 + gl.prec.n : nullptr;
    uint32_t* const prec_Q = (!PREC || prec_in_lds) ? nullptr : gl.prec.queue + (size_t)r * gl.prec.n;
    uint32_t* const prec_S = (!PREC || prec_in_lds) ? nullptr : gl.prec.lsucc + (size_t)r * gl.prec.n;
    // incremental trial refresh (sf_precedence.h: prec_trial_inc) when the scratch lives in HBM: list change / swap candidates are
    // scored against the committed earliest starts without applying them; everything else takes the full evaluation
    PrecInc pinc{};
    const bool prec_incremental = PREC && !prec_in_lds && gl.prec_inc != 0 && gl.prec.lpred != nullptr;
    if (prec_incremental) {
        const size_t pn = (size_t)gl.prec.n, pb = (size_t)r * pn;
        pinc.E = prec_E, pinc.LS = prec_S, pinc.ET = prec_D, pinc.Q1 = prec_Q;
        pinc.LP = gl.prec.lpred + pb, pinc.SE = gl.prec.stamp_e + pb, pinc.SQ = gl.prec.stamp_q + pb, pinc.CH = gl.prec.changed + pb;
        pinc.Q2 = gl.prec.queue2 + pb;
        for (uint32_t t = lane; t < (uint32_t)pn; t += 64) {  // the stamps of earlier launches mean nothing here
            pinc.SE[t] = 0;
            pinc.SQ[t] = 0;
        }
    }
    const bool prec_sweep = PREC && !prec_in_lds && !prec_incremental && gl.prec_sweep != 0 && gl.prec.elane != nullptr;
    PrecSweep psw{};
    uint32_t* const psw_lp = prec_sweep ? gl.prec.lpred + (size_t)r * gl.prec.n : nullptr;
    uint32_t* const psw_pos = prec_sweep ? gl.prec.pos + (size_t)r * gl.prec.n : nullptr;
    uint32_t* const psw_roff = prec_sweep ? gl.prec.roff + (size_t)r * ((size_t)gl.prec.n + 1) : nullptr;
    int32_t* const psw_pmax = prec_sweep ? gl.prec.pmax + (size_t)r * ((size_t)gl.prec.n + 1) : nullptr;
    uint32_t* const psw_rnd = prec_sweep ? gl.prec.rnd + (size_t)r * gl.prec.n : nullptr;
    PrecRec* const psw_rec = prec_sweep ? (PrecRec*)gl.prec.rec + (size_t)r * gl.prec.n : nullptr;
    // one full evaluation of the lists in LDS: typed LDS accessors when the scratch lives there
    auto prec_run = [&]() -> PrecResult {
        if (prec_in_lds) {  // (nobody reads the pop order of this evaluation: the leaf and the recreate run plf_eval)
            if (gl.prec_static)
                return prec_eval<uint16_t, PrecMemLds, false, true>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D,
                                                                    (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16);
            return prec_eval<uint16_t, PrecMemLds, false, false>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D,
                                                                 (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16);
        }
        if (prec_sweep) {  // committed evaluation + what the lane-per-trial sweep reads: list predecessors, order positions, round starts, prefix maxima
            __shared__ uint32_t s_psw_info[4][4];
            uint32_t* info = s_psw_info[wave_in_group];
            const PrecResult pr = prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, prec_E, prec_D, prec_Q, prec_S, psw_lp, info, psw_roff);
            wave_sync();
            const uint32_t pn = (uint32_t)gl.prec.n;
            psw.viol = uni(info[0]);
            psw.ok = uni(info[1]) == 0u && !gl.prec.has_zero_duration;
            psw.rounds = uni(info[2]);
            psw.mk = (int32_t)pr.makespan;
            psw.pen_fixed = pr.penalty - (int64_t)psw.viol - (uni(info[1]) ? (int64_t)pn : 0);
            prec_sync();
            if (psw.ok) {
                for (uint32_t t = lane; t < pn; t += 64) {
                    const uint32_t w = PrecMemGlobal::ld(prec_Q + t);
                    psw_pos[w] = t;
                    uint32_t lo = 0, hi = psw.rounds;  // the round whose range holds position t
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (PrecMemGlobal::ld(psw_roff + mid) <= t)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    psw_rnd[w] = lo;
                }
                int32_t carry = 0;  // prefix maximum of the finish times along the order
                for (uint32_t base = 0; base < pn; base += 64) {
                    const uint32_t t = base + lane;
                    int32_t f = 0;
                    if (t < pn) {
                        const uint32_t w = PrecMemGlobal::ld(prec_Q + t);
                        f = PrecMemGlobal::ld(prec_E + w) + gl.prec.dur[w];
                    }
                    int32_t inc = f;  // inclusive max scan over the 64 lanes
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int32_t up = __shfl_up(inc, o);
                        if ((int)lane >= o) inc = up > inc ? up : inc;
                    }
                    inc = inc > carry ? inc : carry;
                    const int32_t excl = lane == 0 ? carry : (int32_t)__shfl_up(inc, 1);
                    if (t < pn) psw_pmax[t] = lane == 0 ? carry : excl;
                    carry = (int32_t)__shfl(inc, 63);
                }
                if (lane == 0) psw_pmax[pn] = carry;
                prec_sync();
                for (uint32_t t = lane; t < pn; t += 64) {  // the sweep records, one per order position
                    const uint32_t w = PrecMemGlobal::ld(prec_Q + t);
                    PrecRec rc;
                    rc.w = w;
                    rc.dur_w = (uint32_t)gl.prec.dur[w];
                    const uint32_t po = gl.prec.pred_off[w];
                    rc.np = gl.prec.pred_off[w + 1] - po;
                    rc._pad = 0;
                    const uint32_t lpc = PrecMemGlobal::ld(psw_lp + w);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const uint32_t pp = q < 2 ? ((uint32_t)q < rc.np ? gl.prec.pred[po + q] : PREC_NONE) : lpc;
                        rc.p[q] = pp;
                        rc.pos[q] = rc.dur[q] = rc.cfin[q] = 0;
                        if (pp != PREC_NONE) {
                            rc.pos[q] = PrecMemGlobal::ld(psw_pos + pp);
                            rc.dur[q] = (uint32_t)gl.prec.dur[pp];
                            rc.cfin[q] = (uint32_t)(PrecMemGlobal::ld(prec_E + pp) + gl.prec.dur[pp]);
                        }
                    }
                    psw_rec[t] = rc;
                }
                prec_sync();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the records are read with plain loads: drop what the L1 holds of the last commit's
            }
            psw.E = (const PREC_G int32_t*)prec_E, psw.LP = (const PREC_G uint32_t*)psw_lp, psw.LS = (const PREC_G uint32_t*)prec_S;
            psw.TOPO = (const PREC_G uint32_t*)prec_Q, psw.POS = (const PREC_G uint32_t*)psw_pos, psw.ROFF = (const PREC_G uint32_t*)psw_roff;
            psw.RND = (const PREC_G uint32_t*)psw_rnd, psw.PMAX = (const PREC_G int32_t*)psw_pmax, psw.REC = (const PREC_G PrecRec*)psw_rec;
            psw.EL = (PREC_G int32_t*)(gl.prec.elane + (size_t)r * gl.prec.n * 64);
            return pr;
        }
        if (!prec_incremental) return prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, prec_E, prec_D, prec_Q, prec_S);
        // committed evaluation: also the list predecessors, the owner violations, the cycle flag and the makespan multiplicity
        __shared__ uint32_t s_prec_info[4][2];
        uint32_t* info = s_prec_info[wave_in_group];
        const PrecResult pr = prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, prec_E, prec_D, prec_Q, prec_S, pinc.LP, info);
        wave_sync();
        pinc.viol = uni(info[0]);
        pinc.ok = uni(info[1]) == 0u;
        pinc.mk = (int32_t)pr.makespan;
        pinc.pen_fixed = pr.penalty - (int64_t)pinc.viol - (pinc.ok ? 0 : (int64_t)gl.prec.n);
        prec_sync();
        pinc.mk_count = pinc.ok ? uni(prec_count_makespan(gl.prec, prec_E, pinc.mk)) : 0u;
        return pr;
    };
    // full evaluation of a TRIAL state (move kinds the incremental refresh does not cover): its own scratch, so the committed
    // earliest starts / list neighbours survive (trial values, both frontiers and the changed list are free between trials)
    auto prec_run_trial = [&]() -> PrecResult {
        if (prec_sweep) {  // (earliest, in-degree, queue, list successor) that are no part of the committed summary
            const size_t pb = (size_t)r * gl.prec.n;
            return prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, (int32_t*)(gl.prec.changed + pb), prec_D, gl.prec.stamp_q + pb,
                                                      gl.prec.queue2 + pb);
        }
        if (!prec_incremental) return prec_run();
        return prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, (int32_t*)pinc.CH, prec_D, prec_Q, pinc.Q2);
    };
    if (PREC) {
        const PrecResult pr = prec_run();
        prec_pen = pr.penalty;
        prec_mk = pr.makespan;
    }
    // grouped trial evaluator (sf_prec_group.h): T candidates of a replay chunk per pass, G = 64 / T lanes each
    const uint32_t pgrp_T = (PREC && prec_in_lds && gl.prec_static && !gl.prec_static_slim) ? (uint32_t)gl.prec_groups : 0u;
    const uint32_t pgrp_shift = pgrp_T ? (uint32_t)__builtin_ctz(64u / pgrp_T) : 6u;
    uint32_t pgrp_viol = 0;   // wrong-owner items of the committed lists
    uint32_t pgrp_ready = 0;  // nodes without a predecessor in the committed lists
    // one pass: every lane group scores the move it holds (gm uniform inside a group, kind 0 = idle) against the committed lists
    auto pgrp_eval = [&](const PgrpMove& gm, int64_t& gp, int64_t& gmk, bool& gcyc) {
        prec_eval_grouped<uint16_t>(pgs, (uint32_t)gl.prec.n, V, (const PREC_L uint16_t*)s_visits, (const PREC_L uint32_t*)s_off, mem + cv.pgrp, pgrp_shift, gm,
                                    gl.prec.const_penalty + (int64_t)((uint32_t)gl.prec.n - uni(s_off[V])), pgrp_viol, pgrp_ready, gp, gmk, gcyc);
    };
    // the next (up to) T candidates of `todo` -- one per lane: leaf kind and ring words -- through one pass; true on the lanes whose
    // candidate was scored, with its (penalty, makespan, cycle flag)
    auto pgrp_batch = [&](uint64_t& todo, int lkind, uint32_t a0, uint32_t a1, uint32_t ax, int64_t& tp, int64_t& tm_, bool& tcyc) -> bool {
        const uint32_t my_g = lane >> pgrp_shift;
        int src = -1, my_slot = -1;  // the candidate lane my group evaluates; the group that evaluates my candidate
        for (uint32_t q = 0; q < pgrp_T && todo; ++q) {
            const int ci = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            if (my_g == q) src = ci;
            if ((int)lane == ci) my_slot = (int)q;
        }
        const int sl = src < 0 ? (int)lane : src;
        const int ck = __shfl(lkind, sl);
        const uint32_t ca = __shfl(a0, sl), cb = __shfl(a1, sl), cx = __shfl(ax, sl);
        PgrpMove gm;
        gm.kind = src < 0 ? 0u : (uint32_t)list_move_kind_of(ck);
        gm.a = ca >> 16, gm.i = ca & 0xFFFFu, gm.b = cb >> 16, gm.j = cb & 0xFFFFu, gm.el2 = 0;
        gm.ext = list_move_ext_of(ck, ca, cb, cx);
        if (ck == 8192) gm.b = gm.a, gm.j = gm.i + (cb >> 16);  // permute: (list, start, window, rank)
        int64_t gp = 0, gmk = 0;
        bool gcyc = false;
        pgrp_eval(gm, gp, gmk, gcyc);
        const int from = my_slot < 0 ? (int)lane : (my_slot << pgrp_shift);
        tp = (int64_t)shfl_u64((uint64_t)gp, from), tm_ = (int64_t)shfl_u64((uint64_t)gmk, from);
        tcyc = __shfl((int)gcyc, from) != 0;
        return my_slot >= 0;
    };
    // ---- critical-path precedence leaf (kind 16384): per-replica tables, one full evaluation with the cycle flag ----
    const bool plf_on = PREC && gl.plf.on != 0;
    const bool plf_policy = plf_on && gl.plf.policy != 0;  // runtime slot with precedence hooks: route-graph filter + ruins with hooks
    bool plf_cur_cyclic = false;                           // the working lists of this step are cyclic
    PlfRep plf{};
    __shared__ uint32_t s_plf_info[4][4];
    uint32_t* const plf_info = s_plf_info[wave_in_group];
    int64_t* const plf_score = plf_on ? gl.plf.score + (size_t)r * GRC * 4 : nullptr;
    int64_t* const plf_cache = plf_on ? gl.plf.cache + (size_t)r * GL * GRC * 2 : nullptr;  // the filter's evaluation per ring slot
    if (plf_on) {
        const size_t pn = (size_t)gl.prec.n, pc = (size_t)gl.plf.pc;
        plf.latest = gl.plf.latest + (size_t)r * pn, plf.posn = gl.plf.posn + (size_t)r * pn, plf.flag = gl.plf.flag + (size_t)r * pc;
        plf.roff = gl.plf.roff + (size_t)r * (pn + 2), plf.blk = gl.plf.blk + (size_t)r * pn * 2, plf.csw = gl.plf.csw + (size_t)r * pn;
        plf.ssw = gl.plf.ssw + (size_t)r * pn, plf.first = gl.plf.first + (size_t)r * pc, plf.cnl = gl.plf.cnl + (size_t)r * pn;
        plf.msrow = gl.plf.msrow + (size_t)r * (pn + 1), plf.mrrow = gl.plf.mrrow + (size_t)r * (pn + 1), plf.sE = gl.plf.sE + (size_t)r * V, plf.visit = gl.plf.visit + (size_t)r * pn;
    }
    // the recreate's scratch rows -- tails, list predecessors, the two reachability marks, the search queue, Kahn's rounds: 24 bytes per node --
    // borrow the grouped evaluator's trial scratch in LDS when it is there (no trial is in flight during a recreate)
    PlfRep plf_r = plf;
    if (plf_on && pgrp_T) {
        const size_t pn = (size_t)gl.prec.n, shared = pgrp_shared_bytes((int)pn, V);
        const size_t room = pgrp_bytes((int)pn, (int)pgrp_T, V) - shared;
        uint32_t* w = (uint32_t*)(mem + cv.pgrp + shared);
        if (room >= 20 * pn) plf_r.latest = (int32_t*)w, plf_r.first = w + pn, plf_r.visit = w + 2 * pn, plf_r.flag = w + 3 * pn, plf_r.cnl = w + 4 * pn;
        if (room >= 24 * pn + 8) plf_r.roff = w + 5 * pn;
    }
    // full evaluation of the lists in LDS that also reports the cycle flag (and, with `roff`, Kahn's rounds)
    auto plf_eval = [&](bool& cyclic, uint32_t* roff, uint32_t* lp = nullptr) -> PrecResult {
        PrecResult pr;
        if (prec_in_lds && !roff && !lp) {  // the route-graph filter's evaluations: nobody reads their pop order (plf_closes_cycle walks the list successors)
            if (gl.prec_static)
                pr = prec_eval<uint16_t, PrecMemLds, false, true>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D,
                                                                  (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16, nullptr, plf_info, nullptr);
            else
                pr = prec_eval<uint16_t, PrecMemLds, false, false>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D,
                                                                   (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16, nullptr, plf_info, nullptr);
        } else if (prec_in_lds && gl.prec_static)
            pr = prec_eval<uint16_t, PrecMemLds, true, true>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D,
                                                             (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16, lp, plf_info, roff);
        else if (prec_in_lds)
            pr = prec_eval<uint16_t, PrecMemLds>(precm, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_i32*)prec_D, (prec_lds_u16*)prec_Q16,
                                                 (prec_lds_u16*)prec_S16, lp, plf_info, roff);
        else
            pr = prec_eval<uint16_t, PrecMemGlobal>(precm, s_visits, s_off, V, prec_E, prec_D, prec_Q, prec_S, lp, plf_info, roff);
        wave_sync();
        cyclic = uni(plf_info[1]) != 0u;
        return pr;
    };
    auto plf_score_of = [&](const PrecResult& pr) {
        ScoreV<L> sc;
#pragma unroll
        for (int kk = 0; kk < L; ++kk) {
            sc.v[kk] = cur[kk];
            if (kk == gl.prec.hard_level) sc.v[kk] -= pr.penalty - prec_pen;
            if (kk == gl.prec.mk_level) sc.v[kk] -= pr.makespan - prec_mk;
        }
        return sc;
    };
    auto plf_restore_all = [&]() {  // the committed lists back from their HBM copy (written at every commit)
        for (uint32_t t = lane; t <= (uint32_t)V; t += 64) s_off[t] = g_off[t];
        wave_sync();
        const uint32_t tot = uni(s_off[V]);
        for (uint32_t t = lane; t < tot; t += 64) s_visits[t] = (uint16_t)g_visits[t];
        wave_sync();
    };
    // precedence-aware ruin and recreate (move/list_kernel/ruin.rs:131-281 with recreate_precedence_graph): the removed elements go back
    // one per round at the best-scoring insertion over every (element, list, position) that does not close a cycle, the first of
    // equal scores staying.  One element slides through every position with an adjacent exchange (or a list boundary shift) per
    // step, one full evaluation each.
    auto plf_ruin = [&](const PlfMove& m, bool keep, bool hooks, bool skip_empty) -> ScoreV<L> {
        uint32_t vals[PLF_EL_MAX];
#pragma unroll
        for (uint32_t k = 0; k < PLF_EL_MAX; ++k) vals[k] = 0;
        for (uint32_t k = m.n; k-- > 0;) {
            const uint32_t x = plf_list_remove(s_visits, s_off, V, m.el[k] >> 16, m.el[k] & 0xFFFFu);
#pragma unroll
            for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
                if (q == k) vals[q] = x;
        }
        uint32_t remaining = (1u << m.n) - 1u;
        ScoreV<L> last;
#pragma unroll
        for (int kk = 0; kk < L; ++kk) last.v[kk] = cur[kk];
        bool rolled = false;
        for (uint32_t round = 0; round < m.n && !rolled; ++round) {
            bool have = false;
            ScoreV<L> best_sc = last;
            uint32_t b_ri = 0, b_e = 0, b_pos = 0;
            // the lists without the remaining elements: acyclic => every slot of every remaining element is priced from one forward
            // evaluation + one backward pass (plf_best_slot); cyclic => the element slides through the slots, one evaluation each
            bool base_cyc;
            PHR(7)
            const PrecResult base = plf_eval(base_cyc, plf_r.roff, plf_r.first);
            PHR(4)
            if (gl.plf.slow) base_cyc = true;
            if (!base_cyc) {
                const uint32_t rounds = uni(plf_info[2]);
                if (prec_in_lds)
                    plf_tails<PrecMemLds>(precm, plf_r, (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16, rounds);
                else
                    plf_tails<PrecMemGlobal>(precm, plf_r, prec_Q, prec_S, rounds);
            }
            PHR(5)
            const int lvl_order = gl.prec.hard_level < gl.prec.mk_level ? 0 : (gl.prec.hard_level > gl.prec.mk_level ? 1 : 2);
            for (uint32_t ri = 0; ri < m.n; ++ri) {
                if (!((remaining >> ri) & 1u)) continue;
                uint32_t x = 0;
#pragma unroll
                for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
                    if (q == ri) x = vals[q];
                x = uni(x);
                if (!base_cyc) {
                    PlfSlotPick pk{0, 0, 0, 0, 0};
                    if (prec_in_lds)
                        plf_best_slot<PrecMemLds>(pk, precm, plf_r, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_u16*)prec_S16, base.penalty,
                                                  (int32_t)base.makespan, x, hooks, skip_empty, lvl_order);
                    else
                        plf_best_slot<PrecMemGlobal>(pk, precm, plf_r, s_visits, s_off, V, prec_E, prec_S, base.penalty, (int32_t)base.makespan, x, hooks,
                                                     skip_empty, lvl_order);
                    PHR(6)
                    if (pk.found) {
                        const ScoreV<L> sc = plf_score_of(PrecResult{pk.pen, pk.mk});
                        if (!have || score_cmp<L>(sc, best_sc) > 0) {
                            have = true;
                            best_sc = sc, b_ri = ri, b_e = pk.e, b_pos = pk.k;
                        }
                    }
                    continue;
                }
                plf_list_insert(s_visits, s_off, V, 0, 0, x);
                uint32_t e = 0, pos = 0, g = 0;
                for (;;) {
                    const uint32_t others = uni(s_off[e + 1] - s_off[e]) - 1u;  // list e without the sliding element
                    if (!(skip_empty && others == 0u)) {  // skip_empty_destinations (ruin.rs:181-183)
                        bool cyc;
                        const PrecResult pr = plf_eval(cyc, nullptr);
                        if (!(cyc && hooks)) {  // with the hooks an insertion that closes a cycle is not tried (ruin.rs:186-220)
                            const ScoreV<L> sc = plf_score_of(pr);
                            if (!have || score_cmp<L>(sc, best_sc) > 0) {
                                have = true;
                                best_sc = sc, b_ri = ri, b_e = e, b_pos = pos;
                            }
                        }
                    }
                    if (pos < others) {
                        if (lane == 0) {
                            const uint16_t y = s_visits[g + 1];
                            s_visits[g + 1] = (uint16_t)x;
                            s_visits[g] = y;
                        }
                        g += 1, pos += 1;
                    } else if (e + 1 < (uint32_t)V) {
                        if (lane == 0) s_off[e + 1] -= 1;  // the tail of list e becomes the head of list e + 1
                        e += 1, pos = 0;
                    } else
                        break;
                    wave_sync();
                }
                if (lane == 0) s_off[V] -= 1;  // the element sits at the very end: drop it
                wave_sync();
            }
            if (!have) {
                rolled = true;
                break;
            }
            b_ri = uni(b_ri), b_e = uni(b_e), b_pos = uni(b_pos);
            uint32_t bx = 0;
#pragma unroll
            for (uint32_t q = 0; q < PLF_EL_MAX; ++q)
                if (q == b_ri) bx = vals[q];
            bx = uni(bx);
            plf_list_insert(s_visits, s_off, V, b_e, b_pos, bx);
            remaining &= ~(1u << b_ri);
            last = best_sc;
        }
        if (rolled) {  // restore_removed_elements: the move leaves the lists as they were
#pragma unroll
            for (int kk = 0; kk < L; ++kk) last.v[kk] = cur[kk];
        }
        if (rolled || !keep) plf_restore_all();
        return last;
    };
    // one decoded candidate applied to the LDS lists and scored; false = its lists are cyclic (pruned).  keep: leave it applied.
    auto plf_apply = [&](const PlfMove& m) {
        if (m.kind == 10) {
            if (lane < 3) {
                const uint32_t w = lane == 0 ? m.el[0] : (lane == 1 ? m.el[1] : m.el[2]);
                const uint32_t gpos = s_off[w >> 16] + (w & 0xFFFFu);
                const uint16_t x = s_visits[gpos], y = s_visits[gpos + 1];
                s_visits[gpos] = y, s_visits[gpos + 1] = x;
            }
            wave_sync();
        } else
            apply_list_move_wave(lm, s_visits, s_off, s_load, m.kind, m.a, m.ap, m.b, m.bp, m.ext);
        wave_sync();
    };
    auto plf_trial = [&](const PlfMove& m, ScoreV<L>& sc) -> bool {
        if (m.kind == 8) {
            sc = plf_ruin(m, false, true, false);
            return true;
        }
        plf_apply(m);
        bool cyc;
        const PrecResult pr = plf_eval(cyc, nullptr);
        sc = plf_score_of(pr);
        if (m.kind == 10)
            plf_apply(m);  // the same exchanges again
        else {  // one list changed: its items back from the committed copy
            const uint32_t lo = uni(s_off[m.a]), hi = uni(s_off[m.a + 1]);
            for (uint32_t t = lo + lane; t < hi; t += 64) s_visits[t] = (uint16_t)g_visits[t];
            wave_sync();
        }
        return !cyc;
    };
    // ordinary ruin leaf on a precedence model: its candidate record (list, count, positions) as a PlfMove
    auto plf_from_ruin_cand = [&](const uint16_t* cd, PlfMove& m) {
        m.kind = 8;
        m.a = m.b = uni((uint32_t)cd[0]);
        m.n = uni((uint32_t)cd[1]);
        m.ap = m.n, m.bp = 0, m.ext = 0;
#pragma unroll
        for (uint32_t k = 0; k < PLF_EL_MAX; ++k) m.el[k] = (m.a << 16) | uni((uint32_t)cd[2 + k]);
    };
    // Route-graph filter of a runtime list leaf (with_precedence_route_graph, precedence_route.rs:313-317,419-451): does the intra-list
    // candidate just applied to list e close a cycle through a NEW route edge?  After an acyclic working state that is "the lists are
    // cyclic now"; after a cyclic one, every new edge (u, v) -- in the route now, not in the committed route, not a fixed edge -- is
    // tested for v reaching u over the graph of the lists as they are now (plf_reaches).  `cyc` = the cycle flag of the evaluation
    // that just ran on the applied state (its list successors are still in the scratch).
    auto plf_closes_cycle = [&](uint32_t e, bool cyc) -> bool {
        if (!cyc) return false;
        if (!plf_cur_cyclic) return true;
        const uint32_t lo = uni(s_off[e]), hi = uni(s_off[e + 1]);
        bool closes = false;
        for (uint32_t p0 = lo; p0 + 1 < hi && !closes; p0 += 64) {
            const uint32_t p = p0 + lane;
            bool fresh = false;
            uint32_t u = 0, v = 0;
            if (p + 1 < hi) {
                u = s_visits[p], v = s_visits[p + 1];
                const uint32_t w = plf.posn[u];  // u's committed position (in list e: the move is intra-list)
                const uint32_t old_next = (w & 0xFFFFu) + 1u < g_off[e + 1] - g_off[e] ? g_visits[g_off[e] + (w & 0xFFFFu) + 1u] : PREC_NONE;
                fresh = old_next != v;
                for (uint32_t k = gl.prec.succ_off[u]; fresh && k < gl.prec.succ_off[u + 1]; ++k) fresh = gl.prec.succ[k] != v;
            }
            uint64_t fm = __ballot(fresh);
            while (fm && !closes) {
                const int ci = __ffsll((unsigned long long)fm) - 1;
                fm &= fm - 1;
                const uint32_t uu = (uint32_t)__builtin_amdgcn_readlane((int)u, ci), vv = (uint32_t)__builtin_amdgcn_readlane((int)v, ci);
                closes = prec_in_lds ? plf_reaches<PrecMemLds>(precm, plf.visit, plf.cnl, (prec_lds_u16*)prec_S16, vv, uu)
                                     : plf_reaches<PrecMemGlobal>(precm, plf.visit, plf.cnl, prec_S, vv, uu);
            }
        }
        return closes;
    };
    // ring entry (stage << 30 | block, index inside the stage) -> move
    auto plf_decode = [&](uint32_t w0, uint32_t w1, PlfMove& m) {
        const uint32_t stage = w0 >> 30;
        if (stage == 0)
            plf_decode_multi_swap(plf, ((uint64_t)(w0 & 0x3FFFFFFFu) << 32) | w1, m);
        else if (stage == 1)
            plf_decode_multi_ruin(plf, w1, m);
        else
            plf_decode_block(plf_block(plf, w0 & 0x3FFFFFFFu), w1, m);
    };
    // per-launch counters in 32 bits (wave-uniform), folded into the 64-bit sf_stats words before they can wrap
    uint32_t st_steps = 0, st_gen = 0, st_acc = 0, st_applied = 0, st_calc = 0, st_scored = 0, st_sources = 0;
    uint64_t steps_run = 0;
    auto flush_stats = [&]() {
        if (lane == 0) {
            uint64_t* gs = p.stats + (size_t)r * SF_STATS_WORDS;
            gs[0] += st_steps;
            gs[1] += st_gen;
            gs[2] += st_gen;
            gs[3] += st_acc;
            gs[4] += st_applied;
            gs[5] += st_calc;
            gs[6] += st_gen - st_calc;
            gs[7] += st_scored;
            gs[8] += st_sources;
        }
        steps_run += st_steps;
        st_steps = st_gen = st_acc = st_applied = st_calc = st_scored = st_sources = 0;
    };
    uint64_t trace_n = 0;
    const uint64_t step_index0 = dry_run ? 0 : p.step_index[r];
    const uint64_t seed_draws0 = dry_run ? 0 : p.seed_draws[r];
    const int la_idx0 = dry_run ? 0 : p.la_idx[r];
    int la_cursor = la_idx0;  // (la_idx0 + step) % la_size, kept incrementally (no 64-bit division per step)
    bool best_pending = false;  // working == best, snapshot not yet written (see sf_scalar_kernels.hip: deferred clone)
    const FastMod fm_V = make_fastmod(V > 0 ? (uint32_t)V : 1u);  // Barrett remainder by the owner count: no 64-bit division per entity rank
    const FastMod fm_V1 = make_fastmod(V > 1 ? (uint32_t)V - 1u : 1u);
    // coprimality of every candidate permutation stride of the V list owners, once per launch (lane s tests s and s + 64)
    const bool use_cm = V >= 2 && V <= 128;
    const uint64_t cm_lo = use_cm ? __ballot(lane >= 1 && lane < (uint32_t)V && gcd_u32(lane, (uint32_t)V) == 1) : 0ull;
    const uint64_t cm_hi = use_cm ? __ballot(lane + 64 < (uint32_t)V && gcd_u32(lane + 64, (uint32_t)V) == 1) : 0ull;

    for (int64_t step = 0; step < p.n_steps; ++step) {
        // load-balance / balance aggregates of the step snapshot (the tables change only at commit): every lane gets the totals
        int64_t lbv[4] = {0, 0, 0, 0};
        if (tables && sm.grp_level >= 0 && sm.grp_mode >= 1) {
            int64_t s1 = 0, s2 = 0;
            uint32_t nk = 0;
            for (uint32_t v = lane; v < (uint32_t)sm.n_values; v += 64)
                if (t_cnt[v]) {
                    const int64_t x = sm.grp_mode == 2 ? (int64_t)t_cnt[v] : t_sum[v];
                    s1 = wadd(s1, x);
                    s2 = wadd(s2, (int64_t)((uint64_t)x * (uint64_t)x));
                    nk += 1;
                }
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                s1 = wadd(s1, (int64_t)shfl_u64((uint64_t)s1, (int)(lane ^ (uint32_t)o)));
                s2 = wadd(s2, (int64_t)shfl_u64((uint64_t)s2, (int)(lane ^ (uint32_t)o)));
                nk += (uint32_t)__shfl((int)nk, (int)(lane ^ (uint32_t)o));
            }
            lbv[0] = s1, lbv[1] = s2, lbv[2] = (int64_t)nk, lbv[3] = global_stat(sm, s1, s2, nk);
        }
        uint32_t step_pulls = 0;  // pulls of this step (FAST: decides whether the next step fills its rings in lumps)
        uint64_t sidx, sseed;
        if (dry_run) {
            sidx = p.dry_step_index;
            sseed = p.dry_step_seed;
        } else {
            sidx = step_index0 + (uint64_t)step;
            const uint64_t draw = seed_draws0 + (uint64_t)step;
            if (!FAST && p.explicit_seeds && (int64_t)draw < p.n_explicit)
                sseed = p.explicit_seeds[(size_t)r * p.n_explicit + draw];
            else
                sseed = step_seed(p.random_seed + (uint64_t)r, draw);
        }
        sidx = uni64(sidx);
        sseed = uni64(sseed);
        // FAST: SelectionOrder::Random, the default policy's (host-checked): every selection_index call site compiles to the one hash +
        // remainder instead of all three orders (the Shuffled branch alone is two more remainders and a gcd loop per site)
        const StreamCtx ctx{sidx, sseed, FAST ? 3 : p.order};
        ScoreV<L> late;
#pragma unroll
        for (int k = 0; k < L; ++k) late.v[k] = 0;
        const int la_slot = la_cursor;  // LateAcceptance history slot of this step
        if (acceptor == 1 || acceptor == 4) {
#pragma unroll
            for (int k = 0; k < L; ++k) late.v[k] = p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + k];
        }
        ScoreV<L> dla_thr = late;  // DiversifiedLateAcceptance: best step score of the phase minus its tolerance band
        if (acceptor == 4) {
            ScoreV<L> db;
#pragma unroll
            for (int k = 0; k < L; ++k) db.v[k] = (int64_t)uni64((uint64_t)p.dla_best[(size_t)r * 4 + k]);
            dla_thr = dla_threshold<L>(db, p.dla_tolerance);
        }
        int has_best = 0;
        uint64_t equal_count = 0;
        uint32_t accepted = 0;
        ScoreV<L> best;
#pragma unroll
        for (int k = 0; k < L; ++k) best.v[k] = 0;
        uint32_t best_m0 = 0, best_m1 = 0, best_x = 0;
        int best_leaf = 0;
        uint64_t best_ti = 0;  // trace ordinal (within the step) of the forager's current pick

        // entity permutations (selection_index_without_replacement) of the four streams
        uint32_t sc_st = 0, sc_sd = 1, ss_st = 0, ss_sd = 1, lc_st = 0, lc_sd = 1, ls_st = 0, ls_sd = 1, lr_st = 0, lr_sd = 1;
        uint32_t sb_st = 0, sb_sd = 1, sw_st = 0, sw_sd = 1;  // sublist change / swap entity permutations
        uint32_t ko_st = 0, ko_sd = 1;                        // 3-opt entity permutation
        if (has_scalar) {
            ctx.perm_params(ns, SALT_SCALAR_CHANGE_ENTITY ^ identity, sc_st, sc_sd);
            ctx.perm_params(ns, (SALT_SCALAR_SWAP_LEFT ^ identity) ^ OFFSET_MIX, ss_st, ss_sd);
        }
        if (has_list) {
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, SALT_LC_ENTITY ^ ldesc, lc_st, lc_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, SALT_LC_ENTITY ^ ldesc, lc_st, lc_sd);
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, SALT_LS_ENTITY ^ ldesc, ls_st, ls_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, SALT_LS_ENTITY ^ ldesc, ls_st, ls_sd);
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, SALT_LR_ENTITY ^ ldesc, lr_st, lr_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, SALT_LR_ENTITY ^ ldesc, lr_st, lr_sd);
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, SALT_SC_ENTITY ^ ldesc, sb_st, sb_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, SALT_SC_ENTITY ^ ldesc, sb_st, sb_sd);
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, SALT_SS_ENTITY ^ ldesc, sw_st, sw_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, SALT_SS_ENTITY ^ ldesc, sw_st, sw_sd);
            if (use_cm)
                ctx.perm_params_fm(fm_V, fm_V1, (gl.kopt_nearby ? SALT_KN_ENTITY : SALT_KF_ENTITY) ^ ldesc, ko_st, ko_sd, cm_lo, cm_hi);
            else
                ctx.perm_params((uint32_t)V, (gl.kopt_nearby ? SALT_KN_ENTITY : SALT_KF_ENTITY) ^ ldesc, ko_st, ko_sd);
        }
        sc_st = uni(sc_st), sc_sd = uni(sc_sd), ss_st = uni(ss_st), ss_sd = uni(ss_sd);
        lc_st = uni(lc_st), lc_sd = uni(lc_sd), ls_st = uni(ls_st), ls_sd = uni(ls_sd);
        lr_st = uni(lr_st), lr_sd = uni(lr_sd);
        sb_st = uni(sb_st), sb_sd = uni(sb_sd);
        sw_st = uni(sw_st), sw_sd = uni(sw_sd);
        ko_st = uni(ko_st), ko_sd = uni(ko_sd);
        auto ko_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)ko_st + (uint64_t)rank * ko_sd, fm_V); };
        if (gl.kopt_nearby) {  // a fresh cursor per step: no entity open yet
            KoptLds km(mem + cv.kopt);
            if (lane == 0) km.st[15] = 0;
            wave_sync();
        }
        auto sw_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)sw_st + (uint64_t)rank * sw_sd, fm_V); };
        auto sb_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)sb_st + (uint64_t)rank * sb_sd, fm_V); };
        auto lr_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)lr_st + (uint64_t)rank * lr_sd, fm_V); };
        auto lc_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)lc_st + (uint64_t)rank * lc_sd, fm_V); };
        auto ls_ent = [&](uint32_t rank) { return fastmod_u64((uint64_t)ls_st + (uint64_t)rank * ls_sd, fm_V); };
        auto rlen = [&](uint32_t e) { return uni(s_off[e + 1] - s_off[e]); };

        for (int l = 0; l < GL; ++l) {
            lt.put_gen(l, GGen{0, 0, 0, 0, 0, 0, l >= nl});
            lt.set(l, LeafTab::HEAD, 0);
            lt.set(l, LeafTab::TAIL, 0);
            lt.set(l, LeafTab::EX, l >= nl);
            lt.set(l, LeafTab::WCUR, 0);
        }
        if (!FAST && has_list) {  // list permute leaf: its entity permutation of this step (slot.rs:468-499) in the generator state
            for (int l = 0; l < nl; ++l) {
                if (lt.geti(l, LeafTab::KIND) != 8192) continue;
                uint32_t pst, psd;
                if (use_cm)
                    ctx.perm_params_fm(fm_V, fm_V1, SALT_PM_ENTITY ^ ldesc, pst, psd, cm_lo, cm_hi);
                else
                    ctx.perm_params((uint32_t)V, SALT_PM_ENTITY ^ ldesc, pst, psd);
                lt.put_gen(l, GGen{0, 0, uni(pst), uni(psd), 0, 0, V == 0});
            }
        }
        if (!FAST && has_scalar) {  // nearby scalar leaves: ordered_entity's start / stride of this step (change.rs:376-391) in the generator state
            for (int l = 0; l < nl; ++l) {
                const int lk = lt.geti(l, LeafTab::KIND);
                if (lk != 2048 && lk != 4096) continue;
                const uint64_t sa = (lk == 2048 ? SALT_NSC_START : SALT_NSW_START) ^ identity, sb_ = (lk == 2048 ? SALT_NSC_STRIDE : SALT_NSW_STRIDE) ^ identity;
                const uint32_t st0 = (ctx.canonical() || ns <= 1) ? 0u : ctx.random_index(ns, sa);
                const uint32_t sd0 = (ctx.canonical() || ns <= 1) ? 1u : ctx.random_stride(ns, sb_);
                lt.put_gen(l, GGen{0, 0, uni(st0), uni(sd0), 0, 0, ns == 0});
            }
        }
        if (plf_on) {  // critical-path leaf: the committed evaluation again (the trials overwrote its arrays), then the step's analysis
            bool cyc;
            const PrecResult pr = plf_eval(cyc, plf.roff);
            plf_cur_cyclic = cyc;
            const uint32_t rounds = uni(plf_info[2]);
            if (prec_in_lds)
                plf_analyse<PrecMemLds>(precm, plfm, plf, s_visits, s_off, V, (prec_lds_i32*)prec_E, (prec_lds_u16*)prec_Q16, (prec_lds_u16*)prec_S16, rounds,
                                        (int32_t)pr.makespan, cyc);
            else
                plf_analyse<PrecMemGlobal>(precm, plfm, plf, s_visits, s_off, V, prec_E, prec_Q, prec_S, rounds, (int32_t)pr.makespan, cyc);
            plf.nb = uni(plf.nb), plf.C = uni(plf.C), plf.S = uni(plf.S), plf.ms_count = uni64(plf.ms_count), plf.mr_count = uni(plf.mr_count);
        }
        if (pgrp_T) {  // grouped trial evaluator: the committed list edges every trial of this step starts from
            const PgrpLds pl(mem + cv.pgrp, gl.prec.n, V, 0, (int)pgrp_T);
            uint32_t pv = 0, pr = 0;
            pgrp_build_committed<uint16_t>(gl.prec, (const PREC_L uint16_t*)s_visits, (const PREC_L uint32_t*)s_off, V, pl, pv, pr);
            pgrp_viol = uni(pv), pgrp_ready = uni(pr);
        }
        uint32_t exmask = ((1u << GL) - 1u) & ~((1u << nl) - 1u);  // bit l: leaf l is exhausted (wave-uniform mirror of LeafTab::EX)
        // the scheduler's pull order inside one whole cycle, kept across the batches of a step: it is a function of the live set and the running
        // weights, and whole cycles leave both as they found them -- only the pull-by-pull simulation changes them (and drops the cache).  Without
        // it every batch after a child ran dry mid-cycle (the ruin leaf: ten candidates a step) re-ranked the children, nl^2 v_readlane + compares
        uint64_t ro_cache = 0;
        uint32_t ro_live = 0xFFFFFFFFu;  // the live mask `ro_cache` was computed for (all ones: none)
        // nearby leaves: entity order tables of this step (slot.rs:468-499), same layout as the wave engine
        if (has_nearby) {
            const uint32_t total = uni(s_off[V]);
            int ni = 0;
            for (int l = 0; l < nl; ++l) {
                const int lk = lt.geti(l, LeafTab::KIND);
                if (lk != 16 && lk != 32) continue;
                const uint64_t ent_salt = (lk == 16 ? SALT_NEARBY_CHANGE_ENTITY : SALT_NEARBY_SWAP_ENTITY) ^ ldesc;
                uint32_t pst, psd;
                if (use_cm)
                    ctx.perm_params_fm(fm_V, fm_V1, ent_salt, pst, psd, cm_lo, cm_hi);
                else
                    ctx.perm_params((uint32_t)V, ent_salt, pst, psd);
                pst = uni(pst);
                psd = uni(psd);
                uint16_t* ra = nb_route_at + ni * V;
                uint16_t* ro = nb_rank_of + ni * V;
                uint16_t* sb = nb_slot_base + ni * (V + 1);
                for (uint32_t k = lane; k < (uint32_t)V; k += 64) {
                    const uint32_t e = fastmod_u64((uint64_t)pst + (uint64_t)k * psd, fm_V);
                    ra[k] = (uint16_t)e;
                    ro[e] = (uint16_t)k;
                }
                wave_sync();
                uint32_t carry = 0;
                for (uint32_t base = 0; base < (uint32_t)V; base += 64) {
                    const uint32_t k = base + lane;
                    uint32_t v = 0;
                    if (k < (uint32_t)V) {
                        const uint32_t e = ra[k];
                        v = s_off[e + 1] - s_off[e] + 1;
                    }
                    const uint32_t inc = wave_incl_scan(v);
                    if (k < (uint32_t)V) sb[k] = (uint16_t)(carry + inc - v);
                    carry += __shfl(inc, 63);
                }
                if (lane == 0) sb[V] = (uint16_t)carry;
                lt.put_gen(l, GGen{0, 0, 0xFFFFFFFFu, 0, total, 0, total == 0});
                ++ni;
            }
            wave_sync();
        }
        if (RUIN) {  // list ruin leaf: open the cursor (one draw of the per-solve stream), count the source pool
            const uint32_t pool = ruin_open_cursor(gl.ruin, rl, ctx, s_off, V, !dry_run, lane);
            for (int l = 0; l < nl; ++l)
                if (lt.geti(l, LeafTab::KIND) == 1024) lt.put_gen(l, GGen{0, 0, 0, 0, pool, 0, pool == 0 || gl.ruin.moves_per_step <= 0});
            if (PREC && pool != 0) {  // (a precedence model: the recreate is scored by the precedence constraint, with the slot's hooks when declared)
                PHS(0)
                for (uint32_t c = 0; c < (uint32_t)gl.ruin.moves_per_step; ++c) {
                    if (c == 0 && rfast.edge) ruin_build_edges(lm, s_visits, s_off, ruin_sbase, rfast);
                    ruin_next_candidate(gl.ruin, rl, s_off, V, pool, c, lane);
                    PlfMove pm_;
                    plf_from_ruin_cand(rl.cand + (size_t)c * RuinLds::CAND_WORDS, pm_);
                    const ScoreV<L> psc = plf_ruin(pm_, false, plf_policy, gl.ruin.skip_empty != 0);
                    if (lane == 0) {
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) rl.score[(size_t)c * 4 + kk] = psc.v[kk];
                    }
                    wave_sync();
                }
                PHS(3)
            }
            if (!PREC && pool != 0) {
                // Every candidate of the step is generated and scored HERE, before the fill / replay loop, not inside it (round 5).  The
                // leaf's stream depends on the committed state and the step's cursor seed alone, the first fill asked for all of them anyway
                // (moves_per_step <= 16 < the fill threshold), and a recreate is 10^5 clocks of wave-wide work behind a call: inside the
                // loop nest every value of the generators, the scheduler and the replay was live across that call, and the register
                // allocator kept them in scratch for the whole loop (the RUIN instantiations ran the SAME six-leaf work 5x slower than
                // the kernels without the leaf, profiles/r05_phase7_*.txt).  Out here only step-level values cross it.
                PHS(0)
#ifdef SF_RUIN_NO_V2
                const bool v2 = false;
#else
                // trials without touching the lists (sf_ruin_v2.h) when the model is the default policy's shape; sf_ruin.h otherwise, for a
                // candidate whose changed lists outgrow the scratch arena, and for the committed move
                const bool v2 = FAST ? true : (rfast.edge != nullptr && rv2_model_ok(lm));  // (FAST: host-checked)
#endif
                for (uint32_t c = 0; c < (uint32_t)gl.ruin.moves_per_step; ++c) {
                    if (c == 0 && (FAST || rfast.edge)) ruin_build_edges(lm, s_visits, s_off, ruin_sbase, rfast);
                    if (c == 0 && v2) rv2_build_words(s_off, (uint32_t)V, ruin_sbase);
                    ruin_next_candidate(gl.ruin, rl, s_off, V, pool, c, lane);
                    int64_t base_score[L];
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) base_score[kk] = cur[kk];
                    const uint16_t* cand_c = rl.cand + (size_t)c * RuinLds::CAND_WORDS;
                    bool scored = false;
                    if (v2) scored = ruin_trial_v2<L>(lm, s_visits, s_off, s_load, cand_c, rl.work, ruin_sbase, rfast, rfast.slot, (uint32_t)ruin_arena_cap(lm.n_cap, V),
                                                      gl.ruin.skip_empty, base_score, rl.score + (size_t)c * 4);
#ifdef SF_RUIN_V2_CHECK
                    if (scored) {
                        wave_sync();
                        int64_t v2s[L];
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) v2s[kk] = rl.score[(size_t)c * 4 + kk];
                        ruin_recreate<L>(lm, s_visits, s_off, s_load, cand_c, rl.work, ruin_sbase, rfast, gl.ruin.skip_empty, false, base_score, rl.score + (size_t)c * 4);
                        wave_sync();
                        bool same = true;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) same = same && v2s[kk] == rl.score[(size_t)c * 4 + kk];
                        if (lane == 0) {
                            atomicAdd(&g_rv2_check[0], 1ull);
                            if (!same && atomicAdd(&g_rv2_check[1], 1ull) == 0ull) {
                                g_rv2_check[3] = (unsigned long long)r, g_rv2_check[4] = (unsigned long long)c, g_rv2_check[5] = (unsigned long long)cand_c[1];
                                g_rv2_check[6] = (unsigned long long)v2s[L - 1], g_rv2_check[7] = (unsigned long long)rl.score[(size_t)c * 4 + L - 1];
                            }
                        }
                        rv2_build_words(s_off, (uint32_t)V, ruin_sbase);
                    } else if (v2 && lane == 0) {
                        atomicAdd(&g_rv2_check[2], 1ull);
                    }
#endif
                    if constexpr (!FAST) {  // (FAST: the host checked the model, the arena cannot overflow -- sf_ruin.h is not in that kernel at all)
                        if (!scored) {
                            ruin_recreate<L>(lm, s_visits, s_off, s_load, cand_c, rl.work, ruin_sbase, rfast, gl.ruin.skip_empty, false, base_score, rl.score + (size_t)c * 4);
                            if (v2) rv2_build_words(s_off, (uint32_t)V, ruin_sbase);  // (the slot prefix of sf_ruin.h lives in the same words)
                        }
                    }
                    wave_sync();
                }
                PHS(3)
            }
        }
        // union scheduler (vec_union.rs:190-365): StratifiedRandom with equal weights when > 1 leaf
        const uint32_t u_off = nl > 1 ? ctx.random_index((uint32_t)nl, SALT_UNION_OFFSET) : 0u;
        const uint32_t u_str = nl > 1 ? ctx.random_stride((uint32_t)nl, SALT_UNION_STRIDE) : 1u;
        int32_t live_weight = nl;  // total weight of the live children (running weights of the smooth weighted round-robin: leaf table)
        uint32_t u_cur = 0, u_draw = 0;  // RoundRobin cursor / Random draw counter of this step's cursor
        const int u_ord = nl > 1 ? (FAST ? 4 : gl.union_order) : 0;
        if (union_custom) {  // weighted children: a zero weight is an exhausted child from the start (vec_union.rs:214-218)
            live_weight = 0;
#pragma unroll
            for (int l = 0; l < GL; ++l) {
                if (l >= nl) continue;
                if (gl.weight[l] == 0) {
                    exmask |= 1u << l;
                    lt.set(l, LeafTab::EX, 1);
                }
                live_weight += gl.weight[l];
            }
            if (u_ord == 2) u_cur = u_off;  // RotatingRoundRobin starts at the seeded offset
        }
        uint64_t u_order = 0;  // rotated child order of this step, 4 bits per position (nl <= 16)
        for (uint32_t pos = 0; pos < (uint32_t)nl; ++pos) u_order |= (uint64_t)((u_off + pos * u_str) % (uint32_t)nl) << (4u * pos);
        u_order = uni64(u_order);

        PHS(0)
        int done = 0;
        while (!done) {
            // ---- C1: fill every live leaf's ring (or until its stream ends).  A replay batch takes 64 pulls in whole
            // scheduler cycles, i.e. ceil(64 / live leaves) candidates per leaf: keep that many (+8) pending, not 64 ----
            uint32_t fill_thr = 64;
            {
                uint32_t live = 0;
                for (int l = 0; l < nl; ++l) live += ((exmask >> l) & 1u) ? 0u : 1u;
                if (live > 1) fill_thr = (63u + live) / live + 8u;
            }
            const bool lump_fill = prev_pulls >= 2048u;  // (a short step would throw most of a 64-entry lump away when it ends)
            for (int l = 0; l < nl; ++l) {
                const int kind = lt.geti(l, LeafTab::KIND);
                uint32_t* rq = ring + (size_t)l * GRC * 2;
                GGen g = lt.gen(l);
                uint32_t tl = lt.get(l, LeafTab::TAIL);
                const uint32_t hd_l = lt.get(l, LeafTab::HEAD);
                const bool ex_l = ((exmask >> l) & 1u) != 0;
                const uint32_t leaf_max_nearby = lt.get(l, LeafTab::MAXNB), leaf_min = lt.get(l, LeafTab::MINSZ), leaf_max = lt.get(l, LeafTab::MAXSZ);
                // pre_eval + a long step: hysteresis -- a leaf is refilled when it runs below what the next replay batch needs, and then up to 64 pending
                // (a generator call appends at most 64, the ring holds 128), so that the scoring stage below finds ~64 new entries of one kind at a time
                const uint32_t fill_to = (pre_eval && lump_fill) ? (tl - hd_l < fill_thr ? 64u : 0u) : fill_thr;
                if (pre_eval) lt.set(l, LeafTab::TAKEN, tl);  // (TAKEN is free between two replays) first entry the scoring stage has not seen
                while (!ex_l && !g.done && tl - hd_l < fill_to) {
                    st_sources += 1;
                    bool keep = false;
                    uint32_t w0 = 0, w1 = 0, wx = 0;
                    int64_t cache_pen = INT64_MIN, cache_mk = 0;  // route-graph filter: the evaluation of this lane's candidate, if it ran
                    if (!FAST && kind == 1) {  // ---- scalar change ----
                        if (g.a >= ns) {
                            g.done = 1;
                            break;
                        }
                        uint32_t my_row = g.a, my_in = g.b + lane;
                        const uint32_t e0 = (uint32_t)(((uint64_t)sc_st + (uint64_t)my_row * sc_sd) % ns);
                        uint32_t per = value_count(sm, e0) + ((sm.allows_unassigned && (int32_t)s_vals[e0] >= 0) ? 1u : 0u);
                        bool valid = true;
                        for (;;) {
                            if (my_row >= ns) {
                                valid = false;
                                break;
                            }
                            if (my_in < per) break;
                            my_in -= per;
                            ++my_row;
                            if (my_row < ns) {
                                const uint32_t e2 = (uint32_t)(((uint64_t)sc_st + (uint64_t)my_row * sc_sd) % ns);
                                per = value_count(sm, e2) + ((sm.allows_unassigned && (int32_t)s_vals[e2] >= 0) ? 1u : 0u);
                            }
                        }
                        if (valid) {
                            w0 = (uint32_t)(((uint64_t)sc_st + (uint64_t)my_row * sc_sd) % ns);
                            int32_t v = -1;
                            if (my_in < value_count(sm, w0)) v = value_at(sm, ctx, w0, my_in, SALT_SCALAR_CHANGE_VALUE ^ (uint64_t)w0 ^ identity, fm_vc);
                            w1 = (uint32_t)v;
                        }
                        keep = valid;
                        const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
                        if (cnt == 0)
                            g.done = 1;
                        else {
                            g.a = uni(__shfl(my_row, (int)cnt - 1));
                            g.b = uni(__shfl(my_in, (int)cnt - 1)) + 1;
                            if (cnt < 64) g.done = 1;
                        }
                    } else if (!FAST && kind == 2) {  // ---- scalar swap ----
                        if (g.a >= ns) {
                            g.done = 1;
                            break;
                        }
                        const uint32_t left = ns <= 1 ? 0u : (uint32_t)(((uint64_t)ss_st + (uint64_t)g.a * ss_sd) % ns);
                        const uint32_t ro = g.b + lane;
                        if (ro < ns) {
                            const uint32_t right = ns <= 1 ? 0u
                                                           : ctx.selection_index_fm(ro, fm_n, (SALT_SCALAR_SWAP_RIGHT ^ (uint64_t)left ^ (uint64_t)(uint32_t)sm.variable) ^ OFFSET_MIX);
                            if (left < right) {
                                const int32_t lv = (int32_t)s_vals[left], rv = (int32_t)s_vals[right];
                                keep = lv != rv && value_legal(sm, right, lv) && value_legal(sm, left, rv);
                            }
                            w0 = left;
                            w1 = right;
                        }
                        g.b += 64;
                        if (g.b >= ns) {
                            g.b = 0;
                            g.a += 1;
                            if (g.a >= ns) g.done = 1;
                        }
                    } else if (!FAST && (kind == 2048 || kind == 4096)) {  // ---- nearby scalar change / swap: one row per call ----
                        // (scalar_neighborhood/cursor/change.rs:312-374, cursor/swap.rs:346-398) the row is ranked already; the
                        // first max_nearby entries that pass the state filter are the stable top-k, then apply_selection_order
                        if (g.a >= ns) {
                            g.done = 1;
                            break;
                        }
                        const int which = kind == 2048 ? 0 : 1;
                        const uint32_t e = ns <= 1 ? 0u : fastmod_u64((uint64_t)g.c + (uint64_t)g.a * g.d, fm_n);
                        const int32_t cv_ = (int32_t)s_vals[e];
                        const uint32_t r0 = gl.ns_off[which][e], r1 = gl.ns_off[which][e + 1];
                        uint32_t cnt = 0;
                        for (uint32_t b0 = r0; b0 < r1 && cnt < leaf_max_nearby; b0 += 64) {
                            const uint32_t k = b0 + lane;
                            int32_t c = -1;
                            bool ok = false;
                            if (k < r1) {
                                c = gl.ns_val[which][k];
                                if (which == 0) {
                                    ok = c != cv_ && (!gl.ns_dynamic || value_legal(sm, e, c));
                                } else if ((uint32_t)c < ns && (gl.ns_dynamic ? (uint32_t)c != e : (uint32_t)c > e)) {
                                    const int32_t rv = (int32_t)s_vals[c];
                                    ok = rv != cv_ && value_legal(sm, e, rv) && value_legal(sm, (uint32_t)c, cv_);
                                }
                            }
                            const uint64_t okm = __ballot(ok);
                            const uint32_t rank = cnt + mbcnt64(okm);
                            if (ok && rank < leaf_max_nearby) ns_tmp[rank] = (uint16_t)c;
                            cnt += (uint32_t)__popcll(okm);
                        }
                        cnt = uni(cnt < leaf_max_nearby ? cnt : leaf_max_nearby);
                        wave_sync();
                        if (lane < cnt) {
                            const uint32_t idx = ctx.selection_index(lane, cnt, (which == 0 ? SALT_NSC_VALUE : SALT_NSW_TARGET) ^ (uint64_t)e ^ identity);
                            keep = true;
                            w0 = e;
                            w1 = (uint32_t)ns_tmp[idx];
                        } else if (which == 0 && lane == cnt && sm.allows_unassigned && cv_ >= 0) {  // the row's one to-None candidate
                            keep = true;
                            w0 = e;
                            w1 = 0xFFFFFFFFu;
                        }
                        wave_sync();  // the survivors are consumed before the next row overwrites them
                        g.a += 1;
                        if (g.a >= ns) g.done = 1;
                    } else if (!FAST && kind == 8192) {  // ---- list permute (list_kernel/permute.rs:103-205): the (size, permutation) pairs of one start per call ----
                        const uint32_t mn = leaf_min, mx = leaf_max;
                        uint32_t ent = 0, len = 0, start = 0, size_count = 0;
                        for (;;) {  // the current start with at least one window size
                            if (g.a >= (uint32_t)V) break;
                            ent = fastmod_u64((uint64_t)g.c + (uint64_t)g.a * g.d, fm_V);
                            len = rlen(ent);
                            if (len < mn || g.b >= len) {
                                g.a += 1;
                                g.b = 0;
                                g.e = 0;
                                continue;
                            }
                            start = ctx.selection_index(g.b, len, SALT_PM_START ^ (uint64_t)ent ^ ldesc);
                            const uint32_t max_valid = mx < len - start ? mx : len - start;
                            if (max_valid < mn) {
                                g.b += 1;
                                g.e = 0;
                                continue;
                            }
                            size_count = max_valid - mn + 1;
                            break;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        // flat offset f = g.e + lane over the sizes in stream order, factorial(size) - 1 permutations each
                        const uint32_t f = g.e + lane;
                        uint32_t cum = 0, my_size = 0, my_po = 0;
                        bool found = false;
                        for (uint32_t so = 0; so < size_count; ++so) {
                            const uint32_t sz = mn + ctx.selection_index(so, size_count, SALT_PM_SIZE ^ (uint64_t)ent ^ (uint64_t)start);
                            const uint32_t cnt = permute_factorial(sz) - 1u;
                            if (!found && f < cum + cnt) {
                                found = true;
                                my_size = sz;
                                my_po = f - cum;
                            }
                            cum += cnt;
                        }
                        if (found) {
                            const uint32_t rank = ctx.selection_index(my_po, permute_factorial(my_size) - 1u,
                                                                      SALT_PM_ORDER ^ (uint64_t)ent ^ (uint64_t)start ^ (uint64_t)my_size ^ ldesc) + 1u;
                            keep = true;
                            w0 = (ent << 16) | start;
                            w1 = (my_size << 16) | rank;
                        }
                        g.e += 64;
                        if (g.e >= cum) {
                            g.b += 1;
                            g.e = 0;
                        }
                    } else if (DBGK(4) && kind == 4) {  // ---- list change (list_kernel/change.rs:142-241) ----
                        // advance to a source with a non-empty list
                        uint32_t se = 0, slen = 0;
                        for (;;) {
                            if (g.a >= (uint32_t)V) break;
                            se = lc_ent(g.a);
                            slen = rlen(se);
                            if (g.b < slen) break;
                            g.a += 1;
                            g.b = 0;
                            g.c = 0;
                            g.d = 0;
                            g.e = 0;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        const uint32_t sp = ctx.selection_index(g.b, slen, SALT_LC_SOURCE ^ (uint64_t)se ^ ldesc);
                        if (g.c == 0) {  // intra destinations 0..=slen, skipping dp == sp and dp == sp + 1
                            const uint32_t o = g.e + lane;
                            if (o <= slen) {
                                const uint32_t dp = ctx.selection_index(o, slen + 1, SALT_LC_INTRA ^ (uint64_t)se ^ (uint64_t)sp);
                                keep = dp != sp && dp != sp + 1;
                                w0 = (se << 16) | sp;
                                w1 = (se << 16) | dp;
                            }
                            g.e += 64;
                            if (g.e > slen) {
                                g.c = 1;
                                g.d = 0;
                                g.e = 0;
                            }
                        } else {  // inter: every other entity in order, positions 0..=dlen
                            if (g.d == g.a) {
                                g.d += 1;
                                g.e = 0;
                            }
                            if (g.d >= (uint32_t)V) {  // next source position
                                g.b += 1;
                                g.c = 0;
                                g.d = 0;
                                g.e = 0;
                                st_sources -= 1;
                                continue;
                            }
                            const uint32_t de = lc_ent(g.d);
                            const uint32_t dlen = rlen(de);
                            const uint32_t o = g.e + lane;
                            if (o <= dlen) {
                                const uint32_t dp = ctx.selection_index(o, dlen + 1, SALT_LC_INTER ^ (uint64_t)se ^ (uint64_t)de ^ (uint64_t)sp);
                                keep = true;
                                w0 = (se << 16) | sp;
                                w1 = (de << 16) | dp;
                            }
                            g.e += 64;
                            if (g.e > dlen) {
                                g.d += 1;
                                g.e = 0;
                            }
                        }
                    } else if (DBGK(256) && kind == 256) {  // ---- sublist swap (list_kernel/sublist_swap.rs:57-101,228-300) ----
                        const uint32_t mn = leaf_min, mx = leaf_max;
                        uint32_t fent = 0, flen = 0, fstart = 0, sc1 = 0;
                        for (;;) {  // the current first segment
                            if (g.a >= (uint32_t)V) break;
                            fent = sw_ent(g.a);
                            flen = rlen(fent);
                            if (flen < mn || g.b >= flen) {
                                g.a += 1;
                                g.b = 0;
                                g.f = 0;
                                g.d = g.a;
                                g.e = 0;
                                continue;
                            }
                            fstart = ctx.selection_index(g.b, flen, SALT_SS_START ^ (uint64_t)fent ^ ldesc);
                            const uint32_t max_valid = mx < flen - fstart ? mx : flen - fstart;
                            sc1 = max_valid >= mn ? max_valid - mn + 1 : 0u;
                            if (g.f >= sc1) {  // no (more) sizes at this start
                                g.b += 1;
                                g.f = 0;
                                g.d = g.a;
                                g.e = 0;
                                continue;
                            }
                            break;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        const uint32_t fsize = mn + ctx.selection_index(g.f, sc1, SALT_SS_SIZE ^ (uint64_t)fent ^ (uint64_t)fstart);
                        const uint32_t fend = fstart + fsize;
                        if (g.d < g.a) g.d = g.a;
                        if (g.d >= (uint32_t)V) {  // every partner of this first segment is out: next first segment
                            g.f += 1;
                            g.d = g.a;
                            g.e = 0;
                            st_sources -= 1;
                            continue;
                        }
                        // lanes = (second start, size offset) pairs in cursor order; lane group k looks at partner rank g.d + k, so
                        // the 64 / S start slots of one call span several short partner lists
                        const uint32_t S = mx - mn + 1, per = 64u / S;
                        const uint32_t u = lane / S, q = lane % S;
                        const uint32_t rk = g.d + lane;
                        uint32_t sent_k = 0, slen_k = 0;
                        if (rk < (uint32_t)V) {
                            sent_k = fastmod_u64((uint64_t)sw_st + (uint64_t)rk * sw_sd, fm_V);
                            slen_k = s_off[sent_k + 1] - s_off[sent_k];
                            if (slen_k < mn) slen_k = 0;
                        }
                        const uint32_t cnt_k = lane == 0 ? (slen_k > g.e ? slen_k - g.e : 0u) : slen_k;
                        uint32_t grp, so, total;
                        map_slots_to_groups(cnt_k, u, grp, so, total);
                        const uint32_t sent = (uint32_t)__shfl((int)sent_k, (int)grp);
                        const uint32_t slen = (uint32_t)__shfl((int)slen_k, (int)grp);
                        if (grp == 0) so += g.e;
                        const uint32_t nslots = total < per ? total : per;
                        if (u < nslots) {
                            const uint32_t sstart = ctx.selection_index(so, slen, SALT_SS_START ^ (uint64_t)sent ^ ldesc);
                            const uint32_t mv2 = mx < slen - sstart ? mx : slen - sstart;
                            if (mv2 >= mn && q < mv2 - mn + 1) {
                                const uint32_t ssize = mn + ctx.selection_index(q, mv2 - mn + 1, SALT_SS_SIZE ^ (uint64_t)sent ^ (uint64_t)sstart);
                                keep = !(g.d + grp == g.a && (sstart < fend || (fstart == sstart && fend == sstart + ssize)));
                                w0 = (fent << 16) | fstart;
                                w1 = (sent << 16) | sstart;
                                wx = fsize | (ssize << 4);
                            }
                        }
                        if (total <= per) {  // partner ranks g.d .. g.d + 63 consumed
                            g.d += 64;
                            g.e = 0;
                        } else {  // resume after the last start slot of this call
                            const int last_lane = (int)((per - 1) * S);
                            const uint32_t lg = uni((uint32_t)__shfl((int)grp, last_lane)), lo_ = uni((uint32_t)__shfl((int)so, last_lane));
                            const uint32_t ll = uni((uint32_t)__shfl((int)slen, last_lane));
                            if (lo_ + 1 >= ll) {
                                g.d += lg + 1;
                                g.e = 0;
                            } else {
                                g.d += lg;
                                g.e = lo_ + 1;
                            }
                        }
                    } else if (DBGK(128) && kind == 128) {  // ---- sublist change / Or-opt (list_kernel/sublist_change.rs:109-266) ----
                        const uint32_t mn = leaf_min, mx = leaf_max;
                        uint32_t ent = 0, len = 0, start = 0, sc = 0;
                        for (;;) {  // current segment start with at least one legal size
                            if (g.a >= (uint32_t)V) break;
                            ent = sb_ent(g.a);
                            len = rlen(ent);
                            if (len < mn || g.b >= len) {
                                g.a += 1;
                                g.b = 0;
                                g.f = 0;
                                g.c = 0;
                                g.d = 0;
                                g.e = 0;
                                continue;
                            }
                            start = ctx.selection_index(g.b, len, SALT_SC_START ^ (uint64_t)ent ^ ldesc);
                            const uint32_t max_valid = mx < len - start ? mx : len - start;
                            sc = (max_valid > mn ? max_valid - mn : 0u) + (max_valid >= mn ? 1u : 0u);
                            if (sc != 0) break;
                            g.b += 1;
                            g.f = 0;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        const uint32_t z = mn + ctx.selection_index(g.f, sc, SALT_SC_SIZE ^ (uint64_t)ent ^ (uint64_t)start);
                        bool segment_done = false;
                        if (g.c == 0) {  // intra destinations 0..=(len - z) in post-removal coordinates, except `start`
                            const uint32_t post = len - z;
                            const uint32_t o = g.e + lane;
                            if (o <= post) {
                                const uint32_t dp = ctx.selection_index(o, post + 1, SALT_SC_INTRA ^ (uint64_t)ent ^ (uint64_t)start);
                                keep = dp != start;
                                w0 = (ent << 16) | start;
                                w1 = (ent << 16) | dp;
                            }
                            g.e += 64;
                            if (g.e > post) {
                                g.c = 1;
                                g.d = 0;
                                g.e = 0;
                            }
                        } else {
                            // inter destinations: every other entity in leaf order, positions 0..=dlen.  Lane k looks at
                            // destination rank g.d + k, the slots of up to 64 short lists are laid onto the lanes at once.
                            if (g.d >= (uint32_t)V) {
                                segment_done = true;
                            } else {
                                const uint32_t rk = g.d + lane;
                                uint32_t de_k = 0, full_k = 0;
                                if (rk < (uint32_t)V && rk != g.a) {
                                    de_k = fastmod_u64((uint64_t)sb_st + (uint64_t)rk * sb_sd, fm_V);
                                    full_k = s_off[de_k + 1] - s_off[de_k] + 1;
                                }
                                const uint32_t cnt_k = (lane == 0 && full_k) ? full_k - g.e : full_k;  // g.e < full of rank g.d
                                uint32_t grp, o, total;
                                map_slots_to_groups(cnt_k, lane, grp, o, total);
                                const uint32_t de = (uint32_t)__shfl((int)de_k, (int)grp);
                                const uint32_t slots = (uint32_t)__shfl((int)full_k, (int)grp);
                                if (grp == 0) o += g.e;
                                if (lane < total) {
                                    const uint32_t dp = ctx.selection_index(o, slots, SALT_SC_INTER ^ (uint64_t)ent ^ (uint64_t)de ^ (uint64_t)start);
                                    keep = true;
                                    w0 = (ent << 16) | start;
                                    w1 = (de << 16) | dp;
                                }
                                if (total <= 64) {  // all 64 ranks consumed
                                    g.d += 64;
                                    g.e = 0;
                                } else {  // resume after lane 63's slot
                                    const uint32_t lg = uni((uint32_t)__shfl((int)grp, 63)), lo_ = uni((uint32_t)__shfl((int)o, 63));
                                    const uint32_t ls = uni((uint32_t)__shfl((int)slots, 63));
                                    if (lo_ + 1 >= ls) {
                                        g.d += lg + 1;
                                        g.e = 0;
                                    } else {
                                        g.d += lg;
                                        g.e = lo_ + 1;
                                    }
                                }
                            }
                        }
                        wx = z;
                        if (segment_done) {  // advance_segment
                            g.f += 1;
                            if (g.f >= sc) {
                                g.f = 0;
                                g.b += 1;
                            }
                            g.c = 0;
                            g.d = 0;
                            g.e = 0;
                            st_sources -= 1;
                            continue;
                        }
                    } else if (DBGK(16) && (kind == 16 || kind == 32)) {  // ---- nearby list change / swap: one source per call ----
                        if (g.e == 0) {
                            g.done = 1;
                            break;
                        }
                        int ni = 0;  // which nearby table set this leaf owns
                        for (int l2 = 0; l2 < l; ++l2) {
                            const int k2 = lt.geti(l2, LeafTab::KIND);
                            if (k2 == 16 || k2 == 32) ni += 1;
                        }
                        const uint16_t* ra = nb_route_at + ni * V;
                        const uint16_t* ro = nb_rank_of + ni * V;
                        const uint16_t* sb = nb_slot_base + ni * (V + 1);
                        uint16_t* spv = nb_spvec + ni * 64;
                        uint32_t se = 0, len = 0;
                        for (;;) {  // skip empty routes (sources left > 0 guarantees one exists)
                            se = uni((uint32_t)ra[g.a]);
                            len = rlen(se);
                            if (g.b < len) break;
                            g.a += 1;
                            g.b = 0;
                        }
                        if (g.a != g.c || (g.b & ~63u) != g.d) {
                            const uint64_t src_salt = (kind == 16 ? SALT_NEARBY_CHANGE_SOURCE : SALT_NEARBY_SWAP_SOURCE) ^ (uint64_t)se ^ ldesc;
                            const uint32_t oo = (g.b & ~63u) + lane;
                            spv[lane] = (uint16_t)(oo < len ? ctx.selection_index(oo, len, src_salt) : 0u);
                            g.c = g.a;
                            g.d = g.b & ~63u;
                            wave_sync();
                        }
                        const uint32_t sp = uni((uint32_t)spv[g.b & 63u]);
                        const uint32_t sx = uni((uint32_t)s_visits[s_off[se] + sp]);
                        const uint32_t key0 = lane < (uint32_t)lm.dim ? (uint32_t)nb.keys[(size_t)sx * (uint32_t)lm.dim + lane] : NBR_END;
                        tl += nearby_source_to_ring(lm, nb, kind == 16, se, sp, len, g.a, sx, node_slot, s_off, sb, ro, rq, GRC - 1, tl,
                                                    key0, 0u, leaf_max_nearby, 0u);
                        g.b += 1;
                        g.e -= 1;
                        if (g.e == 0) g.done = 1;
                    } else if (DBGK(1) && kind == 512 && leaf_max_nearby == 0) {  // ---- 3-opt, full enumeration (k_opt/full.rs:62-92) ----
                        const uint32_t mseg = leaf_min;
                        uint32_t ent = 0, len = 0;
                        uint64_t mc = 0, mo = 0;
                        for (;;) {
                            if (g.a >= (uint32_t)V) break;
                            ent = ko_ent(g.a);
                            len = rlen(ent);
                            mc = kopt_cut_count(len, mseg) * 7ull;
                            mo = ((uint64_t)g.c << 32) | g.b;
                            if (mo < mc) break;
                            g.a += 1;
                            g.b = 0;
                            g.c = 0;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        const uint64_t o = mo + lane;
                        if (o < mc) {
                            const uint64_t selected = kopt_selection_index64(ctx, o, mc, SALT_KF_MOVE ^ ldesc ^ (uint64_t)ent);
                            uint32_t c1, c2, c3;
                            kopt_unrank(len, mseg, selected / 7ull, c1, c2, c3);
                            keep = true;
                            w0 = (ent << 16) | c1;
                            w1 = (c2 << 16) | c3;
                            wx = (uint32_t)(selected % 7ull);
                        }
                        mo += 64;
                        g.b = (uint32_t)mo;
                        g.c = (uint32_t)(mo >> 32);
                    } else if (DBGK(512) && kind == 512) {  // ---- 3-opt, distance-pruned (k_opt/nearby.rs:106-148, nearby_state.rs) ----
                        const KoptLds km(mem + cv.kopt);
                        const KoptEnv env{&lm,   s_visits, s_off, km, gl.kopt_scratch + (size_t)r * lm.n_cap, ctx, ldesc,
                                          leaf_min, leaf_max_nearby, lane};
                        KoptS ks;
                        kopt_load_state(km.st, ks);
                        uint32_t ntr = 0;
                        while (ntr < KOPT_TRIPLES) {
                            if (!ks.active) {  // load_next_cut_state: the next entity whose route admits three cuts
                                bool opened = false;
                                while (g.a < (uint32_t)V) {
                                    const uint32_t ent = ko_ent(g.a);
                                    g.a += 1;
                                    kopt_open_entity(env, ks, ent, rlen(ent));
                                    if (!ks.done) {
                                        opened = true;
                                        break;
                                    }
                                }
                                if (!opened) {
                                    ks.active = 0;
                                    g.done = 1;
                                    break;
                                }
                            }
                            uint32_t c1 = 0, c2 = 0, c3 = 0;
                            if (kopt_next_cuts(env, ks, c1, c2, c3)) {
                                if (lane == 0) {
                                    km.trip[ntr * 4] = (uint16_t)ks.entity;
                                    km.trip[ntr * 4 + 1] = (uint16_t)c1;
                                    km.trip[ntr * 4 + 2] = (uint16_t)c2;
                                    km.trip[ntr * 4 + 3] = (uint16_t)c3;
                                }
                                ntr += 1;
                            } else {
                                ks.active = 0;
                            }
                        }
                        kopt_store_state(km.st, ks, lane);
                        wave_sync();
                        const uint32_t t = lane / 7u, q = lane % 7u;
                        if (t < ntr) {
                            const uint32_t ent = km.trip[t * 4], c1 = km.trip[t * 4 + 1], c2 = km.trip[t * 4 + 2], c3 = km.trip[t * 4 + 3];
                            keep = true;
                            w0 = (ent << 16) | c1;
                            w1 = (c2 << 16) | c3;
                            wx = ctx.selection_index(q, 7u, kopt_pattern_salt(ldesc, ent, c1, c2, c3));
                        }
                        wave_sync();  // the triples are consumed before the next call overwrites them
                    } else if (PREC && kind == 16384) {  // ---- critical-path precedence leaf (precedence/cursor.rs:182-252): one candidate per call,
                                                         // decoded, applied, scored (and pruned when cyclic) right here.  g.a = stage, g.b = offset
                                                         // inside the stage, g.c = block offset ----
                        // the next candidate of the stream: 0 = decoded into (c0, c1, m), 1 = a stage / block ended (call again), 2 = the stream ended
                        auto plf_next = [&](GGen& gg, uint32_t& c0, uint32_t& c1, PlfMove& m) -> int {
                            if (gg.a == 0) {  // stream offset = gg.d : gg.b (64 bits), ring entry = the selected index (high 30 bits in c0, low 32 in c1)
                                const uint64_t so = ((uint64_t)gg.d << 32) | gg.b;
                                if (so >= plf.ms_count) {
                                    gg.a = 1, gg.b = 0, gg.d = 0;
                                    return 1;
                                }
                                const uint64_t si = (plf.ms_count <= 0xFFFFFFFFull && !gl.plf.force64) ? (uint64_t)ctx.selection_index((uint32_t)so, (uint32_t)plf.ms_count, SALT_PL_MULTI_SWAP ^ ldesc)
                                                                                                  : kopt_selection_index64(ctx, so, plf.ms_count, SALT_PL_MULTI_SWAP ^ ldesc);
                                gg.b += 1;
                                if (gg.b == 0) gg.d += 1;
                                c1 = (uint32_t)si;
                                c0 = (uint32_t)(si >> 32);  // stage 0 in bits 30..31
                                plf_decode_multi_swap(plf, si, m);
                            } else if (gg.a == 1) {
                                if (gg.b >= plf.mr_count) {
                                    gg.a = 2, gg.b = 0, gg.c = 0;
                                    return 1;
                                }
                                c1 = ctx.selection_index(gg.b, plf.mr_count, SALT_PL_MULTI_RUIN ^ ldesc);
                                gg.b += 1;
                                c0 = 1u << 30;
                                plf_decode_multi_ruin(plf, c1, m);
                            } else {
                                if (gg.c >= plf.nb) return 2;
                                const uint32_t bi = ctx.selection_index(gg.c, plf.nb, SALT_PL_BLOCK ^ ldesc);
                                const PlfBlock bl = plf_block(plf, bi);
                                if (gg.b >= bl.moves()) {
                                    gg.c += 1, gg.b = 0;
                                    return 1;
                                }
                                c1 = plf_tiered_index(ctx, bl, gg.b, SALT_PL_MOVE ^ ldesc ^ (uint64_t)bl.e ^ ((uint64_t)bl.start << 16) ^ ((uint64_t)(bl.start + bl.len - 1) << 32));
                                gg.b += 1;
                                c0 = (2u << 30) | bi;
                                plf_decode_block(bl, c1, m);
                            }
                            c0 = uni(c0), c1 = uni(c1);
                            return 0;
                        };
                        PlfMove pm_;
                        const int nx = plf_next(g, w0, w1, pm_);
                        if (nx == 1) {
                            st_sources -= 1;
                            continue;
                        }
                        if (nx == 2) {
                            g.done = 1;
                            break;
                        }
                        if (pgrp_T && pm_.kind != 8) {  // up to T consecutive candidates (no ruins: those recreate on the lists) in one pass
                            const uint32_t my_g = lane >> pgrp_shift;
                            PgrpMove gm;
                            gm.kind = 0, gm.a = gm.i = gm.b = gm.j = gm.ext = gm.el2 = 0;
                            uint32_t bw0 = 0, bw1 = 0;
                            for (uint32_t q = 0;;) {
                                if (my_g == q) {
                                    gm.kind = (uint32_t)pm_.kind, gm.a = pm_.a, gm.i = pm_.ap, gm.b = pm_.b, gm.j = pm_.bp, gm.ext = pm_.ext, gm.el2 = pm_.el[2];
                                    bw0 = w0, bw1 = w1;
                                }
                                if (++q >= pgrp_T) break;
                                const GGen before = g;
                                if (plf_next(g, w0, w1, pm_) != 0 || pm_.kind == 8) {  // a stage boundary or a ruin: the next call's
                                    g = before;
                                    break;
                                }
                            }
                            int64_t gp = 0, gmk = 0;
                            bool gcyc = false;
                            pgrp_eval(gm, gp, gmk, gcyc);
                            keep = gm.kind != 0 && (lane & ((1u << pgrp_shift) - 1u)) == 0 && !gcyc;  // cyclic candidates are pruned
                            w0 = bw0, w1 = bw1;
                            const uint64_t km_ = __ballot(keep);
                            if (keep) {
                                const size_t slot = (size_t)((tl + mbcnt64(km_)) & (GRC - 1)) * 4;
#pragma unroll
                                for (int kk = 0; kk < L; ++kk) {
                                    int64_t v = cur[kk];
                                    if (kk == gl.prec.hard_level) v -= gp - prec_pen;
                                    if (kk == gl.prec.mk_level) v -= gmk - prec_mk;
                                    plf_score[slot + kk] = v;
                                }
                            }
                        } else {
                            ScoreV<L> psc;
                            if (plf_trial(pm_, psc)) {
                                keep = lane == 0;
                                if (lane == 0) {
#pragma unroll
                                    for (int kk = 0; kk < L; ++kk) plf_score[(size_t)(tl & (GRC - 1)) * 4 + kk] = psc.v[kk];
                                }
                            }
                        }
                    } else if (RUIN && kind == 1024) {  // ---- list ruin (list_kernel/ruin.rs:127-144): one candidate per call, scored right here ----
                        if (g.a >= (uint32_t)gl.ruin.moves_per_step) {
                            g.done = 1;
                            break;
                        }
                        {  // scored at the start of the step (above): the ring entries name the candidates
                            const uint32_t n_left = (uint32_t)gl.ruin.moves_per_step - g.a;
                            keep = lane < n_left;
                            w0 = g.a + lane;
                            g.a += n_left;
                            g.done = 1;
                        }
                    } else if (DBGK(64) && kind == 64) {  // ---- list reverse / 2-opt (list_kernel/reverse.rs:68-108) ----
                        uint32_t ent = 0, len = 0;
                        for (;;) {  // entities shorter than two elements are skipped
                            if (g.a >= (uint32_t)V) break;
                            ent = lr_ent(g.a);
                            len = rlen(ent);
                            if (len >= 2 && g.b < len) break;
                            g.a += 1;
                            g.b = 0;
                            g.e = 0;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        // lane k looks at start offset g.b + k of this entity; the (start, end) pairs of up to 64 starts are laid
                        // onto the lanes at once (a short list's whole neighbourhood in one call)
                        const uint32_t bo = g.b + lane;
                        uint32_t start_k = 0, full_k = 0;
                        if (bo < len) {
                            start_k = ctx.selection_index(bo, len, SALT_LR_START ^ (uint64_t)ent ^ ldesc);
                            full_k = len > start_k + 1 ? len - (start_k + 1) : 0u;
                        }
                        // g.e <= end_count of start offset g.b (== only when that count is 0)
                        const uint32_t cnt_k = lane == 0 ? (full_k > g.e ? full_k - g.e : 0u) : full_k;
                        uint32_t grp, o, total;
                        map_slots_to_groups(cnt_k, lane, grp, o, total);
                        const uint32_t start = (uint32_t)__shfl((int)start_k, (int)grp);
                        const uint32_t end_count = (uint32_t)__shfl((int)full_k, (int)grp);
                        if (grp == 0) o += g.e;
                        if (lane < total) {
                            const uint32_t end = start + 2 + ctx.selection_index(o, end_count, SALT_LR_END ^ (uint64_t)ent ^ (uint64_t)start);
                            keep = true;
                            w0 = (ent << 16) | start;
                            w1 = (ent << 16) | end;
                        }
                        if (total <= 64) {  // start offsets g.b .. g.b + 63 consumed
                            g.b += 64;
                            g.e = 0;
                        } else {
                            const uint32_t lg = uni((uint32_t)__shfl((int)grp, 63)), lo_ = uni((uint32_t)__shfl((int)o, 63));
                            const uint32_t lc = uni((uint32_t)__shfl((int)end_count, 63));
                            if (lo_ + 1 >= lc) {
                                g.b += lg + 1;
                                g.e = 0;
                            } else {
                                g.b += lg;
                                g.e = lo_ + 1;
                            }
                        }
                    } else if (DBGK(8)) {  // ---- list swap (list_kernel/swap.rs) ----
                        uint32_t fe = 0, flen = 0;
                        for (;;) {  // entities with an empty list are skipped
                            if (g.a >= (uint32_t)V) break;
                            fe = ls_ent(g.a);
                            flen = rlen(fe);
                            if (flen != 0) break;
                            g.a += 1;
                            g.b = 0;
                            g.c = 0;
                            g.e = 0;
                            g.d = g.a + 1;
                        }
                        if (g.a >= (uint32_t)V) {
                            g.done = 1;
                            break;
                        }
                        if (g.c == 0) {  // intra pairs first < second
                            if (g.b >= flen) {
                                g.c = 1;
                                g.d = g.a + 1;
                                g.b = 0;
                                g.e = 0;
                                st_sources -= 1;
                                continue;
                            }
                            const uint32_t fp = ctx.selection_index(g.b, flen, SALT_LS_FIRST ^ (uint64_t)fe ^ ldesc);
                            const uint32_t second_count = flen > fp + 1 ? flen - (fp + 1) : 0u;
                            const uint32_t o = g.e + lane;
                            if (o < second_count) {
                                const uint32_t sp = fp + 1 + ctx.selection_index(o, second_count, SALT_LS_SECOND ^ (uint64_t)fe ^ (uint64_t)fp);
                                keep = true;
                                w0 = (fe << 16) | fp;
                                w1 = (fe << 16) | sp;
                            }
                            g.e += 64;
                            if (g.e >= second_count) {
                                g.b += 1;
                                g.e = 0;
                            }
                        } else {  // inter with entities after this one in list order
                            uint32_t se2 = 0, slen2 = 0;
                            for (;;) {
                                if (g.d >= (uint32_t)V) break;
                                se2 = ls_ent(g.d);
                                slen2 = rlen(se2);
                                if (slen2 != 0) break;
                                g.d += 1;
                            }
                            if (g.d >= (uint32_t)V) {  // advance_entity
                                g.a += 1;
                                g.c = 0;
                                g.b = 0;
                                g.e = 0;
                                g.d = g.a + 1;
                                st_sources -= 1;
                                continue;
                            }
                            if (g.b >= flen) {
                                g.d += 1;
                                g.b = 0;
                                g.e = 0;
                                st_sources -= 1;
                                continue;
                            }
                            const uint32_t fp = ctx.selection_index(g.b, flen, SALT_LS_IFIRST ^ (uint64_t)fe ^ (uint64_t)se2);
                            const uint32_t o = g.e + lane;
                            if (o < slen2) {
                                const uint32_t sp = ctx.selection_index(o, slen2, SALT_LS_ISECOND ^ (uint64_t)fe ^ (uint64_t)se2 ^ (uint64_t)fp);
                                keep = true;
                                w0 = (fe << 16) | fp;
                                w1 = (se2 << 16) | sp;
                            }
                            g.e += 64;
                            if (g.e >= slen2) {
                                g.b += 1;
                                g.e = 0;
                            }
                        }
                    }
                    if (PREC && plf_policy && kind >= 4 && kind != 512 && kind != 1024 && kind != 2048 && kind != 4096 && kind != 16384) {
                        // runtime slot with precedence hooks: intra-list candidates that close a cycle through the route graph never reach the ring
                        uint64_t chk = __ballot(keep && (kind == 64 || kind == 8192 || (w0 >> 16) == (w1 >> 16)));
                        cache_pen = INT64_MIN, cache_mk = 0;
                        while (pgrp_T && !plf_cur_cyclic && chk) {  // acyclic working state: dropped <=> the lists are cyclic afterwards
                            int64_t tp, tm_;
                            bool tc;
                            if (pgrp_batch(chk, kind, w0, w1, wx, tp, tm_, tc)) {
                                cache_pen = tp, cache_mk = tm_;  // what the replay would evaluate again
                                if (tc) keep = false;
                            }
                        }
                        while (chk) {
                            const int ci = __ffsll((unsigned long long)chk) - 1;
                            chk &= chk - 1;
                            const uint32_t ca = (uint32_t)__builtin_amdgcn_readlane((int)w0, ci), cb = (uint32_t)__builtin_amdgcn_readlane((int)w1, ci);
                            const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)wx, ci);
                            const uint32_t e = ca >> 16;
                            if (kind == 8192)
                                apply_list_move_wave(lm, s_visits, s_off, s_load, 9, e, ca & 0xFFFFu, e, (ca & 0xFFFFu) + (cb >> 16), cb & 0xFFFFu);
                            else
                                apply_list_move_wave(lm, s_visits, s_off, s_load, list_move_kind_of(kind), e, ca & 0xFFFFu, cb >> 16, cb & 0xFFFFu,
                                                     list_move_ext_of(kind, ca, cb, cx));
                            wave_sync();
                            bool cyc;
                            const PrecResult fpr = plf_eval(cyc, nullptr);
                            if ((int)lane == ci) cache_pen = fpr.penalty, cache_mk = fpr.makespan;  // what the replay would evaluate again
                            const bool drop = plf_closes_cycle(e, cyc);
                            const uint32_t lo = uni(s_off[e]), hi = uni(s_off[e + 1]);
                            for (uint32_t t = lo + lane; t < hi; t += 64) s_visits[t] = (uint16_t)g_visits[t];
                            wave_sync();
                            if (drop && (int)lane == ci) keep = false;
                        }
                    }
                    const uint64_t km = __ballot(keep);
                    if (keep) {
                        const uint32_t qi = (tl + mbcnt64(km)) & (GRC - 1);
                        rq[qi * 2] = w0;
                        rq[qi * 2 + 1] = w1;
                        ringx[l * GRC + qi] = (uint8_t)wx;
                        if (PREC && plf_policy) {
                            plf_cache[((size_t)l * GRC + qi) * 2] = cache_pen;
                            plf_cache[((size_t)l * GRC + qi) * 2 + 1] = cache_mk;
                        }
                    }
                    tl += (uint32_t)__popcll(km);
                }
                lt.put_gen(l, g);
                lt.set(l, LeafTab::TAIL, tl);
#ifdef SF_PHASE_PREC  // precedence breakdown: 1 critical-path leaf, 2 permute / change / swap, 3 reverse, 4-7 inside the recreate (PHR), 0 the rest
                PH(kind == 16384 ? 1 : ((kind == 8192 || kind == 4 || kind == 8) ? 2 : (kind == 64 ? 3 : (kind == 1024 ? 7 : 0))))
#elif defined(SF_PHASE_PROFILE)
                PH((kind == 128 || kind == 256) ? 2 : ((kind == 64 || kind == 1024) ? 3 : (kind == 512 ? 4 : 1)))
#endif
            }
            ring_sync();
            if (pre_eval) {
                // ---- scoring stage: the entries this fill round appended, one leaf (= one move kind) at a time ----
                bool any_new = false;
                for (int l = 0; l < nl; ++l) {
                    const int kind = lt.geti(l, LeafTab::KIND);
                    if (RUIN && kind == 1024) continue;  // (list ruin: scored when the step started)
                    const uint32_t t0 = lt.get(l, LeafTab::TAKEN), t1 = lt.get(l, LeafTab::TAIL);
                    for (uint32_t b0 = t0; (int32_t)(t1 - b0) > 0; b0 += 64) {
                        const uint32_t t = b0 + lane;
                        if ((int32_t)(t1 - t) > 0) {
                            const uint32_t qi = t & (GRC - 1);
                            const uint32_t* rq = ring + ((size_t)l * GRC + qi) * 2;
                            const ListDelta d = eval_list_unified(lm, s_visits, s_off, s_load, kind, rq[0], rq[1], (uint32_t)ringx[l * GRC + qi]);
                            int32_t* rd = ringd + ((size_t)l * GRC + qi) * 2;
                            rd[0] = d.doable ? (int32_t)d.d_cap : INT32_MIN;
                            rd[1] = (int32_t)d.d_dist;
                        }
                        any_new = true;
                    }
                }
                if (any_new) ring_sync();
                PHS(6)  // (counted with the replay: it is the replay's trial scoring, moved)
            }

            // ---- C2: lay the next 64 pulls of the union scheduler onto the lanes ----
            uint32_t my_leaf = 0, my_idx = 0;
            for (int l = 0; l < nl; ++l) lt.set(l, LeafTab::TAKEN, 0);
            uint32_t nvalid = 0;
            bool need_more = false;
            {
                int nlive = __popc(~exmask & ((1u << nl) - 1u));
                // Fast path: with equal weights the smooth weighted round-robin is a plain cycle over the
                // live children in rotated order whenever all their running weights are equal (true at the
                // start of every step and after every whole cycle).  Lay out whole cycles directly; the
                // pull-by-pull simulation below handles partial cycles, exhaustion and refills.
                if (nl > 1 && nlive > 1 && !union_custom) {
                    // One cycle of the smooth weighted round-robin with equal weights pulls every live child once, in
                    // descending running weight (ties: rotated order), and leaves the weights as it found them -- provided
                    // max - min < live children (true with equal weights at the start of a step, and again a few pulls
                    // after a child ran dry).  Whole cycles are laid onto the lanes directly; the pull-by-pull simulation
                    // below only handles the transient after an exhaustion, partial cycles and refills.
                    const uint32_t livem = ~exmask & ((1u << nl) - 1u);
                    const int32_t wc = lane < (uint32_t)nl ? (int32_t)lt.w[lane * 16 + LeafTab::WCUR] : 0;  // lane l = leaf l
                    // wave-uniform: the live children's running weights and rotated positions through v_readlane (no LDS
                    // crossbar round trips), rank of every live child = how many live children are pulled before it
                    int32_t wmax = INT32_MIN, wmin = INT32_MAX;
                    uint64_t r_order = 0;  // live leaves by pull order inside a cycle, 4 bits each
                    const bool ro_hit = ro_live == livem;
                    if (!ro_hit) {
                        for (int j = 0; j < nl; ++j) {
                            if (!((livem >> j) & 1u)) continue;
                            const int32_t wj = __builtin_amdgcn_readlane(wc, j);
                            wmax = wj > wmax ? wj : wmax;
                            wmin = wj < wmin ? wj : wmin;
                        }
                    }
                    if (ro_hit || wmax - wmin < nlive) {
                        uint32_t seen = 0;
                        if (ro_hit) {
                            r_order = ro_cache;
                        } else {
                            for (uint32_t pi = 0; pi < (uint32_t)nl; ++pi) {      // rotated position pi holds leaf i
                                const uint32_t i = (uint32_t)(u_order >> (4u * pi)) & 15u;
                                if (!((livem >> i) & 1u)) continue;
                                if (wmax == wmin) {  // equal running weights (the steady state): the rotated order itself
                                    r_order |= (uint64_t)i << (4u * seen);
                                    seen += 1;
                                    continue;
                                }
                                const int32_t wi = __builtin_amdgcn_readlane(wc, (int)i);
                                uint32_t rank = 0;
                                for (uint32_t pj = 0; pj < (uint32_t)nl; ++pj) {
                                    const uint32_t j = (uint32_t)(u_order >> (4u * pj)) & 15u;
                                    if (!((livem >> j) & 1u)) continue;
                                    const int32_t wj = __builtin_amdgcn_readlane(wc, (int)j);
                                    rank += (wj > wi || (wj == wi && pj < pi)) ? 1u : 0u;
                                }
                                r_order |= (uint64_t)i << (4u * rank);
                            }
                            ro_cache = r_order;
                            ro_live = livem;
                        }
                        // my pull t = lane: cycle t / nlive, child = the (t % nlive)-th leaf of the cycle
                        const uint32_t cyc = lane / (uint32_t)nlive, slot = lane % (uint32_t)nlive;
                        const uint32_t leaf = (uint32_t)(r_order >> (4u * slot)) & 15u;
                        const uint32_t hd = lt.w[leaf * 16 + LeafTab::HEAD], tlq = lt.w[leaf * 16 + LeafTab::TAIL];  // per-lane leaf
                        const bool ok = (int32_t)(tlq - (hd + cyc)) > 0;
                        const uint64_t okm = __ballot(ok);
                        const uint32_t upto = okm == ~0ULL ? 64u : (uint32_t)(__ffsll((unsigned long long)~okm) - 1);
                        const uint32_t cycles = upto / (uint32_t)nlive;  // whole cycles only: the running weights stay as they are
                        if (cycles > 0) {
                            nvalid = cycles * (uint32_t)nlive;
                            if (lane < nvalid) {
                                my_leaf = leaf;
                                my_idx = hd + cyc;
                            }
                            for (int l = 0; l < nl; ++l)
                                if (!((exmask >> l) & 1u)) lt.set(l, LeafTab::TAKEN, cycles);
                        }
                    }
                }
                const bool fast_done = nvalid > 0;
                if (!fast_done && nlive > 0) {
                    // Pull-by-pull simulation with the leaf table in registers (lane l = leaf l): one pull is a handful of
                    // DPP / readlane instructions instead of a dozen dependent LDS round trips.
                    const bool isleaf = lane < (uint32_t)nl;
                    ro_live = 0xFFFFFFFFu;  // the running weights change below
                    int32_t wc = isleaf ? (int32_t)lt.w[lane * 16 + LeafTab::WCUR] : 0;
                    const uint32_t hd = isleaf ? lt.w[lane * 16 + LeafTab::HEAD] : 0u, tlv = isleaf ? lt.w[lane * 16 + LeafTab::TAIL] : 0u;
                    const uint32_t dn = isleaf ? lt.w[lane * 16 + LeafTab::DONE] : 1u;
                    uint32_t tk = 0;  // TAKEN was just cleared
                    uint32_t mypos = 0;
                    for (uint32_t pos = 0; pos < (uint32_t)nl; ++pos)
                        if (((uint32_t)(u_order >> (4u * pos)) & 15u) == lane) mypos = pos;
                    int32_t wgt = 1;  // child weight (UnionWeighting); lane l = leaf l
                    if (union_custom) {
                        wgt = 0;
#pragma unroll
                        for (int l = 0; l < GL; ++l)
                            if (lane == (uint32_t)l) wgt = gl.weight[l];
                    }
                    while (nvalid < 64 && nlive > 0) {
                        uint32_t sel = 0;
                        const bool live_l = isleaf && !((exmask >> lane) & 1u);
                        const uint32_t cur_before = u_cur;
                        if (nl > 1 && u_ord == 4) {  // StratifiedRandom: smooth weighted round-robin (vec_union.rs:334-362)
                            if (live_l) wc += wgt;
                            // max running weight, the earlier rotated position on ties: max of (weight << 4 | 15 - position) over
                            // lanes 0..15 by a DPP prefix max (row_shr 1, 2, 4, 8), read at lane 15
                            int32_t key = live_l ? (int32_t)(((uint32_t)wc << 4) | (15u - mypos)) : INT32_MIN;
                            key = max(key, __builtin_amdgcn_update_dpp(INT32_MIN, key, 0x111, 0xf, 0xf, false));
                            key = max(key, __builtin_amdgcn_update_dpp(INT32_MIN, key, 0x112, 0xf, 0xf, false));
                            key = max(key, __builtin_amdgcn_update_dpp(INT32_MIN, key, 0x114, 0xf, 0xf, false));
                            key = max(key, __builtin_amdgcn_update_dpp(INT32_MIN, key, 0x118, 0xf, 0xf, false));
                            const uint32_t kmax = (uint32_t)__builtin_amdgcn_readlane(key, 15);
                            sel = (uint32_t)(u_order >> (4u * (15u - (kmax & 15u)))) & 15u;
                        } else if (nl > 1 && u_ord == 0) {  // Sequential: drain the children in declaration order (:249-262)
                            while (u_cur < (uint32_t)nl && ((exmask >> u_cur) & 1u)) u_cur += 1;
                            sel = u_cur;
                        } else if (nl > 1 && u_ord <= 2) {  // RoundRobin / RotatingRoundRobin: the next live child (:264-283)
                            for (;;) {
                                sel = u_cur % (uint32_t)nl;
                                u_cur = (u_cur + 1u) % (uint32_t)nl;
                                if (!((exmask >> sel) & 1u)) break;
                            }
                        } else if (nl > 1) {  // Random: one seeded draw over the live weights per pull (:285-321)
                            const uint32_t draw = mod_u64(ctx.mixed_seed(0xA11CE5E1EC701000ULL + (uint64_t)u_draw), (uint32_t)live_weight);
                            u_draw += 1;
                            uint32_t cum = 0;
                            for (int i = 0; i < nl; ++i) {
                                if ((exmask >> i) & 1u) continue;
                                cum += (uint32_t)__builtin_amdgcn_readlane(wgt, i);
                                if (draw < cum) {
                                    sel = (uint32_t)i;
                                    break;
                                }
                            }
                        }
                        const uint32_t sel_taken = (uint32_t)__builtin_amdgcn_readlane((int)tk, (int)sel);
                        const uint32_t sel_head = (uint32_t)__builtin_amdgcn_readlane((int)hd, (int)sel);
                        const bool avail = (int32_t)((uint32_t)__builtin_amdgcn_readlane((int)tlv, (int)sel) - (sel_head + sel_taken)) > 0;
                        const bool gdone = __builtin_amdgcn_readlane((int)dn, (int)sel) != 0;
                        if (!avail && !gdone) {
                            // the child has more candidates that are not generated yet: undo this pull's bookkeeping and refill first
                            if (nl > 1 && u_ord == 4 && live_l) wc -= wgt;
                            if (u_ord == 1 || u_ord == 2) u_cur = cur_before;
                            if (u_ord == 3 && nl > 1) u_draw -= 1;
                            need_more = true;
                            break;
                        }
                        if (nl > 1 && u_ord == 4 && lane == sel) wc -= live_weight;
                        if (!avail) {  // exhausted child discovered at this pull
                            exmask |= 1u << sel;
                            nlive -= 1;
                            live_weight -= __builtin_amdgcn_readlane(wgt, (int)sel);
                            continue;
                        }
                        if (lane == sel) tk += 1;
                        if (lane == nvalid) {
                            my_leaf = sel;
                            my_idx = sel_head + sel_taken;
                        }
                        nvalid += 1;
                    }
                    if (isleaf) {
                        lt.w[lane * 16 + LeafTab::WCUR] = (uint32_t)wc;
                        lt.w[lane * 16 + LeafTab::TAKEN] = tk;
                        lt.w[lane * 16 + LeafTab::EX] = (exmask >> lane) & 1u;
                    }
                    wave_sync();
                }
                if (nvalid == 0) {
                    if (need_more) continue;
                    done = 1;  // every child exhausted
                    break;
                }
            }

            PHS(5)
            // ---- C3: trial score, acceptor, forager ----
            {
                const bool valid = lane < nvalid;
                uint32_t m0 = 0, m1 = 0, mx_ = 0;
                int my_kind = (int)lt.w[my_leaf * 16 + LeafTab::KIND];  // per-lane leaf
                if (!FAST) my_kind = my_kind == 2048 ? 1 : (my_kind == 4096 ? 2 : my_kind);  // nearby scalar leaves emit ordinary change / swap moves
                bool doable = false;
                ScoreV<L> sc;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) sc.v[kk] = cur[kk];
                if (valid) {
                    const uint32_t* rq = ring + ((size_t)my_leaf * GRC + (my_idx & (GRC - 1))) * 2;
                    m0 = rq[0];
                    m1 = rq[1];
                    mx_ = ringx[my_leaf * GRC + (my_idx & (GRC - 1))];
                    if (!FAST && my_kind <= 2) {  // (FAST: list-only models -- said at compile time, so that the scalar trial code, the interpreted joins included, is not in those kernels at all:
                                                  // with it in, the six-leaf kernel spilled 157 vector registers instead of 78 and ran 10 % slower, profiles/r06p_six_leaf_scalar_code.txt)
                        const uint32_t* tc = tables ? t_cnt : nullptr;
                        const int64_t* ts = tables ? t_sum : nullptr;
                        const ScalarDelta d = my_kind == 1 ? eval_scalar_move(sm, s_vals, 0, m0, 0u, (int32_t)m1, tc, ts, tables ? lbv : nullptr)
                                                           : eval_scalar_move(sm, s_vals, 1, m0, m1, 0, tc, ts, tables ? lbv : nullptr);
                        doable = d.doable;
                        sc = apply_scalar_delta<L>(sm, cur, d);
                        if (xown_on && doable) {
                            const int64_t dx = (int64_t)xown_scalar_delta(my_kind, m0, m1, s_vals, xown);
#pragma unroll
                            for (int kk = 0; kk < L; ++kk)
                                if (kk == gl.xown_level) sc.v[kk] = wsub(sc.v[kk], (int64_t)((uint64_t)gl.xown_weight * (uint64_t)dx));
                        }
                    } else if (PREC && my_kind == 16384) {  // critical-path leaf: scored when it was generated
                        doable = true;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) sc.v[kk] = plf_score[(size_t)(my_idx & (GRC - 1)) * 4 + kk];
                    } else if (RUIN && my_kind == 1024) {  // list ruin: scored when it was generated
                        doable = true;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) sc.v[kk] = rl.score[(size_t)m0 * 4 + kk];
                    } else if (!FAST && my_kind == 8192) {  // list permute: up to nine legs each way, priced on its own
                        const ListDelta d = eval_list_permute(lm, s_visits, s_off, m0 >> 16, m0 & 0xFFFFu, m1 >> 16, m1 & 0xFFFFu);
                        doable = d.doable;
                        sc = apply_delta<L>(lm, cur, d);
                    } else {
                        ListDelta d;
                        if (pre_eval) {  // scored by the stage above, with every lane on the same kind
                            const int32_t* rd = ringd + ((size_t)my_leaf * GRC + (my_idx & (GRC - 1))) * 2;
                            const int32_t dc = rd[0];
                            d.doable = dc != INT32_MIN;
                            d.d_cap = d.doable ? (int64_t)dc : 0;
                            d.d_dist = (int64_t)rd[1];
                        } else if (unified_eval)  // symmetric matrix: one shared gather for every kind
                            d = eval_list_unified(lm, s_visits, s_off, s_load, my_kind, m0, m1, mx_);
                        else
                            d = my_kind == 256
                                                ? eval_sublist_swap(lm, s_visits, s_off, s_load, m0 >> 16, m0 & 0xFFFFu, (m0 & 0xFFFFu) + (mx_ & 15u),
                                                                    m1 >> 16, m1 & 0xFFFFu, (m1 & 0xFFFFu) + (mx_ >> 4))
                                            : my_kind == 128
                                                ? eval_sublist_change(lm, s_visits, s_off, s_load, m0 >> 16, m0 & 0xFFFFu, (m0 & 0xFFFFu) + mx_,
                                                                      m1 >> 16, m1 & 0xFFFFu)
                                            : my_kind == 512
                                                ? eval_kopt(lm, s_visits, s_off, m0 >> 16, m0 & 0xFFFFu, m1 >> 16, m1 & 0xFFFFu, mx_)
                                            : my_kind == 64
                                                ? eval_list_reverse(lm, s_visits, s_off, m0 >> 16, m0 & 0xFFFFu, m1 & 0xFFFFu)
                                                : eval_list_move_legs<uint16_t, false>(lm, s_visits, s_off, s_load, my_kind == 4 || my_kind == 16,
                                                                                      m0 >> 16, m0 & 0xFFFFu, m1 >> 16, m1 & 0xFFFFu);
                        doable = d.doable;
                        sc = apply_delta<L>(lm, cur, d);
                        if (xown_on && doable) {  // the entities that change lists
                            const int64_t dx = (int64_t)xown_list_delta(my_kind, m0, m1, mx_, s_visits, s_off, s_vals);
#pragma unroll
                            for (int kk = 0; kk < L; ++kk)
                                if (kk == gl.xown_level) sc.v[kk] = wsub(sc.v[kk], (int64_t)((uint64_t)gl.xown_weight * (uint64_t)dx));
                        }
                    }
                }
                ScoreV<L> curv;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) curv.v[kk] = cur[kk];
                doable = doable && valid;
                if (PREC) {
                    uint64_t todo = __ballot(doable && my_kind >= 4 && my_kind != 1024 && my_kind != 16384);
                    if (plf_policy) {  // the route-graph filter evaluated the intra-list candidates when they were generated
                        bool cached = false;
                        if (doable && my_kind >= 4 && my_kind != 1024 && my_kind != 16384) {
                            const size_t ci_ = ((size_t)my_leaf * GRC + (my_idx & (GRC - 1))) * 2;
                            const int64_t cp = plf_cache[ci_];
                            if (cp != INT64_MIN) {
                                cached = true;
                                const int64_t cm2 = plf_cache[ci_ + 1];
#pragma unroll
                                for (int kk = 0; kk < L; ++kk) {
                                    if (kk == gl.prec.hard_level) sc.v[kk] -= cp - prec_pen;
                                    if (kk == gl.prec.mk_level) sc.v[kk] -= cm2 - prec_mk;
                                }
                            }
                        }
                        todo &= ~__ballot(cached);
                    }
                    if (prec_sweep && psw.ok) {  // the list change / swap candidates of the chunk: one lane each, scored together
                        const bool cand = doable && (my_kind == 4 || my_kind == 16 || my_kind == 8 || my_kind == 32);
                        const uint64_t cm_ = __ballot(cand);
                        if (cm_) {
                            int64_t tp = 0, tm_ = 0;
                            prec_trial_sweep64<uint16_t>(precm, psw, (const PREC_L uint16_t*)s_visits, (const PREC_L uint32_t*)s_off, cand, (my_kind == 4 || my_kind == 16) ? 2 : 3, m0 >> 16, m0 & 0xFFFFu,
                                                         m1 >> 16, m1 & 0xFFFFu, tp, tm_);
                            if (cand) {
#pragma unroll
                                for (int kk = 0; kk < L; ++kk) {
                                    if (kk == gl.prec.hard_level) sc.v[kk] -= tp - prec_pen;
                                    if (kk == gl.prec.mk_level) sc.v[kk] -= tm_ - prec_mk;
                                }
                            }
                            todo &= ~cm_;
                        }
                    }
                    while (pgrp_T && todo) {  // T candidates side by side, nothing applied to the lists
                        int64_t tp, tm_;
                        bool tc;
                        if (pgrp_batch(todo, my_kind, m0, m1, mx_, tp, tm_, tc)) {
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) {
                                if (kk == gl.prec.hard_level) sc.v[kk] -= tp - prec_pen;
                                if (kk == gl.prec.mk_level) sc.v[kk] -= tm_ - prec_mk;
                            }
                        }
                    }
                    while (todo) {
                        const int ci = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        const int ck = __builtin_amdgcn_readlane(my_kind, ci);
                        const uint32_t ca = (uint32_t)__builtin_amdgcn_readlane((int)m0, ci), cb = (uint32_t)__builtin_amdgcn_readlane((int)m1, ci);
                        const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)mx_, ci);
                        if (prec_incremental && (ck == 4 || ck == 16 || ck == 8 || ck == 32)) {  // list change / swap: no apply, no undo
                            PrecResult pi;
                            if (prec_trial_inc<uint16_t>(precm, pinc, s_visits, s_off, (ck == 4 || ck == 16) ? 2 : 3, ca >> 16, ca & 0xFFFFu, cb >> 16,
                                                         cb & 0xFFFFu, pi)) {
                                if ((int)lane == ci) {
#pragma unroll
                                    for (int kk = 0; kk < L; ++kk) {
                                        if (kk == gl.prec.hard_level) sc.v[kk] -= pi.penalty - prec_pen;
                                        if (kk == gl.prec.mk_level) sc.v[kk] -= pi.makespan - prec_mk;
                                    }
                                }
                                continue;
                            }
                        }
                        if (ck == 8192)  // permute: (list, start, list, end), ext = rank
                            apply_list_move_wave(lm, s_visits, s_off, s_load, 9, ca >> 16, ca & 0xFFFFu, ca >> 16, (ca & 0xFFFFu) + (cb >> 16), cb & 0xFFFFu);
                        else
                            apply_list_move_wave(lm, s_visits, s_off, s_load, list_move_kind_of(ck), ca >> 16, ca & 0xFFFFu, cb >> 16, cb & 0xFFFFu,
                                                 list_move_ext_of(ck, ca, cb, cx));
                        const PrecResult pr = prec_run_trial();
                        if ((int)lane == ci) {
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) {
                                if (kk == gl.prec.hard_level) sc.v[kk] -= pr.penalty - prec_pen;
                                if (kk == gl.prec.mk_level) sc.v[kk] -= pr.makespan - prec_mk;
                            }
                        }
                        // undo: the owners the move touched, from the committed lists in HBM (written back at every commit)
                        const uint32_t la_ = ca >> 16, lb_ = (ck == 512 || ck == 64 || ck == 8192) ? la_ : (cb >> 16);  // 3-opt / reverse / permute: one owner
                        const uint32_t l_lo = la_ < lb_ ? la_ : lb_, l_hi = la_ < lb_ ? lb_ : la_;
                        for (uint32_t t = l_lo + lane; t <= l_hi + 1; t += 64) s_off[t] = g_off[t];
                        for (uint32_t t = l_lo + lane; t <= l_hi; t += 64) s_load[t] = g_load[t];
                        wave_sync();
                        const uint32_t r_lo = uni(s_off[l_lo]), r_hi = uni(s_off[l_hi + 1]);
                        for (uint32_t t = r_lo + lane; t < r_hi; t += 64) s_visits[t] = (uint16_t)g_visits[t];
                        wave_sync();
                    }
                }
                // Move::requires_score_improvement (evaluation.rs:95-113): a multi-swap of the critical-path leaf that does not beat the last
                // step score is scored and counted, but never reaches the acceptor
                const bool consult = doable && !(PREC && my_kind == 16384 && (m0 >> 30) == 0u && score_cmp<L>(sc, curv) <= 0);
                bool acc = false;
                if (consult) {
                    if (acceptor == 0)
                        acc = score_cmp<L>(sc, curv) > 0;
                    else if (acceptor == 1)
                        acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0;
                    else if (acceptor == 4)
                        acc = score_cmp<L>(sc, curv) >= 0 || score_cmp<L>(sc, late) >= 0 || score_cmp<L>(sc, dla_thr) >= 0;
                }
                SaChunk sach;
                if (annealing) acc = sa_decide<L>(saw, p.sa, consult, sc, curv, lane, sach);
                uint64_t accmask = __ballot(acc);
                bool improving_pick = false;
                ScoreV<L> forager_thr = curv;  // FirstLastStepScoreImproving: the last step score
                if (forager == FORAGER_FIRST_BEST_IMPROVING) {  // the best score ever seen (step.rs:53-58)
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) forager_thr.v[kk] = best_sol[kk];
                }
                const uint32_t nconsumed = forager_chunk_cut<L>(forager, (uint32_t)p.limit, accepted, acc, sc, forager_thr, nvalid, improving_pick);
                const bool consumed = lane < nconsumed;
                if (annealing) sa_commit<L>(saw, p.sa, sach, nconsumed, lane);
                acc = acc && consumed;
                accmask = __ballot(acc);
                if (accmask) {
                    if (improving_pick) {  // BestCandidate::replace by the candidate that ends the step (improving.rs:92-95,205-208)
                        const int sel = (int)nconsumed - 1;
#pragma unroll
                        for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)uni64(shfl_u64((uint64_t)sc.v[kk], sel));
                        best_m0 = __shfl(m0, sel);
                        best_m1 = __shfl(m1, sel);
                        best_x = __shfl(mx_, sel);
                        best_leaf = (int)__shfl(my_leaf, sel);
                        if (TRACE) best_ti = trace_n + (uint64_t)sel;
                        equal_count = 1;
                        has_best = 1;
                    } else if (forager == 1) {
                        if (!has_best) {
                            const int sel = __ffsll((unsigned long long)accmask) - 1;
#pragma unroll
                            for (int kk = 0; kk < L; ++kk) best.v[kk] = (int64_t)shfl_u64((uint64_t)sc.v[kk], sel);
                            best_m0 = __shfl(m0, sel);
                            best_m1 = __shfl(m1, sel);
                            best_x = __shfl(mx_, sel);
                            best_leaf = (int)__shfl(my_leaf, sel);
                            if (TRACE) best_ti = trace_n + (uint64_t)sel;
                            has_best = 1;
                        }
                    } else if (!has_best || __ballot(acc && score_cmp<L>(sc, best) >= 0)) {
                        const ScoreV<L> M = wave_max_score<L>(sc, acc);
                        const int cm = has_best ? score_cmp<L>(M, best) : 1;
                        if (cm >= 0) {
                            const bool newmax = cm > 0;
                            const uint64_t eq_base = newmax ? 0 : equal_count;
                            const bool in_eq = acc && score_cmp<L>(sc, M) == 0;
                            const uint64_t eq = __ballot(in_eq);
                            const uint32_t rank = mbcnt64(eq) + 1u;
                            const uint64_t cntq = eq_base + rank;
                            const bool pick = in_eq && ((newmax && rank == 1) ||
                                                        (p.random_ties && cntq > 1 && reservoir_pick(sseed, cntq)));
                            const uint64_t pm = __ballot(pick);
                            if (pm) {
                                const int sel = 63 - __clzll((unsigned long long)pm);
                                best_m0 = __shfl(m0, sel);
                                best_m1 = __shfl(m1, sel);
                                best_x = __shfl(mx_, sel);
                                best_leaf = (int)__shfl(my_leaf, sel);
                                if (TRACE) best_ti = trace_n + (uint64_t)sel;
                            }
                            best = M;
                            equal_count = eq_base + (uint64_t)__popcll(eq);
                            has_best = 1;
                        }
                    }
                }
                const uint32_t nacc = (uint32_t)__popcll(accmask);
                accepted += nacc;
                st_gen += nconsumed;
                st_acc += nacc;
                st_scored += nvalid;
                st_calc += (uint32_t)__popcll(__ballot(consumed && doable));
                if (tracing && consumed) {
                    const uint64_t ti = trace_n + lane;
                    if ((int64_t)ti < p.trace_cap) {
                        int32_t* tm = p.trace_moves + ti * 6;
                        if (my_kind <= 2) {
                            tm[0] = my_kind == 1 ? 0 : 1;
                            tm[1] = (int32_t)m0;
                            tm[2] = 0;
                            tm[3] = my_kind == 1 ? 0 : (int32_t)m1;
                            tm[4] = 0;
                            tm[5] = my_kind == 1 ? (int32_t)m1 : -1;
                        } else if (PREC && my_kind == 16384) {
                            // decoded one candidate at a time below (the decoders are wave-uniform)
                        } else if (RUIN && my_kind == 1024) {  // a = list, a_pos = count, six 16-bit positions in b / b_pos / value
                            const uint16_t* cd = rl.cand + (size_t)m0 * RuinLds::CAND_WORDS;
                            tm[0] = 8;
                            tm[1] = (int32_t)cd[0];
                            tm[2] = (int32_t)cd[1];
                            tm[3] = (int32_t)((uint32_t)cd[2] | ((uint32_t)cd[3] << 16));
                            tm[4] = (int32_t)((uint32_t)cd[4] | ((uint32_t)cd[5] << 16));
                            tm[5] = (int32_t)((uint32_t)cd[6] | ((uint32_t)cd[7] << 16));
                            if (PREC && plf_policy && cd[1] <= 5) tm[5] |= (int32_t)0x80000000u;  // the move carries the slot's precedence hooks
                        } else if (my_kind == 8192) {  // (9, list, start, list, end, rank)
                            tm[0] = 9;
                            tm[1] = (int32_t)(m0 >> 16);
                            tm[2] = (int32_t)(m0 & 0xFFFFu);
                            tm[3] = (int32_t)(m0 >> 16);
                            tm[4] = (int32_t)((m0 & 0xFFFFu) + (m1 >> 16));
                            tm[5] = (int32_t)(m1 & 0xFFFFu);
                        } else {
                            tm[0] = list_move_kind_of(my_kind);
                            tm[1] = (int32_t)(m0 >> 16);
                            tm[2] = (int32_t)(m0 & 0xFFFFu);
                            tm[3] = (int32_t)(m1 >> 16);
                            tm[4] = (int32_t)(m1 & 0xFFFFu);
                            tm[5] = my_kind == 128 ? (int32_t)((m0 & 0xFFFFu) + mx_)
                                                   : (my_kind == 256 ? (int32_t)((mx_ & 15u) | ((mx_ >> 4) << 16)) : (my_kind == 512 ? (int32_t)mx_ : -1));
                        }
                        for (int kk = 0; kk < L && kk < gl.levels; ++kk) p.trace_scores[ti * gl.levels + kk] = doable ? sc.v[kk] : 0;
                        p.trace_flags[ti] = (doable ? 1 : 0) | (acc ? 2 : 0) | ((doable && !consult) ? 16 : 0) | ((int32_t)my_leaf << 8);  // 16: RejectedByScoreImprovement
                    }
                }
                if (PREC && tracing) {  // critical-path leaf: the wire form of its consumed candidates
                    uint64_t pend = __ballot(consumed && my_kind == 16384);
                    while (pend) {
                        const int ci = __ffsll((unsigned long long)pend) - 1;
                        pend &= pend - 1;
                        PlfMove pm_;
                        plf_decode((uint32_t)__builtin_amdgcn_readlane((int)m0, ci), (uint32_t)__builtin_amdgcn_readlane((int)m1, ci), pm_);
                        const uint64_t ti = trace_n + (uint64_t)ci;
                        if (lane == 0 && (int64_t)ti < p.trace_cap) plf_wire(pm_, p.trace_moves + ti * 6);
                    }
                }
                if (tracing) trace_n += nconsumed;
                // every laid-out pull is consumed unless the forager cut the step (which ends it)
                for (int l = 0; l < nl; ++l) lt.set(l, LeafTab::HEAD, lt.get(l, LeafTab::HEAD) + lt.get(l, LeafTab::TAKEN));
                if (forager_quits(forager, (uint32_t)p.limit, accepted, has_best, improving_pick)) done = 1;
            }
            PHS(6)
        }

        // ---- commit the forager's pick ----
        auto write_best_snapshot = [&]() {
            if (has_list) {
                const uint32_t tot = uni(s_off[V]);
                for (uint32_t t = lane; t < tot; t += 64) lm.best_visits[(size_t)r * lm.n_cap + t] = s_visits[t];
                for (uint32_t t = lane; t <= (uint32_t)V; t += 64) lm.best_off[(size_t)r * (V + 1) + t] = s_off[t];
            }
            for (uint32_t t = lane; t < ns; t += 64) sm.best_vals[(size_t)r * ns + t] = (int32_t)s_vals[t];
        };
        const bool applied = has_best && !dry_run;
        if (applied) {
            if (best_pending) {
                ScoreV<L> bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) bs.v[kk] = best_sol[kk];
                if (!(score_cmp<L>(best, bs) > 0)) {  // leaving the best state: write its snapshot first
                    write_best_snapshot();
                    best_pending = false;
                }
            }
            int kind = lt.geti(uni((uint32_t)best_leaf), LeafTab::KIND);
            if (!FAST) kind = kind == 2048 ? 1 : (kind == 4096 ? 2 : kind);
            const uint32_t a = uni(best_m0), b = uni(best_m1);
            if (!FAST && kind <= 2) {
                if (tracing && lane == 0) {
                    p.trace_applied[0] = 1;
                    if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                    p.trace_applied[1] = kind == 1 ? 0 : 1;
                    p.trace_applied[2] = (int32_t)a;
                    p.trace_applied[3] = 0;
                    p.trace_applied[4] = kind == 1 ? 0 : (int32_t)b;
                    p.trace_applied[5] = 0;
                    p.trace_applied[6] = kind == 1 ? (int32_t)b : -1;
                }
                if (lane == 0) {
                    if (tables) scalar_tables_apply(sm, s_vals, kind == 1 ? 0 : 1, a, b, (int32_t)b, t_cnt, t_sum);
                    if (kind == 1)
                        s_vals[a] = (VT)(int32_t)b;
                    else {
                        const VT t = s_vals[a];
                        s_vals[a] = s_vals[b];
                        s_vals[b] = t;
                    }
                }
                wave_sync();
            } else if (PREC && kind == 16384) {  // critical-path leaf: decode the pick again (the analysis tables are the step's), apply it
                PlfMove pm_;
                plf_decode(a, b, pm_);
                if (tracing && lane == 0) {
                    p.trace_applied[0] = 1;
                    if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                    plf_wire(pm_, p.trace_applied + 1);
                }
                if (pm_.kind == 8) {
                    const ScoreV<L> ignored = plf_ruin(pm_, true, true, false);
                    (void)ignored;
                } else
                    plf_apply(pm_);
            } else if (RUIN && kind == 1024) {  // committed ruin: the same recreate, this time kept
                const uint16_t* cd = rl.cand + (size_t)a * RuinLds::CAND_WORDS;
                if (tracing && lane == 0) {
                    p.trace_applied[0] = 1;
                    if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                    p.trace_applied[1] = 8;
                    p.trace_applied[2] = (int32_t)cd[0];
                    p.trace_applied[3] = (int32_t)cd[1];
                    p.trace_applied[4] = (int32_t)((uint32_t)cd[2] | ((uint32_t)cd[3] << 16));
                    p.trace_applied[5] = (int32_t)((uint32_t)cd[4] | ((uint32_t)cd[5] << 16));
                    p.trace_applied[6] = (int32_t)((uint32_t)cd[6] | ((uint32_t)cd[7] << 16));
                    if (PREC && plf_policy && cd[1] <= 5) p.trace_applied[6] |= (int32_t)0x80000000u;
                }
                if (PREC) {
                    PlfMove pm_;
                    plf_from_ruin_cand(cd, pm_);
                    const ScoreV<L> ignored = plf_ruin(pm_, true, plf_policy, gl.ruin.skip_empty != 0);
                    (void)ignored;
                } else {
                    int64_t base_score[L];
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) base_score[kk] = cur[kk];
                    if constexpr (FAST) {  // the list-preserving recreate once more, this time kept (sf_ruin_v2.h)
                        rv2_build_words(s_off, (uint32_t)V, ruin_sbase);
                        (void)ruin_trial_v2<L>(lm, s_visits, s_off, s_load, cd, rl.work, ruin_sbase, rfast, rfast.slot, (uint32_t)ruin_arena_cap(lm.n_cap, V), gl.ruin.skip_empty,
                                               base_score, rl.score + (size_t)a * 4, true, g_visits);
                    } else {
                        ruin_recreate<L>(lm, s_visits, s_off, s_load, cd, rl.work, ruin_sbase, rfast, gl.ruin.skip_empty, true, base_score,
                                         rl.score + (size_t)a * 4);
                    }
                }
                wave_sync();
                if (has_nearby) {  // any list may have changed: rebuild node -> (route, position)
                    for (uint32_t v = lane; v < (uint32_t)V; v += 64) {
                        const uint32_t o = s_off[v], len = s_off[v + 1] - o;
                        for (uint32_t q = 0; q < len; ++q) node_slot[s_visits[o + q]] = (v << 16) | q;
                    }
                    wave_sync();
                    if (NODEG) ring_sync();
                }
            } else {
                if (tracing && lane == 0) {
                    p.trace_applied[0] = 1;
                    if ((int64_t)best_ti < p.trace_cap) p.trace_flags[best_ti] |= 4;  // Selected + Applied
                    p.trace_applied[1] = list_move_kind_of(kind);
                    p.trace_applied[2] = (int32_t)(a >> 16);
                    p.trace_applied[3] = (int32_t)(a & 0xFFFFu);
                    p.trace_applied[4] = kind == 8192 ? (int32_t)(a >> 16) : (int32_t)(b >> 16);
                    p.trace_applied[5] = kind == 8192 ? (int32_t)((a & 0xFFFFu) + (b >> 16)) : (int32_t)(b & 0xFFFFu);
                    p.trace_applied[6] = kind == 8192 ? (int32_t)(b & 0xFFFFu) : kind == 128 ? (int32_t)((a & 0xFFFFu) + uni(best_x))
                                                      : (kind == 256 ? (int32_t)((uni(best_x) & 15u) | ((uni(best_x) >> 4) << 16))
                                                                     : (kind == 512 ? (int32_t)uni(best_x) : -1));
                }
                if (!FAST && kind == 8192)
                    apply_list_move_wave(lm, s_visits, s_off, s_load, 9, a >> 16, a & 0xFFFFu, a >> 16, (a & 0xFFFFu) + (b >> 16), b & 0xFFFFu);
                else
                    apply_list_move_wave(lm, s_visits, s_off, s_load, list_move_kind_of(kind), a >> 16, a & 0xFFFFu, b >> 16, b & 0xFFFFu,
                                         list_move_ext_of(kind, a, b, uni(best_x)));
                if (has_nearby) {  // refresh node -> (route, position) for the touched routes
                    const uint32_t ra_ = a >> 16, rb_ = (kind == 512 || kind == 8192) ? (a >> 16) : (b >> 16);  // 3-opt / permute: b packs cuts / (size, rank), not a route
                    const uint32_t oa = s_off[ra_], la = s_off[ra_ + 1] - oa;
                    const uint32_t ob = s_off[rb_], lb = s_off[rb_ + 1] - ob;
                    for (uint32_t t = lane; t < la + (ra_ != rb_ ? lb : 0u); t += 64) {
                        if (t < la)
                            node_slot[s_visits[oa + t]] = (ra_ << 16) | t;
                        else
                            node_slot[s_visits[ob + (t - la)]] = (rb_ << 16) | (t - la);
                    }
                    wave_sync();
                    if (NODEG) ring_sync();
                }
                if (xown_on && kind != 512 && kind != 8192 && kind != 64) {  // the two touched lists name their elements again
                    const uint32_t ra_ = a >> 16, rb_ = b >> 16;
                    const uint32_t oa = s_off[ra_], la = s_off[ra_ + 1] - oa;
                    const uint32_t ob = s_off[rb_], lb = s_off[rb_ + 1] - ob;
                    for (uint32_t t = lane; t < la + (ra_ != rb_ ? lb : 0u); t += 64) {
                        const uint32_t e_ = t < la ? (uint32_t)s_visits[oa + t] : (uint32_t)s_visits[ob + (t - la)];
                        if (e_ < ns) xown[e_] = (uint16_t)(t < la ? ra_ : rb_);
                    }
                    ring_sync();
                }
            }
            if (PREC && kind > 2) {  // a list move was committed: the HBM copy the trials undo from, and the constraint's committed state
                const uint32_t tot = uni(s_off[V]);
                for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = s_visits[t];
                for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = s_off[t];
                for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = s_load[t];
                prec_sync();
                const PrecResult pr = prec_run();
                prec_pen = pr.penalty;
                prec_mk = pr.makespan;
            }
#pragma unroll
            for (int kk = 0; kk < L; ++kk) cur[kk] = best.v[kk];
            st_applied += 1;
        } else if (tracing && lane == 0) {
            p.trace_applied[0] = 0;
        }
        if (!dry_run) {
            bool improved = false;
            if (applied) {
                ScoreV<L> cs, bs;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    cs.v[kk] = cur[kk];
                    bs.v[kk] = best_sol[kk];
                }
                improved = score_cmp<L>(cs, bs) > 0;
            }
            if (improved) {  // update_best_solution (scope_progress.rs:89-107): the clone is deferred (best_pending)
                best_pending = true;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) best_sol[kk] = cur[kk];
            }
            if ((acceptor == 1 || acceptor == 4) && lane == 0) {
#pragma unroll
                for (int kk = 0; kk < L; ++kk) p.la_hist[((size_t)r * p.la_size + la_slot) * 4 + kk] = cur[kk];
            }
            if (acceptor == 4 && lane == 0) {  // step_ended: the phase's best step score (diversified_late_acceptance.rs:161-170)
                ScoreV<L> cs, db;
#pragma unroll
                for (int kk = 0; kk < L; ++kk) {
                    cs.v[kk] = cur[kk];
                    db.v[kk] = p.dla_best[(size_t)r * 4 + kk];
                }
                if (score_cmp<L>(cs, db) > 0) {
#pragma unroll
                    for (int kk = 0; kk < L; ++kk) p.dla_best[(size_t)r * 4 + kk] = cur[kk];
                }
            }
            if (annealing) sa_step_ended(saw, p.sa, lane);
            wave_sync();
            st_steps += 1;
            prev_pulls = step_pulls;
            la_cursor = la_cursor + 1 >= p.la_size ? 0 : la_cursor + 1;
        }
        PHS(7)
        if (!dry_run && p.move_budget > 0 && (int64_t)st_gen >= p.move_budget) break;  // work-balanced launch: see sf_solve_moves
        if (!dry_run && p.move_budget == 0 && st_scored >= 0x70000000u) flush_stats();
    }
    PH_DUMP

    if (!dry_run) {
        if (annealing) sa_store(saw, p.sa, r, lane);
        if (RUIN && lane < 4) gl.ruin.rng[(size_t)r * 4 + lane] = rl.prng[lane];
        if (best_pending) {  // the launch ends in a best state: its deferred snapshot
            if (has_list) {
                const uint32_t tot = uni(s_off[V]);
                for (uint32_t t = lane; t < tot; t += 64) lm.best_visits[(size_t)r * lm.n_cap + t] = s_visits[t];
                for (uint32_t t = lane; t <= (uint32_t)V; t += 64) lm.best_off[(size_t)r * (V + 1) + t] = s_off[t];
            }
            for (uint32_t t = lane; t < ns; t += 64) sm.best_vals[(size_t)r * ns + t] = (int32_t)s_vals[t];
        }
        if (has_list) {
            const uint32_t tot = uni(s_off[V]);
            for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = s_visits[t];
            for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = s_off[t];
            for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = s_load[t];
        }
        for (uint32_t t = lane; t < ns; t += 64) g_vals[t] = (int32_t)s_vals[t];
        if (PREC && lane == 0) {
            gl.prec.state[(size_t)r * 2] = prec_pen;
            gl.prec.state[(size_t)r * 2 + 1] = prec_mk;
        }
        if (lane == 0) {
#pragma unroll
            for (int kk = 0; kk < L; ++kk) {
                g_score[kk] = cur[kk];
                p.last_step_score[(size_t)r * 4 + kk] = cur[kk];
                g_best_score[kk] = best_sol[kk];
            }
            p.la_idx[r] = la_cursor;
            p.step_index[r] = step_index0 + steps_run + (uint64_t)st_steps;  // steps actually run (a move budget can end the launch early)
            p.seed_draws[r] = seed_draws0 + steps_run + (uint64_t)st_steps;
        }
        flush_stats();
    }
    if (tracing && lane == 0) *p.trace_count = (int64_t)trace_n;
}

}  // namespace sf
