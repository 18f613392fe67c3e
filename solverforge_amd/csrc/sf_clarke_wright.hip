// Clarke-Wright savings construction on the device (VERDICT round 1 missing #5 / next #9).
//
// Reference semantics restated (paths under crates/solverforge-solver/src/manager/phase_factory/):
//   list_clarke_wright/kernel.rs:59-472      available (empty) owner slots, depot filter, singleton routes, one savings entry per
//                                            unordered pair and metric class, merge passes over the sorted entries until a pass
//                                            merges nothing, owner matching, completion by insertion, commit through replace_route
//   list_clarke_wright/savings.rs:9-18       entry order: saving descending, metric class, left index, right index
//   list_clarke_wright/route_state.rs        routes_match_owners_after_merge; owner_assignment.rs: match_route_owners
//   list_clarke_wright/completion.rs:19-224  completion: elements in (route, visit position) order, each to the (owner, position) of
//                                            least (insertion delta, route length, owner slot, position)
// with the hook bundle of the stock CVRP domain (crates/solverforge-cvrp/src/helpers.rs:40-87, 108-179): ONE metric class (every
// vehicle shares the ProblemData), the model's depot, distance_cost legs (<= i64::MAX / 4, so the clamped i128 sums of
// distance_arithmetic.rs are exact in i64), feasibility = structural only (savings_hooks::feasible, mode 0) or the capacity test of
// route_hooks::feasible (mode 1).  Under one class with uniform owners the reference's owner bookkeeping collapses to closed
// forms, each restated at its use below; the oracle (oracle/sfo_clarke_wright.hpp) keeps the general hook form and the parity
// tests compare the two.
//
// Shape on the device:
//   k_cw_savings   one thread per pair (a < b) in row-major order: key = saving, value = a << 16 | b          (HBM bound)
//   rocprim        stable descending radix sort of the pairs by saving: row-major input order IS (left, right) ascending, so a
//                  stable sort reproduces the reference's total order
//   k_cw_merge     one wavefront per replica, route state in LDS (label, two undirected neighbours, per-label ends / size / load /
//                  reference route index): 64 sorted entries per iteration are tested in parallel, the survivors merge one at a
//                  time (lowest lane first, later lanes re-test); the smaller side is relabelled.  Rejections are permanent when no
//                  demand is negative, so the reference's confirming pass (which merges nothing) is skipped then.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/rocprim.hpp>

#include "sf_construct.hip"

namespace sf {

constexpr uint32_t CW_NONE = 0xFFFFu;

__host__ __device__ inline uint64_t cw_row_base(uint64_t a, uint64_t n) { return a * n - a * (a + 1) / 2; }  // pairs before row a

struct CwCarve {
    size_t load, rload, off, slots, visits, route_of, nbr, first, last, refi, size, reps, present, total;
    __host__ __device__ CwCarve(int V, int n_cap, int dim, int n) {
        size_t o = 0;
        load = o;  // per owner
        o = align_up(o + sizeof(int64_t) * V, 16);
        rload = o;  // per route label
        o = align_up(o + sizeof(int64_t) * n, 16);
        off = o;
        o = align_up(o + sizeof(uint32_t) * (V + 1), 16);
        slots = o;  // available owners in owner order; then owner -> route label
        o = align_up(o + sizeof(uint16_t) * 2 * V, 16);
        visits = o;
        o = align_up(o + sizeof(uint16_t) * n_cap, 16);
        route_of = o;  // later: element positions in (route, visit) order
        o = align_up(o + sizeof(uint16_t) * n, 16);
        nbr = o;
        o = align_up(o + sizeof(uint16_t) * 2 * n, 16);
        first = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        last = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        refi = o;  // later: base of the route in the order array
        o = align_up(o + sizeof(uint16_t) * n, 16);
        size = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        reps = o;
        o = align_up(o + sizeof(uint16_t) * n, 16);
        present = o;
        o = align_up(o + sizeof(uint32_t) * (((size_t)dim + 31) / 32), 16);
        total = o;
    }
};

__device__ __forceinline__ int64_t cw_dist_cost(const ListModel& lm, uint32_t from, uint32_t to) {  // problem_data.rs:28-31
    const int64_t v = lm.mat ? lm.mat[(size_t)from * (size_t)lm.dim + to] : 0;
    return (v >= 0 && v != UNREACHABLE) ? v : MAX_SAFE_LEG_COST;
}

// kernel.rs:141-208: saving = d(depot, left) + d(depot, right) - d(left, right), exact (every leg in 0..=i64::MAX / 4)
__global__ __launch_bounds__(256) void k_cw_savings(ListModel lm, const uint32_t* __restrict__ elements, int n, int64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
    const uint32_t a = blockIdx.y;
    const uint32_t b = a + 1u + blockIdx.x * 256u + threadIdx.x;
    if (b >= (uint32_t)n) return;
    const uint32_t ea = elements[a], eb = elements[b];
    const int64_t s = cw_dist_cost(lm, (uint32_t)lm.depot, ea) + cw_dist_cost(lm, (uint32_t)lm.depot, eb) - cw_dist_cost(lm, ea, eb);
    const uint64_t at = cw_row_base(a, (uint64_t)n) + (b - a - 1u);
    keys[at] = s;
    vals[at] = (a << 16) | b;
}

__device__ __forceinline__ int64_t cw_wave_min_i64(int64_t v) {
#pragma unroll
    for (int mlane = 32; mlane >= 1; mlane >>= 1) {
        const int64_t o = (int64_t)shfl_xor_u64((uint64_t)v, mlane);
        v = o < v ? o : v;
    }
    return v;
}
// exclusive prefix of a per-lane count within the wave + the wave total
__device__ __forceinline__ uint32_t cw_wave_excl(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    total = (uint32_t)__shfl((int)inc, 63, 64);
    return inc - v;
}

// walks a route from `start` (one of its ends) along the undirected neighbour pairs; f(position in the walk, element position)
template <class F>
__device__ __forceinline__ void cw_walk(const lds_u16* nbr, uint32_t start, F f) {
    uint32_t cur = start, prev = CW_NONE, at = 0;
    while (cur != CW_NONE) {
        f(at, cur);
        const uint32_t n0 = nbr[2 * cur], n1 = nbr[2 * cur + 1];
        const uint32_t nxt = n0 != prev ? n0 : n1;
        prev = cur;
        cur = nxt;
        ++at;
    }
}

// feasible_mode: 0 structural (always, ids validated on the host), 1 capacity.  monotone: no negative demand among the elements.
// out_flag[r]: 1 = routes committed, 0 = lists untouched (no available owner / nothing to place / routes could not be matched or
// completed: kernel.rs:88-105, 401-415).  cw_stats[r][4]: merges, merge passes walked, completion trials, routes built.
__global__ __launch_bounds__(64) void k_cw_merge(ListModel lm, const uint32_t* __restrict__ elements, int n, const uint32_t* __restrict__ pairs,
                                                 uint64_t n_pairs, int feasible_mode, int monotone, int32_t* __restrict__ out_flag,
                                                 uint64_t* __restrict__ cw_stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int r = blockIdx.x;
    const int V = lm.V;
    const CwCarve cv(V, lm.n_cap, lm.dim, n);
    lds_i64* load = (lds_i64*)(smem + cv.load);
    lds_i64* rload = (lds_i64*)(smem + cv.rload);
    lds_u32* off = (lds_u32*)(smem + cv.off);
    lds_u16* slots = (lds_u16*)(smem + cv.slots);
    lds_u16* owner_route = slots + V;
    lds_u16* visits = (lds_u16*)(smem + cv.visits);
    lds_u16* route_of = (lds_u16*)(smem + cv.route_of);
    lds_u16* order = route_of;
    lds_u16* nbr = (lds_u16*)(smem + cv.nbr);
    lds_u16* first = (lds_u16*)(smem + cv.first);
    lds_u16* last = (lds_u16*)(smem + cv.last);
    lds_u16* refi = (lds_u16*)(smem + cv.refi);
    lds_u16* rbase = refi;
    lds_u16* size = (lds_u16*)(smem + cv.size);
    lds_u16* reps = (lds_u16*)(smem + cv.reps);
    lds_u32* present = (lds_u32*)(smem + cv.present);
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    int64_t* g_load = lm.load + (size_t)r * V;
    if (lane == 0 && out_flag) out_flag[r] = 0;
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
    // available_entity_slots (kernel.rs:80-82): the owners whose list is empty, in owner order
    uint32_t m = 0;
    for (uint32_t e0 = 0; e0 < (uint32_t)V; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool av = e < (uint32_t)V && off[e + 1] == off[e];
        const uint64_t mask = __ballot(av);
        if (av) slots[m + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)e;
        m += (uint32_t)__popcll(mask);
    }
    wave_sync();
    // singleton routes (kernel.rs:117-133); an element already in a list of this replica has no route
    uint32_t n_un = 0, bad = 0;
    for (uint32_t k0 = 0; k0 < (uint32_t)n; k0 += 64) {
        const uint32_t k = k0 + lane;
        bool un = false, bd = false;
        if (k < (uint32_t)n) {
            const uint32_t x = elements[k];
            un = !((present[x >> 5] >> (x & 31u)) & 1u);
            const int64_t dm = lm.demand ? (int64_t)lm.demand[x] : 0;
            route_of[k] = un ? (uint16_t)k : (uint16_t)CW_NONE;
            first[k] = last[k] = refi[k] = (uint16_t)k;
            size[k] = 1;
            nbr[2 * k] = nbr[2 * k + 1] = (uint16_t)CW_NONE;
            rload[k] = dm;
            bd = un && feasible_mode == 1 && dm > lm.capacity;  // a singleton no owner can take: feasible_for_all_owners = false
        }
        n_un += (uint32_t)__popcll(__ballot(un));
        bad += (uint32_t)__popcll(__ballot(bd));
    }
    wave_sync();
    if (m == 0 || n_un == 0) return;  // kernel.rs:88-105

    // ---- merge passes (kernel.rs:212-357).  One class, uniform owners: can_merge_for_metric_class always holds; the merged
    // route is feasible for all owners or none; routes_match_owners_after_merge holds unless some route OTHER than the two being
    // merged has no feasible owner (route_state.rs:84-103: the candidate stands in for the merged route, the removed one is
    // skipped).  Only singletons can be such routes (demand > capacity), `bad` counts the ones still alone: with negative demands
    // a merge can absorb one.
    const uint32_t bad0 = bad;
    uint64_t merges = 0, passes = 0;
    for (;;) {
        bool merged_in_pass = false;
        ++passes;
        for (uint64_t base = 0; base < n_pairs; base += 64) {
            const uint64_t idx = base + lane;
            const bool valid = idx < n_pairs;
            const uint32_t pk = valid ? pairs[idx] : 0u;
            const uint32_t a = pk >> 16, b = pk & 0xFFFFu;
            uint32_t from = 0;
            for (;;) {
                bool ok = false;
                if (valid && lane >= from) {
                    const uint32_t ra = route_of[a], rb = route_of[b];
                    if (ra != CW_NONE && rb != CW_NONE && ra != rb) {
                        ok = (first[ra] == a || last[ra] == a) && (first[rb] == b || last[rb] == b);
                        if (ok && feasible_mode == 1) {
                            const int64_t la_ = rload[ra], lb_ = rload[rb];
                            ok = la_ + lb_ <= lm.capacity && bad == (uint32_t)(la_ > lm.capacity) + (uint32_t)(lb_ > lm.capacity);
                        }
                    }
                }
                const uint64_t mask = __ballot(ok);
                if (mask == 0ull) break;
                const uint32_t L = (uint32_t)__builtin_ctzll(mask);
                const uint32_t la = (uint32_t)__shfl((int)a, (int)L, 64), lb = (uint32_t)__shfl((int)b, (int)L, 64);
                uint32_t absorbed = 0;
                if (feasible_mode == 1)
                    absorbed = (uint32_t)(rload[route_of[la]] > lm.capacity) + (uint32_t)(rload[route_of[lb]] > lm.capacity);
                wave_sync();
                if (lane == 0) {
                    const uint32_t ri = route_of[la], rj = route_of[lb];
                    // test_ri reversed iff it starts with left; test_rj reversed iff it ends with right (kernel.rs:286-293)
                    const uint32_t nf = first[ri] == la ? last[ri] : first[ri];
                    const uint32_t nl = last[rj] == lb ? first[rj] : last[rj];
                    const bool keep_i = size[ri] >= size[rj];
                    const uint32_t keep = keep_i ? ri : rj, drop = keep_i ? rj : ri;
                    cw_walk(nbr, first[drop], [&](uint32_t, uint32_t cur) { route_of[cur] = (uint16_t)keep; });
                    nbr[2 * la + (nbr[2 * la] == CW_NONE ? 0 : 1)] = (uint16_t)lb;
                    nbr[2 * lb + (nbr[2 * lb] == CW_NONE ? 0 : 1)] = (uint16_t)la;
                    const uint32_t ref = refi[ri], sz = (uint32_t)size[ri] + (uint32_t)size[rj];
                    const int64_t ld = rload[ri] + rload[rj];
                    first[keep] = (uint16_t)nf;
                    last[keep] = (uint16_t)nl;
                    refi[keep] = (uint16_t)ref;  // the merged route keeps the left route's index (routes[ri], kernel.rs:343-355)
                    size[keep] = (uint16_t)sz;
                    size[drop] = 0;
                    rload[keep] = ld;
                }
                wave_sync();
                bad -= uni(absorbed);
                merged_in_pass = true;
                ++merges;
                from = L + 1u;
            }
        }
        if (!merged_in_pass || monotone) break;
    }

    // ---- the constructed routes in route-index order (kernel.rs:359-391): route i of the reference = the label whose refi is i
    uint32_t K = 0, inel = 0;
    for (uint32_t k0 = 0; k0 < (uint32_t)n; k0 += 64) {
        const uint32_t k = k0 + lane;
        bool rep = false, ok = false;
        uint32_t lbl = CW_NONE;
        if (k < (uint32_t)n) {
            lbl = route_of[k];
            rep = lbl != CW_NONE && refi[lbl] == k;
            // feasible owners: all of them or none (a merged route passed the test when it was built; demands only matter in mode 1)
            ok = rep && !(feasible_mode == 1 && rload[lbl] > lm.capacity);
        }
        const uint64_t mk = __ballot(ok);
        if (ok) reps[K + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)lbl;
        K += (uint32_t)__popcll(mk);
        inel += (uint32_t)__popcll(__ballot(rep && !ok));
    }
    wave_sync();
    // match_route_owners (owner_assignment.rs:79-113) with equal feasible sets: the augmenting search hands route i the slot
    // min(K, m) - 1 - i; routes beyond the slots stay unmatched
    if (K == 0) return;
    const bool complete = K > m;
    if (complete && inel != 0) return;  // kernel.rs:395-415: no completion with owner-ineligible routes -> discarded
    // bases of the routes in the order array, then the elements of every route first -> last
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < K; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t lbl = i < K ? (uint32_t)reps[i] : 0u;
        const uint32_t sz = i < K ? (uint32_t)size[lbl] : 0u;
        uint32_t tot;
        const uint32_t ex = cw_wave_excl(sz, lane, tot);
        if (i < K) rbase[lbl] = (uint16_t)(run + ex);
        run += tot;
    }
    wave_sync();
    for (uint32_t i = lane; i < K; i += 64) {
        const uint32_t lbl = reps[i], bs = rbase[lbl];
        cw_walk(nbr, first[lbl], [&](uint32_t at, uint32_t cur) { order[bs + at] = (uint16_t)cur; });
    }
    wave_sync();
    const uint32_t placed_total = run;
    if (tot0 + placed_total > (uint32_t)lm.n_cap) return;  // element capacity of the flat lists

    uint64_t trials = 0;
    if (!complete) {
        // commit (kernel.rs:432-447): route i replaces the (empty) list of slot K - 1 - i
        for (uint32_t e = lane; e < (uint32_t)V; e += 64) owner_route[e] = (uint16_t)CW_NONE;
        wave_sync();
        for (uint32_t i = lane; i < K; i += 64) owner_route[slots[K - 1u - i]] = reps[i];
        wave_sync();
        // new offsets: an owner with a route gets its size (its old list is empty), the others keep their list
        uint32_t acc = 0;
        for (uint32_t e0 = 0; e0 < (uint32_t)V; e0 += 64) {
            const uint32_t e = e0 + lane;
            uint32_t len = 0;
            if (e < (uint32_t)V) {
                const uint32_t lbl = owner_route[e];
                len = lbl != CW_NONE ? (uint32_t)size[lbl] : g_off[e + 1] - g_off[e];
            }
            uint32_t tot;
            const uint32_t ex = cw_wave_excl(len, lane, tot);
            if (e < (uint32_t)V) off[e] = acc + ex;
            acc += tot;
        }
        if (lane == 0) off[V] = acc;
        wave_sync();
        for (uint32_t e = lane; e < (uint32_t)V; e += 64) {
            const uint32_t lbl = owner_route[e], o = off[e];
            if (lbl != CW_NONE) {
                const uint32_t bs = rbase[lbl], sz = size[lbl];
                for (uint32_t t = 0; t < sz; ++t) visits[o + t] = (uint16_t)elements[order[bs + t]];
                load[e] = rload[lbl];
            } else {
                const uint32_t go = g_off[e], len = g_off[e + 1] - go;
                for (uint32_t t = 0; t < len; ++t) visits[o + t] = (uint16_t)g_visits[go + t];
            }
        }
        wave_sync();
    } else {
        // ---- completion by savings insertion (completion.rs:19-224): more routes than owners.  Uniform owners: every element
        // has the same feasible-owner count, so the element order is (route, visit position) = the order array.
        if (bad0 != 0) return;  // an element no owner can take as a singleton: completion returns None (completion.rs:58-60)
        RuinModel rm = ruin_model(lm);
        for (uint32_t q = 0; q < placed_total; ++q) {
            const uint32_t x = elements[uni(order[q])];
            const int64_t dx = lm.demand ? (int64_t)lm.demand[x] : 0;
            int64_t bd = INT64_MAX;
            uint64_t bkey = ~0ull;
            for (uint32_t j0 = 0; j0 < m; j0 += 64) {
                const uint32_t j = j0 + lane;
                if (j >= m) continue;
                const uint32_t e = slots[j], o = off[e], len = off[e + 1] - o;
                trials += len + 1u;
                if (feasible_mode == 1 && load[e] + dx > lm.capacity) continue;
                uint32_t prev = (uint32_t)lm.depot;
                for (uint32_t p = 0; p <= len; ++p) {
                    const uint32_t next = p < len ? (uint32_t)visits[o + p] : (uint32_t)lm.depot;
                    const int64_t delta = cw_dist_cost(lm, prev, x) + cw_dist_cost(lm, x, next) - cw_dist_cost(lm, prev, next);
                    const uint64_t key = ((uint64_t)len << 32) | ((uint64_t)j << 16) | p;
                    if (delta < bd || (delta == bd && key < bkey)) bd = delta, bkey = key;
                    prev = next;
                }
            }
            const int64_t md = cw_wave_min_i64(bkey != ~0ull ? bd : INT64_MAX);
            const uint64_t kmin = uni64(ruin_wave_min_u64((bkey != ~0ull && bd == md) ? bkey : ~0ull));
            if (kmin == ~0ull) return;  // no feasible insertion: completion returns None, nothing is committed
            construct_list_insert(rm, visits, off, load, (uint32_t)slots[(uint32_t)(kmin >> 16) & 0xFFFFu], (uint32_t)kmin & 0xFFFFu, x);
        }
        wave_sync();
    }
    const uint32_t tot = uni(off[V]);
    for (uint32_t t = lane; t < tot; t += 64) g_visits[t] = visits[t];
    for (uint32_t t = lane; t <= (uint32_t)V; t += 64) g_off[t] = off[t];
    for (uint32_t t = lane; t < (uint32_t)V; t += 64) g_load[t] = load[t];
    if (lane == 0) {
        if (out_flag) out_flag[r] = 1;
        if (cw_stats) {
            uint64_t* gs = cw_stats + (size_t)r * 4;
            gs[0] = merges, gs[1] = passes, gs[2] = 0, gs[3] = complete ? m : K;
        }
    }
    if (cw_stats) {  // completion trials: per-lane partial counts
        uint64_t t = trials;
#pragma unroll
        for (int mlane = 32; mlane >= 1; mlane >>= 1) t += shfl_xor_u64(t, mlane);
        if (lane == 0) cw_stats[(size_t)r * 4 + 2] = t;
    }
}

// ---- route-local 2-opt polishing (manager/phase_factory/list_k_opt/kernel.rs:57-220), the step the default construction runs
// after Clarke-Wright.  The reference sweeps (i, j), i < j, per route and reverses route[i..=j] in place at the FIRST improving j,
// then continues with j + 1 on the modified route; a = the element before i (or the depot) and b = route[i] are read once per i.
// A reversal of [i..=j0] only touches positions <= j0 and a later j of the same row reads positions j, j + 1 > j0, so every
// predicate of row i can be evaluated from the route as it stands when the row starts: one wavefront per (route, replica), 64
// values of j per round, the improving lanes applied in ascending order as wave-parallel segment reversals.  Feasibility (mode 1:
// the capacity test of route_hooks::feasible) does not depend on the order of a route: it is one flag per route, an infeasible
// route takes no reversal.  Counters: one generated + evaluated candidate per (i, j), one accepted move per reversal, applied =
// the accepted reversals of a changed route, one step + score calculation per changed route.
__global__ __launch_bounds__(64) void k_list_construct_two_opt(ListModel lm, int feasible_mode, int max_sweeps, uint64_t* stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u16* route = (lds_u16*)smem;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t e = blockIdx.x;
    const int r = blockIdx.y;
    const uint32_t V = (uint32_t)lm.V;
    uint32_t* g_visits = lm.visits + (size_t)r * lm.n_cap;
    const uint32_t* g_off = lm.off + (size_t)r * (V + 1);
    const uint32_t o = g_off[e], n = g_off[e + 1] - o;
    if (n < 4) return;
    for (uint32_t t = lane; t < n; t += 64) route[t] = (uint16_t)g_visits[o + t];
    wave_sync();
    const bool feas = feasible_mode == 0 || lm.load[(size_t)r * V + e] <= lm.capacity;
    const uint32_t depot = (uint32_t)lm.depot;
    uint64_t cand = 0, acc = 0;
    bool changed = false;
    int sweeps = 0;
    for (;;) {
        bool improved = false;
        for (uint32_t i = 0; i + 1 < n; ++i) {
            const uint32_t a = i == 0 ? depot : (uint32_t)uni(route[i - 1]);
            const uint32_t b = uni(route[i]);
            const int64_t dab = cw_dist_cost(lm, a, b);
            for (uint32_t j0 = i + 1; j0 < n; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool imp = false;
                if (j < n) {
                    const uint32_t c = route[j];
                    const uint32_t en = j + 1 < n ? (uint32_t)route[j + 1] : depot;
                    imp = cw_dist_cost(lm, a, c) + cw_dist_cost(lm, b, en) < dab + cw_dist_cost(lm, c, en);
                }
                cand += (n - j0) < 64u ? (n - j0) : 64u;
                uint64_t mask = __ballot(imp);
                if (!feas) mask = 0ull;
                while (mask) {
                    const uint32_t jj = j0 + (uint32_t)__builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    const uint32_t half = (jj - i + 1u) / 2u;
                    wave_sync();
                    for (uint32_t t = lane; t < half; t += 64) {
                        const uint16_t x = route[i + t], y = route[jj - t];
                        route[i + t] = y;
                        route[jj - t] = x;
                    }
                    wave_sync();
                    ++acc;
                    improved = changed = true;
                }
            }
        }
        if (!improved || ++sweeps >= max_sweeps) break;  // max_sweeps = the termination policy (kernel.rs:117-121)
    }
    if (changed)
        for (uint32_t t = lane; t < n; t += 64) g_visits[o + t] = route[t];
    if (stats && lane == 0) {
        uint64_t* gs = stats + (size_t)r * SF_STATS_WORDS;
        atomicAdd((unsigned long long*)&gs[1], (unsigned long long)cand);
        atomicAdd((unsigned long long*)&gs[2], (unsigned long long)cand);
        atomicAdd((unsigned long long*)&gs[7], (unsigned long long)cand);
        atomicAdd((unsigned long long*)&gs[3], (unsigned long long)acc);
        if (changed) {
            atomicAdd((unsigned long long*)&gs[4], (unsigned long long)acc);
            atomicAdd((unsigned long long*)&gs[0], 1ull);
            atomicAdd((unsigned long long*)&gs[5], 1ull);
        }
    }
}

}  // namespace sf
