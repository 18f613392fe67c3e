// Device-side parameter blocks of the list-variable (CVRP-shaped) hot path.
#pragma once
#include <stdint.h>

#include "sf_common.h"
#include "sf_anneal.h"
#include "sf_forager.h"

namespace sf {

constexpr int SF_MAX_LEVELS_CONST = 4;
constexpr int MAX_LEAVES = 2;        // fused kernel: up to two list leaves in one union
constexpr uint32_t NODE_NONE = 0xFFFFFFFFu;
constexpr uint32_t QCAP = 1024;      // per-leaf candidate ring capacity (power of two)
constexpr uint32_t ORD_INTER_BASE = 1u << 20;
constexpr int64_t MAX_PACKED_DISTANCE = (int64_t)1 << 40;  // (distance << 24 | ordinal) key packing
constexpr int SF_STATS_WORDS = 10;  // sf_stats as u64 words

// Immutable problem facts + constraint wiring + per-replica state base pointers (SoA in HBM).
struct ListModel {
    int32_t V;          // list owners (routes)
    int32_t n_cap;      // element capacity of the flat visits array (stride per replica)
    int32_t dim;        // node-id bound (= matrix dimension when a matrix is attached)
    int32_t levels;     // score levels
    // facts
    const int64_t* mat; // dim x dim row-major (MatrixDistanceMeter + distance constraint)
    const uint32_t* mat32;  // optional compact copy: finite legs < 2^32-1 as u32, 0xFFFFFFFF = not finite
    const uint16_t* mat16;  // optional half-size copy when every finite leg < 65535 (0xFFFF = not finite): the trial gathers of the COMPACT
                            // wave kernel read it (2 MB instead of 4 at CVRP-1000: +5 %; no gain at CVRP-5000 or in the generic engine's unified delta, measured)
    int32_t mat_symmetric;  // mat[i][j] == mat[j][i] for every pair (checked on the host at upload)
    int32_t small32;        // every trial delta fits 32-bit arithmetic (all legs finite and < 2^26, small weights / loads)
    int32_t leg16;          // symmetric, compact copy present, every finite leg < 65535, dim <= 65535: 16-bit leg tables (sf_ruin.h)
    // Internal node numbering of the COMPACT wave kernel (nullptr = identity).  When the u16 matrix and the neighbour index no longer fit
    // the L2s, the host renumbers the nodes along a nearest-neighbour chain so that the legs a trial gathers (route neighbours, nearby
    // destinations) sit close to the diagonal of their matrix rows: `perm[external] = internal`, `inv[internal] = external`.  The model
    // handed to that kernel then carries the permuted mat16 / demand / depot and the permuted neighbour index; the replica's lists are
    // mapped on the way into LDS and back out, every other kernel and the whole C ABI keep the caller's ids.  Candidate order does not
    // depend on node ids (equal-distance groups are ordered by enumeration ordinal), so the trajectories are unchanged.
    const uint16_t* perm;
    const uint16_t* inv;
    // [R][dim] u16: the replica's node -> slot table in HBM (L2-resident) instead of its LDS slice -- the NODEG instantiation of the COMPACT
    // wave kernel, taken when that lets more replicas share a CU (CVRP-5000: 29 KB -> 19 KB per replica, 5 -> 8 per CU).  Scratch: rebuilt
    // from the lists at the start of every launch.
    uint16_t* node_tab;
    const int32_t* demand;
    const uint32_t* ne_keys;  // not-exists A-side keys (Customer.id)
    int32_t ne_n;
    int32_t depot;
    int64_t capacity;
    // constraints: level < 0 = absent
    int32_t cap_level, dist_level, ne_level;
    int64_t cap_weight, dist_weight, ne_weight;
    // per-replica committed state
    uint32_t* visits;  // [R][n_cap]  flat CSR values, owner-major
    uint32_t* off;     // [R][V+1]
    int64_t* load;     // [R][V]      per-route demand sum (capacity aggregate)
    int64_t* score;    // [R][SF_MAX_LEVELS] committed (cached) score
    uint32_t* best_visits;  // [R][n_cap]
    uint32_t* best_off;     // [R][V+1]
    int64_t* best_score;    // [R][4]
};

// words per replica of the evaluate_each aggregates: list class 0..2 (capacity, distance, not-exists), scalar class
// 3..11 (unassigned, predicate-join pairs, keyed self-join pairs, grouped weight sum, non-empty groups, pair-cost sum, pairs
// with a non-zero cost, existence weight sum, rows whose existence test holds)
constexpr int SF_EACH_WORDS = 18;  // + list precedence: 12 hard penalty, 13 makespan; consecutive runs: 14 summed excess, 15 groups; 16 matches of the unassigned filter; 17 join of the two planning classes

struct LeafSpec {
    int32_t kind;        // sf_selector_kind
    int32_t max_nearby;
    int32_t descriptor;  // descriptor_index (salts)
};

struct SearchParams {
    int32_t n_leaves;
    LeafSpec leaf[MAX_LEAVES];
    int32_t acceptor;     // 0 HC, 1 LA, 2 never (dry run), 3 simulated annealing (state in `sa`), 4 diversified late acceptance
    int32_t la_size;
    int32_t forager;      // sf_forager_kind: 0 accepted count, 1 first accepted, 2 best score, 3 / 4 improving (sf_forager.h)
    int32_t limit;        // accepted-count limit (forager 4: 0 = none)
    int32_t random_ties;
    int32_t order;        // sf_selection_order
    int32_t dry_run;      // 1: enumerate+score one step, no state change
    int32_t replica_base; // replica of block 0 (single-replica launches)
    int32_t n_launch;     // replicas covered by this launch (wave engine: grid rounding)
    int64_t n_steps;
    int32_t legacy_eval;  // diagnostics / parity tests: force the per-kind trial evaluators of the generic engine
    int64_t move_budget;  // > 0: a replica stops after the step in which its candidates of THIS launch reach the budget
    uint64_t random_seed; // replica r uses random_seed + r
    // dry-run explicit context
    uint64_t dry_step_index, dry_step_seed;
    // optional explicit step seeds [R][n_explicit]
    const uint64_t* explicit_seeds;
    int64_t n_explicit;
    // per-replica search state
    int64_t* last_step_score;  // [R][4]
    int64_t* la_hist;          // [R][la_size][4]
    int32_t* la_idx;           // [R]
    uint64_t* step_index;      // [R] phase step counter
    uint64_t* seed_draws;      // [R]
    uint64_t* stats;           // [R][SF_STATS_WORDS]  (sf_stats layout)
    int32_t* has_best;         // [R]
    // trace of replica `trace_replica` (TRACE kernels only)
    int32_t trace_replica;
    int32_t* trace_moves;      // [cap][6]
    int64_t* trace_scores;     // [cap][levels]
    int32_t* trace_flags;      // [cap]
    int64_t trace_cap;
    int64_t* trace_count;      // [1]
    int32_t* trace_applied;    // [1 + 6]
    SaParams sa;               // acceptor 3
    double dla_tolerance;      // acceptor 4: DiversifiedLateAcceptanceAcceptor::tolerance
    int64_t* dla_best;         // [R][4] acceptor 4: best step score of this phase
};

struct NbrIndex {
    // [dim][dim] u16 per entry: the nodes of every matrix row in ascending (distance, node) order as
    // (same-distance-as-previous flag << 15 | node); non-finite legs (negative / UNREACHABLE,
    // meters.rs:21-23) sort to the end as NBR_END.  Distances themselves are not stored: the
    // (distance, enumeration ordinal) order only needs the group boundaries.
    const uint16_t* keys;
};

}  // namespace sf
