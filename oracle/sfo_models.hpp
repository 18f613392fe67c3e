// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// The four benchmark models of BASELINE.json, expressed with the oracle's
// constraint nodes exactly as the reference examples express them with
// ConstraintFactory streams.
//
// Follows:
//   examples/nqueens/src/domain/board.rs:21-47
//   examples/scalar-graph-coloring/src/domain/graph_coloring.rs:21-44
//   examples/mixed-job-shop/src/domain/job_shop_plan.rs:28-69
//   crates/solverforge/tests/list_clarke_wright_publication/domain/publication_plan.rs:51-65
//       (CVRP "all customers assigned" not-exists)
//   examples/list-tsp/src/domain/tour_plan.rs:37-40 (uni-on-routes weight pattern)
//   crates/solverforge-cvrp/src/problem_data.rs:6-47, meters.rs:10-28
#pragma once
#include <string>
#include <memory>
#include <vector>

#include "sfo_search.hpp"

namespace sfo {

constexpr int64_t UNREACHABLE = INT64_MAX;            // problem_data.rs:6
constexpr int64_t MAX_SAFE_LEG_COST = INT64_MAX / 4;  // problem_data.rs:8

struct CvrpFacts {
    int64_t capacity = 0;
    size_t depot = 0;
    size_t dim = 0;
    std::vector<int32_t> demands;      // per node
    std::vector<int64_t> matrix;       // dim x dim row-major
    std::vector<uint32_t> customers;   // Customer.id facts (node ids)
    bool finite(size_t from, size_t to, int64_t& out) const {  // problem_data.rs:44-47
        if (from >= dim || to >= dim) return false;
        int64_t v = matrix[from * dim + to];
        if (v >= 0 && v != UNREACHABLE) {
            out = v;
            return true;
        }
        return false;
    }
    int64_t distance_cost(size_t from, size_t to) const {  // problem_data.rs:28-31
        int64_t v;
        return finite(from, to, v) ? v : MAX_SAFE_LEG_COST;
    }
    // time windows / service durations / travel times (problem_data.rs:20-23); empty = the model carries none
    std::vector<int64_t> tw_lo, tw_hi, service;  // per node
    std::vector<int64_t> travel;                  // dim x dim row-major
    int64_t departure = 0;
    bool travel_time(size_t from, size_t to, int64_t& out) const {  // problem_data.rs:38-41
        if (from >= dim || to >= dim || travel.size() != dim * dim) return false;
        const int64_t v = travel[from * dim + to];
        if (v >= 0 && v != UNREACHABLE) {
            out = v;
            return true;
        }
        return false;
    }
    // route_is_capacity_feasible / route_is_time_feasible (solverforge-cvrp/src/helpers.rs:168-218): checked i64 accumulation, waiting at a
    // window's start, service before the window's end, a traversable leg back to the depot whose arrival time does not overflow
    bool capacity_feasible(const std::vector<size_t>& route) const {
        int64_t total = 0;
        for (size_t v : route)
            if (v >= dim || __builtin_add_overflow(total, (int64_t)demands[v], &total)) return false;
        return total <= capacity;
    }
    bool time_feasible(const std::vector<size_t>& route) const {
        int64_t t = departure, leg;
        size_t prev = depot;
        for (size_t v : route) {
            if (v >= tw_lo.size() || v >= service.size()) return false;  // (route_is_structurally_valid :147-152)
            if (!travel_time(prev, v, leg) || __builtin_add_overflow(t, leg, &t)) return false;
            if (t < tw_lo[v]) t = tw_lo[v];
            if (service[v] < 0 || __builtin_add_overflow(t, service[v], &t)) return false;
            if (t > tw_hi[v]) return false;
            prev = v;
        }
        int64_t back;
        return travel_time(prev, depot, leg) && !__builtin_add_overflow(t, leg, &back);
    }
};

struct GraphFacts {
    size_t n = 0, n_colors = 0;
    std::vector<uint32_t> adj_off;  // n+1
    std::vector<uint32_t> adj;      // neighbor ids (Node.neighbors, in given order)
};
struct BalanceFacts {  // "bin balance" toy: entity sizes for the grouped-sum constraint
    std::vector<int64_t> size;
};
struct QueensFacts {
    size_t n = 0;
    std::vector<int64_t> column;
};
struct JobShopFacts {
    size_t n_ops = 0, n_machines = 0;
    std::vector<int64_t> job;
};

enum LeafBits : uint32_t {
    LEAF_SCALAR_CHANGE = 1,
    LEAF_SCALAR_SWAP = 2,
    LEAF_LIST_CHANGE = 4,
    LEAF_LIST_SWAP = 8,
    LEAF_NEARBY_LIST_CHANGE = 16,
    LEAF_NEARBY_LIST_SWAP = 32,
    LEAF_LIST_REVERSE = 64,
    LEAF_SUBLIST_CHANGE = 128,
    LEAF_SUBLIST_SWAP = 256,
    LEAF_LIST_RUIN = 1024,  // ListRuinMoveSelectorConfig defaults: 2..=5 elements, 10 moves per step (solverforge-config/src/move_selector.rs:574-587,
                            // list_leaf/spec.rs:245)
    // nearby scalar leaves (default_local_search/policy/scalar.rs:18-65: max_nearby 10, declared between the list rules and the
    // ordinary scalar change / swap pair)
    LEAF_LIST_PERMUTE = 8192,  // ListPermuteMoveSelector (window sizes 2..=5 by default, solverforge-config/src/move_selector.rs:406-416); the
                               // default policy declares it right after ListPrecedence for slots with precedence hooks (policy/list.rs:62-93)
    LEAF_LIST_PRECEDENCE = 16384,  // ListPrecedenceMoveSelector (selector/list_precedence.rs): the critical-path leaf, first list rule of slots with
                                   // precedence hooks (policy/list.rs:24-33,62-93); needs list_slot.precedence
    LEAF_NEARBY_SCALAR_CHANGE = 2048,
    LEAF_NEARBY_SCALAR_SWAP = 4096,
    LEAF_KOPT = 512,  // k = 3; kopt_max_nearby > 0: distance-pruned (default policy with an intra-distance meter), 0: full
};

struct Model {
    ScoreDirector director;
    bool has_scalar = false, has_list = false;
    ScalarSlot scalar_slot;
    ListSlot list_slot;
    LocalSearch search;
    uint32_t leaves = 0;
    size_t max_nearby = 20;
    size_t sublist_min = 1, sublist_max = 3;
    size_t kopt_min_seg = 1, kopt_max_nearby = 20;  // KOptMoveSelectorConfig defaults + DEFAULT_LIST_NEARBY_LIMIT (policy/list.rs:19,144-160)
    size_t permute_min = 2, permute_max = 5;
    size_t scalar_max_nearby = 10, scalar_source_limit = SIZE_MAX;  // NearbyChangeMoveConfig::max_nearby / value_candidate_limit
    UnionOrder union_order = UnionOrder::StratifiedRandom;
    std::vector<uint64_t> union_weights;  // UnionWeighting: one per leaf in union order; empty = equal
    // list ruin leaf: the per-solve stream state (list_leaf/cursor.rs:112-145) is one SmallRng seeded from
    // scoped_seed(random_seed, descriptor, variable, "list_ruin_move_selector"); every cursor open draws one u64 from it
    size_t ruin_min = 2, ruin_max = 5, ruin_moves_per_step = 10, ruin_max_source_len = 0;
    bool ruin_skip_empty = false;
    std::string list_variable_name = "visits";
    mutable SmallRng ruin_rng;
    void seed_ruin_stream(uint64_t random_seed) {
        ruin_rng = SmallRng::seed_from_u64(scoped_seed(random_seed, list_slot.descriptor_index, list_variable_name.c_str(), "list_ruin_move_selector"));
    }

    std::unique_ptr<Cursor> open_leaf(uint32_t leaf, const ScoreDirector& d, const MoveStreamContext& ctx) const {
        constexpr uint32_t FILTERED = LEAF_LIST_CHANGE | LEAF_LIST_SWAP | LEAF_NEARBY_LIST_CHANGE | LEAF_NEARBY_LIST_SWAP | LEAF_LIST_REVERSE |
                                      LEAF_SUBLIST_CHANGE | LEAF_SUBLIST_SWAP | LEAF_LIST_PERMUTE;
        if ((leaf & FILTERED) && list_slot.precedence_policy && list_slot.precedence)
            return std::make_unique<RouteGraphFilterCursor>(open_plain_leaf(leaf, d, ctx), list_slot, d.working);
        return open_plain_leaf(leaf, d, ctx);
    }
    std::unique_ptr<Cursor> open_plain_leaf(uint32_t leaf, const ScoreDirector& d, const MoveStreamContext& ctx) const {
        switch (leaf) {
            case LEAF_SCALAR_CHANGE:
                return std::make_unique<ScalarChangeCursor>(scalar_slot, d.working, ctx);
            case LEAF_SCALAR_SWAP:
                return std::make_unique<ScalarSwapCursor>(scalar_slot, d.working, ctx);
            case LEAF_NEARBY_SCALAR_CHANGE:
                return std::make_unique<NearbyScalarChangeCursor>(scalar_slot, d.working, ctx, scalar_max_nearby, scalar_source_limit);
            case LEAF_NEARBY_SCALAR_SWAP:
                return std::make_unique<NearbyScalarSwapCursor>(scalar_slot, d.working, ctx, scalar_max_nearby);
            case LEAF_LIST_CHANGE:
                return std::make_unique<ListChangeCursor>(list_slot, d.working, ctx);
            case LEAF_LIST_SWAP:
                return std::make_unique<ListSwapCursor>(list_slot, d.working, ctx);
            case LEAF_NEARBY_LIST_CHANGE:
                return std::make_unique<NearbyListChangeCursor>(list_slot, d.working, ctx, max_nearby);
            case LEAF_NEARBY_LIST_SWAP:
                return std::make_unique<NearbyListSwapCursor>(list_slot, d.working, ctx, max_nearby);
            case LEAF_LIST_REVERSE:
                return std::make_unique<ListReverseCursor>(list_slot, d.working, ctx);
            case LEAF_LIST_PRECEDENCE:
                return std::make_unique<ListPrecedenceCursor>(list_slot, d.working, ctx);
            case LEAF_LIST_PERMUTE:
                return std::make_unique<ListPermuteCursor>(list_slot, d.working, ctx, permute_min, permute_max);
            case LEAF_SUBLIST_SWAP:
                return std::make_unique<SublistSwapCursor>(list_slot, d.working, ctx, sublist_min, sublist_max);
            case LEAF_KOPT:
                if (kopt_max_nearby == 0) return std::make_unique<KOptCursor>(list_slot, d.working, ctx, kopt_min_seg);
                return std::make_unique<NearbyKOptCursor>(list_slot, d.working, ctx, kopt_min_seg, kopt_max_nearby);
            case LEAF_LIST_RUIN: {  // open_cursor_with_stream_state (list_leaf/cursor.rs:58-72), slot.rs:446-465
                const uint64_t seed = ruin_rng.next_u64() ^ ctx.offset_seed(0x71578011C0DE0001ULL);
                return std::make_unique<RuinCursor>(list_slot, d.working, seed, ruin_moves_per_step, ruin_min, ruin_max, ruin_max_source_len,
                                                    ruin_skip_empty);
            }
            case LEAF_SUBLIST_CHANGE:  // default sizes 1..=3 (solverforge-config/src/move_selector.rs:713-715)
                return std::make_unique<SublistChangeCursor>(list_slot, d.working, ctx, sublist_min, sublist_max);
        }
        return nullptr;
    }
    // Leaf order = default policy declaration order: list rules first (nearby change, nearby swap
    // / plain change, plain swap), then ordinary scalar change, scalar swap
    // (runtime/compiler/default_local_search/policy.rs:104-108, policy/list.rs:24-33,
    //  policy/scalar.rs:67-106).
    std::unique_ptr<Cursor> open_union(const ScoreDirector& d, const MoveStreamContext& ctx) const {
        static const uint32_t order[] = {LEAF_LIST_PRECEDENCE, LEAF_LIST_PERMUTE, LEAF_NEARBY_LIST_CHANGE, LEAF_LIST_CHANGE, LEAF_NEARBY_LIST_SWAP,
                                         LEAF_LIST_SWAP,          LEAF_SUBLIST_CHANGE, LEAF_SUBLIST_SWAP, LEAF_LIST_REVERSE,
                                         LEAF_KOPT,               LEAF_LIST_RUIN,          LEAF_NEARBY_SCALAR_CHANGE, LEAF_NEARBY_SCALAR_SWAP,
                                         LEAF_SCALAR_CHANGE,      LEAF_SCALAR_SWAP};
        std::vector<std::unique_ptr<Cursor>> children;
        for (uint32_t leaf : order)
            if (leaves & leaf) children.push_back(open_leaf(leaf, d, ctx));
        if (children.size() == 1) return std::move(children[0]);
        return std::make_unique<UnionCursor>(std::move(children), union_order, ctx, union_weights);
    }
    void wire_search() {
        search.director = &director;
        search.open_cursor = [this](const ScoreDirector& d, const MoveStreamContext& ctx) {
            return open_union(d, ctx);
        };
    }
};

inline std::unique_ptr<UniConstraint> make_unassigned(size_t desc, size_t var, const Score& w,
                                                      const char* name) {
    // for_each(entities).unassigned().penalize(w)
    auto c = std::make_unique<UniConstraint>();
    c->name = name;
    c->impact = Impact::Penalty;
    c->source = ChangeSource::descriptor(desc);
    c->count = [desc](const Solution& s) { return s.classes[desc].n; };
    c->filter = [desc, var](const Solution& s, size_t i) { return s.classes[desc].vars[var][i] == NONE; };
    c->weight = [w](const Solution&, size_t) { return w; };
    return c;
}

// ---- N-queens (board.rs:21-47) --------------------------------------------
inline std::unique_ptr<Model> make_nqueens(size_t n, const int64_t* rows) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<QueensFacts>();
    facts->n = n;
    facts->column.resize(n);
    for (size_t i = 0; i < n; ++i) facts->column[i] = (int64_t)i;
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n;
    s.classes[0].vars.assign(1, std::vector<int64_t>(rows, rows + n));
    s.facts = facts;
    m->director.constraints.members.push_back(make_unassigned(0, 0, Score::of(1, 0), "Unassigned queen"));
    auto conflict = std::make_unique<CrossBiConstraint>();
    conflict->name = "Queen conflict";
    conflict->impact = Impact::Penalty;
    conflict->a_source = conflict->b_source = ChangeSource::descriptor(0);
    conflict->a_count = conflict->b_count = [](const Solution& s) { return s.classes[0].n; };
    conflict->key_a = conflict->key_b = [](const Solution&, size_t) { return (int64_t)0; };  // predicate join
    const QueensFacts* qf = facts.get();
    conflict->filter = [qf](const Solution& s, size_t a, size_t b) {
        int64_t ca = qf->column[a], cb = qf->column[b];
        if (ca >= cb) return false;
        int64_t ra = s.classes[0].vars[0][a], rb = s.classes[0].vars[0][b];
        if (ra == NONE || rb == NONE) return false;
        int64_t dr = ra > rb ? ra - rb : rb - ra;
        int64_t dc = ca > cb ? ca - cb : cb - ca;
        return ra == rb || dr == dc;
    };
    conflict->weight = [](const Solution&, size_t, size_t) { return Score::of(1, 0); };
    m->director.constraints.members.push_back(std::move(conflict));
    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n; ++v) out.push_back((int64_t)v);
    };
    m->leaves = LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

// ---- graph colouring (graph_coloring.rs:21-44) ----------------------------
// `indexed`: the INDEXED CPU baseline (PartnerEqualConstraint) instead of the reference's dense predicate join
inline void partners_from_adjacency(size_t n, const std::vector<uint32_t>& off, const std::vector<uint32_t>& adj, std::vector<uint32_t>& poff,
                                    std::vector<uint32_t>& pn) {  // {a, b} is a pair iff b is in neighbors(a) for the lower index a
    std::vector<std::vector<uint32_t>> lists(n);
    for (size_t a = 0; a < n; ++a) {
        std::vector<uint32_t> seen;
        for (uint32_t p = off[a]; p < off[a + 1]; ++p) {
            uint32_t b = adj[p];
            if (b >= n || b <= a || std::find(seen.begin(), seen.end(), b) != seen.end()) continue;
            seen.push_back(b);
            lists[a].push_back(b);
            lists[b].push_back((uint32_t)a);
        }
    }
    poff.assign(n + 1, 0);
    pn.clear();
    for (size_t a = 0; a < n; ++a) {
        poff[a] = (uint32_t)pn.size();
        pn.insert(pn.end(), lists[a].begin(), lists[a].end());
    }
    poff[n] = (uint32_t)pn.size();
}
inline std::unique_ptr<Model> make_graph_coloring(size_t n, size_t n_colors, const uint32_t* adj_off,
                                                  const uint32_t* adj, const int64_t* colors, bool indexed = false) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<GraphFacts>();
    facts->n = n;
    facts->n_colors = n_colors;
    facts->adj_off.assign(adj_off, adj_off + n + 1);
    facts->adj.assign(adj, adj + adj_off[n]);
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n;
    s.classes[0].vars.assign(1, std::vector<int64_t>(colors, colors + n));
    s.facts = facts;
    m->director.constraints.members.push_back(make_unassigned(0, 0, Score::of(1, 0), "Unassigned color"));
    auto conflict = std::make_unique<CrossBiConstraint>();
    conflict->name = "Adjacent color conflict";
    conflict->impact = Impact::Penalty;
    conflict->a_source = conflict->b_source = ChangeSource::descriptor(0);
    conflict->a_count = conflict->b_count = [](const Solution& s) { return s.classes[0].n; };
    conflict->key_a = conflict->key_b = [](const Solution&, size_t) { return (int64_t)0; };
    const GraphFacts* gf = facts.get();
    conflict->filter = [gf](const Solution& s, size_t a, size_t b) {
        if (!(a < b)) return false;  // left.id < right.id (ids are indices)
        bool adjacent = false;       // left.neighbors.contains(&right.id): linear Vec scan
        for (uint32_t p = gf->adj_off[a]; p < gf->adj_off[a + 1]; ++p)
            if (gf->adj[p] == b) {
                adjacent = true;
                break;
            }
        if (!adjacent) return false;
        int64_t ca = s.classes[0].vars[0][a];
        return ca != NONE && ca == s.classes[0].vars[0][b];
    };
    conflict->weight = [](const Solution&, size_t, size_t) { return Score::of(1, 0); };
    if (indexed) {
        auto ix = std::make_unique<PartnerEqualConstraint>();
        ix->name = "Adjacent color conflict (indexed)";
        ix->impact = Impact::Penalty;
        ix->source = ChangeSource::descriptor(0);
        ix->count = [](const Solution& s) { return s.classes[0].n; };
        ix->value = [](const Solution& s, size_t e) { return s.classes[0].vars[0][e]; };
        partners_from_adjacency(n, facts->adj_off, facts->adj, ix->poff, ix->pn);
        ix->weight = Score::of(1, 0);
        m->director.constraints.members.push_back(std::move(ix));
    } else
        m->director.constraints.members.push_back(std::move(conflict));
    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n_colors](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n_colors; ++v) out.push_back((int64_t)v);
    };
    m->leaves = LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

// ---- bin balance: keyed self-join + grouped sum on one scalar variable ------------------------
// The node shapes the reference pins in constraint/tests/bi_incr.rs:21-311 (self-join keyed by the
// planning value, penalty per pair) and constraint/tests/grouped.rs:26-147 + cross_bi_incr.rs:97-120
// (group by key, sum collector, weight(key, sum)), on a model with a ScalarChange/ScalarSwap
// neighbourhood: entities carry a size, the planning value is a bin.
//   level 0: unassigned entity (uni), 1 each
//   level 1: pairs of entities in the same bin (IncrementalBiConstraint keyed by bin), `w_pair` each
//   level 1: per bin, weight(bin, sum of sizes): cap < 0 -> sum^2 ; cap >= 0 -> max(0, sum - cap)
inline std::unique_ptr<Model> make_balance(size_t n, size_t n_bins, const int64_t* bins, const int64_t* sizes,
                                           int64_t w_pair, int64_t cap, size_t arity = 2, int64_t balance_base = 1000) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<BalanceFacts>();
    facts->size.assign(sizes, sizes + n);
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n;
    s.classes[0].vars.assign(1, std::vector<int64_t>(bins, bins + n));
    s.facts = facts;
    const BalanceFacts* bf = facts.get();
    m->director.constraints.members.push_back(make_unassigned(0, 0, Score::of(1, 0), "Unassigned bin"));

    auto pairs = std::make_unique<SelfJoinBiConstraint>();
    pairs->name = "Same bin pair";
    pairs->impact = Impact::Penalty;
    pairs->source = ChangeSource::descriptor(0);
    pairs->count = [](const Solution& s) { return s.classes[0].n; };
    pairs->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    pairs->filter = [](const Solution& s, size_t a, size_t) { return s.classes[0].vars[0][a] != NONE; };
    pairs->weight = [w_pair](const Solution&, size_t, size_t) { return Score::of(0, w_pair); };
    if (arity <= 2) {
        m->director.constraints.members.push_back(std::move(pairs));
    } else {  // tri / quad / penta: tuples of assigned entities sharing a bin
        auto tuples = std::make_unique<SelfJoinNaryConstraint>();
        tuples->name = "Same bin tuple";
        tuples->arity = arity;
        tuples->impact = Impact::Penalty;
        tuples->source = ChangeSource::descriptor(0);
        tuples->count = [](const Solution& s) { return s.classes[0].n; };
        tuples->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        tuples->filter = [](const Solution& s, const size_t* idx) { return s.classes[0].vars[0][idx[0]] != NONE; };
        tuples->weight = [w_pair](const Solution&, const size_t*) { return Score::of(0, w_pair); };
        m->director.constraints.members.push_back(std::move(tuples));
    }

    if (cap == -2) {  // fairness instead of the per-bin load: group_by(load_balance(bin, size)).penalize(unfairness)
        auto fair = std::make_unique<LoadBalanceConstraint>();
        fair->name = "Bin fairness";
        fair->impact = Impact::Penalty;
        fair->source = ChangeSource::descriptor(0);
        fair->count = [](const Solution& s) { return s.classes[0].n; };
        fair->filter = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i] != NONE; };
        fair->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        fair->metric = [bf](const Solution&, size_t i) { return bf->size[i]; };
        fair->weight = [](int64_t unfairness) { return Score::of(0, unfairness); };
        m->director.constraints.members.push_back(std::move(fair));
    }
    auto load = std::make_unique<GroupedConstraint>();
    load->name = "Bin load";
    load->impact = Impact::Penalty;
    load->source = ChangeSource::descriptor(0);
    load->count = [](const Solution& s) { return s.classes[0].n; };
    load->filter = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i] != NONE; };
    load->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    load->value = [bf](const Solution&, size_t i) { return bf->size[i]; };
    load->weight = [cap](int64_t, int64_t sum) {
        if (cap < 0) return Score::of(0, (int64_t)((uint64_t)sum * (uint64_t)sum));
        int64_t over = wrap_sub(sum, cap);
        return Score::of(0, over > 0 ? over : 0);
    };
    if (cap == -3) {  // BalanceConstraint (constraint/balance.rs): `balance_base` (default 1000) soft per unit of the standard deviation of the bin COUNTS
        auto bal = std::make_unique<BalanceConstraint>();
        bal->name = "Bin count balance";
        bal->impact = Impact::Penalty;
        bal->source = ChangeSource::descriptor(0);
        bal->count = [](const Solution& s) { return s.classes[0].n; };
        bal->filter = [](const Solution&, size_t) { return true; };
        bal->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        bal->base_score = Score::of(0, balance_base);
        m->director.constraints.members.push_back(std::move(bal));
    }
    if (cap != -2 && cap != -3) m->director.constraints.members.push_back(std::move(load));

    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n_bins](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n_bins; ++v) out.push_back((int64_t)v);
    };
    m->leaves = LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

// ---- CVRP -------------------------------------------------------------------
// ---- assignment: keyed cross-join with a fact class + exists / not-exists of planning entities per fact row -------------
// The node shapes the reference pins in constraint/tests/cross_bi_incr.rs:60-83,205-381 (shifts joined with employees by
// equal(shift.employee_id, employee.id), a filter and a weight on the pair) and constraint/tests/exists.rs:34-190
// (for_each(A).if_exists / if_not_exists(B, equal keys)), on a model with a ScalarChange / ScalarSwap neighbourhood:
// n entities ("shifts") choose one of n_values fact rows ("employees").
//   level 0: unassigned entity (uni), 1 each
//   level 1: cross_bi A x B keyed by the planning value: filter cost[a][b] != 0, weight cost[a][b]
//   level `ex_level`: for every fact row b with an entity holding it (ex_mode 1) / with none (ex_mode 0): ex_weight * row_w[b]
struct AssignFacts {
    size_t n = 0, n_values = 0;
    std::vector<int64_t> cost;   // [n][n_values]
    std::vector<int64_t> row_w;  // [n_values]
    std::vector<int64_t> cost2;  // [n][n_values]: a second keyed cross-join on another level (cost2_level >= 0), weight 1
};
inline std::unique_ptr<Model> make_assignment(size_t n, size_t n_values, const int64_t* values, const int64_t* cost, int64_t cost_weight,
                                              const int64_t* row_w, int32_t ex_mode, int32_t ex_level, int64_t ex_weight, const int64_t* cost2 = nullptr,
                                              int32_t cost2_level = -1) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<AssignFacts>();
    facts->n = n;
    facts->n_values = n_values;
    facts->cost.assign(cost, cost + n * n_values);
    facts->row_w.assign(row_w, row_w + n_values);
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n;
    s.classes[0].vars.assign(1, std::vector<int64_t>(values, values + n));
    s.facts = facts;
    const AssignFacts* af = facts.get();
    m->director.constraints.members.push_back(make_unassigned(0, 0, Score::of(1, 0), "Unassigned"));

    auto join = std::make_unique<CrossBiConstraint>();
    join->name = "Pair cost";
    join->impact = Impact::Penalty;
    join->a_source = ChangeSource::descriptor(0);
    join->b_source = ChangeSource::fixed();  // problem facts
    join->a_count = [](const Solution& s) { return s.classes[0].n; };
    join->b_count = [af](const Solution&) { return af->n_values; };
    join->key_a = [](const Solution& s, size_t a) { return s.classes[0].vars[0][a]; };  // None never equals a row id
    join->key_b = [](const Solution&, size_t b) { return (int64_t)b; };
    join->filter = [af](const Solution& s, size_t a, size_t b) { return s.classes[0].vars[0][a] != NONE && af->cost[a * af->n_values + b] != 0; };
    join->weight = [af, cost_weight](const Solution&, size_t a, size_t b) { return Score::of(0, wrap_mul(cost_weight, af->cost[a * af->n_values + b])); };
    m->director.constraints.members.push_back(std::move(join));
    if (cost2 && cost2_level >= 0) {  // the same node shape once more, on `cost2_level` (uni filters / weights compiled onto two levels, round 6)
        facts->cost2.assign(cost2, cost2 + n * n_values);
        auto j2 = std::make_unique<CrossBiConstraint>();
        j2->name = "Pair cost 2";
        j2->impact = Impact::Penalty;
        j2->a_source = ChangeSource::descriptor(0);
        j2->b_source = ChangeSource::fixed();
        j2->a_count = [](const Solution& s) { return s.classes[0].n; };
        j2->b_count = [af](const Solution&) { return af->n_values; };
        j2->key_a = [](const Solution& s, size_t a) { return s.classes[0].vars[0][a]; };
        j2->key_b = [](const Solution&, size_t b) { return (int64_t)b; };
        j2->filter = [af](const Solution& s, size_t a, size_t b) { return s.classes[0].vars[0][a] != NONE && af->cost2[a * af->n_values + b] != 0; };
        j2->weight = [af, cost2_level](const Solution&, size_t a, size_t b) { return Score::level(cost2_level, af->cost2[a * af->n_values + b]); };
        m->director.constraints.members.push_back(std::move(j2));
    }

    if (ex_level >= 0) {
        auto ex = std::make_unique<ExistsConstraint>();
        ex->name = ex_mode ? "Row in use" : "Row unused";
        ex->impact = Impact::Penalty;
        ex->mode = ex_mode ? ExistenceMode::Exists : ExistenceMode::NotExists;
        ex->a_source = ChangeSource::fixed();  // the fact rows
        ex->parent_source = ChangeSource::descriptor(0);
        ex->a_count = [af](const Solution&) { return af->n_values; };
        ex->parent_count = [](const Solution& s) { return s.classes[0].n; };
        ex->filter_a = [](const Solution&, size_t) { return true; };
        ex->filter_parent = [](const Solution& s, size_t p) { return s.classes[0].vars[0][p] != NONE; };
        ex->key_a = [](const Solution&, size_t b) { return (int64_t)b; };
        ex->flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) { out.push_back(s.classes[0].vars[0][p]); };
        ex->weight = [af, ex_level, ex_weight](const Solution&, size_t b) { return Score::level(ex_level, wrap_mul(ex_weight, af->row_w[b])); };
        ex->indexed_usize = true;
        m->director.constraints.members.push_back(std::move(ex));
    }
    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n_values](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n_values; ++v) out.push_back((int64_t)v);
    };
    m->leaves = LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

inline int64_t cvrp_route_distance(const CvrpFacts& f, const std::vector<uint32_t>& route) {
    if (route.empty()) return 0;
    int64_t total = f.distance_cost(f.depot, route[0]);
    for (size_t i = 0; i + 1 < route.size(); ++i) total = wrap_add(total, f.distance_cost(route[i], route[i + 1]));
    return wrap_add(total, f.distance_cost(route.back(), f.depot));
}
inline int64_t cvrp_route_load(const CvrpFacts& f, const std::vector<uint32_t>& route) {
    int64_t total = 0;
    for (uint32_t v : route) total = wrap_add(total, (int64_t)f.demands[v]);
    return total;
}

inline std::unique_ptr<Model> make_cvrp(size_t n_customers, size_t n_vehicles, int64_t capacity, size_t depot,
                                        size_t dim, const int32_t* demands, const int64_t* matrix,
                                        const uint32_t* customers, const uint32_t* route_off,
                                        const uint32_t* route_vals) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<CvrpFacts>();
    facts->capacity = capacity;
    facts->depot = depot;
    facts->dim = dim;
    facts->demands.assign(demands, demands + dim);
    facts->matrix.assign(matrix, matrix + dim * dim);
    facts->customers.assign(customers, customers + n_customers);
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n_vehicles;
    s.classes[0].lists.resize(n_vehicles);
    for (size_t v = 0; v < n_vehicles; ++v)
        s.classes[0].lists[v].assign(route_vals + route_off[v], route_vals + route_off[v + 1]);
    s.facts = facts;
    const CvrpFacts* cf = facts.get();

    auto assigned = std::make_unique<ExistsConstraint>();
    assigned->name = "all_customers_assigned";
    assigned->impact = Impact::Penalty;
    assigned->mode = ExistenceMode::NotExists;
    assigned->a_source = ChangeSource::fixed();           // problem facts
    assigned->parent_source = ChangeSource::descriptor(0);  // routes
    assigned->a_count = [cf](const Solution&) { return cf->customers.size(); };
    assigned->parent_count = [](const Solution& s) { return s.classes[0].n; };
    assigned->filter_a = [](const Solution&, size_t) { return true; };
    assigned->filter_parent = [](const Solution&, size_t) { return true; };
    assigned->key_a = [cf](const Solution&, size_t i) { return (int64_t)cf->customers[i]; };
    assigned->flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) {
        for (uint32_t v : s.classes[0].lists[p]) out.push_back((int64_t)v);
    };
    assigned->weight = [](const Solution&, size_t) { return Score::of(1, 0); };
    assigned->indexed_usize = true;
    m->director.constraints.members.push_back(std::move(assigned));

    auto cap = std::make_unique<UniConstraint>();
    cap->name = "vehicle_capacity";
    cap->impact = Impact::Penalty;
    cap->source = ChangeSource::descriptor(0);
    cap->count = [](const Solution& s) { return s.classes[0].n; };
    cap->filter = [](const Solution&, size_t) { return true; };
    cap->weight = [cf](const Solution& s, size_t r) {
        int64_t over = wrap_sub(cvrp_route_load(*cf, s.classes[0].lists[r]), cf->capacity);
        return Score::of(over > 0 ? over : 0, 0);
    };
    m->director.constraints.members.push_back(std::move(cap));

    auto dist = std::make_unique<UniConstraint>();
    dist->name = "total_distance";
    dist->impact = Impact::Penalty;
    dist->source = ChangeSource::descriptor(0);
    dist->count = [](const Solution& s) { return s.classes[0].n; };
    dist->filter = [](const Solution&, size_t) { return true; };
    dist->weight = [cf](const Solution& s, size_t r) {
        return Score::of(0, cvrp_route_distance(*cf, s.classes[0].lists[r]));
    };
    m->director.constraints.members.push_back(std::move(dist));

    m->has_list = true;
    m->list_slot.descriptor_index = 0;
    // MatrixDistanceMeter (meters.rs:10-28)
    m->list_slot.meter = [cf](const Solution& s, size_t se, size_t sp, size_t de, size_t dp) -> double {
        const auto& sv = s.classes[0].lists[se];
        const auto& dv = s.classes[0].lists[de];
        if (sp >= sv.size() || dp >= dv.size()) return std::numeric_limits<double>::infinity();
        int64_t v;
        if (!cf->finite(sv[sp], dv[dp], v)) return std::numeric_limits<double>::infinity();
        return (double)v;
    };
    m->leaves = LEAF_NEARBY_LIST_CHANGE | LEAF_NEARBY_LIST_SWAP;
    m->wire_search();
    return m;
}

// ---- plain list model for selector goldens (Vehicle/Plan toy domain of
// heuristic/selector/tests/*.rs and benches/selector_cursor_gate.rs) -------------
enum class ToyMeter { Equal, Position };
inline std::unique_ptr<Model> make_list_toy(size_t n_entities, const uint32_t* off, const uint32_t* vals,
                                            ToyMeter meter) {
    auto m = std::make_unique<Model>();
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n_entities;
    s.classes[0].lists.resize(n_entities);
    for (size_t v = 0; v < n_entities; ++v) s.classes[0].lists[v].assign(vals + off[v], vals + off[v + 1]);
    m->has_list = true;
    m->list_slot.descriptor_index = 0;
    if (meter == ToyMeter::Equal)  // EqualDistanceMeter (tests/nearby_list.rs)
        m->list_slot.meter = [](const Solution&, size_t, size_t, size_t, size_t) { return 1.0; };
    else  // PositionDistanceMeter (benches/selector_cursor_gate.rs:131-146)
        m->list_slot.meter = [](const Solution&, size_t se, size_t sp, size_t de, size_t dp) {
            double de_ = (double)(se > de ? se - de : de - se);
            double dp_ = (double)(sp > dp ? sp - dp : dp - sp);
            return de_ * 100.0 + dp_;
        };
    m->leaves = LEAF_LIST_CHANGE;
    m->wire_search();
    return m;
}

// ---- mixed job shop (job_shop_plan.rs:28-69) mapped to BendableScore<2,1> -----
// level 0 = hard[0] unassigned machine, level 1 = hard[1] unscheduled operation,
// level 2 = soft[0] same-job-same-machine.  (The reference example itself uses
// HardSoftScore; Bendable is a BASELINE.json requirement)
inline std::unique_ptr<Model> make_jobshop(size_t n_ops, size_t n_machines, const int64_t* job,
                                           const int64_t* machine_idx, const uint32_t* seq_off,
                                           const uint32_t* seq_vals, bool bendable, bool indexed = false,
                                           const int64_t* duration = nullptr, int owner_match_level = -1) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<JobShopFacts>();
    facts->n_ops = n_ops;
    facts->n_machines = n_machines;
    facts->job.assign(job, job + n_ops);
    Solution& s = m->director.working;
    s.classes.resize(2);
    s.classes[0].n = n_ops;
    s.classes[0].vars.assign(1, std::vector<int64_t>(machine_idx, machine_idx + n_ops));
    s.classes[1].n = n_machines;
    s.classes[1].lists.resize(n_machines);
    for (size_t v = 0; v < n_machines; ++v)
        s.classes[1].lists[v].assign(seq_vals + seq_off[v], seq_vals + seq_off[v + 1]);
    s.facts = facts;
    const JobShopFacts* jf = facts.get();
    Score w_unassigned = bendable ? Score::level(0, 1) : Score::of(1, 0);
    Score w_unscheduled = bendable ? Score::level(1, 1) : Score::of(1, 0);
    Score w_reuse = bendable ? Score::level(2, 1) : Score::of(0, 1);
    if (bendable) {
        m->director.levels = 3;
        m->director.hard_levels = 2;
    }
    m->director.constraints.members.push_back(make_unassigned(0, 0, w_unassigned, "Unassigned operation machine"));

    auto unsched = std::make_unique<ExistsConstraint>();
    unsched->name = "Unscheduled operation";
    unsched->impact = Impact::Penalty;
    unsched->mode = ExistenceMode::NotExists;
    unsched->a_source = ChangeSource::descriptor(0);
    unsched->parent_source = ChangeSource::descriptor(1);
    unsched->a_count = [](const Solution& s) { return s.classes[0].n; };
    unsched->parent_count = [](const Solution& s) { return s.classes[1].n; };
    unsched->filter_a = [](const Solution&, size_t) { return true; };
    unsched->filter_parent = [](const Solution&, size_t) { return true; };
    unsched->key_a = [](const Solution&, size_t i) { return (int64_t)i; };  // operation.id
    unsched->flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) {
        for (uint32_t v : s.classes[1].lists[p]) out.push_back((int64_t)v);
    };
    unsched->weight = [w_unscheduled](const Solution&, size_t) { return w_unscheduled; };
    m->director.constraints.members.push_back(std::move(unsched));

    auto reuse = std::make_unique<CrossBiConstraint>();
    reuse->name = "Same job machine reuse";
    reuse->impact = Impact::Penalty;
    reuse->a_source = reuse->b_source = ChangeSource::descriptor(0);
    reuse->a_count = reuse->b_count = [](const Solution& s) { return s.classes[0].n; };
    reuse->key_a = reuse->key_b = [](const Solution&, size_t) { return (int64_t)0; };
    reuse->filter = [jf](const Solution& s, size_t a, size_t b) {
        if (!(a < b)) return false;
        if (jf->job[a] != jf->job[b]) return false;
        int64_t ma = s.classes[0].vars[0][a];
        return ma != NONE && ma == s.classes[0].vars[0][b];
    };
    reuse->weight = [w_reuse](const Solution&, size_t, size_t) { return w_reuse; };
    if (indexed) {  // partners = the other operations of the same job
        auto ix = std::make_unique<PartnerEqualConstraint>();
        ix->name = "Same job machine reuse (indexed)";
        ix->impact = Impact::Penalty;
        ix->source = ChangeSource::descriptor(0);
        ix->count = [](const Solution& s) { return s.classes[0].n; };
        ix->value = [](const Solution& s, size_t e) { return s.classes[0].vars[0][e]; };
        std::unordered_map<int64_t, std::vector<uint32_t>> members;
        for (size_t e = 0; e < n_ops; ++e) members[jf->job[e]].push_back((uint32_t)e);
        ix->poff.assign(n_ops + 1, 0);
        for (size_t e = 0; e < n_ops; ++e) {
            ix->poff[e] = (uint32_t)ix->pn.size();
            for (uint32_t o2 : members[jf->job[e]])
                if (o2 != e) ix->pn.push_back(o2);
        }
        ix->poff[n_ops] = (uint32_t)ix->pn.size();
        ix->weight = w_reuse;
        m->director.constraints.members.push_back(std::move(ix));
    } else
        m->director.constraints.members.push_back(std::move(reuse));

    if (owner_match_level >= 0) {
        // A join of the TWO planning classes, both sides moving (constraint/cross_bi_incremental/incremental.rs:93-137: an insert / retract
        // of an operation runs insert_a / retract_a, of a machine insert_b / retract_b; state.rs:372-461): every operation whose assigned
        // machine does not schedule it -- for_each(Operation).join(for_each(Machine), equal(op.machine_idx, machine.id))
        // .filter(|op, m| !m.sequence.contains(op.id)).penalize(1).  Scalar moves change the A side's key, list moves the B side's filter.
        auto mm = std::make_unique<CrossBiConstraint>();
        mm->name = "Operation on a machine that does not schedule it";
        mm->impact = Impact::Penalty;
        mm->a_source = ChangeSource::descriptor(0);
        mm->b_source = ChangeSource::descriptor(1);
        mm->a_count = [](const Solution& s) { return s.classes[0].n; };
        mm->b_count = [](const Solution& s) { return s.classes[1].n; };
        mm->key_a = [](const Solution& s, size_t a) { return s.classes[0].vars[0][a]; };  // NONE (-1) joins no machine
        mm->key_b = [](const Solution&, size_t b) { return (int64_t)b; };
        mm->filter = [](const Solution& s, size_t a, size_t b) {
            for (uint32_t v : s.classes[1].lists[b])
                if ((size_t)v == a) return false;
            return true;
        };
        const Score w_mm = bendable ? Score::level(owner_match_level, 1) : Score::of(owner_match_level == 0 ? 1 : 0, owner_match_level == 0 ? 0 : 1);
        mm->weight = [w_mm](const Solution&, size_t, size_t) { return w_mm; };
        m->director.constraints.members.push_back(std::move(mm));
    }

    if (duration) {  // the makespan objective: ListPrecedenceMakespanConstraint over the job order and the machine sequences
        auto dur = std::make_shared<std::vector<int64_t>>(duration, duration + n_ops);
        auto c = std::make_unique<ListPrecedenceConstraint>();
        c->name = "listPrecedenceMakespan";
        c->is_hard = true;
        c->list_descriptor = 1;
        c->node_count = [n_ops](const Solution&) { return n_ops; };
        c->node_duration = [dur](const Solution&, size_t n) { return (*dur)[n]; };
        c->fixed_successors = [jf, n_ops](const Solution&, size_t n, std::vector<size_t>& out) {  // the next operation of the same job
            if (n + 1 < n_ops && jf->job[n + 1] == jf->job[n]) out.push_back(n + 1);
        };
        c->owner_count = [](const Solution& s) { return s.classes[1].n; };
        c->list_len = [](const Solution& s, size_t o) { return s.classes[1].lists[o].size(); };
        c->list_get = [](const Solution& s, size_t o, size_t p) { return (int64_t)s.classes[1].lists[o][p]; };
        c->hard = bendable ? Score::level(1, 1) : Score::of(1, 0);
        c->soft = bendable ? Score::level(2, 1) : Score::of(0, 1);
        m->director.constraints.members.push_back(std::move(c));
        auto hooks = std::make_shared<PrecedenceHooks>();  // the same facts as the hooks of the critical-path leaf / the slot's policy
        hooks->node_count = n_ops;
        hooks->durations = *dur;
        hooks->successors.resize(n_ops);
        for (size_t v = 0; v + 1 < n_ops; ++v)
            if (jf->job[v + 1] == jf->job[v]) hooks->successors[v].push_back(v + 1);
        m->list_slot.precedence = hooks;
    }
    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n_machines](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n_machines; ++v) out.push_back((int64_t)v);
    };
    m->has_list = true;
    m->list_slot.descriptor_index = 1;
    m->leaves = LEAF_LIST_CHANGE | LEAF_LIST_SWAP | LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

// ---- precedence shop: owners (machines) hold lists of node ids (operations); the only constraint is the
// ListPrecedenceMakespanConstraint (constraint/list_precedence.rs:13-707) over the fixed successor relation (job order) and
// the consecutive list items.  HardSoftScore by default; `hard_level` / `soft_level` place the two penalties on other levels. -
struct PrecedenceFacts {
    std::vector<int64_t> duration, expected_owner;  // expected_owner: NONE = no expectation
    std::vector<uint32_t> succ_off, succ;
    bool has_owner = false;
};
inline std::unique_ptr<Model> make_precedence_shop(size_t n_nodes, size_t n_owners, const int64_t* duration, const uint32_t* succ_off,
                                                   const uint32_t* succ, const int64_t* expected_owner, const uint32_t* list_off,
                                                   const uint32_t* list_vals, int levels, int hard_levels, int hard_level, int soft_level) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<PrecedenceFacts>();
    facts->duration.assign(duration, duration + n_nodes);
    facts->succ_off.assign(succ_off, succ_off + n_nodes + 1);
    facts->succ.assign(succ, succ + succ_off[n_nodes]);
    if (expected_owner) {
        facts->expected_owner.assign(expected_owner, expected_owner + n_nodes);
        facts->has_owner = true;
    }
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n_owners;
    s.classes[0].lists.resize(n_owners);
    for (size_t v = 0; v < n_owners; ++v) s.classes[0].lists[v].assign(list_vals + list_off[v], list_vals + list_off[v + 1]);
    s.facts = facts;
    const PrecedenceFacts* pf = facts.get();
    m->director.levels = levels;
    m->director.hard_levels = hard_levels;
    auto c = std::make_unique<ListPrecedenceConstraint>();
    c->name = "listPrecedenceMakespan";
    c->is_hard = true;
    c->list_descriptor = 0;
    c->node_count = [pf](const Solution&) { return pf->duration.size(); };
    c->node_duration = [pf](const Solution&, size_t n) { return pf->duration[n]; };
    c->fixed_successors = [pf](const Solution&, size_t n, std::vector<size_t>& out) {
        for (uint32_t t = pf->succ_off[n]; t < pf->succ_off[n + 1]; ++t) out.push_back((size_t)pf->succ[t]);
    };
    c->owner_count = [](const Solution& s) { return s.classes[0].n; };
    c->list_len = [](const Solution& s, size_t o) { return s.classes[0].lists[o].size(); };
    c->list_get = [](const Solution& s, size_t o, size_t p) { return (int64_t)s.classes[0].lists[o][p]; };
    if (pf->has_owner) c->expected_owner = [pf](const Solution&, size_t n) { return pf->expected_owner[n]; };
    c->hard = Score::level(hard_level, 1);
    c->soft = Score::level(soft_level, 1);
    m->director.constraints.members.push_back(std::move(c));
    m->has_list = true;
    m->list_slot.descriptor_index = 0;
    {  // the constraint's graph facts double as the hooks of the critical-path leaf (fixed_successors / node_duration)
        auto hooks = std::make_shared<PrecedenceHooks>();
        hooks->node_count = n_nodes;
        hooks->durations = facts->duration;
        hooks->successors.resize(n_nodes);
        for (size_t v = 0; v < n_nodes; ++v)
            for (uint32_t t = succ_off[v]; t < succ_off[v + 1]; ++t) hooks->successors[v].push_back((size_t)succ[t]);
        m->list_slot.precedence = hooks;
    }
    m->leaves = LEAF_LIST_CHANGE | LEAF_LIST_SWAP;
    m->wire_search();
    return m;
}

// ---- shift scheduling (examples/minimal-shift-scheduling/src/domain/schedule.rs:21-83): shifts choose a nurse ---------------
//   hard: unassigned shift (uni); two shifts of one nurse on one day (predicate cross-join, :31-43)
//   soft: long work streaks -- group_by(nurse, consecutive_runs(day)).penalize(sum over runs of max(0, point_count - limit)) (:45-59)
//   soft (count_weight > 0): group_by(nurse, count()).penalize(count^2) (the grouped count node; the example's complemented
//         |count - target| form needs the complement node, which this build does not restate)
struct ShiftFacts {
    std::vector<int64_t> day, required;  // required[i]: weight of "unassigned required shift" for shift i (0 = not required)
};
inline std::unique_ptr<Model> make_shift_schedule(size_t n_shifts, size_t n_nurses, const int64_t* nurse_idx, const int64_t* day, int64_t limit,
                                                  int64_t w_streak, int64_t count_weight, int64_t target = -1,
                                                  const int64_t* required = nullptr, int64_t presence_lo = -1, int64_t presence_hi = -1,
                                                  int64_t presence_cap = 0, int64_t presence_mode = 0) {
    auto m = std::make_unique<Model>();
    auto facts = std::make_shared<ShiftFacts>();
    facts->day.assign(day, day + n_shifts);
    Solution& s = m->director.working;
    s.classes.resize(1);
    s.classes[0].n = n_shifts;
    s.classes[0].vars.assign(1, std::vector<int64_t>(nurse_idx, nurse_idx + n_shifts));
    s.facts = facts;
    if (required) facts->required.assign(required, required + n_shifts);
    const ShiftFacts* sf = facts.get();
    {  // for_each(shifts).filter(required && nurse_idx.is_none()).penalize(ONE_HARD) (schedule.rs:23-27)
        auto un = make_unassigned(0, 0, Score::of(1, 0), "Unassigned required shift");
        if (required) {
            un->filter = [sf](const Solution& s, size_t i) { return sf->required[i] != 0 && s.classes[0].vars[0][i] == NONE; };
            un->weight = [sf](const Solution&, size_t i) { return Score::of(sf->required[i], 0); };
        }
        m->director.constraints.members.push_back(std::move(un));
    }

    auto clash = std::make_unique<CrossBiConstraint>();
    clash->name = "One shift per nurse day";
    clash->impact = Impact::Penalty;
    clash->a_source = clash->b_source = ChangeSource::descriptor(0);
    clash->a_count = clash->b_count = [](const Solution& s) { return s.classes[0].n; };
    clash->key_a = clash->key_b = [](const Solution&, size_t) { return (int64_t)0; };
    clash->filter = [sf](const Solution& s, size_t a, size_t b) {
        if (!(a < b) || sf->day[a] != sf->day[b]) return false;
        int64_t na = s.classes[0].vars[0][a];
        return na != NONE && na == s.classes[0].vars[0][b];
    };
    clash->weight = [](const Solution&, size_t, size_t) { return Score::of(1, 0); };
    m->director.constraints.members.push_back(std::move(clash));

    auto streak = std::make_unique<GroupedRunsConstraint>();
    streak->name = "Long work streaks";
    streak->impact = Impact::Penalty;
    streak->source = ChangeSource::descriptor(0);
    streak->count = [](const Solution& s) { return s.classes[0].n; };
    streak->filter = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i] != NONE; };
    streak->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    streak->point = [sf](const Solution&, size_t i) { return sf->day[i]; };
    streak->weight = [limit, w_streak](int64_t, const Runs& runs) {
        int64_t excess = 0;
        for (auto& r : runs.runs) excess += (int64_t)r.point_count > limit ? (int64_t)r.point_count - limit : 0;  // saturating_sub
        return Score::of(0, w_streak * excess);
    };
    if (presence_lo >= 0) {  // group_by(nurse, indexed_presence(day)).penalize(w * min(count_in(lo..hi), cap)); cap 0 = uncapped, 1 = any_in
        streak->name = "Days worked in the window";
        streak->presence_weight = [presence_lo, presence_hi, presence_cap, presence_mode, w_streak](int64_t, const IndexedPresenceAccumulator& p) {
            if (presence_mode == 1) {  // "consecutive off bounds": complement_runs(lo..hi), each run's point_count.saturating_sub(cap)
                int64_t excess = 0;
                for (auto& r : p.complement_runs(presence_lo, presence_hi).runs)
                    excess += (int64_t)r.point_count > presence_cap ? (int64_t)r.point_count - presence_cap : 0;
                return Score::of(0, w_streak * excess);
            }
            int64_t c = (int64_t)p.count_in(presence_lo, presence_hi);
            if (presence_cap == 1) c = p.any_in(presence_lo, presence_hi) ? 1 : 0;
            else if (presence_cap > 0 && c > presence_cap) c = presence_cap;
            return Score::of(0, w_streak * c);
        };
    }
    m->director.constraints.members.push_back(std::move(streak));

    if (count_weight > 0 && target >= 0) {  // balanced workload (schedule.rs:61-74): count per nurse, complemented by the nurses
        auto bal = std::make_unique<ComplementedGroupedConstraint>();
        bal->name = "Balanced workload";
        bal->impact = Impact::Penalty;
        bal->a_source = ChangeSource::descriptor(0);
        bal->b_source = ChangeSource::fixed();  // nurses are problem facts
        bal->a_count = [](const Solution& s) { return s.classes[0].n; };
        bal->b_count = [n_nurses](const Solution&) { return n_nurses; };
        bal->key_a = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        bal->key_b = [](const Solution&, size_t b) { return (int64_t)b; };
        bal->value = [](const Solution&, size_t) { return (int64_t)1; };
        bal->default_b = [](const Solution&, size_t) { return (int64_t)0; };
        bal->weight = [count_weight, target](int64_t, int64_t c) { return Score::of(0, count_weight * (c > target ? c - target : target - c)); };
        m->director.constraints.members.push_back(std::move(bal));
    } else if (count_weight > 0) {
        auto load = std::make_unique<GroupedConstraint>();
        load->name = "Workload";
        load->impact = Impact::Penalty;
        load->source = ChangeSource::descriptor(0);
        load->count = [](const Solution& s) { return s.classes[0].n; };
        load->filter = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i] != NONE; };
        load->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        load->value = [](const Solution&, size_t) { return (int64_t)1; };
        load->weight = [count_weight](int64_t, int64_t c) { return Score::of(0, count_weight * c * c); };
        m->director.constraints.members.push_back(std::move(load));
    }
    m->has_scalar = true;
    m->scalar_slot.descriptor_index = 0;
    m->scalar_slot.variable_index = 0;
    m->scalar_slot.allows_unassigned = true;
    m->scalar_slot.values_for_entity = [n_nurses](const Solution&, size_t, std::vector<int64_t>& out) {
        out.clear();
        for (size_t v = 0; v < n_nurses; ++v) out.push_back((int64_t)v);
    };
    m->leaves = LEAF_SCALAR_CHANGE | LEAF_SCALAR_SWAP;
    m->wire_search();
    return m;
}

}  // namespace sfo
