// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/, smoke() and
// bench.py's cpu_baseline leg.  The product path (solverforge_amd/) never links this.
#include <chrono>
#include <cstring>

#include "sfo_clarke_wright.hpp"
#include "sfo_models.hpp"

using namespace sfo;

extern "C" {

// Wire format of one candidate move; identical to sf_move_t in include/solverforge_amd.h.
struct sfo_move_t {
    int32_t kind;   // 0 Change, 1 Swap, 2 ListChange, 3 ListSwap
    int32_t a;      // Change: entity; Swap: left entity; List*: source / first entity
    int32_t a_pos;  // List*: source / first position
    int32_t b;      // Swap: right entity; List*: destination / second entity
    int32_t b_pos;  // List*: destination / second position (pre-removal coords for ListChange)
    int32_t value;  // Change: to_value (-1 = None)
};

static void to_wire(const Move& m, sfo_move_t* w) {
    w->kind = (int32_t)m.kind;
    w->a = (int32_t)m.a;
    w->a_pos = (int32_t)m.a_pos;
    w->b = (int32_t)m.b;
    w->b_pos = (int32_t)m.b_pos;
    w->value = (int32_t)m.to_value;
    if (m.kind == Move::Ruin) {  // a = list, a_pos = count, six 16-bit positions in b / b_pos / value
        w->b = (int32_t)((uint32_t)m.ruin_idx[0] | ((uint32_t)m.ruin_idx[1] << 16));
        w->b_pos = (int32_t)((uint32_t)m.ruin_idx[2] | ((uint32_t)m.ruin_idx[3] << 16));
        w->value = (int32_t)((uint32_t)m.ruin_idx[4] | ((uint32_t)m.ruin_idx[5] << 16));
        if (m.prec && m.a_pos <= 5) {  // the move carries precedence hooks (critical-path leaf, or a ruin leaf of a slot with the policy; <= 5 elements): bit 31 = precedence hooks, bit 30 = two source lists, the second in bits 16..29
            uint32_t flags = 0x8000u;
            if (m.ruin_multi) flags |= 0x4000u | (uint32_t)m.ruin_src[m.a_pos - 1];
            w->value = (int32_t)((uint32_t)m.ruin_idx[4] | (flags << 16));
        }
    }
    if (m.kind == Move::MultiSwap) {  // a = swaps, (list | first << 16) per swap, value = second - first, one byte per swap
        int32_t* slot[3] = {&w->a_pos, &w->b, &w->b_pos};
        w->a = (int32_t)m.a_pos;
        w->value = 0;
        for (size_t i = 0; i < 3; ++i) {
            *slot[i] = i < m.a_pos ? (int32_t)((uint32_t)m.ms_entity[i] | ((uint32_t)m.ms_first[i] << 16)) : 0;
            if (i < m.a_pos) w->value |= (int32_t)(((uint32_t)(m.ms_second[i] - m.ms_first[i]) & 0xFFu) << (8 * i));
        }
    }
}
static Move from_wire(const Model& model, const sfo_move_t& w) {
    Move m;
    m.kind = (Move::Kind)w.kind;
    m.a = (size_t)w.a;
    m.a_pos = (size_t)w.a_pos;
    m.b = (size_t)w.b;
    m.b_pos = (size_t)w.b_pos;
    m.to_value = w.value;
    if (m.kind == Move::Change || m.kind == Move::Swap) {
        m.descriptor = model.scalar_slot.descriptor_index;
        m.variable = model.scalar_slot.variable_index;
        m.allows_unassigned = model.scalar_slot.allows_unassigned;
    } else {
        m.descriptor = model.list_slot.descriptor_index;
    }
    if (m.kind == Move::Ruin) {
        const uint32_t w3[3] = {(uint32_t)w.b, (uint32_t)w.b_pos, (uint32_t)w.value};
        for (int i = 0; i < 6; ++i) m.ruin_idx[i] = (uint16_t)(w3[i / 2] >> (16 * (i & 1)));
        m.b = m.a;
        m.b_pos = 0;
        m.to_value = NONE;
        m.allows_unassigned = model.ruin_skip_empty;
        if (model.list_slot.precedence_policy) m.prec = model.list_slot.precedence.get();  // the slot's ruin leaf recreates with the hooks
        if (m.a_pos <= 5 && ((uint32_t)w.value & 0x80000000u)) {  // a ruin that says so itself (critical-path leaf: no skip_empty)
            const uint32_t flags = (uint32_t)w.value >> 16;
            m.ruin_idx[5] = 0;
            m.prec = model.list_slot.precedence.get();
            if (!model.list_slot.precedence_policy || (flags & 0x4000u)) m.allows_unassigned = false;
            for (size_t i = 0; i < 8; ++i) m.ruin_src[i] = (uint16_t)m.a;
            if (flags & 0x4000u) {
                m.ruin_multi = true;
                m.ruin_src[m.a_pos - 1] = (uint16_t)(flags & 0x3FFFu);
            }
        }
    }
    if (m.kind == Move::MultiSwap) {
        const uint32_t slot[3] = {(uint32_t)w.a_pos, (uint32_t)w.b, (uint32_t)w.b_pos};
        m.a_pos = (size_t)w.a;
        m.require_improvement = true;
        for (size_t i = 0; i < m.a_pos && i < 3; ++i) {
            m.ms_entity[i] = (uint16_t)(slot[i] & 0xFFFFu);
            m.ms_first[i] = (uint16_t)(slot[i] >> 16);
            m.ms_second[i] = (uint16_t)(m.ms_first[i] + (int8_t)(((uint32_t)w.value >> (8 * i)) & 0xFFu));
        }
        m.a = m.b = m.ms_entity[0];
        m.b_pos = 0;
        m.to_value = NONE;
    }
    return m;
}

// ---- stream context -----------------------------------------------------------
uint64_t sfo_splitmix64(uint64_t v) { return splitmix64(v); }
uint64_t sfo_step_seed(uint64_t random_seed, uint64_t draw) { return sf_step_seed(random_seed, draw); }
static MoveStreamContext mk_ctx(uint64_t idx, uint64_t seed, int32_t order) {
    return MoveStreamContext(idx, seed).with_selection_order((SelectionOrder)order);
}
uint64_t sfo_ctx_mixed_seed(uint64_t idx, uint64_t seed, uint64_t salt) { return mk_ctx(idx, seed, 3).mixed_seed(salt); }
uint64_t sfo_ctx_random_index(uint64_t idx, uint64_t seed, uint64_t len, uint64_t salt) {
    return mk_ctx(idx, seed, 3).random_index(len, salt);
}
uint64_t sfo_ctx_random_stride(uint64_t idx, uint64_t seed, uint64_t len, uint64_t salt) {
    return mk_ctx(idx, seed, 3).random_stride(len, salt);
}
uint64_t sfo_ctx_selection_index(uint64_t idx, uint64_t seed, int32_t order, uint64_t offset, uint64_t len,
                                 uint64_t salt) {
    return mk_ctx(idx, seed, order).selection_index(offset, len, salt);
}
uint64_t sfo_ctx_selection_index_wo(uint64_t idx, uint64_t seed, int32_t order, uint64_t offset, uint64_t len,
                                    uint64_t salt) {
    return mk_ctx(idx, seed, order).selection_index_without_replacement(offset, len, salt);
}
int32_t sfo_reservoir_pick(uint64_t step_seed, uint64_t equal_count) {
    return reservoir_pick(step_seed, equal_count) ? 1 : 0;
}
// stable bounded top-k over (distance) with ordinal payload; returns kept count
int32_t sfo_sort_and_limit(const double* dist, int32_t n, int32_t max_nearby, int32_t* out_idx) {
    std::vector<NearbyCandidate> c;
    for (int32_t i = 0; i < n; ++i) c.push_back({(size_t)i, (size_t)i, dist[i]});
    sort_and_limit_nearby_candidates(c, (size_t)max_nearby);
    for (size_t i = 0; i < c.size(); ++i) out_idx[i] = (int32_t)c[i].entity;
    return (int32_t)c.size();
}

// ---- models -------------------------------------------------------------------
void* sfo_nqueens_create(int32_t n, const int64_t* rows) { return make_nqueens((size_t)n, rows).release(); }
void* sfo_graph_coloring_create(int32_t n, int32_t n_colors, const uint32_t* adj_off, const uint32_t* adj,
                                const int64_t* colors) {
    return make_graph_coloring((size_t)n, (size_t)n_colors, adj_off, adj, colors).release();
}
// INDEXED CPU baselines (partner-indexed predicate join instead of the reference's dense one): same scores, O(partners) per insert
void* sfo_graph_coloring_create_indexed(int32_t n, int32_t n_colors, const uint32_t* adj_off, const uint32_t* adj, const int64_t* colors) {
    return make_graph_coloring((size_t)n, (size_t)n_colors, adj_off, adj, colors, true).release();
}
void* sfo_jobshop_create_indexed(int32_t n_ops, int32_t n_machines, const int64_t* job, const int64_t* machine_idx, const uint32_t* seq_off,
                                 const uint32_t* seq_vals, int32_t bendable) {
    return make_jobshop((size_t)n_ops, (size_t)n_machines, job, machine_idx, seq_off, seq_vals, bendable != 0, true).release();
}
void* sfo_balance_create(int32_t n, int32_t n_bins, const int64_t* bins, const int64_t* sizes, int64_t w_pair,
                         int64_t cap) {
    return make_balance((size_t)n, (size_t)n_bins, bins, sizes, w_pair, cap).release();
}
void* sfo_balance_create_nary(int32_t n, int32_t n_bins, const int64_t* bins, const int64_t* sizes, int64_t w_tuple,
                              int64_t cap, int32_t arity) {
    return make_balance((size_t)n, (size_t)n_bins, bins, sizes, w_tuple, cap, (size_t)arity).release();
}
void* sfo_balance_create_base(int32_t n, int32_t n_bins, const int64_t* bins, const int64_t* sizes, int64_t w_pair, int64_t balance_base) {
    return make_balance((size_t)n, (size_t)n_bins, bins, sizes, w_pair, -3, 2, balance_base).release();  // BalanceConstraint with a chosen base score
}
void* sfo_assignment_create(int32_t n, int32_t n_values, const int64_t* values, const int64_t* cost, int64_t cost_weight, const int64_t* row_w,
                            int32_t ex_mode, int32_t ex_level, int64_t ex_weight) {
    return make_assignment((size_t)n, (size_t)n_values, values, cost, cost_weight, row_w, ex_mode, ex_level, ex_weight).release();
}
void* sfo_assignment_create2(int32_t n, int32_t n_values, const int64_t* values, const int64_t* cost, int64_t cost_weight, const int64_t* row_w,
                             int32_t ex_mode, int32_t ex_level, int64_t ex_weight, const int64_t* cost2, int32_t cost2_level) {
    return make_assignment((size_t)n, (size_t)n_values, values, cost, cost_weight, row_w, ex_mode, ex_level, ex_weight, cost2, cost2_level).release();
}
void* sfo_cvrp_create(int32_t n_customers, int32_t n_vehicles, int64_t capacity, int32_t depot, int32_t dim,
                      const int32_t* demands, const int64_t* matrix, const uint32_t* customers,
                      const uint32_t* route_off, const uint32_t* route_vals) {
    return make_cvrp((size_t)n_customers, (size_t)n_vehicles, capacity, (size_t)depot, (size_t)dim, demands,
                     matrix, customers, route_off, route_vals)
        .release();
}
void* sfo_precedence_shop_create(int32_t n_nodes, int32_t n_owners, const int64_t* duration, const uint32_t* succ_off, const uint32_t* succ,
                                 const int64_t* expected_owner, const uint32_t* list_off, const uint32_t* list_vals, int32_t levels,
                                 int32_t hard_levels, int32_t hard_level, int32_t soft_level) {
    return make_precedence_shop((size_t)n_nodes, (size_t)n_owners, duration, succ_off, succ, expected_owner, list_off, list_vals, levels,
                                hard_levels, hard_level, soft_level)
        .release();
}
void* sfo_shift_schedule_create(int32_t n_shifts, int32_t n_nurses, const int64_t* nurse_idx, const int64_t* day, int64_t limit, int64_t w_streak,
                                int64_t count_weight, int64_t target, const int64_t* required) {
    return make_shift_schedule((size_t)n_shifts, (size_t)n_nurses, nurse_idx, day, limit, w_streak, count_weight, target, required).release();
}
void* sfo_shift_schedule_create_presence(int32_t n_shifts, int32_t n_nurses, const int64_t* nurse_idx, const int64_t* day, int64_t lo, int64_t hi,
                                         int64_t cap, int64_t mode, int64_t w, int64_t count_weight, int64_t target, const int64_t* required) {
    return make_shift_schedule((size_t)n_shifts, (size_t)n_nurses, nurse_idx, day, 0, w, count_weight, target, required, lo, hi, cap, mode).release();
}
void* sfo_list_toy_create(int32_t n_entities, const uint32_t* off, const uint32_t* vals, int32_t meter) {
    return make_list_toy((size_t)n_entities, off, vals, meter == 0 ? ToyMeter::Equal : ToyMeter::Position)
        .release();
}
void* sfo_jobshop_create_makespan(int32_t n_ops, int32_t n_machines, const int64_t* job, const int64_t* machine_idx, const uint32_t* seq_off,
                                  const uint32_t* seq_vals, int32_t bendable, int32_t indexed, const int64_t* duration) {
    return make_jobshop((size_t)n_ops, (size_t)n_machines, job, machine_idx, seq_off, seq_vals, bendable != 0, indexed != 0, duration).release();
}
void* sfo_jobshop_create_owner_match(int32_t n_ops, int32_t n_machines, const int64_t* job, const int64_t* machine_idx, const uint32_t* seq_off,
                                     const uint32_t* seq_vals, int32_t bendable, int32_t owner_match_level) {
    return make_jobshop((size_t)n_ops, (size_t)n_machines, job, machine_idx, seq_off, seq_vals, bendable != 0, false, nullptr, owner_match_level).release();
}
void* sfo_jobshop_create(int32_t n_ops, int32_t n_machines, const int64_t* job, const int64_t* machine_idx,
                         const uint32_t* seq_off, const uint32_t* seq_vals, int32_t bendable) {
    return make_jobshop((size_t)n_ops, (size_t)n_machines, job, machine_idx, seq_off, seq_vals, bendable != 0)
        .release();
}
void sfo_model_destroy(void* h) { delete (Model*)h; }

void sfo_model_score(void* h, int64_t* out4) {  // Director::calculate_score
    Score s = ((Model*)h)->director.calculate_score();
    std::memcpy(out4, s.v, sizeof(s.v));
}
void sfo_model_fresh_score(void* h, int64_t* out4) {  // Director::fresh_score (evaluate_all)
    Score s = ((Model*)h)->director.fresh_score();
    std::memcpy(out4, s.v, sizeof(s.v));
}
void sfo_model_reset(void* h) { ((Model*)h)->director.reset(); }
// ConstraintSet::evaluate_each (api/constraint_set/incremental.rs:172,237-244): per constraint, in declaration
// order, its score on the working solution and its match count.  Returns the number of constraints.
int32_t sfo_model_evaluate_each(void* h, int64_t* out_scores4, int64_t* out_counts, int32_t cap) {
    Model* m = (Model*)h;
    int32_t n = 0;
    for (auto& c : m->director.constraints.members) {
        if (n < cap) {
            Score s = c->evaluate(m->director.working);
            std::memcpy(out_scores4 + (size_t)n * 4, s.v, sizeof(s.v));
            out_counts[n] = (int64_t)c->match_count(m->director.working);
        }
        ++n;
    }
    return n;
}

// acceptor: 0 HillClimbing, 1 LateAcceptance(size).  forager: 0 AcceptedCount(limit), 1 FirstAccepted, 2 BestScore,
// 3 FirstBestScoreImproving, 4 FirstLastStepScoreImproving(limit; <= 0 = None).
// union_order: 0 Sequential 1 RoundRobin 2 RotatingRoundRobin 3 Random 4 StratifiedRandom; -1 = default policy
void sfo_model_configure(void* h, int32_t acceptor, int32_t la_size, int32_t forager, int32_t limit,
                         int32_t random_ties, int32_t selection_order, uint32_t leaves, int32_t max_nearby,
                         int32_t union_order, uint64_t random_seed, int32_t public_entity_order) {
    Model* m = (Model*)h;
    if (acceptor == 0)
        m->search.acceptor = std::make_unique<HillClimbingAcceptor>();
    else
        m->search.acceptor = std::make_unique<LateAcceptanceAcceptor>((size_t)la_size);
    m->search.forager.kind = (Forager::Kind)forager;
    m->search.forager.accepted_count_limit = limit > 0 ? (size_t)limit : 0;  // 0 = None (FirstLastStepScoreImproving)
    m->search.forager.best.random_ties = random_ties != 0;
    m->search.selection_order = (SelectionOrder)selection_order;
    m->search.random_seed = random_seed;
    m->leaves = leaves;
    m->max_nearby = (size_t)max_nearby;
    m->list_slot.public_selector_entity_order = public_entity_order != 0;
    m->seed_ruin_stream(random_seed);
    size_t n_leaves = (size_t)__builtin_popcount(leaves);
    if (union_order < 0)
        m->union_order = n_leaves <= 1 ? UnionOrder::Sequential : UnionOrder::StratifiedRandom;
    else
        m->union_order = (UnionOrder)union_order;
}
// SimulatedAnnealingAcceptor (simulated_annealing.rs): mode 0 Single(temps[0]), 1 PerLevel(temps[0..levels]),
// 2 Calibrated(sample_size, target, fallback).  Replaces the acceptor sfo_model_configure installed.
void sfo_model_configure_annealing(void* h, int32_t mode, const double* temps, int32_t levels, int32_t hard_levels,
                                   double decay_rate, double hill_climbing_temperature, int32_t never_accept_hard,
                                   int32_t sample_size, double target_probability, double fallback_temperature,
                                   uint64_t seed) {
    Model* m = (Model*)h;
    auto sa = std::make_unique<SimulatedAnnealingAcceptor>();
    sa->mode = (SimulatedAnnealingAcceptor::Mode)mode;
    sa->levels = levels;
    sa->hard_levels = hard_levels;
    if (mode == 0) sa->single_temperature = temps[0];
    if (mode == 1) sa->level_temperatures.assign(temps, temps + levels);
    sa->decay_rate = decay_rate;
    sa->hill_climbing_temperature = hill_climbing_temperature;
    sa->never_accept_hard_regression = never_accept_hard != 0;
    sa->sample_size = (size_t)sample_size;
    sa->target_acceptance_probability = target_probability;
    sa->fallback_temperature = fallback_temperature;
    sa->rng = SmallRng::seed_from_u64(seed);
    m->search.acceptor = std::move(sa);
}
// Nearby scalar sources of the model's scalar slot (scalar_access.rs:261-340) as data: which = 0 nearby VALUE candidates per
// entity, 1 nearby ENTITY candidates per left entity; rows in source order (CSR), `dist` = the distance meter's value per row
// entry or NULL (meter None: the source order ranks).  dynamic != 0: a DynamicScalarVariableSlot (legality re-check, directional
// swaps).  max_nearby / source_limit (0 = none) of the two nearby leaves.
void sfo_model_set_nearby_scalar(void* h, int32_t which, int32_t n_rows, const uint32_t* off, const int64_t* cand, const double* dist,
                                 int32_t dynamic, int32_t max_nearby, int64_t source_limit) {
    Model* m = (Model*)h;
    ScalarSlot& sl = m->scalar_slot;
    sl.dynamic = dynamic != 0;
    m->scalar_max_nearby = (size_t)max_nearby;
    m->scalar_source_limit = source_limit > 0 ? (size_t)source_limit : SIZE_MAX;
    std::vector<std::vector<int64_t>> rows((size_t)n_rows);
    auto table = std::make_shared<std::vector<std::vector<std::pair<int64_t, double>>>>((size_t)n_rows);
    for (int32_t r = 0; r < n_rows; ++r)
        for (uint32_t k = off[r]; k < off[r + 1]; ++k) {
            rows[(size_t)r].push_back(cand[k]);
            if (dist) (*table)[(size_t)r].push_back({cand[k], dist[k]});
        }
    auto meter = [table](size_t row, int64_t c) -> double {
        for (auto& kv : (*table)[row])
            if (kv.first == c) return kv.second;
        return std::nan("");
    };
    if (which == 0) {
        sl.has_nearby_values = true;
        sl.nearby_values = std::move(rows);
        if (dist) sl.nearby_value_distance = meter;
    } else {
        sl.has_nearby_entities = true;
        sl.nearby_entities = std::move(rows);
        if (dist) sl.nearby_entity_distance = [meter](size_t l, size_t r) { return meter(l, (int64_t)r); };
    }
}
// DiversifiedLateAcceptanceAcceptor(late_acceptance_size, tolerance).  Replaces the acceptor sfo_model_configure installed.
void sfo_model_configure_diversified(void* h, int32_t la_size, double tolerance) {
    ((Model*)h)->search.acceptor = std::make_unique<DiversifiedLateAcceptanceAcceptor>((size_t)la_size, tolerance);
}
// acceptor state after the steps run so far: temperatures[levels], rng state[4], calibrating flag
void sfo_model_annealing_state(void* h, double* out_temps, uint64_t* out_rng, int32_t* out_calibrating) {
    auto* sa = dynamic_cast<SimulatedAnnealingAcceptor*>(((Model*)h)->search.acceptor.get());
    if (!sa) return;
    for (size_t k = 0; k < sa->current.size(); ++k) out_temps[k] = sa->current[k];
    for (int i = 0; i < 4; ++i) out_rng[i] = sa->rng.s[i];
    *out_calibrating = sa->calibrating ? 1 : 0;
}
// xoshiro256++ from an explicit state: n outputs (known-answer check of the SmallRng restatement)
void sfo_xoshiro256pp(const uint64_t* state4, int32_t n, uint64_t* out) {
    SmallRng r;
    for (int i = 0; i < 4; ++i) r.s[i] = state4[i];
    for (int32_t i = 0; i < n; ++i) out[i] = r.next_u64();
}
void sfo_small_rng_seed(uint64_t seed, uint64_t* out_state4) {
    SmallRng r = SmallRng::seed_from_u64(seed);
    for (int i = 0; i < 4; ++i) out_state4[i] = r.s[i];
}
void sfo_model_set_sublist_sizes(void* h, int32_t min_size, int32_t max_size) {
    Model* m = (Model*)h;
    m->sublist_min = (size_t)min_size;
    m->sublist_max = (size_t)max_size;
}
void sfo_model_set_permute(void* h, int32_t min_window_size, int32_t max_window_size) {  // ListPermuteMoveConfig (defaults 2..=5)
    ((Model*)h)->permute_min = (size_t)min_window_size;
    ((Model*)h)->permute_max = (size_t)max_window_size;
}
// the list slot declares its precedence hooks to the compiled runtime leaves (route-graph filter of intra-list candidates, ruins with hooks)
void sfo_model_set_precedence_policy(void* h, int32_t on) { ((Model*)h)->list_slot.precedence_policy = on != 0; }
void sfo_model_set_kopt(void* h, int32_t min_segment_len, int32_t max_nearby) {  // max_nearby 0 = full enumeration
    Model* m = (Model*)h;
    m->kopt_min_seg = (size_t)min_segment_len;
    m->kopt_max_nearby = (size_t)max_nearby;
}
// ListRuinMoveSelectorConfig (solverforge-config/src/move_selector.rs:552-587); max_source_list_len 0 = None.  Re-seeds the
// leaf's per-solve stream: scoped_seed(random_seed, descriptor, variable_name, "list_ruin_move_selector").
void sfo_model_set_ruin(void* h, int32_t min_count, int32_t max_count, int32_t moves_per_step, int32_t max_source_list_len,
                        int32_t skip_empty_destinations, const char* variable_name) {
    Model* m = (Model*)h;
    m->ruin_min = (size_t)min_count;
    m->ruin_max = (size_t)max_count;
    m->ruin_moves_per_step = (size_t)moves_per_step;
    m->ruin_max_source_len = (size_t)max_source_list_len;
    m->ruin_skip_empty = skip_empty_destinations != 0;
    if (variable_name) m->list_variable_name = variable_name;
    m->seed_ruin_stream(m->search.random_seed);
}
uint64_t sfo_scoped_seed(uint64_t base_seed, uint64_t descriptor_index, const char* variable_name, const char* selector_kind) {
    return scoped_seed(base_seed, (size_t)descriptor_index, variable_name, selector_kind);
}
uint64_t sfo_hash_str(const char* s) { return hash_str(s); }
uint64_t sfo_siphash(int32_t c, int32_t d, uint64_t k0, uint64_t k1, const uint8_t* data, int64_t len) {
    return siphash(c, d, k0, k1, data, (size_t)len);
}
// n draws of SmallRng::seed_from_u64(seed).random_range(low..=high) (known-answer hook for the device restatement)
void sfo_random_range_stream(uint64_t seed, uint64_t low, uint64_t high_inclusive, int32_t n, uint64_t* out) {
    SmallRng r = SmallRng::seed_from_u64(seed);
    for (int32_t i = 0; i < n; ++i) out[i] = r.random_range_inclusive(low, high_inclusive);
}
// UnionWeighting of the root union: one weight per leaf in union order (n = 0: equal)
void sfo_model_set_union_weights(void* h, const uint64_t* weights, int32_t n) {
    ((Model*)h)->union_weights.assign(weights, weights + n);
}
// ValueSource::EntitySlice for the scalar slot: entity e draws from values[off[e] .. off[e + 1])
void sfo_model_set_value_lists(void* h, const uint32_t* off, const int64_t* values, int32_t n) {
    Model* m = (Model*)h;
    auto o = std::make_shared<std::vector<uint32_t>>(off, off + n + 1);
    auto v = std::make_shared<std::vector<int64_t>>(values, values + off[n]);
    m->scalar_slot.values_for_entity = [o, v](const Solution&, size_t e, std::vector<int64_t>& out) {
        out.assign(v->begin() + (*o)[e], v->begin() + (*o)[e + 1]);
    };
}
void sfo_model_set_step_seeds(void* h, const uint64_t* seeds, int32_t n) {
    ((Model*)h)->search.explicit_step_seeds.assign(seeds, seeds + n);
}
void sfo_model_phase_start(void* h) { ((Model*)h)->search.phase_start(); }
void sfo_model_steps(void* h, int64_t n) {
    Model* m = (Model*)h;
    for (int64_t i = 0; i < n; ++i) m->search.step();
}
// Runs steps until `seconds` elapsed (checked per step); returns steps executed.
int64_t sfo_model_steps_timed(void* h, double seconds) {
    Model* m = (Model*)h;
    auto t0 = std::chrono::steady_clock::now();
    int64_t n = 0;
    for (;;) {
        m->search.step();
        ++n;
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el >= seconds) break;
    }
    return n;
}
// One traced step: every pulled candidate with doable/score/accepted; returns count (<= cap recorded).
int32_t sfo_model_step_traced(void* h, sfo_move_t* moves, int64_t* scores4, int32_t* flags, int32_t cap,
                              int32_t* applied, sfo_move_t* applied_move) {
    Model* m = (Model*)h;
    std::vector<StepTrace> tr;
    m->search.trace = &tr;
    m->search.step();
    m->search.trace = nullptr;
    int32_t n = (int32_t)tr.size();
    for (int32_t i = 0; i < n && i < cap; ++i) {
        to_wire(tr[i].move, &moves[i]);
        std::memcpy(&scores4[4 * i], tr[i].score.v, 4 * sizeof(int64_t));
        flags[i] = (tr[i].doable ? 1 : 0) | (tr[i].accepted ? 2 : 0) | (tr[i].selected ? 4 : 0) | tr[i].gate | ((int32_t)tr[i].selector << 8);
    }
    *applied = m->search.last_step_applied ? 1 : 0;
    if (m->search.last_step_applied) to_wire(m->search.last_applied_move, applied_move);
    return n;
}
void sfo_model_stats(void* h, uint64_t* out8) {
    const SolverStats& s = ((Model*)h)->search.stats;
    out8[0] = s.step_count;
    out8[1] = s.moves_generated;
    out8[2] = s.moves_evaluated;
    out8[3] = s.moves_accepted;
    out8[4] = s.moves_applied;
    out8[5] = s.score_calculations;
    out8[6] = s.moves_not_doable;
    out8[7] = 0;
}
void sfo_model_last_step_score(void* h, int64_t* out4) {
    std::memcpy(out4, ((Model*)h)->search.last_step_score.v, 4 * sizeof(int64_t));
}
void sfo_model_best_score(void* h, int64_t* out4) {
    std::memcpy(out4, ((Model*)h)->search.best_score.v, 4 * sizeof(int64_t));
}

// Enumerate the candidate stream of one leaf (leaf = single LeafBits bit) or of the configured
// union (leaf = 0) for MoveStreamContext(step_index, step_seed).with_selection_order(order).
int64_t sfo_model_enumerate(void* h, uint32_t leaf, uint64_t step_index, uint64_t step_seed, int32_t order,
                            sfo_move_t* out, int64_t cap) {
    Model* m = (Model*)h;
    MoveStreamContext ctx = mk_ctx(step_index, step_seed, order);
    std::unique_ptr<Cursor> cur = leaf == 0 ? m->open_union(m->director, ctx) : m->open_leaf(leaf, m->director, ctx);
    int64_t n = 0;
    Move mv;
    while (cur->next(mv)) {
        if (n < cap && out) to_wire(mv, &out[n]);
        ++n;
    }
    return n;
}
// FNV-1a style order hash + count of a leaf stream (benches/selector_cursor_gate.rs mix()).
int64_t sfo_model_enumerate_count(void* h, uint32_t leaf, uint64_t step_index, uint64_t step_seed,
                                  int32_t order) {
    return sfo_model_enumerate(h, leaf, step_index, step_seed, order, nullptr, 0);
}

// n x evaluate_candidate (phase/localsearch/evaluation.rs:20-115): state unchanged.
void sfo_model_evaluate_moves(void* h, const sfo_move_t* moves, int64_t n, int64_t* scores4, int32_t* doable) {
    Model* m = (Model*)h;
    m->director.calculate_score();
    for (int64_t i = 0; i < n; ++i) {
        Move mv = from_wire(*m, moves[i]);
        if (!move_is_doable(m->director, mv)) {
            doable[i] = 0;
            std::memset(&scores4[4 * i], 0, 4 * sizeof(int64_t));
            continue;
        }
        doable[i] = 1;
        DirectorScoreState st = m->director.snapshot_score_state();
        MoveUndo u = move_do(m->director, mv);
        Score sc = m->director.calculate_score();
        move_undo(m->director, mv, u);
        m->director.restore_score_state(st);
        std::memcpy(&scores4[4 * i], sc.v, 4 * sizeof(int64_t));
    }
}
// ScalarCandidates (planning/scalar/candidate.rs:85-188) as CompoundScalarMoves: candidate i = edits[offsets[i] .. offsets[i + 1]),
// every edit a Change-shaped sfo_move_t (a = entity, value = to_value).  n x evaluate_candidate, state unchanged.
static std::vector<ScalarEditO> edits_of(const Model& m, const sfo_move_t* edits, int64_t b, int64_t e) {
    std::vector<ScalarEditO> v;
    std::vector<int64_t> values;
    for (int64_t k = b; k < e; ++k) {
        const int64_t to = (int64_t)edits[k].value;
        bool legal = to == NONE ? m.scalar_slot.allows_unassigned : false;
        if (to != NONE && edits[k].a >= 0 && (size_t)edits[k].a < m.director.working.classes[m.scalar_slot.descriptor_index].n) {
            values.clear();
            m.scalar_slot.values_for_entity(m.director.working, (size_t)edits[k].a, values);
            legal = std::find(values.begin(), values.end(), to) != values.end();
        }
        v.push_back({m.scalar_slot.descriptor_index, m.scalar_slot.variable_index, (size_t)edits[k].a, to, legal});
    }
    return v;
}
// One local-search step over a ScalarCandidateProvider's output (GroupedScalarMoveSelector, grouped_scalar_step): returns the number
// of kept candidates; out_kept[n] provider indices in pull order, out_scores4 / out_flags per consumed candidate, *out_consumed,
// *out_selected = ordinal of the committed candidate or -1.
int64_t sfo_model_step_grouped_gated(void* h, const sfo_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n, int64_t group_name_len,
                                     int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_scores4, int32_t* out_flags, int64_t* out_consumed,
                                     int64_t* out_selected);
int64_t sfo_model_step_grouped(void* h, const sfo_move_t* edits, const int64_t* offsets, int64_t n, int64_t group_name_len, int64_t max_moves_per_step,
                               int64_t* out_kept, int64_t* out_scores4, int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected) {
    return sfo_model_step_grouped_gated(h, edits, offsets, nullptr, n, group_name_len, max_moves_per_step, out_kept, out_scores4, out_flags, out_consumed,
                                        out_selected);
}
int64_t sfo_model_step_grouped_gated(void* h, const sfo_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n, int64_t group_name_len,
                                     int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_scores4, int32_t* out_flags, int64_t* out_consumed,
                                     int64_t* out_selected) {
    Model* m = (Model*)h;
    std::vector<std::vector<ScalarEditO>> provided;
    for (int64_t i = 0; i < n; ++i) provided.push_back(edits_of(*m, edits, offsets[i], offsets[i + 1]));
    // group_name_len < 0: the candidates are a cursor's own pull order (RuntimeProviderCursor): no activation
    GroupedStepTrace t = grouped_scalar_step(m->search, provided, (size_t)(group_name_len < 0 ? 0 : group_name_len),
                                             max_moves_per_step > 0 ? (size_t)max_moves_per_step : 256,
                                             gates ? std::vector<int32_t>(gates, gates + n) : std::vector<int32_t>(), group_name_len < 0);
    for (size_t i = 0; i < t.kept.size(); ++i) out_kept[i] = (int64_t)t.kept[i];
    for (size_t i = 0; i < t.scores.size(); ++i) {
        std::memcpy(&out_scores4[4 * i], t.scores[i].v, 4 * sizeof(int64_t));
        out_flags[i] = t.flags[i];
    }
    *out_consumed = (int64_t)t.scores.size();
    *out_selected = t.selected;
    return (int64_t)t.kept.size();
}
void sfo_model_evaluate_compound(void* h, const sfo_move_t* edits, const int64_t* offsets, int64_t n, int64_t* scores4, int32_t* doable) {
    Model* m = (Model*)h;
    m->director.calculate_score();
    for (int64_t i = 0; i < n; ++i) {
        std::vector<ScalarEditO> ed = edits_of(*m, edits, offsets[i], offsets[i + 1]);
        if (!compound_is_doable(m->director, ed)) {
            doable[i] = 0;
            std::memset(&scores4[4 * i], 0, 4 * sizeof(int64_t));
            continue;
        }
        doable[i] = 1;
        DirectorScoreState st = m->director.snapshot_score_state();
        std::vector<int64_t> u = compound_do(m->director, ed);
        Score sc = m->director.calculate_score();
        compound_undo(m->director, ed, u);
        m->director.restore_score_state(st);
        std::memcpy(&scores4[4 * i], sc.v, 4 * sizeof(int64_t));
    }
}
void sfo_model_apply_compound(void* h, const sfo_move_t* edits, int64_t n_edits) {  // committed do_move of one candidate
    Model* m = (Model*)h;
    m->director.calculate_score();
    compound_do(m->director, edits_of(*m, edits, 0, n_edits));
    m->director.calculate_score();
}
void sfo_model_apply_move(void* h, const sfo_move_t* mv) {  // committed do_move
    Model* m = (Model*)h;
    m->director.calculate_score();
    Move mm = from_wire(*m, *mv);
    move_do(m->director, mm);
    m->director.calculate_score();
}
// ListCheapestInsertionPhase over the list class: `elements` = the unassigned elements in source order
void sfo_model_construct_list_cheapest(void* h, const uint32_t* elements, int32_t n) {
    Model* m = (Model*)h;
    // `elements` = the unassigned ones in source order; a slot with the precedence policy hands the phase its hooks (defaults/stages.rs:266-275)
    construct_list_cheapest(m->director, m->list_slot.descriptor_index, std::vector<uint32_t>(elements, elements + n), &m->search.stats,
                            m->list_slot.precedence_policy ? m->list_slot.precedence.get() : nullptr);
}
// ListRegretInsertionPhase (list_construction/regret.rs:223-260) without owner / order-key / precedence hooks
void sfo_model_construct_list_regret(void* h, const uint32_t* elements, int32_t n, const int64_t* order_keys, const int64_t* owners) {
    Model* m = (Model*)h;
    construct_list_regret(m->director, m->list_slot.descriptor_index, std::vector<uint32_t>(elements, elements + n), &m->search.stats,
                          order_keys ? std::vector<int64_t>(order_keys, order_keys + n) : std::vector<int64_t>(),
                          owners ? std::vector<int64_t>(owners, owners + n) : std::vector<int64_t>());
}
// ListClarkeWrightPhase over a CVRP model's list class (list_clarke_wright/kernel.rs) with the solverforge-cvrp hook bundle
// (crates/solverforge-cvrp/src/helpers.rs:40-87): one shared metric class (every vehicle shares the ProblemData), the model's depot,
// distance_cost legs; feasible_mode 0 = savings_hooks::feasible (structural only: capacity stays scoreable), 1 = the capacity
// test of route_hooks::feasible (helpers.rs:169-179).  `elements` = the unassigned elements in source order.  Returns 1 when
// routes were committed, 0 when the phase left the lists untouched; stats[0..5] = savings pairs, merge trials, merges, merge
// passes, completion trials.
int32_t sfo_model_construct_list_clarke_wright(void* h, const uint32_t* elements, int32_t n, int32_t feasible_mode, uint64_t* stats) {
    Model* m = (Model*)h;
    const CvrpFacts* cf = static_cast<const CvrpFacts*>(m->director.working.facts.get());
    const size_t desc = m->list_slot.descriptor_index;
    EntityClass& c = m->director.working.classes[desc];
    m->director.calculate_score();
    ClarkeWrightHooks hk;
    hk.entity_count = c.n;
    hk.source_values.assign(elements, elements + n);
    hk.route_len = [&c](size_t e) { return c.lists[e].size(); };
    hk.depot = [cf](size_t) { return cf->depot; };
    hk.metric_class = [](size_t) { return (size_t)0; };
    hk.distance = [cf](size_t, size_t a, size_t b) { return cf->distance_cost(a, b); };
    hk.feasible = [cf, feasible_mode](size_t, const std::vector<size_t>& route) {
        for (size_t v : route)
            if (v >= cf->dim) return false;
        if (feasible_mode == 0) return true;
        int64_t total = 0;
        for (size_t v : route)
            if (__builtin_add_overflow(total, (int64_t)cf->demands[v], &total)) return false;
        return total <= cf->capacity;
    };
    hk.replace_route = [m, desc, &c](size_t e, const std::vector<size_t>& route) {
        m->director.before_variable_changed(desc, e);
        c.lists[e].assign(route.begin(), route.end());
        m->director.after_variable_changed(desc, e);
    };
    std::vector<size_t> bound((size_t)n);
    for (size_t i = 0; i < (size_t)n; ++i) bound[i] = i;
    ClarkeWrightStats st;
    const bool committed = clarke_wright(hk, bound, &st);
    m->director.calculate_score();
    if (stats) stats[0] = st.savings_pairs, stats[1] = st.merge_trials, stats[2] = st.merges, stats[3] = st.merge_passes, stats[4] = st.completion_trials;
    return committed ? 1 : 0;
}
// Time windows / service durations / travel times of a CVRP model (solverforge-cvrp/src/problem_data.rs:20-23): lo / hi / service per node,
// travel dim x dim row-major.  Read by feasible_mode 2 of the list k-opt phase; nothing on the scoring path uses them (as in the stock crate).
void sfo_model_set_time_windows(void* h, const int64_t* lo, const int64_t* hi, const int64_t* service, const int64_t* travel, int64_t departure) {
    Model* m = (Model*)h;
    CvrpFacts* cf = const_cast<CvrpFacts*>(static_cast<const CvrpFacts*>(m->director.working.facts.get()));
    const size_t n = cf->dim;
    cf->tw_lo.assign(lo, lo + n), cf->tw_hi.assign(hi, hi + n), cf->service.assign(service, service + n);
    cf->travel.assign(travel, travel + n * n);
    cf->departure = departure;
}
// 1 = the route passes the stock route_hooks::feasible (capacity + time windows; an empty route always does)
int32_t sfo_model_route_feasible(void* h, const uint32_t* route, int32_t n) {
    Model* m = (Model*)h;
    const CvrpFacts* cf = static_cast<const CvrpFacts*>(m->director.working.facts.get());
    std::vector<size_t> r(route, route + n);
    return (r.empty() || (cf->capacity_feasible(r) && cf->time_feasible(r))) ? 1 : 0;
}
// ListKOptPhase (route-local 2-opt) over a CVRP model's list class with the stock route hooks (solverforge-cvrp/src/helpers.rs
// route_hooks: depot, distance_cost legs; feasible_mode 0 = no feasibility hook, 1 = the capacity test of route_hooks::feasible, 2 = the complete
// hook with the time windows of sfo_model_set_time_windows).
// stats[0..4] = candidates, accepted reversals, applied (committed) reversals, steps (changed routes).
void sfo_model_construct_list_k_opt(void* h, int32_t k, int32_t feasible_mode, int32_t max_sweeps, uint64_t* stats) {
    Model* m = (Model*)h;
    const CvrpFacts* cf = static_cast<const CvrpFacts*>(m->director.working.facts.get());
    const size_t desc = m->list_slot.descriptor_index;
    EntityClass& c = m->director.working.classes[desc];
    m->director.calculate_score();
    ListKOptHooks hk;
    hk.entity_count = c.n;
    hk.route_values = [&c](size_t e) { return std::vector<size_t>(c.lists[e].begin(), c.lists[e].end()); };
    hk.replace_route = [m, desc, &c](size_t e, const std::vector<size_t>& route) {
        m->director.before_variable_changed(desc, e);
        c.lists[e].assign(route.begin(), route.end());
        m->director.after_variable_changed(desc, e);
        m->director.calculate_score();
    };
    hk.depot = [cf](size_t) { return cf->depot; };
    hk.distance = [cf](size_t, size_t a, size_t b) { return cf->distance_cost(a, b); };
    if (feasible_mode == 1)
        hk.feasible = [cf](size_t, const std::vector<size_t>& route) {
            int64_t total = 0;
            for (size_t v : route)
                if (v >= cf->dim || __builtin_add_overflow(total, (int64_t)cf->demands[v], &total)) return false;
            return total <= cf->capacity;
        };
    if (feasible_mode == 2)  // the complete route_hooks::feasible: capacity + time windows (helpers.rs:109-119; sfo_model_set_time_windows)
        hk.feasible = [cf](size_t, const std::vector<size_t>& route) { return route.empty() || (cf->capacity_feasible(route) && cf->time_feasible(route)); };
    ListKOptStats st;
    list_k_opt(hk, (size_t)k, &st, (size_t)max_sweeps);
    m->director.calculate_score();
    SolverStats& ss = m->search.stats;
    ss.moves_generated += st.candidates, ss.moves_evaluated += st.candidates, ss.moves_accepted += st.accepted, ss.moves_applied += st.applied;
    ss.step_count += st.steps, ss.score_calculations += st.steps;
    if (stats) stats[0] = st.candidates, stats[1] = st.accepted, stats[2] = st.applied, stats[3] = st.steps;
}
// ListConstructionPhase (round robin); order_keys / owners parallel to elements, may be NULL
void sfo_model_construct_list_round_robin(void* h, const uint32_t* elements, int32_t n, const int64_t* order_keys, const int64_t* owners) {
    Model* m = (Model*)h;
    std::vector<int64_t> k, o;
    if (order_keys) k.assign(order_keys, order_keys + n);
    if (owners) o.assign(owners, owners + n);
    construct_list_round_robin(m->director, m->list_slot.descriptor_index, std::vector<uint32_t>(elements, elements + n), k, o, &m->search.stats);
}
void sfo_model_construct_first_fit(void* h) {
    Model* m = (Model*)h;
    construct_first_fit(m->director, m->scalar_slot, &m->search.stats);
}

int32_t sfo_model_get_vars(void* h, int32_t desc, int32_t var, int64_t* out) {
    const EntityClass& c = ((Model*)h)->director.working.classes[(size_t)desc];
    for (size_t i = 0; i < c.n; ++i) out[i] = c.vars[(size_t)var][i];
    return (int32_t)c.n;
}
// CSR download of a list variable; returns total element count.
int32_t sfo_model_get_lists(void* h, int32_t desc, uint32_t* off, uint32_t* vals) {
    const EntityClass& c = ((Model*)h)->director.working.classes[(size_t)desc];
    uint32_t t = 0;
    for (size_t e = 0; e < c.n; ++e) {
        off[e] = t;
        for (uint32_t v : c.lists[e]) vals[t++] = v;
    }
    off[c.n] = t;
    return (int32_t)t;
}

}  // extern "C"
