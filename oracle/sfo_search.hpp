// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// CPU restatement of the local-search step loop, acceptors, foragers, counters
// and first-fit construction.
//
// Follows (paths under crates/solverforge-solver/src/):
//   phase/localsearch/phase.rs:237-320           (phase loop)
//   phase/localsearch/phase/step.rs:30-225       (execute_step)
//   phase/localsearch/phase/candidates.rs:47-285 (evaluate_candidates)
//   phase/localsearch/evaluation.rs:20-115       (evaluate_candidate)
//   phase/localsearch/forager.rs:70-425          (BestCandidate / AcceptedCount / FirstAccepted / BestScore)
//   phase/localsearch/acceptor/hill_climbing.rs:33-41, late_acceptance.rs:89-125,
//   simulated_annealing.rs:11-430 (calibration :48-88, is_accepted :338-375, step_ended :417-430)
//   scope/solver/scope_progress.rs:89-107        (update_best_solution)
//   stats/solver.rs:23,112-119,246               (counter definitions)
//   phase/construction/forager_step.rs:149-226, decision.rs:56-64, evaluation.rs:6-18 (first fit)
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <vector>

#include "sfo_moves.hpp"

namespace sfo {

// Build-defined step-seed stream.  The reference draws step seeds from rand 0.10.1
// StdRng (ChaCha12; scope/solver/scope_core.rs:463-466, phase/step.rs:60-64) whose
// source is not in the tree => parity unpinned.  Both this oracle and the HIP path
// use this documented splitmix64 counter stream instead; parity tests may also pass
// explicit step_seed arrays.
inline uint64_t sf_step_seed(uint64_t random_seed, uint64_t draw_index) {
    return splitmix64(random_seed + draw_index * 0x9E3779B97F4A7C15ULL);
}

struct Acceptor {
    virtual ~Acceptor() = default;
    virtual bool is_accepted(const Score& last_step, const Score& move_score) = 0;
    virtual void phase_started(const Score&) {}
    virtual void step_started() {}
    virtual void step_ended(const Score&) {}
};

struct HillClimbingAcceptor : Acceptor {  // hill_climbing.rs:33-41
    bool is_accepted(const Score& last, const Score& mv) override { return mv > last; }
};

struct LateAcceptanceAcceptor : Acceptor {  // late_acceptance.rs:89-125
    size_t size;
    std::vector<Score> history;
    std::vector<bool> filled;
    size_t current = 0;
    explicit LateAcceptanceAcceptor(size_t n) : size(n), history(n), filled(n, false) {}
    bool is_accepted(const Score& last, const Score& mv) override {
        if (mv >= last) return true;
        if (filled[current]) return mv >= history[current];
        return true;
    }
    void phase_started(const Score& initial) override {
        for (size_t i = 0; i < size; ++i) {
            history[i] = initial;
            filled[i] = true;
        }
        current = 0;
    }
    void step_ended(const Score& step_score) override {
        history[current] = step_score;
        filled[current] = true;
        current = (current + 1) % size;
    }
};

// diversified_late_acceptance.rs:40-176 -- late acceptance plus "within tolerance of the best step score of this phase":
// threshold = best - |best|.multiply(tolerance), every level rounded half away from zero (score/macros.rs:61-71).
// The default acceptor of grouped scalar-only models (runtime/compiler/default_local_search/policy.rs:52-55).
struct DiversifiedLateAcceptanceAcceptor : Acceptor {
    size_t size;
    std::vector<Score> history;
    std::vector<bool> filled;
    size_t current = 0;
    bool has_best = false;
    Score best;
    double tolerance;
    DiversifiedLateAcceptanceAcceptor(size_t n, double tol) : size(n), history(n), filled(n, false), tolerance(tol) {}
    static int64_t f64_as_i64(double x) {  // Rust `as i64`: saturating, NaN -> 0
        if (x != x) return 0;
        if (x >= 9223372036854775808.0) return INT64_MAX;
        if (x <= -9223372036854775808.0) return INT64_MIN;
        return (int64_t)x;
    }
    static Score threshold(const Score& best, double tolerance) {  // (:139-146)
        Score t;
        for (int i = 0; i < MAX_LEVELS; ++i) {
            int64_t a = best.v[i] < 0 ? wrap_neg(best.v[i]) : best.v[i];
            t.v[i] = wrap_sub(best.v[i], f64_as_i64(std::round((double)a * tolerance)));
        }
        return t;
    }
    bool is_accepted(const Score& last, const Score& mv) override {  // (:118-150)
        if (mv >= last) return true;
        if (filled[current]) {
            if (mv >= history[current]) return true;
        } else {
            return true;  // no history yet
        }
        if (has_best && mv >= threshold(best, tolerance)) return true;
        return false;
    }
    void phase_started(const Score& initial) override {  // (:152-159)
        for (size_t i = 0; i < size; ++i) {
            history[i] = initial;
            filled[i] = true;
        }
        current = 0;
        best = initial;
        has_best = true;
    }
    void step_ended(const Score& step_score) override {  // (:161-176)
        if (!has_best || step_score > best) {
            best = step_score;
            has_best = true;
        }
        history[current] = step_score;
        filled[current] = true;
        current = (current + 1) % size;
    }
};

// simulated_annealing.rs:11-430.  `levels` = Score::levels_count(), `hard_levels` = how many
// leading levels carry ScoreLevel::Hard (HardSoft: 1; Bendable<H,S>: H).
struct SimulatedAnnealingAcceptor : Acceptor {
    enum Mode { Single = 0, PerLevel = 1, Calibrated = 2 };
    Mode mode = Calibrated;
    double single_temperature = 0.0;
    std::vector<double> level_temperatures;
    // SimulatedAnnealingCalibration::default (:30-38)
    size_t sample_size = 128;
    double target_acceptance_probability = 0.80;
    double fallback_temperature = 1.0;
    double decay_rate = 0.999985;               // DEFAULT_DECAY_RATE (:11)
    double hill_climbing_temperature = 1.0e-9;  // DEFAULT_HILL_CLIMBING_TEMPERATURE (:12)
    bool never_accept_hard_regression = false;  // HardRegressionPolicy (:18-21)
    int levels = 2, hard_levels = 1;
    SmallRng rng;

    std::vector<double> current;  // current_temperatures
    bool calibrating = false;     // calibration_state.is_some()
    std::vector<std::vector<int64_t>> samples_by_level;
    size_t samples_seen = 0;

    std::vector<double> calibrated_temperatures() const {  // CalibrationState::temperatures :72-87
        const double denominator = -std::log(target_acceptance_probability);
        std::vector<double> out;
        for (const auto& samples : samples_by_level) {
            if (samples.empty()) {
                out.push_back(fallback_temperature);
            } else {
                __int128 total = 0;
                for (int64_t v : samples) total += (__int128)v;
                const double mean = (double)total / (double)samples.size();
                out.push_back(std::max(mean / denominator, fallback_temperature));
            }
        }
        return out;
    }

    bool is_accepted(const Score& last, const Score& mv) override {  // :338-375
        if (mv >= last) return true;
        int level = -1;
        for (int k = 0; k < levels; ++k)
            if (last.v[k] != mv.v[k]) {
                level = k;
                break;
            }
        if (level < 0) return false;
        const int64_t delta = wrap_sub(mv.v[level], last.v[level]);
        if (delta >= 0) return false;  // worsening_delta_at_first_difference :289-296
        if (never_accept_hard_regression && level < hard_levels) return false;
        if (calibrating) {
            const int64_t abs = delta == INT64_MIN ? INT64_MAX : -delta;  // saturating_abs
            samples_by_level[(size_t)level].push_back(abs);
            samples_seen += 1;
            if (samples_seen < sample_size) return false;
            current = calibrated_temperatures();  // finalize_calibration_if_ready :257-266
            calibrating = false;
        }
        const double temperature = current[(size_t)level];
        if (temperature <= hill_climbing_temperature) return false;
        const double probability = std::exp((double)delta / temperature);
        return rng.random_f64() < probability;
    }
    void phase_started(const Score&) override {  // :377-415
        calibrating = false;
        samples_seen = 0;
        if (mode == Single)
            current.assign((size_t)levels, single_temperature);
        else if (mode == PerLevel)
            current = level_temperatures;
        else {
            current.assign((size_t)levels, 0.0);
            samples_by_level.assign((size_t)levels, {});
            calibrating = true;
        }
    }
    void step_ended(const Score&) override {  // :417-430
        if (calibrating) return;
        for (double& t : current) {
            t *= decay_rate;
            if (t < hill_climbing_temperature) t = hill_climbing_temperature;
        }
    }
};

inline bool reservoir_pick(uint64_t step_seed, uint64_t equal_count) {  // forager.rs:143-148
    uint64_t mixed =
        splitmix64(step_seed ^ (equal_count * 0x9E3779B97F4A7C15ULL) ^ 0xF04A63E239B74D11ULL);
    return mixed % equal_count == 0;
}

struct BestCandidate {  // forager.rs:70-141
    bool has = false;
    size_t index = 0;
    Score score;
    uint64_t equal_count = 0;
    uint64_t step_seed = 0;
    bool random_ties = true;
    void reset(uint64_t seed) {
        has = false;
        equal_count = 0;
        step_seed = seed;
    }
    void consider(size_t idx, const Score& sc) {
        if (!has) {
            has = true;
            index = idx;
            score = sc;
            equal_count = 1;
            return;
        }
        int c = cmp(sc, score);
        if (c < 0) return;
        if (c > 0) {
            equal_count = 1;
            index = idx;
            score = sc;
            return;
        }
        ++equal_count;
        if (random_ties && reservoir_pick(step_seed, equal_count)) {
            index = idx;
            score = sc;
        }
    }
};

struct Forager {
    // AcceptedCount / FirstAccepted / BestScore: forager.rs:157-420; FirstBestScoreImproving /
    // FirstLastStepScoreImproving: forager/improving.rs:17-227 (accepted_count_limit 0 = None)
    enum Kind { AcceptedCount, FirstAccepted, BestScore, FirstBestScoreImproving, FirstLastStepScoreImproving } kind = AcceptedCount;
    size_t accepted_count_limit = 256;
    size_t accepted_count = 0;
    BestCandidate best;
    Score ref_best_score, ref_last_step_score;  // step_started arguments (improving foragers)
    bool found_improving = false;
    void step_started(uint64_t seed, const Score& best_score = Score::zero(), const Score& last_step_score = Score::zero()) {
        accepted_count = 0;
        best.reset(seed);
        ref_best_score = best_score;
        ref_last_step_score = last_step_score;
        found_improving = false;
    }
    void replace(size_t idx, const Score& sc) {  // BestCandidate::replace (forager.rs:118-124)
        best.has = true;
        best.index = idx;
        best.score = sc;
        best.equal_count = 1;
    }
    void add_move_index(size_t idx, const Score& sc) {
        switch (kind) {
            case AcceptedCount:  // forager.rs:232-239
                if (accepted_count >= accepted_count_limit) return;
                ++accepted_count;
                best.consider(idx, sc);
                return;
            case FirstAccepted:  // forager.rs:318-325
                if (!best.has) {
                    best.has = true;
                    best.index = idx;
                    best.score = sc;
                }
                return;
            case BestScore:  // forager.rs:410-412
                best.consider(idx, sc);
                return;
            case FirstBestScoreImproving:  // improving.rs:91-100
                if (sc > ref_best_score) {
                    found_improving = true;
                    replace(idx, sc);
                    return;
                }
                if (found_improving) return;
                best.consider(idx, sc);
                return;
            case FirstLastStepScoreImproving:  // improving.rs:196-211
                if (found_improving || (accepted_count_limit > 0 && accepted_count >= accepted_count_limit)) return;
                ++accepted_count;
                if (sc > ref_last_step_score) {
                    found_improving = true;
                    replace(idx, sc);
                    return;
                }
                best.consider(idx, sc);
                return;
        }
    }
    bool is_quit_early() const {
        switch (kind) {
            case AcceptedCount:
                return accepted_count >= accepted_count_limit;
            case FirstAccepted:
                return best.has;
            case BestScore:
                return false;
            case FirstBestScoreImproving:  // improving.rs:102-104
                return found_improving;
            case FirstLastStepScoreImproving:  // improving.rs:213-218
                return found_improving || (accepted_count_limit > 0 && accepted_count >= accepted_count_limit);
        }
        return false;
    }
    int64_t limit_for_context() const {  // LocalSearchForager::accepted_count_limit
        if (kind == AcceptedCount) return (int64_t)accepted_count_limit;
        if (kind == FirstAccepted) return 1;
        if (kind == FirstLastStepScoreImproving && accepted_count_limit > 0) return (int64_t)accepted_count_limit;
        return -1;
    }
};

struct SolverStats {  // stats/solver.rs
    uint64_t step_count = 0;
    uint64_t moves_generated = 0;
    uint64_t moves_evaluated = 0;  // includes not-doable candidates (evaluation.rs:33-49)
    uint64_t moves_accepted = 0;
    uint64_t moves_applied = 0;
    uint64_t score_calculations = 0;  // only scored trials (evaluation.rs:60)
    uint64_t moves_not_doable = 0;
};

using CursorFactory =
    std::function<std::unique_ptr<Cursor>(const ScoreDirector&, const MoveStreamContext&)>;

struct StepTrace {  // one record per pulled candidate (for order/score parity tests)
    Move move;
    bool doable;
    Score score;
    bool accepted;
    size_t selector = 0;    // cursor.selector_index(candidate_id) (candidates.rs:84)
    bool selected = false;  // the forager's pick, committed at the end of the step (step.rs:122-147)
    int gate = 0;           // 8 RejectedByHardImprovement, 16 RejectedByScoreImprovement (evaluation.rs:75-113)
};

struct LocalSearch {
    ScoreDirector* director = nullptr;
    CursorFactory open_cursor;
    std::unique_ptr<Acceptor> acceptor;
    Forager forager;
    SelectionOrder selection_order = SelectionOrder::Random;
    uint64_t random_seed = 0;
    std::vector<uint64_t> explicit_step_seeds;  // optional override

    SolverStats stats;
    Score last_step_score;
    bool has_best = false;
    Score best_score;
    Solution best_solution;
    uint64_t phase_step_index = 0;
    uint64_t seed_draws = 0;
    std::vector<StepTrace>* trace = nullptr;  // when set, records every pulled candidate of a step
    bool last_step_applied = false;
    Move last_applied_move;

    void update_best_solution() {  // scope_progress.rs:89-107
        Score current = director->calculate_score();
        if (!has_best || current > best_score) {
            best_solution = director->working;
            best_solution.has_score = true;
            best_solution.score = current;
            best_score = current;
            has_best = true;
        }
    }
    void phase_start() {  // phase.rs:250-261 (+ initialize_working_solution_as_best)
        last_step_score = director->calculate_score();
        update_best_solution();
        acceptor->phase_started(last_step_score);
        phase_step_index = 0;
    }
    uint64_t next_step_seed() {
        uint64_t seed = seed_draws < explicit_step_seeds.size() ? explicit_step_seeds[seed_draws]
                                                                 : sf_step_seed(random_seed, seed_draws);
        ++seed_draws;
        return seed;
    }
    // One local-search step (step.rs:30-225 around candidates.rs:47-285).
    void step() {
        uint64_t step_index = phase_step_index;
        uint64_t step_seed = next_step_seed();
        // best score ever seen, else the last step score (step.rs:53-58,65)
        forager.step_started(step_seed, has_best ? best_score : last_step_score, last_step_score);
        acceptor->step_started();
        MoveStreamContext ctx(step_index, step_seed, forager.limit_for_context());
        ctx = ctx.with_selection_order(selection_order);
        std::unique_ptr<Cursor> cursor = open_cursor(*director, ctx);
        if (trace) trace->clear();

        std::vector<Move> kept;  // candidate ids are stream indices; keep moves to apply the winner
        size_t candidate_index = 0;
        Move mv;
        while (!forager.is_quit_early()) {
            if (!cursor->next(mv)) break;
            size_t id = candidate_index++;
            kept.push_back(mv);
            ++stats.moves_generated;
            ++stats.moves_evaluated;
            // evaluate_candidate (evaluation.rs:20-115)
            if (!move_is_doable(*director, mv)) {
                ++stats.moves_not_doable;
                if (trace) trace->push_back({mv, false, Score::zero(), false, cursor->last_selector(), false});
                continue;
            }
            DirectorScoreState st = director->snapshot_score_state();
            MoveUndo undo = move_do(*director, mv);
            Score move_score = director->calculate_score();
            move_undo(*director, mv, undo);
            director->restore_score_state(st);
            ++stats.score_calculations;
            if (mv.require_improvement && !(move_score > last_step_score)) {  // RejectedByScoreImprovement (evaluation.rs:95-113): never reaches the acceptor
                if (trace) trace->push_back({mv, true, move_score, false, cursor->last_selector(), false, 16});
                continue;
            }
            bool accepted = acceptor->is_accepted(last_step_score, move_score);
            if (trace) trace->push_back({mv, true, move_score, accepted, cursor->last_selector(), false});
            if (accepted) {
                ++stats.moves_accepted;
                forager.add_move_index(id, move_score);
            }
        }
        last_step_applied = false;
        if (forager.best.has) {  // pick_move_index + apply_owned_candidate
            const Move& winner = kept[forager.best.index];
            if (trace) (*trace)[forager.best.index].selected = true;
            MoveUndo ignored = move_do(*director, winner);
            (void)ignored;
            director->calculate_score();
            ++stats.moves_applied;
            last_step_score = forager.best.score;
            last_step_applied = true;
            last_applied_move = winner;
            update_best_solution();
            forager.best.has = false;
        }
        acceptor->step_ended(last_step_score);  // always (step.rs:216-221)
        ++phase_step_index;
        ++stats.step_count;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// One local-search step whose cursor is a GroupedScalarMoveSelector over a ScalarCandidateProvider
// (builder/selector/grouped_scalar.rs:82-176, the Candidates arm of GroupedScalarCursor::activate): `provided` = what the
// provider returned for the working solution (a host closure in the reference).  The cursor applies the selection order
// (salt 0xC0A1_E5CE_AAA0_0001 ^ group_name.len()), skips candidates without edits, repeats of a kept candidate and candidates
// with two edits on one (descriptor, entity, variable), builds the CompoundScalarMove (every value legal:
// compound_move_for_group_candidate, phase/construction/grouped_scalar/move_build.rs:9-24) and keeps it when it is doable, up to
// max_moves_per_step (default 256 for a candidate-backed group, grouped_scalar.rs:27-40).  The kept candidates are then pulled
// through acceptor and forager like any other cursor (candidates.rs:47-285).
// ---------------------------------------------------------------------------------------------------------------------------
struct GroupedStepTrace {
    std::vector<size_t> kept;      // provider indices in pull order
    std::vector<Score> scores;     // per consumed candidate
    std::vector<int32_t> flags;    // bit0 doable, bit1 accepted, bit2 selected
    int64_t selected = -1;         // ordinal (in `kept`) of the committed candidate
};
// hard_score_delta (phase/hard_delta.rs:11-35): +1 Improving, 0 Neutral, -1 Worse by the first differing HARD level; -2 = the score has no hard level
inline int hard_score_delta(const Score& previous, const Score& candidate, int hard_levels) {
    for (int k = 0; k < hard_levels; ++k) {
        if (candidate.v[k] > previous.v[k]) return 1;
        if (candidate.v[k] < previous.v[k]) return -1;
    }
    return hard_levels > 0 ? 0 : -2;
}
// gates[i] of provided candidate i: bit 0 Move::requires_hard_improvement, bit 1 Move::requires_score_improvement (evaluation.rs:75-113)
inline GroupedStepTrace grouped_scalar_step(LocalSearch& ls, const std::vector<std::vector<ScalarEditO>>& provided, size_t group_name_len,
                                            size_t max_moves_per_step, const std::vector<int32_t>& gates = {}, bool cursor_order = false) {
    GroupedStepTrace out;
    ScoreDirector& d = *ls.director;
    uint64_t step_index = ls.phase_step_index;
    uint64_t step_seed = ls.next_step_seed();
    ls.forager.step_started(step_seed, ls.has_best ? ls.best_score : ls.last_step_score, ls.last_step_score);
    ls.acceptor->step_started();
    MoveStreamContext ctx(step_index, step_seed, ls.forager.limit_for_context());
    ctx = ctx.with_selection_order(ls.selection_order);
    // activate(): order, filter, cap
    std::vector<size_t> order(provided.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    if (!ctx.is_canonical()) {
        std::vector<size_t> canonical = order;
        for (size_t o = 0; o < order.size(); ++o)
            order[o] = canonical[ctx.selection_index(o, canonical.size(), 0xC0A1E5CEAAA00001ULL ^ (uint64_t)group_name_len)];
    }
    auto same = [](const std::vector<ScalarEditO>& a, const std::vector<ScalarEditO>& b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i].descriptor != b[i].descriptor || a[i].entity != b[i].entity || a[i].variable != b[i].variable || a[i].to_value != b[i].to_value)
                return false;
        return true;
    };
    // cursor_order: `provided` IS the pull order of a cursor that did its own activation (RuntimeProviderCursor::next_candidate,
    // runtime/provider_cursor.rs:447-466, behind the leaf of runtime/compiler/executor/local_search/leaf.rs:362-402): the step loop of
    // phase/candidates.rs pulls it as it stands
    if (cursor_order) {
        for (size_t i = 0; i < provided.size(); ++i) out.kept.push_back(i);
        order.clear();
    }
    for (size_t idx : order) {
        if (out.kept.size() >= max_moves_per_step) break;
        const auto& cand = provided[idx];
        if (cand.empty()) continue;
        bool seen = false;
        for (size_t k : out.kept) seen = seen || same(provided[k], cand);
        if (seen) continue;
        bool dup_target = false;
        for (size_t i = 0; i < cand.size() && !dup_target; ++i)
            for (size_t j = 0; j < i; ++j)
                if (cand[i].descriptor == cand[j].descriptor && cand[i].entity == cand[j].entity && cand[i].variable == cand[j].variable) dup_target = true;
        if (dup_target) continue;
        bool legal = true;
        for (auto& e : cand) legal = legal && e.legal && e.entity < d.working.classes[e.descriptor].n;
        if (!legal) continue;
        if (!compound_is_doable(d, cand)) continue;
        out.kept.push_back(idx);
    }
    // the pull loop (candidates.rs:47-285)
    size_t pulled = 0;
    while (!ls.forager.is_quit_early()) {
        if (pulled >= out.kept.size()) break;
        const auto& cand = provided[out.kept[pulled]];
        size_t id = pulled++;
        ++ls.stats.moves_generated;
        ++ls.stats.moves_evaluated;
        if (!compound_is_doable(d, cand)) {
            ++ls.stats.moves_not_doable;
            out.scores.push_back(Score::zero());
            out.flags.push_back(0);
            continue;
        }
        DirectorScoreState st = d.snapshot_score_state();
        std::vector<int64_t> undo = compound_do(d, cand);
        Score move_score = d.calculate_score();
        compound_undo(d, cand, undo);
        d.restore_score_state(st);
        ++ls.stats.score_calculations;
        const int32_t gate = gates.empty() ? 0 : gates[out.kept[id]];
        const bool hard_rejected = (gate & 1) && hard_score_delta(ls.last_step_score, move_score, d.hard_levels) != 1;
        if (hard_rejected || ((gate & 2) && !(move_score > ls.last_step_score))) {  // RejectedByHardImprovement / RejectedByScoreImprovement: the
            out.scores.push_back(move_score);                                       // acceptor is not asked
            out.flags.push_back(1 | (hard_rejected ? 8 : 16));
            continue;
        }
        bool accepted = ls.acceptor->is_accepted(ls.last_step_score, move_score);
        out.scores.push_back(move_score);
        out.flags.push_back(1 | (accepted ? 2 : 0));
        if (accepted) {
            ++ls.stats.moves_accepted;
            ls.forager.add_move_index(id, move_score);
        }
    }
    ls.last_step_applied = false;
    if (ls.forager.best.has) {
        out.selected = (int64_t)ls.forager.best.index;
        out.flags[ls.forager.best.index] |= 4;
        (void)compound_do(d, provided[out.kept[ls.forager.best.index]]);
        d.calculate_score();
        ++ls.stats.moves_applied;
        ls.last_step_score = ls.forager.best.score;
        ls.last_step_applied = true;
        ls.update_best_solution();
        ls.forager.best.has = false;
    }
    ls.acceptor->step_ended(ls.last_step_score);
    ++ls.phase_step_index;
    ++ls.stats.step_count;
    return out;
}

// First-fit construction over one scalar slot (phase/construction/forager_step.rs:149-226,
// decision.rs:56-64): entities in index order; when keep-current is legal (the variable
// allows unassigned) the baseline is the current score and the first candidate value whose
// trial score is strictly better is selected; otherwise the first doable candidate wins.
inline void construct_first_fit(ScoreDirector& d, const ScalarSlot& slot, SolverStats* stats = nullptr) {
    d.calculate_score();
    EntityClass& c = d.working.classes[slot.descriptor_index];
    std::vector<int64_t> values;
    for (size_t e = 0; e < c.n; ++e) {
        if (c.vars[slot.variable_index][e] != NONE) continue;
        values.clear();
        slot.values_for_entity(d.working, e, values);
        bool keep_current_legal = slot.allows_unassigned;
        Score baseline = d.calculate_score();
        bool chosen = false;
        int64_t chosen_value = NONE;
        for (int64_t v : values) {
            Move m;
            m.kind = Move::Change;
            m.descriptor = slot.descriptor_index;
            m.variable = slot.variable_index;
            m.a = e;
            m.to_value = v;
            m.allows_unassigned = slot.allows_unassigned;
            if (!move_is_doable(d, m)) continue;
            if (!keep_current_legal) {
                chosen = true;
                chosen_value = v;
                break;
            }
            DirectorScoreState st = d.snapshot_score_state();
            MoveUndo u = move_do(d, m);
            Score sc = d.calculate_score();
            move_undo(d, m, u);
            d.restore_score_state(st);
            if (stats) {
                ++stats->moves_evaluated;
                ++stats->score_calculations;
            }
            if (sc > baseline) {
                chosen = true;
                chosen_value = v;
                break;
            }
        }
        if (chosen) {
            Move m;
            m.kind = Move::Change;
            m.descriptor = slot.descriptor_index;
            m.variable = slot.variable_index;
            m.a = e;
            m.to_value = chosen_value;
            m.allows_unassigned = slot.allows_unassigned;
            move_do(d, m);
            d.calculate_score();
        }
    }
}

// List cheapest-insertion construction (manager/phase_factory/list_construction/cheapest/kernel.rs:57-150, live.rs:64-170): the
// unassigned elements in (construction order key, source index) order -- no order key here -- are placed one by one; every
// (list, position) is trial-inserted and fully scored (one score_calculation each), the strictly best score wins (the first of
// equals stays), the insertion is committed as one accepted + applied step.  Unrestricted owners, no precedence hooks.
// precedence_downstream (cheapest/kernel.rs:162-229): with the phase's precedence hooks the elements are re-ranked (stable) by the longest
// duration-weighted chain of fixed successors AMONG the elements still to place, longest first; a cyclic successor relation leaves the
// order alone.  `hooks` = the slot's precedence hooks (element = node index), null = the phase has none.
inline std::vector<uint32_t> cheapest_precedence_order(const std::vector<uint32_t>& elements, const PrecedenceHooks* hooks) {
    if (!hooks) return elements;
    const size_t m = elements.size();
    std::vector<int64_t> position(hooks->node_count, -1);
    for (size_t i = 0; i < m; ++i) {
        if ((size_t)elements[i] >= hooks->node_count) return elements;
        position[elements[i]] = (int64_t)i;
    }
    std::vector<std::vector<size_t>> succ(m);
    std::vector<size_t> preds(m, 0);
    for (size_t i = 0; i < m; ++i)
        for (size_t to : hooks->successors[elements[i]]) {
            if (to >= hooks->node_count || position[to] < 0) continue;
            succ[i].push_back((size_t)position[to]);
            preds[(size_t)position[to]] += 1;
        }
    std::vector<size_t> ready, topo;
    for (size_t i = 0; i < m; ++i)
        if (preds[i] == 0) ready.push_back(i);
    while (!ready.empty()) {
        size_t i = ready.back();
        ready.pop_back();
        topo.push_back(i);
        for (size_t s2 : succ[i])
            if (--preds[s2] == 0) ready.push_back(s2);
    }
    if (topo.size() != m) return elements;
    std::vector<int64_t> down(m);
    for (size_t i = 0; i < m; ++i) down[i] = hooks->durations[elements[i]];
    for (size_t t = m; t-- > 0;) {
        size_t i = topo[t];
        int64_t tail = 0;
        for (size_t s2 : succ[i]) tail = std::max(tail, down[s2]);
        down[i] = hooks->durations[elements[i]] + tail;
    }
    std::vector<size_t> idx(m);
    for (size_t i = 0; i < m; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return down[a] > down[b]; });
    std::vector<uint32_t> out;
    for (size_t i : idx) out.push_back(elements[i]);
    return out;
}
inline void construct_list_cheapest(ScoreDirector& d, size_t descriptor, const std::vector<uint32_t>& unassigned_in, SolverStats* stats = nullptr,
                                    const PrecedenceHooks* hooks = nullptr) {
    d.calculate_score();
    EntityClass& c = d.working.classes[descriptor];
    if (unassigned_in.empty() || c.n == 0) return;
    const std::vector<uint32_t> unassigned = cheapest_precedence_order(unassigned_in, hooks);
    for (uint32_t element : unassigned) {
        bool have = false;
        size_t best_e = 0, best_p = 0;
        Score best_score;
        for (size_t e = 0; e < c.n; ++e) {
            const size_t len = c.lists[e].size();
            for (size_t pos = 0; pos <= len; ++pos) {
                DirectorScoreState st = d.snapshot_score_state();
                d.before_variable_changed(descriptor, e);
                c.lists[e].insert(c.lists[e].begin() + (ptrdiff_t)pos, element);
                d.after_variable_changed(descriptor, e);
                Score sc = d.calculate_score();
                d.before_variable_changed(descriptor, e);
                c.lists[e].erase(c.lists[e].begin() + (ptrdiff_t)pos);
                d.after_variable_changed(descriptor, e);
                d.restore_score_state(st);
                if (stats) ++stats->score_calculations, ++stats->moves_generated, ++stats->moves_evaluated;  // live.rs:118-127
                if (!have || sc > best_score) {
                    have = true;
                    best_e = e, best_p = pos, best_score = sc;
                }
            }
        }
        if (!have) continue;
        d.before_variable_changed(descriptor, best_e);
        c.lists[best_e].insert(c.lists[best_e].begin() + (ptrdiff_t)best_p, element);
        d.after_variable_changed(descriptor, best_e);
        d.calculate_score();
        if (stats) {
            ++stats->moves_accepted;
            ++stats->moves_applied;
            ++stats->step_count;
        }
    }
}

// Regret-insertion list construction (manager/phase_factory/list_construction/regret/kernel/execute.rs:52-204 over
// kernel/evaluation.rs:120-230 and kernel/mod.rs:19-75): every round prices every (list, position) of every unassigned element with a
// full score trial; an element's regret is best - second best trial score (Forced, above every finite regret, when it has one slot
// only; a second slot that ties the best gives regret zero); the element with the greatest regret is placed at its best slot --
// ties: the better best score, then the longer precedence downstream (zero without hooks), then the earlier of the unassigned order
// (construction order key, source index).  Inside an element the first of equal trial scores stays.  No precedence hooks;
// `order_keys` (parallel to the elements, may be empty) = element_order_key; `owners_in` = the owner hook (list_placement.rs:54-69)
// below the work budget of kernel/fallback.rs:58-84 (the caller checks it: the bounded fallbacks are not restated).  Counters: one generated + evaluated candidate and one score calculation per trial
// (evaluation.rs:60-66 over phase/construction/telemetry.rs:81-93), one accepted + applied step per placed element.
inline void construct_list_regret(ScoreDirector& d, size_t descriptor, const std::vector<uint32_t>& unassigned_in, SolverStats* stats = nullptr,
                                  const std::vector<int64_t>& order_keys = {}, const std::vector<int64_t>& owners_in = {}) {
    d.calculate_score();
    EntityClass& c = d.working.classes[descriptor];
    if (unassigned_in.empty() || c.n == 0) return;
    std::vector<uint32_t> unassigned = unassigned_in;
    std::vector<int64_t> owners = owners_in;  // parallel to the elements: -1 unrestricted, otherwise the owner hook's value
    if (!order_keys.empty()) {  // execute.rs:81-88: sort_by_key((construction_order_key, source_index)); `unassigned_in` is in source order
        std::vector<size_t> order(unassigned.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return order_keys[a] < order_keys[b]; });
        for (size_t i = 0; i < order.size(); ++i) unassigned[i] = unassigned_in[order[i]];
        if (!owners_in.empty())
            for (size_t i = 0; i < order.size(); ++i) owners[i] = owners_in[order[i]];
    }
    while (!unassigned.empty()) {
        bool have_choice = false, choice_forced = false;
        Score choice_regret, choice_score;
        size_t choice_li = 0, choice_e = 0, choice_p = 0;
        for (size_t li = 0; li < unassigned.size(); ++li) {
            const uint32_t element = unassigned[li];
            bool have = false, have_second = false;
            size_t best_e = 0, best_p = 0;
            Score best_score, second;
            // candidate_entities (mod.rs:104-114): every list, the fixed owner's list, or none (an owner hook value >= entity count)
            size_t e_lo = 0, e_hi = c.n;
            if (!owners.empty() && owners[li] >= 0) {
                if ((size_t)owners[li] < c.n)
                    e_lo = (size_t)owners[li], e_hi = e_lo + 1;
                else
                    e_hi = 0;
            }
            for (size_t e = e_lo; e < e_hi; ++e) {
                const size_t len = c.lists[e].size();
                for (size_t pos = 0; pos <= len; ++pos) {
                    DirectorScoreState st = d.snapshot_score_state();
                    d.before_variable_changed(descriptor, e);
                    c.lists[e].insert(c.lists[e].begin() + (ptrdiff_t)pos, element);
                    d.after_variable_changed(descriptor, e);
                    Score sc = d.calculate_score();
                    d.before_variable_changed(descriptor, e);
                    c.lists[e].erase(c.lists[e].begin() + (ptrdiff_t)pos);
                    d.after_variable_changed(descriptor, e);
                    d.restore_score_state(st);
                    if (stats) ++stats->score_calculations, ++stats->moves_generated, ++stats->moves_evaluated;
                    if (!have) {
                        have = true;
                        best_e = e, best_p = pos, best_score = sc;
                    } else if (sc > best_score) {
                        second = best_score, have_second = true;
                        best_e = e, best_p = pos, best_score = sc;
                    } else if (!have_second || sc > second) {
                        second = sc, have_second = true;
                    }
                }
            }
            if (!have) continue;
            const bool forced = !have_second;  // RegretValue::Forced orders above every Finite (mod.rs:46-56)
            const Score regret = forced ? Score::zero() : best_score - second;
            bool better = !have_choice;
            if (have_choice) {
                const int rc = forced != choice_forced ? (forced ? 1 : -1) : (forced ? 0 : cmp(regret, choice_regret));
                better = rc > 0 || (rc == 0 && best_score > choice_score);
            }
            if (better) {
                have_choice = true, choice_forced = forced;
                choice_regret = regret, choice_score = best_score;
                choice_li = li, choice_e = best_e, choice_p = best_p;
            }
        }
        if (!have_choice) break;
        const uint32_t element = unassigned[choice_li];
        unassigned.erase(unassigned.begin() + (ptrdiff_t)choice_li);
        if (!owners.empty()) owners.erase(owners.begin() + (ptrdiff_t)choice_li);
        d.before_variable_changed(descriptor, choice_e);
        c.lists[choice_e].insert(c.lists[choice_e].begin() + (ptrdiff_t)choice_p, element);
        d.after_variable_changed(descriptor, choice_e);
        d.calculate_score();
        if (stats) {
            ++stats->moves_accepted;
            ++stats->moves_applied;
            ++stats->step_count;
        }
    }
}

// Round-robin list construction (manager/phase_factory/list_construction/round_robin/kernel.rs:71-175): the unassigned elements in
// (construction order key, source index) order; an unrestricted element is appended to the round-robin cursor's owner and advances
// it, an element with a fixed owner (list_placement.rs:54-69: owner hook value < entity count) is appended there without advancing,
// an element whose hook names no valid owner is skipped.  One generated + evaluated candidate, one accepted + applied step and one
// score calculation per appended element.  `unassigned` in source order; order_keys / owners (parallel to it) may be empty:
// owners[k] = -1 unrestricted, >= 0 the owner hook's value.
inline void construct_list_round_robin(ScoreDirector& d, size_t descriptor, const std::vector<uint32_t>& unassigned,
                                       const std::vector<int64_t>& order_keys, const std::vector<int64_t>& owners, SolverStats* stats = nullptr) {
    d.calculate_score();
    EntityClass& c = d.working.classes[descriptor];
    const size_t n_entities = c.n;
    if (n_entities == 0 || unassigned.empty()) return;
    std::vector<size_t> order(unassigned.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    if (!order_keys.empty()) std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return order_keys[a] < order_keys[b]; });
    size_t entity_idx = 0;
    for (size_t k : order) {
        size_t target = entity_idx;
        bool advance = true;
        if (!owners.empty() && owners[k] >= 0) {
            if ((size_t)owners[k] >= n_entities) continue;  // OwnerRestriction::Invalid
            target = (size_t)owners[k], advance = false;
        }
        if (stats) ++stats->moves_generated, ++stats->moves_evaluated, ++stats->moves_accepted;
        d.before_variable_changed(descriptor, target);
        c.lists[target].push_back(unassigned[k]);
        d.after_variable_changed(descriptor, target);
        d.calculate_score();
        if (stats) ++stats->moves_applied, ++stats->step_count, ++stats->score_calculations;
        if (advance) entity_idx = (entity_idx + 1) % n_entities;
    }
}

}  // namespace sfo
