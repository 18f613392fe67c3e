// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// CPU restatement of the incremental constraint nodes and the ScoreDirector.
// Data structures deliberately mirror the reference (hash-indexed match sets,
// per-candidate retract/insert) so this is also a fair CPU baseline.
//
// Follows:
//   crates/solverforge-scoring/src/stream/collection_extract.rs:51-94  (ChangeSource)
//   crates/solverforge-scoring/src/constraint/incremental.rs:19-193    (uni)
//   crates/solverforge-scoring/src/constraint/nary_incremental/bi.rs:12-313 (self-join bi)
//   crates/solverforge-scoring/src/constraint/cross_bi_incremental/{state,incremental}.rs
//   crates/solverforge-scoring/src/constraint/exists.rs:42-437 + exists/key_state.rs
//   crates/solverforge-scoring/src/constraint/grouped/{state,scorer,shared_set}.rs
//   crates/solverforge-scoring/src/constraint/list_precedence.rs:13-707 (ListPrecedenceMakespanConstraint)
//   crates/solverforge-scoring/src/stream/collector/runs.rs:11-229 (consecutive_runs collector)
//   crates/solverforge-scoring/src/constraint/complemented/{state,helpers,incremental}.rs (complemented grouped)
//   crates/solverforge-scoring/src/api/constraint_set/incremental.rs:339-407 (tuple fold)
//   crates/solverforge-scoring/src/director/score_director/incremental.rs:141-218
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "sfo_core.hpp"

namespace sfo {

constexpr int64_t NONE = -1;  // Option<usize>::None for scalar planning variables

// Generic working solution: entity classes addressed by descriptor_index
// (crates/solverforge-core/src/domain/descriptor/solution.rs:16-33).
struct EntityClass {
    size_t n = 0;
    std::vector<std::vector<int64_t>> vars;    // vars[variable_index][entity]; NONE = unassigned
    std::vector<std::vector<uint32_t>> lists;  // lists[entity] — the class's list variable (if any)
};

struct Solution {
    std::vector<EntityClass> classes;
    std::shared_ptr<const void> facts;  // immutable problem facts (model specific)
    bool has_score = false;
    Score score;
};

enum class Impact { Penalty, Reward };

struct ChangeSource {  // collection_extract.rs:51-94
    enum Kind { Unknown, Static, Descriptor } kind = Unknown;
    size_t index = 0;
    static ChangeSource unknown() { return {Unknown, 0}; }
    static ChangeSource fixed() { return {Static, 0}; }
    static ChangeSource descriptor(size_t i) { return {Descriptor, i}; }
    bool reacts_to(size_t d) const {
        return kind == Unknown || (kind == Descriptor && index == d);
    }
    bool owns_descriptor(size_t d) const { return kind == Descriptor && index == d; }
    bool same_index_domain(const ChangeSource& o) const {
        return kind == Descriptor && o.kind == Descriptor && index == o.index;
    }
    bool assert_localizes(size_t d, const std::string& name) const {
        if (owns_descriptor(d)) return true;
        if (reacts_to(d))
            throw std::runtime_error("constraint `" + name +
                                     "` cannot localize entity indexes");  // panic! in the reference
        return false;
    }
};

struct Constraint {
    std::string name;
    bool is_hard = false;
    virtual ~Constraint() = default;
    virtual Score evaluate(const Solution& s) const = 0;
    virtual size_t match_count(const Solution& s) const = 0;
    virtual Score initialize(const Solution& s) = 0;
    virtual Score on_insert(const Solution& s, size_t entity, size_t descriptor) = 0;
    virtual Score on_retract(const Solution& s, size_t entity, size_t descriptor) = 0;
    virtual void reset() = 0;
};

using CountFn = std::function<size_t(const Solution&)>;
using Filter1 = std::function<bool(const Solution&, size_t)>;
using Weight1 = std::function<Score(const Solution&, size_t)>;
using Key1 = std::function<int64_t(const Solution&, size_t)>;
using Filter2 = std::function<bool(const Solution&, size_t, size_t)>;
using Weight2 = std::function<Score(const Solution&, size_t, size_t)>;

inline Score apply_impact(Impact impact, const Score& base) {
    return impact == Impact::Penalty ? -base : base;
}

// ---- uni (constraint/incremental.rs:19-193): stateless ---------------------
struct UniConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Filter1 filter;
    Weight1 weight;

    Score evaluate(const Solution& s) const override {
        Score total;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) total = total + apply_impact(impact, weight(s, i));
        return total;
    }
    size_t match_count(const Solution& s) const override {
        size_t c = 0, n = count(s);
        for (size_t i = 0; i < n; ++i) c += filter(s, i) ? 1 : 0;
        return c;
    }
    Score initialize(const Solution& s) override { return evaluate(s); }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        if (e >= count(s)) return Score::zero();
        return filter(s, e) ? apply_impact(impact, weight(s, e)) : Score::zero();
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        if (e >= count(s)) return Score::zero();
        return filter(s, e) ? -apply_impact(impact, weight(s, e)) : Score::zero();
    }
    void reset() override {}
};

struct PairHash {
    size_t operator()(const std::pair<size_t, size_t>& p) const {
        return std::hash<uint64_t>()(((uint64_t)p.first << 32) ^ (uint64_t)p.second ^
                                     ((uint64_t)p.first >> 32));
    }
};
using Pair = std::pair<size_t, size_t>;
using PairSet = std::unordered_set<Pair, PairHash>;

// ---- keyed self-join (nary_incremental/bi.rs:12-313) ----------------------
struct SelfJoinBiConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Key1 key;
    Filter2 filter;  // (low, high)
    Weight2 weight;  // (low, high)

    std::unordered_map<size_t, PairSet> entity_to_matches;
    PairSet matches;
    std::unordered_map<int64_t, std::unordered_set<size_t>> key_to_indices;
    std::unordered_map<size_t, int64_t> index_to_key;

    Score compute(const Solution& s, size_t a, size_t b) const {
        return apply_impact(impact, weight(s, a, b));
    }
    Score insert_entity(const Solution& s, size_t index) {  // bi.rs:78-130
        size_t n = count(s);
        if (index >= n) return Score::zero();
        int64_t k = key(s, index);
        index_to_key[index] = k;
        key_to_indices[k].insert(index);
        Score total;
        auto it = key_to_indices.find(k);
        if (it != key_to_indices.end()) {
            for (size_t other : it->second) {
                if (other == index) continue;
                size_t low = index < other ? index : other;
                size_t high = index < other ? other : index;
                if (filter(s, low, high)) {
                    Pair p{low, high};
                    if (matches.insert(p).second) {
                        entity_to_matches[low].insert(p);
                        entity_to_matches[high].insert(p);
                        total = total + compute(s, low, high);
                    }
                }
            }
        }
        return total;
    }
    Score retract_entity(const Solution& s, size_t index) {  // bi.rs:132-166
        auto ik = index_to_key.find(index);
        if (ik != index_to_key.end()) {
            auto kb = key_to_indices.find(ik->second);
            if (kb != key_to_indices.end()) {
                kb->second.erase(index);
                if (kb->second.empty()) key_to_indices.erase(kb);
            }
            index_to_key.erase(ik);
        }
        auto em = entity_to_matches.find(index);
        if (em == entity_to_matches.end()) return Score::zero();
        PairSet pairs = std::move(em->second);
        entity_to_matches.erase(em);
        size_t n = count(s);
        Score total;
        for (const Pair& p : pairs) {
            matches.erase(p);
            size_t other = p.first == index ? p.second : p.first;
            auto om = entity_to_matches.find(other);
            if (om != entity_to_matches.end()) {
                om->second.erase(p);
                if (om->second.empty()) entity_to_matches.erase(om);
            }
            // weight recomputed on retract from the pre-change state (bi.rs:158-162)
            if (p.first < n && p.second < n) total = total - compute(s, p.first, p.second);
        }
        return total;
    }

    Score evaluate(const Solution& s) const override {  // bi.rs:181-206
        size_t n = count(s);
        std::unordered_map<int64_t, std::vector<size_t>> tmp;
        for (size_t i = 0; i < n; ++i) tmp[key(s, i)].push_back(i);
        Score total;
        for (auto& kv : tmp) {
            auto& idx = kv.second;
            for (size_t i = 0; i < idx.size(); ++i)
                for (size_t j = i + 1; j < idx.size(); ++j)
                    if (filter(s, idx[i], idx[j])) total = total + compute(s, idx[i], idx[j]);
        }
        return total;
    }
    size_t match_count(const Solution& s) const override {
        size_t n = count(s), c = 0;
        std::unordered_map<int64_t, std::vector<size_t>> tmp;
        for (size_t i = 0; i < n; ++i) tmp[key(s, i)].push_back(i);
        for (auto& kv : tmp) {
            auto& idx = kv.second;
            for (size_t i = 0; i < idx.size(); ++i)
                for (size_t j = i + 1; j < idx.size(); ++j)
                    if (filter(s, idx[i], idx[j])) ++c;
        }
        return c;
    }
    Score initialize(const Solution& s) override {
        reset();
        Score total;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i) total = total + insert_entity(s, i);
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        return insert_entity(s, e);
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        return retract_entity(s, e);
    }
    void reset() override {
        entity_to_matches.clear();
        matches.clear();
        key_to_indices.clear();
        index_to_key.clear();
    }
};

// ---- tri / quad / penta self-join (constraint/nary_incremental/higher_arity/shared.rs:71-417) ----------------
// One node for arity 3..5 (the reference stamps the same scaffolding per arity with a macro): entities sharing a
// key form index-sorted tuples; insert enumerates the combinations of OTHER members of the key bucket in increasing
// index order (for_each_other_indices_combination :48-69), retract replays the stored tuples.
struct SelfJoinNaryConstraint : Constraint {
    using Tuple = std::array<size_t, 5>;  // unused slots = SIZE_MAX
    struct TupleHash {
        size_t operator()(const Tuple& t) const {
            size_t h = 1469598103934665603ull;
            for (size_t v : t) h = (h ^ v) * 1099511628211ull;
            return h;
        }
    };
    using TupleSet = std::unordered_set<Tuple, TupleHash>;
    size_t arity = 3;
    Impact impact;
    ChangeSource source;
    CountFn count;
    Key1 key;
    std::function<bool(const Solution&, const size_t* idx)> filter;    // indices ascending
    std::function<Score(const Solution&, const size_t* idx)> weight;

    std::unordered_map<size_t, TupleSet> entity_to_matches;
    TupleSet matches;
    std::unordered_map<int64_t, std::unordered_set<size_t>> key_to_indices;
    std::unordered_map<size_t, int64_t> index_to_key;

    Score compute(const Solution& s, const Tuple& t) const { return apply_impact(impact, weight(s, t.data())); }
    // every (arity - 1)-subset of `others` (ascending values, none equal to `index`), shared.rs:48-69
    template <class F>
    static void other_combinations(const std::vector<size_t>& sorted_others, size_t need, size_t from, std::vector<size_t>& pick, F&& f) {
        if (need == 0) {
            f(pick);
            return;
        }
        for (size_t i = from; i + need <= sorted_others.size(); ++i) {
            pick.push_back(sorted_others[i]);
            other_combinations(sorted_others, need - 1, i + 1, pick, f);
            pick.pop_back();
        }
    }
    Score insert_entity(const Solution& s, size_t index) {  // shared.rs:171-229
        if (index >= count(s)) return Score::zero();
        int64_t k = key(s, index);
        index_to_key[index] = k;
        key_to_indices[k].insert(index);
        std::vector<size_t> others;
        for (size_t o : key_to_indices[k])
            if (o != index) others.push_back(o);
        std::sort(others.begin(), others.end());  // the result is a sum over a set of tuples: order-independent
        Score total;
        std::vector<size_t> pick;
        other_combinations(others, arity - 1, 0, pick, [&](const std::vector<size_t>& c) {
            Tuple t;
            t.fill(SIZE_MAX);
            t[0] = index;
            for (size_t i = 0; i < c.size(); ++i) t[i + 1] = c[i];
            std::sort(t.begin(), t.begin() + arity);
            if (matches.count(t)) return;
            if (filter(s, t.data()) && matches.insert(t).second) {
                for (size_t i = 0; i < arity; ++i) entity_to_matches[t[i]].insert(t);
                total = total + compute(s, t);
            }
        });
        return total;
    }
    Score retract_entity(const Solution& s, size_t index) {  // shared.rs:231-270
        auto ik = index_to_key.find(index);
        if (ik != index_to_key.end()) {
            auto kb = key_to_indices.find(ik->second);
            if (kb != key_to_indices.end()) {
                kb->second.erase(index);
                if (kb->second.empty()) key_to_indices.erase(kb);
            }
            index_to_key.erase(ik);
        }
        auto em = entity_to_matches.find(index);
        if (em == entity_to_matches.end()) return Score::zero();
        TupleSet tuples = std::move(em->second);
        entity_to_matches.erase(em);
        size_t n = count(s);
        Score total;
        for (const Tuple& t : tuples) {
            matches.erase(t);
            bool in_range = true;
            for (size_t i = 0; i < arity; ++i) {
                if (t[i] >= n) in_range = false;
                if (t[i] == index) continue;
                auto om = entity_to_matches.find(t[i]);
                if (om != entity_to_matches.end()) {
                    om->second.erase(t);
                    if (om->second.empty()) entity_to_matches.erase(om);
                }
            }
            if (in_range) total = total - compute(s, t);
        }
        return total;
    }
    template <class F>
    void for_each_tuple(const Solution& s, F&& f) const {  // evaluate / match_count: shared.rs:287-316
        size_t n = count(s);
        std::unordered_map<int64_t, std::vector<size_t>> tmp;
        for (size_t i = 0; i < n; ++i) tmp[key(s, i)].push_back(i);
        for (auto& kv : tmp) {
            std::vector<size_t> pick;
            other_combinations(kv.second, arity, 0, pick, [&](const std::vector<size_t>& c) {
                Tuple t;
                t.fill(SIZE_MAX);
                for (size_t i = 0; i < c.size(); ++i) t[i] = c[i];
                if (filter(s, t.data())) f(t);
            });
        }
    }
    Score evaluate(const Solution& s) const override {
        Score total;
        for_each_tuple(s, [&](const Tuple& t) { total = total + compute(s, t); });
        return total;
    }
    size_t match_count(const Solution& s) const override {
        size_t c = 0;
        for_each_tuple(s, [&](const Tuple&) { ++c; });
        return c;
    }
    Score initialize(const Solution& s) override {
        reset();
        Score total;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i) total = total + insert_entity(s, i);
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        return insert_entity(s, e);
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        return retract_entity(s, e);
    }
    void reset() override {
        entity_to_matches.clear();
        matches.clear();
        key_to_indices.clear();
        index_to_key.clear();
    }
};

// ---- cross join A x B (cross_bi_incremental/{state,incremental}.rs) -------
// Predicate joins use a constant key for both sides (stream/join_target.rs:82-110).
struct CrossBiConstraint : Constraint {
    Impact impact;
    ChangeSource a_source, b_source;
    CountFn a_count, b_count;
    Key1 key_a, key_b;
    Filter2 filter;  // (a_idx, b_idx)
    Weight2 weight;  // (a_idx, b_idx)

    struct MatchRow {
        Pair pair;
        Score score;
        size_t a_pos, b_pos;
    };
    std::unordered_map<Pair, size_t, PairHash> matches;
    std::vector<MatchRow> match_rows;
    std::unordered_map<size_t, std::vector<size_t>> a_to_matches, b_to_matches;
    std::unordered_map<int64_t, std::vector<size_t>> a_by_key, b_by_key;
    std::unordered_map<size_t, int64_t> a_index_to_key, b_index_to_key;

    Score compute(const Solution& s, size_t a, size_t b) const {
        return apply_impact(impact, weight(s, a, b));
    }
    Score add_match(const Solution& s, size_t a, size_t b) {  // state.rs:260-296
        Pair pair{a, b};
        if (matches.count(pair)) return Score::zero();
        if (!filter(s, a, b)) return Score::zero();
        Score score = compute(s, a, b);  // frozen at add time (state.rs:280-295)
        size_t row = match_rows.size();
        auto& ab = a_to_matches[a];
        size_t a_pos = ab.size();
        ab.push_back(row);
        auto& bb = b_to_matches[b];
        size_t b_pos = bb.size();
        bb.push_back(row);
        match_rows.push_back({pair, score, a_pos, b_pos});
        matches[pair] = row;
        return score;
    }
    void remove_from_bucket(std::unordered_map<size_t, std::vector<size_t>>& buckets, size_t idx,
                            size_t pos, bool a_side) {
        auto it = buckets.find(idx);
        if (it == buckets.end()) return;
        auto& v = it->second;
        v[pos] = v.back();
        v.pop_back();
        if (pos < v.size()) {
            if (a_side)
                match_rows[v[pos]].a_pos = pos;
            else
                match_rows[v[pos]].b_pos = pos;
        }
        if (v.empty()) buckets.erase(it);
    }
    Score remove_match_at(size_t row_idx) {  // state.rs:298-322
        if (row_idx >= match_rows.size()) return Score::zero();
        MatchRow row = match_rows[row_idx];
        matches.erase(row.pair);
        remove_from_bucket(a_to_matches, row.pair.first, row.a_pos, true);
        remove_from_bucket(b_to_matches, row.pair.second, row.b_pos, false);
        size_t last = match_rows.size() - 1;
        match_rows[row_idx] = match_rows[last];
        match_rows.pop_back();
        if (row_idx != last) {
            MatchRow& moved = match_rows[row_idx];
            matches[moved.pair] = row_idx;
            auto am = a_to_matches.find(moved.pair.first);
            if (am != a_to_matches.end()) am->second[moved.a_pos] = row_idx;
            auto bm = b_to_matches.find(moved.pair.second);
            if (bm != b_to_matches.end()) bm->second[moved.b_pos] = row_idx;
        }
        return -row.score;
    }
    static void remove_index_from_key_bucket(std::unordered_map<int64_t, std::vector<size_t>>& m,
                                             int64_t k, size_t idx) {
        auto it = m.find(k);
        if (it == m.end()) return;
        auto& v = it->second;
        for (size_t p = 0; p < v.size(); ++p)
            if (v[p] == idx) {
                v[p] = v.back();
                v.pop_back();
                break;
            }
        if (v.empty()) m.erase(it);
    }
    Score insert_a(const Solution& s, size_t a) {  // state.rs:372-401
        if (a >= a_count(s)) return Score::zero();
        int64_t k = key_a(s, a);
        a_index_to_key[a] = k;
        a_by_key[k].push_back(a);
        std::vector<size_t> bs;
        auto it = b_by_key.find(k);
        if (it != b_by_key.end()) bs = it->second;  // cloned like the reference
        Score total;
        for (size_t b : bs) total = total + add_match(s, a, b);
        return total;
    }
    Score retract_a(size_t a) {  // state.rs:403-418
        auto ik = a_index_to_key.find(a);
        if (ik != a_index_to_key.end()) {
            remove_index_from_key_bucket(a_by_key, ik->second, a);
            a_index_to_key.erase(ik);
        }
        Score total;
        for (;;) {
            auto it = a_to_matches.find(a);
            if (it == a_to_matches.end() || it->second.empty()) break;
            total = total + remove_match_at(it->second.back());
        }
        return total;
    }
    Score insert_b(const Solution& s, size_t b) {  // state.rs:420-446
        if (b >= b_count(s)) return Score::zero();
        int64_t k = key_b(s, b);
        b_index_to_key[b] = k;
        b_by_key[k].push_back(b);
        std::vector<size_t> as;
        auto it = a_by_key.find(k);
        if (it != a_by_key.end()) as = it->second;
        Score total;
        for (size_t a : as) total = total + add_match(s, a, b);
        return total;
    }
    Score retract_b(size_t b) {  // state.rs:448-461
        auto ik = b_index_to_key.find(b);
        if (ik != b_index_to_key.end()) {
            remove_index_from_key_bucket(b_by_key, ik->second, b);
            b_index_to_key.erase(ik);
        }
        Score total;
        for (;;) {
            auto it = b_to_matches.find(b);
            if (it == b_to_matches.end() || it->second.empty()) break;
            total = total + remove_match_at(it->second.back());
        }
        return total;
    }

    Score evaluate(const Solution& s) const override {  // incremental.rs:27-47
        size_t na = a_count(s), nb = b_count(s);
        std::unordered_map<int64_t, std::vector<size_t>> bk;
        for (size_t b = 0; b < nb; ++b) bk[key_b(s, b)].push_back(b);
        Score total;
        for (size_t a = 0; a < na; ++a) {
            auto it = bk.find(key_a(s, a));
            if (it == bk.end()) continue;
            for (size_t b : it->second)
                if (filter(s, a, b)) total = total + compute(s, a, b);
        }
        return total;
    }
    size_t match_count(const Solution& s) const override {
        size_t na = a_count(s), nb = b_count(s), c = 0;
        std::unordered_map<int64_t, std::vector<size_t>> bk;
        for (size_t b = 0; b < nb; ++b) bk[key_b(s, b)].push_back(b);
        for (size_t a = 0; a < na; ++a) {
            auto it = bk.find(key_a(s, a));
            if (it == bk.end()) continue;
            for (size_t b : it->second) c += filter(s, a, b) ? 1 : 0;
        }
        return c;
    }
    Score initialize(const Solution& s) override {  // incremental.rs:71-91
        reset();
        size_t na = a_count(s), nb = b_count(s);
        for (size_t a = 0; a < na; ++a) {
            int64_t k = key_a(s, a);
            a_index_to_key[a] = k;
            a_by_key[k].push_back(a);
        }
        for (size_t b = 0; b < nb; ++b) {
            int64_t k = key_b(s, b);
            b_index_to_key[b] = k;
            b_by_key[k].push_back(b);
        }
        Score total;
        for (size_t a = 0; a < na; ++a) {
            auto it = b_by_key.find(key_a(s, a));
            if (it == b_by_key.end()) continue;
            std::vector<size_t> bs = it->second;
            for (size_t b : bs) total = total + add_match(s, a, b);
        }
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {  // incremental.rs:93-115
        bool a_changed = a_source.assert_localizes(d, name);
        bool b_changed = b_source.assert_localizes(d, name);
        Score total;
        if (!a_changed && !b_changed) return total;
        if (a_changed) total = total + insert_a(s, e);
        if (b_changed) total = total + insert_b(s, e);
        return total;
    }
    Score on_retract(const Solution&, size_t e, size_t d) override {  // incremental.rs:117-137
        bool a_changed = a_source.assert_localizes(d, name);
        bool b_changed = b_source.assert_localizes(d, name);
        Score total;
        if (!a_changed && !b_changed) return total;
        if (a_changed) total = total + retract_a(e);
        if (b_changed) total = total + retract_b(e);
        return total;
    }
    void reset() override {
        matches.clear();
        match_rows.clear();
        a_to_matches.clear();
        b_to_matches.clear();
        a_by_key.clear();
        b_by_key.clear();
        a_index_to_key.clear();
        b_index_to_key.clear();
    }
};

// ---- exists / not-exists over a (flattened) B side (constraint/exists.rs) ---
enum class ExistenceMode { Exists, NotExists };
using FlattenFn = std::function<void(const Solution&, size_t parent, std::vector<int64_t>& keys_out)>;

struct ExistsConstraint : Constraint {
    Impact impact;
    ExistenceMode mode;
    ChangeSource a_source, parent_source;
    CountFn a_count, parent_count;
    Filter1 filter_a, filter_parent;
    Key1 key_a;
    FlattenFn flatten;  // keys of the parent's flattened B items (a plain B row = one key)
    Weight1 weight;
    bool indexed_usize = true;  // usize keys use dense Vec tables (exists/key_state.rs:33-60)

    struct ASlot {
        bool has_key = false;
        int64_t key = 0;
        size_t bucket_pos = 0;
        Score score;
    };
    std::vector<ASlot> a_slots;
    // dense storage
    std::vector<std::vector<size_t>> d_a_indices;
    std::vector<Score> d_a_totals;
    std::vector<size_t> d_b_counts;
    // hashed storage
    std::unordered_map<int64_t, std::vector<size_t>> h_a_indices;
    std::unordered_map<int64_t, Score> h_a_totals;
    std::unordered_map<int64_t, size_t> h_b_counts;

    bool matches_count(size_t c) const {
        return mode == ExistenceMode::Exists ? c > 0 : c == 0;
    }
    size_t b_count(int64_t k) const {
        if (indexed_usize) return (size_t)k < d_b_counts.size() ? d_b_counts[(size_t)k] : 0;
        auto it = h_b_counts.find(k);
        return it == h_b_counts.end() ? 0 : it->second;
    }
    void inc_b(int64_t k, size_t by) {
        if (indexed_usize) {
            if (d_b_counts.size() <= (size_t)k) d_b_counts.resize((size_t)k + 1, 0);
            d_b_counts[(size_t)k] += by;
        } else
            h_b_counts[k] += by;
    }
    void dec_b(int64_t k, size_t by) {
        if (indexed_usize) {
            if ((size_t)k < d_b_counts.size())
                d_b_counts[(size_t)k] = d_b_counts[(size_t)k] >= by ? d_b_counts[(size_t)k] - by : 0;
        } else {
            auto it = h_b_counts.find(k);
            if (it != h_b_counts.end()) {
                it->second = it->second >= by ? it->second - by : 0;
                if (it->second == 0) h_b_counts.erase(it);
            }
        }
    }
    Score a_total(int64_t k) const {
        if (indexed_usize) return (size_t)k < d_a_totals.size() ? d_a_totals[(size_t)k] : Score::zero();
        auto it = h_a_totals.find(k);
        return it == h_a_totals.end() ? Score::zero() : it->second;
    }
    void add_a_total(int64_t k, const Score& sc) {
        if (indexed_usize) {
            if (d_a_totals.size() <= (size_t)k) d_a_totals.resize((size_t)k + 1);
            d_a_totals[(size_t)k] = d_a_totals[(size_t)k] + sc;
        } else
            h_a_totals[k] = h_a_totals[k] + sc;
    }
    std::vector<size_t>& a_bucket(int64_t k) {
        if (indexed_usize) {
            if (d_a_indices.size() <= (size_t)k) d_a_indices.resize((size_t)k + 1);
            return d_a_indices[(size_t)k];
        }
        return h_a_indices[k];
    }
    Score compute(const Solution& s, size_t a) const { return apply_impact(impact, weight(s, a)); }

    Score retract_a(size_t idx) {  // exists.rs:168-186
        if (idx >= a_slots.size()) return Score::zero();
        ASlot slot = a_slots[idx];
        if (!slot.has_key) return Score::zero();
        Score contribution = matches_count(b_count(slot.key)) ? slot.score : Score::zero();
        auto& bucket = a_bucket(slot.key);
        if (slot.bucket_pos < bucket.size()) {
            bucket[slot.bucket_pos] = bucket.back();
            bucket.pop_back();
            if (slot.bucket_pos < bucket.size()) a_slots[bucket[slot.bucket_pos]].bucket_pos = slot.bucket_pos;
        }
        add_a_total(slot.key, -slot.score);
        a_slots[idx] = ASlot{};
        return -contribution;
    }
    Score insert_a(const Solution& s, size_t idx) {  // exists.rs:188-216
        size_t n = a_count(s);
        if (idx >= n) return Score::zero();
        if (a_slots.size() < n) a_slots.resize(n);
        if (!filter_a(s, idx)) {
            a_slots[idx] = ASlot{};
            return Score::zero();
        }
        int64_t k = key_a(s, idx);
        auto& bucket = a_bucket(k);
        size_t pos = bucket.size();
        bucket.push_back(idx);
        Score sc = compute(s, idx);
        add_a_total(k, sc);
        Score contribution = matches_count(b_count(k)) ? sc : Score::zero();
        a_slots[idx] = ASlot{true, k, pos, sc};
        return contribution;
    }
    Score key_existence_delta(int64_t k, size_t old_c, size_t new_c) const {  // exists.rs:218-231
        bool o = matches_count(old_c), n = matches_count(new_c);
        if (o == n) return Score::zero();
        return n ? a_total(k) : -a_total(k);
    }
    using KeyCounts = std::vector<std::pair<int64_t, size_t>>;
    Score update_key_counts(const KeyCounts& kc, bool insert) {  // exists.rs:233-247
        Score total;
        for (auto& e : kc) {
            size_t old_c = b_count(e.first);
            if (insert)
                inc_b(e.first, e.second);
            else
                dec_b(e.first, e.second);
            total = total + key_existence_delta(e.first, old_c, b_count(e.first));
        }
        return total;
    }
    KeyCounts parent_key_counts(const Solution& s, size_t idx) const {  // exists.rs:249-272
        KeyCounts kc;
        if (idx >= parent_count(s)) return kc;
        if (!filter_parent(s, idx)) return kc;
        std::vector<int64_t> keys;
        flatten(s, idx, keys);
        for (int64_t k : keys) {
            bool found = false;
            for (auto& e : kc)
                if (e.first == k) {
                    e.second += 1;
                    found = true;
                    break;
                }
            if (!found) kc.push_back({k, 1});
        }
        return kc;
    }
    void rebuild_b_counts(const Solution& s) {
        d_b_counts.clear();
        h_b_counts.clear();
        size_t np = parent_count(s);
        std::vector<int64_t> keys;
        for (size_t p = 0; p < np; ++p) {
            if (!filter_parent(s, p)) continue;
            keys.clear();
            flatten(s, p, keys);
            for (int64_t k : keys) inc_b(k, 1);
        }
    }
    Score evaluate(const Solution& s) const override {  // exists.rs:337-352
        ExistsConstraint tmp;
        tmp.indexed_usize = indexed_usize;
        tmp.parent_count = parent_count;
        tmp.filter_parent = filter_parent;
        tmp.flatten = flatten;
        tmp.rebuild_b_counts(s);
        Score total;
        size_t n = a_count(s);
        for (size_t a = 0; a < n; ++a) {
            if (!filter_a(s, a)) continue;
            if (matches_count(tmp.b_count(key_a(s, a)))) total = total + compute(s, a);
        }
        return total;
    }
    size_t match_count(const Solution& s) const override {
        ExistsConstraint tmp;
        tmp.indexed_usize = indexed_usize;
        tmp.parent_count = parent_count;
        tmp.filter_parent = filter_parent;
        tmp.flatten = flatten;
        tmp.rebuild_b_counts(s);
        size_t n = a_count(s), c = 0;
        for (size_t a = 0; a < n; ++a)
            if (filter_a(s, a) && matches_count(tmp.b_count(key_a(s, a)))) ++c;
        return c;
    }
    Score initialize(const Solution& s) override {  // exists.rs:358-362
        reset();
        rebuild_b_counts(s);
        size_t n = a_count(s);
        a_slots.assign(n, ASlot{});
        Score total;
        for (size_t a = 0; a < n; ++a) total = total + insert_a(s, a);
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {  // exists.rs:364-391
        bool a_changed = a_source.assert_localizes(d, name);
        bool p_changed = parent_source.assert_localizes(d, name);
        bool same = a_source.same_index_domain(parent_source) && a_changed && p_changed;
        Score total;
        if (same) {
            KeyCounts kc = parent_key_counts(s, e);
            total = total + update_key_counts(kc, true);
            total = total + insert_a(s, e);
            return total;
        }
        if (p_changed) total = total + update_key_counts(parent_key_counts(s, e), true);
        if (a_changed) total = total + insert_a(s, e);
        return total;
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {  // exists.rs:393-420
        bool a_changed = a_source.assert_localizes(d, name);
        bool p_changed = parent_source.assert_localizes(d, name);
        bool same = a_source.same_index_domain(parent_source) && a_changed && p_changed;
        Score total;
        if (same) {
            KeyCounts kc = parent_key_counts(s, e);
            total = total + retract_a(e);
            total = total + update_key_counts(kc, false);
            return total;
        }
        if (a_changed) total = total + retract_a(e);
        if (p_changed) total = total + update_key_counts(parent_key_counts(s, e), false);
        return total;
    }
    void reset() override {
        a_slots.clear();
        d_a_indices.clear();
        d_a_totals.clear();
        d_b_counts.clear();
        h_a_indices.clear();
        h_a_totals.clear();
        h_b_counts.clear();
    }
};

// ---- grouped uni: group_by(key, count|sum) -> weight(key, result) ---------
// (grouped/state.rs:43-247, grouped/scorer.rs:46-152, stream/collector/{sum,count}.rs)
using GroupWeight = std::function<Score(int64_t key, int64_t result)>;
using Value1 = std::function<int64_t(const Solution&, size_t)>;

struct GroupedConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Filter1 filter;
    Key1 key;
    Value1 value;  // count collector: value == 1 per entity; sum collector: mapped value
    GroupWeight weight;

    struct Group {
        int64_t key;
        int64_t acc;
        size_t count;
    };
    std::vector<Group> groups;
    std::unordered_map<int64_t, size_t> group_of_key;
    std::unordered_map<size_t, size_t> entity_groups;
    std::unordered_map<size_t, int64_t> entity_retractions;
    std::vector<size_t> changed_groups;
    std::vector<Score> cached_scores;

    Score compute(int64_t k, int64_t r) const { return apply_impact(impact, weight(k, r)); }
    void mark_changed(size_t g) {
        for (size_t c : changed_groups)
            if (c == g) return;
        changed_groups.push_back(g);
    }
    size_t group_id_for_key(int64_t k) {
        auto it = group_of_key.find(k);
        if (it != group_of_key.end()) return it->second;
        size_t g = groups.size();
        groups.push_back({k, 0, 0});
        group_of_key[k] = g;
        return g;
    }
    void insert_entity(const Solution& s, size_t idx) {  // state.rs:216-229
        size_t g = group_id_for_key(key(s, idx));
        if (groups[g].count == 0) groups[g].acc = 0;
        int64_t v = value(s, idx);
        groups[g].acc = wrap_add(groups[g].acc, v);
        groups[g].count += 1;
        entity_groups[idx] = g;
        entity_retractions[idx] = v;
        mark_changed(g);
    }
    void retract_entity(size_t idx) {  // state.rs:231-244
        auto eg = entity_groups.find(idx);
        if (eg == entity_groups.end()) return;
        size_t g = eg->second;
        entity_groups.erase(eg);
        auto er = entity_retractions.find(idx);
        if (er == entity_retractions.end()) return;
        int64_t v = er->second;
        entity_retractions.erase(er);
        groups[g].acc = wrap_sub(groups[g].acc, v);
        groups[g].count = groups[g].count > 0 ? groups[g].count - 1 : 0;
        mark_changed(g);
    }
    Score replace_cached(size_t slot, const Score& sc) {  // scorer.rs:145-152
        while (cached_scores.size() <= slot) cached_scores.push_back(Score::zero());
        Score prev = cached_scores[slot];
        cached_scores[slot] = sc;
        return sc - prev;
    }
    Score refresh_changed() {  // scorer.rs:89-101
        Score delta;
        for (size_t g : changed_groups) {
            if (g >= groups.size()) continue;
            Score sc = groups[g].count == 0 ? Score::zero() : compute(groups[g].key, groups[g].acc);
            delta = delta + replace_cached(g, sc);
        }
        return delta;
    }
    Score evaluate(const Solution& s) const override {
        std::unordered_map<int64_t, int64_t> acc;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i) {
            if (!filter(s, i)) continue;
            int64_t k = key(s, i);
            acc[k] = wrap_add(acc[k], value(s, i));
        }
        Score total;
        for (auto& kv : acc) total = total + compute(kv.first, kv.second);
        return total;
    }
    size_t match_count(const Solution& s) const override {
        std::unordered_set<int64_t> keys;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) keys.insert(key(s, i));
        return keys.size();
    }
    Score initialize(const Solution& s) override {
        reset();
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) insert_entity(s, i);
        changed_groups.clear();
        cached_scores.clear();
        Score total;
        for (size_t g = 0; g < groups.size(); ++g) {
            Score sc = groups[g].count == 0 ? Score::zero() : compute(groups[g].key, groups[g].acc);
            replace_cached(g, sc);
            total = total + sc;
        }
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        changed_groups.clear();
        if (!source.assert_localizes(d, name)) return refresh_changed();
        if (e < count(s) && filter(s, e)) insert_entity(s, e);
        return refresh_changed();
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        (void)s;
        changed_groups.clear();
        if (!source.assert_localizes(d, name)) return refresh_changed();
        retract_entity(e);
        return refresh_changed();
    }
    void reset() override {
        groups.clear();
        group_of_key.clear();
        entity_groups.clear();
        entity_retractions.clear();
        changed_groups.clear();
        cached_scores.clear();
    }
};

// ---- complemented grouped constraint (constraint/complemented/{state,helpers,incremental}.rs; stream
// grouped_stream/base.rs:142-200 `.complement(B, key_b, default)`) -------------------------------------------------------------
// Groups the A entities by an optional key with a sum / count collector, then scores EVERY B row: with its key's grouped
// result when a group exists, else with default(b).  B rows are indexed by key (several rows may share one).
struct ComplementedGroupedConstraint : Constraint {
    Impact impact;
    ChangeSource a_source, b_source;
    CountFn a_count, b_count;
    Key1 key_a;  // NONE = the key function returns None (entity skipped)
    Key1 key_b;
    Value1 value;      // collector extract (count: 1)
    Value1 default_b;  // default result of a B row without a group
    GroupWeight weight;

    struct Group {
        int64_t acc = 0;
        size_t count = 0;
    };
    std::unordered_map<int64_t, Group> groups;
    std::unordered_map<size_t, int64_t> entity_groups, entity_retractions;
    std::unordered_map<int64_t, std::vector<size_t>> b_by_key;
    std::unordered_map<size_t, int64_t> b_index_to_key;

    Score compute(int64_t k, int64_t r) const { return apply_impact(impact, weight(k, r)); }
    Score b_score_for_index(const Solution& s, int64_t k, size_t b) const {  // helpers.rs:67-82
        if (b >= b_count(s)) return Score::zero();
        auto it = groups.find(k);
        return it != groups.end() ? compute(k, it->second.acc) : compute(k, default_b(s, b));
    }
    Score key_score(const Solution& s, int64_t k) const {  // helpers.rs:84-91
        auto it = b_by_key.find(k);
        if (it == b_by_key.end()) return Score::zero();
        Score t;
        for (size_t b : it->second) t = t + b_score_for_index(s, k, b);
        return t;
    }
    Score insert_entity(const Solution& s, size_t idx) {  // helpers.rs:29-65
        int64_t k = key_a(s, idx);
        if (k == NONE) return Score::zero();
        int64_t v = value(s, idx);
        Score old = key_score(s, k);
        Group& g = groups[k];
        g.acc = wrap_add(g.acc, v);
        g.count += 1;
        entity_groups[idx] = k;
        entity_retractions[idx] = v;
        return key_score(s, k) - old;
    }
    Score retract_entity(const Solution& s, size_t idx) {  // helpers.rs:137-165
        auto eg = entity_groups.find(idx);
        if (eg == entity_groups.end()) return Score::zero();
        int64_t k = eg->second;
        entity_groups.erase(eg);
        auto er = entity_retractions.find(idx);
        if (er == entity_retractions.end()) return Score::zero();
        int64_t v = er->second;
        entity_retractions.erase(er);
        Score old = key_score(s, k);
        auto g = groups.find(k);
        if (g == groups.end()) return Score::zero();
        g->second.acc = wrap_sub(g->second.acc, v);
        g->second.count = g->second.count > 0 ? g->second.count - 1 : 0;
        if (g->second.count == 0) groups.erase(g);
        return key_score(s, k) - old;
    }
    void remove_index_from_key_bucket(int64_t k, size_t idx) {  // helpers.rs:93-108
        auto it = b_by_key.find(k);
        if (it == b_by_key.end()) return;
        auto& v = it->second;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == idx) {
                v[i] = v.back();
                v.pop_back();
                break;
            }
        if (v.empty()) b_by_key.erase(it);
    }
    Score insert_b(const Solution& s, size_t b) {  // helpers.rs:117-124
        if (b >= b_count(s)) return Score::zero();
        int64_t k = key_b(s, b);
        auto old = b_index_to_key.find(b);
        if (old != b_index_to_key.end()) remove_index_from_key_bucket(old->second, b);
        b_index_to_key[b] = k;
        b_by_key[k].push_back(b);
        return b_score_for_index(s, k, b);
    }
    Score retract_b(const Solution& s, size_t b) {  // helpers.rs:126-134
        auto it = b_index_to_key.find(b);
        if (it == b_index_to_key.end()) return Score::zero();
        int64_t k = it->second;
        b_index_to_key.erase(it);
        Score delta = -b_score_for_index(s, k, b);
        remove_index_from_key_bucket(k, b);
        return delta;
    }
    Score evaluate(const Solution& s) const override {  // incremental.rs:31-52
        std::unordered_map<int64_t, int64_t> acc;
        size_t na = a_count(s), nb = b_count(s);
        for (size_t i = 0; i < na; ++i) {
            int64_t k = key_a(s, i);
            if (k == NONE) continue;
            acc[k] = wrap_add(acc.count(k) ? acc[k] : 0, value(s, i));
        }
        Score total;
        for (size_t b = 0; b < nb; ++b) {
            int64_t k = key_b(s, b);
            auto it = acc.find(k);
            total = total + (it != acc.end() ? compute(k, it->second) : compute(k, default_b(s, b)));
        }
        return total;
    }
    size_t match_count(const Solution& s) const override { return b_count(s); }  // incremental.rs:54-57
    Score initialize(const Solution& s) override {  // incremental.rs:59-83
        reset();
        size_t na = a_count(s), nb = b_count(s);
        for (size_t b = 0; b < nb; ++b) {
            int64_t k = key_b(s, b);
            b_by_key[k].push_back(b);
            b_index_to_key[b] = k;
        }
        Score total;
        for (size_t b = 0; b < nb; ++b) total = total + compute(key_b(s, b), default_b(s, b));
        for (size_t i = 0; i < na; ++i) total = total + insert_entity(s, i);
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {  // incremental.rs:85-104
        bool a_changed = a_source.assert_localizes(d, name), b_changed = b_source.assert_localizes(d, name);
        Score total;
        if (a_changed && e < a_count(s)) total = total + insert_entity(s, e);
        if (b_changed) total = total + insert_b(s, e);
        return total;
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {  // incremental.rs:106-125
        bool a_changed = a_source.assert_localizes(d, name), b_changed = b_source.assert_localizes(d, name);
        Score total;
        if (a_changed) total = total + retract_entity(s, e);
        if (b_changed) total = total + retract_b(s, e);
        return total;
    }
    void reset() override {
        groups.clear();
        entity_groups.clear();
        entity_retractions.clear();
        b_by_key.clear();
        b_index_to_key.clear();
    }
};

// ---- consecutive_runs collector (stream/collector/runs.rs:11-229) ------------------------------------------------------------
// Points (i64) with multiplicity in an ordered map; the result lists the maximal runs of consecutive unique points, each with
// its unique-point count and its item count (duplicates included).
struct Run {
    int64_t start, end;
    size_t point_count, item_count;
};
struct Runs {
    std::vector<Run> runs;
    size_t point_count = 0, item_count = 0;
};
struct RunsAccumulator {  // runs.rs:126-172
    std::map<int64_t, size_t> points;
    size_t item_count = 0;
    void accumulate(int64_t v) {
        ++points[v];
        ++item_count;
    }
    void retract(int64_t v) {
        auto it = points.find(v);
        if (it == points.end()) return;
        it->second = it->second > 0 ? it->second - 1 : 0;
        item_count = item_count > 0 ? item_count - 1 : 0;
        if (it->second == 0) points.erase(it);
    }
    Runs finish() const {  // runs_from_counts_and_item_count (:178-229)
        Runs r;
        r.point_count = points.size();
        r.item_count = item_count;
        bool open = false;
        int64_t start = 0, prev = 0;
        size_t pc = 0, ic = 0;
        for (auto& kv : points) {
            if (!open) {
                open = true;
                start = prev = kv.first;
                pc = 1;
                ic = kv.second;
            } else if (prev != INT64_MAX && prev + 1 == kv.first) {  // checked_add(1) == Some(point)
                prev = kv.first;
                pc += 1;
                ic += kv.second;
            } else {
                r.runs.push_back({start, prev, pc, ic});
                start = prev = kv.first;
                pc = 1;
                ic = kv.second;
            }
        }
        if (open) r.runs.push_back({start, prev, pc, ic});
        return r;
    }
    void reset() {
        points.clear();
        item_count = 0;
    }
};

// ---- indexed_presence collector (stream/collector/indexed_presence.rs:1-147) ---------------------------------------------------
// The same ordered point -> multiplicity map as consecutive_runs, read as a presence set: membership, unique-point and item
// counts, runs of present points, runs of ABSENT points inside a horizon (runs_from_counts over a map of the missing points, one
// item each), counts inside a half-open range.  Oracle only so far: the device keeps the same per-(value, point) count table for
// SF_C_RUNS_VALUE, a presence-scored constraint kind is not declared yet.
struct IndexedPresenceAccumulator {  // indexed_presence.rs:101-147
    RunsAccumulator acc;
    void accumulate(int64_t v) { acc.accumulate(v); }
    void retract(int64_t v) { acc.retract(v); }
    void reset() {
        acc.points.clear();
        acc.item_count = 0;
    }
    bool contains(int64_t index) const { return acc.points.count(index) != 0; }  // :49-51
    size_t count() const { return acc.points.size(); }                            // :54-56
    size_t item_count() const { return acc.item_count; }                          // :59-61
    bool is_empty() const { return acc.points.empty(); }                          // :64-66
    Runs runs() const { return acc.finish(); }                                    // :68-70
    Runs complement_runs(int64_t start, int64_t end) const {                      // :72-90
        RunsAccumulator missing;
        if (start >= end) return missing.finish();
        for (int64_t index = start; index < end;) {
            if (!acc.points.count(index)) missing.accumulate(index);
            if (index == INT64_MAX) break;  // checked_add(1) == None
            ++index;
        }
        return missing.finish();
    }
    size_t count_in(int64_t start, int64_t end) const {  // :92-97
        if (start >= end) return 0;
        size_t n = 0;
        for (auto it = acc.points.lower_bound(start); it != acc.points.end() && it->first < end; ++it) ++n;
        return n;
    }
    bool any_in(int64_t start, int64_t end) const { return count_in(start, end) > 0; }  // :99-101
};

// for_each(E).filter(f).group_by(key, consecutive_runs(point)).penalize(w(key, runs)): the grouped node
// (constraint/grouped/state.rs:43-247, scorer.rs:46-152) with the runs accumulator per group
struct GroupedRunsConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Filter1 filter;
    Key1 key;
    Value1 point;
    std::function<Score(int64_t, const Runs&)> weight;
    // set instead of `weight` for group_by(key, indexed_presence(point)): the same accumulator read as a presence set
    std::function<Score(int64_t, const IndexedPresenceAccumulator&)> presence_weight;
    Score weigh(int64_t k, const RunsAccumulator& acc) const {
        if (presence_weight) {
            IndexedPresenceAccumulator p;
            p.acc = acc;
            return presence_weight(k, p);
        }
        return weight(k, acc.finish());
    }

    struct Group {
        int64_t key;
        RunsAccumulator acc;
        size_t count;
    };
    std::vector<Group> groups;
    std::unordered_map<int64_t, size_t> group_of_key;
    std::unordered_map<size_t, size_t> entity_groups;
    std::unordered_map<size_t, int64_t> entity_retractions;
    std::vector<size_t> changed_groups;
    std::vector<Score> cached_scores;

    Score compute(const Group& g) const { return g.count == 0 ? Score::zero() : apply_impact(impact, weigh(g.key, g.acc)); }
    void mark_changed(size_t g) {
        for (size_t c : changed_groups)
            if (c == g) return;
        changed_groups.push_back(g);
    }
    size_t group_id_for_key(int64_t k) {
        auto it = group_of_key.find(k);
        if (it != group_of_key.end()) return it->second;
        size_t g = groups.size();
        groups.push_back({k, RunsAccumulator(), 0});
        group_of_key[k] = g;
        return g;
    }
    void insert_entity(const Solution& s, size_t idx) {  // state.rs:216-229
        size_t g = group_id_for_key(key(s, idx));
        if (groups[g].count == 0) groups[g].acc.reset();
        int64_t v = point(s, idx);
        groups[g].acc.accumulate(v);
        groups[g].count += 1;
        entity_groups[idx] = g;
        entity_retractions[idx] = v;
        mark_changed(g);
    }
    void retract_entity(size_t idx) {  // state.rs:231-244
        auto eg = entity_groups.find(idx);
        if (eg == entity_groups.end()) return;
        size_t g = eg->second;
        entity_groups.erase(eg);
        auto er = entity_retractions.find(idx);
        if (er == entity_retractions.end()) return;
        int64_t v = er->second;
        entity_retractions.erase(er);
        groups[g].acc.retract(v);
        groups[g].count = groups[g].count > 0 ? groups[g].count - 1 : 0;
        mark_changed(g);
    }
    Score replace_cached(size_t slot, const Score& sc) {  // scorer.rs:145-152
        while (cached_scores.size() <= slot) cached_scores.push_back(Score::zero());
        Score prev = cached_scores[slot];
        cached_scores[slot] = sc;
        return sc - prev;
    }
    Score refresh_changed() {  // scorer.rs:89-101
        Score delta;
        for (size_t g : changed_groups)
            if (g < groups.size()) delta = delta + replace_cached(g, compute(groups[g]));
        return delta;
    }
    Score evaluate(const Solution& s) const override {
        std::map<int64_t, RunsAccumulator> acc;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) acc[key(s, i)].accumulate(point(s, i));
        Score total;
        for (auto& kv : acc) total = total + apply_impact(impact, weigh(kv.first, kv.second));
        return total;
    }
    size_t match_count(const Solution& s) const override {
        std::unordered_set<int64_t> keys;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) keys.insert(key(s, i));
        return keys.size();
    }
    Score initialize(const Solution& s) override {
        reset();
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) insert_entity(s, i);
        changed_groups.clear();
        cached_scores.clear();
        Score total;
        for (size_t g = 0; g < groups.size(); ++g) {
            Score sc = compute(groups[g]);
            replace_cached(g, sc);
            total = total + sc;
        }
        return total;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        changed_groups.clear();
        if (!source.assert_localizes(d, name)) return refresh_changed();
        if (e < count(s) && filter(s, e)) insert_entity(s, e);
        return refresh_changed();
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        (void)s;
        changed_groups.clear();
        if (!source.assert_localizes(d, name)) return refresh_changed();
        retract_entity(e);
        return refresh_changed();
    }
    void reset() override {
        groups.clear();
        group_of_key.clear();
        entity_groups.clear();
        entity_retractions.clear();
        changed_groups.clear();
        cached_scores.clear();
    }
};

// ---- load_balance collector (stream/collector/load_balance.rs:100-226) -------------------------------------
// Incremental squared deviation of the per-key loads: integral = sum(x^2), fraction numerator = -(sum x)^2, kept by
// the reference's four-term update; unfairness = round(sqrt(fraction / n + integral)) in f64.
struct LoadBalanceAccumulator {
    std::unordered_map<int64_t, size_t> item_counts;
    std::unordered_map<int64_t, int64_t> loads;
    int64_t sum = 0;
    int64_t squared_deviation_integral = 0;
    int64_t squared_deviation_fraction_numerator = 0;

    void update_squared_deviation(int64_t old_value, int64_t new_value) {  // :144-163
        int64_t term1 = wrap_sub(wrap_mul(new_value, new_value), wrap_mul(old_value, old_value));
        int64_t sum_others = wrap_mul(2, wrap_sub(sum, old_value));
        int64_t new_sum = wrap_add(wrap_sub(sum, old_value), new_value);
        int64_t sum_diff = wrap_sub(sum, new_sum);
        int64_t term3 = wrap_sub(wrap_mul(new_sum, new_sum), wrap_mul(sum, sum));
        int64_t term4 = wrap_mul(2, wrap_sub(wrap_mul(old_value, sum), wrap_mul(new_value, new_sum)));
        int64_t fraction_delta = wrap_add(wrap_add(wrap_mul(sum_others, sum_diff), term3), term4);
        squared_deviation_integral = wrap_add(squared_deviation_integral, term1);
        squared_deviation_fraction_numerator = wrap_add(squared_deviation_fraction_numerator, fraction_delta);
    }
    void add_to_metric(int64_t key, int64_t diff) {  // :123-132
        auto it = loads.find(key);
        int64_t old_value = it == loads.end() ? 0 : it->second;
        int64_t new_value = wrap_add(old_value, diff);
        if (old_value != new_value) {
            loads[key] = new_value;
            update_squared_deviation(old_value, new_value);
            sum = wrap_add(sum, diff);
        }
    }
    void reset_metric(int64_t key) {  // :134-141
        auto it = loads.find(key);
        if (it == loads.end()) return;
        int64_t old_value = it->second;
        loads.erase(it);
        if (old_value != 0) {
            update_squared_deviation(old_value, 0);
            sum = wrap_sub(sum, old_value);
        }
    }
    int64_t unfairness() const {  // compute_unfairness :167-184 (f64::round = half away from zero = llround)
        size_t n = item_counts.size();
        if (n == 0) return 0;
        double tmp = n == 1 ? (double)squared_deviation_fraction_numerator + (double)squared_deviation_integral
                            : (double)squared_deviation_fraction_numerator / (double)n + (double)squared_deviation_integral;
        if (!(tmp >= 0.0)) return 0;  // sqrt of a negative radicand is NaN and `NaN as i64` is 0 in Rust
        return (int64_t)std::round(std::sqrt(tmp));
    }
    void accumulate(int64_t key, int64_t metric) {  // :192-201
        if (metric == 0) return;
        item_counts[key] += 1;
        add_to_metric(key, metric);
    }
    void retract(int64_t key, int64_t metric) {  // :203-218
        if (metric == 0) return;
        auto it = item_counts.find(key);
        if (it == item_counts.end() || it->second == 0) return;
        it->second -= 1;
        if (it->second == 0) {
            item_counts.erase(it);
            reset_metric(key);
        } else {
            add_to_metric(key, -metric);
        }
    }
    void reset() {  // :228-234
        item_counts.clear();
        loads.clear();
        sum = 0;
        squared_deviation_integral = 0;
        squared_deviation_fraction_numerator = 0;
    }
};

// for_each(E).filter(f).group_by(load_balance(key, metric)).penalize(w(unfairness)): the grouped node
// (constraint/grouped/state.rs:216-244, scorer.rs:89-152) with the unit group key and the load-balance accumulator.
struct LoadBalanceConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Filter1 filter;
    Key1 key;
    Value1 metric;
    std::function<Score(int64_t unfairness)> weight;

    LoadBalanceAccumulator acc;
    size_t group_count = 0;  // entities in the (single) group: an empty group scores zero (scorer.rs:89-101)
    std::unordered_map<size_t, std::pair<int64_t, int64_t>> entity_retractions;
    Score cached;

    Score current() const { return group_count == 0 ? Score::zero() : apply_impact(impact, weight(acc.unfairness())); }
    Score refresh() {
        Score sc = current();
        Score d = sc - cached;
        cached = sc;
        return d;
    }
    Score evaluate(const Solution& s) const override {
        LoadBalanceAccumulator a;
        size_t n = count(s), members = 0;
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) {
                a.accumulate(key(s, i), metric(s, i));
                ++members;
            }
        return members == 0 ? Score::zero() : apply_impact(impact, weight(a.unfairness()));
    }
    size_t match_count(const Solution& s) const override {
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) return 1;  // one group
        return 0;
    }
    Score initialize(const Solution& s) override {
        reset();
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i)) {
                int64_t k = key(s, i), m = metric(s, i);
                acc.accumulate(k, m);
                entity_retractions[i] = {k, m};
                ++group_count;
            }
        cached = current();
        return cached;
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name)) return Score::zero();
        if (e < count(s) && filter(s, e)) {
            int64_t k = key(s, e), m = metric(s, e);
            acc.accumulate(k, m);
            entity_retractions[e] = {k, m};
            ++group_count;
        }
        return refresh();
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        (void)s;
        if (!source.assert_localizes(d, name)) return Score::zero();
        auto it = entity_retractions.find(e);
        if (it != entity_retractions.end()) {
            acc.retract(it->second.first, it->second.second);
            entity_retractions.erase(it);
            group_count = group_count > 0 ? group_count - 1 : 0;
        }
        return refresh();
    }
    void reset() override {
        acc.reset();
        group_count = 0;
        entity_retractions.clear();
        cached = Score::zero();
    }
};

// ---- ConstraintSet tuple fold (api/constraint_set/incremental.rs:339-407) ----
// ---- INDEXED CPU baseline for predicate joins (not a reference node) ---------------------------------------------------------
// The reference evaluates `left.id < right.id && partner(left, right) && equal values` as a cross-join with a constant key: every
// insert tests all n rows of the other side (CrossBiConstraint above, the dense-faithful baseline).  A CPU implementation that
// indexes the join by its partner relation does O(partners) work per insert instead; SURVEY.md §7 asks for both numbers.  Same
// scores by construction (each matched pair counted once, frozen weight = `weight`); stateless.
struct PartnerEqualConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Key1 value;                            // NONE = unassigned
    std::vector<uint32_t> poff, pn;        // symmetric partner CSR
    Score weight;
    size_t matches_of(const Solution& s, size_t e) const {
        int64_t v = value(s, e);
        if (v == NONE) return 0;
        size_t c = 0;
        for (uint32_t p = poff[e]; p < poff[e + 1]; ++p) c += value(s, pn[p]) == v ? 1 : 0;
        return c;
    }
    Score times(size_t c) const {
        Score r;
        for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = wrap_mul(weight.v[i], (int64_t)c);
        return apply_impact(impact, r);
    }
    Score evaluate(const Solution& s) const override { return times(match_count(s)); }
    size_t match_count(const Solution& s) const override {
        size_t c = 0, n = count(s);
        for (size_t e = 0; e < n; ++e) c += matches_of(s, e);
        return c / 2;  // every pair is seen from both ends
    }
    Score initialize(const Solution& s) override { return evaluate(s); }
    Score on_insert(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name) || e >= count(s)) return Score::zero();
        return times(matches_of(s, e));
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {
        if (!source.assert_localizes(d, name) || e >= count(s)) return Score::zero();
        return -times(matches_of(s, e));
    }
    void reset() override {}
};

// ---- BalanceConstraint (constraint/balance.rs:83-372): base_score x population standard deviation of the per-key entity
// COUNTS, one global statistic; `Score::multiply(f64)` rounds every level half away from zero (score/macros.rs:61-63) ----------
struct BalanceConstraint : Constraint {
    Impact impact;
    ChangeSource source;
    CountFn count;
    Filter1 filter;
    Key1 key;  // NONE = the key function returns None (entity skipped)
    Score base_score;
    std::unordered_map<int64_t, int64_t> counts;
    std::unordered_map<size_t, int64_t> entity_keys;
    int64_t group_count = 0, total_count = 0, sum_squared = 0;

    static Score multiply(const Score& s, double f) {
        Score r;
        for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = (int64_t)std::round((double)s.v[i] * f);
        return r;
    }
    static double std_dev_of(int64_t groups, int64_t total, int64_t sum_sq) {  // compute_std_dev (:162-173)
        if (groups == 0) return 0.0;
        double n = (double)groups, mean = (double)total / n;
        double variance = ((double)sum_sq / n) - (mean * mean);
        return variance <= 0.0 ? 0.0 : std::sqrt(variance);
    }
    Score compute_score() const { return apply_impact(impact, multiply(base_score, std_dev_of(group_count, total_count, sum_squared))); }
    Score evaluate(const Solution& s) const override {  // (:213-236)
        std::unordered_map<int64_t, int64_t> c;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i) && key(s, i) != NONE) ++c[key(s, i)];
        if (c.empty()) return Score::zero();
        int64_t total = 0, sq = 0;
        for (auto& kv : c) total += kv.second, sq += kv.second * kv.second;
        return apply_impact(impact, multiply(base_score, std_dev_of((int64_t)c.size(), total, sq)));
    }
    size_t match_count(const Solution& s) const override {  // groups deviating from the mean by more than 0.5 (:238-266)
        std::unordered_map<int64_t, int64_t> c;
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i)
            if (filter(s, i) && key(s, i) != NONE) ++c[key(s, i)];
        if (c.empty()) return 0;
        int64_t total = 0;
        for (auto& kv : c) total += kv.second;
        double mean = (double)total / (double)c.size();
        size_t m = 0;
        for (auto& kv : c) m += std::fabs((double)kv.second - mean) > 0.5 ? 1 : 0;
        return m;
    }
    Score initialize(const Solution& s) override {  // (:268-292)
        reset();
        size_t n = count(s);
        for (size_t i = 0; i < n; ++i) {
            if (!filter(s, i) || key(s, i) == NONE) continue;
            int64_t k = key(s, i), oc = counts.count(k) ? counts[k] : 0, nc = oc + 1;
            counts[k] = nc;
            entity_keys[i] = k;
            if (oc == 0) ++group_count;
            ++total_count;
            sum_squared += nc * nc - oc * oc;
        }
        return compute_score();
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {  // (:294-329)
        if (!source.assert_localizes(d, name)) return Score::zero();
        if (e >= count(s) || !filter(s, e) || key(s, e) == NONE) return Score::zero();
        Score old = compute_score();
        int64_t k = key(s, e), oc = counts.count(k) ? counts[k] : 0, nc = oc + 1;
        counts[k] = nc;
        entity_keys[e] = k;
        if (oc == 0) ++group_count;
        ++total_count;
        sum_squared += nc * nc - oc * oc;
        return compute_score() - old;
    }
    Score on_retract(const Solution& s, size_t e, size_t d) override {  // (:331-366)
        if (!source.assert_localizes(d, name)) return Score::zero();
        if (e >= count(s)) return Score::zero();
        auto it = entity_keys.find(e);
        if (it == entity_keys.end()) return Score::zero();
        int64_t k = it->second;
        entity_keys.erase(it);
        Score old = compute_score();
        int64_t oc = counts.count(k) ? counts[k] : 0;
        if (oc == 0) return Score::zero();
        int64_t nc = oc - 1;
        if (nc == 0) {
            counts.erase(k);
            --group_count;
        } else
            counts[k] = nc;
        --total_count;
        sum_squared += nc * nc - oc * oc;
        return compute_score() - old;
    }
    void reset() override {
        counts.clear();
        entity_keys.clear();
        group_count = total_count = sum_squared = 0;
    }
};

// ---- ListPrecedenceMakespanConstraint (constraint/list_precedence.rs:13-707) ---------------------------------------------------
// Nodes = list elements with a duration; edges = the fixed successor relation plus every pair of consecutive elements of
// every owner's list (edges are counted: one that is both fixed and on a route exists once in the graph).  Score =
// HardSoft(-(invalid fixed edges + invalid list items + wrong-owner items + assignment penalty + cycle penalty), -makespan),
// makespan = the longest duration-weighted path; a cyclic graph costs node_count hard and has makespan 0.  The incremental
// state (earliest starts refreshed over the descendants of changed edges, the cached cycle, the full Kahn rebuild) follows the
// reference line by line so the GraphRefreshKind sequence of its tests can be pinned.  `hard` / `soft` = the score of one
// unit of penalty / makespan (the reference fixes HardSoftScore::of(1, 0) / of(0, 1)).
struct ListPrecedenceConstraint : Constraint {
    using Edge = std::pair<size_t, size_t>;
    enum class Refresh { Skipped, Incremental, CycleDetected, CycleRecovered, Full };
    struct RouteChange {
        std::vector<Edge> added, removed;
        bool empty() const { return added.empty() && removed.empty(); }
    };
    struct Snapshot {
        std::vector<size_t> elements;
        std::vector<Edge> edges;
        size_t invalid = 0, violation = 0;
    };
    struct State {  // ListPrecedenceState (:188-210)
        size_t node_count = 0;
        std::vector<int64_t> durations;
        std::vector<std::vector<std::pair<size_t, size_t>>> edge_counts;
        std::vector<std::vector<size_t>> successors, predecessors;
        std::vector<size_t> assigned_counts;
        std::vector<std::vector<size_t>> owner_elements;
        std::vector<std::vector<Edge>> owner_edges;
        std::vector<size_t> owner_invalid, owner_violation;
        size_t invalid_fixed_edges = 0, owner_invalid_total = 0, owner_violation_total = 0, assignment_penalty = 0, cycle_penalty = 0;
        std::vector<Edge> cycle_added_edges;
        std::vector<int64_t> earliest, finishes;
        size_t hard_penalty = 0;
        int64_t makespan = 0;
        size_t last_visited = 0;  // GraphRefreshKind::Incremental { visited }

        State() = default;
        State(size_t n, size_t owners, std::vector<int64_t> dur)  // (:268-292)
            : node_count(n), durations(std::move(dur)), edge_counts(n), successors(n), predecessors(n), assigned_counts(n, 0),
              owner_elements(owners), owner_edges(owners), owner_invalid(owners, 0), owner_violation(owners, 0), assignment_penalty(n),
              earliest(n, 0), finishes(n, 0) {}
        static int64_t sat_add(int64_t a, int64_t b) {
            int64_t r;
            return __builtin_add_overflow(a, b, &r) ? (b > 0 ? INT64_MAX : INT64_MIN) : r;
        }
        static size_t penalty_of(size_t count) { return count == 0 ? 1 : (count == 1 ? 0 : count - 1); }  // (:690-696)
        static void remove_node(std::vector<size_t>& v, size_t node) {  // swap_remove of the first match (:702-707)
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i] == node) {
                    v[i] = v.back();
                    v.pop_back();
                    return;
                }
        }
        bool add_edge(Edge e) {  // (:294-307)
            for (auto& tc : edge_counts[e.first])
                if (tc.first == e.second) {
                    ++tc.second;
                    return false;
                }
            edge_counts[e.first].push_back({e.second, 1});
            successors[e.first].push_back(e.second);
            predecessors[e.second].push_back(e.first);
            return true;
        }
        bool remove_edge(Edge e) {  // (:309-326)
            auto& ec = edge_counts[e.first];
            for (size_t i = 0; i < ec.size(); ++i)
                if (ec[i].first == e.second) {
                    if (--ec[i].second == 0) {
                        ec[i] = ec.back();
                        ec.pop_back();
                        remove_node(successors[e.first], e.second);
                        remove_node(predecessors[e.second], e.first);
                        return true;
                    }
                    return false;
                }
            return false;
        }
        bool contains_edge(Edge e) const {  // (:683-688)
            for (auto& tc : edge_counts[e.first])
                if (tc.first == e.second && tc.second > 0) return true;
            return false;
        }
        void adjust_assignment(size_t node, size_t new_count) {  // (:328-341)
            size_t old_count = assigned_counts[node];
            if (old_count == new_count) return;
            assignment_penalty = assignment_penalty - penalty_of(old_count) + penalty_of(new_count);
            assigned_counts[node] = new_count;
        }
        RouteChange replace_owner_route(size_t owner, Snapshot snap) {  // (:395-481, diff_owner_assignments / diff_owner_edges)
            RouteChange change;
            std::vector<size_t> old_elements = std::move(owner_elements[owner]);
            {
                std::map<size_t, std::pair<size_t, size_t>> counts;
                for (size_t n : old_elements) ++counts[n].first;
                for (size_t n : snap.elements) ++counts[n].second;
                for (auto& kv : counts) {
                    if (kv.second.first == kv.second.second) continue;
                    size_t cur = assigned_counts[kv.first];
                    size_t upd = (cur >= kv.second.first ? cur - kv.second.first : 0) + kv.second.second;
                    adjust_assignment(kv.first, upd);
                }
            }
            owner_elements[owner] = std::move(snap.elements);
            std::vector<Edge> old_edges = std::move(owner_edges[owner]);
            {
                std::map<Edge, std::pair<size_t, size_t>> counts;
                for (auto& e : old_edges) ++counts[e].first;
                for (auto& e : snap.edges) ++counts[e].second;
                for (auto& kv : counts) {  // the reference iterates a HashMap: the ORDER of the change's edges is unspecified there
                    if (kv.second.first > kv.second.second) {
                        for (size_t i = 0; i < kv.second.first - kv.second.second; ++i)
                            if (remove_edge(kv.first)) change.removed.push_back(kv.first);
                    } else if (kv.second.second > kv.second.first) {
                        for (size_t i = 0; i < kv.second.second - kv.second.first; ++i)
                            if (add_edge(kv.first)) change.added.push_back(kv.first);
                    }
                }
            }
            owner_edges[owner] = std::move(snap.edges);
            owner_invalid_total = owner_invalid_total - owner_invalid[owner] + snap.invalid;
            owner_invalid[owner] = snap.invalid;
            owner_violation_total = owner_violation_total - owner_violation[owner] + snap.violation;
            owner_violation[owner] = snap.violation;
            return change;
        }
        int64_t max_finish() const {  // (:634-636)
            int64_t m = 0;
            bool any = false;
            for (int64_t f : finishes) m = any ? std::max(m, f) : f, any = true;
            return any ? m : 0;
        }
        void mark_cyclic_without_cache() {  // (:597-603)
            std::fill(earliest.begin(), earliest.end(), 0);
            std::fill(finishes.begin(), finishes.end(), 0);
            cycle_added_edges.clear();
            cycle_penalty = node_count;
            makespan = 0;
        }
        void rebuild_graph_summary() {  // Kahn (:553-589)
            std::vector<size_t> indegree(node_count);
            for (size_t i = 0; i < node_count; ++i) indegree[i] = predecessors[i].size();
            std::vector<int64_t> e(node_count, 0), f(node_count, 0);
            std::deque<size_t> ready;
            for (size_t i = 0; i < node_count; ++i)
                if (indegree[i] == 0) ready.push_back(i);
            size_t processed = 0;
            int64_t mk = 0;
            while (!ready.empty()) {
                size_t node = ready.front();
                ready.pop_front();
                ++processed;
                int64_t fin = sat_add(e[node], durations[node]);
                f[node] = fin;
                mk = std::max(mk, fin);
                for (size_t s : successors[node]) {
                    e[s] = std::max(e[s], fin);
                    if (--indegree[s] == 0) ready.push_back(s);
                }
            }
            if (processed < node_count)
                mark_cyclic_without_cache();
            else {
                earliest = std::move(e);
                finishes = std::move(f);
                cycle_penalty = 0;
                cycle_added_edges.clear();
                makespan = mk;
            }
        }
        bool reaches(size_t start, size_t target, size_t visit_id, std::vector<size_t>& visited, std::vector<size_t>& stack) const {  // (:653-681)
            if (start == target) return true;
            stack.clear();
            stack.push_back(start);
            while (!stack.empty()) {
                size_t node = stack.back();
                stack.pop_back();
                if (visited[node] == visit_id) continue;
                visited[node] = visit_id;
                for (size_t s : successors[node]) {
                    if (s == target) return true;
                    if (visited[s] != visit_id) stack.push_back(s);
                }
            }
            return false;
        }
        bool added_edges_introduce_cycle(const std::vector<Edge>& added) const {  // (:638-651)
            if (added.empty()) return false;
            std::vector<size_t> visited(node_count, 0), stack;
            for (size_t i = 0; i < added.size(); ++i)
                if (reaches(added[i].second, added[i].first, i + 1, visited, stack)) return true;
            return false;
        }
        bool recover_cached_cycle(const RouteChange& change) {  // (:605-620)
            if (cycle_added_edges.empty() || !change.added.empty()) return false;
            for (auto& e : cycle_added_edges)
                if (contains_edge(e)) return false;
            cycle_penalty = 0;
            cycle_added_edges.clear();
            makespan = max_finish();
            return true;
        }
        void replace_earliest(size_t node, int64_t ne) {  // (:622-632)
            int64_t old_finish = finishes[node];
            earliest[node] = ne;
            int64_t nf = sat_add(ne, durations[node]);
            finishes[node] = nf;
            if (nf >= makespan)
                makespan = nf;
            else if (old_finish == makespan)
                makespan = max_finish();
        }
        Refresh refresh_graph_after_route_change(const RouteChange& change) {  // (:494-551)
            if (change.empty()) return Refresh::Skipped;
            if (cycle_penalty > 0) {
                if (recover_cached_cycle(change)) return Refresh::CycleRecovered;
                rebuild_graph_summary();
                return Refresh::Full;
            }
            if (added_edges_introduce_cycle(change.added)) {
                if (change.removed.empty()) {  // mark_cyclic_from_route_change: the acyclic earliest / finishes stay cached
                    cycle_added_edges = change.added;
                    cycle_penalty = node_count;
                    makespan = 0;
                } else
                    mark_cyclic_without_cache();
                return Refresh::CycleDetected;
            }
            std::vector<char> queued(node_count, 0);
            std::deque<size_t> queue;
            auto seed = [&](size_t node) {
                if (node < node_count && !queued[node]) {
                    queued[node] = 1;
                    queue.push_back(node);
                }
            };
            for (auto& e : change.added) seed(e.second);
            for (auto& e : change.removed) seed(e.second);
            size_t visited = 0;
            while (!queue.empty()) {
                size_t node = queue.front();
                queue.pop_front();
                queued[node] = 0;
                ++visited;
                int64_t ne = 0;
                bool any = false;
                for (size_t p : predecessors[node]) {
                    int64_t v = sat_add(earliest[p], durations[p]);
                    ne = any ? std::max(ne, v) : v;
                    any = true;
                }
                if (ne == earliest[node]) continue;
                replace_earliest(node, ne);
                for (size_t s : successors[node])
                    if (!queued[s]) {
                        queued[s] = 1;
                        queue.push_back(s);
                    }
            }
            makespan = max_finish();
            last_visited = visited;
            return Refresh::Incremental;
        }
        void refresh_penalty() {  // refresh_score_from_cached_graph (:483-492)
            hard_penalty = invalid_fixed_edges + owner_invalid_total + owner_violation_total + assignment_penalty + cycle_penalty;
        }
    };

    size_t list_descriptor = 0;
    std::function<size_t(const Solution&)> node_count, owner_count;
    std::function<int64_t(const Solution&, size_t)> node_duration;
    std::function<void(const Solution&, size_t, std::vector<size_t>&)> fixed_successors;
    std::function<size_t(const Solution&, size_t)> list_len;
    std::function<int64_t(const Solution&, size_t, size_t)> list_get;        // NONE = None
    std::function<int64_t(const Solution&, size_t)> expected_owner;          // may be empty; NONE = no expectation
    Score hard = Score::of(1, 0), soft = Score::of(0, 1);
    bool has_state = false;
    State state;

    Score score_of(const State& st) const {  // HardSoftScore::of(-hard_penalty, makespan.saturating_neg())
        Score r;
        for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = -(hard.v[i] * (int64_t)st.hard_penalty + soft.v[i] * st.makespan);
        return r;
    }
    Snapshot owner_route_snapshot(const State& st, const Solution& s, size_t owner) const {  // (:357-393)
        Snapshot snap;
        size_t len = list_len(s, owner);
        bool has_prev = false;
        size_t prev = 0;
        for (size_t pos = 0; pos < len; ++pos) {
            int64_t got = list_get(s, owner, pos);
            if (got == NONE || (size_t)got >= st.node_count) {
                ++snap.invalid;
                has_prev = false;
                continue;
            }
            size_t node = (size_t)got;
            snap.elements.push_back(node);
            if (expected_owner) {
                int64_t ex = expected_owner(s, node);
                if (ex != NONE && (size_t)ex != owner) ++snap.violation;
            }
            if (has_prev) snap.edges.push_back({prev, node});
            prev = node;
            has_prev = true;
        }
        return snap;
    }
    State build_state(const Solution& s) const {  // (:62-90)
        size_t n = node_count(s), owners = owner_count(s);
        std::vector<int64_t> dur(n);
        for (size_t i = 0; i < n; ++i) dur[i] = node_duration(s, i);
        State st(n, owners, std::move(dur));
        std::vector<size_t> succ;
        for (size_t node = 0; node < n; ++node) {
            succ.clear();
            fixed_successors(s, node, succ);
            for (size_t t : succ) {
                if (t < n)
                    st.add_edge({node, t});
                else
                    ++st.invalid_fixed_edges;
            }
        }
        for (size_t o = 0; o < owners; ++o) st.replace_owner_route(o, owner_route_snapshot(st, s, o));
        st.rebuild_graph_summary();
        st.refresh_penalty();
        return st;
    }
    Score evaluate(const Solution& s) const override { return score_of(build_state(s)); }
    size_t match_count(const Solution& s) const override {  // (:96-99)
        State st = build_state(s);
        return st.hard_penalty + (st.makespan > 0 ? 1 : 0);
    }
    Score initialize(const Solution& s) override {
        state = build_state(s);
        has_state = true;
        return score_of(state);
    }
    Score on_insert(const Solution& s, size_t e, size_t d) override {  // (:129-152)
        if (d != list_descriptor || !has_state || e >= state.owner_edges.size()) return Score::zero();
        Score before = score_of(state);
        RouteChange ch = state.replace_owner_route(e, owner_route_snapshot(state, s, e));
        state.refresh_graph_after_route_change(ch);
        state.refresh_penalty();
        return score_of(state) - before;
    }
    Score on_retract(const Solution&, size_t e, size_t d) override {  // (:154-174)
        if (d != list_descriptor || !has_state || e >= state.owner_edges.size()) return Score::zero();
        Score before = score_of(state);
        RouteChange ch = state.replace_owner_route(e, Snapshot());
        state.refresh_graph_after_route_change(ch);
        state.refresh_penalty();
        return score_of(state) - before;
    }
    void reset() override {
        has_state = false;
        state = State();
    }
};

struct ConstraintSet {
    std::vector<std::unique_ptr<Constraint>> members;
    Score evaluate_all(const Solution& s) const {
        Score t;
        for (auto& c : members) t = t + c->evaluate(s);
        return t;
    }
    Score initialize_all(const Solution& s) {
        Score t;
        for (auto& c : members) t = t + c->initialize(s);
        return t;
    }
    Score on_insert_all(const Solution& s, size_t e, size_t d) {
        Score t;
        for (auto& c : members) t = t + c->on_insert(s, e, d);
        return t;
    }
    Score on_retract_all(const Solution& s, size_t e, size_t d) {
        Score t;
        for (auto& c : members) t = t + c->on_retract(s, e, d);
        return t;
    }
    void reset_all() {
        for (auto& c : members) c->reset();
    }
};

// ---- ScoreDirector (director/score_director/incremental.rs:64-394) -------
struct DirectorScoreState {
    bool solution_has_score;
    Score solution_score;
    bool initialized;
    Score committed_score;
};

struct ScoreDirector {
    Solution working;
    ConstraintSet constraints;
    Score cached;
    bool initialized = false;
    int levels = 2, hard_levels = 1;
    // stats mirrored from crates/solverforge-solver/src/stats/solver.rs
    uint64_t retract_calls = 0, insert_calls = 0;

    Score calculate_score() {  // incremental.rs:141-149
        if (!initialized) {
            cached = constraints.initialize_all(working);
            initialized = true;
        }
        working.has_score = true;
        working.score = cached;
        return cached;
    }
    Score fresh_score() const {  // incremental.rs:151-155 (clone + evaluate_all)
        Solution clone = working;
        return constraints.evaluate_all(clone);
    }
    void before_variable_changed(size_t d, size_t e) {  // incremental.rs:157-169
        if (!initialized) return;
        ++retract_calls;
        cached = cached + constraints.on_retract_all(working, e, d);
    }
    void after_variable_changed(size_t d, size_t e) {  // incremental.rs:171-185
        if (!initialized) return;
        ++insert_calls;
        cached = cached + constraints.on_insert_all(working, e, d);
    }
    DirectorScoreState snapshot_score_state() const {  // incremental.rs:193-201
        return {working.has_score, working.score, initialized, cached};
    }
    void restore_score_state(const DirectorScoreState& st) {  // incremental.rs:203-218
        working.has_score = st.solution_has_score;
        working.score = st.solution_score;
        if (st.initialized) {
            cached = st.committed_score;
            initialized = true;
        } else {
            constraints.reset_all();
            cached = Score::zero();
            initialized = false;
        }
    }
    void reset() {
        constraints.reset_all();
        initialized = false;
        cached = Score::zero();
    }
    size_t entity_count(size_t d) const { return d < working.classes.size() ? working.classes[d].n : 0; }
};

}  // namespace sfo
