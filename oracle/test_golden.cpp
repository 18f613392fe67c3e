// ORACLE — TEST INFRASTRUCTURE ONLY.
// Pins the oracle's constraint nodes against the reference's own known-answer tests.
// Every CASE cites the reference test it re-derives (values copied by hand from the
// assertions, never source text).  Output: one "ok <name>" / "FAIL <name>" line per case;
// exit status = number of failures.  Driven by tests/test_oracle_golden.py.
#include <cstdio>
#include <string>

#include <array>

#include "sfo_clarke_wright.hpp"
#include "sfo_models.hpp"

using namespace sfo;

static int failures = 0;
#define CHECK(name, ...)                           \
    do {                                           \
        if ((__VA_ARGS__))                         \
            std::printf("ok %s\n", name);          \
        else {                                     \
            std::printf("FAIL %s (line %d)\n", name, __LINE__); \
            ++failures;                            \
        }                                          \
    } while (0)

static Score soft(int64_t v) { return Score::level(0, v); }  // SoftScore::of(v): single level

// Toy solution: class 0 rows with two int columns (vars[0], vars[1]).
static Solution two_col(std::vector<int64_t> c0, std::vector<int64_t> c1) {
    Solution s;
    s.classes.resize(1);
    s.classes[0].n = c0.size();
    s.classes[0].vars = {c0, c1};
    return s;
}

// Row-conflict self-join of constraint/tests/bi_incr.rs: key = row, filter a.col < b.col.
static SelfJoinBiConstraint row_conflict(Impact impact, Weight2 w, Filter2 f = nullptr) {
    SelfJoinBiConstraint c;
    c.name = "Row conflict";
    c.impact = impact;
    c.source = ChangeSource::descriptor(0);
    c.count = [](const Solution& s) { return s.classes[0].n; };
    c.key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    c.filter = f ? f : [](const Solution& s, size_t a, size_t b) {
        return s.classes[0].vars[1][a] < s.classes[0].vars[1][b];
    };
    c.weight = w;
    return c;
}
static Weight2 const_w(int64_t v) {
    return [v](const Solution&, size_t, size_t) { return soft(v); };
}

static void bi_incr_cases() {
    {  // bi_incr.rs:21-48 test_evaluate_no_conflicts
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 1, 2}, {0, 1, 2});
        CHECK("bi_incr.evaluate_no_conflicts", c.evaluate(s) == soft(0) && c.match_count(s) == 0);
    }
    {  // bi_incr.rs:50-76 test_evaluate_with_conflicts
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 0, 2}, {0, 1, 2});
        CHECK("bi_incr.evaluate_with_conflicts", c.evaluate(s) == soft(-1) && c.match_count(s) == 1);
    }
    {  // bi_incr.rs:78-117 test_incremental_insert: deltas 0, -1, 0
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 0, 2}, {0, 1, 2});
        c.initialize(s);
        c.reset();
        bool ok = c.on_insert(s, 0, 0) == soft(0);
        ok = ok && c.on_insert(s, 1, 0) == soft(-1);
        ok = ok && c.on_insert(s, 2, 0) == soft(0);
        CHECK("bi_incr.incremental_insert", ok);
    }
    {  // bi_incr.rs:119-143 test_incremental_retract: +1
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 0}, {0, 1});
        c.initialize(s);
        c.reset();
        c.on_insert(s, 0, 0);
        c.on_insert(s, 1, 0);
        CHECK("bi_incr.incremental_retract", c.on_retract(s, 0, 0) == soft(1));
    }
    {  // bi_incr.rs:145-168 test_reward_type: +2
        auto c = row_conflict(Impact::Reward, const_w(2), [](const Solution& s, size_t a, size_t b) {
            int64_t ca = s.classes[0].vars[1][a], cb = s.classes[0].vars[1][b];
            return ca < cb && (ca - cb == 1 || cb - ca == 1);
        });
        Solution s = two_col({0, 0}, {0, 1});
        CHECK("bi_incr.reward_type", c.evaluate(s) == soft(2));
    }
    {  // bi_incr.rs:170-194 test_dynamic_weight: -3
        auto c = row_conflict(Impact::Penalty, [](const Solution& s, size_t a, size_t b) {
            int64_t d = s.classes[0].vars[1][b] - s.classes[0].vars[1][a];
            return soft(d < 0 ? -d : d);
        });
        Solution s = two_col({0, 0}, {0, 3});
        CHECK("bi_incr.dynamic_weight", c.evaluate(s) == soft(-3));
    }
    {  // bi_incr.rs:196-219 test_multiple_conflicts: -3, 3 matches
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 0, 0}, {0, 1, 2});
        CHECK("bi_incr.multiple_conflicts", c.evaluate(s) == soft(-3) && c.match_count(s) == 3);
    }
    {  // bi_incr.rs:221-246 test_reset: insert after reset -> 0
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0, 0}, {0, 1});
        c.initialize(s);
        c.reset();
        c.on_insert(s, 0, 0);
        c.on_insert(s, 1, 0);
        c.reset();
        CHECK("bi_incr.reset", c.on_insert(s, 0, 0) == soft(0));
    }
    {  // bi_incr.rs:248-273 test_in_constraint_set: evaluate_all == -1
        ConstraintSet set;
        set.members.push_back(std::make_unique<SelfJoinBiConstraint>(row_conflict(Impact::Penalty, const_w(1))));
        Solution s = two_col({0, 0, 2}, {0, 1, 2});
        CHECK("bi_incr.in_constraint_set", set.evaluate_all(s) == soft(-1));
    }
    {  // bi_incr.rs:275-311 test_out_of_bounds: zero deltas
        auto c = row_conflict(Impact::Penalty, const_w(1));
        Solution s = two_col({0}, {0});
        c.initialize(s);
        CHECK("bi_incr.out_of_bounds", c.on_insert(s, 100, 0) == soft(0) && c.on_retract(s, 100, 0) == soft(0));
    }
}

// cross_bi_incr.rs: Schedule{shifts(employee_id, day) = class 0, employees(id, unavailable_day) = class 1}
static Solution schedule(std::vector<int64_t> shift_emp, std::vector<int64_t> shift_day,
                         std::vector<int64_t> emp_id, std::vector<int64_t> emp_unavail) {
    Solution s;
    s.classes.resize(2);
    s.classes[0].n = shift_emp.size();
    s.classes[0].vars = {shift_emp, shift_day};
    s.classes[1].n = emp_id.size();
    s.classes[1].vars = {emp_id, emp_unavail};  // one unavailable day per employee (NONE = none)
    return s;
}
static CrossBiConstraint unavailable_employee() {  // cross_bi_incr.rs:60-83
    CrossBiConstraint c;
    c.name = "Unavailable employee";
    c.impact = Impact::Penalty;
    c.a_source = ChangeSource::descriptor(0);
    c.b_source = ChangeSource::descriptor(1);
    c.a_count = [](const Solution& s) { return s.classes[0].n; };
    c.b_count = [](const Solution& s) { return s.classes[1].n; };
    c.key_a = [](const Solution& s, size_t a) { return s.classes[0].vars[0][a]; };
    c.key_b = [](const Solution& s, size_t b) { return s.classes[1].vars[0][b]; };
    c.filter = [](const Solution& s, size_t a, size_t b) {
        return s.classes[0].vars[0][a] != NONE && s.classes[1].vars[1][b] == s.classes[0].vars[1][a];
    };
    c.weight = [](const Solution&, size_t, size_t) { return soft(1); };
    return c;
}

static void cross_bi_cases() {
    Solution sample = schedule({0, 0}, {5, 6}, {0}, {5});  // sample_schedule() cross_bi_incr.rs:110-128
    {  // cross_bi_incr.rs:205-216 evaluate / match_count without initialize
        auto c = unavailable_employee();
        CHECK("cross_bi.evaluate_without_initialize", c.evaluate(sample) == soft(-1) && c.match_count(sample) == 1);
    }
    {  // cross_bi_incr.rs:247-259 incremental updates: init -1, retract +1, insert -1
        auto c = unavailable_employee();
        bool ok = c.initialize(sample) == soft(-1);
        ok = ok && c.on_retract(sample, 0, 0) == soft(1);
        ok = ok && c.on_insert(sample, 0, 0) == soft(-1);
        CHECK("cross_bi.incremental_updates", ok);
    }
    {  // cross_bi_incr.rs:261-273 b-side retract/insert: unavailable day 5 -> 6 keeps total -1
        auto c = unavailable_employee();
        Solution s = sample;
        Score total = c.initialize(s);
        bool ok = total == soft(-1);
        total = total + c.on_retract(s, 0, 1);
        s.classes[1].vars[1][0] = 6;
        total = total + c.on_insert(s, 0, 1);
        CHECK("cross_bi.b_side_updates", ok && total == soft(-1) && total == c.evaluate(s));
    }
    {  // cross_bi_incr.rs:372-381 unrelated descriptor is a no-op
        auto c = unavailable_employee();
        Score initial = c.initialize(sample);
        CHECK("cross_bi.unrelated_descriptor_noop", initial == soft(-1) && c.on_retract(sample, 0, 2) == soft(0));
    }
    {  // cross_bi_incr.rs:307-335 filter sees source indexes: (shift 1, employee 0), weight = day 6 -> -6
        CrossBiConstraint c = unavailable_employee();
        c.filter = [](const Solution&, size_t a, size_t b) { return a == 1 && b == 0; };
        c.weight = [](const Solution& s, size_t a, size_t) { return soft(s.classes[0].vars[1][a]); };
        Solution two = schedule({0, 0}, {5, 6}, {0, 1}, {NONE, NONE});
        CHECK("cross_bi.filter_source_indexes", c.match_count(two) == 1 && c.evaluate(two) == soft(-6));
    }
    {  // same-descriptor predicate join fires both sides and tests (i,i) (incremental.rs:93-115)
        CrossBiConstraint c;
        c.name = "pair";
        c.impact = Impact::Penalty;
        c.a_source = c.b_source = ChangeSource::descriptor(0);
        c.a_count = c.b_count = [](const Solution& s) { return s.classes[0].n; };
        c.key_a = c.key_b = [](const Solution&, size_t) { return (int64_t)0; };
        c.filter = [](const Solution& s, size_t a, size_t b) {
            return a < b && s.classes[0].vars[0][a] == s.classes[0].vars[0][b];
        };
        c.weight = [](const Solution&, size_t, size_t) { return soft(1); };
        Solution s = two_col({1, 1, 2, 1}, {0, 0, 0, 0});
        Score total = c.initialize(s);
        bool ok = total == soft(-3) && total == c.evaluate(s);
        total = total + c.on_retract(s, 1, 0);
        s.classes[0].vars[0][1] = 2;
        total = total + c.on_insert(s, 1, 0);
        ok = ok && total == soft(-2) && total == c.evaluate(s);
        CHECK("cross_bi.same_descriptor_both_sides", ok);
    }
}

static void exists_cases() {
    {  // exists.rs:92-121 flattened not-exists: init -3, after route=[1,2,3] -> 0
        Solution s;
        s.classes.resize(1);
        s.classes[0].n = 1;
        s.classes[0].lists = {{}};
        std::vector<int64_t> customers = {1, 2, 3};
        ExistsConstraint c;
        c.name = "missing assignment";
        c.impact = Impact::Penalty;
        c.mode = ExistenceMode::NotExists;
        c.a_source = ChangeSource::fixed();
        c.parent_source = ChangeSource::descriptor(0);
        c.a_count = [customers](const Solution&) { return customers.size(); };
        c.parent_count = [](const Solution& s) { return s.classes[0].n; };
        c.filter_a = [](const Solution&, size_t) { return true; };
        c.filter_parent = [](const Solution&, size_t) { return true; };
        c.key_a = [customers](const Solution&, size_t i) { return customers[i]; };
        c.flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) {
            for (uint32_t v : s.classes[0].lists[p]) out.push_back(v);
        };
        c.weight = [](const Solution&, size_t) { return soft(1); };
        Score total = c.initialize(s);
        bool ok = total == soft(-3);
        total = total + c.on_retract(s, 0, 0);
        s.classes[0].lists[0] = {1, 2, 3};
        total = total + c.on_insert(s, 0, 0);
        CHECK("exists.flattened_not_exists_route_change", ok && total == c.evaluate(s) && total == soft(0));
    }
    {  // exists.rs:34-78 exists: tasks(assignee) static, workers(available) descriptor 0; 0 -> -2
        // class 0 = workers {id, available}; tasks held as facts
        Solution s = two_col({0, 1}, {1, 1});
        std::vector<int64_t> assignee = {0, 0, 1};
        ExistsConstraint c;
        c.name = "unavailable worker";
        c.impact = Impact::Penalty;
        c.mode = ExistenceMode::Exists;
        c.a_source = ChangeSource::fixed();
        c.parent_source = ChangeSource::descriptor(0);
        c.a_count = [assignee](const Solution&) { return assignee.size(); };
        c.parent_count = [](const Solution& s) { return s.classes[0].n; };
        c.filter_a = [assignee](const Solution&, size_t i) { return assignee[i] != NONE; };
        c.filter_parent = [](const Solution& s, size_t p) { return s.classes[0].vars[1][p] == 0; };  // !available
        c.key_a = [assignee](const Solution&, size_t i) { return assignee[i]; };
        c.flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) {
            out.push_back(s.classes[0].vars[0][p]);  // SelfFlatten: one B row = one key (worker.id)
        };
        c.weight = [](const Solution&, size_t) { return soft(1); };
        Score total = c.initialize(s);
        bool ok = total == soft(0);
        total = total + c.on_retract(s, 0, 0);
        s.classes[0].vars[1][0] = 0;  // worker 0 becomes unavailable
        total = total + c.on_insert(s, 0, 0);
        CHECK("exists.b_descriptor_change_updates_all_a", ok && total == c.evaluate(s) && total == soft(-2));
    }
    {  // exists.rs:139-190 same source on both sides: -2 -> 0 when the enabled peer is disabled
        Solution s = two_col({1, 1}, {0, 1});  // {key, enabled}
        ExistsConstraint c;
        c.name = "key has enabled peer";
        c.impact = Impact::Penalty;
        c.mode = ExistenceMode::Exists;
        c.a_source = c.parent_source = ChangeSource::descriptor(0);
        c.a_count = c.parent_count = [](const Solution& s) { return s.classes[0].n; };
        c.filter_a = [](const Solution&, size_t) { return true; };
        c.filter_parent = [](const Solution& s, size_t p) { return s.classes[0].vars[1][p] != 0; };
        c.key_a = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        c.flatten = [](const Solution& s, size_t p, std::vector<int64_t>& out) {
            out.push_back(s.classes[0].vars[0][p]);
        };
        c.weight = [](const Solution&, size_t) { return soft(1); };
        Score total = c.initialize(s);
        bool ok = total == soft(-2);
        total = total + c.on_retract(s, 1, 0);
        s.classes[0].vars[1][1] = 0;
        total = total + c.on_insert(s, 1, 0);
        CHECK("exists.same_source_consistent", ok && total == c.evaluate(s) && total == soft(0));
    }
}

static GroupedConstraint workload(Impact impact, GroupWeight w) {  // grouped.rs helper shape
    GroupedConstraint c;
    c.name = "Workload";
    c.impact = impact;
    c.source = ChangeSource::descriptor(0);
    c.count = [](const Solution& s) { return s.classes[0].n; };
    c.filter = [](const Solution&, size_t) { return true; };
    c.key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    c.value = [](const Solution&, size_t) { return (int64_t)1; };  // count()
    c.weight = w;
    return c;
}

// constraint/tests/{tri,quad,penta}_incr.rs: key = team, tuples of `arity` tasks sharing a team
static SelfJoinNaryConstraint cluster(size_t arity, Impact impact, int64_t w, bool only_indexes_from_one = false) {
    SelfJoinNaryConstraint c;
    c.name = "Cluster";
    c.arity = arity;
    c.impact = impact;
    c.source = ChangeSource::descriptor(0);
    c.count = [](const Solution& s) { return s.classes[0].n; };
    c.key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    c.filter = [arity, only_indexes_from_one](const Solution&, const size_t* idx) {
        if (!only_indexes_from_one) return true;
        for (size_t i = 0; i < arity; ++i)
            if (idx[i] != i + 1) return false;  // (1, 2, 3[, 4[, 5]]): the filter sees SOURCE indexes
        return true;
    };
    c.weight = [w](const Solution&, const size_t*) { return soft(w); };
    return c;
}
static void nary_cases() {
    const char* names[6] = {"", "", "", "tri", "quad", "penta"};
    for (size_t arity = 3; arity <= 5; ++arity) {
        std::string pre = std::string(names[arity]) + "_incr.";
        std::vector<int64_t> same(arity, 1), zeros;
        {  // evaluate: `arity` tasks on team 1 + one on team 2 -> one tuple, -1
            std::vector<int64_t> t = same;
            t.push_back(2);
            Solution s = two_col(t, std::vector<int64_t>(t.size(), 0));
            auto c = cluster(arity, Impact::Penalty, 1);
            CHECK((pre + "evaluate").c_str(), c.evaluate(s) == soft(-1) && c.match_count(s) == 1);
        }
        {  // multiple: arity + 1 tasks on one team -> C(arity + 1, arity) = arity + 1 tuples (-4 / -5 / -6)
            std::vector<int64_t> t(arity + 1, 1);
            Solution s = two_col(t, std::vector<int64_t>(t.size(), 0));
            auto c = cluster(arity, Impact::Penalty, 1);
            CHECK((pre + "multiple").c_str(), c.evaluate(s) == soft(-(int64_t)(arity + 1)));
        }
        {  // incremental: initialize -1, retract task 0 -> +1, re-insert -> -1
            Solution s = two_col(same, std::vector<int64_t>(arity, 0));
            auto c = cluster(arity, Impact::Penalty, 1);
            bool ok = c.initialize(s) == soft(-1);
            ok = ok && c.on_retract(s, 0, 0) == soft(1);
            ok = ok && c.on_insert(s, 0, 0) == soft(-1);
            CHECK((pre + "incremental").c_str(), ok);
        }
        {  // filter_receives_source_indexes: arity + 1 tasks, only (1, 2, ..) passes -> one match
            std::vector<int64_t> t(arity + 1, 1);
            Solution s = two_col(t, std::vector<int64_t>(t.size(), 0));
            auto c = cluster(arity, Impact::Penalty, 1, true);
            CHECK((pre + "filter_source_indexes").c_str(),
                  c.match_count(s) == 1 && c.evaluate(s) == soft(-1) && c.initialize(s) == soft(-1));
        }
        {  // reward: +5
            Solution s = two_col(same, std::vector<int64_t>(arity, 0));
            auto c = cluster(arity, Impact::Reward, 5);
            CHECK((pre + "reward").c_str(), c.evaluate(s) == soft(5));
        }
    }
}

// stream/collector/tests/collector.rs:142-260 (the six load_balance tests)
static void load_balance_cases() {
    {  // test_perfectly_balanced
        LoadBalanceAccumulator a;
        a.accumulate(0, 1), a.accumulate(1, 1);
        CHECK("load_balance.perfectly_balanced", a.unfairness() == 0);
    }
    {  // test_unbalanced: loads [2, 1] -> sqrt(0.5) rounds to 1; test_retract: back to balanced
        LoadBalanceAccumulator a;
        a.accumulate(0, 1), a.accumulate(0, 1), a.accumulate(1, 1);
        CHECK("load_balance.unbalanced", a.unfairness() == 1);
        a.retract(0, 1);
        CHECK("load_balance.retract", a.unfairness() == 0);
    }
    {  // test_empty, test_single_item
        LoadBalanceAccumulator a;
        CHECK("load_balance.empty", a.unfairness() == 0);
        a.accumulate(0, 1), a.accumulate(0, 1), a.accumulate(0, 1);
        CHECK("load_balance.single_item", a.loads[0] == 3 && a.unfairness() == 0);
    }
    {  // test_load_balance_standard_deviation: A=2; +B=1 -> 1; +B -> 0; -B -> 1; -B -> 0; -A -> 0
        LoadBalanceAccumulator a;
        bool ok = a.unfairness() == 0;
        a.accumulate(100, 2);
        ok = ok && a.unfairness() == 0;
        a.accumulate(200, 1);
        ok = ok && a.unfairness() == 1;
        a.accumulate(200, 1);
        ok = ok && a.unfairness() == 0;
        a.retract(200, 1);
        ok = ok && a.unfairness() == 1;
        a.retract(200, 1);
        ok = ok && a.unfairness() == 0 && a.item_counts.size() == 1;
        a.retract(100, 2);
        ok = ok && a.unfairness() == 0 && a.item_counts.empty() && a.sum == 0 && a.squared_deviation_integral == 0 &&
             a.squared_deviation_fraction_numerator == 0;
        CHECK("load_balance.standard_deviation", ok);
    }
    {  // zero metrics are skipped (load_balance.rs:194-196); the closed form integral = sum x^2, fraction = -(sum x)^2
        LoadBalanceAccumulator a;
        a.accumulate(7, 0);
        bool ok = a.item_counts.empty();
        a.accumulate(1, 5), a.accumulate(2, 3), a.accumulate(3, 9), a.accumulate(2, 4);
        ok = ok && a.squared_deviation_integral == 25 + 49 + 81 && a.squared_deviation_fraction_numerator == -(21 * 21) && a.sum == 21;
        CHECK("load_balance.closed_form", ok);
    }
}

static void grouped_cases() {
    {  // grouped.rs:26-58 evaluate: counts 3,1 with weight count^2 -> -10
        auto c = workload(Impact::Penalty, [](int64_t, int64_t n) { return soft(n * n); });
        Solution s = two_col({1, 1, 1, 2}, {0, 0, 0, 0});
        CHECK("grouped.evaluate", c.evaluate(s) == soft(-10));
    }
    {  // grouped.rs:60-103 incremental: init -3, retract +1, insert -1
        auto c = workload(Impact::Penalty, [](int64_t, int64_t n) { return soft(n); });
        Solution s = two_col({1, 1, 2}, {0, 0, 0});
        bool ok = c.initialize(s) == soft(-3);
        ok = ok && c.on_retract(s, 0, 0) == soft(1);
        ok = ok && c.on_insert(s, 0, 0) == soft(-1);
        CHECK("grouped.incremental", ok);
    }
    {  // grouped.rs:105-125 reward: +2
        auto c = workload(Impact::Reward, [](int64_t, int64_t n) { return soft(n); });
        Solution s = two_col({1, 1}, {0, 0});
        CHECK("grouped.reward", c.evaluate(s) == soft(2));
    }
    {  // grouped.rs:127-147 weight can use key: 1*1 + 2*2 -> -5
        auto c = workload(Impact::Penalty, [](int64_t k, int64_t n) { return soft(k * n); });
        Solution s = two_col({1, 2, 2}, {0, 0, 0});
        CHECK("grouped.weight_uses_key", c.evaluate(s) == soft(-5));
    }
}

// director/tests/benchmarks.rs:116-182: incremental == calculate_full after 1000 do/undo moves.
static void director_case() {
    const size_t n = 100;
    ScoreDirector d;
    Solution& s = d.working;
    s.classes.resize(1);
    s.classes[0].n = n;
    s.classes[0].vars.assign(3, std::vector<int64_t>(n));
    for (size_t i = 0; i < n; ++i) {
        s.classes[0].vars[0][i] = 0;                      // employee_id = Some(0)
        s.classes[0].vars[1][i] = (int64_t)(i % 24);      // start_hour
        s.classes[0].vars[2][i] = (int64_t)(i % 24) + 1;  // end_hour
    }
    auto full = [](const Solution& s) {  // benchmarks.rs:45-70 calculate_full
        int64_t penalty = 0;
        size_t n = s.classes[0].n;
        auto& e = s.classes[0].vars[0];
        auto& st = s.classes[0].vars[1];
        auto& en = s.classes[0].vars[2];
        for (size_t i = 0; i < n; ++i)
            if (e[i] == NONE) ++penalty;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = i + 1; j < n; ++j)
                if (e[i] != NONE && e[i] == e[j] && st[i] < en[j] && st[j] < en[i]) penalty += 10;
        return soft(-penalty);
    };
    d.constraints.members.push_back(make_unassigned(0, 0, soft(1), "Unassigned"));
    auto ov = std::make_unique<SelfJoinBiConstraint>();
    ov->name = "Overlapping";
    ov->impact = Impact::Penalty;
    ov->source = ChangeSource::descriptor(0);
    ov->count = [](const Solution& s) { return s.classes[0].n; };
    ov->key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    ov->filter = [](const Solution& s, size_t a, size_t b) {
        return a < b && s.classes[0].vars[1][a] < s.classes[0].vars[2][b] &&
               s.classes[0].vars[1][b] < s.classes[0].vars[2][a];
    };
    ov->weight = [](const Solution&, size_t, size_t) { return soft(10); };
    d.constraints.members.push_back(std::move(ov));
    bool ok = d.calculate_score() == full(d.working);
    for (size_t i = 0; i < 1000; ++i) {
        size_t idx = i % n;
        int64_t old = d.working.classes[0].vars[0][idx];
        d.before_variable_changed(0, idx);
        d.working.classes[0].vars[0][idx] = (int64_t)(i % 5) + 1;
        d.after_variable_changed(0, idx);
        ok = ok && d.cached == full(d.working);
        d.before_variable_changed(0, idx);
        d.working.classes[0].vars[0][idx] = old;
        d.after_variable_changed(0, idx);
    }
    CHECK("director.incremental_equals_full", ok && d.cached == full(d.working));
}

// heuristic/move/tests/list_reverse.rs:63-171: do / undo of a reversal, doability bounds
static void list_reverse_cases() {
    auto mk = [](std::vector<uint32_t> cities) {
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = 1;
        d.working.classes[0].lists = {cities};
        return d;
    };
    auto rev = [](size_t start, size_t end) {
        Move m;
        m.kind = Move::ListReverse;
        m.a = m.b = 0;
        m.a_pos = start;
        m.b_pos = end;
        return m;
    };
    {
        ScoreDirector d = mk({1, 2, 3, 4, 5});
        Move m = rev(1, 4);
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 4, 3, 2, 5});
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4, 5});
        CHECK("list_reverse.reverse_segment", ok);
    }
    {
        ScoreDirector d = mk({1, 2, 3, 4});
        Move m = rev(0, 4);
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({4, 3, 2, 1});
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4});
        CHECK("list_reverse.reverse_entire_list", ok);
    }
    {
        ScoreDirector d = mk({1, 2, 3});
        CHECK("list_reverse.single_element_not_doable", !move_is_doable(d, rev(1, 2)));
        CHECK("list_reverse.out_of_bounds_not_doable", !move_is_doable(d, rev(1, 10)));
    }
}

// heuristic/move/tests/list_ruin.rs:191-470 (ruin + greedy recreate: do / undo, first-position tie, recreate order by
// score, doability, final-position bookkeeping) and the SipHash / scoped_seed plumbing of the ruin seed
static void list_ruin_cases() {
    auto mk = [](std::vector<std::vector<uint32_t>> routes, bool prefer_four_before_two) {
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = routes.size();
        d.working.classes[0].lists = routes;
        if (prefer_four_before_two) {  // RouteScoreConstraint(prefer_four_before_two), list_ruin.rs:158-174
            auto c = std::make_unique<UniConstraint>();
            c->name = "routeScore";
            c->impact = Impact::Reward;
            c->source = ChangeSource::descriptor(0);
            c->count = [](const Solution&) { return (size_t)1; };  // scores routes[0] only
            c->filter = [](const Solution&, size_t) { return true; };
            c->weight = [](const Solution& s, size_t) {
                const auto& r = s.classes[0].lists[0];
                size_t p4 = SIZE_MAX, p2 = SIZE_MAX;
                for (size_t i = 0; i < r.size(); ++i) {
                    if (r[i] == 4 && p4 == SIZE_MAX) p4 = i;
                    if (r[i] == 2 && p2 == SIZE_MAX) p2 = i;
                }
                return soft(p4 < p2 ? 100 : 0);
            };
            d.constraints.members.push_back(std::move(c));
        }
        return d;
    };
    auto ruin = [](size_t entity, std::vector<uint16_t> idx) {
        Move m;
        m.kind = Move::Ruin;
        m.a = m.b = entity;
        std::sort(idx.begin(), idx.end());  // ListRuinMove::new sorts (single_ruin_source)
        m.a_pos = idx.size();
        for (size_t i = 0; i < idx.size() && i < 8; ++i) m.ruin_idx[i] = idx[i];
        return m;
    };
    auto sorted = [](std::vector<uint32_t> v) {
        std::sort(v.begin(), v.end());
        return v;
    };
    {  // ruin_single_element: constant score -> the first tried position (entity 0, position 0) wins
        ScoreDirector d = mk({{1, 2, 3, 4, 5}}, false);
        d.calculate_score();
        Move m = ruin(0, {2});
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({3, 1, 2, 4, 5});
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4, 5});
        CHECK("list_ruin.single_element_first_position_wins", ok);
    }
    {  // ruin_multiple_elements + ruin_unordered_indices
        ScoreDirector d = mk({{1, 2, 3, 4, 5}}, false);
        d.calculate_score();
        Move m = ruin(0, {3, 1});
        bool ok = move_is_doable(d, m) && m.ruin_idx[0] == 1 && m.ruin_idx[1] == 3;
        MoveUndo u = move_do(d, m);
        ok = ok && sorted(d.working.classes[0].lists[0]) == std::vector<uint32_t>({1, 2, 3, 4, 5}) && u.placements.size() == 2;
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4, 5});
        CHECK("list_ruin.multiple_elements_do_undo", ok);
    }
    {  // ruin_recreate_can_choose_removed_elements_out_of_removal_order
        ScoreDirector d = mk({{1, 2, 3, 4}}, true);
        d.calculate_score();
        Move m = ruin(0, {1, 3});
        MoveUndo u = move_do(d, m);
        const auto& r = d.working.classes[0].lists[0];
        size_t p4 = std::find(r.begin(), r.end(), 4u) - r.begin(), p2 = std::find(r.begin(), r.end(), 2u) - r.begin();
        bool ok = p4 < p2 && d.calculate_score() == soft(100) && d.calculate_score() == d.fresh_score();
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4}) && d.cached == d.fresh_score();
        CHECK("list_ruin.recreate_picks_by_score_not_removal_order", ok);
    }
    {  // two routes: elements may move across lists, undo restores both exactly
        ScoreDirector d = mk({{1, 2, 3, 4}, {5, 6, 7}}, true);
        d.calculate_score();
        Move m = ruin(1, {0, 2});
        MoveUndo u = move_do(d, m);
        bool ok = d.working.classes[0].lists[0].size() + d.working.classes[0].lists[1].size() == 7 && d.cached == d.fresh_score();
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4}) &&
             d.working.classes[0].lists[1] == std::vector<uint32_t>({5, 6, 7}) && d.cached == d.fresh_score();
        CHECK("list_ruin.undo_restores_every_list", ok);
    }
    {  // ruin_recreate_restores_multiple_source_entities (:307-338): new_multi_source over two lists, four elements, do + undo
        ScoreDirector d = mk({{1, 2, 3, 4}, {5, 6, 7}}, true);
        d.calculate_score();
        Move m;
        m.kind = Move::Ruin;
        m.a = m.b = 0;
        m.a_pos = 4;
        m.ruin_multi = true;
        const uint16_t idx[4] = {1, 3, 0, 2}, src[4] = {0, 0, 1, 1};
        for (int i = 0; i < 4; ++i) m.ruin_idx[i] = idx[i], m.ruin_src[i] = src[i];
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0].size() + d.working.classes[0].lists[1].size() == 7 && d.cached == d.fresh_score();
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4}) && d.working.classes[0].lists[1] == std::vector<uint32_t>({5, 6, 7}) &&
             d.cached == d.fresh_score();
        CHECK("list_ruin.recreate_restores_multiple_source_entities", ok);
    }
    {  // precedence_ruin_restores_original_when_recreate_has_no_safe_position (:443-468): two nodes that precede each other (elements 0, 1
       // for the reference's 1, 2): wherever the ruined element goes, a cycle closes -> nothing is placed, the list comes back
        ScoreDirector d = mk({{0, 1}}, false);
        d.calculate_score();
        PrecedenceHooks h;
        h.node_count = 2;
        h.durations = {1, 1};
        h.successors = {{1}, {0}};
        Move m = ruin(0, {0});
        m.prec = &h;
        MoveUndo u = move_do(d, m);
        CHECK("list_ruin.precedence_ruin_restores_original_when_recreate_has_no_safe_position",
              u.placements.empty() && d.working.classes[0].lists[0] == std::vector<uint32_t>({0, 1}));
    }
    {  // heuristic/move/tests/list_multi_swap.rs:72-134: three independent intra-list swaps as one move, do + undo; two swaps in one list
       // are not doable
        ScoreDirector d = mk({{1, 2, 3}, {10, 20, 30}, {100, 200, 300}}, false);
        d.calculate_score();
        Move m;
        m.kind = Move::MultiSwap;
        m.a_pos = 3;
        const uint16_t e[3] = {0, 1, 2}, f[3] = {0, 0, 1}, g[3] = {2, 1, 2};
        for (int i = 0; i < 3; ++i) m.ms_entity[i] = e[i], m.ms_first[i] = f[i], m.ms_second[i] = g[i];
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        const auto& L = d.working.classes[0].lists;
        ok = ok && L[0] == std::vector<uint32_t>({3, 2, 1}) && L[1] == std::vector<uint32_t>({20, 10, 30}) && L[2] == std::vector<uint32_t>({100, 300, 200});
        move_undo(d, m, u);
        ok = ok && L[0] == std::vector<uint32_t>({1, 2, 3}) && L[1] == std::vector<uint32_t>({10, 20, 30}) && L[2] == std::vector<uint32_t>({100, 200, 300});
        CHECK("list_multi_swap.applies_independent_intra_list_swaps_and_undoes", ok);
        Move bad;
        bad.kind = Move::MultiSwap;
        bad.a_pos = 2;
        bad.ms_entity[0] = bad.ms_entity[1] = 0;
        bad.ms_first[0] = 0, bad.ms_second[0] = 1, bad.ms_first[1] = 1, bad.ms_second[1] = 2;
        CHECK("list_multi_swap.rejects_overlapping_entities", !move_is_doable(d, bad));
    }
    {
        ScoreDirector d = mk({{1, 2, 3}}, false);
        CHECK("list_ruin.empty_indices_not_doable", !move_is_doable(d, ruin(0, {})));
        CHECK("list_ruin.out_of_bounds_not_doable", !move_is_doable(d, ruin(0, {0, 10})));
    }
    {  // computes_exact_final_positions_for_same_entity_reinsertion + undo_positions_do_not_underflow...
        std::vector<RuinPlacement> pl = {{0, 0, 0}, {0, 0, 1}, {0, 0, 2}, {0, 1, 3}};
        std::vector<size_t> cur = ruin_final_positions(pl);
        bool ok = cur == std::vector<size_t>({3, 2, 0, 1});
        std::vector<size_t> order;
        for (size_t i = pl.size(); i-- > 0;) {
            size_t at = cur[i];
            order.push_back(at);
            for (size_t j = 0; j < i; ++j)
                if (cur[j] > at) cur[j] -= 1;
        }
        CHECK("list_ruin.final_positions_and_removal_order", ok && order == std::vector<size_t>({1, 0, 0, 0}));
    }
    {  // SipHash-2-4 reference vector of the SipHash paper (key 00..0f, message 00..0e): the round function and the
       // finalisation that DefaultHasher (SipHash-1-3) shares
        uint8_t msg[15];
        for (int i = 0; i < 15; ++i) msg[i] = (uint8_t)i;
        CHECK("siphash24.paper_vector", siphash(2, 4, 0x0706050403020100ULL, 0x0f0e0d0c0b0a0908ULL, msg, 15) == 0xa129ca6149be45e5ULL);
        // seeded ruin streams are reproducible (list_leaf/tests/parity.rs:252-266) and scoped by selector kind
        CHECK("scoped_seed.deterministic_and_scoped",
              scoped_seed(1337, 0, "visits", "list_ruin_move_selector") == scoped_seed(1337, 0, "visits", "list_ruin_move_selector") &&
                  scoped_seed(1337, 0, "visits", "list_ruin_move_selector") != scoped_seed(1337, 0, "visits", "other") &&
                  scoped_seed(1337, 0, "visits", "list_ruin_move_selector") != scoped_seed(1337, 1, "visits", "list_ruin_move_selector"));
    }
    {  // random_range stays inside its bounds and hits both ends (Canon's method, 32-bit path)
        SmallRng r = SmallRng::seed_from_u64(7);
        bool ok = true, lo = false, hi = false;
        for (int i = 0; i < 2000; ++i) {
            uint64_t v = r.random_range_inclusive(2, 5);
            ok = ok && v >= 2 && v <= 5;
            lo = lo || v == 2;
            hi = hi || v == 5;
        }
        CHECK("small_rng.random_range_bounds", ok && lo && hi);
    }
}

// constraint/tests/balance.rs:21-304: equal / unequal distributions, unassigned filtered, incremental retract / insert, empty,
// single key, reward
static void balance_cases() {
    auto mk = [](Impact impact) {
        BalanceConstraint c;
        c.name = "Balance";
        c.impact = impact;
        c.source = ChangeSource::descriptor(0);
        c.count = [](const Solution& s) { return s.classes[0].n; };
        c.filter = [](const Solution&, size_t) { return true; };
        c.key = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
        c.base_score = soft(1000);
        return c;
    };
    auto sol = [](std::vector<int64_t> emp) {
        Solution s;
        s.classes.resize(1);
        s.classes[0].n = emp.size();
        s.classes[0].vars = {emp};
        return s;
    };
    CHECK("balance.equal_distribution", mk(Impact::Penalty).evaluate(sol({0, 0, 1, 1})) == soft(0));
    CHECK("balance.unequal_distribution", mk(Impact::Penalty).evaluate(sol({0, 0, 0, 1})) == soft(-1000));
    CHECK("balance.filters_unassigned", mk(Impact::Penalty).evaluate(sol({0, 1, NONE, NONE})) == soft(0));
    {
        BalanceConstraint c = mk(Impact::Penalty);
        Solution s = sol({0, 0, 1, 1});
        bool ok = c.initialize(s) == soft(0);
        ok = ok && c.on_retract(s, 0, 0) == soft(-500);  // counts 1 and 2: std dev 0.5
        ok = ok && c.on_insert(s, 0, 0) == soft(500);
        CHECK("balance.incremental", ok);
        CHECK("balance.unrelated_descriptor_is_noop", c.on_insert(s, 0, 1) == Score::zero());
    }
    CHECK("balance.empty_solution", mk(Impact::Penalty).evaluate(sol({})) == soft(0));
    CHECK("balance.single_employee", mk(Impact::Penalty).evaluate(sol({0, 0, 0})) == soft(0));
    CHECK("balance.reward", mk(Impact::Reward).evaluate(sol({0, 0, 0, 1})) == soft(1000));
}

// manager/phase_factory/list_construction/cheapest/kernel.rs:57-150 with a constant score (cheapest/tests.rs zero_score): the
// first tried (list, position) wins every time, so the elements pile up in front of list 0 in reverse source order
static void cheapest_insertion_cases() {
    ScoreDirector d;
    d.working.classes.resize(1);
    d.working.classes[0].n = 2;
    d.working.classes[0].lists = {{}, {}};
    SolverStats st;
    construct_list_cheapest(d, 0, {0, 1, 2}, &st);
    CHECK("cheapest_insertion.first_position_wins_ties",
          d.working.classes[0].lists[0] == std::vector<uint32_t>({2, 1, 0}) && d.working.classes[0].lists[1].empty());
    // trials: element k sees (k + 1) + 1 slots -> 2 + 3 + 4
    CHECK("cheapest_insertion.one_score_calculation_per_trial", st.score_calculations == 9 && st.moves_generated == 9 && st.moves_evaluated == 9 && st.moves_applied == 3 && st.step_count == 3);
}

// heuristic/move/tests/compound_scalar.rs:146-262: several edits applied and undone atomically, every edit applied before the
// first after-notification, no-op and illegal candidates rejected
// manager/phase_factory/list_construction/cheapest/tests.rs:244-258 (precedence_downstream_breaks_cheapest_ties_...): elements [1, 0], one
// empty route, every insertion scores the same, hooks = unit durations + "0 precedes 1": element 0 has the longer downstream chain, goes
// first, and element 1 then takes the first of two equal slots -> [1, 0]
// phase/localsearch/phase/tests/foraging.rs:113-134 (score_improvement_required_move_rejects_worse_before_acceptor): two candidates that
// require a score improvement, trial scores -5 and 3 from 0, an acceptor that accepts everything (LateAcceptance whose history is far
// below), AcceptedCount(1): the -5 never reaches the acceptor, so the step takes the 3; both are evaluated.  And phase/hard_delta.rs.
static void gate_cases() {
    Solution s;
    s.classes.resize(1);
    s.classes[0].n = 1;
    s.classes[0].vars.assign(1, std::vector<int64_t>{0});
    ScoreDirector d;
    d.working = s;
    d.levels = 2;
    d.hard_levels = 1;
    auto c = std::make_unique<UniConstraint>();  // soft score = {0, -5, +3}[value]
    c->name = "value cost";
    c->impact = Impact::Penalty;
    c->source = ChangeSource::descriptor(0);
    c->count = [](const Solution& sol) { return sol.classes[0].n; };
    c->filter = [](const Solution&, size_t) { return true; };
    c->weight = [](const Solution& sol, size_t i) {
        static const int64_t cost[3] = {0, 5, -3};
        return Score::of(0, cost[sol.classes[0].vars[0][i]]);
    };
    d.constraints.members.push_back(std::move(c));
    LocalSearch ls;
    ls.director = &d;
    ls.acceptor = std::make_unique<LateAcceptanceAcceptor>(1);  // accepts whatever is >= the late score
    ls.forager.kind = Forager::AcceptedCount;
    ls.forager.accepted_count_limit = 1;
    ls.selection_order = SelectionOrder::Original;
    ls.phase_start();
    static_cast<LateAcceptanceAcceptor*>(ls.acceptor.get())->history.assign(1, Score::of(0, -100));  // "always accept"
    std::vector<std::vector<ScalarEditO>> provided = {{{0, 0, 0, 1, true}}, {{0, 0, 0, 2, true}}};
    GroupedStepTrace t = grouped_scalar_step(ls, provided, 0, 256, {2, 2});
    CHECK("gates.score_improvement_required_move_rejects_worse_before_acceptor",
          ls.stats.moves_evaluated == 2 && ls.stats.moves_applied == 1 && d.calculate_score() == Score::of(0, 3) && t.flags.size() == 2 && t.flags[0] == (1 | 16) &&  // 16: RejectedByScoreImprovement
              (t.flags[1] & 6) == 6);
    bool ok = hard_score_delta(Score::of(-2, 0), Score::of(-1, -50), 1) == 1 && hard_score_delta(Score::of(-1, 0), Score::of(-1, 9), 1) == 0 &&
              hard_score_delta(Score::of(-1, 0), Score::of(-3, 9), 1) == -1 && hard_score_delta(Score::of(0, 0), Score::of(0, 9), 0) == -2;
    CHECK("gates.hard_score_delta", ok);
}

// manager/phase_factory/list_construction/regret/tests.rs:305-326 (regret_requires_only_the_explicit_source_key...): a score that
// no insertion changes, one list, three elements -> every regret ties, the first unassigned element goes to the first slot each
// round: [3, 2, 1] in the reference's payloads = source indices [2, 1, 0].  And a two-list case worked by hand (kernel/mod.rs:58-75).
static void regret_insertion_cases() {
    {
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = 1;
        d.working.classes[0].lists = {{}};
        SolverStats st;
        construct_list_regret(d, 0, {0, 1, 2}, &st);
        CHECK("list_regret.constant_score_piles_up_in_reverse_source_order",
              d.working.classes[0].lists[0] == (std::vector<uint32_t>{2, 1, 0}) && st.step_count == 3 && st.moves_applied == 3 &&
                  st.score_calculations == 3 + 2 * 2 + 3 && st.moves_generated == st.score_calculations);
    }
    {  // construction order keys {2, 0, 1} (execute.rs:81-88): with every regret and score tied the smallest key goes first each round
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = 1;
        d.working.classes[0].lists = {{}};
        construct_list_regret(d, 0, {0, 1, 2}, nullptr, {2, 0, 1});
        CHECK("list_regret.order_keys_rank_the_unassigned_elements", d.working.classes[0].lists[0] == (std::vector<uint32_t>{0, 2, 1}));
    }
    {  // soft = -sum over lists of (position + 1) * weight[element], weights {1, 5, 3}: two empty lists.
        // round 1: every element has two equal slots (regret 0) -> the best score decides: element 0 (score -1) to list 0.
        // round 2: element 1: list 0 front -5 - 2 = -7 (0 shifts), list 0 back -1 - 10 = -11, list 1 -1 - 5 = -6 -> best -6, second -7, regret 1;
        //          element 2: front -3 - 2 = -5, back -1 - 6 = -7, list 1 -1 - 3 = -4 -> regret 1, score -4 > -6 -> element 2 to list 1.
        // round 3: element 1: list 0 front -4 - 5 - 2 + 1 ... computed by the director; checked against the expected lists below
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = 2;
        d.working.classes[0].lists = {{}, {}};
        auto c = std::make_unique<UniConstraint>();
        c->name = "position weight";
        c->impact = Impact::Penalty;
        c->source = ChangeSource::descriptor(0);
        c->count = [](const Solution& sol) { return sol.classes[0].n; };
        c->filter = [](const Solution&, size_t) { return true; };
        c->weight = [](const Solution& sol, size_t e) {
            static const int64_t w[3] = {1, 5, 3};
            int64_t t = 0;
            const auto& l = sol.classes[0].lists[e];
            for (size_t p = 0; p < l.size(); ++p) t += (int64_t)(p + 1) * w[l[p]];
            return Score::of(0, t);
        };
        d.constraints.members.push_back(std::move(c));
        construct_list_regret(d, 0, {0, 1, 2}, nullptr);
        // round 3, element 1: list 0 = [0]: front -(5 + 2) - 3 = -10, back -(1 + 10) - 3 = -14; list 1 = [2]: front -(5 + 6) - 1 = -12, back -(3 + 10) - 1 = -14
        CHECK("list_regret.greatest_regret_then_best_score",
              d.working.classes[0].lists[0] == (std::vector<uint32_t>{1, 0}) && d.working.classes[0].lists[1] == (std::vector<uint32_t>{2}) &&
                  d.calculate_score() == Score::of(0, -10));
    }
}

static void cheapest_precedence_cases() {
    Solution s;
    s.classes.resize(1);
    s.classes[0].n = 1;
    s.classes[0].lists = {{}};
    PrecedenceHooks h;
    h.node_count = 2;
    h.durations = {1, 1};
    h.successors = {{1}, {}};
    {
        ScoreDirector d;
        d.working = s;
        construct_list_cheapest(d, 0, {1, 0}, nullptr, &h);
        CHECK("list_cheapest.precedence_downstream_breaks_cheapest_ties", d.working.classes[0].lists[0] == std::vector<uint32_t>{1, 0});
    }
    {  // without the hooks the source order decides: 1 first, then 0 in front of it
        ScoreDirector d;
        d.working = s;
        construct_list_cheapest(d, 0, {1, 0}, nullptr, nullptr);
        CHECK("list_cheapest.source_order_without_hooks", d.working.classes[0].lists[0] == std::vector<uint32_t>{0, 1});
    }
}

static void compound_scalar_cases() {
    struct SnapshotOnInsert : Constraint {  // records (left[0], left[1]) at every after_variable_changed (RecordingCompoundDirector)
        std::vector<std::pair<int64_t, int64_t>>* log;
        Score evaluate(const Solution&) const override { return Score::zero(); }
        size_t match_count(const Solution&) const override { return 0; }
        Score initialize(const Solution&) override { return Score::zero(); }
        Score on_insert(const Solution& s, size_t, size_t) override {
            log->push_back({s.classes[0].vars[0][0], s.classes[0].vars[0][1]});
            return Score::zero();
        }
        Score on_retract(const Solution&, size_t, size_t) override { return Score::zero(); }
        void reset() override {}
    };
    auto mk = [](std::vector<int64_t> left, std::vector<int64_t> right) {
        ScoreDirector d;
        d.working.classes.resize(2);
        d.working.classes[0].n = left.size();
        d.working.classes[0].vars = {left};
        d.working.classes[1].n = right.size();
        d.working.classes[1].vars = {right};
        return d;
    };
    {
        ScoreDirector d = mk(std::vector<int64_t>(8, 0), std::vector<int64_t>(8, 1));
        d.calculate_score();
        std::vector<ScalarEditO> ed = {{0, 0, 0, 2, true}, {1, 0, 0, 3, true}};
        bool ok = compound_is_doable(d, ed);
        std::vector<int64_t> u = compound_do(d, ed);
        ok = ok && d.working.classes[0].vars[0][0] == 2 && d.working.classes[1].vars[0][0] == 3;
        compound_undo(d, ed, u);
        ok = ok && d.working.classes[0].vars[0][0] == 0 && d.working.classes[1].vars[0][0] == 1;
        CHECK("compound_scalar.applies_and_undoes_atomically", ok);
    }
    {
        ScoreDirector d = mk({0, 1}, {});
        std::vector<std::pair<int64_t, int64_t>> log;
        auto c = std::make_unique<SnapshotOnInsert>();
        c->log = &log;
        d.constraints.members.push_back(std::move(c));
        d.calculate_score();
        std::vector<ScalarEditO> ed = {{0, 0, 0, 2, true}, {0, 0, 1, 3, true}};
        compound_do(d, ed);
        bool ok = d.working.classes[0].vars[0] == std::vector<int64_t>({2, 3}) && log.size() == 2 &&
                  log[0] == std::make_pair<int64_t, int64_t>(2, 3) && log[1] == std::make_pair<int64_t, int64_t>(2, 3);
        CHECK("compound_scalar.all_edits_before_after_notifications", ok);
    }
    {
        ScoreDirector d = mk({0}, {1});
        CHECK("compound_scalar.rejects_noop", !compound_is_doable(d, {{0, 0, 0, 0, true}}));
        CHECK("compound_scalar.rejects_illegal", !compound_is_doable(d, {{0, 0, 0, 2, false}}));
        CHECK("compound_scalar.rejects_empty", !compound_is_doable(d, {}));
    }
}

// heuristic/move/tests/k_opt.rs:86-222 (do / undo / doability), selector/k_opt/tests.rs:80-147 and
// selector/tests/k_opt.rs (first combination, 35 combinations, 245 moves, all doable),
// benches/selector_cursor_gate.rs:370-382 (4,760 moves on an 18-element route)
static void k_opt_cases() {
    auto mk = [](std::vector<uint32_t> cities) {
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = 1;
        d.working.classes[0].lists = {cities};
        return d;
    };
    auto kopt = [](size_t c1, size_t c2, size_t c3, size_t pattern) {
        const size_t cuts[3] = {c1, c2, c3};
        return make_kopt_move(0, 0, cuts, pattern);
    };
    {
        ScoreDirector d = mk({1, 2, 3, 4, 5, 6, 7, 8});
        Move m = kopt(2, 4, 6, 3);
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 5, 6, 3, 4, 7, 8});
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4, 5, 6, 7, 8});
        CHECK("k_opt.three_opt_swap_segments", ok);
    }
    {
        ScoreDirector d = mk({1, 2, 3, 4, 5, 6, 7, 8});
        Move m = kopt(2, 4, 6, 0);
        bool ok = move_is_doable(d, m);
        MoveUndo u = move_do(d, m);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 4, 3, 5, 6, 7, 8});
        move_undo(d, m, u);
        ok = ok && d.working.classes[0].lists[0] == std::vector<uint32_t>({1, 2, 3, 4, 5, 6, 7, 8});
        CHECK("k_opt.three_opt_reverse_segment", ok);
    }
    {
        ScoreDirector d = mk({1, 2, 3});
        CHECK("k_opt.invalid_cuts_not_doable", !move_is_doable(d, kopt(2, 4, 10, 0)));
        ScoreDirector e = mk({1, 2, 3, 4, 5, 6, 7, 8});
        CHECK("k_opt.cuts_not_sorted_not_doable", !move_is_doable(e, kopt(4, 2, 6, 0)));
    }
    {
        std::vector<size_t> cuts;
        bool ok = cut_combination_at(3, 8, 1, 0, cuts) && cuts == std::vector<size_t>({1, 2, 3});
        ok = ok && count_cut_combinations(3, 8, 1) == 35 && count_cut_combinations(3, 6, 2) == 0;
        ok = ok && kopt_binomial(5, 2) == 10 && kopt_binomial(7, 3) == 35 && kopt_binomial(10, 5) == 252;
        // every rank maps to a distinct increasing triple inside 1..=7, in lexicographic order
        std::vector<size_t> prev;
        for (size_t r = 0; r < 35; ++r) {
            ok = ok && cut_combination_at(3, 8, 1, r, cuts) && cuts[0] >= 1 && cuts[0] < cuts[1] && cuts[1] < cuts[2] && cuts[2] <= 7;
            if (r) ok = ok && prev < cuts;
            prev = cuts;
        }
        ok = ok && !cut_combination_at(3, 8, 1, 35, cuts);
        CHECK("k_opt.cut_combinations", ok);
    }
    {
        ScoreDirector d = mk({1, 2, 3, 4, 5, 6, 7, 8});
        ListSlot slot;
        KOptCursor cur(slot, d.working, MoveStreamContext(), 1);
        Move m;
        size_t count = 0;
        bool all_doable = true;
        while (cur.next(m)) {
            ++count;
            all_doable = all_doable && move_is_doable(d, m);
        }
        CHECK("k_opt.selector_generates_245_doable_moves", count == 245 && all_doable);
        std::vector<uint32_t> route(18);
        for (size_t i = 0; i < 18; ++i) route[i] = (uint32_t)i;
        ScoreDirector g = mk(route);
        KOptCursor gate(slot, g.working, MoveStreamContext(), 1);
        count = 0;
        while (gate.next(m)) ++count;
        CHECK("k_opt.selector_gate_4760", count == 4760);
    }
}

// heuristic/selector/tests/list_permute.rs:121-160 (the two cases without owner restrictions) + nth_permutation's order
static void list_permute_cases() {
    auto mk = [](std::vector<std::vector<uint32_t>> lists) {
        Solution s;
        s.classes.resize(1);
        s.classes[0].n = lists.size();
        s.classes[0].lists = lists;
        return s;
    };
    ListSlot slot;
    {
        Solution s = mk({{1, 2, 3}});
        ListPermuteCursor c(slot, s, MoveStreamContext(), 2, 3);
        std::vector<Move> moves;
        Move m;
        while (c.next(m)) moves.push_back(m);
        bool ok = moves.size() == 7 && moves[0].a_pos == 0 && moves[0].b_pos == 2 && nth_permutation(2, (size_t)moves[0].to_value) == std::vector<size_t>{1, 0};
        for (auto& mv : moves) ok = ok && mv.a == 0;
        ScoreDirector d;
        d.working = s;
        for (auto& mv : moves) ok = ok && move_is_doable(d, mv);
        CHECK("list_permute.enumerates_windows_without_batching_moves", ok);
    }
    {
        Solution s = mk({{1, 2, 3, 4}, {100, 101, 102}});
        ListPermuteCursor c(slot, s, MoveStreamContext(), 2, 3);
        std::vector<std::tuple<size_t, size_t, size_t, int64_t>> sig;
        Move m;
        while (c.next(m)) sig.push_back({m.a, m.a_pos, m.b_pos, m.to_value});
        // count_list_permute_moves_for_len: len 4 -> starts 0..3: (1 + 5) + (1 + 5) + 1 + 0 = 13; len 3 -> 7
        std::vector<std::tuple<size_t, size_t, size_t, int64_t>> uniq = sig;
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        CHECK("list_permute.size_matches_streamed_unique_candidates", sig.size() == 20 && uniq.size() == sig.size());
    }
    {
        bool ok = nth_permutation(3, 0) == std::vector<size_t>{0, 1, 2} && nth_permutation(3, 1) == std::vector<size_t>{0, 2, 1} &&
                  nth_permutation(3, 5) == std::vector<size_t>{2, 1, 0} && nth_permutation(4, 9) == std::vector<size_t>{1, 2, 3, 0};
        CHECK("list_permute.nth_permutation_is_lexicographic", ok);
    }
}

// heuristic/selector/scalar_neighborhood/tests.rs:165-204 (both cases): a dynamic slot whose nearby sources decline the row
// (return false) -> the ordinary candidate values with the source limit / every entity; meters value and |left - right|.
static void nearby_scalar_cases() {
    auto mk = [](std::vector<int64_t> values, std::vector<int64_t> candidates) {
        Solution s;
        s.classes.resize(1);
        s.classes[0].n = values.size();
        s.classes[0].vars.assign(1, values);
        ScalarSlot slot;
        slot.allows_unassigned = false;
        slot.dynamic = true;
        slot.values_for_entity = [candidates](const Solution&, size_t, std::vector<int64_t>& out) { out = candidates; };
        slot.nearby_value_distance = [](size_t, int64_t v) { return (double)v; };
        slot.nearby_entity_distance = [](size_t l, size_t r) { return (double)(l > r ? l - r : r - l); };
        return std::make_pair(s, slot);
    };
    {
        auto [s, slot] = mk({0}, {0, 2, 1});
        NearbyScalarChangeCursor c(slot, s, MoveStreamContext(), 2, 2);
        std::vector<std::pair<size_t, int64_t>> got;
        Move m;
        while (c.next(m)) got.push_back({m.a, m.to_value});
        CHECK("nearby_scalar.change_row_fallback_with_source_limit", got == std::vector<std::pair<size_t, int64_t>>{{0, 2}});
    }
    {
        auto [s, slot] = mk({0, 1, 2}, {0, 1, 2});
        NearbyScalarSwapCursor c(slot, s, MoveStreamContext(), 1);
        std::vector<std::pair<size_t, size_t>> got;
        Move m;
        while (c.next(m)) got.push_back({m.a, m.b});
        CHECK("nearby_scalar.swap_all_entity_row_fallback", got == std::vector<std::pair<size_t, size_t>>{{0, 1}, {1, 0}, {2, 1}});
    }
    {  // static slot: canonical orientation right > left only (swap.rs:362-366), the declared source rows rank by source order
        auto [s, slot] = mk({0, 1, 2, 0}, {0, 1, 2});
        slot.dynamic = false;
        slot.nearby_entity_distance = nullptr;
        slot.has_nearby_entities = true;
        slot.nearby_entities = {{3, 2, 1}, {0, 3, 2}, {1, 0, 3}, {2, 1, 0}};
        NearbyScalarSwapCursor c(slot, s, MoveStreamContext(), 2);
        std::vector<std::pair<size_t, size_t>> got;
        Move m;
        while (c.next(m)) got.push_back({m.a, m.b});
        // row 0: 3 has the same value (skipped), then 2, 1; row 1: 3, 2; row 2: 3; row 3: nothing to the right
        CHECK("nearby_scalar.swap_static_canonical_orientation",
              got == std::vector<std::pair<size_t, size_t>>{{0, 2}, {0, 1}, {1, 3}, {1, 2}, {2, 3}});
    }
}

// phase/localsearch/acceptor/diversified_late_acceptance/tests.rs:17-68 (every case; SoftScore = one level)
static void diversified_late_acceptance_cases() {
    auto S = [](int64_t v) { return Score::level(0, v); };
    {
        DiversifiedLateAcceptanceAcceptor a(5, 0.1);
        a.phase_started(S(-100));
        CHECK("dla.accepts_improving_moves", a.is_accepted(S(-100), S(-90)));
    }
    {
        DiversifiedLateAcceptanceAcceptor a(3, 0.1);
        a.phase_started(S(-100));
        CHECK("dla.accepts_late_equal", a.is_accepted(S(-90), S(-100)));
    }
    {
        DiversifiedLateAcceptanceAcceptor a(3, 0.1);
        a.phase_started(S(-100));
        a.step_ended(S(-80));
        a.step_ended(S(-70));
        a.step_ended(S(-60));
        CHECK("dla.diversification_accepts_within_tolerance", a.is_accepted(S(-60), S(-65)));  // -60 - round(60 * 0.1) = -66
    }
    {
        DiversifiedLateAcceptanceAcceptor a(3, 0.05);
        a.phase_started(S(-100));
        a.step_ended(S(-40));
        a.step_ended(S(-40));
        a.step_ended(S(-40));
        CHECK("dla.rejects_outside_tolerance", !a.is_accepted(S(-40), S(-50)));  // late -40, threshold -42
    }
    {
        DiversifiedLateAcceptanceAcceptor a(3, 0.1);
        a.phase_started(S(-100));
        a.step_ended(S(-80));
        a.step_ended(S(-70));
        a.step_ended(S(-60));
        CHECK("dla.history_cycles", a.is_accepted(S(-60), S(-75)));  // history[0] = -80
    }
}

// phase/localsearch/acceptor/simulated_annealing/tests.rs:40-273 (every case) + the published
// xoshiro256++ test vector for the SmallRng restatement.
static void simulated_annealing_cases() {
    auto mk = [](SimulatedAnnealingAcceptor::Mode mode, std::vector<double> temps, double decay, double hc,
                 bool never_hard, uint64_t seed, int levels) {
        SimulatedAnnealingAcceptor a;
        a.mode = mode;
        a.levels = levels;
        a.hard_levels = levels > 1 ? 1 : 0;
        if (mode == SimulatedAnnealingAcceptor::Single) a.single_temperature = temps[0];
        if (mode == SimulatedAnnealingAcceptor::PerLevel) a.level_temperatures = temps;
        a.decay_rate = decay;
        a.hill_climbing_temperature = hc;
        a.never_accept_hard_regression = never_hard;
        a.rng = SmallRng::seed_from_u64(seed);
        return a;
    };
    auto soft1 = [](int64_t v) { return Score::level(0, v); };  // SoftScore: one level
    {
        // xoshiro256plusplus.c reference vector (state 1,2,3,4), as asserted by rand's own test
        SmallRng r;
        r.s[0] = 1, r.s[1] = 2, r.s[2] = 3, r.s[3] = 4;
        const uint64_t expect[10] = {41943041ULL,           58720359ULL,           3588806011781223ULL,
                                     3591011842654386ULL,   9228616714210784205ULL, 9973669472204895162ULL,
                                     14011001112246962877ULL, 12406186145184390807ULL, 15849039046786891736ULL,
                                     10450023813501588000ULL};
        bool ok = true;
        for (uint64_t e : expect) ok = ok && r.next_u64() == e;
        CHECK("sa.xoshiro256pp_reference_vector", ok);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Single, {1000.0}, 0.99, 1.0e-9, false, 42, 1);
        CHECK("sa.accepts_improving_and_equal", a.is_accepted(soft1(-10), soft1(-5)) && a.is_accepted(soft1(-10), soft1(-10)));
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Single, {1000000.0}, 0.99, 1.0e-9, false, 42, 1);
        a.phase_started(soft1(0));
        int acc = 0;
        for (int i = 0; i < 100; ++i) acc += a.is_accepted(soft1(-10), soft1(-11));
        CHECK("sa.high_temperature_accepts_most", acc > 90);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Single, {0.001}, 0.99, 1.0e-9, false, 42, 1);
        a.phase_started(soft1(0));
        int acc = 0;
        for (int i = 0; i < 100; ++i) acc += a.is_accepted(soft1(-10), soft1(-20));
        CHECK("sa.low_temperature_rejects_most", acc < 5);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::PerLevel, {100.0}, 0.5, 20.0, false, 42, 1);
        a.phase_started(soft1(0));
        bool ok = a.current[0] == 100.0;
        a.step_ended(soft1(0));
        ok = ok && a.current[0] == 50.0;
        a.step_ended(soft1(0));
        ok = ok && a.current[0] == 25.0;
        a.step_ended(soft1(0));
        ok = ok && a.current[0] == 20.0;
        CHECK("sa.temperature_decays_to_hill_climbing_threshold", ok);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::PerLevel, {1.0e-9, 1.0e12}, 1.0, 1.0e-9, false, 42, 2);
        a.phase_started(Score::of(-10, -1000000));
        bool ok = true;
        for (int i = 0; i < 100; ++i) ok = ok && !a.is_accepted(Score::of(-10, -1000000), Score::of(-11, 0));
        CHECK("sa.soft_improvement_does_not_mask_hard_regression", ok);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Single, {0.0}, 1.0, 1.0e-9, false, 42, 2);
        CHECK("sa.hard_improvement_with_soft_regression_accepted", a.is_accepted(Score::of(-2, 0), Score::of(-1, -1000000)));
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::PerLevel, {0.0, 1000000.0}, 1.0, 1.0e-9, false, 42, 2);
        a.phase_started(Score::of(0, 0));
        int acc = 0;
        for (int i = 0; i < 100; ++i) acc += a.is_accepted(Score::of(0, -10), Score::of(0, -11));
        CHECK("sa.soft_regression_uses_soft_temperature", acc > 90);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::PerLevel, {1.0e12, 1.0e12}, 1.0, 1.0e-9, true, 42, 2);
        a.phase_started(Score::of(0, 0));
        CHECK("sa.never_accept_hard_regression", !a.is_accepted(Score::of(-10, 0), Score::of(-11, 10000)));
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::PerLevel, {100.0, 100.0}, 0.1, 1.1, false, 42, 2);
        HillClimbingAcceptor hill;
        a.phase_started(Score::of(0, 0));
        for (int i = 0; i < 3; ++i) a.step_ended(Score::of(0, 0));
        const Score pairs[3][2] = {{Score::of(0, 0), Score::of(0, -1)},
                                   {Score::of(-1, 0), Score::of(-2, 10000)},
                                   {Score::of(-1, 0), Score::of(0, -10000)}};
        bool ok = true;
        for (auto& pr : pairs) ok = ok && a.is_accepted(pr[0], pr[1]) == hill.is_accepted(pr[0], pr[1]);
        CHECK("sa.cooled_matches_hill_climbing", ok);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Calibrated, {}, 1.0, 1.0e-9, false, 42, 2);
        a.sample_size = 2;
        a.target_acceptance_probability = 0.5;
        a.fallback_temperature = 1.0;
        a.phase_started(Score::of(0, 0));
        bool ok = !a.is_accepted(Score::of(0, 0), Score::of(-4, 0));
        (void)a.is_accepted(Score::of(0, 0), Score::of(0, -10));
        // 4/ln2 = 5.77..., 10/ln2 = 14.42...
        ok = ok && a.current[0] > 5.0 && a.current[1] > 14.0 && !a.calibrating;
        ok = ok && a.current[0] == 4.0 / -std::log(0.5) && a.current[1] == 10.0 / -std::log(0.5);
        CHECK("sa.sampled_calibration_per_level", ok);
    }
    {
        auto a = mk(SimulatedAnnealingAcceptor::Calibrated, {}, 0.999, 1.0e-9, false, 42, 2);
        auto b = mk(SimulatedAnnealingAcceptor::Calibrated, {}, 0.999, 1.0e-9, false, 42, 2);
        a.phase_started(Score::of(-576, -1000));
        b.phase_started(Score::of(-576, -1000));
        CHECK("sa.seeded_calibration_same_start", a.current == b.current && a.current[0] == 0.0 && a.calibrating);
    }
}

// phase/localsearch/forager/tests.rs: the forager known answers (values re-derived from the assertions)
static void forager_cases() {
    auto soft1 = [](int64_t v) { return Score::level(0, v); };
    auto mk = [&](Forager::Kind k, size_t limit) {
        Forager f;
        f.kind = k;
        f.accepted_count_limit = limit;
        f.best.random_ties = false;
        return f;
    };
    {  // tests.rs:30-75 accepted_count_forager_retains_only_the_first_strict_best_candidate
        Forager f = mk(Forager::AcceptedCount, 4);
        f.step_started(0, soft1(0), soft1(0));
        f.add_move_index(0, soft1(-10));
        f.add_move_index(1, soft1(-12));
        f.add_move_index(2, soft1(-5));
        f.add_move_index(3, soft1(-5));
        CHECK("forager.accepted_count_first_strict_best", f.best.has && f.best.index == 2 && f.best.score == soft1(-5));
    }
    {  // tests.rs:77-120 test_accepted_count_forager_quits_at_limit
        Forager f = mk(Forager::AcceptedCount, 3);
        f.step_started(0, soft1(0), soft1(0));
        f.add_move_index(0, soft1(-10));
        bool ok = !f.is_quit_early();
        f.add_move_index(1, soft1(-5));
        ok = ok && !f.is_quit_early();
        f.add_move_index(2, soft1(-8));
        CHECK("forager.accepted_count_quits_at_limit", ok && f.is_quit_early());
    }
    {  // tests.rs:462-499 test_first_best_score_improving_quits_on_improvement
        Forager f = mk(Forager::FirstBestScoreImproving, 0);
        f.step_started(0, soft1(-10), soft1(0));
        f.add_move_index(0, soft1(-15));
        bool ok = !f.is_quit_early();
        f.add_move_index(1, soft1(-5));
        ok = ok && f.is_quit_early();
        CHECK("forager.first_best_score_improving", ok && f.best.has && f.best.index == 1 && f.best.score == soft1(-5));
    }
    {  // tests.rs:501-540 test_first_last_step_improving_quits_on_improvement
        Forager f = mk(Forager::FirstLastStepScoreImproving, 0);
        f.step_started(0, soft1(0), soft1(-10));
        f.add_move_index(0, soft1(-15));
        bool ok = !f.is_quit_early();
        f.add_move_index(1, soft1(-5));
        ok = ok && f.is_quit_early();
        CHECK("forager.first_last_step_improving", ok && f.best.has && f.best.index == 1 && f.best.score == soft1(-5));
    }
    {  // tests.rs:542-587 first_last_step_improving_falls_back_to_accepted_count_limit
        Forager f = mk(Forager::FirstLastStepScoreImproving, 2);
        f.step_started(0, soft1(0), soft1(-10));
        f.add_move_index(0, soft1(-15));
        bool ok = !f.is_quit_early();
        f.add_move_index(1, soft1(-12));
        ok = ok && f.is_quit_early();
        f.add_move_index(2, soft1(-5));  // released: the limit was reached
        CHECK("forager.first_last_step_improving_limit", ok && f.best.has && f.best.index == 1 && f.best.score == soft1(-12) &&
                                                              f.limit_for_context() == 2);
    }
    {  // improving.rs:91-100: after an improving candidate every later accepted candidate is released
        Forager f = mk(Forager::FirstBestScoreImproving, 0);
        f.step_started(0, soft1(-10), soft1(0));
        f.add_move_index(0, soft1(-5));
        f.add_move_index(1, soft1(-20));
        f.add_move_index(2, soft1(-1));  // a later improving candidate replaces (score > best_score is tested first)
        CHECK("forager.first_best_score_improving_later_better_replaces", f.best.index == 2 && f.limit_for_context() == -1);
    }
}

// constraint/list_precedence.rs:905-1176 (the constraint's own test module): the two-task plan (durations 2, 3; fixed edge
// 0 -> 1; expected owners 0, 1), the four-task plan and the bare graph states.
struct PrecPlan {
    std::vector<int64_t> duration, next, owner;
};
static ListPrecedenceConstraint prec_constraint(std::shared_ptr<PrecPlan> plan, bool with_owner) {
    ListPrecedenceConstraint c;
    c.name = "listPrecedenceMakespan";
    c.list_descriptor = 0;
    c.node_count = [plan](const Solution&) { return plan->duration.size(); };
    c.node_duration = [plan](const Solution&, size_t n) { return plan->duration[n]; };
    c.fixed_successors = [plan](const Solution&, size_t n, std::vector<size_t>& out) {
        if (plan->next[n] != NONE) out.push_back((size_t)plan->next[n]);
    };
    c.owner_count = [](const Solution& s) { return s.classes[0].n; };
    c.list_len = [](const Solution& s, size_t o) { return s.classes[0].lists[o].size(); };
    c.list_get = [](const Solution& s, size_t o, size_t p) { return (int64_t)s.classes[0].lists[o][p]; };
    if (with_owner) c.expected_owner = [plan](const Solution&, size_t n) { return plan->owner[n]; };
    return c;
}
static Solution prec_routes(std::vector<std::vector<uint32_t>> routes) {
    Solution s;
    s.classes.resize(1);
    s.classes[0].n = routes.size();
    s.classes[0].lists = routes;
    return s;
}
static ListPrecedenceConstraint::State prec_graph(size_t n, std::vector<int64_t> dur, std::vector<std::pair<size_t, size_t>> edges) {
    ListPrecedenceConstraint::State st(n, 0, dur);
    for (auto& e : edges) st.add_edge(e);
    st.rebuild_graph_summary();
    st.refresh_penalty();
    return st;
}
static void list_precedence_cases() {
    using C = ListPrecedenceConstraint;
    using R = C::Refresh;
    auto two = std::make_shared<PrecPlan>(PrecPlan{{2, 3}, {1, NONE}, {0, 1}});
    auto four = std::make_shared<PrecPlan>(PrecPlan{{1, 1, 1, 1}, {NONE, NONE, NONE, NONE}, {0, 0, 0, 0}});
    {  // evaluates_fixed_and_list_precedence_makespan (:905-913)
        auto c = prec_constraint(two, false);
        CHECK("list_precedence.evaluate", c.evaluate(prec_routes({{0}, {1}})) == Score::of(0, -5));
    }
    {  // acyclic_graph_route_change_uses_incremental_descendant_refresh (:915-934)
        auto st = prec_graph(4, {2, 3, 5, 7}, {{0, 1}, {1, 2}, {0, 3}});
        bool ok = st.cycle_penalty == 0 && st.makespan == 10;
        C::RouteChange ch;
        if (st.remove_edge({1, 2})) ch.removed.push_back({1, 2});
        ok = ok && st.refresh_graph_after_route_change(ch) == R::Incremental && st.last_visited == 1;
        ok = ok && st.cycle_penalty == 0 && st.earliest[2] == 0 && st.earliest[3] == 2 && st.makespan == 9;
        CHECK("list_precedence.incremental_descendant_refresh", ok);
    }
    {  // cycle_introducing_route_change_marks_cyclic_without_graph_rebuild (:936-953)
        auto st = prec_graph(3, {1, 1, 1}, {{0, 1}, {1, 2}});
        bool ok = st.cycle_penalty == 0 && st.makespan == 3;
        C::RouteChange ch;
        if (st.add_edge({2, 0})) ch.added.push_back({2, 0});
        ok = ok && st.refresh_graph_after_route_change(ch) == R::CycleDetected && st.cycle_penalty == 3 && st.makespan == 0;
        CHECK("list_precedence.cycle_detected", ok);
        // cycle_retraction_recovers_cached_acyclic_state_without_graph_rebuild (:955-981)
        C::RouteChange undo;
        if (st.remove_edge({2, 0})) undo.removed.push_back({2, 0});
        ok = ok && st.refresh_graph_after_route_change(undo) == R::CycleRecovered && st.cycle_penalty == 0 &&
             st.earliest == std::vector<int64_t>({0, 1, 2}) && st.makespan == 3;
        CHECK("list_precedence.cycle_recovered", ok);
    }
    {  // cycle_introduced_with_removed_edges_recovers_by_full_rebuild (:983-1013)
        auto st = prec_graph(3, {10, 10, 1}, {{0, 1}, {1, 2}});
        bool ok = st.makespan == 21;
        C::RouteChange ch;
        if (st.remove_edge({0, 1})) ch.removed.push_back({0, 1});
        if (st.add_edge({2, 1})) ch.added.push_back({2, 1});
        ok = ok && st.refresh_graph_after_route_change(ch) == R::CycleDetected && st.cycle_penalty == 3 && st.cycle_added_edges.empty();
        C::RouteChange undo;
        if (st.remove_edge({2, 1})) undo.removed.push_back({2, 1});
        ok = ok && st.refresh_graph_after_route_change(undo) == R::Full && st.cycle_penalty == 0 &&
             st.earliest == std::vector<int64_t>({0, 0, 10}) && st.makespan == 11;
        CHECK("list_precedence.cycle_with_removed_edges_full_rebuild", ok);
    }
    {  // owner_route_replacement_diffs_unchanged_prefix_edges (:1015-1031)
        auto c = prec_constraint(four, false);
        C::State st(4, 1, {1, 1, 1, 1});
        st.replace_owner_route(0, c.owner_route_snapshot(st, prec_routes({{0, 1, 2}}), 0));
        st.rebuild_graph_summary();
        auto ch = st.replace_owner_route(0, c.owner_route_snapshot(st, prec_routes({{0, 1, 3}}), 0));
        std::sort(ch.added.begin(), ch.added.end());
        std::sort(ch.removed.begin(), ch.removed.end());
        bool ok = ch.removed == std::vector<C::Edge>({{1, 2}}) && ch.added == std::vector<C::Edge>({{1, 3}}) &&
                  st.assigned_counts == std::vector<size_t>({1, 1, 0, 1}) && st.owner_edges[0] == std::vector<C::Edge>({{0, 1}, {1, 3}});
        CHECK("list_precedence.route_replacement_diff", ok);
    }
    {  // cyclic_state_with_unmatched_change_uses_full_graph_refresh (:1033-1056)
        auto st = prec_graph(4, {1, 1, 1, 1}, {{0, 1}, {1, 2}});
        C::RouteChange ch;
        if (st.add_edge({2, 0})) ch.added.push_back({2, 0});
        bool ok = st.refresh_graph_after_route_change(ch) == R::CycleDetected;
        C::RouteChange un;
        if (st.add_edge({2, 3})) un.added.push_back({2, 3});
        ok = ok && st.refresh_graph_after_route_change(un) == R::Full && st.cycle_penalty == 4 && st.makespan == 0;
        CHECK("list_precedence.cyclic_unmatched_change_full_refresh", ok);
    }
    {  // partial_cycle_penalizes_whole_precedence_schedule (:1058-1065): assignment penalty 4 + cycle penalty 4
        auto st = prec_graph(4, {1, 1, 1, 10}, {{0, 1}, {1, 0}});
        CHECK("list_precedence.partial_cycle", st.cycle_penalty == 4 && st.makespan == 0 && st.hard_penalty == 8);
    }
    auto director_with = [&](std::vector<std::vector<uint32_t>> routes, bool owner) {
        ScoreDirector d;
        d.working = prec_routes(routes);
        d.constraints.members.push_back(std::make_unique<ListPrecedenceConstraint>(prec_constraint(two, owner)));
        return d;
    };
    {  // penalizes_cycles_incrementally (:1067-1079)
        auto d = director_with({{0, 1}}, false);
        bool ok = d.calculate_score() == Score::of(0, -5);
        d.before_variable_changed(0, 0);
        d.working.classes[0].lists[0] = {1, 0};
        d.after_variable_changed(0, 0);
        Score sc = d.calculate_score();
        ok = ok && sc == Score::of(-2, 0) && d.fresh_score() == sc;
        CHECK("list_precedence.cycles_incrementally", ok);
    }
    {  // repeated_route_updates_keep_incremental_score_fresh (:1081-1101)
        auto d = director_with({{0}, {1}}, false);
        bool ok = d.fresh_score() == d.calculate_score();
        std::vector<std::pair<size_t, std::vector<uint32_t>>> upd = {{0, {0, 1}}, {1, {}}, {0, {1, 0}}, {1, {0, 1}}};
        for (auto& u : upd) {
            d.before_variable_changed(0, u.first);
            d.working.classes[0].lists[u.first] = u.second;
            d.after_variable_changed(0, u.first);
            ok = ok && d.fresh_score() == d.calculate_score();
        }
        CHECK("list_precedence.repeated_updates_fresh", ok);
    }
    {  // retract_removes_cached_owner_route_before_insert (:1103-1118)
        auto c = prec_constraint(two, false);
        Solution s = prec_routes({{0}, {1}});
        Score sc = c.initialize(s);
        bool ok = sc == Score::of(0, -5);
        sc = sc + c.on_retract(s, 0, 0);
        ok = ok && sc == Score::of(-1, -5);
        s.classes[0].lists[0] = {0, 1};
        sc = sc + c.on_insert(s, 0, 0);
        ok = ok && sc == Score::of(-1, -5) && c.evaluate(s) == sc;
        CHECK("list_precedence.retract_before_insert", ok);
    }
    {  // incremental_route_update_only_reads_changed_owner_route (:1120-1154): the scores of that sequence
        auto d = director_with({{0}, {1}}, false);
        bool ok = d.calculate_score() == Score::of(0, -5);
        d.before_variable_changed(0, 0);
        d.working.classes[0].lists[0] = {0, 1};
        d.after_variable_changed(0, 0);
        ok = ok && d.calculate_score() == Score::of(-1, -5) && d.fresh_score() == d.calculate_score();
        CHECK("list_precedence.incremental_route_update", ok);
    }
    {  // penalizes_missing_duplicate_and_wrong_owner_assignments (:1156-1164)
        auto c = prec_constraint(two, true);
        CHECK("list_precedence.duplicate_and_wrong_owner", c.evaluate(prec_routes({{0, 1}, {1}})) == Score::of(-2, -5));
    }
    {  // ignores_unrelated_descriptor_changes (:1166-1176)
        auto d = director_with({{0}, {1}}, false);
        Score sc = d.calculate_score();
        d.before_variable_changed(1, 0);
        d.working.classes[0].lists[0] = {1, 0};
        d.after_variable_changed(1, 0);
        CHECK("list_precedence.unrelated_descriptor", d.calculate_score() == sc);
    }
}

// stream/collector/tests/collector.rs:266-320,399-469 (the seven consecutive_runs tests)
// heuristic/selector/tests/list_precedence.rs:170-623 (all twelve cases): tasks = (duration, fixed successor or none), routes per
// machine; index_to_element = identity.
static void list_precedence_selector_cases() {
    struct Task {
        int64_t duration;
        int64_t next;  // -1 = none
    };
    struct Fixture {
        Solution s;
        ListSlot slot;
    };
    auto mk = [](std::vector<Task> tasks, std::vector<std::vector<uint32_t>> routes) {
        Fixture f;
        f.s.classes.resize(1);
        f.s.classes[0].n = routes.size();
        f.s.classes[0].lists = routes;
        auto h = std::make_shared<PrecedenceHooks>();
        h->node_count = tasks.size();
        for (auto& t : tasks) {
            h->durations.push_back(t.duration);
            h->successors.push_back(t.next >= 0 ? std::vector<size_t>{(size_t)t.next} : std::vector<size_t>{});
        }
        f.slot.precedence = h;
        return f;
    };
    auto stream = [](Fixture& f, MoveStreamContext ctx = MoveStreamContext()) {
        ListPrecedenceCursor c(f.slot, f.s, ctx);
        std::vector<Move> moves;
        Move m;
        while (c.next(m)) moves.push_back(m);
        return moves;
    };
    auto size_of = [](Fixture& f) {
        std::vector<size_t> entities(f.s.classes[0].n);
        for (size_t e = 0; e < entities.size(); ++e) entities[e] = e;
        return precedence_selector_size(critical_analysis(*f.slot.precedence, f.s.classes[0].lists, entities));
    };
    auto is = [](const Move& m, Move::Kind k, size_t a_pos, size_t b_pos) { return m.kind == k && m.a == 0 && m.b == 0 && m.a_pos == a_pos && m.b_pos == b_pos; };
    auto sig = [](const Move& m) {
        std::vector<int64_t> v{(int64_t)m.kind, (int64_t)m.a, (int64_t)m.a_pos, (int64_t)m.b, (int64_t)m.b_pos, m.to_value, (int64_t)m.ruin_multi};
        for (int i = 0; i < 8; ++i) v.push_back(m.kind == Move::Ruin ? m.ruin_idx[i] : 0), v.push_back(m.kind == Move::Ruin && m.ruin_multi ? m.ruin_src[i] : 0);
        for (int i = 0; i < 4; ++i)
            if (m.kind == Move::MultiSwap) v.push_back(m.ms_entity[i]), v.push_back(m.ms_first[i]), v.push_back(m.ms_second[i]);
        return v;
    };
    auto all_unique = [&](const std::vector<Move>& moves) {
        std::vector<std::vector<int64_t>> sigs;
        for (auto& m : moves) sigs.push_back(sig(m));
        std::sort(sigs.begin(), sigs.end());
        return std::unique(sigs.begin(), sigs.end()) == sigs.end();
    };
    auto ruin_window = [](const Move& m) {
        std::vector<size_t> w;
        for (size_t i = 0; i < m.a_pos; ++i) w.push_back(m.ruin_idx[i]);
        return w;
    };
    {  // :170-240
        Fixture f = mk({{3, -1}, {4, -1}, {5, -1}}, {{0, 1, 2}});
        auto mv = stream(f);
        bool ok = size_of(f) == 24 && mv.size() == 24;
        ok = ok && is(mv[0], Move::ListChange, 0, 2) && is(mv[1], Move::ListChange, 1, 3) && is(mv[2], Move::ListChange, 0, 3) &&
             is(mv[3], Move::ListChange, 2, 0) && is(mv[4], Move::ListChange, 2, 1) && is(mv[5], Move::ListChange, 1, 0);
        ok = ok && is(mv[6], Move::ListSwap, 0, 1) && is(mv[7], Move::ListSwap, 0, 2) && is(mv[8], Move::ListSwap, 1, 2);
        ok = ok && is(mv[9], Move::ListReverse, 0, 2) && is(mv[10], Move::ListReverse, 0, 3) && is(mv[11], Move::ListReverse, 1, 3);
        // SublistSwap [0,1) <-> [1,3) and [0,2) <-> [2,3)
        ok = ok && is(mv[12], Move::SublistSwap, 0, 1) && mv[12].to_value == (1 | (2 << 16)) && is(mv[13], Move::SublistSwap, 0, 2) &&
             mv[13].to_value == (2 | (1 << 16));
        ok = ok && mv[14].kind == Move::Ruin && mv[14].a == 0 && !mv[14].ruin_multi && mv[14].prec != nullptr && ruin_window(mv[14]) == std::vector<size_t>{0, 1, 2};
        ok = ok && is(mv[15], Move::SublistChange, 0, 1) && mv[15].to_value == 2 && is(mv[16], Move::SublistChange, 1, 0) && mv[16].to_value == 3;
        ok = ok && is(mv[17], Move::ListPermute, 0, 2) && nth_permutation(2, (size_t)mv[17].to_value) == std::vector<size_t>{1, 0};
        ok = ok && is(mv[18], Move::ListPermute, 0, 3) && nth_permutation(3, (size_t)mv[18].to_value) == std::vector<size_t>{0, 2, 1};
        CHECK("list_precedence_selector.emits_critical_block_moves", ok);
    }
    {  // :242-274
        Fixture f = mk({{3, -1}, {4, -1}, {5, -1}, {2, 2}}, {{0, 1, 2}, {3}});
        auto mv = stream(f);
        CHECK("list_precedence_selector.size_matches_streamed_unique_candidates", mv.size() == size_of(f) && all_unique(mv) && !mv.empty());
    }
    {  // :276-307
        std::vector<Task> tasks;
        for (int i = 0; i < 7; ++i) tasks.push_back({i + 1, -1});
        Fixture f = mk(tasks, {{0, 1, 2, 3, 4, 5, 6}});
        auto mv = stream(f);
        std::vector<std::vector<size_t>> windows;
        for (auto& m : mv)
            if (m.kind == Move::Ruin) windows.push_back(ruin_window(m));
        CHECK("list_precedence_selector.streams_all_critical_ruin_windows",
              mv.size() == size_of(f) && windows == std::vector<std::vector<size_t>>{{0, 1, 2, 3, 4}, {1, 2, 3, 4, 5}, {2, 3, 4, 5, 6}});
    }
    {  // :309-335
        std::vector<Task> tasks(6, Task{5, -1});
        Fixture f = mk(tasks, {{0, 1, 2}, {3, 4, 5}});
        auto mv = stream(f);
        std::vector<Move> multi;
        for (auto& m : mv)
            if (m.kind == Move::Ruin && m.ruin_multi) multi.push_back(m);
        bool ok = mv.size() == size_of(f) && multi.size() == 9;
        for (auto& m : multi) ok = ok && m.a_pos == 2;
        ok = ok && !multi.empty() && multi[0].ruin_src[0] == 0 && multi[0].ruin_src[1] == 1;
        CHECK("list_precedence_selector.streams_multi_block_ruin_candidates", ok);
    }
    {  // :337-383
        Fixture f = mk({{5, -1}, {5, 4}, {10, -1}, {5, 5}, {5, -1}, {5, -1}}, {{0, 1}, {2, 3}, {4, 5}});
        auto mv = stream(f);
        bool any = false;
        for (auto& m : mv)
            if (m.kind == Move::MultiSwap && m.require_improvement && m.a_pos == 3 && m.ms_entity[0] == 0 && m.ms_first[0] == 0 && m.ms_second[0] == 1 &&
                m.ms_entity[1] == 1 && m.ms_first[1] == 0 && m.ms_second[1] == 1 && m.ms_entity[2] == 2 && m.ms_first[2] == 0 && m.ms_second[2] == 1)
                any = true;
        CHECK("list_precedence_selector.streams_fixed_successor_support_multi_swaps", mv.size() == size_of(f) && any);
    }
    {  // :385-429
        Fixture f = mk({{3, -1}, {4, -1}, {5, -1}, {6, -1}}, {{0, 1, 2, 3}});
        auto canonical = stream(f);
        auto contextual = stream(f, MoveStreamContext(11, 23, 8).with_selection_order(SelectionOrder::Shuffled));
        std::vector<std::vector<int64_t>> a, b;
        for (auto& m : canonical) a.push_back(sig(m));
        for (auto& m : contextual) b.push_back(sig(m));
        bool differs = a != b;
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        CHECK("list_precedence_selector.applies_stream_context_order", a.size() == b.size() && a == b && differs);
    }
    {  // :431-475: MoveStreamContext::new(11, 23, Some(8)) keeps the default order; the first tier stays the adjacent changes
        Fixture f = mk({{3, -1}, {4, -1}, {5, -1}, {6, -1}}, {{0, 1, 2, 3}});
        auto mv = stream(f, MoveStreamContext(11, 23, 8));
        bool ok = mv.size() >= 3;
        for (size_t i = 0; i < 3 && ok; ++i) ok = mv[i].kind == Move::ListChange && mv[i].a == 0 && mv[i].b == 0 && mv[i].b_pos == mv[i].a_pos + 2;
        CHECK("list_precedence_selector.stream_context_preserves_adjacent_priority_tier_for_short_blocks", ok);
    }
    {  // :477-520
        Fixture f = mk({{3, 1}, {2, -1}, {5, -1}}, {{0, 2}, {1}});
        auto mv = stream(f);
        bool ok = mv.size() == 6 && is(mv[0], Move::ListChange, 0, 2) && is(mv[1], Move::ListChange, 1, 0) && is(mv[2], Move::ListSwap, 0, 1) &&
                  is(mv[3], Move::ListReverse, 0, 2) && mv[4].kind == Move::Ruin && mv[4].a == 0 && ruin_window(mv[4]) == std::vector<size_t>{0, 1} &&
                  is(mv[5], Move::ListPermute, 0, 2) && nth_permutation(2, (size_t)mv[5].to_value) == std::vector<size_t>{1, 0};
        CHECK("list_precedence_selector.ignores_noncritical_route_arcs", ok);
    }
    {  // :522-547
        Fixture f = mk({{1, 1}, {1, -1}}, {{0, 1}});
        auto mv = stream(f);
        CHECK("list_precedence_selector.skips_moves_that_force_fixed_successor_cycles",
              size_of(f) == 1 && mv.size() == 1 && mv[0].kind == Move::Ruin && mv[0].a == 0 && ruin_window(mv[0]) == std::vector<size_t>{0, 1});
    }
    {  // :549-585
        Fixture f = mk({{1, -1}, {5, -1}, {5, -1}, {10, 1}}, {{0, 1, 2}, {3}});
        auto mv = stream(f);
        bool any = false, all = true;
        for (auto& m : mv)
            if (m.kind == Move::SublistChange) {
                any = any || (m.a_pos == 1 && m.to_value == 3 && m.b_pos == 0);
                all = all && m.a_pos != m.b_pos;
            }
        CHECK("list_precedence_selector.sublist_destinations_use_route_coordinates", mv.size() == size_of(f) && any && all);
    }
    {  // :587-611
        Fixture f = mk({{3, 2}, {1, -1}, {5, -1}}, {{0, 1}, {2}});
        auto mv = stream(f);
        bool any = false;
        for (auto& m : mv) any = any || (m.kind == Move::ListChange && m.a == 0 && m.a_pos == 0 && m.b_pos == 2);
        CHECK("list_precedence_selector.emits_singleton_critical_node_relocations", mv.size() == size_of(f) && any);
    }
    {  // heuristic/selector/tests/list_ruin.rs:279-303 (precedence_ruin_recreate_skips_cycle_forming_insertions): elements 0, 1, 2 for the
       // reference's 1, 2, 3 (index_to_element = identity here), fixed successor 0 -> 1, no constraint (every insertion scores the same):
       // the ruined middle element may not go in front of its predecessor, so the first acyclic position puts it back
        Fixture f = mk({{1, 1}, {1, -1}, {1, -1}}, {{0, 1, 2}});
        ScoreDirector d;
        d.working = f.s;
        Move m;
        m.kind = Move::Ruin;
        m.a = m.b = 0;
        m.a_pos = 1;
        m.ruin_idx[0] = 1;
        m.prec = f.slot.precedence.get();
        MoveUndo ignored = move_do(d, m);
        (void)ignored;
        bool with_hooks = d.working.classes[0].lists[0] == std::vector<uint32_t>{0, 1, 2};
        ScoreDirector d2;
        d2.working = f.s;
        m.prec = nullptr;  // without the hooks the first position wins
        MoveUndo ignored2 = move_do(d2, m);
        (void)ignored2;
        CHECK("list_ruin.precedence_ruin_recreate_skips_cycle_forming_insertions", with_hooks && d2.working.classes[0].lists[0] == std::vector<uint32_t>{1, 0, 2});
    }
    {  // :613-623
        Fixture f = mk({{1, 1}, {1, -1}}, {{1, 0}});
        CHECK("list_precedence_selector.skips_cyclic_current_graph", size_of(f) == 0 && stream(f).empty());
    }
}

static void runs_cases() {
    auto feed = [](std::vector<int64_t> v) {
        RunsAccumulator a;
        for (int64_t x : v) a.accumulate(x);
        return a;
    };
    {  // test_consecutive_runs_empty
        Runs r = RunsAccumulator().finish();
        CHECK("runs.empty", r.runs.empty() && r.point_count == 0 && r.item_count == 0);
    }
    {  // test_consecutive_runs_one_run
        Runs r = feed({3, 1, 2}).finish();
        CHECK("runs.one_run", r.runs.size() == 1 && r.runs[0].start == 1 && r.runs[0].end == 3 && r.runs[0].point_count == 3 && r.runs[0].item_count == 3);
    }
    {  // test_consecutive_runs_multiple_runs
        Runs r = feed({8, 1, 2, 4, 5, 10}).finish();
        bool ok = r.runs.size() == 4 && r.runs[0].start == 1 && r.runs[0].end == 2 && r.runs[1].start == 4 && r.runs[1].end == 5 &&
                  r.runs[2].start == 8 && r.runs[2].end == 8 && r.runs[3].start == 10 && r.runs[3].end == 10;
        CHECK("runs.multiple_runs", ok);
    }
    {  // test_consecutive_runs_duplicates_count_items_not_points
        Runs r = feed({1, 1, 2, 4, 4, 4}).finish();
        bool ok = r.point_count == 3 && r.item_count == 6 && r.runs[0].point_count == 2 && r.runs[0].item_count == 3 &&
                  r.runs[1].point_count == 1 && r.runs[1].item_count == 3;
        CHECK("runs.duplicates", ok);
    }
    {  // test_consecutive_runs_negative_indexes
        Runs r = feed({-3, -2, -1, 1}).finish();
        CHECK("runs.negative_indexes", r.runs.size() == 2 && r.runs[0].start == -3 && r.runs[0].end == -1 && r.runs[1].start == 1 && r.runs[1].end == 1);
    }
    {  // test_consecutive_runs_i64_max_boundary
        Runs r = feed({INT64_MIN, INT64_MAX - 1, INT64_MAX}).finish();
        bool ok = r.runs.size() == 2 && r.runs[0].start == INT64_MIN && r.runs[0].end == INT64_MIN && r.runs[1].start == INT64_MAX - 1 &&
                  r.runs[1].end == INT64_MAX;
        CHECK("runs.i64_max_boundary", ok);
    }
    {  // test_consecutive_runs_insert_retract_parity
        RunsAccumulator a = feed({1, 2, 2, 3, 7});
        a.retract(2);
        Runs r = a.finish();
        bool ok = r.runs.size() == 2 && r.runs[0].item_count == 3 && r.item_count == 4;
        a.retract(2);
        r = a.finish();
        ok = ok && r.runs.size() == 3 && r.point_count == 3 && r.item_count == 3;
        CHECK("runs.insert_retract_parity", ok);
    }
}

// constraint/tests/complemented.rs:34-420: shifts (class 0, vars[0] = employee id or NONE) grouped by employee with count(),
// complemented by the employees (class 1, vars[0] = id)
static ComplementedGroupedConstraint complemented(std::function<Score(int64_t, int64_t)> w, int64_t dflt = 0) {
    ComplementedGroupedConstraint c;
    c.name = "Shift count";
    c.impact = Impact::Penalty;
    c.a_source = ChangeSource::descriptor(0);
    c.b_source = ChangeSource::descriptor(1);
    c.a_count = [](const Solution& s) { return s.classes[0].n; };
    c.b_count = [](const Solution& s) { return s.classes[1].n; };
    c.key_a = [](const Solution& s, size_t i) { return s.classes[0].vars[0][i]; };
    c.key_b = [](const Solution& s, size_t i) { return s.classes[1].vars[0][i]; };
    c.value = [](const Solution&, size_t) { return (int64_t)1; };
    c.default_b = [dflt](const Solution&, size_t) { return dflt; };
    c.weight = w;
    return c;
}
static Solution shifts_employees(std::vector<int64_t> shifts, std::vector<int64_t> employees) {
    Solution s;
    s.classes.resize(2);
    s.classes[0].n = shifts.size();
    s.classes[0].vars = {shifts};
    s.classes[1].n = employees.size();
    s.classes[1].vars = {employees};
    return s;
}
static void complemented_cases() {
    auto lin = [](int64_t, int64_t c) { return soft(c); };
    auto sq = [](int64_t, int64_t c) { return soft(c * c); };
    {  // test_complemented_evaluate (:34-70), test_complemented_skips_none_keys (:72-112)
        auto c = complemented(lin);
        CHECK("complemented.evaluate", c.evaluate(shifts_employees({0, 0}, {0, 1})) == soft(-2));
        CHECK("complemented.skips_none_keys", c.evaluate(shifts_employees({0, 0, NONE, NONE}, {0, 1})) == soft(-2));
    }
    {  // test_complemented_incremental (:114-168)
        auto c = complemented(lin);
        Solution s = shifts_employees({0, 0, 1}, {0, 1, 2});
        bool ok = c.initialize(s) == soft(-3) && c.on_retract(s, 0, 0) == soft(1) && c.on_insert(s, 0, 0) == soft(-1);
        CHECK("complemented.incremental", ok);
    }
    {  // test_complemented_incremental_with_none_keys (:170-217)
        auto c = complemented(lin);
        Solution s = shifts_employees({0, NONE, 0}, {0, 1});
        bool ok = c.initialize(s) == soft(-2) && c.on_retract(s, 1, 0) == soft(0) && c.on_insert(s, 1, 0) == soft(0);
        CHECK("complemented.incremental_none_keys", ok);
    }
    {  // test_complemented_with_default (:219-255)
        auto c = complemented(sq);
        CHECK("complemented.with_default", c.evaluate(shifts_employees({0, 0, 0}, {0, 1, 2})) == soft(-9));
    }
    {  // test_complemented_incremental_matches_evaluate (:257-314)
        auto c = complemented(sq);
        Solution s = shifts_employees({0, 0, 1}, {0, 1});
        Score t = c.initialize(s);
        bool ok = t == c.evaluate(s) && t == soft(-5);
        t = t + c.on_retract(s, 2, 0);
        ok = ok && t == soft(-4);
        t = t + c.on_insert(s, 2, 0);
        CHECK("complemented.incremental_matches_evaluate", ok && t == soft(-5));
    }
    {  // test_complemented_b_side_insert_and_retract (:316-353)
        auto c = complemented(lin);
        Solution s = shifts_employees({0}, {0});
        Score t = c.initialize(s);
        bool ok = t == soft(-1);
        t = t + c.on_retract(s, 0, 1);
        s.classes[1].vars[0][0] = 2;
        t = t + c.on_insert(s, 0, 1);
        CHECK("complemented.b_side", ok && t == soft(0) && t == c.evaluate(s));
    }
    {  // test_complemented_missing_group_weight_can_use_complement_key (:355-378)
        auto c = complemented([](int64_t k, int64_t cnt) { return soft(k + cnt); });
        CHECK("complemented.complement_key_weight", c.evaluate(shifts_employees({1}, {1, 3})) == soft(-5));
    }
    {  // test_complemented_duplicate_complement_keys_match_incremental (:380-420): default 5, two B rows share key 0
        auto c = complemented(lin, 5);
        Solution s = shifts_employees({0}, {0, 0, 1});
        Score t = c.initialize(s);
        bool ok = t == soft(-7) && t == c.evaluate(s);
        t = t + c.on_retract(s, 0, 0);
        s.classes[0].vars[0][0] = NONE;
        t = t + c.on_insert(s, 0, 0);
        ok = ok && t == soft(-15) && t == c.evaluate(s);
        t = t + c.on_retract(s, 0, 0);
        s.classes[0].vars[0][0] = 0;
        t = t + c.on_insert(s, 0, 0);
        CHECK("complemented.duplicate_complement_keys", ok && t == soft(-7) && t == c.evaluate(s));
    }
}

// ---- Clarke-Wright savings construction (manager/phase_factory/list_clarke_wright/tests.rs, tests/metric_class.rs) ----------
// The reference's toy Plan: customer_values = the declared elements (source order), routes = owner lists; an element's source
// key is its value (usize_element_source_key).
struct CwPlan {
    std::vector<size_t> customer_values;
    std::vector<std::vector<size_t>> routes;
};
static std::vector<size_t> cw_unassigned(const CwPlan& p) {  // runtime_list_source.rs:185-223
    std::vector<size_t> out;
    for (size_t i = 0; i < p.customer_values.size(); ++i) {
        bool assigned = false;
        for (auto& r : p.routes)
            for (size_t v : r) assigned |= v == p.customer_values[i];
        if (!assigned) out.push_back(i);
    }
    return out;
}
static ClarkeWrightHooks cw_hooks(CwPlan& p, std::function<int64_t(size_t, size_t, size_t)> distance,
                                  std::function<bool(size_t, const std::vector<size_t>&)> feasible, bool shared_class = false) {
    ClarkeWrightHooks h;
    h.entity_count = p.routes.size();
    h.source_values = p.customer_values;
    h.route_len = [&p](size_t e) { return p.routes[e].size(); };
    h.depot = [](size_t) { return (size_t)0; };
    h.metric_class = shared_class ? std::function<size_t(size_t)>([](size_t) { return (size_t)0; })
                                  : std::function<size_t(size_t)>([](size_t e) { return e; });  // unique_metric_class (list_clarke_wright.rs:184-186)
    h.distance = distance;
    h.feasible = feasible;
    h.replace_route = [&p](size_t e, const std::vector<size_t>& r) { p.routes[e] = r; };
    return h;
}
static std::vector<size_t> cw_sorted(std::vector<size_t> v) {
    std::sort(v.begin(), v.end());
    return v;
}
static std::vector<size_t> cw_assigned_sorted(const CwPlan& p) {
    std::vector<size_t> out;
    for (auto& r : p.routes) out.insert(out.end(), r.begin(), r.end());
    return cw_sorted(out);
}
static bool cw_any_sorted(const CwPlan& p, std::vector<size_t> want) {
    for (auto& r : p.routes)
        if (cw_sorted(r) == want) return true;
    return false;
}
static bool cw_any_exact(const CwPlan& p, std::vector<size_t> want) {
    for (auto& r : p.routes)
        if (r == want) return true;
    return false;
}
static void clarke_wright_cases() {
    using V = std::vector<size_t>;
    auto abs_distance = [](size_t, size_t a, size_t b) { return (int64_t)(a > b ? a - b : b - a); };
    auto len3 = [](size_t, const V& r) { return r.size() <= 3; };
    auto single_visit = [](size_t, const V& r) { return r.size() <= 1; };
    CHECK("clarke_wright.sum_two_minus_one", cw_sum_two_minus_one(10, 8, 3) == 15 && cw_sum_two_minus_one(INT64_MAX, INT64_MAX, -1) == INT64_MAX &&
                                                 cw_sum_two_minus_one(INT64_MIN, INT64_MIN, 1) == INT64_MIN);  // distance_arithmetic.rs:24-40
    {  // clarke_wright_hooks_receive_actual_list_values (tests.rs:231-290)
        CwPlan p{{10, 20, 30}, {{}}};
        std::vector<size_t> seen;
        auto h = cw_hooks(
            p, [&](size_t, size_t a, size_t b) { seen.push_back(a), seen.push_back(b); return (int64_t)(a > b ? a - b : b - a); },
            [&](size_t, const V& r) { for (size_t v : r) seen.push_back(v); return r.size() <= 3; });
        clarke_wright(h, cw_unassigned(p));
        bool ok = cw_sorted(p.routes[0]) == V{10, 20, 30}, big = false;
        for (size_t v : seen) {
            big |= v >= 10;
            ok = ok && (v == 0 || v == 10 || v == 20 || v == 30);
        }
        CHECK("clarke_wright.hooks_receive_actual_list_values", ok && big);
    }
    {  // clarke_wright_route_feasible_preserves_capacity_hook_behavior (tests.rs:293-340)
        CwPlan p{{10, 20, 30}, {{}, {}}};
        auto h = cw_hooks(p, abs_distance, [](size_t, const V& r) { size_t s = 0; for (size_t v : r) s += v; return s <= 30; });
        clarke_wright(h, cw_unassigned(p));
        bool ok = cw_assigned_sorted(p) == V{10, 20, 30};
        for (auto& r : p.routes) {
            size_t s = 0;
            for (size_t v : r) s += v;
            ok = ok && s <= 30;
        }
        CHECK("clarke_wright.capacity_hook", ok);
    }
    {  // clarke_wright_extreme_distances_do_not_overflow_savings (tests.rs:343-391)
        CwPlan p{{10, 20, 30}, {{}, {}, {}}};
        auto h = cw_hooks(p, [](size_t, size_t a, size_t b) { return a == b ? (int64_t)0 : INT64_MAX; }, single_visit);
        clarke_wright(h, cw_unassigned(p));
        bool ok = cw_assigned_sorted(p) == V{10, 20, 30};
        for (auto& r : p.routes) ok = ok && r.size() <= 1;
        CHECK("clarke_wright.extreme_distances", ok);
    }
    {  // clarke_wright_respects_fixed_element_owner (tests.rs:394-431)
        CwPlan p{{10, 11}, {{}, {}}};
        auto h = cw_hooks(p, abs_distance, len3);
        h.element_owner = [&p](size_t s) { return (int64_t)(p.customer_values[s] % 2); };
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.fixed_element_owner", p.routes[0] == V{10} && p.routes[1] == V{11});
    }
    {  // clarke_wright_keeps_unrestricted_elements_when_owner_hook_exists (tests.rs:434-482)
        CwPlan p{{10, 11, 12}, {{}, {}, {}}};
        auto h = cw_hooks(p, abs_distance, single_visit);
        h.element_owner = [&p](size_t s) { return p.customer_values[s] == 11 ? (int64_t)1 : (int64_t)-1; };
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.mixed_element_owner", p.routes[1] == V{11} && cw_assigned_sorted(p) == V{10, 11, 12});
    }
    {  // clarke_wright_preserves_preassigned_routes (tests.rs:485-531): value 0 is the depot of the empty slots and is filtered
        CwPlan p{{0, 10, 20, 30}, {{20}, {}, {}}};
        auto h = cw_hooks(p, abs_distance, len3);
        clarke_wright(h, cw_unassigned(p));
        V rest;
        for (size_t e = 1; e < 3; ++e) rest.insert(rest.end(), p.routes[e].begin(), p.routes[e].end());
        CHECK("clarke_wright.preserves_preassigned_routes", p.routes[0] == V{20} && cw_sorted(rest) == V{10, 30});
    }
    {  // clarke_wright_assigns_constructed_routes_to_feasible_owners (tests.rs:534-576)
        CwPlan p{{10, 11, 20, 21}, {{}, {}}};
        auto h = cw_hooks(p, abs_distance, [](size_t e, const V& r) {
            if (r.size() > 2) return false;
            for (size_t v : r)
                if (e == 0 ? v >= 20 : e == 1 ? v < 20 : true) return false;
            return true;
        });
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.feasible_owners", cw_sorted(p.routes[0]) == V{10, 11} && cw_sorted(p.routes[1]) == V{20, 21});
    }
    {  // clarke_wright_uses_owner_depots_for_savings (tests.rs:579-625)
        CwPlan p{{10, 20}, {{}, {}}};
        std::vector<std::array<size_t, 3>> calls;
        auto h = cw_hooks(p, [&](size_t e, size_t a, size_t b) { calls.push_back({e, a, b}); return (int64_t)(a > b ? a - b : b - a); }, single_visit);
        h.depot = [](size_t e) { return 100 + e; };
        clarke_wright(h, cw_unassigned(p));
        bool a = false, b = false;
        for (auto& c : calls) a |= c == std::array<size_t, 3>{0, 100, 10}, b |= c == std::array<size_t, 3>{1, 101, 10};
        CHECK("clarke_wright.owner_depots", a && b);
    }
    {  // clarke_wright_skips_merge_that_breaks_global_owner_matching (tests.rs:628-682)
        CwPlan p{{1, 2, 3, 4}, {{}, {}, {}}};
        auto h = cw_hooks(
            p,
            [](size_t, size_t a, size_t b) -> int64_t {
                if (a == 0 || b == 0) return 100;
                const size_t lo = std::min(a, b), hi = std::max(a, b);
                return lo == 1 && hi == 2 ? 0 : lo == 2 && hi == 3 ? 1 : 100;
            },
            [](size_t e, const V& r) {
                const V s = cw_sorted(r);
                return e == 0 ? (s == V{4} || s == V{1, 2, 3}) : e == 1 ? s == V{3} : e == 2 ? s == V{1, 2} : false;
            });
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.global_owner_matching",
              cw_assigned_sorted(p) == V{1, 2, 3, 4} && cw_any_sorted(p, {1, 2}) && cw_any_exact(p, {3}) && cw_any_exact(p, {4}));
    }
    {  // clarke_wright_completes_unmatched_routes_with_savings_insertion (tests.rs:685-739)
        CwPlan p{{1, 2, 3, 4, 5}, {{}, {}, {}}};
        auto feas = [](size_t e, const V& r) {
            const size_t cap = e == 0 ? 5 : (e == 1 || e == 2) ? 11 : 0;
            size_t s = 0;
            for (size_t v : r) s += (v == 1 || v == 2) ? 6 : (v >= 3 && v <= 5) ? 5 : 100;
            return s <= cap;
        };
        auto h = cw_hooks(
            p,
            [](size_t, size_t a, size_t b) -> int64_t {
                if (a == 0 || b == 0) return 100;
                const size_t lo = std::min(a, b), hi = std::max(a, b);
                return lo == 3 && hi == 4 ? 0 : lo == 1 && hi == 5 ? 1 : 90;
            },
            feas);
        ClarkeWrightStats st;
        clarke_wright(h, cw_unassigned(p), &st);
        bool ok = cw_assigned_sorted(p) == V{1, 2, 3, 4, 5} && st.completed_by_insertion;
        for (size_t e = 0; e < 3; ++e) ok = ok && feas(e, p.routes[e]);
        CHECK("clarke_wright.completion_by_insertion", ok);
    }
    {  // clarke_wright_computes_savings_once_per_metric_class (tests/metric_class.rs:59-104)
        CwPlan p{{1, 2, 3, 4}, {{}, {}, {}, {}, {}, {}}};
        size_t calls = 0;
        auto h = cw_hooks(p, [&](size_t, size_t a, size_t b) { ++calls; return (int64_t)(a > b ? a - b : b - a); }, len3, true);
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.savings_once_per_metric_class", calls == 4 * 3 / 2 * 3);
    }
    {  // clarke_wright_keeps_feasibility_owner_specific_with_shared_metric_class (tests/metric_class.rs:107-150)
        CwPlan p{{1, 2}, {{}, {}}};
        auto h = cw_hooks(p, abs_distance, [](size_t e, const V& r) {
            return e == 0 ? r.size() <= 1 : e == 1 ? (r.size() <= 1 || cw_sorted(r) == V{1, 2}) : false;
        }, true);
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.owner_specific_feasibility_shared_class", p.routes[0].size() <= 1 && cw_sorted(p.routes[1]) == V{1, 2});
    }
    {  // clarke_wright_checks_owner_matching_inside_shared_metric_class (tests/metric_class.rs:153-219)
        CwPlan p{{1, 2, 3, 4}, {{}, {}, {}}};
        auto h = cw_hooks(
            p,
            [](size_t, size_t a, size_t b) -> int64_t {
                if (a == 0 || b == 0) return 100;
                const size_t lo = std::min(a, b), hi = std::max(a, b);
                return lo == 3 && hi == 4 ? 0 : lo == 1 && hi == 2 ? 1 : 100;
            },
            [](size_t e, const V& r) {
                const V s = cw_sorted(r);
                if (s == V{1, 2} || s == V{3, 4}) return e == 0;
                return s.size() == 1 && s[0] >= 1 && s[0] <= 4;
            },
            true);
        clarke_wright(h, cw_unassigned(p));
        CHECK("clarke_wright.owner_matching_inside_shared_class",
              cw_assigned_sorted(p) == V{1, 2, 3, 4} && cw_any_sorted(p, {3, 4}) && !cw_any_sorted(p, {1, 2}));
    }
}
// list_clarke_wright/tests/compiled_parity.rs:468-476 (public / static / dynamic round robin: elements 1..4 over two empty routes ->
// [[1, 3], [2, 4]]), list_construction/round_robin/tests.rs:99-105 (accepted / applied counted), list_placement.rs:54-69 + round_robin/
// kernel.rs:114-122 (a fixed owner does not advance the cursor, an out-of-range owner is skipped)
static void round_robin_cases() {
    auto mk = [](size_t n) {
        ScoreDirector d;
        d.working.classes.resize(1);
        d.working.classes[0].n = n;
        d.working.classes[0].lists.assign(n, {});
        return d;
    };
    using L = std::vector<std::vector<uint32_t>>;
    {
        ScoreDirector d = mk(2);
        SolverStats st;
        construct_list_round_robin(d, 0, {1, 2, 3, 4}, {}, {}, &st);
        CHECK("round_robin.compiled_parity_routes", d.working.classes[0].lists == L{{1, 3}, {2, 4}});
        CHECK("round_robin.counters", st.moves_generated == 4 && st.moves_evaluated == 4 && st.moves_accepted == 4 && st.moves_applied == 4 && st.step_count == 4);
    }
    {
        ScoreDirector d = mk(3);
        construct_list_round_robin(d, 0, {10, 11, 12, 13, 14}, {}, {-1, 2, -1, 7, -1});
        CHECK("round_robin.fixed_owner_does_not_advance_invalid_is_skipped", d.working.classes[0].lists == L{{10}, {12}, {11, 14}});
    }
    {
        ScoreDirector d = mk(2);
        construct_list_round_robin(d, 0, {5, 6, 7, 8}, {3, 1, 3, 0}, {});
        CHECK("round_robin.order_key_then_source_index", d.working.classes[0].lists == L{{8, 5}, {6, 7}});
    }
}
// manager/phase_factory/list_k_opt.rs:325-395 (route-local 2-opt: improves [1,3,2,4] to [1,2,3,4] on the line metric; a feasibility
// hook that rejects the reversal keeps the route; extreme distances neither overflow nor accept a wrapped improvement)
static void list_k_opt_cases() {
    using V = std::vector<size_t>;
    auto run = [](V route, std::function<int64_t(size_t, size_t, size_t)> dist, std::function<bool(size_t, const V&)> feas, size_t k = 2) {
        std::vector<V> routes{route};
        ListKOptHooks h;
        h.entity_count = 1;
        h.route_values = [&](size_t e) { return routes[e]; };
        h.replace_route = [&](size_t e, const V& r) { routes[e] = r; };
        h.depot = [](size_t) { return (size_t)0; };
        h.distance = dist;
        h.feasible = feas;
        list_k_opt(h, k);
        return routes[0];
    };
    auto line = [](size_t, size_t a, size_t b) { return (int64_t)(a > b ? a - b : b - a); };
    CHECK("list_k_opt.improves_route", run({1, 3, 2, 4}, line, nullptr) == V{1, 2, 3, 4});
    CHECK("list_k_opt.feasibility_hook_rejects", run({1, 3, 2, 4}, line, [](size_t, const V& r) { return r.size() > 2 && r[1] == 3 && r[2] == 2; }) == V{1, 3, 2, 4});
    CHECK("list_k_opt.extreme_distances", run({1, 3, 2, 4}, [](size_t, size_t a, size_t b) { return a == b ? (int64_t)0 : INT64_MAX; }, nullptr) == V{1, 3, 2, 4});
    CHECK("list_k_opt.k_other_than_2_is_a_no_op", run({1, 3, 2, 4}, line, nullptr, 3) == V{1, 3, 2, 4});
    CHECK("list_k_opt.sum_two", cw_sum_two(4, 7) == 11 && cw_sum_two(INT64_MAX, 1) == INT64_MAX && cw_sum_two(INT64_MIN, -1) == INT64_MIN);
}
// stream/collector/tests/collector.rs:333-400 (indexed_presence: membership / counts / runs, complement runs, retract and reset)
static void indexed_presence_cases() {
    {
        IndexedPresenceAccumulator a;
        for (int64_t v : {4, 2, 3, 7, 7}) a.accumulate(v);
        Runs r = a.runs();
        bool ok = a.contains(3) && !a.contains(5) && a.count() == 4 && a.item_count() == 5 && a.count_in(2, 5) == 3 && a.any_in(7, 8);
        ok = ok && r.runs.size() == 2 && r.runs[0].start == 2 && r.runs[0].end == 4 && r.runs[1].start == 7 && r.runs[1].item_count == 2;
        CHECK("indexed_presence.active_runs_and_membership", ok);
    }
    {
        IndexedPresenceAccumulator a;
        for (int64_t v : {0, 2, 5}) a.accumulate(v);
        Runs c = a.complement_runs(0, 7);
        bool ok = c.runs.size() == 3 && c.runs[0].start == 1 && c.runs[0].end == 1 && c.runs[1].start == 3 && c.runs[1].end == 4 &&
                  c.runs[2].start == 6 && c.runs[2].end == 6;
        CHECK("indexed_presence.complement_runs", ok && a.complement_runs(5, 5).runs.empty());
    }
    {
        IndexedPresenceAccumulator a;
        a.accumulate(-1), a.accumulate(-1), a.accumulate(0);
        a.retract(-1);
        bool ok = a.contains(-1) && a.item_count() == 2;
        a.retract(-1);
        ok = ok && !a.contains(-1);
        a.reset();
        CHECK("indexed_presence.retract_and_reset", ok && a.is_empty());
    }
}

int main() {
    indexed_presence_cases();
    list_k_opt_cases();
    round_robin_cases();
    clarke_wright_cases();
    complemented_cases();
    runs_cases();
    list_precedence_cases();
    list_precedence_selector_cases();
    forager_cases();
    k_opt_cases();
    simulated_annealing_cases();
    diversified_late_acceptance_cases();
    nearby_scalar_cases();
    list_permute_cases();
    list_reverse_cases();
    list_ruin_cases();
    compound_scalar_cases();
    cheapest_insertion_cases();
    cheapest_precedence_cases();
    regret_insertion_cases();
    gate_cases();
    balance_cases();
    bi_incr_cases();
    cross_bi_cases();
    exists_cases();
    grouped_cases();
    load_balance_cases();
    nary_cases();
    director_case();
    std::printf("%s %d failures\n", failures ? "FAILED" : "PASSED", failures);
    return failures;
}
