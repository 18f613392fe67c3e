// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/Makefile).  CPU restatement of the reference's canonical Clarke-Wright
// savings construction; never linked into or called by the product library.
//
// Reference (paths under crates/solverforge-solver/src/manager/phase_factory/):
//   list_clarke_wright/kernel.rs:59-472          run_clarke_wright_in_phase: available slots, depot filter, singleton routes,
//                                                savings per representative owner slot, sorted merge passes, owner matching,
//                                                completion, commit
//   list_clarke_wright/savings.rs:1-18           SavingsEntry order: saving desc, metric class, left, right
//   list_clarke_wright/route_state.rs:6-183      ConstructedRoute, routes_match_owners_after_merge (+ the per-class shortcut)
//   list_clarke_wright/owner_assignment.rs:7-113 owner slots, representatives (BTreeMap order), feasible owners, the augmenting
//                                                match_route_owners
//   list_clarke_wright/completion.rs:19-224      completion by savings insertion when routes outnumber matched owners
//   list_clarke_wright.rs:123-187                owner_allows / route_owner_allows / insertion_delta
//   distance_arithmetic.rs:1-17                  sum_two_minus_one: exact i128 sum clamped to i64
// Pinned to the known answers of list_clarke_wright/tests.rs and tests/metric_class.rs (oracle/test_golden.cpp,
// test_clarke_wright_*).  The hooks below are the ClarkeWrightAccess protocol (list_clarke_wright.rs:37-50) as closures.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <set>
#include <vector>

namespace sfo {

inline int64_t cw_sum_two_minus_one(int64_t left, int64_t right, int64_t minus) {  // distance_arithmetic.rs:7-16
    __int128 v = (__int128)left + (__int128)right - (__int128)minus;
    if (v > (__int128)INT64_MAX) return INT64_MAX;
    if (v < (__int128)INT64_MIN) return INT64_MIN;
    return (int64_t)v;
}

// ClarkeWrightAccess (list_clarke_wright.rs:37-50).  `source` = the declared elements in source order (their route values);
// an element is named by its source index everywhere below, exactly like the reference.
struct ClarkeWrightHooks {
    size_t entity_count = 0;
    std::vector<size_t> source_values;                                     // route_value(source_index.element(i))
    std::function<size_t(size_t)> route_len;                               // current list length of an owner
    std::function<size_t(size_t)> depot;                                   // savings_depot(owner)
    std::function<size_t(size_t)> metric_class;                            // savings_metric_class(owner)
    std::function<int64_t(size_t, size_t, size_t)> distance;               // savings_distance(owner, from, to)
    std::function<bool(size_t, const std::vector<size_t>&)> feasible;      // savings_feasible(owner, route values)
    std::function<int64_t(size_t)> element_owner;                          // -1 = unrestricted (element_owner -> None), by source index
    std::function<void(size_t, const std::vector<size_t>&)> replace_route; // replace_route(owner, route values)
};

struct ClarkeWrightStats {
    uint64_t savings_pairs = 0, merge_trials = 0, merges = 0, merge_passes = 0, completion_trials = 0;
    bool completed_by_insertion = false, discarded = false;
};

namespace cw_detail {

struct OwnerSlot {
    size_t owner_idx, metric_class;
};
struct Route {  // route_state.rs:6-28
    std::vector<size_t> visits;
    bool scored = false;
    size_t scored_class = 0;
    bool feasible_all_owners = false, feasible_all_class_owners = false;
    bool can_merge_for(size_t c) const { return !scored || scored_class == c; }
};

inline bool owner_allows(const ClarkeWrightHooks& h, size_t entity, size_t element) {  // list_clarke_wright.rs:123-137
    const int64_t o = h.element_owner ? h.element_owner(element) : -1;
    if (o < 0) return true;
    return (size_t)o < h.entity_count && (size_t)o == entity;
}
inline bool route_owner_allows(const ClarkeWrightHooks& h, size_t entity, const std::vector<size_t>& route) {
    for (size_t e : route)
        if (!owner_allows(h, entity, e)) return false;
    return true;
}
inline std::vector<size_t> route_values(const ClarkeWrightHooks& h, const std::vector<size_t>& route) {
    std::vector<size_t> out;
    out.reserve(route.size());
    for (size_t i : route) out.push_back(h.source_values[i]);
    return out;
}

// owner_assignment.rs:52-77
inline std::vector<size_t> feasible_owners(const ClarkeWrightHooks& h, const std::vector<OwnerSlot>& slots, const std::vector<size_t>& route,
                                           bool scored, size_t scored_class) {
    std::vector<size_t> out;
    const std::vector<size_t> values = route_values(h, route);
    for (const OwnerSlot& s : slots) {
        if (scored && s.metric_class != scored_class) continue;
        if (!h.feasible(s.owner_idx, values)) continue;
        if (!route_owner_allows(h, s.owner_idx, route)) continue;
        out.push_back(s.owner_idx);
    }
    return out;
}

// owner_assignment.rs:95-113 (augmenting path; `seen` is shared along one route's search)
inline bool assign_route(size_t route_idx, const std::vector<std::vector<size_t>>& sets, std::map<size_t, size_t>& owner_to_route,
                         std::set<size_t>& seen) {
    for (size_t owner : sets[route_idx]) {
        if (!seen.insert(owner).second) continue;
        auto it = owner_to_route.find(owner);
        bool ok = true;
        if (it != owner_to_route.end()) {
            const size_t existing = it->second;
            ok = assign_route(existing, sets, owner_to_route, seen);
        }
        if (ok) {
            owner_to_route[owner] = route_idx;
            return true;
        }
    }
    return false;
}
// owner_assignment.rs:79-93: -1 = unmatched
inline std::vector<int64_t> match_route_owners(const std::vector<std::vector<size_t>>& sets) {
    std::vector<size_t> order(sets.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (sets[a].size() != sets[b].size()) return sets[a].size() < sets[b].size();
        return a < b;
    });
    std::map<size_t, size_t> owner_to_route;
    for (size_t r : order) {
        std::set<size_t> seen;
        (void)assign_route(r, sets, owner_to_route, seen);
    }
    std::vector<int64_t> out(sets.size(), -1);
    for (auto& kv : owner_to_route) out[kv.second] = (int64_t)kv.first;
    return out;
}

// route_state.rs:122-183
inline bool match_by_metric_class(const std::vector<Route>& routes, size_t merged, size_t removed, size_t cand_class, bool cand_all_class,
                                  const std::vector<OwnerSlot>& slots) {
    std::map<size_t, size_t> route_count, owner_count;
    size_t non_empty = 0;
    for (const OwnerSlot& s : slots) ++owner_count[s.metric_class];
    for (size_t i = 0; i < routes.size(); ++i) {
        const Route& r = routes[i];
        if (i == removed || r.visits.empty()) continue;
        ++non_empty;
        const bool scored = i == merged ? true : r.scored;
        const size_t cls = i == merged ? cand_class : r.scored_class;
        const bool all_class = i == merged ? cand_all_class : r.feasible_all_class_owners;
        if (scored) {
            if (all_class)
                ++route_count[cls];
            else
                return false;
        } else if (!r.feasible_all_owners) {
            return false;
        }
    }
    if (non_empty > slots.size()) return false;
    for (auto& kv : route_count) {
        auto it = owner_count.find(kv.first);
        if (it == owner_count.end() || kv.second > it->second) return false;
    }
    return true;
}

// route_state.rs:57-120
inline bool routes_match_owners_after_merge(const ClarkeWrightHooks& h, const std::vector<Route>& routes, size_t merged, size_t removed,
                                            const std::vector<size_t>& candidate, size_t cand_class, bool cand_all_class,
                                            const std::vector<OwnerSlot>& slots) {
    if (match_by_metric_class(routes, merged, removed, cand_class, cand_all_class, slots)) return true;
    std::vector<std::vector<size_t>> sets;
    for (size_t i = 0; i < routes.size(); ++i) {
        const std::vector<size_t>* rv;
        bool scored;
        size_t cls;
        if (i == merged) {
            rv = &candidate, scored = true, cls = cand_class;
        } else if (i == removed || routes[i].visits.empty()) {
            continue;
        } else {
            rv = &routes[i].visits, scored = routes[i].scored, cls = routes[i].scored_class;
        }
        std::vector<size_t> fo = feasible_owners(h, slots, *rv, scored, cls);
        if (fo.empty()) return false;
        sets.push_back(std::move(fo));
    }
    if (sets.size() > slots.size()) return true;
    for (int64_t o : match_route_owners(sets))
        if (o < 0) return false;
    return true;
}

// list_clarke_wright.rs:153-182
inline int64_t insertion_delta(const ClarkeWrightHooks& h, size_t owner, const std::vector<size_t>& route, size_t insert_idx, size_t element) {
    const size_t depot = h.depot(owner);
    const size_t value = h.source_values[element];
    const size_t prev = insert_idx == 0 ? depot : h.source_values[route[insert_idx - 1]];
    const size_t next = insert_idx < route.size() ? h.source_values[route[insert_idx]] : depot;
    return cw_sum_two_minus_one(h.distance(owner, prev, value), h.distance(owner, value, next), h.distance(owner, prev, next));
}

// completion.rs:19-224: false = None (no completion)
inline bool complete_by_insertion(const ClarkeWrightHooks& h, const std::vector<OwnerSlot>& slots, const std::vector<Route>& routes,
                                  std::vector<std::pair<size_t, std::vector<size_t>>>& out, ClarkeWrightStats* stats) {
    struct Assignment {
        size_t owner_idx;
        std::vector<size_t> route;
    };
    std::vector<Assignment> asg;
    for (const OwnerSlot& s : slots) asg.push_back({s.owner_idx, {}});
    struct Key {
        size_t feasible_owner_count, route_idx, visit_position, element_idx;
        bool operator<(const Key& o) const {
            if (feasible_owner_count != o.feasible_owner_count) return feasible_owner_count < o.feasible_owner_count;
            if (route_idx != o.route_idx) return route_idx < o.route_idx;
            if (visit_position != o.visit_position) return visit_position < o.visit_position;
            return element_idx < o.element_idx;
        }
    };
    std::vector<Key> order;
    for (size_t ri = 0; ri < routes.size(); ++ri) {
        if (routes[ri].visits.empty()) continue;
        for (size_t vp = 0; vp < routes[ri].visits.size(); ++vp) {
            const size_t e = routes[ri].visits[vp];
            const std::vector<size_t> value{h.source_values[e]};
            size_t cnt = 0;
            for (const OwnerSlot& s : slots)
                if (owner_allows(h, s.owner_idx, e) && h.feasible(s.owner_idx, value)) ++cnt;
            if (cnt == 0) return false;
            order.push_back({cnt, ri, vp, e});
        }
    }
    std::sort(order.begin(), order.end());
    for (const Key& k : order) {
        const size_t e = k.element_idx;
        bool have = false;
        int64_t best_delta = 0;
        size_t best_a = 0, best_i = 0;
        for (size_t a = 0; a < asg.size(); ++a) {
            const size_t owner = asg[a].owner_idx;
            if (!owner_allows(h, owner, e)) continue;
            for (size_t ins = 0; ins <= asg[a].route.size(); ++ins) {
                if (stats) ++stats->completion_trials;
                std::vector<size_t> cand = asg[a].route;
                cand.insert(cand.begin() + (std::ptrdiff_t)ins, e);
                if (!h.feasible(owner, route_values(h, cand))) continue;
                if (!route_owner_allows(h, owner, cand)) continue;
                const int64_t delta = insertion_delta(h, owner, asg[a].route, ins, e);
                bool better = !have;
                if (have) {  // (delta, route len, assignment index) strictly less
                    const size_t bl = asg[best_a].route.size(), cl = asg[a].route.size();
                    better = delta < best_delta || (delta == best_delta && (cl < bl || (cl == bl && a < best_a)));
                }
                if (better) have = true, best_delta = delta, best_a = a, best_i = ins;
            }
        }
        if (!have) return false;
        asg[best_a].route.insert(asg[best_a].route.begin() + (std::ptrdiff_t)best_i, e);
    }
    out.clear();
    for (Assignment& a : asg)
        if (!a.route.empty()) out.push_back({a.owner_idx, route_values(h, a.route)});
    return true;
}

}  // namespace cw_detail

// kernel.rs:59-472.  `unassigned` = source indices of the bound unassigned elements, increasing
// (runtime_list_source.rs:185-223).  Returns true when routes were committed through replace_route.
inline bool clarke_wright(const ClarkeWrightHooks& h, const std::vector<size_t>& bound_unassigned, ClarkeWrightStats* stats = nullptr) {
    using namespace cw_detail;
    const size_t n_entities = h.entity_count, n_elements = h.source_values.size();
    if (n_entities == 0 || n_elements == 0) return false;
    std::vector<size_t> available;
    for (size_t e = 0; e < n_entities; ++e)
        if (h.route_len(e) == 0) available.push_back(e);
    std::set<size_t> depot_values;
    for (size_t e : available) depot_values.insert(h.depot(e));
    std::vector<size_t> unassigned;
    for (size_t s : bound_unassigned)
        if (!depot_values.count(h.source_values[s])) unassigned.push_back(s);
    if (unassigned.empty() || available.empty()) return false;

    std::vector<OwnerSlot> slots;
    for (size_t e : available) slots.push_back({e, h.metric_class(e)});
    std::vector<OwnerSlot> reps;  // BTreeMap: first owner of every class, in class order
    {
        std::map<size_t, size_t> first;
        for (const OwnerSlot& s : slots) first.emplace(s.metric_class, s.owner_idx);
        for (auto& kv : first) reps.push_back({kv.second, kv.first});
    }
    const size_t n = unassigned.size();
    std::vector<Route> routes(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t s = unassigned[i];
        const std::vector<size_t> single{h.source_values[s]};
        bool all = true;
        for (const OwnerSlot& sl : slots)
            if (!(h.feasible(sl.owner_idx, single) && owner_allows(h, sl.owner_idx, s))) {
                all = false;
                break;
            }
        routes[i].visits = {s};
        routes[i].feasible_all_owners = all;
    }
    std::vector<int64_t> route_of(n_elements, -1);
    for (size_t i = 0; i < n; ++i) route_of[unassigned[i]] = (int64_t)i;

    struct Entry {
        int64_t saving;
        size_t metric_class, left, right;
    };
    std::vector<Entry> savings;
    savings.reserve(n * (n - 1) / 2 * reps.size());
    for (const OwnerSlot& rep : reps) {
        const size_t owner = rep.owner_idx;
        for (size_t a = 0; a < n; ++a)
            for (size_t b = a + 1; b < n; ++b) {
                const size_t depot = h.depot(owner);
                const size_t lv = h.source_values[unassigned[a]], rv = h.source_values[unassigned[b]];
                const int64_t sv = cw_sum_two_minus_one(h.distance(owner, depot, lv), h.distance(owner, depot, rv), h.distance(owner, lv, rv));
                savings.push_back({sv, rep.metric_class, unassigned[a], unassigned[b]});
                if (stats) ++stats->savings_pairs;
            }
    }
    std::sort(savings.begin(), savings.end(), [](const Entry& l, const Entry& r) {  // savings.rs:9-18 (a total order: no ties survive)
        if (l.saving != r.saving) return l.saving > r.saving;
        if (l.metric_class != r.metric_class) return l.metric_class < r.metric_class;
        if (l.left != r.left) return l.left < r.left;
        return l.right < r.right;
    });

    for (;;) {
        bool merged_in_pass = false;
        if (stats) ++stats->merge_passes;
        for (const Entry& en : savings) {
            if (stats) ++stats->merge_trials;
            const int64_t ri_ = route_of[en.left], rj_ = route_of[en.right];
            if (ri_ < 0 || rj_ < 0) continue;
            const size_t ri = (size_t)ri_, rj = (size_t)rj_;
            if (ri == rj || !routes[ri].can_merge_for(en.metric_class) || !routes[rj].can_merge_for(en.metric_class)) continue;
            const bool i_end = routes[ri].visits.front() == en.left || routes[ri].visits.back() == en.left;
            const bool j_end = routes[rj].visits.front() == en.right || routes[rj].visits.back() == en.right;
            if (!i_end || !j_end) continue;
            std::vector<size_t> test_ri = routes[ri].visits;
            if (test_ri.front() == en.left) std::reverse(test_ri.begin(), test_ri.end());
            std::vector<size_t> test_rj = routes[rj].visits;
            if (test_rj.back() == en.right) std::reverse(test_rj.begin(), test_rj.end());
            std::vector<size_t> cand = test_ri;
            cand.insert(cand.end(), test_rj.begin(), test_rj.end());
            const std::vector<size_t> fo = feasible_owners(h, slots, cand, true, en.metric_class);
            if (fo.empty()) continue;
            size_t class_owners = 0;
            for (const OwnerSlot& s : slots)
                if (s.metric_class == en.metric_class) ++class_owners;
            const bool all_class = fo.size() == class_owners;
            if (!routes_match_owners_after_merge(h, routes, ri, rj, cand, en.metric_class, all_class, slots)) continue;
            routes[ri].visits = test_ri;
            routes[ri].scored = true;
            routes[ri].scored_class = en.metric_class;
            routes[ri].feasible_all_owners = false;
            routes[ri].feasible_all_class_owners = all_class;
            routes[rj].visits.clear();
            routes[rj].scored = false;
            routes[rj].feasible_all_owners = false;
            routes[rj].feasible_all_class_owners = false;
            for (size_t s : test_rj) route_of[s] = (int64_t)ri;
            routes[ri].visits.insert(routes[ri].visits.end(), test_rj.begin(), test_rj.end());
            merged_in_pass = true;
            if (stats) ++stats->merges;
        }
        if (!merged_in_pass) break;
    }

    std::vector<Route> non_empty;
    for (Route& r : routes)
        if (!r.visits.empty()) non_empty.push_back(r);
    std::vector<Route> assignable;
    std::vector<std::vector<size_t>> sets;
    size_t ineligible = 0;
    for (const Route& r : non_empty) {
        std::vector<size_t> fo = feasible_owners(h, slots, r.visits, r.scored, r.scored_class);
        if (fo.empty()) {
            ++ineligible;
            continue;
        }
        assignable.push_back(r);
        sets.push_back(std::move(fo));
    }
    const std::vector<int64_t> route_to_owner = match_route_owners(sets);
    size_t matched = 0;
    for (int64_t o : route_to_owner)
        if (o >= 0) ++matched;
    std::vector<std::pair<size_t, std::vector<size_t>>> completed;
    bool have_completion = false;
    if (matched < assignable.size() && ineligible == 0) {
        have_completion = complete_by_insertion(h, slots, non_empty, completed, stats);
        if (stats) stats->completed_by_insertion = have_completion;
    }
    if (matched < assignable.size() && !have_completion) {
        if (stats) stats->discarded = true;
        return false;
    }
    if (have_completion) {
        for (auto& kv : completed) h.replace_route(kv.first, kv.second);
        return true;
    }
    if (matched > 0) {
        for (size_t i = 0; i < assignable.size(); ++i) {
            if (route_to_owner[i] < 0) continue;
            h.replace_route((size_t)route_to_owner[i], route_values(h, assignable[i].visits));
        }
        return true;
    }
    if (stats) stats->discarded = true;
    return false;
}

// ---- route-local 2-opt polishing (manager/phase_factory/list_k_opt/kernel.rs:57-220; the step the default construction runs after
// Clarke-Wright).  Per owner with >= 4 visits: sweeps over (i, j), i < j, reversing route[i..=j] in place whenever
// d(a, c) + d(b, e) < d(a, b) + d(c, e) and the reversed route is feasible -- with a = the element before i (or the depot) and
// b = route[i] READ ONCE PER i (they are not refreshed after a reversal inside the j loop: restated as written), c = route[j],
// e = the element after j (or the depot) read from the current route -- until a sweep improves nothing; a changed route is committed
// as one step.  k != 2 is a scored no-op.  distance_arithmetic.rs:1-5 sum_two: exact sum clamped to i64.
inline int64_t cw_sum_two(int64_t l, int64_t r) {
    __int128 v = (__int128)l + (__int128)r;
    if (v > (__int128)INT64_MAX) return INT64_MAX;
    if (v < (__int128)INT64_MIN) return INT64_MIN;
    return (int64_t)v;
}
struct ListKOptHooks {
    size_t entity_count = 0;
    std::function<std::vector<size_t>(size_t)> route_values;
    std::function<void(size_t, const std::vector<size_t>&)> replace_route;
    std::function<size_t(size_t)> depot;
    std::function<int64_t(size_t, size_t, size_t)> distance;
    std::function<bool(size_t, const std::vector<size_t>&)> feasible;  // empty = always
};
struct ListKOptStats {
    uint64_t candidates = 0, accepted = 0, applied = 0, steps = 0;
};
// max_sweeps (0 = unlimited) stands in for the construction's termination policy (kernel.rs:117-121): on an asymmetric metric the
// 2-opt delta ignores the reversed inner legs and the sweeps need not converge.
inline void list_k_opt(const ListKOptHooks& h, size_t k, ListKOptStats* stats = nullptr, size_t max_sweeps = 0) {
    if (k != 2) return;
    for (size_t ent = 0; ent < h.entity_count; ++ent) {
        const size_t depot = h.depot(ent);
        std::vector<size_t> route = h.route_values(ent);
        const size_t n = route.size();
        if (n < 4) continue;
        bool changed = false;
        uint64_t pending = 0;
        size_t sweeps = 0;
        for (;;) {
            bool improved = false;
            for (size_t i = 0; i + 1 < n; ++i) {
                const size_t a = i == 0 ? depot : route[i - 1];
                const size_t b = route[i];
                for (size_t j = i + 1; j < n; ++j) {
                    const size_t c = route[j];
                    const size_t e = j + 1 < n ? route[j + 1] : depot;
                    const int64_t proposed = cw_sum_two(h.distance(ent, a, c), h.distance(ent, b, e));
                    const int64_t current = cw_sum_two(h.distance(ent, a, b), h.distance(ent, c, e));
                    if (stats) ++stats->candidates;
                    if (proposed < current) {
                        std::reverse(route.begin() + (std::ptrdiff_t)i, route.begin() + (std::ptrdiff_t)j + 1);
                        if (h.feasible && !h.feasible(ent, route)) {
                            std::reverse(route.begin() + (std::ptrdiff_t)i, route.begin() + (std::ptrdiff_t)j + 1);
                            continue;
                        }
                        if (stats) ++stats->accepted;
                        ++pending;
                        improved = changed = true;
                    }
                }
            }
            if (!improved || (max_sweeps && ++sweeps >= max_sweeps)) break;
        }
        if (changed) {
            h.replace_route(ent, route);
            if (stats) stats->applied += pending, ++stats->steps;
        }
    }
}

}  // namespace sfo
