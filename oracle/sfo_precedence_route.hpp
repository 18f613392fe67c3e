// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// CPU restatement of the precedence route graph behind the critical-path list neighbourhood and the precedence-aware
// ruin recreate:
//   heuristic/selector/precedence_route.rs:1-561 (build_precedence_route_graph, the intra-list cycle tests, the
//   multi-swap / insertion cycle tests over the whole graph)
// (paths under crates/solverforge-solver/src/).  Nodes are list elements, identified with their index
// (index_to_element = identity, as in the reference's tests/list_precedence.rs:80-86); an element value that is not a node
// index breaks the route chain, as `node_index` returning None does.
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace sfo {

// The slot's precedence hooks as data: node count, durations (node_duration) and the fixed successor lists in hook order
// (duplicates and foreign values are kept: the graph build filters them the way the reference does).
struct PrecedenceHooks {
    size_t node_count = 0;
    std::vector<int64_t> durations;
    std::vector<std::vector<size_t>> successors;
};

struct PrecedenceRouteGraph {
    using Edge = std::pair<size_t, size_t>;
    std::vector<std::vector<size_t>> fixed_successors, fixed_predecessors, successors, predecessors, route_nodes;

    static bool has(const std::vector<size_t>& v, size_t x) { return std::find(v.begin(), v.end(), x) != v.end(); }
    static bool has_edge(const std::vector<Edge>& v, Edge e) { return std::find(v.begin(), v.end(), e) != v.end(); }

    void push_edge(size_t from, size_t to) {  // (:306-311)
        if (!has(successors[from], to)) {
            successors[from].push_back(to);
            predecessors[to].push_back(from);
        }
    }

    // build_precedence_route_graph (:53-112): fixed edges first, then one chain per owner list
    static PrecedenceRouteGraph build(const PrecedenceHooks& h, const std::vector<std::vector<uint32_t>>& lists) {
        const size_t n = h.node_count;
        PrecedenceRouteGraph g;
        g.fixed_successors.assign(n, {});
        g.fixed_predecessors.assign(n, {});
        g.successors.assign(n, {});
        g.predecessors.assign(n, {});
        for (size_t from = 0; from < n; ++from)
            for (size_t to : h.successors[from]) {
                if (to >= n) continue;
                g.push_edge(from, to);
                if (!has(g.fixed_successors[from], to)) {
                    g.fixed_successors[from].push_back(to);
                    g.fixed_predecessors[to].push_back(from);
                }
            }
        for (const auto& list : lists) {
            std::vector<size_t> nodes;
            bool have_previous = false;
            size_t previous = 0;
            for (uint32_t element : list) {
                if ((size_t)element >= n) {
                    have_previous = false;
                    continue;
                }
                if (have_previous) g.push_edge(previous, (size_t)element);
                nodes.push_back((size_t)element);
                previous = (size_t)element;
                have_previous = true;
            }
            g.route_nodes.push_back(std::move(nodes));
        }
        return g;
    }

    const std::vector<size_t>* route(size_t entity) const { return entity < route_nodes.size() ? &route_nodes[entity] : nullptr; }

    bool node_route_position(size_t node, size_t& entity, size_t& position) const {  // (:158-169)
        for (size_t e = 0; e < route_nodes.size(); ++e)
            for (size_t p = 0; p < route_nodes[e].size(); ++p)
                if (route_nodes[e][p] == node) {
                    entity = e;
                    position = p;
                    return true;
                }
        return false;
    }

    // ---- the routes a move leaves behind (:320-392) ----
    static std::vector<size_t> after_list_change(const std::vector<size_t>& r, size_t source, size_t dest) {
        std::vector<size_t> a = r;
        size_t node = a[source];
        a.erase(a.begin() + (ptrdiff_t)source);
        a.insert(a.begin() + (ptrdiff_t)(dest > source ? dest - 1 : dest), node);
        return a;
    }
    static std::vector<size_t> after_sublist_change(const std::vector<size_t>& r, size_t s, size_t e, size_t dest) {
        std::vector<size_t> a = r;
        std::vector<size_t> moved(a.begin() + (ptrdiff_t)s, a.begin() + (ptrdiff_t)e);
        a.erase(a.begin() + (ptrdiff_t)s, a.begin() + (ptrdiff_t)e);
        a.insert(a.begin() + (ptrdiff_t)dest, moved.begin(), moved.end());
        return a;
    }
    static bool after_sublist_swap(const std::vector<size_t>& r, size_t fs, size_t fe, size_t ss, size_t se, std::vector<size_t>& a) {
        size_t ls = fs <= ss ? fs : ss, le = fs <= ss ? fe : se, rs = fs <= ss ? ss : fs, re = fs <= ss ? se : fe;
        if (le > rs || re > r.size()) return false;
        a.assign(r.begin(), r.begin() + (ptrdiff_t)ls);
        a.insert(a.end(), r.begin() + (ptrdiff_t)rs, r.begin() + (ptrdiff_t)re);
        a.insert(a.end(), r.begin() + (ptrdiff_t)le, r.begin() + (ptrdiff_t)rs);
        a.insert(a.end(), r.begin() + (ptrdiff_t)ls, r.begin() + (ptrdiff_t)le);
        a.insert(a.end(), r.begin() + (ptrdiff_t)re, r.end());
        return true;
    }
    static std::vector<size_t> after_permutation(const std::vector<size_t>& r, size_t start, const std::vector<size_t>& perm) {
        std::vector<size_t> a = r;
        for (size_t k = 0; k < perm.size(); ++k) a[start + k] = r[start + perm[k]];
        return a;
    }

    static std::vector<Edge> route_edges(const std::vector<size_t>& r) {
        std::vector<Edge> e;
        for (size_t i = 0; i + 1 < r.size(); ++i) e.push_back({r[i], r[i + 1]});
        return e;
    }
    static void changed_route_edges(const std::vector<size_t>& before, const std::vector<size_t>& after, std::vector<Edge>& removed,
                                    std::vector<Edge>& added) {  // (:394-412): appended in route order
        std::vector<Edge> be = route_edges(before), ae = route_edges(after);
        for (Edge e : be)
            if (!has_edge(ae, e)) removed.push_back(e);
        for (Edge e : ae)
            if (!has_edge(be, e)) added.push_back(e);
    }

    bool fixed(size_t from, size_t to) const { return from < fixed_successors.size() && has(fixed_successors[from], to); }

    // reaches_with_route_delta (:519-556): `source` reaches `target` over the graph minus the removed (non-fixed) edges plus
    // the active added ones
    bool reaches(size_t source, size_t target, const std::vector<Edge>& removed, const std::vector<Edge>& added) const {
        const size_t n = successors.size();
        if (source >= n || target >= n) return false;
        std::vector<char> visited(n, 0);
        std::vector<size_t> stack{source};
        while (!stack.empty()) {
            size_t node = stack.back();
            stack.pop_back();
            if (node == target) return true;
            if (visited[node]) continue;
            visited[node] = 1;
            for (size_t s : successors[node]) {
                if (has_edge(removed, {node, s}) && !fixed(node, s)) continue;
                if (s < n && !visited[s]) stack.push_back(s);
            }
            for (Edge e : added)
                if (e.first == node && e.second < n && !visited[e.second]) stack.push_back(e.second);
        }
        return false;
    }
    bool edge_active(size_t from, size_t to, const std::vector<Edge>& removed, const std::vector<Edge>& added) const {  // (:503-516)
        if (has_edge(added, {from, to})) return true;
        return from < successors.size() && has(successors[from], to) && (!has_edge(removed, {from, to}) || fixed(from, to));
    }
    // added_edges_introduce_cycle (:419-451): the added edges join one at a time; the first that closes a cycle decides
    bool added_edges_introduce_cycle(const std::vector<Edge>& removed, const std::vector<Edge>& added) const {
        std::vector<Edge> active;
        for (Edge e : added) {
            if (edge_active(e.first, e.second, removed, active)) continue;
            if (reaches(e.second, e.first, removed, active)) return true;
            active.push_back(e);
        }
        return false;
    }
    bool route_move_introduces_cycle(const std::vector<size_t>& before, const std::vector<size_t>& after) const {  // (:313-317)
        std::vector<Edge> removed, added;
        changed_route_edges(before, after, removed, added);
        return added_edges_introduce_cycle(removed, added);
    }

    // route_delta_has_cycle (:453-500): one Kahn pass over the whole graph after the delta
    bool route_delta_has_cycle(const std::vector<Edge>& removed, const std::vector<Edge>& added) const {
        const size_t n = successors.size();
        std::vector<std::vector<size_t>> adjacency(n);
        std::vector<size_t> indegree(n, 0);
        for (size_t from = 0; from < n; ++from)
            for (size_t to : successors[from]) {
                if (to >= n) continue;
                if (has_edge(removed, {from, to}) && !fixed(from, to)) continue;
                if (!has(adjacency[from], to)) {
                    adjacency[from].push_back(to);
                    ++indegree[to];
                }
            }
        for (Edge e : added) {
            if (e.first >= n || e.second >= n) continue;
            if (!has(adjacency[e.first], e.second)) {
                adjacency[e.first].push_back(e.second);
                ++indegree[e.second];
            }
        }
        std::vector<size_t> ready;
        for (size_t node = 0; node < n; ++node)
            if (indegree[node] == 0) ready.push_back(node);
        size_t visited = 0;
        while (!ready.empty()) {
            size_t node = ready.back();
            ready.pop_back();
            ++visited;
            for (size_t s : adjacency[node])
                if (--indegree[s] == 0) ready.push_back(s);
        }
        return visited != n;
    }

    // ---- the public tests (:171-304) ----
    bool intra_list_change_introduces_cycle(size_t entity, size_t source, size_t dest) const {
        const auto* r = route(entity);
        return r && route_move_introduces_cycle(*r, after_list_change(*r, source, dest));
    }
    bool intra_list_swap_introduces_cycle(size_t entity, size_t first, size_t second) const {
        const auto* r = route(entity);
        if (!r) return false;
        std::vector<size_t> a = *r;
        std::swap(a[first], a[second]);
        return route_move_introduces_cycle(*r, a);
    }
    bool intra_list_reverse_introduces_cycle(size_t entity, size_t start, size_t end) const {
        const auto* r = route(entity);
        if (!r) return false;
        std::vector<size_t> a = *r;
        std::reverse(a.begin() + (ptrdiff_t)start, a.begin() + (ptrdiff_t)end);
        return route_move_introduces_cycle(*r, a);
    }
    bool intra_sublist_change_introduces_cycle(size_t entity, size_t s, size_t e, size_t dest) const {
        const auto* r = route(entity);
        return r && route_move_introduces_cycle(*r, after_sublist_change(*r, s, e, dest));
    }
    bool intra_sublist_swap_introduces_cycle(size_t entity, size_t fs, size_t fe, size_t ss, size_t se) const {
        const auto* r = route(entity);
        std::vector<size_t> a;
        if (!r || !after_sublist_swap(*r, fs, fe, ss, se, a)) return false;
        return route_move_introduces_cycle(*r, a);
    }
    bool intra_list_permutation_introduces_cycle(size_t entity, size_t start, const std::vector<size_t>& perm) const {
        const auto* r = route(entity);
        return r && route_move_introduces_cycle(*r, after_permutation(*r, start, perm));
    }
    struct SwapCoord {
        size_t entity, first, second;
    };
    bool multi_intra_list_swaps_introduce_cycle(const std::vector<SwapCoord>& swaps) const {  // (:257-280)
        std::vector<Edge> removed, added;
        for (const SwapCoord& s : swaps) {
            const auto* r = route(s.entity);
            if (!r || s.first >= r->size() || s.second >= r->size()) continue;
            std::vector<size_t> a = *r;
            std::swap(a[s.first], a[s.second]);
            changed_route_edges(*r, a, removed, added);
        }
        return route_delta_has_cycle(removed, added);
    }
    // insertion_introduces_cycle (:282-304): `element` between `previous` and `next` (SIZE_MAX = none)
    bool insertion_introduces_cycle(size_t previous, size_t element, size_t next) const {
        std::vector<Edge> removed, added;
        if (previous != SIZE_MAX) {
            added.push_back({previous, element});
            if (next != SIZE_MAX) removed.push_back({previous, next});
        }
        if (next != SIZE_MAX) added.push_back({element, next});
        return route_delta_has_cycle(removed, added);
    }
};

}  // namespace sfo
