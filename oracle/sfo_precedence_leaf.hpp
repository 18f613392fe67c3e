// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// CPU restatement of the critical-path precedence neighbourhood (ListPrecedenceMoveSelector and the compiled runtime leaf
// RuntimeListNeighborhoodSpec::Precedence share it):
//   heuristic/selector/list_precedence.rs:1-256                      the public selector (open_cursor_with_context, size)
//   heuristic/selector/list_kernel/precedence/analysis.rs:1-192      earliest / latest starts, critical arcs -> critical blocks
//   heuristic/selector/list_kernel/precedence/coordinates.rs:1-492   per-block move families, tiered order, cycle pruning
//   heuristic/selector/list_kernel/precedence/support.rs:1-195       cross-block support swaps and multi-ruin coordinates
//   heuristic/selector/list_kernel/precedence/cursor.rs:1-290        the streamed cursor (multi-swaps, multi-ruins, blocks)
//   heuristic/selector/list_kernel/precedence/emission.rs:1-295      coordinates -> moves (ruins carry the precedence hooks,
//                                                                    multi-swaps require a score improvement)
//   runtime/compiler/executor/list_leaf/cursor/probe.rs:259-293      the runtime leaf analyses every entity in index order
// (paths under crates/solverforge-solver/src/).  Pinned to heuristic/selector/tests/list_precedence.rs in test_golden.cpp.
#pragma once
#include <deque>

#include "sfo_precedence_route.hpp"

namespace sfo {

constexpr size_t CRITICAL_PERMUTE_MAX_WINDOW_SIZE = 5, CRITICAL_RUIN_MAX_SIZE = 5, CRITICAL_SUBLIST_MAX_SIZE = 3;  // coordinates.rs:11-13

struct CriticalBlock {  // route positions start..=end of one entity whose consecutive arcs are all critical (coordinates.rs:15-89)
    size_t entity, start, end, route_len;
    size_t len() const { return end - start + 1; }
    size_t change_move_count() const { return len() * (route_len ? route_len - 1 : 0); }
    size_t adjacent_change_move_count() const { return len() ? len() - 1 : 0; }
    static bool valid_non_adjacent_dest(size_t source, size_t source_offset, size_t destination, size_t block_len) {  // (:394-403)
        return destination != source && destination != source + 1 && !(source_offset + 1 < block_len && destination == source + 2);
    }
    std::vector<size_t> boundary_offsets() const {  // (:342-352)
        std::vector<size_t> o{0};
        if (len() - 1 != 0) o.push_back(len() - 1);
        return o;
    }
    size_t boundary_change_move_count() const {  // (:328-340)
        size_t count = 0;
        for (size_t so : boundary_offsets())
            for (size_t dest = 0; dest <= route_len; ++dest) count += valid_non_adjacent_dest(start + so, so, dest, len());
        return count;
    }
    size_t swap_move_count() const { return len() * (len() - 1) / 2; }
    size_t reverse_move_count() const { return len() * (len() - 1) / 2; }
    size_t adjacent_sublist_swap_move_count() const {  // (:405-427)
        size_t bl = len();
        if (bl < 3) return 0;
        size_t max_size = std::min(CRITICAL_SUBLIST_MAX_SIZE, bl), count = 0;
        for (size_t s = 0; s < bl; ++s)
            for (size_t fs = 1; fs <= max_size; ++fs) {
                size_t second_start = s + fs;
                if (second_start >= bl) break;
                for (size_t ss = 1; ss <= max_size; ++ss)
                    if (fs != 1 || ss != 1) count += second_start + ss <= bl;
            }
        return count;
    }
    size_t ruin_move_count() const { return len() < 2 ? 0 : len() - std::min(len(), CRITICAL_RUIN_MAX_SIZE) + 1; }
    size_t sublist_change_move_count() const {  // (:444-456)
        size_t bl = len();
        if (bl < 2 || route_len < 2) return 0;
        size_t max_size = std::min(std::min(CRITICAL_SUBLIST_MAX_SIZE, bl), route_len), count = 0;
        for (size_t size = 2; size <= max_size; ++size) count += (bl - size + 1) * (route_len - size);
        return count;
    }
    size_t permute_move_count() const {  // (:429-442)
        size_t bl = len();
        if (bl < 2) return 0;
        size_t max_window = std::min(std::min(CRITICAL_PERMUTE_MAX_WINDOW_SIZE, MAX_LIST_PERMUTE_WINDOW_SIZE), bl), count = 0;
        for (size_t s = 0; s < bl; ++s)
            for (size_t size = 2; size <= std::min(max_window, bl - s); ++size) count += permute_factorial(size) - 1;
        return count;
    }
    size_t move_count() const {
        return change_move_count() + swap_move_count() + reverse_move_count() + adjacent_sublist_swap_move_count() + ruin_move_count() +
               sublist_change_move_count() + permute_move_count();
    }
};

struct CriticalAnalysis {
    std::vector<CriticalBlock> blocks;
    PrecedenceRouteGraph graph;
};

// critical_analysis_from_graph (analysis.rs:56-112) over graph_summary (:119-168): nothing when the current graph is cyclic
inline CriticalAnalysis critical_analysis(const PrecedenceHooks& h, const std::vector<std::vector<uint32_t>>& lists,
                                          const std::vector<size_t>& selected_entities) {
    CriticalAnalysis out;
    out.graph = PrecedenceRouteGraph::build(h, lists);
    const size_t n = h.node_count;
    if (n == 0) return out;
    const auto& succ = out.graph.successors;
    const auto& pred = out.graph.predecessors;
    auto sat_add = [](int64_t a, int64_t b) {
        int64_t r;
        return __builtin_add_overflow(a, b, &r) ? (b > 0 ? INT64_MAX : INT64_MIN) : r;
    };
    auto sat_sub = [](int64_t a, int64_t b) {
        int64_t r;
        return __builtin_sub_overflow(a, b, &r) ? (b > 0 ? INT64_MIN : INT64_MAX) : r;
    };
    std::vector<size_t> indegree(n);
    std::vector<int64_t> earliest(n, 0);
    std::deque<size_t> ready;
    for (size_t v = 0; v < n; ++v) {
        indegree[v] = pred[v].size();
        if (indegree[v] == 0) ready.push_back(v);
    }
    std::vector<size_t> topo;
    while (!ready.empty()) {
        size_t v = ready.front();
        ready.pop_front();
        topo.push_back(v);
        int64_t finish = sat_add(earliest[v], h.durations[v]);
        for (size_t s : succ[v]) {
            earliest[s] = std::max(earliest[s], finish);
            if (--indegree[s] == 0) ready.push_back(s);
        }
    }
    if (topo.size() != n) return out;
    int64_t makespan = 0;
    for (size_t v : topo) makespan = std::max(makespan, sat_add(earliest[v], h.durations[v]));
    std::vector<int64_t> latest(n, INT64_MAX);
    for (size_t i = n; i-- > 0;) {
        size_t v = topo[i];
        if (succ[v].empty())
            latest[v] = sat_sub(makespan, h.durations[v]);
        else {
            int64_t best = INT64_MAX;
            for (size_t s : succ[v]) best = std::min(best, sat_sub(latest[s], h.durations[v]));
            latest[v] = best;
        }
    }
    auto critical_node = [&](size_t v) { return earliest[v] == latest[v]; };
    auto critical_arc = [&](size_t from, size_t to) {  // (:170-182)
        return PrecedenceRouteGraph::has(succ[from], to) && critical_node(from) && critical_node(to) &&
               sat_add(earliest[from], h.durations[from]) == earliest[to];
    };
    for (size_t entity : selected_entities) {
        const auto* nodes = out.graph.route(entity);
        if (!nodes) continue;
        size_t position = 0;
        while (position < nodes->size()) {
            bool starts_arc = position + 1 < nodes->size() && critical_arc((*nodes)[position], (*nodes)[position + 1]);
            if (!starts_arc) {
                if (critical_node((*nodes)[position])) out.blocks.push_back({entity, position, position, nodes->size()});
                ++position;
                continue;
            }
            size_t start = position;
            ++position;
            while (position + 1 < nodes->size() && critical_arc((*nodes)[position], (*nodes)[position + 1])) ++position;
            out.blocks.push_back({entity, start, position, nodes->size()});
            ++position;
        }
    }
    return out;
}

// ---- coordinates of one block's move families (coordinates.rs:91-246) ----
struct PrecCoord {
    size_t a = 0, b = 0, c = 0, d = 0;
};
inline bool prec_boundary_change(const CriticalBlock& bl, size_t offset, PrecCoord& out) {  // (:354-371)
    for (size_t so : bl.boundary_offsets()) {
        size_t source = bl.start + so;
        for (size_t dest = 0; dest <= bl.route_len; ++dest) {
            if (!CriticalBlock::valid_non_adjacent_dest(source, so, dest, bl.len())) continue;
            if (offset == 0) {
                out = {source, dest, 0, 0};
                return true;
            }
            --offset;
        }
    }
    return false;
}
inline bool prec_interior_change(const CriticalBlock& bl, size_t offset, PrecCoord& out) {  // (:373-392)
    for (size_t so = 0; so < bl.len(); ++so) {
        if (so == 0 || so + 1 == bl.len()) continue;
        size_t source = bl.start + so;
        for (size_t dest = 0; dest <= bl.route_len; ++dest) {
            if (!CriticalBlock::valid_non_adjacent_dest(source, so, dest, bl.len())) continue;
            if (offset == 0) {
                out = {source, dest, 0, 0};
                return true;
            }
            --offset;
        }
    }
    return false;
}
inline PrecCoord prec_non_adjacent_change(const CriticalBlock& bl, size_t offset) {  // (:132-143)
    PrecCoord c;
    size_t boundary = bl.boundary_change_move_count();
    if (offset < boundary) {
        prec_boundary_change(bl, offset, c);
        return c;
    }
    prec_interior_change(bl, offset - boundary, c);
    return c;
}
inline PrecCoord prec_critical_pair(const CriticalBlock& bl, size_t offset, size_t end_extra) {  // critical_swap / critical_reverse (:145-169)
    for (size_t f = 0; f < bl.len(); ++f)
        for (size_t s = f + 1; s < bl.len(); ++s) {
            if (offset == 0) return {bl.start + f, bl.start + s + end_extra, 0, 0};
            --offset;
        }
    return {};
}
inline PrecCoord prec_adjacent_sublist_swap(const CriticalBlock& bl, size_t offset) {  // (:171-204)
    size_t max_size = std::min(CRITICAL_SUBLIST_MAX_SIZE, bl.len());
    for (size_t s = 0; s < bl.len(); ++s)
        for (size_t fs = 1; fs <= max_size; ++fs) {
            size_t second_start = s + fs;
            if (second_start >= bl.len()) break;
            for (size_t ss = 1; ss <= max_size; ++ss) {
                if (fs == 1 && ss == 1) continue;
                size_t second_end = second_start + ss;
                if (second_end > bl.len()) continue;
                if (offset == 0) return {bl.start + s, bl.start + second_start, bl.start + second_start, bl.start + second_end};
                --offset;
            }
        }
    return {};
}
inline PrecCoord prec_sublist_change(const CriticalBlock& bl, size_t offset) {  // critical_sublist_change (:206-228): (source_start, size, destination)
    size_t max_size = std::min(std::min(CRITICAL_SUBLIST_MAX_SIZE, bl.len()), bl.route_len);
    for (size_t size = 2; size <= max_size; ++size)
        for (size_t ss = 0; ss + size <= bl.len(); ++ss)
            for (size_t dest = 0; dest + size <= bl.route_len; ++dest) {
                if (dest == bl.start + ss) continue;
                if (offset == 0) return {ss, size, dest, 0};
                --offset;
            }
    return {};
}
inline PrecCoord prec_permutation(size_t block_len, size_t offset) {  // critical_permutation (:230-252): (start, size, rank)
    size_t max_window = std::min(std::min(CRITICAL_PERMUTE_MAX_WINDOW_SIZE, MAX_LIST_PERMUTE_WINDOW_SIZE), block_len);
    for (size_t s = 0; s < block_len; ++s)
        for (size_t size = 2; size <= std::min(max_window, block_len - s); ++size) {
            size_t count = permute_factorial(size) - 1;
            if (offset < count) return {s, size, offset + 1, 0};
            offset -= count;
        }
    return {};
}

// One block-local move index -> the move (cursor.rs:83-179) or its cycle test (coordinates.rs:265-326)
struct PrecDecoded {
    enum Family { Change, Swap, Reverse, SublistSwap, Ruin, SublistChange, Permute } family = Change;
    PrecCoord c;
};
inline PrecDecoded prec_decode(const CriticalBlock& bl, size_t move_index) {
    PrecDecoded o;
    size_t adjacent = bl.adjacent_change_move_count(), change = bl.change_move_count();
    if (move_index < adjacent) {
        o.c = {bl.start + move_index, bl.start + move_index + 2, 0, 0};
        return o;
    }
    if (move_index < change) {
        o.c = prec_non_adjacent_change(bl, move_index - adjacent);
        return o;
    }
    size_t base = change;
    if (move_index < base + bl.swap_move_count()) {
        o.family = PrecDecoded::Swap;
        o.c = prec_critical_pair(bl, move_index - base, 0);
        return o;
    }
    base += bl.swap_move_count();
    if (move_index < base + bl.reverse_move_count()) {
        o.family = PrecDecoded::Reverse;
        o.c = prec_critical_pair(bl, move_index - base, 1);
        return o;
    }
    base += bl.reverse_move_count();
    if (move_index < base + bl.adjacent_sublist_swap_move_count()) {
        o.family = PrecDecoded::SublistSwap;
        o.c = prec_adjacent_sublist_swap(bl, move_index - base);
        return o;
    }
    base += bl.adjacent_sublist_swap_move_count();
    if (move_index < base + bl.ruin_move_count()) {
        o.family = PrecDecoded::Ruin;
        o.c = {bl.start + (move_index - base), std::min(bl.len(), CRITICAL_RUIN_MAX_SIZE), 0, 0};  // critical_ruin_indices (:190-204): first index, count
        return o;
    }
    base += bl.ruin_move_count();
    if (move_index < base + bl.sublist_change_move_count()) {
        o.family = PrecDecoded::SublistChange;
        o.c = prec_sublist_change(bl, move_index - base);
        return o;
    }
    base += bl.sublist_change_move_count();
    o.family = PrecDecoded::Permute;
    o.c = prec_permutation(bl.len(), move_index - base);
    return o;
}
inline bool prec_move_introduces_route_cycle(const CriticalBlock& bl, size_t move_index, const PrecedenceRouteGraph& g) {
    const auto* route = g.route(bl.entity);
    if (!route || route->size() != bl.route_len) return false;
    PrecDecoded m = prec_decode(bl, move_index);
    switch (m.family) {
        case PrecDecoded::Change:
            return g.intra_list_change_introduces_cycle(bl.entity, m.c.a, m.c.b);
        case PrecDecoded::Swap:
            return g.intra_list_swap_introduces_cycle(bl.entity, m.c.a, m.c.b);
        case PrecDecoded::Reverse:
            return g.intra_list_reverse_introduces_cycle(bl.entity, m.c.a, m.c.b);
        case PrecDecoded::SublistSwap:
            return g.intra_sublist_swap_introduces_cycle(bl.entity, m.c.a, m.c.b, m.c.c, m.c.d);
        case PrecDecoded::Ruin:
            return false;
        case PrecDecoded::SublistChange:
            return g.intra_sublist_change_introduces_cycle(bl.entity, bl.start + m.c.a, bl.start + m.c.a + m.c.b, m.c.c);
        case PrecDecoded::Permute:
            return g.intra_list_permutation_introduces_cycle(bl.entity, bl.start + m.c.a, nth_permutation(m.c.b, m.c.c));
    }
    return false;
}
inline size_t prec_filtered_move_count(const CriticalBlock& bl, const PrecedenceRouteGraph& g) {  // (:254-263)
    size_t count = 0;
    for (size_t i = 0; i < bl.move_count(); ++i) count += !prec_move_introduces_route_cycle(bl, i, g);
    return count;
}

// ---- cross-block support swaps and multi-ruins (support.rs:1-195) ----
struct AdjacentSwap {
    size_t entity, position;
    bool operator==(const AdjacentSwap& o) const { return entity == o.entity && position == o.position; }
};
inline void push_unique_swap(std::vector<AdjacentSwap>& v, AdjacentSwap s) {
    if (std::find(v.begin(), v.end(), s) == v.end()) v.push_back(s);
}
inline std::vector<AdjacentSwap> critical_adjacent_swaps(const std::vector<CriticalBlock>& blocks) {  // (:22-36)
    std::vector<AdjacentSwap> v;
    for (const auto& b : blocks)
        for (size_t p = b.start; p < b.end; ++p) push_unique_swap(v, {b.entity, p});
    return v;
}
inline void push_support_adjacent_swaps(const PrecedenceRouteGraph& g, size_t node, std::vector<AdjacentSwap>& v) {  // (:163-188)
    size_t entity, position;
    if (!g.node_route_position(node, entity, position)) return;
    const auto* route = g.route(entity);
    if (!route) return;
    if (position > 0) push_unique_swap(v, {entity, position - 1});
    if (position + 1 < route->size()) push_unique_swap(v, {entity, position});
}
inline std::vector<AdjacentSwap> support_adjacent_swaps(const std::vector<CriticalBlock>& blocks, const PrecedenceRouteGraph& g) {  // (:38-62)
    std::vector<AdjacentSwap> v;
    for (const auto& b : blocks) {
        const auto* route = g.route(b.entity);
        if (!route) continue;
        for (size_t p = b.start; p <= b.end; ++p) {
            if (p >= route->size()) continue;
            size_t node = (*route)[p];
            for (size_t s : g.fixed_successors[node]) push_support_adjacent_swaps(g, s, v);
            for (size_t q : g.fixed_predecessors[node]) push_support_adjacent_swaps(g, q, v);
        }
    }
    return v;
}
inline size_t multi_support_swap_count(const std::vector<AdjacentSwap>& critical, const std::vector<AdjacentSwap>& support) {  // (:64-84)
    size_t count = 0;
    for (size_t i = 0; i < critical.size(); ++i)
        for (size_t j = i + 1; j < critical.size(); ++j) {
            if (critical[i].entity == critical[j].entity) continue;
            for (const auto& s : support) count += s.entity != critical[i].entity && s.entity != critical[j].entity;
        }
    return count;
}
inline std::vector<PrecedenceRouteGraph::SwapCoord> multi_support_swaps(const std::vector<AdjacentSwap>& critical, const std::vector<AdjacentSwap>& support,
                                                                        size_t offset) {  // (:86-109)
    for (size_t i = 0; i < critical.size(); ++i)
        for (size_t j = i + 1; j < critical.size(); ++j) {
            if (critical[i].entity == critical[j].entity) continue;
            for (const auto& s : support) {
                if (s.entity == critical[i].entity || s.entity == critical[j].entity) continue;
                if (offset == 0)
                    return {{critical[i].entity, critical[i].position, critical[i].position + 1},
                            {critical[j].entity, critical[j].position, critical[j].position + 1},
                            {s.entity, s.position, s.position + 1}};
                --offset;
            }
        }
    return {};
}
inline size_t multi_critical_ruin_count(const std::vector<CriticalBlock>& blocks) {  // (:124-137)
    size_t count = 0;
    for (size_t i = 0; i < blocks.size(); ++i)
        for (size_t j = i + 1; j < blocks.size(); ++j) count += blocks[i].len() * blocks[j].len();
    return count;
}
// multi_critical_ruin_sources (:139-161): (entity, position) of the two ruined elements
inline bool multi_critical_ruin_sources(const std::vector<CriticalBlock>& blocks, size_t offset, size_t out[4]) {
    for (size_t i = 0; i < blocks.size(); ++i)
        for (size_t j = i + 1; j < blocks.size(); ++j) {
            size_t sc = blocks[j].len(), pairs = blocks[i].len() * sc;
            if (offset >= pairs) {
                offset -= pairs;
                continue;
            }
            out[0] = blocks[i].entity, out[1] = blocks[i].start + offset / sc, out[2] = blocks[j].entity, out[3] = blocks[j].start + offset % sc;
            return true;
        }
    return false;
}

// ListPrecedenceMoveSelector::size (list_precedence.rs:199-209)
inline size_t precedence_selector_size(const CriticalAnalysis& an) {
    size_t total = 0;
    for (const auto& b : an.blocks) total += prec_filtered_move_count(b, an.graph);
    auto critical = critical_adjacent_swaps(an.blocks);
    auto support = support_adjacent_swaps(an.blocks, an.graph);
    size_t ms = multi_support_swap_count(critical, support);
    for (size_t o = 0; o < ms; ++o) total += !an.graph.multi_intra_list_swaps_introduce_cycle(multi_support_swaps(critical, support, o));
    return total + multi_critical_ruin_count(an.blocks);
}

}  // namespace sfo
