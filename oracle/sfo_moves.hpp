// ORACLE — TEST INFRASTRUCTURE ONLY (see sfo_core.hpp header).
//
// CPU restatement of the in-scope moves and seeded candidate cursors.
//
// Follows:
//   heuristic/move/traits.rs:30-95, move/change.rs:118-221, move/swap.rs:150-215
//   heuristic/selector/scalar_neighborhood/move/apply.rs:10-335 (doable / apply_one / apply_many)
//   heuristic/move/list_kernel/change.rs:16-153, list_kernel/swap.rs:17-110
//   heuristic/selector/scalar_neighborhood/cursor/change.rs:27-121, cursor/swap.rs:22-160,
//       cursor.rs:371-378 (slot_identity)
//   heuristic/selector/list_kernel/change.rs:25-241, list_kernel/swap.rs:25-270
//   heuristic/selector/list_kernel/nearby_change.rs:17-233, nearby_swap.rs:17-260
//   heuristic/selector/list_kernel/reverse.rs:12-108, sublist_change.rs:13-266, sublist_swap.rs:13-330
//   heuristic/move/list_kernel/reverse.rs:15-57, sublist_change.rs:18-130, sublist_swap.rs:17-160
//   heuristic/selector/nearby_list_support.rs:3-34 (stable bounded top-k)
//   heuristic/selector/list_support.rs:13-26,62-70
//   runtime/compiler/executor/list_leaf/cursor/slot.rs:196-499 (runtime leaf entity order)
//   heuristic/selector/decorator/vec_union.rs:190-365 (UnionScheduler)
//   heuristic/selector/list_kernel/ruin.rs:1-181, heuristic/move/list_kernel/ruin.rs:1-435 (list ruin + greedy recreate)
// (all paths under crates/solverforge-solver/src/)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <vector>

#include "sfo_precedence_route.hpp"
#include "sfo_scoring.hpp"

namespace sfo {

struct Move {
    enum Kind : int32_t { Change = 0, Swap = 1, ListChange = 2, ListSwap = 3, ListReverse = 4, SublistChange = 5, SublistSwap = 6, KOpt = 7, Ruin = 8, ListPermute = 9, MultiSwap = 10 } kind = Change;
    size_t descriptor = 0;
    size_t variable = 0;
    // Change: a = entity, to_value.  Swap: a = left entity, b = right entity.
    // ListChange: (a, a_pos) -> (b, b_pos) [pre-removal destination coords].
    // ListSwap: (a, a_pos) <-> (b, b_pos).
    // ListReverse: reverse list a over [a_pos, b_pos) (b = a).
    // SublistChange: segment [a_pos, to_value) of list a -> list b at position b_pos (post-removal
    //                coordinates when a == b).
    // SublistSwap: segment [a_pos, a_pos + (to_value & 0xFFFF)) of list a <-> segment
    //                [b_pos, b_pos + (to_value >> 16)) of list b.
    // KOpt (3-opt, one list): list a cut at positions a_pos < b < b_pos (NOTE: `b` carries the middle
    //                cut, not an entity), reconnected by THREE_OPT_RECONNECTIONS[to_value].
    // ListPermute: the window [a_pos, b_pos) of list a (b = a) reordered by the to_value-th permutation of its positions in
    //                lexicographic order (nth_permutation, selector/list_kernel/permute.rs:260-272; rank 0 = identity is no move).
    // Ruin (list ruin-and-recreate, one source list): list a loses the a_pos elements at the ascending positions
    //                ruin_idx[0..a_pos) and every removed element is greedily re-inserted (move/list_kernel/ruin.rs:131-281);
    //                `allows_unassigned` carries skip_empty_destinations.  `ruin_multi`: a multi-source ruin (new_multi_source,
    //                move/list_ruin.rs:80-100): ruin_src[i] = the list ruin_idx[i] belongs to, entries sorted by (list, position)
    //                as merged_ruin_sources leaves them (a = the first list).  `prec` != null: the move carries the precedence
    //                hooks (with_precedence_hooks, :148-158) and the recreate skips insertions that close a cycle.
    // MultiSwap (move/list_kernel/multi_swap.rs:13-128): a_pos swaps (ms_entity[i], ms_first[i]) <-> (ms_entity[i], ms_second[i]) in
    //                pairwise different lists, applied as one move; `require_improvement` = with_require_score_improvement.
    size_t a = 0, a_pos = 0, b = 0, b_pos = 0;
    int64_t to_value = NONE;
    bool allows_unassigned = false;
    uint16_t ruin_idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // SmallVec<[usize; 8]> of the reference; this build caps a ruin at 6
    bool ruin_multi = false;
    uint16_t ruin_src[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const PrecedenceHooks* prec = nullptr;  // not owned; not part of the move's identity
    uint16_t ms_entity[4] = {0, 0, 0, 0}, ms_first[4] = {0, 0, 0, 0}, ms_second[4] = {0, 0, 0, 0};
    bool require_improvement = false;  // Move::requires_score_improvement (evaluation.rs:95-113)
    size_t ruin_source(size_t i) const { return ruin_multi ? (size_t)ruin_src[i] : a; }
};
inline bool operator==(const Move& x, const Move& y) {
    return x.kind == y.kind && x.descriptor == y.descriptor && x.variable == y.variable && x.a == y.a &&
           x.a_pos == y.a_pos && x.b == y.b && x.b_pos == y.b_pos && x.to_value == y.to_value &&
           (x.kind != Move::Ruin || (std::equal(x.ruin_idx, x.ruin_idx + 8, y.ruin_idx) && x.ruin_multi == y.ruin_multi &&
                                     (!x.ruin_multi || std::equal(x.ruin_src, x.ruin_src + 8, y.ruin_src)) && (x.prec != nullptr) == (y.prec != nullptr))) &&
           (x.kind != Move::MultiSwap || (std::equal(x.ms_entity, x.ms_entity + 4, y.ms_entity) && std::equal(x.ms_first, x.ms_first + 4, y.ms_first) &&
                                          std::equal(x.ms_second, x.ms_second + 4, y.ms_second) && x.require_improvement == y.require_improvement));
}

struct RuinPlacement {  // RuinUndo entry (move/list_kernel/ruin.rs:16): where a removed element went
    size_t entity, position, removed_index;
};
struct MoveUndo {
    int64_t old_a = NONE, old_b = NONE;
    std::vector<uint32_t> old_list;  // KOpt: the route before the reconnection (k_opt_do_move returns it)
    std::vector<RuinPlacement> placements;  // Ruin: in insertion order; empty when the recreate was rolled back
};

// final_positions_after_ordered_insertions (move/list_kernel/ruin.rs:81-95): where each placed element sits after all the
// later insertions into the same list
inline std::vector<size_t> ruin_final_positions(const std::vector<RuinPlacement>& placements) {
    std::vector<size_t> cur;
    for (size_t i = 0; i < placements.size(); ++i) {
        for (size_t j = 0; j < i; ++j)
            if (placements[j].entity == placements[i].entity && cur[j] >= placements[i].position) cur[j] += 1;
        cur.push_back(placements[i].position);
    }
    return cur;
}

// Reconnection patterns of a k-opt move (heuristic/move/k_opt_reconnection.rs:54-58,203-211):
// `order[p]` = which of the k+1 segments sits at position p afterwards, bit i of `reverse` = segment i
// (ORIGINAL index) is reversed.
struct KOptReconnection {
    uint8_t order[6];
    uint8_t reverse;
    uint8_t len;
};
static const KOptReconnection THREE_OPT_RECONNECTIONS[7] = {
    {{0, 1, 2, 3, 0, 0}, 0b0010, 4}, {{0, 1, 2, 3, 0, 0}, 0b0100, 4}, {{0, 1, 2, 3, 0, 0}, 0b0110, 4},
    {{0, 2, 1, 3, 0, 0}, 0b0000, 4}, {{0, 2, 1, 3, 0, 0}, 0b0010, 4}, {{0, 2, 1, 3, 0, 0}, 0b0100, 4},
    {{0, 2, 1, 3, 0, 0}, 0b0110, 4},
};

// nth_permutation (selector/list_kernel/permute.rs:260-272): the rank-th permutation of 0..len in lexicographic order
constexpr size_t MAX_LIST_PERMUTE_WINDOW_SIZE = 8;  // move/list_permute.rs:16
inline size_t permute_factorial(size_t v) {
    size_t f = 1;
    for (size_t i = 2; i <= v; ++i) f *= i;
    return f;
}
inline std::vector<size_t> nth_permutation(size_t len, size_t rank) {
    std::vector<size_t> remaining(len), perm;
    for (size_t i = 0; i < len; ++i) remaining[i] = i;
    for (size_t position = 0; position < len; ++position) {
        size_t step = permute_factorial(len - position - 1);
        size_t index = rank / step;
        rank %= step;
        perm.push_back(remaining[index]);
        remaining.erase(remaining.begin() + (ptrdiff_t)index);
    }
    return perm;
}

}  // namespace sfo
#include "sfo_precedence_leaf.hpp"
namespace sfo {

inline bool move_is_doable(const ScoreDirector& d, const Move& m) {
    const Solution& s = d.working;
    const EntityClass& c = s.classes[m.descriptor];
    switch (m.kind) {
        case Move::Change: {  // scalar_neighborhood/move/apply.rs:15-24,219-230
            if (m.a >= c.n) return false;
            if (c.vars[m.variable][m.a] == m.to_value) return false;
            return m.to_value != NONE || m.allows_unassigned;
        }
        case Move::Swap: {  // apply.rs:25-35,232-245
            if (m.a == m.b || m.a >= c.n || m.b >= c.n) return false;
            return c.vars[m.variable][m.a] != c.vars[m.variable][m.b];
        }
        case Move::ListChange: {  // move/list_kernel/change.rs:44-71
            size_t src_len = c.lists[m.a].size();
            if (m.a_pos >= src_len) return false;
            bool intra = m.a == m.b;
            size_t max_dst = intra ? src_len : c.lists[m.b].size();
            if (m.b_pos > max_dst) return false;
            return !intra || (m.a_pos != m.b_pos && m.b_pos != m.a_pos + 1);
        }
        case Move::ListSwap: {  // move/list_kernel/swap.rs:30-56
            if (m.a_pos >= c.lists[m.a].size() || m.b_pos >= c.lists[m.b].size()) return false;
            if (m.a == m.b && m.a_pos == m.b_pos) return false;
            return c.lists[m.a][m.a_pos] != c.lists[m.b][m.b_pos];
        }
        case Move::ListReverse:  // move/list_kernel/reverse.rs:22-36
            return m.a < c.lists.size() && m.b_pos > m.a_pos + 1 && m.b_pos <= c.lists[m.a].size();
        case Move::ListPermute: {  // permute_is_doable (move/list_kernel/permute.rs:22-38): a window inside the list, a non-identity permutation
            if (m.a >= c.lists.size() || !(m.a_pos < m.b_pos) || m.b_pos > c.lists[m.a].size()) return false;
            size_t len = m.b_pos - m.a_pos;
            if (len < 2 || len > MAX_LIST_PERMUTE_WINDOW_SIZE) return false;
            return m.to_value >= 1 && (size_t)m.to_value < permute_factorial(len);
        }
        case Move::KOpt: {  // k_opt_is_doable (move/list_kernel/k_opt.rs:13-41): cuts inside the list, strictly increasing
            if (m.to_value < 0 || m.to_value >= 7 || m.a >= c.lists.size()) return false;
            size_t len = c.lists[m.a].size();
            if (m.a_pos > len || m.b > len || m.b_pos > len) return false;
            return m.b > m.a_pos && m.b_pos > m.b;
        }
        case Move::Ruin: {  // ruin_is_doable without an owner binding (move/list_kernel/ruin.rs:97-113)
            if (m.a_pos == 0 || m.a_pos > 8 || m.a >= c.lists.size()) return false;
            for (size_t i = 0; i < m.a_pos; ++i)
                if (m.ruin_source(i) >= c.lists.size() || m.ruin_idx[i] >= c.lists[m.ruin_source(i)].size()) return false;
            return true;
        }
        case Move::MultiSwap: {  // multi_swap_is_doable (move/list_kernel/multi_swap.rs:30-60)
            if (m.a_pos == 0 || m.a_pos > 4) return false;
            for (size_t i = 0; i < m.a_pos; ++i) {
                size_t e = m.ms_entity[i], f = m.ms_first[i], g = m.ms_second[i];
                if (f == g) return false;
                for (size_t j = 0; j < i; ++j)
                    if (m.ms_entity[j] == e) return false;
                if (e >= c.lists.size() || f >= c.lists[e].size() || g >= c.lists[e].size()) return false;
                if (c.lists[e][f] == c.lists[e][g]) return false;
            }
            return true;
        }
        case Move::SublistSwap: {  // move/list_kernel/sublist_swap.rs:17-43
            if (m.to_value < 0) return false;
            size_t fs = m.a_pos, fe = m.a_pos + (size_t)(m.to_value & 0xFFFF);
            size_t ss = m.b_pos, se = m.b_pos + (size_t)(m.to_value >> 16);
            if (fs >= fe || ss >= se) return false;
            if (fe > c.lists[m.a].size() || se > c.lists[m.b].size()) return false;
            return !(m.a == m.b && fs < se && ss < fe);
        }
        case Move::SublistChange: {  // move/list_kernel/sublist_change.rs:18-50
            size_t start = m.a_pos, end = (size_t)m.to_value;
            if (m.to_value < 0 || start >= end) return false;
            size_t src_len = c.lists[m.a].size();
            if (end > src_len) return false;
            size_t max_dst = m.a == m.b ? src_len - (end - start) : c.lists[m.b].size();
            if (m.b_pos > max_dst) return false;
            return m.a != m.b || m.b_pos != start;
        }
    }
    return false;
}

inline MoveUndo move_do(ScoreDirector& d, const Move& m) {
    MoveUndo u;
    EntityClass& c = d.working.classes[m.descriptor];
    switch (m.kind) {
        case Move::Change: {  // apply_one (apply.rs:288-303)
            u.old_a = c.vars[m.variable][m.a];
            d.before_variable_changed(m.descriptor, m.a);
            c.vars[m.variable][m.a] = m.to_value;
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::Swap: {  // apply_many (apply.rs:305-335)
            u.old_a = c.vars[m.variable][m.a];
            u.old_b = c.vars[m.variable][m.b];
            d.before_variable_changed(m.descriptor, m.a);
            d.before_variable_changed(m.descriptor, m.b);
            c.vars[m.variable][m.a] = u.old_b;
            c.vars[m.variable][m.b] = u.old_a;
            d.after_variable_changed(m.descriptor, m.a);
            d.after_variable_changed(m.descriptor, m.b);
            break;
        }
        case Move::ListChange: {  // move/list_kernel/change.rs:73-120
            bool intra = m.a == m.b;
            d.before_variable_changed(m.descriptor, m.a);
            if (!intra) d.before_variable_changed(m.descriptor, m.b);
            uint32_t value = c.lists[m.a][m.a_pos];
            c.lists[m.a].erase(c.lists[m.a].begin() + (ptrdiff_t)m.a_pos);
            size_t dst = (intra && m.b_pos > m.a_pos) ? m.b_pos - 1 : m.b_pos;  // adjusted_destination
            c.lists[m.b].insert(c.lists[m.b].begin() + (ptrdiff_t)dst, value);
            d.after_variable_changed(m.descriptor, m.a);
            if (!intra) d.after_variable_changed(m.descriptor, m.b);
            break;
        }
        case Move::ListSwap: {  // move/list_kernel/swap.rs:58-110
            bool intra = m.a == m.b;
            uint32_t first = c.lists[m.a][m.a_pos];
            uint32_t second = c.lists[m.b][m.b_pos];
            d.before_variable_changed(m.descriptor, m.a);
            if (!intra) d.before_variable_changed(m.descriptor, m.b);
            c.lists[m.a][m.a_pos] = second;
            c.lists[m.b][m.b_pos] = first;
            d.after_variable_changed(m.descriptor, m.a);
            if (!intra) d.after_variable_changed(m.descriptor, m.b);
            break;
        }
        case Move::ListReverse: {  // move/list_kernel/reverse.rs:38-57
            d.before_variable_changed(m.descriptor, m.a);
            std::reverse(c.lists[m.a].begin() + (ptrdiff_t)m.a_pos, c.lists[m.a].begin() + (ptrdiff_t)m.b_pos);
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::ListPermute: {  // permute_do_move (move/list_kernel/permute.rs:40-72): remove the window, re-insert it reordered
            d.before_variable_changed(m.descriptor, m.a);
            auto& l = c.lists[m.a];
            u.old_list.assign(l.begin() + (ptrdiff_t)m.a_pos, l.begin() + (ptrdiff_t)m.b_pos);
            std::vector<size_t> perm = nth_permutation(m.b_pos - m.a_pos, (size_t)m.to_value);
            for (size_t k = 0; k < perm.size(); ++k) l[m.a_pos + k] = u.old_list[perm[k]];
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::SublistSwap: {  // apply_sublist_swap (move/list_kernel/sublist_swap.rs:78-160): the segments
                                   // trade places, each keeping its internal order
            bool intra = m.a == m.b;
            size_t fs = m.a_pos, fe = m.a_pos + (size_t)(m.to_value & 0xFFFF);
            size_t ss = m.b_pos, se = m.b_pos + (size_t)(m.to_value >> 16);
            d.before_variable_changed(m.descriptor, m.a);
            if (!intra) d.before_variable_changed(m.descriptor, m.b);
            if (intra) {
                auto& l = c.lists[m.a];
                size_t es = fs < ss ? fs : ss, ee = fs < ss ? fe : se, ls = fs < ss ? ss : fs, le = fs < ss ? se : fe;
                std::vector<uint32_t> out(l.begin(), l.begin() + (ptrdiff_t)es);
                out.insert(out.end(), l.begin() + (ptrdiff_t)ls, l.begin() + (ptrdiff_t)le);
                out.insert(out.end(), l.begin() + (ptrdiff_t)ee, l.begin() + (ptrdiff_t)ls);
                out.insert(out.end(), l.begin() + (ptrdiff_t)es, l.begin() + (ptrdiff_t)ee);
                out.insert(out.end(), l.begin() + (ptrdiff_t)le, l.end());
                l = out;
            } else {
                auto& la = c.lists[m.a];
                auto& lb = c.lists[m.b];
                std::vector<uint32_t> sa(la.begin() + (ptrdiff_t)fs, la.begin() + (ptrdiff_t)fe);
                std::vector<uint32_t> sb(lb.begin() + (ptrdiff_t)ss, lb.begin() + (ptrdiff_t)se);
                la.erase(la.begin() + (ptrdiff_t)fs, la.begin() + (ptrdiff_t)fe);
                la.insert(la.begin() + (ptrdiff_t)fs, sb.begin(), sb.end());
                lb.erase(lb.begin() + (ptrdiff_t)ss, lb.begin() + (ptrdiff_t)se);
                lb.insert(lb.begin() + (ptrdiff_t)ss, sa.begin(), sa.end());
            }
            d.after_variable_changed(m.descriptor, m.a);
            if (!intra) d.after_variable_changed(m.descriptor, m.b);
            break;
        }
        case Move::KOpt: {  // k_opt_do_move (move/list_kernel/k_opt.rs:43-96)
            d.before_variable_changed(m.descriptor, m.a);
            auto& l = c.lists[m.a];
            u.old_list = l;
            const KOptReconnection& r = THREE_OPT_RECONNECTIONS[m.to_value];
            const size_t bounds[5] = {0, m.a_pos, m.b, m.b_pos, l.size()};
            std::vector<uint32_t> out;
            out.reserve(l.size());
            for (size_t p = 0; p < r.len; ++p) {
                size_t seg = r.order[p];
                std::vector<uint32_t> part(u.old_list.begin() + (ptrdiff_t)bounds[seg], u.old_list.begin() + (ptrdiff_t)bounds[seg + 1]);
                if ((r.reverse >> seg) & 1) std::reverse(part.begin(), part.end());
                out.insert(out.end(), part.begin(), part.end());
            }
            l = out;
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::Ruin: {  // ruin_do_move (move/list_kernel/ruin.rs:131-281): no owner binding.  Every trial insertion is a full
                            // before / insert / after / calculate_score; with precedence hooks the route graph is rebuilt from
                            // the working lists every round and insertions that close a cycle are skipped (:186-220).
            struct Removed {
                size_t removed_index, source, original_position;
                uint32_t value;
            };
            std::vector<Removed> remaining;
            for (size_t i0 = 0; i0 < m.a_pos;) {  // one (source, ascending indices) group at a time (:146-164)
                size_t source = m.ruin_source(i0), i1 = i0;
                while (i1 < m.a_pos && m.ruin_source(i1) == source) ++i1;
                d.before_variable_changed(m.descriptor, source);
                std::vector<Removed> part;
                for (size_t i = i1; i-- > i0;) {  // remove from the back
                    size_t index = m.ruin_idx[i];
                    part.push_back({i, source, index, c.lists[source][index]});
                    c.lists[source].erase(c.lists[source].begin() + (ptrdiff_t)index);
                }
                std::reverse(part.begin(), part.end());
                remaining.insert(remaining.end(), part.begin(), part.end());
                d.after_variable_changed(m.descriptor, source);
                i0 = i1;
            }
            const std::vector<Removed> removed = remaining;
            const bool skip_empty = m.allows_unassigned;
            const size_t entity_count = c.n;
            while (!remaining.empty()) {
                PrecedenceRouteGraph graph;
                if (m.prec) graph = PrecedenceRouteGraph::build(*m.prec, c.lists);  // recreate_precedence_graph (ruin_access.rs:129-147)
                bool have = false;
                size_t best_ri = 0, best_e = 0, best_p = 0;
                Score best_score;
                for (size_t ri = 0; ri < remaining.size(); ++ri) {
                    for (size_t e = 0; e < entity_count; ++e) {
                        size_t len = c.lists[e].size();
                        if (skip_empty && len == 0) continue;
                        for (size_t pos = 0; pos <= len; ++pos) {
                            if (m.prec && (size_t)remaining[ri].value < m.prec->node_count) {
                                auto node_of = [&](size_t p) { return (size_t)c.lists[e][p] < m.prec->node_count ? (size_t)c.lists[e][p] : SIZE_MAX; };
                                size_t previous = pos > 0 ? node_of(pos - 1) : SIZE_MAX, next = pos < len ? node_of(pos) : SIZE_MAX;
                                if (graph.insertion_introduces_cycle(previous, (size_t)remaining[ri].value, next)) continue;
                            }
                            d.before_variable_changed(m.descriptor, e);
                            c.lists[e].insert(c.lists[e].begin() + (ptrdiff_t)pos, remaining[ri].value);
                            d.after_variable_changed(m.descriptor, e);
                            Score cand = d.calculate_score();
                            if (!have || cand > best_score) {  // strict: the first of equal scores stays (:228-237)
                                have = true;
                                best_ri = ri, best_e = e, best_p = pos, best_score = cand;
                            }
                            d.before_variable_changed(m.descriptor, e);
                            c.lists[e].erase(c.lists[e].begin() + (ptrdiff_t)pos);
                            d.after_variable_changed(m.descriptor, e);
                        }
                    }
                }
                if (!have) {  // restore_removed_elements (:250-253,375-404)
                    std::vector<size_t> cur = ruin_final_positions(u.placements);
                    for (size_t i = u.placements.size(); i-- > 0;) {
                        size_t e = u.placements[i].entity, at = cur[i];
                        d.before_variable_changed(m.descriptor, e);
                        c.lists[e].erase(c.lists[e].begin() + (ptrdiff_t)at);
                        d.after_variable_changed(m.descriptor, e);
                        for (size_t j = 0; j < i; ++j)
                            if (u.placements[j].entity == e && cur[j] > at) cur[j] -= 1;
                    }
                    // restore_values (:406-435): removed is already sorted by (source, original position)
                    for (size_t i0 = 0; i0 < removed.size();) {
                        size_t source = removed[i0].source, i1 = i0;
                        d.before_variable_changed(m.descriptor, source);
                        for (; i1 < removed.size() && removed[i1].source == source; ++i1)
                            c.lists[source].insert(c.lists[source].begin() + (ptrdiff_t)removed[i1].original_position, removed[i1].value);
                        d.after_variable_changed(m.descriptor, source);
                        i0 = i1;
                    }
                    u.placements.clear();
                    break;
                }
                Removed r = remaining[best_ri];
                remaining.erase(remaining.begin() + (ptrdiff_t)best_ri);
                d.before_variable_changed(m.descriptor, best_e);
                c.lists[best_e].insert(c.lists[best_e].begin() + (ptrdiff_t)best_p, r.value);
                d.after_variable_changed(m.descriptor, best_e);
                u.placements.push_back({best_e, best_p, r.removed_index});
            }
            break;
        }
        case Move::MultiSwap: {  // multi_swap_do_move (move/list_kernel/multi_swap.rs:62-112): read every pair, notify every list
                                 // (the lists are pairwise different: is_doable), write, notify
            uint32_t first_value[4], second_value[4];
            for (size_t i = 0; i < m.a_pos; ++i) {
                first_value[i] = c.lists[m.ms_entity[i]][m.ms_first[i]];
                second_value[i] = c.lists[m.ms_entity[i]][m.ms_second[i]];
            }
            for (size_t i = 0; i < m.a_pos; ++i) d.before_variable_changed(m.descriptor, m.ms_entity[i]);
            for (size_t i = 0; i < m.a_pos; ++i) {
                c.lists[m.ms_entity[i]][m.ms_first[i]] = second_value[i];
                c.lists[m.ms_entity[i]][m.ms_second[i]] = first_value[i];
            }
            for (size_t i = 0; i < m.a_pos; ++i) d.after_variable_changed(m.descriptor, m.ms_entity[i]);
            break;
        }
        case Move::SublistChange: {  // apply_sublist_change (move/list_kernel/sublist_change.rs:88-130)
            bool intra = m.a == m.b;
            d.before_variable_changed(m.descriptor, m.a);
            if (!intra) d.before_variable_changed(m.descriptor, m.b);
            auto& src = c.lists[m.a];
            std::vector<uint32_t> seg(src.begin() + (ptrdiff_t)m.a_pos, src.begin() + (ptrdiff_t)m.to_value);
            src.erase(src.begin() + (ptrdiff_t)m.a_pos, src.begin() + (ptrdiff_t)m.to_value);
            auto& dst = c.lists[m.b];
            dst.insert(dst.begin() + (ptrdiff_t)m.b_pos, seg.begin(), seg.end());
            d.after_variable_changed(m.descriptor, m.a);
            if (!intra) d.after_variable_changed(m.descriptor, m.b);
            break;
        }
    }
    return u;
}

inline void move_undo(ScoreDirector& d, const Move& m, const MoveUndo& u) {
    EntityClass& c = d.working.classes[m.descriptor];
    switch (m.kind) {
        case Move::Change: {
            d.before_variable_changed(m.descriptor, m.a);
            c.vars[m.variable][m.a] = u.old_a;
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::Swap: {
            d.before_variable_changed(m.descriptor, m.a);
            d.before_variable_changed(m.descriptor, m.b);
            c.vars[m.variable][m.a] = u.old_a;
            c.vars[m.variable][m.b] = u.old_b;
            d.after_variable_changed(m.descriptor, m.a);
            d.after_variable_changed(m.descriptor, m.b);
            break;
        }
        case Move::ListChange: {  // change_undo_move (move/list_kernel/change.rs:122-153)
            bool intra = m.a == m.b;
            size_t dst = (intra && m.b_pos > m.a_pos) ? m.b_pos - 1 : m.b_pos;
            d.before_variable_changed(m.descriptor, m.b);
            if (!intra) d.before_variable_changed(m.descriptor, m.a);
            uint32_t value = c.lists[m.b][dst];
            c.lists[m.b].erase(c.lists[m.b].begin() + (ptrdiff_t)dst);
            c.lists[m.a].insert(c.lists[m.a].begin() + (ptrdiff_t)m.a_pos, value);
            d.after_variable_changed(m.descriptor, m.b);
            if (!intra) d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::SublistSwap: {  // the inverse exchanges the segments where they now sit
            Move inv = m;
            size_t za = (size_t)(m.to_value & 0xFFFF), zb = (size_t)(m.to_value >> 16);
            if (m.a == m.b) {
                if (m.a_pos < m.b_pos) {  // early segment now has size zb; the late one ends where it ended
                    inv.a_pos = m.a_pos;
                    inv.b_pos = m.b_pos + zb - za;
                } else {
                    inv.b_pos = m.b_pos;
                    inv.a_pos = m.a_pos + za - zb;
                }
            }
            inv.to_value = (int64_t)(zb | (za << 16));
            MoveUndo ignored = move_do(d, inv);
            (void)ignored;
            break;
        }
        case Move::SublistChange: {  // inverse relocation (move/segment_layout.rs:48-72)
            Move inv = m;
            size_t len = (size_t)m.to_value - m.a_pos;
            inv.a = m.b;
            inv.a_pos = m.b_pos;
            inv.to_value = (int64_t)(m.b_pos + len);
            inv.b = m.a;
            inv.b_pos = m.a_pos;
            MoveUndo ignored = move_do(d, inv);
            (void)ignored;
            break;
        }
        case Move::Ruin: {  // ruin_undo_move (move/list_kernel/ruin.rs:283-322) + restore_values (:406-435)
            if (u.placements.empty()) break;  // the recreate was rolled back inside do_move
            std::vector<size_t> cur = ruin_final_positions(u.placements);
            struct Back {
                size_t source, original_position;
                uint32_t value;
            };
            std::vector<Back> values;
            for (size_t i = u.placements.size(); i-- > 0;) {
                size_t e = u.placements[i].entity, at = cur[i];
                d.before_variable_changed(m.descriptor, e);
                uint32_t value = c.lists[e][at];
                c.lists[e].erase(c.lists[e].begin() + (ptrdiff_t)at);
                size_t ri = u.placements[i].removed_index;  // removed_source_entry
                values.push_back({m.ruin_source(ri), (size_t)m.ruin_idx[ri], value});
                d.after_variable_changed(m.descriptor, e);
                for (size_t j = 0; j < i; ++j)
                    if (u.placements[j].entity == e && cur[j] > at) cur[j] -= 1;
            }
            std::sort(values.begin(), values.end(), [](const Back& x, const Back& y) {
                return x.source != y.source ? x.source < y.source : x.original_position < y.original_position;
            });
            for (size_t i0 = 0; i0 < values.size();) {
                size_t source = values[i0].source, i1 = i0;
                d.before_variable_changed(m.descriptor, source);
                for (; i1 < values.size() && values[i1].source == source; ++i1)
                    c.lists[source].insert(c.lists[source].begin() + (ptrdiff_t)values[i1].original_position, values[i1].value);
                d.after_variable_changed(m.descriptor, source);
                i0 = i1;
            }
            break;
        }
        case Move::ListPermute: {  // permute_undo_move (move/list_kernel/permute.rs:74-101): the original window back
            d.before_variable_changed(m.descriptor, m.a);
            for (size_t k = 0; k < u.old_list.size(); ++k) c.lists[m.a][m.a_pos + k] = u.old_list[k];
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::KOpt: {  // k_opt_undo_move (move/list_kernel/k_opt.rs:98-119): put the old route back
            d.before_variable_changed(m.descriptor, m.a);
            c.lists[m.a] = u.old_list;
            d.after_variable_changed(m.descriptor, m.a);
            break;
        }
        case Move::MultiSwap:      // multi_swap_undo_move (:114-128) = the same exchange again
        case Move::ListSwap:       // swap is its own inverse
        case Move::ListReverse: {  // so is a reversal
            MoveUndo ignored = move_do(d, m);
            (void)ignored;
            break;
        }
    }
}

// ---------------------------------------------------------------------------
// CompoundScalarMove (heuristic/move/compound_scalar.rs:207-330): several ScalarEdits of one ScalarCandidate
// (planning/scalar/candidate.rs:85-188) scored and applied as ONE move.
// ---------------------------------------------------------------------------
struct ScalarEditO {
    size_t descriptor, variable, entity;
    int64_t to_value;  // NONE = unassign
    bool legal;        // value_is_legal of the slot (:186-204): None only when the variable allows it, Some(v) only from the entity's value list
};
inline bool compound_is_doable(const ScoreDirector& d, const std::vector<ScalarEditO>& edits) {  // is_doable_on (:254-270)
    if (edits.empty()) return false;
    bool changes = false;
    for (const ScalarEditO& e : edits) {
        const EntityClass& c = d.working.classes[e.descriptor];
        if (e.entity >= c.n) return false;
        if (!e.legal) return false;
        changes = changes || c.vars[e.variable][e.entity] != e.to_value;
    }
    return changes;
}
inline std::vector<std::pair<size_t, size_t>> compound_affected(const std::vector<ScalarEditO>& edits) {  // unique_affected_entities (:395-405)
    std::vector<std::pair<size_t, size_t>> a;
    for (const ScalarEditO& e : edits) {
        std::pair<size_t, size_t> k{e.descriptor, e.entity};
        if (std::find(a.begin(), a.end(), k) == a.end()) a.push_back(k);
    }
    return a;
}
// do_move (:291-308): old values first, retract every affected entity, apply every edit in order, insert in reverse order
inline std::vector<int64_t> compound_do(ScoreDirector& d, const std::vector<ScalarEditO>& edits) {
    std::vector<int64_t> undo;
    auto affected = compound_affected(edits);
    for (const ScalarEditO& e : edits) undo.push_back(d.working.classes[e.descriptor].vars[e.variable][e.entity]);
    for (auto& a : affected) d.before_variable_changed(a.first, a.second);
    for (const ScalarEditO& e : edits) d.working.classes[e.descriptor].vars[e.variable][e.entity] = e.to_value;
    for (size_t i = affected.size(); i-- > 0;) d.after_variable_changed(affected[i].first, affected[i].second);
    return undo;
}
inline void compound_undo(ScoreDirector& d, const std::vector<ScalarEditO>& edits, const std::vector<int64_t>& undo) {  // (:310-321)
    auto affected = compound_affected(edits);
    for (auto& a : affected) d.before_variable_changed(a.first, a.second);
    for (size_t i = 0; i < edits.size(); ++i) d.working.classes[edits[i].descriptor].vars[edits[i].variable][edits[i].entity] = undo[i];
    for (size_t i = affected.size(); i-- > 0;) d.after_variable_changed(affected[i].first, affected[i].second);
}

// ---------------------------------------------------------------------------
// Cursors
// ---------------------------------------------------------------------------
struct Cursor {
    virtual ~Cursor() = default;
    virtual bool next(Move& out) = 0;
    // MoveCursor::selector_index of the candidate `next` just returned (vec_union.rs:505-509: the child cursor index)
    virtual size_t last_selector() const { return 0; }
};

struct ScalarSlot {
    size_t descriptor_index = 0;
    size_t variable_index = 0;
    bool allows_unassigned = false;
    bool empty_value_source = false;
    // canonical candidate values for one entity (ValueSource; builder/context/scalar/variable.rs:138-192)
    std::function<void(const Solution&, size_t entity, std::vector<int64_t>& out)> values_for_entity;
    uint64_t identity() const {  // cursor.rs:371-378
        return ((uint64_t)descriptor_index << 32) ^ (uint64_t)variable_index;
    }
    // ---- nearby sources (builder/context/scalar_access.rs:261-340): the slot's hooks as data --------------------------------
    // nearby_value_candidates / nearby_entity_candidates: one row per entity in SOURCE order (absent = the hook is None: the
    // cursor falls back to the ordinary candidate values with the source limit / to every entity); the distance meters as a
    // value per row entry (absent = the meter is None: the distance is the source order, change.rs:332-334, swap.rs:385-387).
    bool dynamic = false;  // DynamicScalarVariableSlot: nearby change re-checks value legality, nearby swap is directional
    bool has_nearby_values = false, has_nearby_entities = false;
    std::vector<std::vector<int64_t>> nearby_values, nearby_entities;
    // distance meters: called with (entity, candidate); NaN = None
    std::function<double(size_t entity, int64_t value)> nearby_value_distance;
    std::function<double(size_t left, size_t right)> nearby_entity_distance;
};

// NearbyTopK (heuristic/selector/nearby_support.rs:20-90): the best `limit` candidates by (distance total order, source order,
// candidate), non-finite distances dropped, returned in that order.
struct RankedNearby {
    int64_t candidate;
    double distance;
    size_t order;
};
inline int f64_total_cmp(double a, double b) {  // f64::total_cmp on the values that reach here (finite, -0.0 < +0.0)
    int64_t x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    x ^= (int64_t)((uint64_t)(x >> 63) >> 1);
    y ^= (int64_t)((uint64_t)(y >> 63) >> 1);
    return x < y ? -1 : (x > y ? 1 : 0);
}
inline std::vector<int64_t> nearby_top_k(std::vector<RankedNearby> all, size_t limit) {
    std::vector<RankedNearby> kept;
    for (auto& c : all)
        if (limit != 0 && std::isfinite(c.distance)) kept.push_back(c);
    std::sort(kept.begin(), kept.end(), [](const RankedNearby& l, const RankedNearby& r) {
        int c = f64_total_cmp(l.distance, r.distance);
        if (c != 0) return c < 0;
        if (l.order != r.order) return l.order < r.order;
        return l.candidate < r.candidate;
    });
    if (kept.size() > limit) kept.resize(limit);
    std::vector<int64_t> out;
    for (auto& c : kept) out.push_back(c.candidate);
    return out;
}

// Scalar change leaf (scalar_neighborhood/cursor/change.rs:27-121).
struct ScalarChangeCursor : Cursor {
    static constexpr uint64_t VALUE_SALT = 0xC4A46E0000000000ULL;
    static constexpr uint64_t ENTITY_SALT = 0xC4A46E0000000001ULL;
    struct Row {
        size_t entity;
        std::vector<int64_t> values;
        bool current_assigned;
    };
    ScalarSlot slot;
    std::vector<Row> rows;
    size_t row_offset = 0, value_offset = 0;
    bool unassigned_pending = false;

    ScalarChangeCursor(const ScalarSlot& sl, const Solution& solution_in, const MoveStreamContext& ctx)
        : slot(sl) {
        Solution solution = solution_in;  // the leaf clones the working solution (cursor.rs:89)
        uint64_t identity = slot.identity();
        size_t n = solution.classes[slot.descriptor_index].n;
        rows.reserve(n);
        std::vector<int64_t> canonical;
        for (size_t off = 0; off < n; ++off) {
            size_t e = ctx.selection_index_without_replacement(off, n, ENTITY_SALT ^ identity);
            canonical.clear();
            slot.values_for_entity(solution, e, canonical);
            size_t vc = canonical.size();
            Row row;
            row.entity = e;
            row.values.resize(vc);
            for (size_t vo = 0; vo < vc; ++vo)
                row.values[vo] = canonical[ctx.selection_index(vo, vc, VALUE_SALT ^ (uint64_t)e ^ identity)];
            row.current_assigned = solution.classes[slot.descriptor_index].vars[slot.variable_index][e] != NONE;
            rows.push_back(std::move(row));
        }
    }
    bool next(Move& out) override {
        for (;;) {
            if (row_offset >= rows.size()) return false;
            Row& row = rows[row_offset];
            if (value_offset < row.values.size()) {
                out = Move{};
                out.kind = Move::Change;
                out.descriptor = slot.descriptor_index;
                out.variable = slot.variable_index;
                out.a = row.entity;
                out.to_value = row.values[value_offset++];
                out.allows_unassigned = slot.allows_unassigned;
                return true;
            }
            if (!unassigned_pending && slot.allows_unassigned && row.current_assigned) {
                unassigned_pending = true;
                out = Move{};
                out.kind = Move::Change;
                out.descriptor = slot.descriptor_index;
                out.variable = slot.variable_index;
                out.a = row.entity;
                out.to_value = NONE;
                out.allows_unassigned = slot.allows_unassigned;
                return true;
            }
            ++row_offset;
            value_offset = 0;
            unassigned_pending = false;
        }
    }
};

// Scalar swap leaf (scalar_neighborhood/cursor/swap.rs:22-160).
struct ScalarSwapCursor : Cursor {
    static constexpr uint64_t LEFT_SALT = 0x5A095CA1AA000001ULL;
    static constexpr uint64_t RIGHT_SALT = 0x5A095CA1AA000002ULL;
    static constexpr uint64_t ENTITY_STRIDE_MIX = 0xD1B54A32D192ED03ULL;
    ScalarSlot slot;
    MoveStreamContext ctx;
    std::vector<int64_t> current_values;
    std::vector<std::vector<int64_t>> legal_values;
    size_t left_offset = 0, right_offset = 0;

    ScalarSwapCursor(const ScalarSlot& sl, const Solution& solution_in, const MoveStreamContext& c)
        : slot(sl), ctx(c) {
        Solution solution = solution_in;
        size_t n = solution.classes[slot.descriptor_index].n;
        current_values = solution.classes[slot.descriptor_index].vars[slot.variable_index];
        legal_values.resize(n);
        for (size_t e = 0; e < n; ++e) slot.values_for_entity(solution, e, legal_values[e]);
    }
    bool destination_is_legal(size_t e, int64_t value) const {  // swap.rs:103-123
        if (slot.empty_value_source) return value != NONE;
        if (value == NONE) return slot.allows_unassigned;
        for (int64_t v : legal_values[e])
            if (v == value) return true;
        return false;
    }
    bool next(Move& out) override {  // swap.rs:125-160
        size_t n = current_values.size();
        uint64_t identity = slot.identity();
        while (left_offset < n) {
            size_t left = n <= 1 ? 0
                                 : ctx.selection_index_without_replacement(
                                       left_offset, n, (LEFT_SALT ^ identity) ^ ENTITY_STRIDE_MIX);
            while (right_offset < n) {
                size_t right =
                    n <= 1 ? 0
                           : ctx.selection_index(right_offset, n,
                                                 (RIGHT_SALT ^ (uint64_t)left ^ (uint64_t)slot.variable_index) ^
                                                     ENTITY_STRIDE_MIX);
                ++right_offset;
                if (left >= right) continue;
                int64_t lv = current_values[left], rv = current_values[right];
                if (lv == rv || !destination_is_legal(left, rv) || !destination_is_legal(right, lv)) continue;
                out = Move{};
                out.kind = Move::Swap;
                out.descriptor = slot.descriptor_index;
                out.variable = slot.variable_index;
                out.a = left;
                out.b = right;
                out.allows_unassigned = slot.allows_unassigned;
                return true;
            }
            ++left_offset;
            right_offset = 0;
        }
        return false;
    }
};

// Nearby scalar leaves (scalar_neighborhood/cursor/change.rs:123-392, cursor/swap.rs:162-414).  The native (eager) and the
// dynamic (lazy) source timing read the same cursor-open snapshot, so both yield the stream restated here.
inline size_t nearby_ordered_entity(size_t n, size_t offset, const MoveStreamContext& ctx, uint64_t start_salt, uint64_t stride_salt,
                                    uint64_t identity) {  // change.rs:376-391
    if (n <= 1) return 0;
    size_t start = ctx.start_offset(n, start_salt ^ identity);
    size_t stride = ctx.stride(n, stride_salt ^ identity);
    return (start + offset * stride) % n;
}
template <class T>
inline void apply_selection_order(const MoveStreamContext& ctx, std::vector<T>& values, uint64_t salt) {  // iter.rs:152-161
    if (ctx.is_canonical()) return;
    std::vector<T> canonical = values;
    for (size_t o = 0; o < values.size(); ++o) values[o] = canonical[ctx.selection_index(o, canonical.size(), salt)];
}

struct NearbyScalarChangeCursor : Cursor {
    static constexpr uint64_t ENTITY_START_SALT = 0xC4A46E00AAAA0001ULL, ENTITY_STRIDE_SALT = 0xC4A46E00AAAA0002ULL;
    static constexpr uint64_t VALUE_SALT = 0xC4A46E00AAAA0003ULL;
    struct Row {
        size_t entity;
        std::vector<int64_t> values;
        bool unassigned_pending;
    };
    ScalarSlot slot;
    std::vector<Row> rows;
    size_t row_offset = 0, value_offset = 0;

    static bool value_is_legal(const ScalarSlot& slot, const Solution& s, size_t e, int64_t v) {  // variable.rs:207-227
        std::vector<int64_t> legal;
        slot.values_for_entity(s, e, legal);
        return std::find(legal.begin(), legal.end(), v) != legal.end();
    }
    // rank_nearby_values (change.rs:312-374)
    static std::vector<int64_t> rank(const ScalarSlot& slot, const Solution& s, size_t e, size_t max_nearby, size_t source_limit,
                                     const MoveStreamContext& ctx) {
        if (max_nearby == 0 || source_limit == 0) return {};
        const int64_t current = s.classes[slot.descriptor_index].vars[slot.variable_index][e];
        std::vector<RankedNearby> all;
        size_t order = 0;
        auto visit = [&](int64_t value) {
            size_t source_order = order++;
            if (current == value || (slot.dynamic && !value_is_legal(slot, s, e, value))) return;
            double dist = slot.nearby_value_distance ? slot.nearby_value_distance(e, value) : std::nan("");
            all.push_back(RankedNearby{value, dist != dist ? (double)source_order : dist, source_order});
        };
        if (slot.has_nearby_values) {
            const auto& row = slot.nearby_values[e];
            for (size_t i = 0; i < row.size() && i < source_limit; ++i) visit(row[i]);
        } else {  // visit_candidate_values(.., Some(source_limit), ..)
            std::vector<int64_t> vals;
            slot.values_for_entity(s, e, vals);
            for (size_t i = 0; i < vals.size() && i < source_limit; ++i) visit(vals[i]);
        }
        std::vector<int64_t> values = nearby_top_k(std::move(all), max_nearby);
        apply_selection_order(ctx, values, VALUE_SALT ^ (uint64_t)e ^ slot.identity());
        return values;
    }
    NearbyScalarChangeCursor(const ScalarSlot& sl, const Solution& solution_in, const MoveStreamContext& ctx, size_t max_nearby, size_t source_limit)
        : slot(sl) {
        Solution solution = solution_in;
        size_t n = solution.classes[slot.descriptor_index].n;
        for (size_t off = 0; off < n; ++off) {
            size_t e = nearby_ordered_entity(n, off, ctx, ENTITY_START_SALT, ENTITY_STRIDE_SALT, slot.identity());
            Row row;
            row.entity = e;
            row.values = rank(slot, solution, e, max_nearby, source_limit, ctx);
            row.unassigned_pending = slot.allows_unassigned && solution.classes[slot.descriptor_index].vars[slot.variable_index][e] != NONE;
            rows.push_back(std::move(row));
        }
    }
    bool next(Move& out) override {  // change.rs:186-216
        for (;;) {
            if (row_offset >= rows.size()) return false;
            Row& row = rows[row_offset];
            out = Move{};
            out.kind = Move::Change;
            out.descriptor = slot.descriptor_index;
            out.variable = slot.variable_index;
            out.a = row.entity;
            out.allows_unassigned = slot.allows_unassigned;
            if (value_offset < row.values.size()) {
                out.to_value = row.values[value_offset++];
                return true;
            }
            if (row.unassigned_pending) {
                row.unassigned_pending = false;
                out.to_value = NONE;
                return true;
            }
            ++row_offset;
            value_offset = 0;
        }
    }
};

struct NearbyScalarSwapCursor : Cursor {
    static constexpr uint64_t ENTITY_START_SALT = 0x5A095CA1AAAA0001ULL, ENTITY_STRIDE_SALT = 0x5A095CA1AAAA0002ULL;
    static constexpr uint64_t TARGET_SALT = 0x5A095CA1AAAA0003ULL;
    struct Row {
        size_t left;
        std::vector<int64_t> rights;
    };
    ScalarSlot slot;
    std::vector<Row> rows;
    size_t row_offset = 0, right_offset = 0;

    static bool destination_is_legal(const ScalarSlot& slot, const Solution& s, size_t e, int64_t value) {  // swap.rs:103-123
        if (slot.empty_value_source) return value != NONE;
        if (value == NONE) return slot.allows_unassigned;
        return NearbyScalarChangeCursor::value_is_legal(slot, s, e, value);
    }
    // rank_nearby_entities (swap.rs:346-398)
    static std::vector<int64_t> rank(const ScalarSlot& slot, const Solution& s, size_t left, size_t n, size_t max_nearby,
                                     const MoveStreamContext& ctx) {
        if (max_nearby == 0) return {};
        const auto& vals = s.classes[slot.descriptor_index].vars[slot.variable_index];
        const int64_t left_value = vals[left];
        std::vector<RankedNearby> all;
        size_t order = 0;
        auto visit = [&](size_t right) {
            size_t source_order = order++;
            bool allowed = slot.dynamic ? right != left : right > left;  // DynamicDirectional / StaticCanonical
            if (!allowed || right >= n) return;
            const int64_t right_value = vals[right];
            if (left_value == right_value || !destination_is_legal(slot, s, left, right_value) || !destination_is_legal(slot, s, right, left_value))
                return;
            double dist = slot.nearby_entity_distance ? slot.nearby_entity_distance(left, right) : std::nan("");
            all.push_back(RankedNearby{(int64_t)right, dist != dist ? (double)source_order : dist, source_order});
        };
        if (slot.has_nearby_entities) {
            const auto& row = slot.nearby_entities[left];
            for (size_t i = 0; i < row.size() && i < n; ++i) visit((size_t)row[i]);  // the source is visited with limit = entity_count
        } else {
            for (size_t r = 0; r < n; ++r) visit(r);
        }
        std::vector<int64_t> ents = nearby_top_k(std::move(all), max_nearby);
        apply_selection_order(ctx, ents, TARGET_SALT ^ (uint64_t)left ^ slot.identity());
        return ents;
    }
    NearbyScalarSwapCursor(const ScalarSlot& sl, const Solution& solution_in, const MoveStreamContext& ctx, size_t max_nearby) : slot(sl) {
        Solution solution = solution_in;
        size_t n = solution.classes[slot.descriptor_index].n;
        for (size_t off = 0; off < n; ++off) {
            size_t left = nearby_ordered_entity(n, off, ctx, ENTITY_START_SALT, ENTITY_STRIDE_SALT, slot.identity());
            rows.push_back(Row{left, rank(slot, solution, left, n, max_nearby, ctx)});
        }
    }
    bool next(Move& out) override {  // swap.rs:226-246
        for (;;) {
            if (row_offset >= rows.size()) return false;
            Row& row = rows[row_offset];
            if (right_offset < row.rights.size()) {
                out = Move{};
                out.kind = Move::Swap;
                out.descriptor = slot.descriptor_index;
                out.variable = slot.variable_index;
                out.a = row.left;
                out.b = (size_t)row.rights[right_offset++];
                out.allows_unassigned = slot.allows_unassigned;
                return true;
            }
            ++row_offset;
            right_offset = 0;
        }
    }
};

// ---- list leaves -----------------------------------------------------------
using DistanceMeter =
    std::function<double(const Solution&, size_t src_e, size_t src_p, size_t dst_e, size_t dst_p)>;

struct ListSlot {
    size_t descriptor_index = 0;
    // PrecedencePolicy::Explicit of the slot (successors + durations): what the critical-path leaf analyses and what its ruins
    // carry into the recreate; null = the slot declares no precedence hooks
    std::shared_ptr<PrecedenceHooks> precedence;
    // the compiled runtime slot's precedence policy (list_leaf/cursor/slot.rs:191-404, ruin_access.rs:195-217): with hooks declared, every
    // list cursor of the slot drops intra-list candidates that close a cycle through the route graph and the ruin leaf recreates with
    // the hooks.  false = the public selectors, which know nothing of the hooks (only ListPrecedenceMoveSelector takes them).
    bool precedence_policy = false;
    DistanceMeter meter;  // CrossEntityDistanceMeter (selector/nearby_list_change.rs:22-31)
    // Entity order profile.  The compiled runtime leaf orders entities WITHOUT replacement
    // (list_leaf/cursor/slot.rs:468-499); the public selectors re-index WITH replacement
    // (selector/list_support.rs:13-26).  Both are identity under SelectionOrder::Original.
    bool public_selector_entity_order = false;
};

inline void selected_entities(const ListSlot& slot, const Solution& s, const MoveStreamContext& ctx,
                              uint64_t salt, std::vector<size_t>& entities,
                              std::vector<size_t>& route_lens) {
    const EntityClass& c = s.classes[slot.descriptor_index];
    size_t n = c.n;
    entities.resize(n);
    route_lens.resize(n);
    for (size_t off = 0; off < n; ++off) {
        size_t e;
        if (n <= 1)
            e = off;
        else if (slot.public_selector_entity_order)
            e = ctx.selection_index(off, n, salt);
        else
            e = ctx.selection_index_without_replacement(off, n, salt);
        entities[off] = e;
        route_lens[off] = c.lists[e].size();
    }
}

inline Move make_list_move(Move::Kind k, size_t desc, size_t a, size_t ap, size_t b, size_t bp) {
    Move m;
    m.kind = k;
    m.descriptor = desc;
    m.a = a;
    m.a_pos = ap;
    m.b = b;
    m.b_pos = bp;
    return m;
}

// Full list change (selector/list_kernel/change.rs:25-241).
struct ListChangeCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x1157C4A46E000001ULL;
    static constexpr uint64_t SALT_SOURCE = 0x1157C4A46E000002ULL;
    static constexpr uint64_t SALT_INTRA = 0x1157C4A46E000003ULL;
    static constexpr uint64_t SALT_INTER = 0x1157C4A46E000004ULL;
    size_t desc;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t src_idx = 0, src_pos_offset = 0;
    bool stage_intra = true;
    size_t intra_dst_offset = 0, dst_idx = 0, inter_dst_pos_offset = 0;

    ListChangeCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c)
        : desc(slot.descriptor_index), ctx(c) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    void advance_source_position() {
        ++src_pos_offset;
        stage_intra = true;
        intra_dst_offset = 0;
        dst_idx = 0;
        inter_dst_pos_offset = 0;
        while (src_idx < route_lens.size() && src_pos_offset >= route_lens[src_idx]) {
            ++src_idx;
            src_pos_offset = 0;
        }
    }
    bool next(Move& out) override {
        for (;;) {
            if (src_idx >= entities.size()) return false;
            size_t se = entities[src_idx];
            size_t slen = route_lens[src_idx];
            if (slen == 0) {
                ++src_idx;
                continue;
            }
            size_t sp = ctx.selection_index(src_pos_offset, slen, SALT_SOURCE ^ (uint64_t)se ^ (uint64_t)desc);
            if (stage_intra) {
                while (intra_dst_offset <= slen) {
                    size_t dp = ctx.selection_index(intra_dst_offset, slen + 1,
                                                    SALT_INTRA ^ (uint64_t)se ^ (uint64_t)sp);
                    ++intra_dst_offset;
                    if (sp == dp || dp == sp + 1) continue;
                    out = make_list_move(Move::ListChange, desc, se, sp, se, dp);
                    return true;
                }
                stage_intra = false;
                dst_idx = 0;
                inter_dst_pos_offset = 0;
            } else {
                while (dst_idx < entities.size()) {
                    if (dst_idx == src_idx) {
                        ++dst_idx;
                        inter_dst_pos_offset = 0;
                        continue;
                    }
                    size_t de = entities[dst_idx];
                    size_t dlen = route_lens[dst_idx];
                    if (inter_dst_pos_offset <= dlen) {
                        size_t dp = ctx.selection_index(inter_dst_pos_offset, dlen + 1,
                                                        SALT_INTER ^ (uint64_t)se ^ (uint64_t)de ^ (uint64_t)sp);
                        ++inter_dst_pos_offset;
                        out = make_list_move(Move::ListChange, desc, se, sp, de, dp);
                        return true;
                    }
                    ++dst_idx;
                    inter_dst_pos_offset = 0;
                }
                advance_source_position();
            }
        }
    }
};

// Full list swap (selector/list_kernel/swap.rs:25-270).
struct ListSwapCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x11575A0900000001ULL;
    static constexpr uint64_t SALT_FIRST = 0x11575A0900000002ULL;
    static constexpr uint64_t SALT_SECOND = 0x11575A0900000003ULL;
    static constexpr uint64_t SALT_INTER_FIRST = 0x11575A0900000004ULL;
    static constexpr uint64_t SALT_INTER_SECOND = 0x11575A0900000005ULL;
    size_t desc;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t entity_idx = 0;
    bool stage_intra = true;
    size_t first_off = 0, second_off = 0, destination_idx = 1, inter_first_off = 0, inter_second_off = 0;

    ListSwapCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c)
        : desc(slot.descriptor_index), ctx(c) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    void advance_entity() {
        ++entity_idx;
        stage_intra = true;
        first_off = second_off = 0;
        destination_idx = entity_idx + 1;
        inter_first_off = inter_second_off = 0;
    }
    bool next(Move& out) override {
        for (;;) {
            if (entity_idx >= entities.size()) return false;
            size_t fe = entities[entity_idx];
            size_t flen = route_lens[entity_idx];
            if (flen == 0) {
                advance_entity();
                continue;
            }
            if (stage_intra) {
                while (first_off < flen) {
                    size_t fp = ctx.selection_index(first_off, flen, SALT_FIRST ^ (uint64_t)fe ^ (uint64_t)desc);
                    size_t second_count = flen > fp + 1 ? flen - (fp + 1) : 0;
                    if (second_off < second_count) {
                        size_t sp = fp + 1 +
                                    ctx.selection_index(second_off, second_count,
                                                        SALT_SECOND ^ (uint64_t)fe ^ (uint64_t)fp);
                        ++second_off;
                        out = make_list_move(Move::ListSwap, desc, fe, fp, fe, sp);
                        return true;
                    }
                    ++first_off;
                    second_off = 0;
                }
                stage_intra = false;
                destination_idx = entity_idx + 1;
                inter_first_off = inter_second_off = 0;
            } else {
                while (destination_idx < entities.size()) {
                    size_t se = entities[destination_idx];
                    size_t slen = route_lens[destination_idx];
                    if (slen == 0) {
                        ++destination_idx;
                        continue;
                    }
                    while (inter_first_off < flen) {
                        size_t fp = ctx.selection_index(inter_first_off, flen,
                                                        SALT_INTER_FIRST ^ (uint64_t)fe ^ (uint64_t)se);
                        if (inter_second_off < slen) {
                            size_t sp = ctx.selection_index(
                                inter_second_off, slen,
                                SALT_INTER_SECOND ^ (uint64_t)fe ^ (uint64_t)se ^ (uint64_t)fp);
                            ++inter_second_off;
                            out = make_list_move(Move::ListSwap, desc, fe, fp, se, sp);
                            return true;
                        }
                        ++inter_first_off;
                        inter_second_off = 0;
                    }
                    ++destination_idx;
                    inter_first_off = inter_second_off = 0;
                }
                advance_entity();
            }
        }
    }
};

// Contiguous-window permutation leaf (selector/list_kernel/permute.rs:22-205; compiled runtime leaf: list_leaf/cursor/slot.rs:273-296,
// entity salt 0x91D7_9E8A_0000_0001 ^ descriptor): per entity, per start (ordered_index), per window size min..=max that fits,
// every non-identity permutation of the window in ordered_index order of its rank.  No owner restrictions, no precedence graph.
struct ListPermuteCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x91D79E8A00000001ULL, SALT_START = 0x91D79E8A00000002ULL;
    static constexpr uint64_t SALT_SIZE = 0x91D79E8A00000003ULL, SALT_ORDER = 0x91D79E8A00000004ULL;
    size_t desc;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t entity_idx = 0, start_offset = 0, size_offset = 0, permutation_offset = 0;
    bool has_window = false;
    size_t win_start = 0, win_size = 0;
    size_t min_size, max_size;

    ListPermuteCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t min_window_size, size_t max_window_size)
        : desc(slot.descriptor_index), ctx(c), min_size(min_window_size), max_size(max_window_size) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    bool next(Move& out) override {
        for (;;) {
            if (entity_idx >= entities.size()) return false;
            size_t entity = entities[entity_idx], route_len = route_lens[entity_idx];
            auto advance_entity = [&]() {
                ++entity_idx;
                start_offset = size_offset = permutation_offset = 0;
                has_window = false;
            };
            auto advance_start = [&]() {
                ++start_offset;
                size_offset = permutation_offset = 0;
                has_window = false;
            };
            if (route_len < min_size) {
                advance_entity();
                continue;
            }
            if (has_window) {
                size_t count = permute_factorial(win_size) - 1;
                if (permutation_offset < count) {
                    size_t rank = ctx.selection_index(permutation_offset, count,
                                                      SALT_ORDER ^ (uint64_t)entity ^ (uint64_t)win_start ^ (uint64_t)win_size ^ (uint64_t)desc) + 1;
                    ++permutation_offset;
                    out = make_list_move(Move::ListPermute, desc, entity, win_start, entity, win_start + win_size);
                    out.to_value = (int64_t)rank;
                    return true;
                }
                has_window = false;
                ++size_offset;
                permutation_offset = 0;
            }
            if (start_offset >= route_len) {
                advance_entity();
                continue;
            }
            size_t start = ctx.selection_index(start_offset, route_len, SALT_START ^ (uint64_t)entity ^ (uint64_t)desc);
            size_t max_valid = std::min(max_size, route_len - start);
            if (max_valid < min_size) {
                advance_start();
                continue;
            }
            size_t size_count = max_valid - min_size + 1;
            if (size_offset >= size_count) {
                advance_start();
                continue;
            }
            win_size = min_size + ctx.selection_index(size_offset, size_count, SALT_SIZE ^ (uint64_t)entity ^ (uint64_t)start);
            win_start = start;
            has_window = true;
        }
    }
};


// Critical-path precedence leaf (selector/list_kernel/precedence/cursor.rs:22-290): the multi-swaps first, then the multi-block
// ruins, then the blocks in stream order, each block's families through the tiered index; candidates whose routes would close a
// cycle are pruned before they count.  Entities in index order (runtime leaf: probe.rs:286; FromSolutionEntitySelector in the
// reference's tests).
struct ListPrecedenceCursor : Cursor {
    static constexpr uint64_t SALT_BLOCK = 0xC9171EAF5EED0001ULL, SALT_MOVE = 0xC9171EAF5EED0002ULL, SALT_MULTI_RUIN = 0xC9171EAF5EED0003ULL;
    static constexpr uint64_t SALT_MULTI_SWAP = 0xC9171EAF5EED0004ULL;
    static constexpr uint64_t SALT_TIER_ADJACENT = 0xAD1ACE1700000001ULL, SALT_TIER_BOUNDARY = 0xAD1ACE1700000002ULL, SALT_TIER_REST = 0xAD1ACE1700000003ULL;
    size_t desc;
    MoveStreamContext ctx;
    const PrecedenceHooks* hooks;
    CriticalAnalysis an;
    std::vector<AdjacentSwap> critical_swaps, support_swaps;
    size_t block_index = 0, move_index = 0, multi_swap_index = 0, multi_swap_count = 0, multi_ruin_index = 0, multi_ruin_count = 0;

    ListPrecedenceCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c) : desc(slot.descriptor_index), ctx(c), hooks(slot.precedence.get()) {
        if (!hooks) return;  // RuntimeListSlotCursor::Empty (slot.rs:298-303)
        const EntityClass& cls = s.classes[desc];
        std::vector<size_t> entities(cls.n);
        for (size_t e = 0; e < cls.n; ++e) entities[e] = e;
        an = critical_analysis(*hooks, cls.lists, entities);
        critical_swaps = critical_adjacent_swaps(an.blocks);
        support_swaps = support_adjacent_swaps(an.blocks, an.graph);
        multi_swap_count = multi_support_swap_count(critical_swaps, support_swaps);
        multi_ruin_count = multi_critical_ruin_count(an.blocks);
    }
    size_t tiered_index(const CriticalBlock& bl, size_t offset, uint64_t salt) const {  // tiered_precedence_move_index (coordinates.rs:91-130)
        size_t adjacent = bl.adjacent_change_move_count();
        if (offset < adjacent) return ctx.selection_index(offset, adjacent, salt ^ SALT_TIER_ADJACENT);
        size_t boundary = bl.boundary_change_move_count();
        if (offset < adjacent + boundary) return adjacent + ctx.selection_index(offset - adjacent, boundary, salt ^ SALT_TIER_BOUNDARY);
        size_t rest = bl.move_count() - adjacent - boundary;
        return adjacent + boundary + ctx.selection_index(offset - adjacent - boundary, rest, salt ^ SALT_TIER_REST);
    }
    Move ruin_move(const size_t* src, const size_t* idx, size_t count) const {  // precedence_ruin (emission.rs:104-141)
        Move m;
        m.kind = Move::Ruin;
        m.descriptor = desc;
        m.a = m.b = src[0];
        m.a_pos = count;
        m.prec = hooks;
        for (size_t i = 0; i < count; ++i) {
            m.ruin_idx[i] = (uint16_t)idx[i];
            m.ruin_src[i] = (uint16_t)src[i];
            if (src[i] != src[0]) m.ruin_multi = true;
        }
        return m;
    }
    Move emit(const CriticalBlock& bl, size_t index) const {  // push_move (cursor.rs:83-179)
        PrecDecoded p = prec_decode(bl, index);
        const size_t e = bl.entity;
        switch (p.family) {
            case PrecDecoded::Change:
                return make_list_move(Move::ListChange, desc, e, p.c.a, e, p.c.b);
            case PrecDecoded::Swap:
                return make_list_move(Move::ListSwap, desc, e, p.c.a, e, p.c.b);
            case PrecDecoded::Reverse:
                return make_list_move(Move::ListReverse, desc, e, p.c.a, e, p.c.b);
            case PrecDecoded::SublistSwap: {
                Move m = make_list_move(Move::SublistSwap, desc, e, p.c.a, e, p.c.c);
                m.to_value = (int64_t)((p.c.b - p.c.a) | ((p.c.d - p.c.c) << 16));
                return m;
            }
            case PrecDecoded::Ruin: {
                size_t src[8], idx[8];
                for (size_t i = 0; i < p.c.b; ++i) src[i] = e, idx[i] = p.c.a + i;
                return ruin_move(src, idx, p.c.b);
            }
            case PrecDecoded::SublistChange: {
                Move m = make_list_move(Move::SublistChange, desc, e, bl.start + p.c.a, e, p.c.c);
                m.to_value = (int64_t)(bl.start + p.c.a + p.c.b);
                return m;
            }
            case PrecDecoded::Permute: {
                Move m = make_list_move(Move::ListPermute, desc, e, bl.start + p.c.a, e, bl.start + p.c.a + p.c.b);
                m.to_value = (int64_t)p.c.c;
                return m;
            }
        }
        return Move{};
    }
    bool next(Move& out) override {
        for (;;) {
            if (multi_swap_index < multi_swap_count) {
                size_t index = ctx.selection_index(multi_swap_index, multi_swap_count, SALT_MULTI_SWAP ^ (uint64_t)desc);
                ++multi_swap_index;
                auto swaps = multi_support_swaps(critical_swaps, support_swaps, index);
                if (an.graph.multi_intra_list_swaps_introduce_cycle(swaps)) continue;
                Move m;  // emit_multi_swap (emission.rs:280-294)
                m.kind = Move::MultiSwap;
                m.descriptor = desc;
                m.a_pos = swaps.size();
                m.a = m.b = swaps.empty() ? 0 : swaps[0].entity;
                m.require_improvement = true;
                for (size_t i = 0; i < swaps.size() && i < 4; ++i)
                    m.ms_entity[i] = (uint16_t)swaps[i].entity, m.ms_first[i] = (uint16_t)swaps[i].first, m.ms_second[i] = (uint16_t)swaps[i].second;
                out = m;
                return true;
            }
            if (multi_ruin_index < multi_ruin_count) {
                size_t index = ctx.selection_index(multi_ruin_index, multi_ruin_count, SALT_MULTI_RUIN ^ (uint64_t)desc);
                ++multi_ruin_index;
                size_t s[4] = {0, 0, 0, 0};
                multi_critical_ruin_sources(an.blocks, index, s);
                // merged_ruin_sources (move/list_kernel/ruin.rs:34-54): one source when both elements sit in one list, sorted by entity
                size_t src[2] = {s[0], s[2]}, idx[2] = {s[1], s[3]};
                if (src[0] > src[1] || (src[0] == src[1] && idx[0] > idx[1])) std::swap(src[0], src[1]), std::swap(idx[0], idx[1]);
                if (src[0] == src[1] && idx[0] == idx[1])
                    out = ruin_move(src, idx, 1);
                else
                    out = ruin_move(src, idx, 2);
                return true;
            }
            if (block_index >= an.blocks.size()) return false;
            const CriticalBlock& bl = an.blocks[ctx.selection_index(block_index, an.blocks.size(), SALT_BLOCK ^ (uint64_t)desc)];
            if (move_index < bl.move_count()) {
                size_t index = tiered_index(bl, move_index, SALT_MOVE ^ (uint64_t)desc ^ (uint64_t)bl.entity ^ ((uint64_t)bl.start << 16) ^ ((uint64_t)bl.end << 32));
                ++move_index;
                if (prec_move_introduces_route_cycle(bl, index, an.graph)) continue;
                out = emit(bl, index);
                return true;
            }
            ++block_index;
            move_index = 0;
        }
    }
};

// Intra-list reversal / 2-opt (selector/list_kernel/reverse.rs:12-108): per entity with len >= 2,
// start = ordered_index(start_offset, len), end = start + 2 + ordered_index(end_offset, len - start - 1),
// i.e. every range [start, end) of at least two elements.
struct ListReverseCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x11572A0700000001ULL;
    static constexpr uint64_t SALT_START = 0x11572A0700000002ULL;
    static constexpr uint64_t SALT_END = 0x11572A0700000003ULL;
    size_t desc;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t entity_idx = 0, start_offset = 0, end_offset = 0;

    ListReverseCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c)
        : desc(slot.descriptor_index), ctx(c) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    bool next(Move& out) override {
        for (;;) {
            if (entity_idx >= entities.size()) return false;
            size_t entity = entities[entity_idx];
            size_t len = route_lens[entity_idx];
            if (len < 2) {
                ++entity_idx;
                start_offset = end_offset = 0;
                continue;
            }
            while (start_offset < len) {
                size_t start = ctx.selection_index(start_offset, len, SALT_START ^ (uint64_t)entity ^ (uint64_t)desc);
                size_t end_count = len > start + 1 ? len - (start + 1) : 0;
                if (end_offset < end_count) {
                    size_t end = start + 2 + ctx.selection_index(end_offset, end_count, SALT_END ^ (uint64_t)entity ^ (uint64_t)start);
                    ++end_offset;
                    out = make_list_move(Move::ListReverse, desc, entity, start, entity, end);
                    return true;
                }
                ++start_offset;
                end_offset = 0;
            }
            ++entity_idx;
            start_offset = end_offset = 0;
        }
    }
};

// Contiguous sublist relocation / Or-opt (selector/list_kernel/sublist_change.rs:13-266): source
// entity -> segment start -> segment size (min..=max) -> intra destinations 0..=(len - size) except
// the start itself (post-removal coordinates) -> every other entity, positions 0..=len.
struct SublistChangeCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x5B157C4A46E00001ULL, SALT_START = 0x5B157C4A46E00002ULL;
    static constexpr uint64_t SALT_SIZE = 0x5B157C4A46E00003ULL, SALT_INTRA = 0x5B157C4A46E00004ULL;
    static constexpr uint64_t SALT_INTER = 0x5B157C4A46E00005ULL;
    size_t desc, min_size, max_size;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t source_idx = 0, start_offset = 0, size_offset = 0;
    bool stage_intra = true;
    size_t intra_offset = 0, destination_idx = 0, inter_offset = 0;

    SublistChangeCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t mn, size_t mx)
        : desc(slot.descriptor_index), min_size(mn), max_size(mx), ctx(c) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    size_t size_count(size_t len, size_t start) const {  // sublist_change.rs:109-115
        size_t max_valid = std::min(max_size, len > start ? len - start : 0);
        return (max_valid > min_size ? max_valid - min_size : 0) + (max_valid >= min_size ? 1 : 0);
    }
    // (entity, len, start, end, size) of the current segment; size 0 = nothing here
    bool current(size_t& ent, size_t& len, size_t& start, size_t& end, size_t& size) const {
        if (source_idx >= entities.size()) return false;
        ent = entities[source_idx];
        len = route_lens[source_idx];
        start = end = size = 0;
        if (len < min_size) return true;
        start = ctx.selection_index(start_offset, len, SALT_START ^ (uint64_t)ent ^ (uint64_t)desc);
        size_t sc = size_count(len, start);
        if (sc == 0) return true;
        size_t so = ctx.selection_index(size_offset, sc, SALT_SIZE ^ (uint64_t)ent ^ (uint64_t)start);
        size = min_size + so;
        end = start + size;
        return true;
    }
    void advance_segment() {
        size_t ent, len, start, end, size;
        if (!current(ent, len, start, end, size)) return;
        size_t sc = size_count(len, start);
        ++size_offset;
        if (size_offset >= sc) {
            size_offset = 0;
            ++start_offset;
        }
        while (source_idx < route_lens.size() && start_offset >= route_lens[source_idx]) {
            ++source_idx;
            start_offset = size_offset = 0;
        }
        stage_intra = true;
        intra_offset = destination_idx = inter_offset = 0;
    }
    bool next(Move& out) override {
        for (;;) {
            size_t ent, len, start, end, size;
            if (!current(ent, len, start, end, size)) return false;
            if (len < min_size || size == 0) {
                advance_segment();
                continue;
            }
            if (stage_intra) {
                size_t post = len - size;
                while (intra_offset <= post) {
                    size_t dp = ctx.selection_index(intra_offset, post + 1, SALT_INTRA ^ (uint64_t)ent ^ (uint64_t)start);
                    ++intra_offset;
                    if (dp == start) continue;
                    out = make_list_move(Move::SublistChange, desc, ent, start, ent, dp);
                    out.to_value = (int64_t)end;
                    return true;
                }
                stage_intra = false;
                destination_idx = inter_offset = 0;
            } else {
                while (destination_idx < entities.size()) {
                    if (destination_idx == source_idx) {
                        ++destination_idx;
                        continue;
                    }
                    size_t de = entities[destination_idx], dlen = route_lens[destination_idx];
                    if (inter_offset <= dlen) {
                        size_t dp = ctx.selection_index(inter_offset, dlen + 1,
                                                        SALT_INTER ^ (uint64_t)ent ^ (uint64_t)de ^ (uint64_t)start);
                        ++inter_offset;
                        out = make_list_move(Move::SublistChange, desc, ent, start, de, dp);
                        out.to_value = (int64_t)end;
                        return true;
                    }
                    ++destination_idx;
                    inter_offset = 0;
                }
                advance_segment();
            }
        }
    }
};

// Contiguous sublist exchange (selector/list_kernel/sublist_swap.rs:13-330).  A segment cursor walks
// (start, size) of one entity; the move stream pairs every first segment with the segments of the
// same entity that start at or after its end, then with every segment of the later entities.
struct SublistSegmentCursor {
    size_t entity = 0, len = 0, min_size = 1, max_size = 3, desc = 0;
    const MoveStreamContext* ctx = nullptr;
    size_t start_offset = 0, size_offset = 0, size_count = 0, cur_start = 0;
    bool has_start = false;
    bool next(size_t& start, size_t& end) {  // sublist_swap.rs:57-101
        if (len < min_size) return false;
        for (;;) {
            if (has_start) {
                if (size_offset < size_count) {
                    size_t size = min_size + ctx->selection_index(size_offset, size_count,
                                                                  0x5B1575A090000003ULL ^ (uint64_t)entity ^ (uint64_t)cur_start);
                    ++size_offset;
                    start = cur_start;
                    end = cur_start + size;
                    return true;
                }
                has_start = false;
            }
            if (start_offset >= len) return false;
            size_t st = ctx->selection_index(start_offset, len, 0x5B1575A090000002ULL ^ (uint64_t)entity ^ (uint64_t)desc);
            ++start_offset;
            size_t max_valid = std::min(max_size, len - st);
            if (max_valid < min_size) continue;
            has_start = true;
            cur_start = st;
            size_count = max_valid - min_size + 1;
            size_offset = 0;
        }
    }
};

struct SublistSwapCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x5B1575A090000001ULL;
    size_t desc, min_size, max_size;
    MoveStreamContext ctx;
    std::vector<size_t> entities, route_lens;
    size_t first_idx = 0, second_idx = 0;
    SublistSegmentCursor first_segments, second_segments;
    bool have_first_cursor = false, have_first = false, have_second_cursor = false;
    size_t fs = 0, fe = 0;

    SublistSwapCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t mn, size_t mx)
        : desc(slot.descriptor_index), min_size(mn), max_size(mx), ctx(c) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    SublistSegmentCursor segment_cursor(size_t idx) const {
        SublistSegmentCursor sc;
        sc.entity = entities[idx];
        sc.len = route_lens[idx];
        sc.min_size = min_size;
        sc.max_size = max_size;
        sc.desc = desc;
        sc.ctx = &ctx;
        return sc;
    }
    bool next_first() {  // next_first_segment
        for (;;) {
            if (first_idx >= entities.size()) return false;
            if (!have_first_cursor) {
                first_segments = segment_cursor(first_idx);
                have_first_cursor = true;
            }
            if (first_segments.next(fs, fe)) {
                have_first = true;
                second_idx = first_idx;
                have_second_cursor = false;
                return true;
            }
            ++first_idx;
            have_first_cursor = false;
            have_first = false;
            second_idx = first_idx;
            have_second_cursor = false;
        }
    }
    bool next(Move& out) override {
        for (;;) {
            if (first_idx >= entities.size()) return false;
            if (!have_first && !next_first()) return false;
            size_t first_entity = entities[first_idx];
            if (second_idx < first_idx) {
                second_idx = first_idx;
                have_second_cursor = false;
            }
            while (second_idx < entities.size()) {
                size_t second_entity = entities[second_idx];
                if (!have_second_cursor) {
                    second_segments = segment_cursor(second_idx);
                    have_second_cursor = true;
                }
                size_t ss, se;
                while (second_segments.next(ss, se)) {
                    if (first_idx == second_idx) {
                        if (ss < fe) continue;
                        if (fs == ss && fe == se) continue;
                    }
                    out = make_list_move(Move::SublistSwap, desc, first_entity, fs, second_entity, ss);
                    out.to_value = (int64_t)((fe - fs) | ((se - ss) << 16));
                    return true;
                }
                ++second_idx;
                have_second_cursor = false;
            }
            have_first = false;
            second_idx = first_idx;
            have_second_cursor = false;
        }
    }
};

struct NearbyCandidate {
    size_t entity, position;
    double distance;
};

// Stable bounded top-k (selector/nearby_list_support.rs:3-34): equivalent to a stable
// sort by partial_cmp (incomparable == Equal) followed by truncate(max_nearby).
inline void sort_and_limit_nearby_candidates(std::vector<NearbyCandidate>& c, size_t max_nearby) {
    if (max_nearby == 0) {
        c.clear();
        return;
    }
    size_t retained = 0;
    for (size_t read = 0; read < c.size(); ++read) {
        NearbyCandidate cand = c[read];
        // partition_point(|existing| existing.partial_cmp(cand) != Greater)
        size_t lo = 0, hi = retained;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            bool greater = c[mid].distance > cand.distance;  // NaN compares false => "Equal"
            if (!greater)
                lo = mid + 1;
            else
                hi = mid;
        }
        size_t insertion = lo;
        if (insertion >= max_nearby) continue;
        size_t next_retained = retained + 1 < max_nearby ? retained + 1 : max_nearby;
        for (size_t i = next_retained - 1; i > insertion; --i) c[i] = c[i - 1];
        c[insertion] = cand;
        retained = next_retained;
    }
    c.resize(retained);
}

// Nearby list change (selector/list_kernel/nearby_change.rs:17-233).
struct NearbyListChangeCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0xA1EA2B17C4A40001ULL;
    static constexpr uint64_t SALT_SOURCE = 0xA1EA2B17C4A40002ULL;
    size_t desc;
    MoveStreamContext ctx;
    Solution solution;  // the runtime leaf clones the solution on open (slot.rs:238)
    DistanceMeter meter;
    size_t max_nearby;
    std::vector<size_t> entities, route_lens;
    size_t source_idx = 0, source_pos_offset = 0;
    size_t cur_e = 0, cur_p = 0;
    std::vector<NearbyCandidate> candidates;
    std::vector<std::pair<size_t, size_t>> destinations;
    size_t destination_offset = 0;
    uint64_t probes = 0;  // distance-meter calls (generation cost)

    NearbyListChangeCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t k)
        : desc(slot.descriptor_index), ctx(c), solution(s), meter(slot.meter), max_nearby(k) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    bool load_next_source() {
        while (source_idx < entities.size()) {
            size_t se = entities[source_idx];
            size_t slen = route_lens[source_idx];
            if (slen == 0) {
                ++source_idx;
                source_pos_offset = 0;
                continue;
            }
            while (source_pos_offset < slen) {
                size_t sp = ctx.selection_index(source_pos_offset, slen,
                                                SALT_SOURCE ^ (uint64_t)se ^ (uint64_t)desc);
                ++source_pos_offset;
                candidates.clear();
                for (size_t dp = 0; dp <= slen; ++dp) {
                    if (dp == sp || dp == sp + 1) continue;
                    size_t ref = dp < slen - 1 ? dp : slen - 1;  // min(dp, len.saturating_sub(1))
                    double dist = meter(solution, se, sp, se, ref);
                    ++probes;
                    if (std::isfinite(dist)) candidates.push_back({se, dp, dist});
                }
                for (size_t di = 0; di < entities.size(); ++di) {
                    if (di == source_idx) continue;
                    size_t de = entities[di];
                    size_t dlen = route_lens[di];
                    for (size_t dp = 0; dp <= dlen; ++dp) {
                        size_t last = dlen > 0 ? dlen - 1 : 0;
                        size_t ref = dp < last ? dp : last;
                        double dist = meter(solution, se, sp, de, ref);
                        ++probes;
                        if (std::isfinite(dist)) candidates.push_back({de, dp, dist});
                    }
                }
                sort_and_limit_nearby_candidates(candidates, max_nearby);
                if (candidates.empty()) continue;
                cur_e = se;
                cur_p = sp;
                destinations.clear();
                for (auto& cd : candidates) destinations.push_back({cd.entity, cd.position});
                destination_offset = 0;
                return true;
            }
            ++source_idx;
            source_pos_offset = 0;
        }
        return false;
    }
    bool next(Move& out) override {
        if (destination_offset >= destinations.size() && !load_next_source()) return false;
        auto& d = destinations[destination_offset++];
        out = make_list_move(Move::ListChange, desc, cur_e, cur_p, d.first, d.second);
        return true;
    }
};

// Nearby list swap (selector/list_kernel/nearby_swap.rs:17-260).
struct NearbyListSwapCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0xA1EA25A090000001ULL;
    static constexpr uint64_t SALT_SOURCE = 0xA1EA25A090000002ULL;
    size_t desc;
    MoveStreamContext ctx;
    Solution solution;
    DistanceMeter meter;
    size_t max_nearby;
    std::vector<size_t> entities, route_lens;
    size_t source_idx = 0, source_pos_offset = 0;
    size_t cur_e = 0, cur_p = 0;
    std::vector<NearbyCandidate> candidates;
    std::vector<std::pair<size_t, size_t>> destinations;
    size_t destination_offset = 0;
    uint64_t probes = 0;

    NearbyListSwapCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t k)
        : desc(slot.descriptor_index), ctx(c), solution(s), meter(slot.meter), max_nearby(k) {
        selected_entities(slot, s, ctx, SALT_ENTITY ^ (uint64_t)desc, entities, route_lens);
    }
    bool load_next_source() {
        while (source_idx < entities.size()) {
            size_t se = entities[source_idx];
            size_t slen = route_lens[source_idx];
            if (slen == 0) {
                ++source_idx;
                source_pos_offset = 0;
                continue;
            }
            while (source_pos_offset < slen) {
                size_t sp = ctx.selection_index(source_pos_offset, slen,
                                                SALT_SOURCE ^ (uint64_t)se ^ (uint64_t)desc);
                ++source_pos_offset;
                candidates.clear();
                for (size_t dp = sp + 1; dp < slen; ++dp) {
                    double dist = meter(solution, se, sp, se, dp);
                    ++probes;
                    if (std::isfinite(dist)) candidates.push_back({se, dp, dist});
                }
                for (size_t di = 0; di < entities.size(); ++di) {
                    if (di <= source_idx) continue;
                    size_t de = entities[di];
                    size_t dlen = route_lens[di];
                    if (dlen == 0) continue;
                    for (size_t dp = 0; dp < dlen; ++dp) {
                        double dist = meter(solution, se, sp, de, dp);
                        ++probes;
                        if (std::isfinite(dist)) candidates.push_back({de, dp, dist});
                    }
                }
                sort_and_limit_nearby_candidates(candidates, max_nearby);
                if (candidates.empty()) continue;
                cur_e = se;
                cur_p = sp;
                destinations.clear();
                for (auto& cd : candidates) destinations.push_back({cd.entity, cd.position});
                destination_offset = 0;
                return true;
            }
            ++source_idx;
            source_pos_offset = 0;
        }
        return false;
    }
    bool next(Move& out) override {
        if (destination_offset >= destinations.size() && !load_next_source()) return false;
        auto& d = destinations[destination_offset++];
        out = make_list_move(Move::ListSwap, desc, cur_e, cur_p, d.first, d.second);
        return true;
    }
};

// ---- union of leaves (selector/decorator/vec_union.rs:190-365) -------------
enum class UnionOrder { Sequential, RoundRobin, RotatingRoundRobin, Random, StratifiedRandom };

struct UnionScheduler {
    size_t current_cursor = 0;
    UnionOrder order;
    std::vector<bool> exhausted;
    size_t live = 0;
    size_t cursor_offset = 0, cursor_stride = 1;
    MoveStreamContext ctx;
    uint64_t random_draw = 0;
    std::vector<uint64_t> weights;
    std::vector<__int128> weighted_current;
    uint64_t total_live_weight = 0;

    UnionScheduler(size_t n, UnionOrder o, const MoveStreamContext& c, const std::vector<uint64_t>& w)
        : order(o), ctx(c), weights(w) {
        exhausted.resize(n);
        for (size_t i = 0; i < n; ++i) {
            exhausted[i] = weights[i] == 0;
            if (!exhausted[i]) ++live;
            total_live_weight += weights[i];
        }
        if (o == UnionOrder::RotatingRoundRobin || o == UnionOrder::StratifiedRandom)
            cursor_offset = ctx.random_index(n, 0xA11CE5E1EC700001ULL);
        if (o == UnionOrder::StratifiedRandom) cursor_stride = ctx.random_stride(n, 0xA11CE5E1EC700002ULL);
        current_cursor = o == UnionOrder::StratifiedRandom ? 0 : cursor_offset;
        weighted_current.assign(n, 0);
    }
    // next_child(i) pulls from child i; returns false when that child is exhausted.
    template <class F>
    bool next(size_t n, F&& next_child, size_t& which) {
        switch (order) {
            case UnionOrder::Sequential:
                while (current_cursor < n) {
                    if (next_child(current_cursor)) {
                        which = current_cursor;
                        return true;
                    }
                    ++current_cursor;
                }
                return false;
            case UnionOrder::RoundRobin:
            case UnionOrder::RotatingRoundRobin:
                while (live > 0) {
                    size_t i = current_cursor % n;
                    current_cursor = (current_cursor + 1) % n;
                    if (exhausted[i]) continue;
                    if (next_child(i)) {
                        which = i;
                        return true;
                    }
                    exhausted[i] = true;
                    --live;
                }
                return false;
            case UnionOrder::Random:
                while (live > 0) {
                    uint64_t draw = ctx.random_seed(0xA11CE5E1EC701000ULL + random_draw) % total_live_weight;
                    ++random_draw;
                    uint64_t cumulative = 0;
                    size_t pick = n;
                    for (size_t i = 0; i < n; ++i) {
                        if (exhausted[i]) continue;
                        cumulative += weights[i];
                        if (draw < cumulative) {
                            pick = i;
                            break;
                        }
                    }
                    if (next_child(pick)) {
                        which = pick;
                        return true;
                    }
                    exhausted[pick] = true;
                    --live;
                    total_live_weight -= weights[pick];
                }
                return false;
            case UnionOrder::StratifiedRandom:
                while (live > 0) {
                    size_t selected = n;
                    __int128 selected_weight = 0;
                    bool have = false;
                    for (size_t pos = 0; pos < n; ++pos) {
                        size_t i = (cursor_offset + pos * cursor_stride) % n;
                        if (exhausted[i]) continue;
                        weighted_current[i] += (__int128)weights[i];
                        if (!have || weighted_current[i] > selected_weight) {
                            selected = i;
                            selected_weight = weighted_current[i];
                            have = true;
                        }
                    }
                    weighted_current[selected] -= (__int128)total_live_weight;
                    if (next_child(selected)) {
                        which = selected;
                        return true;
                    }
                    exhausted[selected] = true;
                    --live;
                    total_live_weight -= weights[selected];
                }
                return false;
        }
        return false;
    }
};

// ---- k-opt (3-opt) leaves ----------------------------------------------------------------
inline size_t kopt_binomial(size_t n, size_t k) {  // selector/k_opt/iterators.rs:163-178
    if (k > n) return 0;
    if (k == 0 || k == n) return 1;
    k = std::min(k, n - k);
    size_t result = 1;
    for (size_t i = 0; i < k; ++i) result = result * (n - i) / (i + 1);
    return result;
}
inline size_t count_cut_combinations(size_t k, size_t len, size_t min_seg) {  // iterators.rs:98-106
    size_t min_len = (k + 1) * min_seg;
    if (len < min_len) return 0;
    return kopt_binomial(len - min_len + k, k);
}
// rank -> cut positions in lexicographic order (iterators.rs:108-161)
inline bool cut_combination_at(size_t k, size_t len, size_t min_seg, size_t rank, std::vector<size_t>& cuts) {
    if (k == 0 || min_seg == 0 || len < (k + 1) * min_seg) return false;
    size_t choice_count = len - (k + 1) * min_seg + k;
    if (rank >= kopt_binomial(choice_count, k)) return false;
    cuts.clear();
    size_t start = 0;
    for (size_t position = 0; position < k; ++position) {
        size_t remaining = k - position - 1;
        size_t maximum = choice_count - (k - position);
        bool found = false;
        size_t selected = 0;
        for (size_t candidate = start; candidate <= maximum; ++candidate) {
            size_t suffix = kopt_binomial(choice_count - candidate - 1, remaining);
            if (rank < suffix) {
                selected = candidate;
                found = true;
                break;
            }
            rank -= suffix;
        }
        if (!found) return false;
        cuts.push_back(selected + min_seg + position * (min_seg - 1));
        start = selected + 1;
    }
    return true;
}
inline Move make_kopt_move(size_t desc, size_t entity, const size_t cuts[3], size_t pattern) {
    Move m;
    m.kind = Move::KOpt;
    m.descriptor = desc;
    m.a = entity;
    m.a_pos = cuts[0];
    m.b = cuts[1];
    m.b_pos = cuts[2];
    m.to_value = (int64_t)pattern;
    return m;
}

// Full-enumeration k-opt cursor (selector/list_kernel/k_opt/full.rs:12-101): per entity (order without
// replacement), move offset -> selection_index over cut_count * patterns -> (cut rank, pattern).
struct KOptCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x4B0F7E1171000001ULL, SALT_MOVE = 0x4B0F7E1171000002ULL;
    size_t desc, k = 3, min_seg;
    MoveStreamContext ctx;
    std::vector<std::pair<size_t, size_t>> entity_lens;
    size_t entity_offset = 0, move_offset = 0;
    KOptCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t min_segment_len)
        : desc(slot.descriptor_index), min_seg(min_segment_len), ctx(c) {
        const EntityClass& cls = s.classes[desc];
        std::vector<std::pair<size_t, size_t>> canonical;
        for (size_t e = 0; e < cls.n; ++e) canonical.push_back({e, cls.lists[e].size()});
        entity_lens = canonical;  // apply_selection_order_without_replacement (iter.rs:159-173)
        if (!ctx.is_canonical())
            for (size_t off = 0; off < canonical.size(); ++off)
                entity_lens[off] = canonical[ctx.selection_index_without_replacement(off, canonical.size(), SALT_ENTITY ^ (uint64_t)desc)];
    }
    bool next(Move& out) override {
        for (;;) {
            if (entity_offset >= entity_lens.size()) return false;
            size_t entity = entity_lens[entity_offset].first, route_len = entity_lens[entity_offset].second;
            size_t move_count = count_cut_combinations(k, route_len, min_seg) * 7;
            if (move_offset >= move_count) {
                ++entity_offset;
                move_offset = 0;
                continue;
            }
            size_t selected = ctx.selection_index(move_offset, move_count, SALT_MOVE ^ (uint64_t)desc ^ (uint64_t)entity);
            ++move_offset;
            std::vector<size_t> cuts;
            cut_combination_at(k, route_len, min_seg, selected / 7, cuts);
            out = make_kopt_move(desc, entity, cuts.data(), selected % 7);
            return true;
        }
    }
};

// Lazy distance-pruned cut generation (selector/list_kernel/k_opt/nearby_state.rs:22-241), k = 3.
struct NearbyCutState {
    size_t entity, k = 3, len, max_nearby, min_seg;
    std::vector<std::pair<size_t, size_t>> stack;  // (position, index in the level's cache)
    std::vector<std::vector<size_t>> nearby_cache;
    std::vector<size_t> first_positions;
    size_t first_offset = 0;
    MoveStreamContext ctx;
    uint64_t salt;
    bool done = false;
    const Solution* sol;
    const DistanceMeter* meter;

    NearbyCutState(size_t e, size_t len_, size_t min_segment_len, size_t max_nearby_, const MoveStreamContext& c,
                   uint64_t salt_, const Solution* s, const DistanceMeter* m)
        : entity(e), len(len_), max_nearby(max_nearby_), min_seg(min_segment_len), ctx(c), salt(salt_), sol(s), meter(m) {
        if (len < (k + 1) * min_seg) {  // :71-88
            done = true;
            return;
        }
        size_t maximum_first = len - min_seg * k;
        std::vector<size_t> canonical;
        for (size_t p = min_seg; p <= maximum_first; ++p) canonical.push_back(p);
        first_positions = canonical;
        if (!ctx.is_canonical())
            for (size_t off = 0; off < canonical.size(); ++off)
                first_positions[off] = canonical[ctx.selection_index_without_replacement(off, canonical.size(), salt ^ 0x4B0F7E1172EA0001ULL)];
        stack.push_back({first_positions[0], 0});
        nearby_cache.push_back({});
    }
    std::vector<size_t> nearby_positions(size_t origin) const {  // :22-52: stable sort by distance, truncate
        std::vector<std::pair<size_t, double>> pos;
        for (size_t p = 0; p < len; ++p)
            if (p != origin) pos.push_back({p, (*meter)(*sol, entity, origin, entity, p)});
        std::stable_sort(pos.begin(), pos.end(), [](const auto& l, const auto& r) { return l.second < r.second; });
        if (pos.size() > max_nearby) pos.resize(max_nearby);
        std::vector<size_t> out;
        for (auto& pr : pos) out.push_back(pr.first);
        return out;
    }
    bool backtrack() {  // :159-193
        while (!stack.empty()) {
            stack.pop_back();
            nearby_cache.pop_back();
            if (!stack.empty()) {
                size_t cache_index = nearby_cache.size();
                if (cache_index > 0) {
                    const std::vector<size_t>& cache = nearby_cache[cache_index - 1];
                    size_t next_index = stack.back().second + 1;
                    if (next_index < cache.size()) {
                        stack.back().second = next_index;
                        size_t position = stack.back().first, next_position = cache[next_index];
                        if (next_position > position) {
                            stack.back() = {next_position, next_index};
                            return true;
                        }
                    }
                }
            } else {
                first_offset += 1;
                if (first_offset < first_positions.size()) {
                    stack.push_back({first_positions[first_offset], 0});
                    nearby_cache.push_back({});
                    return true;
                }
            }
        }
        return false;
    }
    void extend_stack() {  // :113-157
        while (stack.size() < k && !done) {
            size_t last_position = stack.back().first;
            std::vector<size_t> nearby = nearby_positions(last_position);
            size_t remaining_cuts = k - stack.size();
            size_t minimum_position = last_position + min_seg;
            size_t maximum_position = len - min_seg * remaining_cuts;
            std::vector<size_t> valid;
            for (size_t p : nearby)
                if (p >= minimum_position && p <= maximum_position) valid.push_back(p);
            if (!ctx.is_canonical()) {  // apply_selection_order: WITH replacement under Random (iter.rs:149-157)
                std::vector<size_t> canonical = valid;
                uint64_t s2 = salt ^ 0x4B0F7E1172EA0002ULL ^ ((uint64_t)last_position * 0x9E3779B97F4A7C15ULL) ^ (uint64_t)stack.size();
                for (size_t off = 0; off < valid.size(); ++off) valid[off] = canonical[ctx.selection_index(off, canonical.size(), s2)];
            }
            if (valid.empty()) {
                if (!backtrack()) {
                    done = true;
                    return;
                }
            } else {
                size_t next_position = valid[0];
                nearby_cache.push_back(valid);
                stack.push_back({next_position, 0});
            }
        }
    }
    void advance() {  // :195-219
        if (done || stack.empty()) {
            done = true;
            return;
        }
        const std::vector<size_t>& cache = nearby_cache.back();
        size_t next_index = stack.back().second + 1;
        if (next_index < cache.size()) {
            stack.back() = {cache[next_index], next_index};
            return;
        }
        if (backtrack())
            extend_stack();
        else
            done = true;
    }
    bool next_cuts(size_t out[3]) {  // :221-240
        extend_stack();
        if (done || stack.size() != k) return false;
        for (size_t i = 0; i < k; ++i) out[i] = stack[i].first;
        advance();
        return true;
    }
};

// Distance-pruned k-opt cursor (selector/list_kernel/k_opt/nearby.rs:16-148): entities without
// replacement, per entity the lazy cut stream, per cut set the 7 patterns in selection_index order.
struct NearbyKOptCursor : Cursor {
    static constexpr uint64_t SALT_ENTITY = 0x4B0F7E1172EA0003ULL, SALT_STATE = 0x4B0F7E1172EA0004ULL;
    static constexpr uint64_t SALT_PATTERN = 0x4B0F7E1172EA0005ULL;
    size_t desc, min_seg, max_nearby;
    MoveStreamContext ctx;
    Solution solution;  // the cursor works on a clone taken at open (slot.rs:436)
    DistanceMeter meter;
    std::vector<std::pair<size_t, size_t>> entity_lens;
    size_t entity_offset = 0;
    std::unique_ptr<NearbyCutState> state;
    bool has_pending = false;
    size_t pending[3] = {0, 0, 0}, pending_entity = 0, pattern_offset = 0;

    NearbyKOptCursor(const ListSlot& slot, const Solution& s, const MoveStreamContext& c, size_t min_segment_len, size_t max_nearby_)
        : desc(slot.descriptor_index), min_seg(min_segment_len), max_nearby(max_nearby_), ctx(c), solution(s), meter(slot.meter) {
        const EntityClass& cls = s.classes[desc];
        std::vector<std::pair<size_t, size_t>> canonical;
        for (size_t e = 0; e < cls.n; ++e) canonical.push_back({e, cls.lists[e].size()});
        entity_lens = canonical;
        if (!ctx.is_canonical())
            for (size_t off = 0; off < canonical.size(); ++off)
                entity_lens[off] = canonical[ctx.selection_index_without_replacement(off, canonical.size(), SALT_ENTITY ^ (uint64_t)desc)];
    }
    bool load_next_cut_state() {  // :85-104
        while (entity_offset < entity_lens.size()) {
            size_t entity = entity_lens[entity_offset].first, route_len = entity_lens[entity_offset].second;
            ++entity_offset;
            auto st = std::make_unique<NearbyCutState>(entity, route_len, min_seg, max_nearby, ctx,
                                                       SALT_STATE ^ (uint64_t)desc ^ (uint64_t)entity, &solution, &meter);
            if (!st->done) {
                state = std::move(st);
                return true;
            }
        }
        return false;
    }
    bool next(Move& out) override {  // :112-148
        for (;;) {
            if (has_pending) {
                if (pattern_offset < 7) {
                    uint64_t salt = SALT_PATTERN ^ (uint64_t)desc;
                    for (size_t i = 0; i < 3; ++i)
                        salt ^= ((uint64_t)pending_entity * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)pending[i] * 0xBF58476D1CE4E5B9ULL);
                    size_t pattern = ctx.selection_index(pattern_offset, 7, salt);
                    ++pattern_offset;
                    out = make_kopt_move(desc, pending_entity, pending, pattern);
                    return true;
                }
                has_pending = false;
                pattern_offset = 0;
            }
            if (!state && !load_next_cut_state()) return false;
            size_t cuts[3];
            if (state->next_cuts(cuts)) {
                std::sort(cuts, cuts + 3);
                for (size_t i = 0; i < 3; ++i) pending[i] = cuts[i];
                pending_entity = state->entity;
                has_pending = true;
                pattern_offset = 0;
                continue;
            }
            state.reset();
        }
    }
};

// ---- list ruin leaf (selector/list_kernel/ruin.rs:38-144; runtime pool list_leaf/cursor/probe.rs:218-238) ----------------
// Unrestricted source pool = every non-empty list (optionally no longer than max_source_list_len) frozen at cursor open;
// each pull draws the source list, the ruin count and a partial Fisher-Yates of its positions from the cursor's SmallRng.
// `random_range` is rand's (crate source not under /root/reference): PARITY UNPINNED, see SmallRng in sfo_core.hpp.
struct RuinCursor : Cursor {
    size_t descriptor;
    SmallRng rng;
    std::vector<std::pair<size_t, size_t>> pool;  // (entity, list_len)
    size_t remaining_moves, min_ruin_count, max_ruin_count;
    bool skip_empty_destinations;
    const PrecedenceHooks* hooks = nullptr;  // runtime slot with precedence successors: the moves recreate with them (ruin_access.rs:195-217)
    RuinCursor(const ListSlot& slot, const Solution& s, uint64_t seed, size_t moves_per_step, size_t min_count, size_t max_count,
               size_t max_source_list_len, bool skip_empty)
        : descriptor(slot.descriptor_index),
          rng(SmallRng::seed_from_u64(seed)),
          remaining_moves(moves_per_step),
          min_ruin_count(min_count),
          max_ruin_count(max_count),
          skip_empty_destinations(skip_empty),
          hooks(slot.precedence_policy ? slot.precedence.get() : nullptr) {
        const EntityClass& c = s.classes[descriptor];
        for (size_t e = 0; e < c.n; ++e) {
            size_t len = c.lists[e].size();
            if (len > 0 && (max_source_list_len == 0 || len <= max_source_list_len)) pool.push_back({e, len});
        }
    }
    bool next(Move& out) override {
        if (remaining_moves == 0 || pool.empty()) return false;
        --remaining_moves;
        auto [entity, list_len] = pool[(size_t)rng.random_range(0, pool.size())];
        size_t mn = std::min(min_ruin_count, list_len), mx = std::min(max_ruin_count, list_len);
        size_t ruin_count = mn == mx ? mn : (size_t)rng.random_range_inclusive(mn, mx);  // choose_ruin_count (:78-86)
        std::vector<size_t> indices(list_len);
        for (size_t i = 0; i < list_len; ++i) indices[i] = i;
        for (size_t i = 0; i < ruin_count; ++i) std::swap(indices[i], indices[(size_t)rng.random_range(i, list_len)]);
        indices.resize(ruin_count);
        std::sort(indices.begin(), indices.end());  // single_ruin_source (move/list_kernel/ruin.rs:28-32)
        Move m;
        m.kind = Move::Ruin;
        m.descriptor = descriptor;
        m.a = entity;
        m.b = entity;
        m.a_pos = ruin_count;
        m.allows_unassigned = skip_empty_destinations;
        m.prec = hooks;
        for (size_t i = 0; i < ruin_count && i < 8; ++i) m.ruin_idx[i] = (uint16_t)indices[i];
        out = m;
        return true;
    }
};


// The route-graph filter of a runtime list cursor (with_precedence_route_graph: list_kernel/change.rs:174-183, swap.rs:188-197,
// nearby_change.rs:144-153, nearby_swap.rs:147-156, reverse.rs:98-102, sublist_change.rs:207-216, sublist_swap.rs:288-300,
// permute.rs:128-137): an INTRA-list candidate whose added route edges close a cycle is skipped before it is emitted; the cursor's
// own state does not depend on the verdict, so filtering its output is the same stream.  The graph is the one of the cursor open.
struct RouteGraphFilterCursor : Cursor {
    std::unique_ptr<Cursor> inner;
    PrecedenceRouteGraph graph;
    RouteGraphFilterCursor(std::unique_ptr<Cursor> c, const ListSlot& slot, const Solution& s)
        : inner(std::move(c)), graph(PrecedenceRouteGraph::build(*slot.precedence, s.classes[slot.descriptor_index].lists)) {}
    bool closes_cycle(const Move& m) const {
        if (m.a != m.b && m.kind != Move::ListReverse && m.kind != Move::ListPermute) return false;
        switch (m.kind) {
            case Move::ListChange:
                return graph.intra_list_change_introduces_cycle(m.a, m.a_pos, m.b_pos);
            case Move::ListSwap:
                return graph.intra_list_swap_introduces_cycle(m.a, m.a_pos, m.b_pos);
            case Move::ListReverse:
                return graph.intra_list_reverse_introduces_cycle(m.a, m.a_pos, m.b_pos);
            case Move::SublistChange:
                return graph.intra_sublist_change_introduces_cycle(m.a, m.a_pos, (size_t)m.to_value, m.b_pos);
            case Move::SublistSwap:
                return graph.intra_sublist_swap_introduces_cycle(m.a, m.a_pos, m.a_pos + (size_t)(m.to_value & 0xFFFF), m.b_pos,
                                                                 m.b_pos + (size_t)(m.to_value >> 16));
            case Move::ListPermute:
                return graph.intra_list_permutation_introduces_cycle(m.a, m.a_pos, nth_permutation(m.b_pos - m.a_pos, (size_t)m.to_value));
            default:
                return false;
        }
    }
    bool next(Move& out) override {
        while (inner->next(out))
            if (!closes_cycle(out)) return true;
        return false;
    }
    size_t last_selector() const override { return inner->last_selector(); }
};

struct UnionCursor : Cursor {
    std::vector<std::unique_ptr<Cursor>> children;
    UnionScheduler sched;
    size_t last_child = 0;
    UnionCursor(std::vector<std::unique_ptr<Cursor>> ch, UnionOrder order, const MoveStreamContext& ctx, const std::vector<uint64_t>& weights = {})
        : children(std::move(ch)),
          sched(children.size(), order, ctx, weights.size() == children.size() ? weights : std::vector<uint64_t>(children.size(), 1)) {}
    bool next(Move& out) override {
        return sched.next(
            children.size(), [&](size_t i) { return children[i]->next(out); }, last_child);
    }
    size_t last_selector() const override { return last_child; }
};

}  // namespace sfo
