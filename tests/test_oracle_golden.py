"""CPU tests: the oracle against the reference's own golden vectors."""
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "selector_goldens.json")))


def _tuples(moves):
    return [[int(m["a"]), int(m["a_pos"]), int(m["b"]), int(m["b_pos"])] for m in moves]


def test_constraint_node_goldens(oracle):
    """bi_incr / cross_bi_incr / exists / grouped / director known answers (oracle/test_golden.cpp)."""
    exe = os.path.join(os.path.dirname(oracle._LIB), "test_golden")
    out = subprocess.run([exe], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert len(lines) >= 24
    assert all(l.startswith("ok") for l in lines), out.stdout
    assert out.returncode == 0
    # list_clarke_wright/tests.rs (10), tests/metric_class.rs (3), distance_arithmetic.rs (1 line for its three tests)
    assert sum(l.startswith("ok clarke_wright.") for l in lines) == 14
    # round_robin: compiled_parity.rs known answer + counters + owner / order-key semantics; list_k_opt.rs:325-395 + sum_two
    assert sum(l.startswith("ok round_robin.") for l in lines) == 4
    assert sum(l.startswith("ok list_k_opt.") for l in lines) == 5
    assert sum(l.startswith("ok indexed_presence.") for l in lines) == 3  # stream/collector/tests/collector.rs:333-400


def test_list_change_canonical_order(oracle):
    g = GOLD["list_change_order"]
    m = oracle.Model.list_toy(g["routes"])
    assert _tuples(m.enumerate(oracle.LEAF_LIST_CHANGE)) == g["expected"]


def test_list_swap_canonical_order(oracle):
    g = GOLD["list_swap_order"]
    m = oracle.Model.list_toy(g["routes"])
    assert _tuples(m.enumerate(oracle.LEAF_LIST_SWAP)) == g["expected"]


def test_nearby_change_stable_tie_order(oracle):
    g = GOLD["nearby_change_stable_ties"]
    m = oracle.Model.list_toy(g["routes"], meter=g["meter"])
    m.configure(max_nearby=g["max_nearby"])
    assert _tuples(m.enumerate(oracle.LEAF_NEARBY_LIST_CHANGE))[:3] == g["expected_prefix"]


def test_nearby_swap_pairs(oracle):
    g = GOLD["nearby_swap_pairs"]
    m = oracle.Model.list_toy(g["routes"], meter=g["meter"])
    m.configure(max_nearby=g["max_nearby"])
    assert _tuples(m.enumerate(oracle.LEAF_NEARBY_LIST_SWAP)) == g["expected"]


def test_known_answer_candidate_counts(oracle):
    g = GOLD["candidate_counts"]
    routes = [[v * 1000 + i for i in range(g["visits_per_vehicle"])] for v in range(g["vehicles"])]
    m = oracle.Model.list_toy(routes, meter="position")
    m.configure(max_nearby=g["max_nearby"])
    assert m.enumerate_count(oracle.LEAF_LIST_CHANGE) == g["list_change"]
    assert m.enumerate_count(oracle.LEAF_LIST_SWAP) == g["list_swap"]
    assert m.enumerate_count(oracle.LEAF_NEARBY_LIST_CHANGE) == g["nearby_change"]
    assert m.enumerate_count(oracle.LEAF_NEARBY_LIST_SWAP) == g["nearby_swap"]


def test_bounded_top_k_matches_stable_sort(oracle):
    """nearby_list_support.rs:52-73 property, same LCG inputs (incl. -0.0 ties)."""
    import ctypes as C

    L = oracle.lib()
    for length in range(0, 96):
        state = (0x9E3779B9 ^ length) & 0xFFFFFFFF
        dist = []
        for index in range(length):
            state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
            dist.append(-0.0 if index % 17 == 0 else float(state % 11))
        d = np.array(dist, dtype=np.float64)
        for max_nearby in range(0, length + 3):
            out = np.zeros(max(length, 1), dtype=np.int32)
            n = L.sfo_sort_and_limit(d.ctypes.data_as(C.c_void_p), length, max_nearby, out.ctypes.data_as(C.c_void_p))
            expected = sorted(range(length), key=lambda i: (d[i] + 0.0, i))[:max_nearby]  # stable; -0.0 == 0.0
            assert list(out[:n]) == expected, (length, max_nearby)


def test_stream_context_is_a_permutation(oracle):
    """selection_index_without_replacement visits every row once; stride is coprime (iter.rs:94-147)."""
    from math import gcd

    L = oracle.lib()
    for length in [1, 2, 3, 7, 16, 100, 1000]:
        for seed in [0, 41, 2**63 + 5]:
            idx = [L.sfo_ctx_selection_index_wo(7, seed, oracle.ORDER_RANDOM, o, length, 0xABCDEF) for o in range(length)]
            assert sorted(idx) == list(range(length))
            assert gcd(L.sfo_ctx_random_stride(7, seed, length, 1234), length) == 1
    # canonical orders are the identity (iter.rs:175-180)
    assert [L.sfo_ctx_selection_index(3, 9, oracle.ORDER_ORIGINAL, o, 10, 5) for o in range(10)] == list(range(10))


def test_splitmix64_known_values(oracle):
    """splitmix64 (iter.rs:193-198) against the published reference outputs of the algorithm
    (seed 0 stream: 0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F)."""
    golden = 0x9E3779B97F4A7C15
    assert oracle.splitmix64(0) == 0xE220A8397B1DCDAF
    assert oracle.splitmix64(golden) == 0x6E789E6AA1B965F4
    assert oracle.splitmix64((2 * golden) & (2**64 - 1)) == 0x06C45D188009454F
    from solverforge_amd import datasets

    assert [int(v) for v in datasets.stream(0, 3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_sublist_change_canonical_order_and_count(oracle):
    g = GOLD["sublist_change_order"]
    m = oracle.Model.list_toy(g["routes"])
    m.set_sublist_sizes(g["min"], g["max"])
    mv = m.enumerate(oracle.LEAF_SUBLIST_CHANGE)
    assert [[int(x["a"]), int(x["a_pos"]), int(x["value"]), int(x["b"]), int(x["b_pos"])] for x in mv] == g["expected"]
    c = GOLD["candidate_counts_sublist"]
    routes = [[v * 1000 + i for i in range(c["visits_per_vehicle"])] for v in range(c["vehicles"])]
    m = oracle.Model.list_toy(routes)
    m.set_sublist_sizes(c["min"], c["max"])
    assert m.enumerate_count(oracle.LEAF_SUBLIST_CHANGE) == c["sublist_change"]


def test_sublist_swap_canonical_order_and_count(oracle):
    g = GOLD["sublist_swap_order"]
    m = oracle.Model.list_toy(g["routes"])
    m.set_sublist_sizes(g["min"], g["max"])
    mv = m.enumerate(oracle.LEAF_SUBLIST_SWAP)
    got = [[int(x["a"]), int(x["a_pos"]), int(x["a_pos"]) + (int(x["value"]) & 0xFFFF), int(x["b"]), int(x["b_pos"]),
            int(x["b_pos"]) + (int(x["value"]) >> 16)] for x in mv]
    assert got == g["expected"]
    c = GOLD["candidate_counts_sublist"]
    routes = [[v * 1000 + i for i in range(c["visits_per_vehicle"])] for v in range(c["vehicles"])]
    m = oracle.Model.list_toy(routes)
    m.set_sublist_sizes(c["min"], c["max"])
    assert m.enumerate_count(oracle.LEAF_SUBLIST_SWAP) == c["sublist_swap"]


def test_oracle_incremental_equals_fresh_with_segment_leaves(oracle):
    """FullAssert on the oracle itself (director/tests/benchmarks.rs:116-182 style): do / score / undo of
    reversals, sublist relocations and sublist exchanges leaves the committed score == full recalculation."""
    from solverforge_amd import datasets

    p = datasets.make_cvrp(28, 4, 35, seed=21)
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    bits = oracle.LEAF_LIST_REVERSE | oracle.LEAF_SUBLIST_CHANGE | oracle.LEAF_SUBLIST_SWAP
    o.configure(leaves=bits, random_seed=1, la_size=5, limit=25)
    o.phase_start()
    for _ in range(12):
        o.steps(5)
        assert (o.score() == o.fresh_score()).all()
        assert sorted(c for r in o.get_lists(0) for c in r) == list(range(1, 29))


def test_load_balance_model_incremental_equals_fresh(oracle):
    """Bin fairness (load_balance collector, stream/collector/load_balance.rs): the incremental node's score equals a
    from-scratch evaluation after every committed step, and the closed form round(sqrt(sum x^2 - (sum x)^2 / keys))."""
    import numpy as np
    from solverforge_amd import datasets

    n, k = 50, 6
    r = datasets.stream(5, 2 * n)
    bins = (r[:n] % np.uint64(k + 1)).astype(np.int64) - 1
    sizes = (r[n:] % np.uint64(500)).astype(np.int64) + 1
    o = oracle.Model.balance(k, bins, sizes, w_pair=0, cap=-2)
    o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=2, la_size=4, limit=20)
    o.phase_start()
    for _ in range(30):
        o.steps(1)
        assert (o.score() == o.fresh_score()).all()
        vals = o.get_vars(0, 0)
        loads = np.array([sizes[vals == b].sum() for b in range(k) if (vals == b).any()], dtype=np.int64)
        rad = float(-(int(loads.sum()) ** 2)) / float(len(loads)) + float(int((loads * loads).sum())) if len(loads) > 1 else 0.0
        unfair = int(np.floor(np.sqrt(rad) + 0.5)) if len(loads) > 1 else 0
        assert o.score()[1] == -unfair and o.score()[0] == -int((vals < 0).sum())


BALANCE_FMA_COUNTS = [1, 6, 2, 6, 9, 4, 5, 7, 2, 9]  # variance 33.3 - 5.1 * 5.1: a fused multiply-subtract rounds 5 * sigma to 14


def balance_fma_bins():
    import numpy as np

    return np.concatenate([np.full(c, b, dtype=np.int64) for b, c in enumerate(BALANCE_FMA_COUNTS)])


def test_balance_variance_rounds_the_square_before_subtracting(oracle):
    """BalanceConstraint (constraint/balance.rs:162-173): variance = sum_sq / n - mean * mean with the product rounded first
    (Rust never contracts a * b - c into an FMA).  With these counts and base score 5 the reference gives 13; a contracted
    evaluation gives 14."""
    import numpy as np

    bins = balance_fma_bins()
    o = oracle.Model.balance(len(BALANCE_FMA_COUNTS), bins, np.ones(len(bins), dtype=np.int64), w_pair=0, cap=-3, balance_base=5)
    assert list(o.score()[:2]) == [0, -13] and list(o.fresh_score()[:2]) == [0, -13]


def test_constructed_graph_start_equals_oracle_first_fit(oracle):
    """datasets.construct_graph (the C2 start state) is the oracle's first-fit construction phase."""
    from solverforge_amd import datasets

    for seed, k in ((1, 3), (2, 5), (3, 16)):
        g = datasets.make_graph(120, 700, k, seed=seed)
        o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
        o.construct_first_fit()
        c = datasets.construct_graph(g)
        assert (c["colors"] == o.get_vars(0, 0)).all()
        if k == 3:
            assert (c["colors"] < 0).any()  # three colours cannot colour this graph greedily: those vertices stay unassigned
        if k == 16:
            assert (c["colors"] >= 0).all()


def test_indexed_cpu_baseline_follows_the_dense_faithful_oracle(oracle):
    """The INDEXED CPU baseline (PartnerEqualConstraint: the predicate join indexed by its partner relation, SURVEY 7 "report
    both") is not a reference node: it must reproduce the dense-faithful oracle's trajectory exactly (graph colouring, job shop)."""
    import numpy as np
    from solverforge_amd import datasets

    g = datasets.make_graph(300, 1500, 6, seed=3)
    g["colors"] = (datasets.stream(102, 300) % np.uint64(7)).astype(np.int64) - 1
    a = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    b = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"], indexed=True)
    assert (a.score() == b.score()).all() and (b.score() == b.fresh_score()).all()
    for o in (a, b):
        o.configure(leaves=3, random_seed=5, la_size=7, limit=30)
        o.phase_start()
        o.steps(40)
    assert (a.get_vars(0, 0) == b.get_vars(0, 0)).all() and (a.score() == b.score()).all() and a.stats() == b.stats()
    sa, ca = a.evaluate_each()
    sb, cb = b.evaluate_each()
    assert (sa == sb).all() and (ca == cb).all()
    p = datasets.construct_jobshop(datasets.make_jobshop(12, 4))
    a = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"])
    b = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], indexed=True)
    for o in (a, b):
        o.configure(leaves=4 | 8 | 1 | 2, random_seed=2, la_size=5, limit=20)
        o.phase_start()
        o.steps(40)
    assert a.get_lists(1) == b.get_lists(1) and (a.get_vars(0, 0) == b.get_vars(0, 0)).all() and (a.score() == b.score()).all()


def test_clarke_wright_cvrp_adapter_properties(oracle):
    """The CVRP adapter of the Clarke-Wright oracle (stock savings hooks, one metric class): every customer routed once; the
    structural mode ends in ONE route (capacity stays scoreable, solverforge-cvrp/src/helpers.rs:71-87), the capacity mode in
    capacity-feasible routes; an over-subscribed fleet leaves the lists untouched (kernel.rs:401-415)."""
    from solverforge_amd import datasets

    p = datasets.make_cvrp(200, 20, 55, seed=0)
    p["routes"] = [[] for _ in p["routes"]]
    for mode in (0, 1):
        o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
        committed, st = o.construct_list_clarke_wright(p["customers"], mode)
        lists = o.get_lists(0)
        assert committed and int(st[0]) == 200 * 199 // 2 and int(st[4]) == 0
        assert sorted(c for rt in lists for c in rt) == list(range(1, 201))
        routes = [rt for rt in lists if rt]
        if mode == 0:
            assert len(routes) == 1 and int(st[2]) == 199
        else:
            assert all(sum(int(p["demands"][c]) for c in rt) <= 55 for rt in routes) and o.score()[0] == 0
            assert int(st[2]) == 200 - len(routes)
    p2 = datasets.make_cvrp(300, 20, 55, seed=0)
    p2["routes"] = [[] for _ in p2["routes"]]
    o = oracle.Model.cvrp(p2["capacity"], p2["depot"], p2["demands"], p2["matrix"], p2["customers"], p2["routes"])
    committed, st = o.construct_list_clarke_wright(p2["customers"], 1)
    assert not committed and int(st[4]) > 0 and all(not rt for rt in o.get_lists(0))


def test_construction_pipeline_oracle_properties(oracle):
    """Round robin reproduces datasets.make_cvrp's fill; ListKOpt after Clarke-Wright never worsens the distance level and keeps
    every route's visit set; an over-capacity route takes no reversal under the capacity hook."""
    from solverforge_amd import datasets

    p = datasets.make_cvrp(200, 20, 55, seed=0)
    start = [list(map(int, rt)) for rt in p["routes"]]
    p["routes"] = [[] for _ in p["routes"]]
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_round_robin(p["customers"])
    assert o.get_lists(0) == start
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_clarke_wright(p["customers"], 1)
    before, s0 = o.get_lists(0), o.score()[:2].copy()
    st = o.construct_list_k_opt(2, 1)
    after = o.get_lists(0)
    assert [sorted(rt) for rt in after] == [sorted(rt) for rt in before]
    assert o.score()[0] == s0[0] and o.score()[1] >= s0[1] and int(st[1]) == int(st[2]) > 0
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.construct_list_clarke_wright(p["customers"], 0)  # one over-capacity route
    one = o.get_lists(0)
    st = o.construct_list_k_opt(2, 1)
    assert o.get_lists(0) == one and int(st[1]) == 0 and int(st[0]) == 200 * 199 // 2


def test_precedence_selector_goldens_are_in_the_binary(oracle):
    """heuristic/selector/tests/list_precedence.rs (12 cases), list_ruin.rs:279-303, list_construction/cheapest/tests.rs:244-258."""
    exe = os.path.join(os.path.dirname(oracle._LIB), "test_golden")
    out = subprocess.run([exe], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert sum(l.startswith("ok list_precedence_selector.") for l in lines) == 12
    assert "ok list_ruin.precedence_ruin_recreate_skips_cycle_forming_insertions" in lines
    assert "ok list_cheapest.precedence_downstream_breaks_cheapest_ties" in lines
    # heuristic/move/tests/list_multi_swap.rs:72-134, move/tests/list_ruin.rs:307-338,443-468, phase/tests/foraging.rs:113-134
    for name in ("list_multi_swap.applies_independent_intra_list_swaps_and_undoes", "list_multi_swap.rejects_overlapping_entities",
                 "list_ruin.recreate_restores_multiple_source_entities", "list_ruin.precedence_ruin_restores_original_when_recreate_has_no_safe_position",
                 "gates.score_improvement_required_move_rejects_worse_before_acceptor", "gates.hard_score_delta"):
        assert "ok " + name in lines, name


def _regret_by_deltas(p, order_keys=None, owners=None):
    """Regret insertion restated independently of the oracle's director: leg and capacity deltas on plain Python lists
    (regret/kernel/execute.rs:52-204, evaluation.rs:120-230, mod.rs:19-75).  Returns the lists and every placement."""
    M, dep, dem, cap = p["matrix"], int(p["depot"]), p["demands"], int(p["capacity"])
    lists = [list(rt) for rt in p["routes"]]
    load = [sum(int(dem[x]) for x in rt) for rt in lists]
    placed = {c for rt in lists for c in rt}
    un = [int(c) for c in p["customers"] if int(c) not in placed]
    if order_keys is not None:
        keys = {int(c): int(k) for c, k in zip(p["customers"], order_keys)}
        un.sort(key=lambda c: keys[c])  # stable: source index breaks ties
    own = {int(c): -1 for c in p["customers"]} if owners is None else {int(c): int(w) for c, w in zip(p["customers"], owners)}
    steps = []
    while un:
        choice = None
        for li, x in enumerate(un):
            best, second = None, None
            for e, l in enumerate(lists):
                if own[x] >= 0 and own[x] != e:  # candidate_entities: the fixed owner only (none when the hook names no list)
                    continue
                for pos in range(len(l) + 1):
                    prev = l[pos - 1] if pos > 0 else dep
                    nxt = l[pos] if pos < len(l) else dep
                    dd = int(M[prev, x]) + int(M[x, nxt]) - (int(M[prev, nxt]) if l else 0)
                    dc = max(0, load[e] + int(dem[x]) - cap) - max(0, load[e] - cap)
                    sc = (-dc, -dd)
                    if best is None:
                        best = (sc, e, pos)
                    elif sc > best[0]:
                        second, best = best[0], (sc, e, pos)
                    elif second is None or sc > second:
                        second = sc
            if best is None:
                continue
            forced = second is None
            regret = None if forced else (best[0][0] - second[0], best[0][1] - second[1])
            if choice is None:
                better = True
            else:
                rc = (1 if forced else -1) if forced != choice[0] else (0 if forced else (regret > choice[1]) - (regret < choice[1]))
                better = rc > 0 or (rc == 0 and best[0] > choice[2])
            if best is None:
                continue
            if better:
                choice = (forced, regret, best[0], li, best[1], best[2])
        if choice is None:
            break
        x = un.pop(choice[3])
        lists[choice[4]].insert(choice[5], x)
        load[choice[4]] += int(dem[x])
        steps.append((x, choice[4], choice[5]))
    return lists, steps


@pytest.mark.parametrize("n,v,cap,seed,keep,keys,owners", [
    (24, 4, 40, 1, 0, False, False), (30, 5, 25, 2, 2, False, False), (18, 1, 400, 3, 0, False, False), (26, 4, 30, 4, 0, True, False),
    (22, 3, 12, 5, 1, True, False), (28, 4, 30, 6, 0, False, True), (25, 5, 20, 7, 2, True, True)])
def test_regret_insertion_oracle_equals_an_independent_delta_restatement(oracle, n, v, cap, seed, keep, keys, owners):
    """The oracle scores every trial through its incremental director (all three CVRP constraints); the restatement above prices
    the same trial from two legs and one load.  Same placements, one by one; every customer exactly once; counters."""
    from solverforge_amd import datasets

    p = datasets.make_cvrp(n, v, cap, seed=seed)
    p["routes"] = [rt if i < keep else [] for i, rt in enumerate(p["routes"])]
    ks = np.random.default_rng(seed).integers(0, 3, n).astype(np.int64) if keys else None
    o = oracle.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    placed = {c for rt in p["routes"] for c in rt}
    miss = [i for i, c in enumerate(p["customers"]) if int(c) not in placed]
    ow = None
    if owners:  # a third of the customers have a fixed owner, a few of them one that is no list at all
        rng = np.random.default_rng(seed + 100)
        ow = np.full(n, -1, dtype=np.int64)
        pick = rng.choice(n, n // 3, replace=False)
        ow[pick] = rng.integers(0, v + 2, len(pick))
    o.construct_list_regret([int(p["customers"][i]) for i in miss], None if ks is None else ks[miss], None if ow is None else ow[miss])
    lists, steps = _regret_by_deltas(p, ks, ow)
    assert o.get_lists(0) == lists
    never = set() if ow is None else {int(p["customers"][i]) for i in miss if ow[i] >= v}
    assert sorted(c for rt in lists for c in rt) == sorted(int(c) for c in p["customers"] if int(c) not in never)
    st = o.stats()
    assert st["moves_applied"] == st["step_count"] == len(miss) - len(never)
    if ow is None:
        slots_before = sum(len(rt) for rt in p["routes"]) + v
        trials = sum((len(miss) - k) * (slots_before + k) for k in range(len(miss)))  # round k: every slot of every remaining element
        assert st["score_calculations"] == st["moves_generated"] == st["moves_evaluated"] == trials
    else:
        for c, e, _ in steps:
            w = int(ow[list(p["customers"]).index(c)])
            assert w < 0 or w == e


def test_regret_goldens_are_in_the_binary(oracle):
    """list_construction/regret/tests.rs:305-326 (constant score), a two-list case worked by hand, order keys."""
    exe = os.path.join(os.path.dirname(oracle._LIB), "test_golden")
    lines = subprocess.run([exe], capture_output=True, text=True).stdout.splitlines()
    for name in ("list_regret.constant_score_piles_up_in_reverse_source_order", "list_regret.greatest_regret_then_best_score",
                 "list_regret.order_keys_rank_the_unassigned_elements"):
        assert "ok " + name in lines, name


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_precedence_leaf_properties_on_random_shops(oracle, seed):
    """Size-independent properties of the critical-path leaf and the slot's precedence policy on random job shops: every streamed
    candidate is doable, no candidate but a ruin leaves a cyclic graph behind (hard penalty < node count after any acyclic-start trial),
    the filtered streams of the other leaves are sub-sequences of the unfiltered ones, and a local search under the nine-leaf policy
    never ends worse than it started."""
    import sys

    sys.path.insert(0, os.path.dirname(HERE))
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(3 + seed, 3, seed=seed)
    n = len(p["durations"])
    bits_leaf, bits_other = 16384, 4 | 8 | 64 | 128 | 256 | 8192

    def mk(policy):
        o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
        o.set_precedence_policy(policy)
        return o

    o = mk(False)
    o.configure(leaves=bits_leaf, random_seed=seed, selection_order=3)
    start = o.score()[:2].copy()
    assert start[0] == 0  # the step-major start is acyclic and on the expected machines
    moves = o.enumerate(0, 0, 17 + seed, 3)
    sc, doable = o.evaluate_moves(moves)
    assert len(moves) > 0 and doable.all()
    assert (sc[:, 0] > -n).all()  # nothing cyclic survives the pruning (ruins recreate acyclically with the hooks)
    kinds = set(int(k) for k in moves["kind"])
    assert {2, 3, 4, 8} <= kinds
    plain, filt = mk(False), mk(True)
    dropped = 0
    for leaf in (4, 8, 64, 128, 256, 8192):  # leaf by leaf: inside a union the scheduler interleaves the shorter streams differently
        plain.configure(leaves=leaf, random_seed=seed, selection_order=3)
        filt.configure(leaves=leaf, random_seed=seed, selection_order=3)
        a = [tuple(int(x[k]) for k in ("kind", "a", "a_pos", "b", "b_pos", "value")) for x in plain.enumerate(0, 1, 5, 3)]
        fm = filt.enumerate(0, 1, 5, 3)
        b = [tuple(int(x[k]) for k in ("kind", "a", "a_pos", "b", "b_pos", "value")) for x in fm]
        it = iter(a)
        assert all(any(y == x for y in it) for x in b) and len(b) <= len(a), leaf  # a sub-sequence, in order
        dropped += len(a) - len(b)
        fs, fd = filt.evaluate_moves(fm)
        intra = [i for i, x in enumerate(b) if x[1] == x[3] or x[0] in (4, 9)]
        assert (fs[intra, 0] > -n).all(), leaf  # no surviving intra-list candidate is cyclic
        ps, pd = plain.evaluate_moves(plain.enumerate(0, 1, 5, 3))
        kept = set(b)
        gone = [i for i, x in enumerate(a) if x not in kept]
        assert (ps[gone, 0] <= -n).all(), leaf  # and every dropped one is
    assert dropped > 0
    o = mk(True)
    o.configure(leaves=bits_leaf | bits_other | 1024, random_seed=seed, la_size=20, limit=64)
    o.set_ruin(2, 4, 3)
    o.phase_start()
    o.steps(25)
    best = o.best_score()[:2]
    assert tuple(best) >= tuple(start)
    assert (o.fresh_score()[:2] == o.score()[:2]).all()
