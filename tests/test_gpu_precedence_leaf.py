"""GPU parity tests (through the C ABI): the critical-path precedence leaf (ListPrecedenceMoveSelector,
heuristic/selector/list_precedence.rs:121-210 over list_kernel/precedence/{analysis,coordinates,support,cursor,emission}.rs -- the first
list leaf of the default policy for slots with precedence hooks, default_local_search/policy/list.rs:24-33,62-93) in the generic N-leaf
engine vs the oracle's cursor (pinned to all twelve cases of heuristic/selector/tests/list_precedence.rs in oracle/test_golden.cpp):
candidate streams with trial scores under Original / Random / Shuffled order -- multi-swaps (score improvement required), two-block
ruins and ruin windows recreated with the precedence hooks, the tiered block families, cyclic candidates pruned -- traced steps with
the committed move, fused multi-replica launches with counters; alone and beside the permute / change / swap leaves; Kahn scratch in
LDS and in HBM."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PREC, PERMUTE = 16384, 8192
BITS = {"precedence": PREC, "permute": PERMUTE, "list_change": 4, "list_swap": 8, "list_reverse": 64, "sublist_change": 128}
COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations", "moves_not_doable"]


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


@pytest.fixture(params=["lds", "hbm", "lds_slow_recreate", "lds_index64", "lds_one_trial", "lds_groups2", "lds_groups16"])
def scratch(request):
    """Kahn scratch in LDS / HBM; slow_recreate: the ruins slide every element through every slot with one evaluation each
    (SF_AMD_PLF_SLOW) instead of pricing all slots from one forward + one backward pass; index64: the multi-swap stream takes its
    64-bit index path (selection index, ring entry, row search) whatever its length (SF_AMD_PLF_FORCE64).  "lds" runs the grouped
    trial evaluator at its default width (8 trials per wavefront at these sizes: replay, route-graph filter and the leaf's generator
    score candidates through position maps, the recreate's rows borrow its LDS); one_trial switches it off (SF_AMD_PREC_GROUPS=0: every
    trial applied, evaluated wave-wide and restored), groups2 / groups16 run 2 trials on 32 lanes each (the recreate's round offsets
    stay in HBM) and 16 trials on 4 lanes each (several pops of a Kahn round per group)."""
    env = {"hbm": ("SF_AMD_PREC_HBM", "1"), "lds_slow_recreate": ("SF_AMD_PLF_SLOW", "1"), "lds_index64": ("SF_AMD_PLF_FORCE64", "1"),
           "lds_one_trial": ("SF_AMD_PREC_GROUPS", "0"), "lds_groups2": ("SF_AMD_PREC_GROUPS", "2"), "lds_groups16": ("SF_AMD_PREC_GROUPS", "16")}
    if request.param in env:
        os.environ[env[request.param][0]] = env[request.param][1]
    yield request.param
    for k, _ in env.values():
        os.environ.pop(k, None)


def _pair(oracle, p, leaves, seed, n_replicas=1, la=5, limit=25, with_owner=True):
    import solverforge_amd as sfa

    d = sfa.build_precedence_shop(p, n_replicas=n_replicas, leaves=leaves, with_owner=with_owner)
    d.configure(sfa.SolverConfig(random_seed=seed, late_acceptance_size=la, accepted_count_limit=limit))
    bits = sum(BITS[x] for x in leaves)

    def mk(s, order=3):
        o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"] if with_owner else None)
        o.configure(leaves=bits, random_seed=s, la_size=la, limit=limit, selection_order=order)
        return o

    return d, mk, bits


@pytest.mark.parametrize("jobs,machines,seed", [(4, 3, 1), (6, 4, 5), (7, 3, 9), (10, 5, 2)])
@pytest.mark.parametrize("leaves", [("precedence",), ("precedence", "permute", "list_change", "list_swap")])
def test_streams_with_trial_scores(oracle, scratch, jobs, machines, seed, leaves):
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(jobs, machines, seed=seed)
    d, mk, bits = _pair(oracle, p, leaves, seed)
    o = mk(seed)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    seen = set()
    for order in (0, 3, 4):
        o = mk(seed, order)
        for si, ss in ((0, 0), (3, 0xBEEF1234)):
            gm, gs, gd = d.open_cursor(si, ss, selection_order=order, cap=1 << 18)
            om = o.enumerate(0, si, ss, order)
            assert len(gm) == len(om) > 0, (order, si)
            assert (_t(gm) == _t(om)).all(), (order, si)
            os_, od = o.evaluate_moves(om)
            assert (gd == od).all() and (gs == os_[:, :2]).all(), (order, si)
            seen |= set(int(k) for k in _t(gm)[:, 0])
            es, ed = d.evaluate_moves(om)  # sf_step_evaluate on host-provided records: every kind, multi-swaps and ruins with hooks included
            assert (ed == od).all() and (es == os_[:, :2]).all(), (order, si)
    assert {2, 3, 4, 8} <= seen  # change, swap, reverse, ruin at least
    if jobs >= 6:
        assert {5, 6, 9, 10} <= seen  # sublist change / swap, permutation, multi-swap


def test_four_level_score_and_no_expected_owner(oracle):
    """The 4-level instantiation (hard penalty on level 1, makespan on level 3), no expected-owner hook."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 3, seed=8)
    R = 2
    d = sfa.build_precedence_shop(p, n_replicas=R, leaves=("precedence", "list_change"), levels=4, hard_levels=2, hard_level=1, makespan_level=3,
                                  with_owner=False)
    d.configure(sfa.SolverConfig(random_seed=6, late_acceptance_size=4, accepted_count_limit=20))

    def mk(seed):
        o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], None, levels=4, hard_levels=2, hard_level=1, soft_level=3)
        o.configure(leaves=PREC | 4, random_seed=seed, la_size=4, limit=20)
        return o

    o = mk(6)
    assert (d.calculate_score()[0] == o.score()[:4]).all()
    gm, gs, gd = d.open_cursor(2, 99, selection_order=3, cap=1 << 18)
    om = o.enumerate(0, 2, 99, 3)
    assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
    os_, od = o.evaluate_moves(om)
    assert (gd == od).all() and (gs == os_[:, :4]).all()
    d.phase_start()
    d.solve_steps(15)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(6 + r)
        o.phase_start()
        o.steps(15)
        assert (scores[r] == o.score()[:4]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r


def test_multi_swaps_and_ruins_through_sf_apply(oracle):
    """Moves the cursor handed out are committed by sf_apply: a multi-swap (three swaps in pairwise different lists, as one move), a
    ruin window and a two-block ruin, both recreated with the precedence hooks."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 4, seed=3)
    d, mk, bits = _pair(oracle, p, ("precedence",), 2)
    o = mk(2)
    d.calculate_score()
    for it, pick in enumerate((10, 8, 10, 8, 8)):
        om = o.enumerate(0, it, 9 + it, 3)
        ms = om[om["kind"] == pick]
        if len(ms) == 0:
            continue
        mv = ms[(7 * it) % len(ms)]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0), it
        assert (d.calculate_score()[0] == o.score()[:2]).all(), it
        assert (d.fresh_score()[0] == o.score()[:2]).all(), it


def test_traced_and_fused_steps(oracle, scratch):
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 4, seed=3)
    R = 3
    for leaves in (("precedence",), ("precedence", "permute", "list_change", "list_swap", "list_reverse")):
        d, mk, bits = _pair(oracle, p, leaves, 11, n_replicas=R)
        o = mk(11)
        d.calculate_score()
        d.phase_start()
        o.phase_start()
        kinds = set()
        for step in range(8):
            gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
            om, os_, of, oap, omv = o.step_traced()
            assert len(gm) == len(om), (leaves, step)
            assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), (leaves, step)
            assert gap == oap
            if gap:
                assert tuple(gmv) == tuple(omv), step
                kinds.add(int(gmv["kind"]))
            assert d.working_lists(0, 0) == o.get_lists(0), step
        d.solve_steps(10)
        d.solve_steps(10)
        scores = d.calculate_score()
        for r in range(R):
            o = mk(11 + r)
            o.phase_start()
            o.steps(28)
            assert (scores[r] == o.score()[:2]).all(), (leaves, r)
            assert d.working_lists(0, r) == o.get_lists(0), (leaves, r)
            gst, ost = d.stats(r), o.stats()
            for c in COUNTERS:
                assert gst[c] == ost[c], (leaves, r, c)
        assert (d.fresh_score() == scores).all()


def test_unscheduled_and_cyclic_states(oracle):
    """A cyclic working state gives the leaf no blocks at all (analysis.rs:61-70); the other leaves still stream."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(5, 3, seed=2)
    # job 0's second operation moves in front of its first one, on that one's machine: a two-node cycle
    seqs = [list(s) for s in p["sequences"]]
    for s in seqs:
        if 1 in s:
            s.remove(1)
    for s in seqs:
        if 0 in s:
            s.insert(s.index(0), 1)
    p["sequences"] = seqs
    d, mk, bits = _pair(oracle, p, ("precedence", "list_swap"), 4)
    o = mk(4)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    gm, gs, gd = d.open_cursor(0, 5, selection_order=3, cap=1 << 18)
    om = o.enumerate(0, 0, 5, 3)
    assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
    assert set(int(k) for k in _t(gm)[:, 0]) == {3}
    d.phase_start()
    o.phase_start()
    d.solve_steps(12)
    o.steps(12)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_validation():
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_cvrp(12, 2, 60, seed=1)
    d = sfa.build_cvrp(p, leaves=("list_change",))
    d.add_precedence_selector(0)  # no precedence constraint on this list class: refused at the first launch
    d.configure(sfa.SolverConfig(random_seed=1))
    d.calculate_score()
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError):
        d.solve_steps(1)
    q = datasets.make_precedence_shop(3, 2, seed=1)
    d = sfa.build_precedence_shop(q, leaves=("precedence",))
    with pytest.raises(sfa.SolverForgeError):
        d.add_precedence_selector(0)  # one such leaf per union
    # (round 4: the 2,048-node limit of the leaf is gone -- test_critical_path_leaf_beyond_2048_nodes runs a 2,400-node shop)


# ---- the runtime slot's precedence policy: route-graph filter on the other list leaves, ruin leaf with the hooks --------------------------
POLICY_LEAVES = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "ruin")
BITS.update({"sublist_swap": 256, "ruin": 1024, "kopt": 512})


def _policy_pair(oracle, p, leaves, seed, n_replicas=1, policy=True, ruin=(2, 5, 4), la=5, limit=25, forager=0, capacity=None):
    import solverforge_amd as sfa

    d = sfa.build_precedence_shop(p, n_replicas=n_replicas, leaves=leaves, ruin=ruin, precedence_policy=policy, element_capacity=capacity)
    d.configure(sfa.SolverConfig(random_seed=seed, late_acceptance_size=la, accepted_count_limit=limit, forager=forager))
    bits = sum(BITS[x] for x in leaves)

    def mk(s, order=3):
        o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
        o.configure(leaves=bits, random_seed=s, la_size=la, limit=limit, selection_order=order, forager=forager)
        o.set_ruin(ruin[0], ruin[1], ruin[2])
        o.set_kopt(1, 0)
        o.set_precedence_policy(policy)
        return o

    return d, mk


def _shuffled(p, seed, same_machine_pairs=True):
    """Sequences in a seeded random order, and a few operations moved onto the machine of their job predecessor (so that
    intra-list moves can close cycles through the job order and cyclic working states occur)."""
    rng = np.random.default_rng(seed)
    seqs = [list(s) for s in p["sequences"]]
    m = p["n_machines"]
    if same_machine_pairs:
        for job in range(0, p["n_jobs"], 2):
            a, b = job * m, job * m + 1
            for s in seqs:
                if b in s:
                    s.remove(b)
            for s in seqs:
                if a in s:
                    s.insert(s.index(a) + 1, b)
    q = dict(p)
    q["sequences"] = seqs
    return q, rng


@pytest.mark.parametrize("leaves", [("list_change", "list_swap"), ("permute", "sublist_change", "sublist_swap", "list_reverse"), POLICY_LEAVES[:-1]])
def test_route_graph_filter_streams(oracle, scratch, leaves):
    """Every intra-list candidate that would close a cycle is gone from the stream (acyclic start), inter-list ones stay."""
    from solverforge_amd import datasets

    p, _ = _shuffled(datasets.make_precedence_shop(6, 4, seed=7), 1)
    for policy in (True, False):
        d, mk = _policy_pair(oracle, p, leaves, 3, policy=policy)
        o = mk(3)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        d.phase_start()
        o.phase_start()
        for order in (0, 3):
            o = mk(3, order)
            o.phase_start()
            gm, gs, gd = d.open_cursor(2, 41, selection_order=order, cap=1 << 18)
            om = o.enumerate(0, 2, 41, order)
            assert len(gm) == len(om) > 0, (policy, order)
            assert (_t(gm) == _t(om)).all(), (policy, order)
            os_, od = o.evaluate_moves(om)
            assert (gd == od).all() and (gs == os_[:, :2]).all(), (policy, order)
        if policy:
            n_policy = len(gm)
        else:
            assert len(gm) > n_policy  # the unfiltered stream is longer


def test_policy_steps_from_acyclic_and_cyclic_starts(oracle, scratch):
    from solverforge_amd import datasets

    base = datasets.make_precedence_shop(6, 4, seed=12)
    for cyclic_start in (False, True):
        p, rng = _shuffled(base, 5)
        if cyclic_start:  # reverse the machines' sequences: job successors in front of their predecessors
            p["sequences"] = [list(reversed(s)) for s in p["sequences"]]
        R = 2
        d, mk = _policy_pair(oracle, p, POLICY_LEAVES, 21, n_replicas=R)
        o = mk(21)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        d.phase_start()
        o.phase_start()
        for step in range(8):
            gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
            om, os_, of, oap, omv = o.step_traced()
            assert len(gm) == len(om), (cyclic_start, step)
            assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), (cyclic_start, step)
            assert gap == oap
            if gap:
                assert tuple(gmv) == tuple(omv), step
        d.solve_steps(12)
        scores = d.calculate_score()
        for r in range(R):
            o = mk(21 + r)
            o.phase_start()
            o.steps(20)
            assert (scores[r] == o.score()[:2]).all(), (cyclic_start, r)
            assert d.working_lists(0, r) == o.get_lists(0), (cyclic_start, r)
            gst, ost = d.stats(r), o.stats()
            for c in COUNTERS:
                assert gst[c] == ost[c], (cyclic_start, r, c)
        assert (d.fresh_score() == scores).all()


def test_complete_default_policy_nine_leaves(oracle):
    """LIST_POLICY_TABLE of a slot with precedence hooks and no distance meter (policy/list.rs:24-33): ListPrecedence + ListPermute, plain
    change, plain swap, sublist change, sublist swap, reverse, unbounded 3-opt, ruin -- nine leaves in one StratifiedRandom union."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 3, seed=14)
    leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
    R = 2
    d, mk = _policy_pair(oracle, p, leaves, 31, n_replicas=R, ruin=(2, 5, 3), limit=40)
    o = mk(31)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    leaf_ids = set()
    for step in range(6):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        leaf_ids |= set(int(f) >> 8 for f in gf)
    assert leaf_ids == set(range(9))
    d.solve_steps(14)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(31 + r)
        o.phase_start()
        o.steps(20)
        assert (scores[r] == o.score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        for c in COUNTERS:
            assert gst[c] == ost[c], (r, c)


def test_ruin_leaf_on_a_precedence_model_without_the_policy(oracle):
    """The public ListRuinMoveSelector knows no hooks: its recreate scores cyclic insertions like any other."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(5, 3, seed=4)
    d, mk = _policy_pair(oracle, p, ("ruin", "list_swap"), 8, policy=False, ruin=(1, 6, 5))
    o = mk(8)
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(6):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om) and (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
    d.solve_steps(10)
    o.steps(10)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_mixed_job_shop_under_the_list_policy(oracle):
    """Two classes (scalar machine choice + machine sequences, BendableScore<2,1>) with the makespan objective: the critical-path leaf,
    permute, the ruin leaf and the slot's precedence policy beside the scalar change / swap leaves.  The flattened not-exists of the
    list class costs every insertion of a recreate round the same, so the recreate is still decided by the precedence constraint."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(8, 4), seed=3)
    p["durations"] = (datasets.stream(5, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
    leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "ruin", "change", "swap")
    bits = PREC | PERMUTE | 4 | 8 | 128 | 1024 | 1 | 2
    ruin = (2, 4, 3)
    for policy in (True, False):
        R = 2
        d = sfa.build_jobshop(p, n_replicas=R, leaves=leaves, makespan=True, ruin=ruin, precedence_policy=policy)
        d.configure(sfa.SolverConfig(random_seed=5, late_acceptance_size=6, accepted_count_limit=30))

        def mk(seed):
            o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, durations=p["durations"])
            o.configure(leaves=bits, random_seed=seed, la_size=6, limit=30)
            o.set_ruin(ruin[0], ruin[1], ruin[2], variable_name="sequence")
            o.set_precedence_policy(policy)
            return o

        o = mk(5)
        assert (d.calculate_score()[0] == o.score()[:3]).all()
        d.phase_start()
        o.phase_start()
        for step in range(8):
            gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
            om, os_, of, oap, omv = o.step_traced()
            assert len(gm) == len(om), (policy, step)
            assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :3]).all(), (policy, step)
            assert gap == oap
            if gap:
                assert tuple(gmv) == tuple(omv), step
        d.solve_steps(12)
        scores = d.calculate_score()
        for r in range(R):
            o = mk(5 + r)
            o.phase_start()
            o.steps(20)
            assert (scores[r] == o.score()[:3]).all(), (policy, r)
            assert d.working_lists(1, r) == o.get_lists(1), (policy, r)
            assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), (policy, r)
        assert (d.fresh_score() == scores).all()


@pytest.mark.parametrize("policy", [False, True])
@pytest.mark.parametrize("start", ["empty", "partial", "cyclic"])
def test_cheapest_insertion_construction_on_a_precedence_model(oracle, policy, start):
    """ListCheapestInsertionPhase on a list class scored by the precedence constraint: every slot priced from one forward + one
    backward pass (or one evaluation per slot when the lists are already cyclic); with the slot's precedence policy the phase has
    the hooks and ranks the elements by their downstream chain (cheapest/kernel.rs:162-229, pinned in oracle/test_golden.cpp)."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(7, 4, seed=6, scheduled=start != "empty")
    n = len(p["durations"])
    rng = np.random.default_rng(3)
    if start != "empty":
        seqs = [list(s) for s in p["sequences"]]
        for _ in range(n // 2):  # take half of the operations out again
            v = int(rng.integers(len(seqs)))
            if seqs[v]:
                seqs[v].pop(int(rng.integers(len(seqs[v]))))
        if start == "cyclic":
            seqs = [list(reversed(s)) for s in seqs]
            a, b = 0, 1  # job 0's second operation in front of its first on one machine: the lists are cyclic before the phase starts
            for s in seqs:
                for x in (a, b):
                    if x in s:
                        s.remove(x)
            seqs[0] = [b, a] + seqs[0]
        p["sequences"] = seqs
    elements = np.arange(n, dtype=np.uint32)[rng.permutation(n)]
    R = 2
    d = sfa.build_precedence_shop(p, n_replicas=R, leaves=("list_change",), precedence_policy=policy)
    d.configure(sfa.SolverConfig(random_seed=1))
    o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
    o.configure(leaves=4, random_seed=1)
    o.set_precedence_policy(policy)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    placed = set(x for s in p["sequences"] for x in s)
    missing = [int(x) for x in elements if int(x) not in placed]
    sc = d.construct_list_cheapest(0, elements)
    o.construct_list_cheapest(missing)
    for r in range(R):
        assert d.working_lists(0, r) == o.get_lists(0), (policy, start, r)
        assert (sc[r] == o.score()[:2]).all()
    assert (d.fresh_score() == sc).all()
    gst, ost = d.stats(0), o.stats()
    for c in ("step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"):
        assert gst[c] == ost[c], c
    d.phase_start()
    o.phase_start()
    d.solve_steps(5)
    o.steps(5)
    assert d.working_lists(0, 0) == o.get_lists(0)


def test_round_robin_construction_on_a_precedence_model(oracle):
    """The round-robin phase scores no trial: unassigned operations are dealt onto the machines in order, the committed score carries the
    precedence constraint."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 4, seed=9, scheduled=False)
    n = len(p["durations"])
    d = sfa.build_precedence_shop(p, n_replicas=2, leaves=("list_change", "list_swap"))
    d.configure(sfa.SolverConfig(random_seed=3))
    o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
    o.configure(leaves=4 | 8, random_seed=3)
    d.calculate_score()
    rng = np.random.default_rng(2)
    els = rng.permutation(n).astype(np.uint32)
    owners = np.where(rng.random(n) < 0.5, p["expected_owner"][els], -1).astype(np.int32)
    sc = d.construct_list_round_robin(0, els, owners=owners)
    o.construct_list_round_robin(els, owners=owners)
    assert d.working_lists(0, 1) == o.get_lists(0)
    assert (sc[0] == o.score()[:2]).all() and (d.fresh_score()[1] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(8)
    o.steps(8)
    assert d.working_lists(0, 0) == o.get_lists(0) and (d.calculate_score()[0] == o.score()[:2]).all()


def test_precedence_ruin_rolls_back_when_nothing_is_safe(oracle):
    """move/tests/list_ruin.rs:443-468 on the device: two nodes that precede each other -- wherever a ruined element would go a cycle
    closes, so the recreate places nothing, the lists come back and the candidate scores like the current state (ruin leaf with the
    slot's hooks, traced steps and sf_step_evaluate of a flagged record)."""
    import solverforge_amd as sfa

    p = {"durations": np.array([1, 1, 2], dtype=np.int64), "successors": [[1], [0], []], "expected_owner": np.array([0, 0, 1], dtype=np.int64),
         "sequences": [[0, 1], [2]], "n_jobs": 1, "n_machines": 2}
    d, mk = _policy_pair(oracle, p, ("ruin", "list_swap"), 5, ruin=(1, 1, 4))
    o = mk(5)
    start = o.score()[:2].copy()
    assert (d.calculate_score()[0] == start).all()
    d.phase_start()
    o.phase_start()
    for step in range(4):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 16)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om) and (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        assert d.working_lists(0, 0) == o.get_lists(0), step
    rec = np.zeros(1, dtype=gm.dtype)
    rec[0] = (8, 0, 1, 0, 0, -2147483648)  # ruin position 0 of list 0 with the hooks
    es, ed = d.evaluate_moves(rec)
    hs, hd = o.evaluate_moves(rec)
    assert (ed == hd).all() and (es == hs[:, :2]).all()


def test_high_occupancy_instantiation(oracle):
    """Launches with more than eight replicas per CU take the PREC kernels built for four workgroups per CU (MODE 2): same results."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(5, 3, seed=4)
    R = 2304
    d, mk = _policy_pair(oracle, p, POLICY_LEAVES, 40, n_replicas=R, ruin=(2, 4, 3))
    d.calculate_score()
    d.phase_start()
    d.solve_steps(6)
    scores = d.calculate_score()
    for r in (0, 1, 777, R - 1):
        o = mk(40 + r)
        o.phase_start()
        o.steps(6)
        assert (scores[r] == o.score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        for c in COUNTERS:
            assert gst[c] == ost[c], (r, c)
    assert (d.fresh_score() == scores).all()


# ---- the reference's DEFAULT forager of a model whose list slot supports precedence moves: FirstLastStepScoreImproving(256) --------------
# compile_default_local_search_components (runtime/compiler/default_local_search/policy.rs:62-71); forager/improving.rs:111-227.  The step
# ends at the first accepted candidate that beats the last step score (it is the pick whatever came before) or at `limit` accepted
# candidates; the critical-path leaf's multi-swaps additionally pass the requires_score_improvement gate before they reach the acceptor.
NINE_LEAVES = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "ruin")
FLSI = 4


def test_configure_default_picks_the_precedence_forager(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(4, 3, seed=2)
    d = sfa.build_precedence_shop(p, leaves=NINE_LEAVES, precedence_policy=True)
    cfg = d.configure_default(random_seed=7)
    assert (cfg.acceptor, cfg.late_acceptance_size, cfg.forager, cfg.accepted_count_limit) == (sfa.Acceptor.LATE_ACCEPTANCE, 400, FLSI, 256)
    # a list slot without precedence hooks: AcceptedCount(256)
    d2 = sfa.build_precedence_shop(p, leaves=("list_change", "list_swap"), precedence_policy=False)
    cfg2 = d2.configure_default(random_seed=7)
    assert (cfg2.acceptor, cfg2.forager, cfg2.accepted_count_limit) == (sfa.Acceptor.LATE_ACCEPTANCE, sfa.Forager.ACCEPTED_COUNT, 256)
    # the default components run: 12 fused steps == oracle under the same components
    o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
    o.configure(leaves=sum(BITS[x] for x in NINE_LEAVES), random_seed=7, la_size=400, limit=256, forager=FLSI)
    o.set_ruin()
    o.set_kopt(1, 0)
    o.set_precedence_policy(True)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    d.solve_steps(12)
    o.steps(12)
    assert (d.calculate_score()[0] == o.score()[:2]).all() and d.working_lists(0, 0) == o.get_lists(0)
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c


@pytest.mark.parametrize("limit", [256, 7, 0])
@pytest.mark.parametrize("cyclic_start", [False, True])
def test_nine_leaf_policy_under_the_default_forager(oracle, scratch, limit, cyclic_start):
    """Traced + fused steps of the complete nine-leaf policy under FirstLastStepScoreImproving: limit 256 = the reference default (the
    early quit decides almost every step), 7 = the accepted-count cut and the improving cut race inside one replay chunk, 0 = no limit
    (the grouped-scalar default's form).  Every pull, trial score, flag (incl. the multi-swaps the improvement gate turns away), the
    committed move and all counters == oracle."""
    from solverforge_amd import datasets

    p, _ = _shuffled(datasets.make_precedence_shop(6, 4, seed=12), 5)
    if cyclic_start:
        p["sequences"] = [list(reversed(s)) for s in p["sequences"]]
    R = 3
    d, mk = _policy_pair(oracle, p, NINE_LEAVES, 51, n_replicas=R, ruin=(2, 5, 3), la=6, limit=limit, forager=FLSI)
    o = mk(51)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    gated = improving = 0
    for step in range(10):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), (step, len(gm), len(om))
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
        gated += int(((gf >> 4) & 1).sum())
        improving += int(len(gm) > 0 and (gf[-1] & 6) == 6)  # the step ended on an accepted candidate that is the pick
    d.solve_steps(15)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(51 + r)
        o.phase_start()
        o.steps(25)
        assert (scores[r] == o.score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
        gst, ost = d.stats(r), o.stats()
        for c in COUNTERS:
            assert gst[c] == ost[c], (r, c)
    assert (d.fresh_score() == scores).all()
    if not cyclic_start:
        assert improving > 0  # the early quit was exercised


def test_default_forager_on_the_mixed_job_shop(oracle):
    """Bendable<2,1> mixed job shop with the makespan objective under LateAcceptance + FirstLastStepScoreImproving(256): list leaves incl.
    the critical-path leaf and the ruin leaf with hooks beside the scalar pair."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(8, 4), seed=3)
    p["durations"] = (datasets.stream(5, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
    leaves = ("precedence", "permute", "list_change", "list_swap", "sublist_change", "ruin", "change", "swap")
    bits = PREC | PERMUTE | 4 | 8 | 128 | 1024 | 1 | 2
    ruin = (2, 4, 3)
    R = 2
    d = sfa.build_jobshop(p, n_replicas=R, leaves=leaves, makespan=True, ruin=ruin, precedence_policy=True)
    cfg = d.configure_default(random_seed=5)
    assert (cfg.forager, cfg.accepted_count_limit) == (FLSI, 256)

    def mk(seed):
        o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, durations=p["durations"])
        o.configure(leaves=bits, random_seed=seed, la_size=400, limit=256, forager=FLSI)
        o.set_ruin(ruin[0], ruin[1], ruin[2], variable_name="sequence")
        o.set_precedence_policy(True)
        return o

    o = mk(5)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    d.phase_start()
    o.phase_start()
    for step in range(8):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :3]).all(), step
        assert gap == oap
    d.solve_steps(17)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(5 + r)
        o.phase_start()
        o.steps(25)
        assert (scores[r] == o.score()[:3]).all(), r
        assert d.working_lists(1, r) == o.get_lists(1), r
        assert (d.working_values(0, 0, replica=r) == o.get_vars(0, 0)).all(), r


def test_element_capacity_below_the_node_count(oracle):
    """the recreate's per-node tables (list predecessors, reachability marks) are rows of max(node_count, element_capacity)
    words -- a list class whose capacity is smaller than the precedence graph (some nodes can never be scheduled) must not spill into the
    next replica's rows."""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(6, 3, seed=9)
    n = len(p["durations"])
    drop = {n - 1, n - 2, n - 4}
    q = dict(p)
    q["sequences"] = [[x for x in s if x not in drop] for s in p["sequences"]]
    R = 3
    d, mk = _policy_pair(oracle, q, POLICY_LEAVES, 17, n_replicas=R, ruin=(2, 5, 4), capacity=n - len(drop))
    o = mk(17)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    d.solve_steps(20)
    scores = d.calculate_score()
    for r in range(R):
        o = mk(17 + r)
        o.phase_start()
        o.steps(20)
        assert (scores[r] == o.score()[:2]).all(), r
        assert d.working_lists(0, r) == o.get_lists(0), r
    assert (d.fresh_score() == scores).all()


def test_critical_path_leaf_beyond_2048_nodes(oracle):
    """Round 4 lifted the leaf's 2,048-node limit (64-bit multi-swap stream).  A 60 x 40 shop (2,400 nodes: Kahn scratch in HBM): traced
    steps of the leaf beside change + swap == oracle, then fused steps with the counters.  (At this size the stream still fits 32 bits --
    the 64-bit plumbing is what the index64 parametrisation of the other tests runs; the reference's own cursor counts its triples with
    three nested loops, so no oracle run exists where the count passes 2^32.)"""
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(60, 40, seed=6)
    assert len(p["durations"]) == 2400
    d, mk, bits = _pair(oracle, p, ("precedence", "list_change", "list_swap"), 4, la=5, limit=12)
    o = mk(4)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(3):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
    d.solve_steps(2)
    o.steps(2)
    assert (d.calculate_score()[0] == o.score()[:2]).all() and d.working_lists(0, 0) == o.get_lists(0)
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c
