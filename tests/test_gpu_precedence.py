"""GPU parity tests (through the C ABI): ListPrecedenceMakespanConstraint (constraint/list_precedence.rs) on the device --
full scores of acyclic / cyclic / partly scheduled job shops, evaluate_each, trial scores of every list move kind
(sf_step_evaluate and the generic engine's cursor), committed moves, traced and fused steps vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["lds", "lds_one_trial", "lds_groups16", "lds_static_hbm", "hbm", "hbm_full", "hbm_incremental"])
def scratch_mode(request, monkeypatch):
    """Every test runs with the constraint's scratch in the replica's LDS slice -- the grouped trial evaluator at its default width
    (sf_prec_group.h), switched off (one wave-wide evaluation per applied trial), at 16 trials per wavefront, and without the
    workgroup-shared LDS copy of the static graph (which also switches the groups off) -- and forced into HBM (the layout of large
    graphs) three ways: the default lane-per-trial sweep of the list change / swap trials (prec_trial_sweep64), one full evaluation
    per trial, and the opt-in wave-cooperative incremental refresh (prec_trial_inc).  The library reads the variables at every launch."""
    for k in ("SF_AMD_PREC_HBM", "SF_AMD_PREC_INC", "SF_AMD_PREC_NO_SWEEP", "SF_AMD_PREC_GROUPS", "SF_AMD_PREC_STATIC_HBM", "SF_AMD_PREC_STATIC_SLIM",
              "SF_AMD_PREC_LDS_MAX_KB"):
        monkeypatch.delenv(k, raising=False)
    if request.param.startswith("hbm"):
        monkeypatch.setenv("SF_AMD_PREC_HBM", "1")
    if request.param == "hbm_full":
        monkeypatch.setenv("SF_AMD_PREC_NO_SWEEP", "1")
    if request.param == "hbm_incremental":
        monkeypatch.setenv("SF_AMD_PREC_INC", "1")
    if request.param == "lds_one_trial":
        monkeypatch.setenv("SF_AMD_PREC_GROUPS", "0")
    if request.param == "lds_groups16":
        monkeypatch.setenv("SF_AMD_PREC_GROUPS", "16")
    if request.param == "lds_static_hbm":
        monkeypatch.setenv("SF_AMD_PREC_STATIC_HBM", "1")
    return request.param

LEAF_BITS = {"list_change": 4, "list_swap": 8, "list_reverse": 64, "sublist_change": 128, "sublist_swap": 256, "kopt": 512}


def _t(moves):
    return np.stack([moves["kind"], moves["a"], moves["a_pos"], moves["b"], moves["b_pos"], moves["value"]], axis=1)


def _shuffled(p, seed, drop=0):
    """The scheduled shop with every sequence permuted by the documented stream (cycles likely) and `drop` operations removed."""
    from solverforge_amd import datasets

    q = dict(p)
    seqs = [list(s) for s in p["sequences"]]
    r = datasets.stream(seed, 4096)
    k = 0
    for s in seqs:
        for i in range(len(s) - 1, 0, -1):
            j = int(r[k] % np.uint64(i + 1))
            k += 1
            s[i], s[j] = s[j], s[i]
    for _ in range(drop):
        v = int(r[k] % np.uint64(len(seqs)))
        k += 1
        if seqs[v]:
            seqs[v].pop(int(r[k] % np.uint64(len(seqs[v]))))
            k += 1
    q["sequences"] = seqs
    return q


def _mk(oracle, p, leaves=("list_change", "list_swap"), n_replicas=1, with_owner=True, levels=2, hard_levels=1, hard_level=0, mk_level=1):
    import solverforge_amd as sfa

    d = sfa.build_precedence_shop(p, n_replicas=n_replicas, leaves=leaves, with_owner=with_owner, levels=levels, hard_levels=hard_levels,
                                  hard_level=hard_level, makespan_level=mk_level)
    o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"] if with_owner else None,
                                     levels=levels, hard_levels=hard_levels, hard_level=hard_level, soft_level=mk_level)
    bits = 0
    for name in leaves:
        bits |= LEAF_BITS[name]
    return d, o, bits


def test_reference_known_answers(oracle):
    """The two-task plan of the reference's own tests (list_precedence.rs:905-913, 1067-1079, 1156-1164) through the C ABI."""
    base = {"durations": [2, 3], "successors": [[1], []], "expected_owner": [0, 1]}
    for seqs, owner, want in [([[0], [1]], False, [0, -5]), ([[1, 0], []], False, [-2, 0]), ([[0], []], True, [-1, -5])]:
        p = dict(base, sequences=seqs)
        d, o, _ = _mk(oracle, p, with_owner=owner)
        got = d.calculate_score()[0]
        assert got.tolist() == want and (o.score()[:2] == got).all(), (seqs, got)
        assert (d.fresh_score()[0] == got).all()


@pytest.mark.parametrize("jobs,machines,seed,drop,owner", [(6, 4, 1, 0, True), (10, 5, 2, 3, True), (12, 6, 3, 0, False), (20, 10, 4, 7, True)])
def test_full_scores_and_evaluate_each(oracle, jobs, machines, seed, drop, owner):
    from solverforge_amd import datasets

    p0 = datasets.make_precedence_shop(jobs, machines, seed=seed)
    for p in (p0, _shuffled(p0, seed, drop), datasets.make_precedence_shop(jobs, machines, seed=seed, scheduled=False)):
        d, o, _ = _mk(oracle, p, with_owner=owner)
        got = d.calculate_score()[0]
        assert (got == o.score()[:2]).all(), (got, o.score())
        assert (d.fresh_score()[0] == got).all()
        gs, gc = d.evaluate_each()
        os_, oc = o.evaluate_each()
        assert (gs == os_[:, :2]).all() and (gc == oc).all()
    d, o, _ = _mk(oracle, p0, with_owner=owner)
    assert d.calculate_score()[0][0] == 0 and d.calculate_score()[0][1] < 0  # the step-major schedule is feasible


def test_fan_out_graph_beyond_the_node_record(oracle):
    """A general precedence graph: one hub with 260 fixed successors (the Kahn rounds' node record holds the first successor and an
    out-degree that saturates at 255; the rest come from the CSR), nodes with two and three successors, a chain; most nodes in no
    list.  Full score and the trial score of every list move kind == oracle, in every scratch mode of the fixture."""
    n = 272
    succ = [[] for _ in range(n)]
    succ[0] = list(range(1, 261))
    succ[3] = [261, 262]
    succ[7] = [262, 263, 264]
    for v in range(264, 271):
        succ[v] = [v + 1]
    dur = [int(1 + (v * 7) % 9) for v in range(n)]
    owner = [v % 3 for v in range(n)]
    for seqs in ([[0, 3, 261, 10, 264, 265, 40], [7, 262, 5, 263, 266, 100], [2, 267, 268, 200, 1, 271]],
                 [[3, 0, 261, 264, 10], [262, 7, 263, 5, 270, 269], [268, 267, 2, 1]]):  # the second one is cyclic
        p = {"durations": dur, "successors": succ, "expected_owner": owner, "sequences": seqs}
        leaves = ("list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt")
        d, o, bits = _mk(oracle, p, leaves=leaves)
        o.set_kopt(1, 0)
        assert (d.calculate_score()[0] == o.score()[:2]).all() and (d.fresh_score()[0] == o.score()[:2]).all()
        o.configure(leaves=bits, selection_order=0)
        gm, gs, gd = d.open_cursor(2, 31, selection_order=0, cap=1 << 18)
        om = o.enumerate(0, 2, 31, 0)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
    # a short fused search from the acyclic plan keeps replica 0 on the oracle
    p = {"durations": dur, "successors": succ, "expected_owner": owner, "sequences": [[0, 3, 261, 10, 264, 265, 40], [7, 262, 5, 263, 266, 100], [2, 267, 268, 200, 1, 271]]}
    d, o, bits = _mk(oracle, p, leaves=("list_change", "list_swap", "list_reverse"))
    import solverforge_amd as sfa
    d.calculate_score()
    d.configure(sfa.SolverConfig(random_seed=3, late_acceptance_size=5, accepted_count_limit=20))
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=20)
    d.phase_start()
    o.phase_start()
    d.solve_steps(25)
    o.steps(25)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()


@pytest.mark.parametrize("state", ["scheduled", "shuffled"])
def test_trial_scores_every_list_move_kind(oracle, state):
    from solverforge_amd import datasets

    p = datasets.make_precedence_shop(7, 4, seed=5)
    if state == "shuffled":
        p = _shuffled(p, 9, 2)
    leaves = ("list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt")
    d, o, bits = _mk(oracle, p, leaves=leaves)
    o.set_kopt(1, 0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    for order in (0, 3):
        o.configure(leaves=bits, selection_order=order)
        gm, gs, gd = d.open_cursor(2, 31, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 2, 31, order)
        assert len(gm) == len(om) > 0
        assert (_t(gm) == _t(om)).all()
        os_, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gs == os_[:, :2]).all()
        es, ed = d.evaluate_moves(om)  # sf_step_evaluate on host-provided records
        assert (ed == od).all() and (es == os_[:, :2]).all()
    kinds = set(int(k) for k in _t(gm)[:, 0])
    assert kinds == {2, 3, 4, 5, 6, 7}


def test_apply_traced_and_fused_steps(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = _shuffled(datasets.make_precedence_shop(8, 4, seed=7), 3, 1)
    leaves = ("list_change", "list_swap", "sublist_change", "list_reverse")
    d, o, bits = _mk(oracle, p, leaves=leaves)
    d.calculate_score()
    o.configure(leaves=bits, random_seed=4, la_size=6, limit=24)
    d.configure(sfa.SolverConfig(random_seed=4, late_acceptance_size=6, accepted_count_limit=24))
    rng = np.random.default_rng(2)
    for it in range(6):  # committed moves through sf_apply
        mv = o.enumerate(0, it, 50 + it, 3)
        sc, do = o.evaluate_moves(mv)
        mv = mv[do != 0]
        mv = mv[rng.integers(len(mv))]
        o.apply_move(mv)
        d.apply_move(mv)
        assert d.working_lists(0, 0) == o.get_lists(0), it
        assert (d.calculate_score()[0] == o.score()[:2]).all(), it
        assert (d.fresh_score()[0] == o.score()[:2]).all(), it
    d.phase_start()
    o.phase_start()
    for step in range(20):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gs == os_[:, :2]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
        assert d.working_lists(0, 0) == o.get_lists(0), step
    d.solve_steps(60)
    o.steps(60)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


@pytest.mark.parametrize("jobs,machines,setting", [(30, 20, "default"), (30, 20, "static_in_hbm"), (30, 20, "lds_cap_36"), (3, 70, "default"), (40, 70, "default")])
def test_larger_shops_setup_paths(oracle, monkeypatch, jobs, machines, setting):
    """Round 5: the wave-wide evaluation's set-up walks the lists chunk by chunk with the list offsets held one per lane (fewer than 64 lists) or
    read from the lists' offset table (70 lists here), keeps 16-bit queue / successor arrays in LDS, and reads node records, fixed in-degrees and
    owners from a slim workgroup-shared LDS copy when the full static copy does not fit (600 nodes and more; SF_AMD_PREC_STATIC_SLIM=0 leaves
    them in HBM).  2,800 nodes (40 x 70) put the scratch beyond the round-4 LDS limit: it stays in LDS as long as one replica per CU fits.
    Full scores of scheduled / shuffled / partly assigned states, then fused steps == oracle."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    if setting == "static_in_hbm":
        monkeypatch.setenv("SF_AMD_PREC_STATIC_SLIM", "0")
    if setting == "lds_cap_36":
        monkeypatch.setenv("SF_AMD_PREC_LDS_MAX_KB", "36")
    p0 = datasets.make_precedence_shop(jobs, machines, seed=9)
    for p in (p0, _shuffled(p0, 5, 3)):
        d, o, _ = _mk(oracle, p)
        assert (d.calculate_score()[0] == o.score()[:2]).all()
        assert (d.fresh_score()[0] == o.score()[:2]).all()
    leaves = ("list_change", "list_swap", "sublist_change", "list_reverse")
    d, o, bits = _mk(oracle, p0, leaves=leaves)
    d.calculate_score()
    o.configure(leaves=bits, random_seed=6, la_size=5, limit=12)
    d.configure(sfa.SolverConfig(random_seed=6, late_acceptance_size=5, accepted_count_limit=12))
    d.phase_start()
    o.phase_start()
    steps = 4 if jobs * machines > 2000 else 10
    d.solve_steps(steps)
    o.steps(steps)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for k in ["step_count", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations"]:
        assert gst[k] == ost[k], k


def test_multi_replica_search_leaves_the_cycle_and_improves(oracle):
    """8 replicas, distinct seeds: every replica == its own oracle run; starting cyclic (hard = -n), the search reaches a feasible
    schedule and a shorter makespan than the step-major start."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p0 = datasets.make_precedence_shop(6, 3, seed=11)
    p = _shuffled(p0, 5)
    R = 8
    d, _, bits = _mk(oracle, p, n_replicas=R)
    d.calculate_score()
    d.configure(sfa.SolverConfig(random_seed=100, late_acceptance_size=20, accepted_count_limit=32))
    d.phase_start()
    d.solve_steps(150)
    got = d.calculate_score()
    best = d.best_scores()
    for r in (0, 3, 7):
        o = oracle.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
        o.configure(leaves=bits, random_seed=100 + r, la_size=20, limit=32)
        o.phase_start()
        o.steps(150)
        assert d.working_lists(0, r) == o.get_lists(0), r
        assert (got[r] == o.score()[:2]).all(), r
        assert (best[r] == o.best_score()[:2]).all(), r
    assert (d.fresh_score() == got).all()
    assert best[:, 0].max() == 0


def test_three_levels_and_validation(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = _shuffled(datasets.make_precedence_shop(5, 3, seed=2), 8)
    d, o, bits = _mk(oracle, p, levels=3, hard_levels=2, hard_level=1, mk_level=2)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    o.configure(leaves=bits, random_seed=1, la_size=4, limit=16)
    d.configure(sfa.SolverConfig(random_seed=1, late_acceptance_size=4, accepted_count_limit=16))
    d.phase_start()
    o.phase_start()
    d.solve_steps(30)
    o.steps(30)
    assert d.working_lists(0, 0) == o.get_lists(0)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    # limits: duplicate list items, element ids >= node_count, equal levels
    bad = dict(p, sequences=[[0, 1], [1], []])
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_precedence_shop(bad).calculate_score()
    with pytest.raises(sfa.SolverForgeError):
        sfa.build_precedence_shop(p, hard_level=1, makespan_level=1)
    d2 = sfa.GpuScoreDirector(score_levels=2, hard_levels=1, n_replicas=1)
    d2.add_entity_class(0, 2)
    d2.add_list_variable(0, [[0], [5]], element_capacity=8, element_id_bound=8)
    d2.add_list_precedence(0, [1, 1], [[1], []])
    with pytest.raises(sfa.SolverForgeError):
        d2.calculate_score()


def test_mixed_jobshop_with_makespan(oracle):
    """The mixed job shop (scalar machine_idx + machine sequences, BendableScore<2,1>) with the makespan objective added: list and
    scalar leaves in one union, the precedence constraint on the list class of a two-class model."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(9, 4), seed=3)
    p["durations"] = (datasets.stream(5, p["n_ops"]) % np.uint64(9)).astype(np.int64) + 1
    p["sequences"][2] = p["sequences"][2][::-1]  # one machine against the job order: cycles
    d = sfa.build_jobshop(p, makespan=True)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True, durations=p["durations"])
    bits = 4 | 8 | 1 | 2
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    gs, gc = d.evaluate_each()
    os_, oc = o.evaluate_each()
    assert (gs == os_[:, :3]).all() and (gc == oc).all()
    for order in (0, 3):
        o.configure(leaves=bits, selection_order=order)
        gm, gsc, gd = d.open_cursor(1, 5, selection_order=order, cap=1 << 18)
        om = o.enumerate(0, 1, 5, order)
        assert len(gm) == len(om) > 0 and (_t(gm) == _t(om)).all()
        osc, od = o.evaluate_moves(om)
        assert (gd == od).all() and (gsc == osc[:, :3]).all()
        es, ed = d.evaluate_moves(om)
        assert (ed == od).all() and (es == osc[:, :3]).all()
    o.configure(leaves=bits, random_seed=2, la_size=8, limit=32)
    d.configure(sfa.SolverConfig(random_seed=2, late_acceptance_size=8, accepted_count_limit=32))
    d.phase_start()
    o.phase_start()
    kinds = set()
    for step in range(15):
        gm, gsc, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, osc, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm) == _t(om)).all() and (gf == of).all() and (gsc == osc[:, :3]).all(), step
        assert gap == oap
        if gap:
            assert tuple(gmv) == tuple(omv), step
            kinds.add(int(gmv["kind"]))
    d.solve_steps(80)
    o.steps(80)
    assert d.working_lists(1, 0) == o.get_lists(1)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.fresh_score()[0] == o.score()[:3]).all()
