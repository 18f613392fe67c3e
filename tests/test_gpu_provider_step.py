"""GPU parity test (through the C ABI): a local-search step whose candidates come from the reference's RuntimeProviderCursor -- the ONE
cursor behind the grouped-scalar leaf and the compound conflict-repair leaf of the default scalar policy
(runtime/compiler/default_local_search/policy/scalar.rs:107-190; runtime/provider_cursor.rs:37-492; its leaf,
runtime/compiler/executor/local_search/leaf.rs:362-402).  The cursor (oracle/provider_cursor.py, pinned to runtime/provider_cursor_tests.rs
by tests/test_provider_cursor.py) schedules the providers with the step's MoveStreamContext -- lazy pulls, max_matches_per_step /
max_repairs_per_match / max_moves_per_step, the rotations, normalisation, per-provider dedup scopes -- and what it stores, in pull order,
goes through sf_step_decide_cursor: scored on the device, gated (a repair leaf's require_hard_improvement = gate bit 0,
phase/localsearch/evaluation.rs:75-113), accepted / foraged / committed, and compared with the oracle's step over the same stream:
trial scores, flags (doable, accepted, committed, RejectedByHardImprovement), the pick, the state, the seven counters.  Components are the
default policy's for a model that declares groups and conflict repairs (sf_provider_declare + sf_solver_configure_default)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations", "moves_not_doable"]


def _providers(pc, g, k):
    """Host-side providers over a colouring (the solution a provider sees is the list of colours, None = unassigned)."""
    off, adj = g["adj_off"], g["adj"]
    n = len(off) - 1
    edges = [(a, int(b)) for a in range(n) for b in adj[off[a]:off[a + 1]] if a < int(b)]

    def conflicts(sol):
        return [(a, b) for a, b in edges if sol[a] is not None and sol[a] == sol[b]]

    def group_exchange(sol, limits):  # callback group: colour exchanges along edges whose ends differ, a rotation, one repeat, one no-op
        out = []
        for a, b in edges[::3]:
            if sol[a] != sol[b]:
                out.append(pc.RawProviderCandidate("exchange", (pc.RawProviderEdit(None, "color", a, sol[b]), pc.RawProviderEdit("Node", "color", b, sol[a]))))
        if out:
            out.insert(2, out[0])  # the same candidate again: one dedup scope per provider
        a, b, c = 0, n // 2, n - 1
        out.append(pc.RawProviderCandidate("rotate", (pc.RawProviderEdit(None, "color", a, sol[b]), pc.RawProviderEdit(None, "color", b, sol[c]),
                                                      pc.RawProviderEdit(None, "color", c, sol[a]))))
        out.append(pc.RawProviderCandidate("noop", (pc.RawProviderEdit(None, "color", 1, sol[1]),)))  # resolved, not doable: never stored
        return out

    def group_static(sol, limits):  # static (typed) group: every fifth node to the next colour
        return [pc.StaticCandidate("next_color", (pc.StaticEdit(0, "color", v, ((sol[v] or 0) + 1) % k),)) for v in range(0, n, 5)]

    def repair_callback(sol, limits):  # conflict repair, declared for "conflict": the lower end of a conflicting edge to every other colour
        assert "conflict" in limits.constraints
        out = []
        for a, b in conflicts(sol)[: limits.max_matches_per_step]:
            out += [pc.RawProviderCandidate("recolor_low", (pc.RawProviderEdit(None, "color", a, c),)) for c in range(k) if c != sol[a]]
        return out

    def repair_callback_unassigned(sol, limits):  # declared for "unassigned" AND "conflict": assign the unassigned nodes
        return [pc.RawProviderCandidate("assign", (pc.RawProviderEdit(None, "color", v, (v * 7) % k),)) for v in range(n) if sol[v] is None]

    def repair_static(sol, limits):  # typed repair for "conflict": both ends of a conflicting edge to two fresh colours at once
        out = []
        for a, b in conflicts(sol):
            out.append(pc.StaticCandidate("recolor_both", (pc.StaticEdit(0, "color", a, (sol[a] + 1) % k), pc.StaticEdit(0, "color", b, (sol[b] + 2) % k))))
        return out

    return group_exchange, group_static, repair_callback, repair_callback_unassigned, repair_static


@pytest.mark.parametrize("order,hard_gate,caps", [(3, True, (16, 32, 256)), (4, True, (3, 2, 7)), (0, False, (16, 3, 40)), (3, True, (2, 32, 256))])
def test_provider_cursor_steps(oracle, order, hard_gate, caps):
    import solverforge_amd as sfa
    from oracle import provider_cursor as pc
    from solverforge_amd import datasets

    rng = np.random.default_rng(17)
    k = 5
    g = datasets.make_graph(70, 230, k, seed=4)
    g["colors"] = rng.integers(-1, k, 70).astype(np.int64)
    d = sfa.build_graph_coloring(g)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    # the model declares a scalar group and a conflict repair: the default components follow (policy.rs:52-77)
    d.declare_provider(1, "colour exchanges")
    d.declare_provider(2, "conflict")
    cfg = d.configure_default(random_seed=3)
    assert (cfg.acceptor, cfg.forager, cfg.accepted_count_limit) == (sfa.Acceptor.DIVERSIFIED_LATE_ACCEPTANCE, sfa.Forager.FIRST_LAST_STEP_SCORE_IMPROVING, 0)
    d.configure(sfa.SolverConfig(acceptor=cfg.acceptor, late_acceptance_size=6, forager=cfg.forager, accepted_count_limit=cfg.accepted_count_limit,
                                 selection_order=order, random_seed=3))
    d.configure_diversified(0.02)
    o.configure(acceptor=1, la_size=6, forager=int(cfg.forager), limit=0, selection_order=order, leaves=3, random_seed=3)
    o.configure_diversified(6, 0.02)
    n_steps = 24
    seeds = rng.integers(0, 2**63, n_steps + 8).astype(np.uint64)  # explicit step seeds: the host-side cursor runs under the step's own context
    d.set_step_seeds(seeds)
    o.set_step_seeds(seeds)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()

    slot = pc.ScalarSlot(pc.SlotId(0, 0, "Node", "color"), k, True, get=lambda sol, row: sol[row], entity_count=lambda sol: len(sol))
    group_exchange, group_static, repair_cb, repair_cb_un, repair_static = _providers(pc, g, k)
    registry = pc.RuntimeProviderRegistry([slot])
    group_plan = pc.CompiledProviderPlan(pc.GroupSchedule(None, caps[2] if caps[2] < 256 else None), [
        pc.ProviderBindingPlan(registry.add_callback(group_exchange), pc.POLICY_CALLBACK_GROUP, (slot.id,), rotation_seed_salt=0x51),
        pc.ProviderBindingPlan(registry.add_static_group(group_static), pc.POLICY_STATIC_GROUP, (slot.id,), rotation_seed_salt=0x52, declared_max_moves_per_step=9),
    ], pc.MOVE_GROUPED)
    repair_plan = pc.CompiledProviderPlan(pc.RepairSchedule(("conflict",), caps[0], caps[1], caps[2], False), [
        pc.ProviderBindingPlan(registry.add_callback(repair_cb_un, constraints=("unassigned", "conflict")), pc.POLICY_CALLBACK_REPAIR, (slot.id,), rotation_seed_salt=0x61),
        pc.ProviderBindingPlan(registry.add_callback(repair_cb, constraints=("conflict",)), pc.POLICY_CALLBACK_REPAIR, (slot.id,), rotation_seed_salt=0x61),
        pc.ProviderBindingPlan(registry.add_static_repair("conflict", repair_static), pc.POLICY_STATIC_REPAIR, (slot.id,), declared_schema_index=0,
                               constraint_rotation_seed_salt=0x71, provider_rotation_seed_salt=0x72, spec_rotation_seed_salt=0x73),
    ], pc.MOVE_COMPOUND_CONFLICT_REPAIR)
    reasons = pc.ProviderReasonArena()
    applied = gated = repairs_applied = 0
    for step in range(n_steps):
        values = o.get_vars(0, 0)
        assert (d.working_values(0, 0) == values).all(), step
        sol = [None if v < 0 else int(v) for v in values]
        ctx = pc.MoveStreamContext(step, int(seeds[step]), order)
        # the union's two provider leaves, sequentially: the grouped-scalar leaf, then the compound conflict-repair leaf
        c1, g1, _ = pc.RuntimeProviderCursor(group_plan, sol, ctx, False).drain_for_step_decide(registry, reasons)
        c2, g2, _ = pc.RuntimeProviderCursor(repair_plan, sol, ctx, hard_gate).drain_for_step_decide(registry, reasons)
        cands, gates = c1 + c2, np.concatenate([g1, g2]).astype(np.int32)
        assert len(c2) <= caps[2] and (len(g2) == 0 or (g2 == (1 if hard_gate else 0)).all())
        gs, gf, gsel = d.step_decide_cursor(cands, gates)
        os_, of, osel = o.step_cursor(cands, gates)
        assert len(gf) == len(of) and (gf == of).all(), step
        assert (gs == os_[:, :2]).all(), step
        assert gsel == osel, step
        applied += gsel >= 0
        repairs_applied += gsel >= len(c1)
        gated += int(((gf & 8) != 0).sum())
        assert (d.calculate_score()[0] == o.score()[:2]).all(), step
    assert applied > 3
    assert gated > 0 or not hard_gate
    assert (d.fresh_score()[0] == o.score()[:2]).all()
    assert (d.best_scores()[0] == o.best_score()[:2]).all()
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c
    assert gst["moves_not_doable"] == 0  # the cursor stores doable moves only (provider_cursor.rs:420-437)
    # the fused engine continues the same search state (history, step index, explicit seed draws)
    d.solve_steps(6)
    o.steps(6)
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert (d.calculate_score()[0] == o.score()[:2]).all()


def test_cursor_step_validation(oracle):
    """sf_step_decide_cursor takes a normalised store: malformed records are the caller's error, a stale (no longer doable) candidate is
    pulled and counted like evaluate_candidate does."""
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.make_graph(20, 40, 4, seed=1)
    g["colors"] = (np.arange(20) % 4).astype(np.int64)
    d = sfa.build_graph_coloring(g)
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.LATE_ACCEPTANCE, late_acceptance_size=3, forager=0, accepted_count_limit=4, random_seed=1))
    o.configure(acceptor=1, la_size=3, forager=0, limit=4, leaves=3, random_seed=1)
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for bad in ([[]], [[(0, 1), (0, 2)]], [[(0, 9)]], [[(25, 1)]]):
        with pytest.raises(sfa.SolverForgeError):
            d.step_decide_cursor(bad)
    cands = [[(0, 0)], [(1, 2)], [(1, 2)], [(2, 0), (3, 1)]]  # a no-op (not doable), a repeat (two provider scopes): all pulled
    gs, gf, gsel = d.step_decide_cursor(cands)
    os_, of, osel = o.step_cursor(cands)
    assert (gf == of).all() and (gs == os_[:, :2]).all() and gsel == osel and gf[0] == 0 and len(gf) == 4
    gst, ost = d.stats(0), o.stats()
    for c in COUNTERS:
        assert gst[c] == ost[c], c
    assert gst["moves_not_doable"] == 1
    assert d.step_decide_cursor([])[2] == -1 and o.step_cursor([])[2] == -1  # an empty store: the step still ends
    assert d.stats(0)["step_count"] == o.stats()["step_count"] == 2
