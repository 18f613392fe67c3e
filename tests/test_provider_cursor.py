"""Oracle-side provider cursor (oracle/provider_cursor.py) against the reference's own tests for it
(crates/solverforge-solver/src/runtime/provider_cursor_tests.rs:220-455: the five scenarios below keep their names, fixtures and
assertions) plus the scheduling rules the file's comments state (provider_cursor.rs:107-400).  The step-seeded selection order is pinned to
the C++ oracle's MoveStreamContext (oracle/sfo_core.hpp, itself pinned to iter.rs).  CPU only: the cursor touches no score."""
import numpy as np
import pytest

from oracle import provider_cursor as pc


# ---- the fixture of provider_cursor_tests.rs:25-135: one entity, one scalar variable "worker" with values 0..2 -------------------------
def _slot(n_values=3, allows_unassigned=False, descriptor_index=0, variable_index=0, entity_class="Row", variable_name="worker"):
    return pc.ScalarSlot(pc.SlotId(descriptor_index, variable_index, entity_class, variable_name), n_values, allows_unassigned,
                         get=lambda sol, row: sol[row], entity_count=lambda sol: len(sol))


class CountingGroupProvider:
    def __init__(self, values=(1, 2), reason="candidate"):
        self.pulls, self.values, self.reason, self.limits = 0, values, reason, []

    def __call__(self, solution, limits):
        self.pulls += 1
        self.limits.append(limits)
        return [pc.RawProviderCandidate(self.reason, (pc.RawProviderEdit(None, "worker", 0, v),)) for v in self.values]


def _callback_group_plan(slot, handle, requested_max_moves_per_step=0, salt=0):
    return pc.CompiledProviderPlan(pc.GroupSchedule(None, requested_max_moves_per_step),
                                   [pc.ProviderBindingPlan(handle, pc.POLICY_CALLBACK_GROUP, (slot.id,), rotation_seed_salt=salt)], pc.MOVE_GROUPED)


def test_callback_group_is_lazy_and_explicit_zero_clamps_to_one_candidate():
    slot = _slot()
    provider = CountingGroupProvider()
    registry = pc.RuntimeProviderRegistry([slot])
    plan = _callback_group_plan(slot, registry.add_callback(provider))
    reasons = pc.ProviderReasonArena()
    cursor = pc.RuntimeProviderCursor(plan, [0], pc.MoveStreamContext(), False)
    assert provider.pulls == 0
    i = cursor.next_candidate(registry, reasons)
    assert i is not None and provider.pulls == 1
    selected = cursor.take_candidate(i)
    assert cursor.next_candidate(registry, reasons) is None
    assert provider.pulls == 1
    assert reasons.label(selected.reason) == "candidate" and len(reasons) == 1
    assert provider.limits[0] == pc.GroupLimits(None, 1)  # (the clamped limit is what the callback sees)
    assert selected.kind == pc.MOVE_GROUPED and [(e.entity_index, e.to_value) for e in selected.edits] == [(0, 1)]


def test_static_group_stays_lazy_and_normalizes_typed_candidates_directly():
    slot = _slot()
    pulls = []

    def static_group_candidates(solution, limits):
        pulls.append(limits)
        assert limits.value_candidate_limit == 2 and limits.max_moves_per_step == 2
        return [pc.StaticCandidate("static_candidate", (pc.StaticEdit(0, "worker", 0, v),)) for v in (1, 2)]

    registry = pc.RuntimeProviderRegistry([slot])
    h = registry.add_static_group(static_group_candidates)
    plan = pc.CompiledProviderPlan(pc.GroupSchedule(2, 2), [pc.ProviderBindingPlan(h, pc.POLICY_STATIC_GROUP, (slot.id,))], pc.MOVE_GROUPED)
    reasons = pc.ProviderReasonArena()
    cursor = pc.RuntimeProviderCursor(plan, [0])
    assert not pulls
    first = cursor.take_candidate(cursor.next_candidate(registry, reasons)).reason
    second = cursor.take_candidate(cursor.next_candidate(registry, reasons)).reason
    assert cursor.next_candidate(registry, reasons) is None
    assert len(pulls) == 1 and first == second and reasons.label(first) == "static_candidate" and len(reasons) == 1


def test_static_repair_stays_on_the_typed_candidate_path():
    slot = _slot()
    pulls = []

    def static_repair_candidates(solution, limits):
        pulls.append(limits)
        assert (limits.max_matches_per_step, limits.max_repairs_per_match, limits.max_moves_per_step) == (1, 2, 2)
        return [pc.StaticCandidate("static_repair", (pc.StaticEdit(0, "worker", 0, v),)) for v in (1, 2)]

    registry = pc.RuntimeProviderRegistry([slot])
    h = registry.add_static_repair("hard_constraint", static_repair_candidates)
    plan = pc.CompiledProviderPlan(pc.RepairSchedule(("hard_constraint",), 1, 2, 2, False),
                                   [pc.ProviderBindingPlan(h, pc.POLICY_STATIC_REPAIR, (slot.id,))], pc.MOVE_CONFLICT_REPAIR)
    reasons = pc.ProviderReasonArena()
    cursor = pc.RuntimeProviderCursor(plan, [0])
    assert not pulls
    first = cursor.take_candidate(cursor.next_candidate(registry, reasons))
    second = cursor.take_candidate(cursor.next_candidate(registry, reasons))
    assert cursor.next_candidate(registry, reasons) is None
    assert len(pulls) == 1 and first.reason == second.reason and reasons.label(first.reason) == "static_repair" and len(reasons) == 1
    assert first.kind == pc.MOVE_CONFLICT_REPAIR


def test_provider_reason_arena_reuses_one_id_for_repeated_callback_labels():
    slot = _slot()
    provider = CountingGroupProvider()
    registry = pc.RuntimeProviderRegistry([slot])
    plan = _callback_group_plan(slot, registry.add_callback(provider), requested_max_moves_per_step=2)
    reasons = pc.ProviderReasonArena()
    cursor = pc.RuntimeProviderCursor(plan, [0])
    first = cursor.take_candidate(cursor.next_candidate(registry, reasons))
    second = cursor.take_candidate(cursor.next_candidate(registry, reasons))
    del cursor
    assert provider.pulls == 1 and first.reason == second.reason and len(reasons) == 1 and reasons.label(first.reason) == "candidate"


def test_concurrent_lazy_cursors_share_the_execution_arena_without_retaining_its_borrow():
    slot = _slot()
    provider = CountingGroupProvider()
    registry = pc.RuntimeProviderRegistry([slot])
    plan = _callback_group_plan(slot, registry.add_callback(provider))
    reasons = pc.ProviderReasonArena()
    first, second = pc.RuntimeProviderCursor(plan, [0]), pc.RuntimeProviderCursor(plan, [0])
    assert provider.pulls == 0
    a = first.take_candidate(first.next_candidate(registry, reasons)).reason
    b = second.take_candidate(second.next_candidate(registry, reasons)).reason
    assert provider.pulls == 2 and a == b and len(reasons) == 1 and reasons.label(a) == "candidate"


# ---- the selection order the cursor rotates with ---------------------------------------------------------------------------------------
def test_selection_order_matches_the_oracle_context(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for _ in range(300):
        idx, seed, salt = (int(x) for x in rng.integers(0, 2**63, 3))
        n = int(rng.integers(1, 40))
        for order in (pc.ORDER_ORIGINAL, pc.ORDER_SORTED, pc.ORDER_PROBABILISTIC, pc.ORDER_RANDOM, pc.ORDER_SHUFFLED):
            ctx = pc.MoveStreamContext(idx, seed, order)
            assert ctx.mixed_seed(salt) == L.sfo_ctx_mixed_seed(idx, seed, salt)
            assert ctx.random_stride(n, salt) == L.sfo_ctx_random_stride(idx, seed, n, salt)
            for off in range(n):
                assert ctx.selection_index(off, n, salt) == L.sfo_ctx_selection_index(idx, seed, order, off, n, salt)
    vals = list(range(7))
    pc.MoveStreamContext(3, 9, pc.ORDER_SHUFFLED).apply_selection_order(vals, 11)
    assert sorted(vals) == list(range(7)) and vals != list(range(7))  # a shuffled order is a permutation
    same = list(range(7))
    pc.MoveStreamContext(3, 9, pc.ORDER_ORIGINAL).apply_selection_order(same, 11)
    assert same == list(range(7))


# ---- scheduling rules (provider_cursor.rs:107-400) ---------------------------------------------------------------------------------------
def _drain(cursor, registry, reasons):
    out = []
    while (i := cursor.next_candidate(registry, reasons)) is not None:
        out.append(cursor.take_candidate(i))
    return out


def test_callback_group_caps_in_callback_order_then_rotates_and_static_group_rotates_first():
    slot = _slot(n_values=8)
    ctx = pc.MoveStreamContext(4, 77, pc.ORDER_SHUFFLED)
    values = (1, 2, 3, 4, 5, 6)
    reasons = pc.ProviderReasonArena()
    # callback group: normalise, cap to 4 in callback order, then rotate those four
    provider = CountingGroupProvider(values)
    registry = pc.RuntimeProviderRegistry([slot])
    plan = _callback_group_plan(slot, registry.add_callback(provider), requested_max_moves_per_step=4, salt=21)
    got = [m.edits[0].to_value for m in _drain(pc.RuntimeProviderCursor(plan, [0], ctx), registry, reasons)]
    want = [1, 2, 3, 4]
    ctx.apply_selection_order(want, 21)
    assert got == want and sorted(got) == [1, 2, 3, 4]
    # static group: rotate the provider's whole output, then filter, then stop at max_moves
    registry2 = pc.RuntimeProviderRegistry([slot])
    h = registry2.add_static_group(lambda sol, lim: [pc.StaticCandidate("g", (pc.StaticEdit(0, "worker", 0, v),)) for v in values])
    plan2 = pc.CompiledProviderPlan(pc.GroupSchedule(None, 4), [pc.ProviderBindingPlan(h, pc.POLICY_STATIC_GROUP, (slot.id,), rotation_seed_salt=21)])
    got2 = [m.edits[0].to_value for m in _drain(pc.RuntimeProviderCursor(plan2, [0], ctx), registry2, reasons)]
    want2 = list(values)
    ctx.apply_selection_order(want2, 21)
    assert got2 == want2[:4]
    # declared_max_moves_per_step applies when nothing is requested; an explicit zero on a STATIC group is a no-op (no pull at all)
    pulls = []
    registry3 = pc.RuntimeProviderRegistry([slot])
    h3 = registry3.add_static_group(lambda sol, lim: pulls.append(lim) or [pc.StaticCandidate("g", (pc.StaticEdit(0, "worker", 0, v),)) for v in values])
    plan3 = pc.CompiledProviderPlan(pc.GroupSchedule(None, None), [pc.ProviderBindingPlan(h3, pc.POLICY_STATIC_GROUP, (slot.id,), declared_max_moves_per_step=3)])
    assert len(_drain(pc.RuntimeProviderCursor(plan3, [0]), registry3, reasons)) == 3 and pulls[0].max_moves_per_step == 3
    plan3.schedule = pc.GroupSchedule(None, 0)
    assert _drain(pc.RuntimeProviderCursor(plan3, [0]), registry3, reasons) == [] and len(pulls) == 1


def test_normalisation_drops_duplicates_and_non_doable_candidates_and_raises_like_the_resolver():
    slot = _slot(n_values=3)
    other = _slot(descriptor_index=1, entity_class="Other", variable_name="shift")
    reasons = pc.ProviderReasonArena()

    def run(cands, allowed=None, slots=(slot, other), solution=(0, 1)):
        registry = pc.RuntimeProviderRegistry(list(slots))
        h = registry.add_callback(lambda sol, lim: cands)
        plan = pc.CompiledProviderPlan(pc.GroupSchedule(None, 10), [pc.ProviderBindingPlan(h, pc.POLICY_CALLBACK_GROUP, allowed or (slot.id,))])
        return _drain(pc.RuntimeProviderCursor(plan, list(solution)), registry, reasons)

    E = pc.RawProviderEdit
    C = pc.RawProviderCandidate
    moves = run([C("a", (E(None, "worker", 0, 1),)), C("a", (E("Row", "worker", 0, 1),)),  # the same candidate twice (alias by class name): kept once
                 C("b", (E(None, "worker", 0, 1),)),  # another reason: a different candidate
                 C("a", ()),  # no edits: skipped
                 C("a", (E(None, "worker", 0, 2), E("Row", "worker", 0, 1))),  # two edits of one target: skipped
                 C("a", (E(None, "worker", 0, 0),)),  # changes nothing: resolved, but not doable -> not stored
                 C("a", (E(None, "worker", 0, 0), E(None, "worker", 1, 2)))])  # one edit changes a value: doable
    assert [(reasons.label(m.reason), [(e.entity_index, e.to_value) for e in m.edits]) for m in moves] == [
        ("a", [(0, 1)]), ("b", [(0, 1)]), ("a", [(0, 0), (1, 2)])]
    for cands, kind in (([C("a", (E(None, "nope", 0, 1),))], "UnknownSlot"), ([C("a", (E(None, "shift", 0, 1),))], "SlotOutsideSelector"),
                        ([C("a", (E(None, "worker", 2, 1),))], "EntityIndexOutOfBounds"), ([C("a", (E(None, "worker", 0, 3),))], "IllegalValue"),
                        ([C("a", (E(None, "worker", 0, None),))], "IllegalValue")):
        with pytest.raises(pc.ProviderResolutionError) as e:
            run(cands)
        assert e.value.kind == kind
    assert [m.edits[0].to_value for m in run([C("a", (E(None, "worker", 0, None),))], slots=(_slot(allows_unassigned=True),))] == [None]
    with pytest.raises(ValueError):
        pc.RuntimeProviderRegistry([slot, _slot()])  # duplicate (descriptor, variable)


def test_repair_schedule_callbacks_first_then_static_under_the_three_caps():
    slot = _slot(n_values=16)
    reasons = pc.ProviderReasonArena()
    log = []

    def callback(name, values):
        def pull(solution, limits):
            log.append((name, limits))
            return [pc.RawProviderCandidate(name, (pc.RawProviderEdit(None, "worker", 0, v),)) for v in values]
        return pull

    def static(name, values):
        def pull(solution, limits):
            log.append((name, limits))
            return [pc.StaticCandidate(name, (pc.StaticEdit(0, "worker", 0, v),)) for v in values]
        return pull

    def build(schedule, ctx=pc.MoveStreamContext()):
        registry = pc.RuntimeProviderRegistry([slot])
        bindings = [
            pc.ProviderBindingPlan(registry.add_callback(callback("cb_other", (9,)), constraints=("unrelated",)), pc.POLICY_CALLBACK_REPAIR, (slot.id,)),
            pc.ProviderBindingPlan(registry.add_callback(callback("cb_multi", (1, 2, 3)), constraints=("c1", "c2")), pc.POLICY_CALLBACK_REPAIR, (slot.id,)),
            pc.ProviderBindingPlan(registry.add_static_repair("c1", static("st_c1", (4, 5, 6))), pc.POLICY_STATIC_REPAIR, (slot.id,), declared_schema_index=0),
            pc.ProviderBindingPlan(registry.add_static_repair("c2", static("st_c2", (4, 7))), pc.POLICY_STATIC_REPAIR, (slot.id,), declared_schema_index=1),
        ]
        plan = pc.CompiledProviderPlan(schedule, bindings, pc.MOVE_COMPOUND_CONFLICT_REPAIR)
        return registry, pc.RuntimeProviderCursor(plan, [0], ctx, require_hard_improvement=True)

    # a multi-constraint callback is called once; providers that declare none of the constraints are not called; two repairs per match
    log.clear()
    registry, cursor = build(pc.RepairSchedule(("c1", "c2"), 8, 2, 100, True))
    moves = _drain(cursor, registry, reasons)
    assert [n for n, _ in log] == ["cb_multi", "st_c1", "st_c2"]
    assert log[0][1] == pc.RepairLimits(8, 2, 100, ("c1", "c2"), True) and log[1][1] == pc.RepairLimits(8, 2, 100)
    # st_c2's (4) repeats st_c1's edit under another reason label: a different candidate; the static stream shares ONE dedup scope
    assert [(reasons.label(m.reason), m.edits[0].to_value) for m in moves] == [("cb_multi", 1), ("cb_multi", 2), ("st_c1", 4), ("st_c1", 5), ("st_c2", 4), ("st_c2", 7)]
    assert all(m.require_hard_improvement and m.kind == pc.MOVE_COMPOUND_CONFLICT_REPAIR for m in moves)
    # max_matches_per_step counts provider invocations across callbacks and static repairs
    log.clear()
    registry, cursor = build(pc.RepairSchedule(("c1", "c2"), 2, 2, 100))
    assert len(_drain(cursor, registry, reasons)) == 4 and [n for n, _ in log] == ["cb_multi", "st_c1"]
    # max_moves_per_step stops the stream in the middle of a provider's output, and the static repairs are not reached once it is full
    log.clear()
    registry, cursor = build(pc.RepairSchedule(("c1", "c2"), 8, 3, 3))
    assert [m.edits[0].to_value for m in _drain(cursor, registry, reasons)] == [1, 2, 3] and [n for n, _ in log] == ["cb_multi"]
    # any zero limit or no constraint: nothing is pulled
    for sched in (pc.RepairSchedule((), 8, 2, 8), pc.RepairSchedule(("c1",), 0, 2, 8), pc.RepairSchedule(("c1",), 8, 0, 8), pc.RepairSchedule(("c1",), 8, 2, 0)):
        log.clear()
        registry, cursor = build(sched)
        assert _drain(cursor, registry, reasons) == [] and not log
    # a seeded step rotates constraints (salt ^ max_moves), providers per constraint (salt ^ constraint index) and each provider's output
    # (salt ^ declared schema index) -- same multiset, reproducible, and the callbacks still come first
    ctx = pc.MoveStreamContext(12, 3456, pc.ORDER_SHUFFLED)
    registry, cursor = build(pc.RepairSchedule(("c1", "c2"), 8, 3, 100), ctx)
    a = [(reasons.label(m.reason), m.edits[0].to_value) for m in _drain(cursor, registry, reasons)]
    registry, cursor = build(pc.RepairSchedule(("c1", "c2"), 8, 3, 100), ctx)
    b = [(reasons.label(m.reason), m.edits[0].to_value) for m in _drain(cursor, registry, reasons)]
    assert a == b and a[0][0] == "cb_multi" and sorted(a) == sorted([("cb_multi", 1), ("cb_multi", 2), ("cb_multi", 3), ("st_c1", 4), ("st_c1", 5), ("st_c1", 6), ("st_c2", 4), ("st_c2", 7)])


def test_drain_for_step_decide_hands_the_device_its_arguments():
    slot = _slot(n_values=4)
    registry = pc.RuntimeProviderRegistry([slot])
    h = registry.add_callback(lambda sol, lim: [pc.RawProviderCandidate("swap", (pc.RawProviderEdit(None, "worker", 0, 2), pc.RawProviderEdit(None, "worker", 1, 0))),
                                                pc.RawProviderCandidate("move", (pc.RawProviderEdit(None, "worker", 1, 3),))])
    plan = pc.CompiledProviderPlan(pc.GroupSchedule(None, None), [pc.ProviderBindingPlan(h, pc.POLICY_CALLBACK_GROUP, (slot.id,))])
    reasons = pc.ProviderReasonArena()
    cands, gates, reason_ids = pc.RuntimeProviderCursor(plan, [0, 2], require_hard_improvement=True).drain_for_step_decide(registry, reasons)
    assert cands == [[(0, 2), (1, 0)], [(1, 3)]] and gates.dtype == np.int32 and gates.tolist() == [1, 1]
    assert [reasons.label(r) for r in reason_ids] == ["swap", "move"]
