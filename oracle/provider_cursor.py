"""ORACLE (test infrastructure; the product package never imports this file).  A cited CPU restatement of the reference's compound-provider cursor: the one lazy cursor that schedules grouped-scalar and conflict-repair
providers (crates/solverforge-solver/src/runtime/provider_cursor.rs:1-492), with the normalisation kernel it calls
(builder/context/provider/resolver.rs:90-345), the per-run reason arena (builder/context/provider/types.rs:82-130), the doability rule of the
move it emits (heuristic/move/runtime_compound.rs:121-137) and the step-seeded selection order it rotates with
(heuristic/selector/move_selector/iter.rs:40-200).

Nothing here touches a score.  The cursor decides WHICH provider is pulled WHEN, with which limits, how its output is rotated, capped,
resolved to slots, validated and deduplicated; what comes out is a list of compound scalar candidates, which the device prices in one
launch through `ScoreDirector.step_decide(..., gates=...)` (sf_step_decide_gated; `require_hard_improvement` is gate bit 0).  In a drop-in
the Rust cursor keeps this job; this restatement exists so that the GPU parity tests can drive `sf_step_decide_gated` with
exactly the candidate stream the reference's cursor would hand it (tests/test_gpu_provider_step.py) and is itself pinned to
`runtime/provider_cursor_tests.rs` by `tests/test_provider_cursor.py`.
"""
from __future__ import annotations

from dataclasses import dataclass
from math import gcd
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

_M64 = (1 << 64) - 1


# ---- MoveStreamContext (iter.rs:20-200) ----------------------------------------------------------------------------------------------
ORDER_ORIGINAL, ORDER_SORTED, ORDER_PROBABILISTIC, ORDER_RANDOM, ORDER_SHUFFLED = 0, 1, 2, 3, 4  # = sf_selection_order (include/solverforge_amd.h)


def _splitmix64(v: int) -> int:  # iter.rs:191-196
    v = (v + 0x9E3779B97F4A7C15) & _M64
    v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & _M64
    return v ^ (v >> 31)


@dataclass(frozen=True)
class MoveStreamContext:
    """step_index / step_seed / selection order of the step the cursor runs in (iter.rs:20-45); the default is canonical (Original)."""

    step_index: int = 0
    step_seed: int = 0
    selection_order: int = ORDER_ORIGINAL

    def is_canonical(self) -> bool:  # iter.rs:175-180
        return self.selection_order in (ORDER_ORIGINAL, ORDER_SORTED, ORDER_PROBABILISTIC)

    def mixed_seed(self, salt: int) -> int:  # iter.rs:182-184
        return _splitmix64((self.step_seed ^ ((self.step_index * 0x9E3779B97F4A7C15) & _M64) ^ salt) & _M64)

    def random_index(self, n: int, salt: int) -> int:  # iter.rs:87-92
        return 0 if n <= 1 else self.mixed_seed(salt) % n

    def random_stride(self, n: int, salt: int) -> int:  # iter.rs:94-103
        if n <= 1:
            return 1
        stride = self.mixed_seed(salt) % (n - 1) + 1
        while gcd(stride, n) != 1:
            stride = 1 if stride == n - 1 else stride + 1
        return stride

    def selection_index(self, offset: int, n: int, salt: int) -> int:  # iter.rs:109-128
        assert offset < n
        if self.is_canonical():
            return offset
        if self.selection_order == ORDER_RANDOM:
            return self.random_index(n, salt ^ ((offset * 0xD1B54A32D192ED03) & _M64))
        start = self.random_index(n, salt)
        stride = self.random_stride(n, salt ^ 0xA24BAED4963EE407)
        return (start + offset * stride) % n

    def apply_selection_order(self, values: list, salt: int) -> None:  # iter.rs:149-157 (in place, like the reference)
        if self.is_canonical():
            return
        canonical = list(values)
        for offset in range(len(values)):
            values[offset] = canonical[self.selection_index(offset, len(canonical), salt)]


# ---- slots, edits, candidates ---------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SlotId:
    descriptor_index: int
    variable_index: int
    entity_class: str
    variable_name: str


@dataclass
class ScalarSlot:
    """One scalar planning variable as the providers address it.  `values[solution]` is read through `get`; the legal values of an entity are
    0 .. n_values - 1, plus None when the variable allows unassigned (RuntimeScalarSlot::value_is_legal)."""

    id: SlotId
    n_values: int
    allows_unassigned: bool = False
    get: Callable = None  # (solution, entity_index) -> Optional[int]
    entity_count: Callable = None  # (solution) -> int

    def value_is_legal(self, solution, entity_index: int, to_value: Optional[int]) -> bool:
        if to_value is None:
            return self.allows_unassigned
        return 0 <= to_value < self.n_values


@dataclass(frozen=True)
class RawProviderEdit:  # types.rs: a host callback's edit, addressed by names
    entity_class: Optional[str]
    variable_name: str
    entity_index: int
    to_value: Optional[int]


@dataclass(frozen=True)
class RawProviderCandidate:
    reason: str
    edits: Tuple[RawProviderEdit, ...]


@dataclass(frozen=True)
class StaticEdit:  # ScalarTarget::set: a typed edit, addressed by descriptor index + variable name
    descriptor_index: int
    variable_name: str
    entity_index: int
    to_value: Optional[int]


@dataclass(frozen=True)
class StaticCandidate:  # ScalarCandidate / RepairCandidate
    reason: str
    edits: Tuple[StaticEdit, ...]


@dataclass(frozen=True)
class ResolvedEdit:
    slot_index: int
    descriptor_index: int
    variable_index: int
    entity_index: int
    to_value: Optional[int]


@dataclass(frozen=True)
class ResolvedCandidate:
    reason: int  # id in the run's reason arena
    edits: Tuple[ResolvedEdit, ...]


class ProviderResolutionError(ValueError):
    """resolver.rs: UnknownSlot / SlotOutsideSelector / EntityIndexOutOfBounds / IllegalValue -- the reference raises through its host
    boundary; `kind` names the variant."""

    def __init__(self, kind: str, **details):
        super().__init__("%s %r" % (kind, details))
        self.kind = kind
        self.details = details


class ProviderReasonArena:
    """Per-run reason interning (types.rs:82-130): static and host labels with the same text share one id."""

    def __init__(self):
        self._ids = {}
        self._labels: List[str] = []

    def intern(self, label: str) -> int:
        i = self._ids.get(label)
        if i is None:
            i = len(self._labels)
            if i > 0xFFFFFFFF:
                raise OverflowError("a single solve cannot intern more than u32::MAX reasons")
            self._labels.append(label)
            self._ids[label] = i
        return i

    def label(self, reason_id: int) -> str:
        if not 0 <= reason_id < len(self._labels):
            raise LookupError("runtime compound move refers to a reason outside its run arena")
        return self._labels[reason_id]

    def __len__(self):
        return len(self._labels)


class ProviderNormalizationState:
    """Deduplication scope of one explicit provider result stream (types.rs:372-376)."""

    def __init__(self):
        self.seen_candidates = set()


# ---- registry (builder/context/provider/registry.rs) -----------------------------------------------------------------------------------
@dataclass(frozen=True)
class GroupLimits:  # RuntimeProviderLimits::Group
    value_candidate_limit: Optional[int]
    max_moves_per_step: Optional[int]


@dataclass(frozen=True)
class RepairLimits:  # RuntimeProviderLimits::Repair / RepairLimits
    max_matches_per_step: int
    max_repairs_per_match: int
    max_moves_per_step: int
    constraints: Tuple[str, ...] = ()
    include_soft_matches: bool = False


@dataclass
class _Callback:
    pull: Callable  # (solution, limits) -> [RawProviderCandidate]
    constraints: Tuple[str, ...] = ()  # repair callbacks: the constraints they declare


@dataclass
class _StaticGroup:
    pull: Callable  # (solution, GroupLimits) -> [StaticCandidate]


@dataclass
class _StaticRepair:
    constraint: str
    pull: Callable  # (solution, RepairLimits) -> [StaticCandidate]


HANDLE_CALLBACK, HANDLE_STATIC_GROUP, HANDLE_STATIC_REPAIR = "callback", "static_group", "static_repair"


class RuntimeProviderRegistry:
    """The frozen slots and the providers a plan's handles point at."""

    def __init__(self, slots: Sequence[ScalarSlot]):
        seen = set()
        for s in slots:
            key = (s.id.descriptor_index, s.id.variable_index)
            if key in seen:  # resolver.rs:74-80
                raise ValueError("runtime provider registry has duplicate scalar slot %s.%s" % (s.id.entity_class, s.id.variable_name))
            seen.add(key)
        self.slots = list(slots)
        self.callbacks: List[_Callback] = []
        self.static_groups: List[_StaticGroup] = []
        self.static_repairs: List[_StaticRepair] = []

    def add_callback(self, pull, constraints=()):
        self.callbacks.append(_Callback(pull, tuple(constraints)))
        return (HANDLE_CALLBACK, len(self.callbacks) - 1)

    def add_static_group(self, pull):
        self.static_groups.append(_StaticGroup(pull))
        return (HANDLE_STATIC_GROUP, len(self.static_groups) - 1)

    def add_static_repair(self, constraint, pull):
        self.static_repairs.append(_StaticRepair(constraint, pull))
        return (HANDLE_STATIC_REPAIR, len(self.static_repairs) - 1)

    # -- constraint declarations
    def declares_constraint(self, handle, constraint: str) -> bool:
        kind, i = handle
        if kind == HANDLE_CALLBACK:
            return constraint in self.callbacks[i].constraints
        if kind == HANDLE_STATIC_REPAIR:
            return self.static_repairs[i].constraint == constraint
        return False

    def declares_any_constraint(self, handle, constraints) -> bool:
        return any(self.declares_constraint(handle, c) for c in constraints)

    # -- pulls
    def pull_callback_raw(self, handle, solution, limits):
        assert handle[0] == HANDLE_CALLBACK
        return list(self.callbacks[handle[1]].pull(solution, limits))

    def pull_static_group(self, index, solution, value_candidate_limit, max_moves_per_step):
        return list(self.static_groups[index].pull(solution, GroupLimits(value_candidate_limit, max_moves_per_step)))

    def pull_static_repair(self, index, solution, limits: RepairLimits):
        return list(self.static_repairs[index].pull(solution, limits))

    # -- the one normalisation kernel (resolver.rs:254-345)
    def _allowed(self, slot: ScalarSlot, allowed_slots) -> bool:
        return any(a.descriptor_index == slot.id.descriptor_index and a.variable_index == slot.id.variable_index for a in allowed_slots)

    def _resolve(self, matches, allowed_slots, unknown_details) -> int:
        first = next((i for i, s in enumerate(self.slots) if matches(s)), None)
        if first is None:
            raise ProviderResolutionError("UnknownSlot", **unknown_details)
        index = next((i for i, s in enumerate(self.slots) if matches(s) and self._allowed(s, allowed_slots)), None)
        if index is None:
            s = self.slots[first]
            raise ProviderResolutionError("SlotOutsideSelector", entity_class=s.id.entity_class, variable_name=s.id.variable_name)
        return index

    def resolve_raw_index(self, edit: RawProviderEdit, allowed_slots) -> int:  # resolver.rs:90-125
        return self._resolve(lambda s: (edit.entity_class is None or edit.entity_class == s.id.entity_class) and edit.variable_name == s.id.variable_name,
                             allowed_slots, dict(entity_class=edit.entity_class, variable_name=edit.variable_name))

    def resolve_static_index(self, edit: StaticEdit, allowed_slots) -> int:  # resolver.rs:127-160
        return self._resolve(lambda s: s.id.descriptor_index == edit.descriptor_index and s.id.variable_name == edit.variable_name,
                             allowed_slots, dict(entity_class=None, variable_name=edit.variable_name))

    def _normalize(self, solution, candidates, allowed_slots, state, reasons, resolve) -> List[ResolvedCandidate]:
        normalized = []
        for cand in candidates:
            if not cand.edits:
                continue
            edits, seen_targets, duplicate_target = [], set(), False
            for edit in cand.edits:
                si = resolve(edit, allowed_slots)  # (raises before the duplicate test, like the reference's `?`)
                slot = self.slots[si]
                target = (slot.id.descriptor_index, slot.id.variable_index, edit.entity_index)
                if target in seen_targets:
                    duplicate_target = True
                    break
                seen_targets.add(target)
                if edit.entity_index >= slot.entity_count(solution):
                    raise ProviderResolutionError("EntityIndexOutOfBounds", entity_class=slot.id.entity_class, variable_name=slot.id.variable_name,
                                                  entity_index=edit.entity_index)
                if not slot.value_is_legal(solution, edit.entity_index, edit.to_value):
                    raise ProviderResolutionError("IllegalValue", entity_class=slot.id.entity_class, variable_name=slot.id.variable_name,
                                                  entity_index=edit.entity_index, to_value=edit.to_value)
                edits.append(ResolvedEdit(si, slot.id.descriptor_index, slot.id.variable_index, edit.entity_index, edit.to_value))
            if duplicate_target:
                continue
            reason = reasons.intern(cand.reason)
            key = (reason, tuple((e.descriptor_index, e.variable_index, e.entity_index, e.to_value) for e in edits))
            if key in state.seen_candidates:
                continue
            state.seen_candidates.add(key)
            normalized.append(ResolvedCandidate(reason, tuple(edits)))
        return normalized

    def normalize_or_raise(self, solution, raw, allowed_slots, state, reasons):  # registry.rs:289-307
        return self._normalize(solution, raw, allowed_slots, state, reasons, self.resolve_raw_index)

    def normalize_static_group(self, solution, native, allowed_slots, state, reasons):  # registry.rs:309-326
        return self._normalize(solution, native, allowed_slots, state, reasons, self.resolve_static_index)

    normalize_static_repair = normalize_static_group  # registry.rs:328-345 (same kernel, RepairCandidate)


# ---- compiled plan (runtime/compiler: CompiledProviderPlan) ---------------------------------------------------------------------------
POLICY_CALLBACK_GROUP, POLICY_STATIC_GROUP, POLICY_CALLBACK_REPAIR, POLICY_STATIC_REPAIR = range(4)
MOVE_GROUPED, MOVE_CONFLICT_REPAIR, MOVE_COMPOUND_CONFLICT_REPAIR = range(3)


@dataclass
class ProviderBindingPlan:
    handle: tuple
    policy: int
    allowed_slots: Tuple[SlotId, ...]
    declared_schema_index: int = 0
    rotation_seed_salt: int = 0  # CallbackGroup / StaticGroup / CallbackRepair
    declared_max_moves_per_step: Optional[int] = None  # StaticGroup
    constraint_rotation_seed_salt: int = 0  # StaticRepair
    provider_rotation_seed_salt: int = 0
    spec_rotation_seed_salt: int = 0


@dataclass
class GroupSchedule:  # ProviderSchedule::Group
    value_candidate_limit: Optional[int] = None
    requested_max_moves_per_step: Optional[int] = None


@dataclass
class RepairSchedule:  # ProviderSchedule::Repair
    constraints: Tuple[str, ...]
    max_matches_per_step: int
    max_repairs_per_match: int
    max_moves_per_step: int
    include_soft_matches: bool = False


@dataclass
class CompiledProviderPlan:
    schedule: object
    bindings: List[ProviderBindingPlan]
    move_kind: int = MOVE_GROUPED


@dataclass(frozen=True)
class RuntimeCompoundMove:
    kind: int
    reason: int
    edits: Tuple[ResolvedEdit, ...]
    require_hard_improvement: bool

    def is_doable_on(self, registry: RuntimeProviderRegistry, solution) -> bool:  # runtime_compound.rs:121-137
        if not self.edits:
            return False
        targets = [(e.descriptor_index, e.variable_index, e.entity_index) for e in self.edits]
        if len(set(targets)) != len(targets):
            return False
        changes = False
        for e in self.edits:
            slot = registry.slots[e.slot_index]
            if e.entity_index >= slot.entity_count(solution) or not slot.value_is_legal(solution, e.entity_index, e.to_value):
                return False
            changes = changes or slot.get(solution, e.entity_index) != e.to_value
        return changes


class RuntimeProviderCursor:
    """provider_cursor.rs:37-70.  A provider source is never pulled by the constructor; its first pull happens when `next_candidate` is first
    reached.  The reason arena belongs to the caller and is borrowed per call."""

    def __init__(self, plan: CompiledProviderPlan, solution, context: MoveStreamContext = MoveStreamContext(), require_hard_improvement: bool = False):
        self.plan = plan
        self.solution = solution
        self.context = context
        self.require_hard_improvement = require_hard_improvement
        self._store: List[Optional[RuntimeCompoundMove]] = []
        self._next_index = 0
        self._prepared = False

    # -- provider_cursor.rs:72-105
    def _prepare(self, registry, reasons):
        if self._prepared:
            return
        self._prepared = True
        s = self.plan.schedule
        if isinstance(s, GroupSchedule):
            self._prepare_group(registry, s.value_candidate_limit, s.requested_max_moves_per_step, reasons)
        else:
            self._prepare_repair(registry, list(s.constraints), s.max_matches_per_step, s.max_repairs_per_match, s.max_moves_per_step, s.include_soft_matches, reasons)

    # -- provider_cursor.rs:107-190
    def _prepare_group(self, registry, value_candidate_limit, requested_max_moves_per_step, reasons):
        for binding in self.plan.bindings:
            if binding.policy == POLICY_CALLBACK_GROUP:
                # the public group callback contract treats an explicit zero as one candidate rather than a no-op
                max_moves = max(requested_max_moves_per_step if requested_max_moves_per_step is not None else 256, 1)
            elif binding.policy == POLICY_STATIC_GROUP:
                max_moves = requested_max_moves_per_step if requested_max_moves_per_step is not None else (
                    binding.declared_max_moves_per_step if binding.declared_max_moves_per_step is not None else 256)
            else:
                continue
            if max_moves == 0:
                continue
            state = ProviderNormalizationState()
            if binding.policy == POLICY_CALLBACK_GROUP:
                raw = registry.pull_callback_raw(binding.handle, self.solution, GroupLimits(value_candidate_limit, max_moves))
                candidates = registry.normalize_or_raise(self.solution, raw, binding.allowed_slots, state, reasons)
                # callback order is normalised / deduplicated first, capped in callback order, then step-rotated
                del candidates[max_moves:]
                self.context.apply_selection_order(candidates, binding.rotation_seed_salt)
            else:
                assert binding.handle[0] == HANDLE_STATIC_GROUP, "static group policy must retain a static group handle"
                native = registry.pull_static_group(binding.handle[1], self.solution, value_candidate_limit, max_moves)
                # native groups rotate provider output BEFORE validity / dedup filtering
                self.context.apply_selection_order(native, binding.rotation_seed_salt)
                candidates = registry.normalize_static_group(self.solution, native, binding.allowed_slots, state, reasons)
            for c in candidates:
                if len(self._store) >= max_moves:
                    break
                self._push_candidate(registry, c)

    # -- provider_cursor.rs:192-235
    def _prepare_repair(self, registry, constraints, max_matches_per_step, max_repairs_per_match, max_moves_per_step, include_soft_matches, reasons):
        if not constraints or max_matches_per_step == 0 or max_repairs_per_match == 0 or max_moves_per_step == 0:
            return
        invocations = [0]
        self._prepare_callback_repairs(registry, constraints, max_matches_per_step, max_repairs_per_match, max_moves_per_step, include_soft_matches, invocations, reasons)
        if len(self._store) < max_moves_per_step and invocations[0] < max_matches_per_step:
            self._prepare_static_repairs(registry, constraints, max_matches_per_step, max_repairs_per_match, max_moves_per_step, invocations, reasons)

    # -- provider_cursor.rs:237-310
    def _prepare_callback_repairs(self, registry, constraints, max_matches_per_step, max_repairs_per_match, max_moves_per_step, include_soft_matches, invocations, reasons):
        indexes = [i for i, b in enumerate(self.plan.bindings) if b.policy == POLICY_CALLBACK_REPAIR]
        if not indexes:
            return
        salt = self.plan.bindings[indexes[0]].rotation_seed_salt
        # rotate the complete callback declaration stream before testing constraint membership; a multi-constraint provider is called once
        self.context.apply_selection_order(indexes, salt)
        limits = RepairLimits(max_matches_per_step, max_repairs_per_match, max_moves_per_step, tuple(constraints), include_soft_matches)
        for bi in indexes:
            if len(self._store) >= max_moves_per_step or invocations[0] >= max_matches_per_step:
                break
            binding = self.plan.bindings[bi]
            if not registry.declares_any_constraint(binding.handle, constraints):
                continue
            invocations[0] += 1
            raw = registry.pull_callback_raw(binding.handle, self.solution, limits)
            candidates = registry.normalize_or_raise(self.solution, raw, binding.allowed_slots, ProviderNormalizationState(), reasons)
            del candidates[max_repairs_per_match:]
            for c in candidates:
                if len(self._store) >= max_moves_per_step:
                    break
                self._push_candidate(registry, c)

    # -- provider_cursor.rs:312-400
    def _prepare_static_repairs(self, registry, constraints, max_matches_per_step, max_repairs_per_match, max_moves_per_step, invocations, reasons):
        static_indexes = [i for i, b in enumerate(self.plan.bindings) if b.policy == POLICY_STATIC_REPAIR]
        if not static_indexes:
            return
        first = self.plan.bindings[static_indexes[0]]
        constraint_indexes = list(range(len(constraints)))
        self.context.apply_selection_order(constraint_indexes, first.constraint_rotation_seed_salt ^ max_moves_per_step)
        state = ProviderNormalizationState()  # (one dedup scope for the whole static repair stream)
        for ci in constraint_indexes:
            constraint = constraints[ci]
            indexes = [bi for bi in static_indexes if registry.declares_constraint(self.plan.bindings[bi].handle, constraint)]
            self.context.apply_selection_order(indexes, first.provider_rotation_seed_salt ^ ci)
            for bi in indexes:
                if len(self._store) >= max_moves_per_step or invocations[0] >= max_matches_per_step:
                    return
                invocations[0] += 1
                binding = self.plan.bindings[bi]
                assert binding.handle[0] == HANDLE_STATIC_REPAIR, "static repair policy must retain a static repair handle"
                native = registry.pull_static_repair(binding.handle[1], self.solution, RepairLimits(max_matches_per_step, max_repairs_per_match, max_moves_per_step))
                self.context.apply_selection_order(native, first.spec_rotation_seed_salt ^ binding.declared_schema_index)
                del native[max_repairs_per_match:]
                for c in registry.normalize_static_repair(self.solution, native, binding.allowed_slots, state, reasons):
                    if len(self._store) >= max_moves_per_step:
                        return
                    self._push_candidate(registry, c)

    # -- provider_cursor.rs:420-437
    def _push_candidate(self, registry, candidate: ResolvedCandidate):
        mov = RuntimeCompoundMove(self.plan.move_kind, candidate.reason, candidate.edits, self.require_hard_improvement)
        if mov.is_doable_on(registry, self.solution):
            self._store.append(mov)

    # -- provider_cursor.rs:447-466
    def next_candidate(self, registry, reasons) -> Optional[int]:
        self._prepare(registry, reasons)
        while self._next_index < len(self._store):
            i = self._next_index
            self._next_index += 1
            if self._store[i] is not None:
                return i
        return None

    def take_candidate(self, candidate_id: int) -> RuntimeCompoundMove:
        mov = self._store[candidate_id]
        if mov is None:
            raise LookupError("candidate %d was already taken" % candidate_id)
        self._store[candidate_id] = None
        return mov

    # -- the boundary to the device: everything the cursor holds, in pull order, as sf_step_decide_gated's arguments
    def drain_for_step_decide(self, registry, reasons, device_slot: int = 0):
        """Every remaining candidate, in pull order, as the arguments of `ScoreDirector.step_decide(candidates, gates=gates)`:
        (candidates [[(entity_index, to_value or -1), ...], ...], gates [n] int32 with bit 0 = require_hard_improvement, reason ids [n]).
        The device model has one scalar planning class: every edit must address registry slot `device_slot`."""
        cands, gates, reason_ids = [], [], []
        while True:
            i = self.next_candidate(registry, reasons)
            if i is None:
                break
            mov = self.take_candidate(i)
            if any(e.slot_index != device_slot for e in mov.edits):
                raise NotImplementedError("the device prices edits of one scalar planning variable (slot %d)" % device_slot)
            cands.append([(e.entity_index, -1 if e.to_value is None else e.to_value) for e in mov.edits])
            gates.append(1 if mov.require_hard_improvement else 0)
            reason_ids.append(mov.reason)
        return cands, np.asarray(gates, dtype=np.int32), reason_ids
