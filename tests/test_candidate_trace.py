"""Candidate-trace wire format v3 (reference stats/candidate_trace.rs): the native encoder
(sf_trace_encode_step / sf_trace_digest_*) against an independent restatement (oracle/trace_v3.py), hand-framed bytes,
the published FNV-1a vectors for digest lane one, and — on the GPU — the digest of a traced device run against the
digest of the CPU oracle's run."""
import struct

import numpy as np
import pytest

import solverforge_amd as sfa
from oracle import sfo, trace_v3
from solverforge_amd import candidate_trace as ct
from solverforge_amd import datasets

BITS = {"nearby_change": 16, "nearby_swap": 32, "list_change": 4, "list_swap": 8, "list_reverse": 64,
        "sublist_change": 128, "sublist_swap": 256, "kopt": 512, "change": 1, "swap": 2, "ruin": 1024, "permute": 8192,
        "precedence": 16384}


def t6(m):
    return np.stack([m["kind"], m["a"], m["a_pos"], m["b"], m["b_pos"], m["value"]], axis=1)


def test_digest_lane_one_is_fnv1a64():
    # published FNV-1a 64 vectors (offset basis / prime = candidate_trace.rs:30-31)
    assert ct.digest_of_bytes(b"")[0] == 0xCBF29CE484222325
    assert ct.digest_of_bytes(b"a")[0] == 0xAF63DC4C8601EC8C
    assert ct.digest_of_bytes(b"foobar")[0] == 0x85944171F73967E8
    assert ct.digest_of_bytes(b"")[1] == 0x9E3779B97F4A7C15  # CandidateTraceDigest::empty
    for data in (b"", b"a", b"foobar", bytes(range(256)) * 3):
        assert ct.digest_of_bytes(data) == trace_v3.Digest().update(data).value()


def test_digest_is_incremental():
    t = ct.CandidateTrace()
    mv = np.zeros(3, dtype=sfa.MOVE_DTYPE)
    mv["kind"] = [2, 3, 4]
    mv["a"], mv["a_pos"], mv["b"], mv["b_pos"], mv["value"] = [1, 2, 3], [4, 5, 6], [1, 0, 3], [7, 8, 9], -1
    t.record_step(mv[:2], np.array([3, 1], dtype=np.int32))
    t.record_step(mv[2:], np.array([7 | (2 << 8)], dtype=np.int32))
    assert ct.digest_of_bytes(t.canonical_bytes()) == t.prefix_digest
    assert (t.total_pulls, t.step_index) == (3, 2)


def test_one_pull_framed_by_hand():
    """A list_change pull written out byte by byte from append_canonical_bytes (candidate_trace.rs:778-812,724-739)."""
    q = lambda v: struct.pack("<Q", v)
    s = lambda x: q(len(x)) + x
    expect = (b"\x45" + q(5)            # 0x45, ordinal
              + b"\x02"                 # CandidateTraceSource::LocalSearch
              + q(1) + s(b"Local Search") + q(9)   # phase_index, phase_type, step_index
              + b"\x01" + q(1)          # selector_index Some(1)
              + q(0)                    # candidate_index
              + b"\x00"                 # construction_target None
              + b"\x01"                 # identity Some
              + b"\x4f" + q(0) + b"\x01" + s(b"visits") + s(b"list_change")
              + q(5) + b"".join(b"\x01" + q(v) for v in (2, 1, 2, 4, 3))   # intra move: adjusted destination 4 - 1
              + q(3) + bytes([2, 8, 9]))  # Evaluated, Selected, Applied
    t = ct.CandidateTrace(phase_index=1)
    t.total_pulls, t.step_index = 5, 9
    mv = np.zeros(1, dtype=sfa.MOVE_DTYPE)
    mv["kind"], mv["a"], mv["a_pos"], mv["b"], mv["b_pos"], mv["value"] = 2, 2, 1, 2, 4, -1
    t.record_step(mv, np.array([7 | (1 << 8)], dtype=np.int32))
    assert t.canonical_bytes() == expect
    # to-None scalar change: the value coordinate is Absent (tag 2, no payload)
    t2 = ct.CandidateTrace(scalar_descriptor=3, scalar_variable="machine_idx")
    mv["kind"], mv["a"], mv["value"] = 0, 17, -1
    t2.record_step(mv, np.array([1], dtype=np.int32))
    tail = b"\x4f" + q(3) + b"\x01" + s(b"machine_idx") + s(b"scalar_change") + q(2) + b"\x01" + q(17) + b"\x02" + q(2) + bytes([2, 6])
    assert t2.canonical_bytes().endswith(tail)


def test_unknown_move_kind_is_rejected():
    t = ct.CandidateTrace()
    mv = np.zeros(1, dtype=sfa.MOVE_DTYPE)
    mv["kind"] = 42
    with pytest.raises(sfa.SolverForgeError):
        t.record_step(mv, np.array([1], dtype=np.int32))


def _oracle_models():
    p = datasets.make_cvrp(40, 5, 40, seed=2)
    o = sfo.Model.cvrp(p["capacity"], p["depot"], p["demands"], p["matrix"], p["customers"], p["routes"])
    o.set_kopt(1, 5)
    o.set_sublist_sizes(1, 3)
    leaves = ("nearby_change", "nearby_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt")
    o.configure(acceptor=1, la_size=5, forager=0, limit=40, leaves=sum(BITS[x] for x in leaves), selection_order=3,
                random_seed=4, max_nearby=10)
    yield "cvrp", o, dict(list_scope=(0, "visits")), p, leaves
    j = datasets.construct_jobshop(datasets.make_jobshop(6, 4), seed=1)
    oj = sfo.Model.jobshop(j["job"], j["machine_idx"], j["sequences"], bendable=True)
    oj.set_kopt(1, 0)
    jl = ("list_change", "list_swap", "change", "swap")
    oj.configure(acceptor=1, la_size=5, forager=0, limit=30, leaves=sum(BITS[x] for x in jl), selection_order=3, random_seed=2)
    yield "jobshop", oj, dict(list_scope=(1, "sequence"), scalar_scope=(0, "machine_idx")), j, jl


def test_native_encoder_matches_restatement_on_oracle_traces():
    """Every move family, selector indices of a 6-leaf union, all four disposition shapes."""
    seen_kinds, seen_disp = set(), set()
    for name, o, scopes, _, _ in _oracle_models():
        ls, ss = scopes.get("list_scope", (0, "visits")), scopes.get("scalar_scope", (0, "value"))
        native = ct.CandidateTrace(list_descriptor=ls[0], list_variable=ls[1], scalar_descriptor=ss[0], scalar_variable=ss[1])
        ref = trace_v3.Trace(list_scope=ls, scalar_scope=ss)
        o.phase_start()
        for _ in range(12):
            mv, _, fl, applied, _ = o.step_traced()
            assert int(((fl & 4) != 0).sum()) == (1 if applied else 0)  # exactly the committed pick is Selected
            assert ((fl & 4) == 0).all() or (fl[(fl & 4) != 0] & 3 == 3).all()
            native.record_step(mv, fl)
            ref.record_step(t6(mv), fl)
            seen_kinds |= set(mv["kind"].tolist())
            seen_disp |= set((fl & 7).tolist())
        assert native.canonical_bytes() == ref.canonical_bytes(), name
        assert native.prefix_digest == ref.digest.value(), name
        assert native.total_pulls == ref.total_pulls > 0
    assert seen_kinds == {0, 1, 2, 3, 4, 5, 6, 7}
    assert seen_disp >= {0, 1, 3, 7}


SHOP_LEAVES = ("precedence", "permute", "list_change", "list_swap", "ruin")


def _shop_oracle(policy):
    p = datasets.make_precedence_shop(7, 4, seed=5)
    o = sfo.Model.precedence_shop(p["durations"], p["successors"], p["sequences"], p["expected_owner"])
    if policy:
        o.set_precedence_policy(True)
    o.configure(acceptor=1, la_size=5, forager=0, limit=30, leaves=sum(BITS[x] for x in SHOP_LEAVES), selection_order=3, random_seed=6)
    return p, o


@pytest.mark.parametrize("policy", [False, True])
def test_native_encoder_on_precedence_shop_traces(policy):
    """The critical-path leaf's families: list_ruin (single- and two-source, with hooks), list_multi_swap (rejected by the
    score-improvement gate before the acceptor, evaluation.rs:95-113), list_permute."""
    _, o = _shop_oracle(policy)
    native = ct.CandidateTrace(list_descriptor=0, list_variable="sequence")
    ref = trace_v3.Trace(list_scope=(0, "sequence"))
    kinds, gated, multi = set(), 0, 0
    o.phase_start()
    for _ in range(15):
        mv, _, fl, _, _ = o.step_traced()
        native.record_step(mv, fl)
        ref.record_step(t6(mv), fl)
        kinds |= set(mv["kind"].tolist())
        gated += int(((fl & 16) != 0).sum())
        assert ((fl[(fl & 16) != 0] & 7) == 1).all() and (mv["kind"][(fl & 16) != 0] == 10).all()
        multi += int(((mv["kind"] == 8) & ((mv["value"].astype(np.int64) & 0xC0000000) == 0xC0000000)).sum())
    assert native.canonical_bytes() == ref.canonical_bytes()
    assert native.prefix_digest == ref.digest.value()
    assert {8, 9, 10} <= kinds and gated > 0 and multi > 0


def test_gate_dispositions_and_new_identities_framed_by_hand():
    q = lambda v: struct.pack("<Q", v)
    s = lambda b: q(len(b)) + b
    mv = np.zeros(4, dtype=sfa.MOVE_DTYPE)
    fl = np.array([1 | 8, 1 | 16, 3, 3], dtype=np.int32)
    # two swaps on list 2: (1, 4) and (6, 5)
    mv[0] = (10, 2, 2 | (1 << 16), 2 | (6 << 16), 0, 3 | (0xFF << 8))
    # two-source ruin with hooks: positions 3, 1 of list 4 and position 2 of list 1 -> sources merged, lists ascending
    mv[1] = (8, 4, 3, 3 | (1 << 16), 2, np.int32(np.uint32(0xC0000000 | (1 << 16)).view(np.int32)))
    mv[2] = (8, 0, 2, 5 | (2 << 16), 0, 0)  # plain ruin: positions ascending
    mv[3] = (9, 1, 2, 1, 5, 3)  # window [2, 5), rank 3 of 3! -> (1, 2, 0)
    t = ct.CandidateTrace(list_descriptor=1, list_variable="seq")
    t.record_step(mv, fl)
    coords = lambda xs: q(len(xs)) + b"".join(b"\x01" + q(x) for x in xs)
    ident = lambda fam, xs: b"\x4f" + q(1) + b"\x01" + s(b"seq") + s(fam) + coords(xs)
    body = t.canonical_bytes()
    e0 = ident(b"list_multi_swap", [2, 1, 4, 2, 6, 5]) + q(2) + bytes([2, 4])
    e1 = ident(b"list_ruin", [1, 1, 2, 4, 2, 1, 3]) + q(2) + bytes([2, 5])
    e2 = ident(b"list_ruin", [0, 2, 2, 5]) + q(2) + bytes([2, 7])
    e3 = ident(b"list_permute", [1, 2, 5, 3, 1, 2, 0]) + q(2) + bytes([2, 7])
    for e in (e0, e1, e2, e3):
        assert e in body, e
    ref = trace_v3.Trace(list_scope=(1, "seq"))
    ref.record_step(t6(mv), fl)
    assert ref.canonical_bytes() == body


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [False, True])
def test_gpu_trace_digest_precedence_shop(policy):
    p, o = _shop_oracle(policy)
    d = sfa.build_precedence_shop(p, leaves=SHOP_LEAVES, precedence_policy=policy)
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=5, forager=0, accepted_count_limit=30, selection_order=3, random_seed=6))
    gpu = ct.CandidateTrace(list_descriptor=0, list_variable="sequence", keep_bytes=False)
    cpu = trace_v3.Trace(list_scope=(0, "sequence"))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(20):
        gm, _, gf, gap, _ = d.solve_step_traced(cap=1 << 18)
        om, _, of, oap, _ = o.step_traced()
        assert (gf == of).all(), step
        gpu.record_step(gm, gf)
        cpu.record_step(t6(om), of)
        assert gpu.prefix_digest == cpu.digest.value(), (policy, step)
        assert gap == oap
    assert gpu.total_pulls == cpu.total_pulls > 0


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [0, 1, 2])
def test_gpu_trace_digest_equals_oracle_digest(engine):
    """The whole pull stream of a device run — order, identities, selector indices, dispositions, the committed pick —
    compared with the CPU oracle's as one 128-bit digest (and step by step, so a mismatch is localised)."""
    for name, o, scopes, data, leaves in _oracle_models():
        if name == "cvrp":
            if engine:  # the wave / block engines run the two nearby leaves
                leaves = ("nearby_change", "nearby_swap")
                o.configure(acceptor=1, la_size=5, forager=0, limit=40, leaves=48, selection_order=3, random_seed=4, max_nearby=10)
            d = sfa.build_cvrp(data, leaves=leaves, max_nearby=10, kopt=(1, 5), sublist_sizes=(1, 3))
            if engine:
                d.set_engine(engine)
            cfg = sfa.SolverConfig(acceptor=1, late_acceptance_size=5, forager=0, accepted_count_limit=40, selection_order=3, random_seed=4)
        else:
            if engine:
                continue
            d = sfa.build_jobshop(data, leaves=leaves)
            cfg = sfa.SolverConfig(acceptor=1, late_acceptance_size=5, forager=0, accepted_count_limit=30, selection_order=3, random_seed=2)
        d.configure(cfg)
        ls, ss = scopes.get("list_scope", (0, "visits")), scopes.get("scalar_scope", (0, "value"))
        gpu = ct.CandidateTrace(list_descriptor=ls[0], list_variable=ls[1], scalar_descriptor=ss[0], scalar_variable=ss[1], keep_bytes=False)
        cpu = trace_v3.Trace(list_scope=ls, scalar_scope=ss)
        d.calculate_score()
        d.phase_start()
        o.phase_start()
        for step in range(25):
            gm, _, gf, gap, _ = d.solve_step_traced(cap=1 << 18)
            om, _, of, oap, _ = o.step_traced()
            gpu.record_step(gm, gf)
            cpu.record_step(t6(om), of)
            assert gpu.prefix_digest == cpu.digest.value(), (name, step)
            assert gap == oap
        assert gpu.total_pulls == cpu.total_pulls


@pytest.mark.gpu
@pytest.mark.parametrize("forager", [0, 1, 2])
def test_gpu_trace_digest_scalar_engine(forager):
    g = datasets.make_graph(60, 150, 4, seed=3)
    g["colors"] = (datasets.stream(12, 60) % np.uint64(5)).astype(np.int64) - 1
    d = sfa.build_graph_coloring(g, leaves=("change", "swap"))
    o = sfo.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    o.configure(acceptor=1, la_size=3, forager=forager, limit=7, leaves=3, selection_order=3, random_seed=5)
    d.configure(sfa.SolverConfig(acceptor=1, late_acceptance_size=3, forager=forager, accepted_count_limit=7, selection_order=3, random_seed=5))
    gpu = ct.CandidateTrace(scalar_variable="color", keep_bytes=False)
    cpu = trace_v3.Trace(scalar_scope=(0, "color"))
    d.calculate_score()
    d.phase_start()
    o.phase_start()
    for step in range(6):
        gm, _, gf, _, _ = d.solve_step_traced(cap=1 << 18)
        om, _, of, _, _ = o.step_traced()
        gpu.record_step(gm, gf)
        cpu.record_step(t6(om), of)
        assert gpu.prefix_digest == cpu.digest.value(), (forager, step)
    assert gpu.total_pulls == cpu.total_pulls > 0


def test_native_encoder_matches_committed_fixture():
    """tests/golden/candidate_trace_v3.json (made by tests/golden/make_candidate_trace_golden.py): every move family (ruins with
    one and two sources, permutations and multi-swaps included), all six disposition shapes, selector indices 0..8, two scopes."""
    import json
    import os

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "candidate_trace_v3.json")))
    sc = fx["scope"]
    t = ct.CandidateTrace(phase_index=sc["phase_index"], phase_type=sc["phase_type"], list_descriptor=sc["list"][0],
                          list_variable=sc["list"][1], scalar_descriptor=sc["scalar"][0], scalar_variable=sc["scalar"][1])
    for step in fx["steps"]:
        mv = np.zeros(len(step), dtype=sfa.MOVE_DTYPE)
        for i, pull in enumerate(step):
            mv[i] = tuple(pull["move"])
        t.record_step(mv, np.array([pull["flags"] for pull in step], dtype=np.int32))
    assert t.canonical_bytes().hex() == fx["canonical_bytes_hex"]
    assert [str(v) for v in t.prefix_digest] == fx["prefix_digest"]
    assert t.total_pulls == fx["total_pulls"]


def test_applied_kopt_pull_with_empty_scalar_variable_name():
    """The largest pull (an applied 3-opt move: 10 coordinates + three dispositions) must frame when the scope carries
    no scalar variable name — a list-only model bound from a Rust shim passes NULL / "" there."""
    mv = np.zeros(1, dtype=sfa.MOVE_DTYPE)
    mv["kind"], mv["a"], mv["a_pos"], mv["b"], mv["b_pos"], mv["value"] = 7, 3, 2, 5, 9, 4
    flags = np.array([7 | (5 << 8)], dtype=np.int32)
    for scalar_name in ("", "value"):
        t = ct.CandidateTrace(list_variable="v", scalar_variable=scalar_name, phase_type="")
        t.record_step(mv, flags)
        ref = trace_v3.Digest().update(t.canonical_bytes()).value()
        assert t.prefix_digest == ref
        assert len(t.canonical_bytes()) >= 190
