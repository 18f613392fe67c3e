"""TEST INFRASTRUCTURE (oracle): an independent, pure-Python restatement of the reference's candidate-trace wire
format v3 — `CandidatePullTelemetry::append_canonical_bytes`, `CandidateTraceIdentity::append_canonical_bytes`,
the framing helpers and the two-lane `CandidateTraceDigest` of
crates/solverforge-solver/src/stats/candidate_trace.rs (:24-61, :718-812, :1000-1064).  Only tests/ may import it;
the product path frames traces in the native library (csrc/sf_candidate_trace.inc).

Parity pin: the reference holds NO literal digest or byte golden for this format (its tests compare digests of two
runs of the same code, stats/tests.rs:326).  Lane one of the digest is FNV-1a 64 and is checked against the
published FNV test vectors; lane two and the framing are restated from source: **parity unpinned** beyond that."""
import struct

M64 = (1 << 64) - 1


class Digest:  # candidate_trace.rs:24-61
    def __init__(self):
        self.first = 0xCBF29CE484222325
        self.second = 0x9E3779B97F4A7C15

    def update(self, data):
        a, s = self.first, self.second
        for byte in data:
            a ^= byte
            a = (a * 0x00000100000001B3) & M64
            s ^= (byte + 0x9D) & M64
            s = ((s << 13) | (s >> 51)) & M64
            s = (s * 0xD6E8FEB86659FD93 + 0x9E3779B9) & M64
        self.first, self.second = a, s
        return self

    def value(self):
        return self.first, self.second


def u64(v):  # append_u64 :1015-1017
    return struct.pack("<Q", v)


def string(s):  # append_string :1039-1042
    b = s.encode()
    return u64(len(b)) + b


def coordinates(values):  # append_coordinate_list :1044-1064 (None = Absent)
    out = u64(len(values))
    for v in values:
        out += b"\x02" if v is None else b"\x01" + u64(v)
    return out


def logical_move(descriptor_index, variable_name, family, coords):  # :688-706, :724-739
    return b"\x4f" + u64(descriptor_index) + b"\x01" + string(variable_name) + string(family) + coordinates(coords)


SEGMENT_ORDER = [[0, 1, 2, 3]] * 3 + [[0, 2, 1, 3]] * 4  # k_opt_reconnection.rs:203-234


def identity(move, list_scope, scalar_scope):
    """move = (kind, a, a_pos, b, b_pos, value) in the oracle's wire form (sfo_move_t)."""
    kind, a, a_pos, b, b_pos, value = (int(x) for x in move)
    if kind == 0:  # scalar_neighborhood/move.rs:112-127
        return logical_move(*scalar_scope, "scalar_change", [a, None if value < 0 else value])
    if kind == 1:  # :128-138
        return logical_move(*scalar_scope, "scalar_swap", [a, b])
    if kind == 2:  # move/list_kernel/change.rs:205-226, adjusted_destination :28-34
        adjusted = b_pos - 1 if (a == b and b_pos > a_pos) else b_pos
        return logical_move(*list_scope, "list_change", [a, a_pos, b, b_pos, adjusted])
    if kind == 3:  # move/list_kernel/swap.rs:157-176
        return logical_move(*list_scope, "list_swap", [a, a_pos, b, b_pos])
    if kind == 4:  # move/list_kernel/reverse.rs:99-111
        return logical_move(*list_scope, "list_reverse", [a, a_pos, b_pos])
    if kind == 5:  # move/list_kernel/sublist_change.rs:194-213
        return logical_move(*list_scope, "sublist_change", [a, a_pos, value, b, b_pos])
    if kind == 6:  # move/list_kernel/sublist_swap.rs:257-278
        return logical_move(*list_scope, "sublist_swap",
                            [a, a_pos, a_pos + (value & 0xFFFF), b, b_pos, b_pos + (value >> 16)])
    if kind == 7:  # runtime/compiler/executor/list_leaf/move.rs:403-420
        return logical_move(*list_scope, "k_opt", [a, a_pos, a, b, a, b_pos] + SEGMENT_ORDER[value])
    if kind == 9:  # move/list_kernel/permute.rs:170-188: entity, start, end, len, permutation (nth_permutation of the rank)
        n = b_pos - a_pos
        remaining, perm, rank = list(range(n)), [], value
        for position in range(n):
            step = 1
            for t in range(2, n - position):
                step *= t
            perm.append(remaining.pop(rank // step))
            rank %= step
        return logical_move(*list_scope, "list_permute", [a, a_pos, b_pos, n] + perm)
    if kind == 8:  # runtime/compiler/executor/list_leaf/move.rs:421-433: (entity, count, positions) per source list
        u = lambda x: x & 0xFFFFFFFF
        pos = [(u(w) >> (16 * h)) & 0xFFFF for w in (b, b_pos, value) for h in (0, 1)]
        flagged = a_pos <= 5 and bool(u(value) & 0x80000000)
        # sources merged per list, positions ascending, lists ascending (heuristic/move/list_kernel/ruin.rs:27-52)
        if flagged and (u(value) & 0x40000000):  # the last element comes from the second list
            second = (u(value) >> 16) & 0x3FFF
            groups = sorted([(a, sorted(pos[: a_pos - 1])), (second, [pos[a_pos - 1]])], key=lambda g: g[0])
        else:
            groups = [(a, sorted(pos[:a_pos]))]
        return logical_move(*list_scope, "list_ruin", [x for e, idx in groups for x in [e, len(idx)] + idx])
    if kind == 10:  # list_leaf/move.rs:434-447: (entity, first, second) per swap
        coords = []
        for q, w in enumerate((a_pos, b, b_pos)[:a]):
            w &= 0xFFFFFFFF
            d = (value >> (8 * q)) & 0xFF
            first = w >> 16
            coords += [w & 0xFFFF, first, first + (d - 256 if d > 127 else d)]
        return logical_move(*list_scope, "list_multi_swap", coords)
    raise ValueError(f"move kind {kind}")


def dispositions(flag):  # candidates.rs:127-281, step.rs:122-147,227-243; codes candidate_trace.rs:536-549
    if not flag & 1:
        codes = [2, 3]  # Evaluated, NotDoable
    elif flag & 8:
        codes = [2, 4]  # Evaluated, RejectedByHardImprovement (evaluation.rs:75-93)
    elif flag & 16:
        codes = [2, 5]  # Evaluated, RejectedByScoreImprovement (evaluation.rs:95-113)
    elif not flag & 2:
        codes = [2, 6]  # Evaluated, AcceptorRejected
    elif not flag & 4:
        codes = [2, 7]  # Evaluated, ForagerIgnored
    else:
        codes = [2, 8, 9]  # Evaluated, Selected, Applied
    return u64(len(codes)) + bytes(codes)


def pull(ordinal, phase_index, phase_type, step_index, selector_index, candidate_index, ident, flag):  # :778-812
    return (b"\x45" + u64(ordinal) + b"\x02" + u64(phase_index) + string(phase_type) + u64(step_index)
            + b"\x01" + u64(selector_index) + u64(candidate_index) + b"\x00" + b"\x01" + ident + dispositions(flag))


class Trace:
    def __init__(self, phase_index=0, phase_type="Local Search", list_scope=(0, "visits"), scalar_scope=(0, "value")):
        self.phase_index, self.phase_type = phase_index, phase_type
        self.list_scope, self.scalar_scope = list_scope, scalar_scope
        self.total_pulls = 0
        self.step_index = 0
        self.digest = Digest()
        self.chunks = []

    def record_step(self, moves6, flags):
        for i, (mv, f) in enumerate(zip(moves6, flags)):
            f = int(f)
            b = pull(self.total_pulls + i, self.phase_index, self.phase_type, self.step_index, (f >> 8) & 0xFF, i,
                     identity(mv, self.list_scope, self.scalar_scope), f)
            self.digest.update(b)  # snapshot(): one update per pull (:975-984)
            self.chunks.append(b)
        self.total_pulls += len(flags)
        self.step_index += 1

    def canonical_bytes(self):
        return b"".join(self.chunks)
