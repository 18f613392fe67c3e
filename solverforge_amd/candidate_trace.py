"""Candidate-trace export, wire format v3 (reference: crates/solverforge-solver/src/stats/candidate_trace.rs).

`CandidateTrace` mirrors the part of `CandidateTraceTelemetry` that proves work equivalence: the ordered pulls of a
local-search phase in canonical bytes and their two-lane `prefix_digest` (candidate_trace.rs:975-984).  The framing
and the digest run in the native library (`sf_trace_encode_step`); this class only carries the counters
(`total_pulls`, the phase's step index) from one traced step to the next."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MOVE_DTYPE, SolverForgeError, TraceDigestStruct, TraceScopeStruct, ptr

FORMAT_VERSION = 3  # CANDIDATE_TRACE_FORMAT_VERSION (candidate_trace.rs:15)


class CandidateTrace:
    def __init__(self, phase_index=0, phase_type="Local Search", list_descriptor=0, list_variable="visits",
                 scalar_descriptor=0, scalar_variable="value", keep_bytes=True):
        self._L = _lib.load()
        self._strings = [s.encode() for s in (phase_type, list_variable, scalar_variable)]  # keep the char* alive
        self._scope = TraceScopeStruct(phase_index, self._strings[0], list_descriptor, self._strings[1],
                                       scalar_descriptor, self._strings[2])
        self._digest = TraceDigestStruct()
        self._L.sf_trace_digest_init(C.byref(self._digest))
        self.total_pulls = 0
        self.step_index = 0
        self._chunks = [] if keep_bytes else None

    def record_step(self, moves, flags):
        """Frames the pulls of one traced step (`GpuScoreDirector.solve_step_traced` output)."""
        moves = np.ascontiguousarray(moves, dtype=MOVE_DTYPE)
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        n = len(moves)
        if len(flags) != n:
            raise SolverForgeError("moves and flags differ in length")
        need = self._L.sf_trace_encode_step(C.byref(self._scope), self.total_pulls, self.step_index, ptr(moves),
                                            ptr(flags), n, None, 0, None)
        if need < 0:
            raise SolverForgeError(f"sf_trace_encode_step: {_lib.ERRORS.get(need, need)}")
        out = np.zeros(max(need, 1), dtype=np.uint8)
        got = self._L.sf_trace_encode_step(C.byref(self._scope), self.total_pulls, self.step_index, ptr(moves),
                                           ptr(flags), n, ptr(out), need, C.byref(self._digest))
        if got != need:
            raise SolverForgeError(f"sf_trace_encode_step: {_lib.ERRORS.get(got, got)}")
        if self._chunks is not None:
            self._chunks.append(out[:need].tobytes())
        self.total_pulls += n
        self.step_index += 1

    @property
    def prefix_digest(self):
        return int(self._digest.first), int(self._digest.second)

    def canonical_bytes(self):
        if self._chunks is None:
            raise SolverForgeError("trace was created with keep_bytes=False")
        return b"".join(self._chunks)


def digest_of_bytes(data):
    """CandidateTraceDigest::of_bytes (candidate_trace.rs:42-46)."""
    L = _lib.load()
    d = TraceDigestStruct()
    L.sf_trace_digest_init(C.byref(d))
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    L.sf_trace_digest_update(C.byref(d), ptr(buf) if len(buf) else None, len(buf))
    return int(d.first), int(d.second)
