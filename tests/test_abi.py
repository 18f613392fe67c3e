"""CPU test: the C-ABI library loads and exports every symbol include/solverforge_amd.h declares
(no compute calls without a GPU), and fails loudly when no device exists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "solverforge_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g

    g.build()
    from solverforge_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SYMBOLS) == declared


def test_fails_loudly_without_device():
    import solverforge_amd as sfa
    from solverforge_amd import _lib

    if _lib.load().sf_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(sfa.SolverForgeError, match="NO_DEVICE"):
        sfa.GpuScoreDirector()
