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


def test_device_sources_pass_the_codegen_lint(tmp_path):
    """scripts/lint_device_patterns.py: no struct returned by value from a __noinline__ device function in
    csrc/, and the lint does fire on that shape."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lint_device_patterns", os.path.join(root, "scripts", "lint_device_patterns.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    import glob

    src = sorted(glob.glob(os.path.join(root, "solverforge_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "solverforge_amd", "csrc", "*.hip"))
                 + glob.glob(os.path.join(root, "solverforge_amd", "csrc", "*.inc")))
    assert lint.findings(src) == []
    bad = tmp_path / "bad.h"
    bad.write_text("struct Pick { int a[8]; };\n__device__ __noinline__ Pick best_slot(int x) { Pick p{}; return p; }\n"
                   "__device__ void f(bool c, Pick& o) { o = c ? best_slot(1) : best_slot(2); }\n")
    got = lint.findings([str(bad)])
    assert len(got) == 2 and "by value" in got[0] and "conditional expression" in got[1]
