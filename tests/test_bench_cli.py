"""CPU tests of bench.py's launcher contract: `--gpus N` must never silently run fewer ranks than asked for, and the
product path must refuse to run without a HIP device (no CPU fallback)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _device_count():
    import __graft_entry__ as g

    g.build()
    from solverforge_amd import _lib

    return _lib.load().sf_device_count()


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_gpus_n_refuses_to_run_with_fewer_devices():
    n = _device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0", "--solve-seconds", "0", "--no-pmc"])
    assert r.returncode == 2
    assert b"refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == b""  # no JSON line: nothing to mistake for an N-GPU result


def test_c5_portfolio_refuses_to_run_with_fewer_devices():
    """BASELINE config 5 (CVRP-5000, the 8-GPU portfolio): the same refusal, before any problem data is built."""
    n = _device_count()
    r = _run(["--customers", "5000", "--vehicles", "500", "--replicas", "1280", "--gpus", str(max(n, 0) + 2), "--steps", "1", "--warmup", "0",
              "--solve-seconds", "0", "--no-pmc"])
    assert r.returncode == 2
    assert b"refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == b""


def test_gpus_flag_must_match_the_launcher():
    r = _run(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    assert b"refusing to report one as the other" in r.stderr


def test_no_device_no_number():
    if _device_count() > 0:
        pytest.skip("a HIP device is present")
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert b"SF_ERR_NO_DEVICE" in r.stderr
    assert r.stdout.strip() == b""


def test_launch_shapes_are_whole_residencies():
    """The replica counts bench.py launches by default fill the 256 CUs evenly (round 6: the legs run many residencies per launch; a count that is not a
    multiple of one residency would leave CUs idle at the end of every launch), and the counter window of the M2 leg lies inside a short leg."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.M1_REPLICAS % (24 * 256) == 0  # wave engine, COMPACT slice at CVRP-1000: 24 replicas per CU
    assert b.M2_REPLICAS["default"] % (12 * 256) == 0  # generic engine, FAST + RUIN: 12 per CU
    assert b.M2_REPLICAS["default6"] % (16 * 256) == 0  # six-leaf FAST: 16 per CU
    assert b.C5_REPLICAS % (11 * 256) == 0  # wave engine, launch mode 6 at CVRP-5000: 11 per CU
    assert b.PMC_CHILD_MAX_REPLICAS % (24 * 256) == 0
    assert 0 < b.M2_PMC_TIMED <= b.M2_PMC_WARM  # the window [WARM, WARM + TIMED) of launches: past the construction, a few launches long
    assert [c for p in b.M2_PMC_PASSES for c in p].count("SQ_INSTS_SALU") == 1
