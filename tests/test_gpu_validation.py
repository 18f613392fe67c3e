"""C-ABI argument validation and context independence on the device: bad host inputs come back as
SF_ERR_INVALID instead of reaching device indexing; contexts are independent of the calling thread's current device."""
import threading

import numpy as np
import pytest

import solverforge_amd as sfa
from solverforge_amd import datasets
from solverforge_amd.director import ConstraintKind, GpuScoreDirector, SelectorKind

pytestmark = pytest.mark.gpu


def test_list_variable_csr_is_validated():
    d = GpuScoreDirector(n_replicas=1)
    d.add_entity_class(0, 2)
    L = d._L
    bad_start = np.array([1, 2, 3], dtype=np.uint32)
    vals = np.array([1, 2, 3], dtype=np.uint32)
    assert L.sf_schema_add_list_variable(d._h, 0, sfa._lib.ptr(bad_start), sfa._lib.ptr(vals), 8, 8) == -1
    not_monotonic = np.array([0, 2, 1], dtype=np.uint32)
    assert L.sf_schema_add_list_variable(d._h, 0, sfa._lib.ptr(not_monotonic), sfa._lib.ptr(vals), 8, 8) == -1
    ok_off = np.array([0, 1, 3], dtype=np.uint32)
    assert L.sf_schema_add_list_variable(d._h, 0, sfa._lib.ptr(ok_off), None, 8, 8) == -1           # values missing
    huge = np.array([1, 0x80000001, 3], dtype=np.uint32)                                                # id >= 2^31
    assert L.sf_schema_add_list_variable(d._h, 0, sfa._lib.ptr(ok_off), sfa._lib.ptr(huge), 8, 8) == -1
    assert L.sf_schema_add_list_variable(d._h, 0, sfa._lib.ptr(ok_off), sfa._lib.ptr(vals), 8, 8) == 0
    d.close()


def test_fact_csr_and_depot_and_scalar_length_are_validated():
    d = GpuScoreDirector(n_replicas=1)
    L = d._L
    off = np.array([0, 3, 2], dtype=np.uint32)
    vals = np.array([0, 1, 2], dtype=np.uint32)
    assert L.sf_fact_csr_u32(d._h, 3, 2, sfa._lib.ptr(off), sfa._lib.ptr(vals)) == -1
    off2 = np.array([0, 1, 3], dtype=np.uint32)
    assert L.sf_fact_csr_u32(d._h, 3, 2, sfa._lib.ptr(off2), None) == -1
    d.add_entity_class(0, 4)
    with pytest.raises(sfa.SolverForgeError, match="one value per row"):
        d.add_scalar_variable(0, 0, 3, True, np.zeros(3, dtype=np.int32))
    d.close()
    p = datasets.make_cvrp(12, 3, 55, seed=1)
    for depot in (-1, 13):
        q = dict(p)
        q["depot"] = depot
        dd = sfa.build_cvrp(q, n_replicas=1)
        with pytest.raises(sfa.SolverForgeError, match="depot"):
            dd.calculate_score()
        dd.close()


def test_launch_limits_are_rejected():
    p = datasets.make_cvrp(30, 3, 55, seed=2)
    d = sfa.build_cvrp(p, n_replicas=2)
    d.calculate_score()
    d.configure(sfa.SolverConfig(random_seed=1))
    d.phase_start()
    with pytest.raises(sfa.SolverForgeError, match="2\\^31"):
        d.solve_moves(10, 1 << 31)
    with pytest.raises(sfa.SolverForgeError, match="2\\^31"):
        d.solve_steps(1 << 31)
    d.solve_moves(10, (1 << 31) - 1)  # the largest legal budget: ends by max_steps
    assert d.stats(0)["step_count"] == 10
    d.close()


def test_contexts_are_independent_of_the_calling_thread():
    """Two contexts driven alternately, one of them from a worker thread (the per-thread current device of HIP is
    (re)bound by every entry point): both must reproduce the single-context run bit for bit."""
    p = datasets.make_cvrp(40, 4, 55, seed=5)

    def run(seed, out, steps=15):
        d = sfa.build_cvrp(p, n_replicas=3)
        d.calculate_score()
        d.configure(sfa.SolverConfig(random_seed=seed))
        d.phase_start()
        for _ in range(steps):
            d.solve_steps(1)
        out[seed] = (d.calculate_score().copy(), d.working_lists(0, 1), d.stats(2))
        d.close()

    ref = {}
    run(7, ref)
    run(11, ref)
    got = {}
    d1 = sfa.build_cvrp(p, n_replicas=3)
    d1.calculate_score()
    d1.configure(sfa.SolverConfig(random_seed=7))
    d1.phase_start()
    th = threading.Thread(target=run, args=(11, got))
    th.start()
    for _ in range(15):
        d1.solve_steps(1)
    th.join()
    got[7] = (d1.calculate_score().copy(), d1.working_lists(0, 1), d1.stats(2))
    d1.close()
    for seed in (7, 11):
        assert (got[seed][0] == ref[seed][0]).all()
        assert got[seed][1] == ref[seed][1]
        assert got[seed][2] == ref[seed][2]
