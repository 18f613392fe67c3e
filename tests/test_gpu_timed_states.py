"""Parity AT THE STATES THE BENCHMARK SCRIPTS TIME: the post-construction start of BASELINE
config 4 (mixed job shop 500 x 20: 20 machine sequences of ~500 operations, every operation assigned) under the 4-leaf and
the 8-leaf union, and of config 2 (graph colouring 10k / 100k after first-fit construction) under the reference's default
scalar policy (auto-calibrated SimulatedAnnealing + AcceptedCount(1)): traced candidate order / scores / decisions on the
first steps, then a fused window, against the CPU oracle.  Plus the portfolio exchange between two processes over RCCL."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied", "score_calculations",
            "moves_not_doable"]


def _t(m):
    return np.stack([m["kind"], m["a"], m["a_pos"], m["b"], m["b_pos"], m["value"]], axis=1)


@pytest.mark.parametrize("leaves,bits,traced_steps", [
    (("list_change", "list_swap", "change", "swap"), 4 | 8 | 1 | 2, 12),
    (("list_change", "list_swap", "sublist_change", "sublist_swap", "list_reverse", "kopt", "change", "swap"),
     4 | 8 | 128 | 256 | 64 | 512 | 1 | 2, 10),
])
def test_c4_constructed_jobshop_union_matches_oracle(oracle, leaves, bits, traced_steps):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    p = datasets.construct_jobshop(datasets.make_jobshop(500, 20))  # scripts/jobshop_bench.py's start state
    assert min(len(s) for s in p["sequences"]) > 400
    d = sfa.build_jobshop(p, n_replicas=2, leaves=leaves)
    o = oracle.Model.jobshop(p["job"], p["machine_idx"], p["sequences"], bendable=True)
    if "kopt" in leaves:
        o.set_kopt(1, 0)
    o.configure(leaves=bits, random_seed=0)
    d.configure(sfa.SolverConfig(random_seed=0))
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    d.phase_start()
    o.phase_start()
    for step in range(traced_steps):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 18)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om) > 0, step
        assert (_t(gm) == _t(om)).all(), step          # candidate order of the union cursor
        assert (gs == os_[:, :3]).all(), step          # trial scores
        assert (gf == of).all(), step                  # doable / accepted / selector index / pick
        assert gap == oap, step
        if gap:
            assert tuple(gmv) == tuple(omv), step
    d.solve_steps(25)  # fused launch from the traced state
    o.steps(25)
    assert (d.calculate_score()[0] == o.score()[:3]).all()
    assert (d.fresh_score()[0] == o.score()[:3]).all()
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    assert d.working_lists(1, 0) == o.get_lists(1)
    gst, ost = d.stats(0), o.stats()
    for k in COUNTERS:
        assert gst[k] == ost[k], k


def test_c2_constructed_graph_default_scalar_policy_matches_oracle(oracle):
    import solverforge_amd as sfa
    from solverforge_amd import datasets

    g = datasets.construct_graph(datasets.make_graph(10000, 100000, 16, seed=0))  # scripts/graph_bench.py's start state
    d = sfa.build_graph_coloring(g, n_replicas=2)
    d.configure(sfa.SolverConfig(acceptor=sfa.Acceptor.SIMULATED_ANNEALING, accepted_count_limit=1, random_seed=0))
    o = oracle.Model.graph_coloring(g["n_colors"], g["adj_off"], g["adj"], g["colors"])
    o.configure(leaves=oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP, random_seed=0, limit=1)
    o.configure_annealing(seed=0)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    d.phase_start()
    o.phase_start()
    for step in range(40):
        gm, gs, gf, gap, gmv = d.solve_step_traced(cap=1 << 17)
        om, os_, of, oap, omv = o.step_traced()
        assert len(gm) == len(om), step
        assert (_t(gm)[:, [0, 1, 3, 5]] == _t(om)[:, [0, 1, 3, 5]]).all(), step
        assert (gs == os_[:, :2]).all() and (gf == of).all(), step
        assert gap == oap, step
    d.solve_steps(400)
    o.steps(400)
    assert (d.calculate_score()[0] == o.score()[:2]).all()
    assert (d.working_values(0, 0) == o.get_vars(0, 0)).all()
    gt, gc = d.annealing_state(0)
    ot, _, oc = o.annealing_state()
    assert gc == bool(oc) and (gt == ot[:2]).all()      # calibrated temperatures, bit-equal f64
    gst, ost = d.stats(0), o.stats()
    for k in COUNTERS:
        assert gst[k] == ost[k], k


def _rccl_rank(rank, world, uid_q, out_q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import solverforge_amd as sfa
        from solverforge_amd import datasets, portfolio

        p = datasets.make_cvrp(60, 6, 55, seed=3)
        d = sfa.build_cvrp(p, n_replicas=4)
        d.configure(sfa.SolverConfig(random_seed=portfolio.rank_seed_base(9, rank, 4)))
        d.calculate_score()
        d.phase_start()
        d.solve_steps(30)
        local_best = max(tuple(int(v) for v in s) for s in d.best_scores())
        if rank == 0:
            uid = d.portfolio_unique_id()
            for _ in range(world - 1):
                uid_q.put(uid.tobytes())
        else:
            uid = np.frombuffer(uid_q.get(timeout=60), dtype=np.uint8).copy()
        d.portfolio_init(uid, rank, world)
        best, wr, wrep = d.portfolio_allgather_best()
        routes = d.portfolio_broadcast_best(wr, wrep)
        d.portfolio_destroy()
        out_q.put((rank, "ok", local_best, tuple(int(v) for v in best), int(wr), int(wrep), sum(len(r) for r in routes)))
    except Exception as e:  # reported to the parent: the test decides between fail and skip
        out_q.put((rank, "error", f"{type(e).__name__}: {e}"))


def test_portfolio_exchange_between_two_processes_over_rccl():
    """World size 2 on ONE GPU (the box has one): two processes, one RCCL communicator, ncclAllGather of the best scores
    + ncclBroadcast of the winner's routes.  RCCL may refuse two ranks on one device; then the test skips with its reason
    (the same code path runs with one rank per GPU in bench.py --gpus N)."""
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rccl_rank, args=(r, 2, uid_q, out_q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = []
    try:
        for _ in range(2):
            res.append(out_q.get(timeout=150))
    except Exception:
        for pr in procs:
            pr.kill()
        pytest.skip("RCCL did not bring a 2-rank communicator up on one device within 150 s")
    for pr in procs:
        pr.join(timeout=30)
        if pr.is_alive():
            pr.kill()
    errors = [r for r in res if r[1] != "ok"]
    if errors:
        pytest.skip(f"RCCL refused two ranks on one device: {errors[0][2]}")
    res.sort()
    locals_ = [r[2] for r in res]
    expect = max(locals_)
    for r in res:
        assert r[3] == expect                          # every rank names the same best score
        assert r[4] == locals_.index(expect)           # ... and the same winner (ties: lowest rank)
        assert r[6] == 60                              # the winner's routes arrived: every customer once
    assert res[0][4] == res[1][4] and res[0][5] == res[1][5]
