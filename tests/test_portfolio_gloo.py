"""CPU tests of the multi-GPU (N>1) host path: world_size-2 gloo rendezvous, seed sharding, the
lexicographic best-score exchange rule (identical winner on every rank) and the max/sum-over-ranks
reductions bench.py uses.  The RCCL exchange itself needs GPUs; it applies the same winner rule."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, scores, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from solverforge_amd import portfolio

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wr, ws = portfolio.gloo_allgather_best(dist, scores[rank], rank, world)
        mx = portfolio.max_over_ranks(dist, 1.5 + rank)
        sm = portfolio.sum_over_ranks(dist, 10.0 * (rank + 1))
        dist.barrier()
        out.put((rank, wr, ws, mx, sm, portfolio.rank_seed_base(7, rank, 4096)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scores,expect_rank", [
    ([[0, -120], [0, -100]], 1),       # better soft wins
    ([[0, -100], [-1, -5]], 0),        # hard level dominates
    ([[0, -100], [0, -100]], 0),       # tie keeps the lowest rank
])
def test_world2_exchange_names_one_winner(scores, expect_rank):
    import multiprocessing as mp  # workers import torch; the pytest process stays torch-free

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scores, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, wr, ws, mx, sm, seed in res:
        assert wr == expect_rank
        assert ws == scores[expect_rank]
        assert mx == 2.5 and sm == 30.0
        assert seed == 7 + rank * 4096


def test_pick_winner_rule():
    from solverforge_amd import portfolio

    assert portfolio.pick_winner([[0, -5], [0, -5], [0, -4]]) == (2, [0, -4])
    assert portfolio.pick_winner([[-1, 0], [-2, 100]]) == (0, [-1, 0])
    assert portfolio.pick_winner([[0, 0, -3], [0, 0, -3]]) == (0, [0, 0, -3])
    assert portfolio.better([0, -1], [-1, 1000])


def _bench_leg_replicas():
    """The replica counts bench.py itself uses for its legs (imported, not restated): M1, every M2 policy, the tuned leg, the C5 side leg."""
    sys.path.insert(0, ROOT)
    import bench

    return sorted({bench.M1_REPLICAS, bench.C5_REPLICAS, bench.TUNED["replicas"], *bench.M2_REPLICAS.values()})


@pytest.mark.parametrize("replicas", _bench_leg_replicas())
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_seed_ranges_tile_without_overlap(replicas, world):
    """Every portfolio member of a leg has its own seed: the per-rank ranges are disjoint and consecutive, and rank 0 / replica 0
    is the single-GPU search (same seed as an N = 1 run)."""
    sys.path.insert(0, ROOT)
    from solverforge_amd import portfolio

    ranges = [portfolio.rank_seed_range(0, q, replicas) for q in range(world)]
    assert ranges[0].start == 0
    for a, b in zip(ranges, ranges[1:]):
        assert a.stop == b.start and len(a) == replicas
    seen = set()
    for rg in ranges:
        assert seen.isdisjoint(rg)
        seen.update(rg)
    assert len(seen) == world * replicas
    # what bench.py passes to SolverConfig(random_seed=...) is the start of the rank's range
    assert [portfolio.rank_seed_base(0, q, replicas) for q in range(world)] == [rg.start for rg in ranges]
