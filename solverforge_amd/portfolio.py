"""Multi-GPU portfolio plumbing: one process per GPU, independent seeded searches,
no data-path collective; one best-score exchange at the end.

The exchange itself runs over RCCL/xGMI inside the C ABI (sf_portfolio_allgather_best: one
ncclAllGather of (score levels, rank, replica) + a local lexicographic max).  This module holds the
host-side pieces that do not need a GPU — seed sharding, the lexicographic winner rule, and the
torch.distributed (gloo) rendezvous helpers bench.py uses for barriers / max-over-ranks — so the
N>1 path is covered by world_size-2 CPU tests.  The reference has no counterpart (it is
single-process; SURVEY.md §2 "Parallelism strategies").
"""
from typing import Iterable, List, Sequence, Tuple


def rank_seed_base(base_seed: int, rank: int, replicas_per_rank: int) -> int:
    """Replica r of rank q searches with random_seed = base + q * replicas + r: every portfolio
    member in the job has a distinct seed, and rank 0 / replica 0 is the single-GPU search."""
    return int(base_seed) + int(rank) * int(replicas_per_rank)


def rank_seed_range(base_seed: int, rank: int, replicas_per_rank: int) -> range:
    """The seeds rank `rank` searches with in a leg of `replicas_per_rank` replicas per GPU: consecutive ranks tile the seed
    axis without gaps or overlap, whatever the replica count of the leg (bench.py runs legs of 24,576 / 2,048 / 1,024)."""
    lo = rank_seed_base(base_seed, rank, replicas_per_rank)
    return range(lo, lo + int(replicas_per_rank))


def better(a: Sequence[int], b: Sequence[int]) -> bool:
    """Lexicographic Score ordering, most significant level first
    (crates/solverforge-core/src/score/hard_soft.rs:130-137, bendable.rs:210-230)."""
    return tuple(int(v) for v in a) > tuple(int(v) for v in b)


def pick_winner(scores: Iterable[Sequence[int]]) -> Tuple[int, List[int]]:
    """Index and value of the best score; ties keep the lowest index, so every rank that evaluates
    the same gathered list names the same winner (the rule sf_portfolio_allgather_best applies)."""
    best_i, best = -1, None
    for i, s in enumerate(scores):
        if best is None or better(s, best):
            best_i, best = i, [int(v) for v in s]
    return best_i, best


def gloo_allgather_best(dist, local_best: Sequence[int], rank: int, world: int):
    """CPU exchange with the same semantics as the RCCL one (used by tests and as bench.py's
    reported fallback when RCCL cannot initialise): returns (winner_rank, winner_score)."""
    import torch

    t = torch.tensor([int(v) for v in local_best], dtype=torch.int64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    return pick_winner([g.tolist() for g in gathered])


def torch_rccl_allgather_best(dist, local_best: Sequence[int], rank: int, world: int, device_index: int):
    """The same exchange through torch.distributed's "nccl" backend (= RCCL over xGMI on ROCm) on device
    tensors: one all_gather of the score levels per rank.  Collective: every rank must call it."""
    import torch

    group = dist.new_group(backend="nccl")
    t = torch.tensor([int(v) for v in local_best], dtype=torch.int64, device=f"cuda:{device_index}")
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    torch.cuda.synchronize(device_index)
    return pick_winner([g.cpu().tolist() for g in gathered])


def max_over_ranks(dist, value: float) -> float:
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float) -> float:
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
