"""Host-side check of the arithmetic the CVRP-5000 wave kernel uses instead of its per-step route table (csrc/sf_list_wave.hip: RouteArith, round 6).
The runtime list leaf orders the list owners by the permutation rank -> (start + rank * stride) mod V with gcd(stride, V) = 1
(runtime/compiler/executor/list_leaf/cursor/slot.rs:468-499, heuristic/selector/move_selector/iter.rs:130-147).  The kernel needs the inverse map
(owner -> rank) per neighbour-row entry and an ordinal that orders the destination slots of equal-distance neighbours like the reference's enumeration
order (owners by rank, positions ascending).  CPU only: the same 32-bit Barrett steps in numpy; the device side is covered by tests/test_gpu_cvrp.py."""
import math

import numpy as np
import pytest


def _barrett_mod(x, V):
    """x mod V the way the kernel does it: one multiply-high by floor(2^32 / V), one conditional subtract (x < 2^21, V <= 1022)."""
    recip = (1 << 32) // V if V > 1 else 0
    x = np.asarray(x, dtype=np.uint64)
    q = (x * np.uint64(recip)) >> np.uint64(32)
    r = x - q * np.uint64(V)
    return np.where(r >= V, r - np.uint64(V), r).astype(np.uint64)


def _stride_inverse(stride, V):
    """lane c tests c, c + 64, ...: the first c < V with stride * c == 1 (mod V)"""
    for base in range(0, V, 64):
        c = np.arange(base, base + 64, dtype=np.uint64)
        hit = (c < V) & (_barrett_mod(np.uint64(stride) * c, V) == 1)
        if hit.any():
            return int(c[np.argmax(hit)])
    return 0  # V == 1


@pytest.mark.parametrize("V", [1, 2, 3, 7, 64, 100, 127, 128, 500, 1021, 1022])
def test_rank_of_route_inverts_the_entity_permutation(V):
    rng = np.random.default_rng(V)
    strides = [s for s in range(1, V) if math.gcd(s, V) == 1] or [1]
    for stride in ([strides[0], strides[-1]] + list(rng.choice(strides, size=min(6, len(strides)), replace=False))):
        stride = int(stride)
        for start in {0, V - 1, int(rng.integers(V))}:
            inv = _stride_inverse(stride, V)
            if V > 1:
                assert (stride * inv) % V == 1
            ranks = np.arange(V, dtype=np.uint64)
            route_at = (start + ranks * stride) % V  # the reference's order: rank -> owner
            x = route_at + np.uint64(V) - np.uint64(start)
            x = np.where(x >= V, x - np.uint64(V), x)
            got = _barrett_mod(x * np.uint64(inv), V)
            assert (got == ranks).all(), (V, stride, start)


def test_rank_position_pairs_order_like_slot_ordinals():
    """first-slot-ordinal(rank) + position (the table the kernel used to rebuild every step) and (rank << 16 | position) sort destination slots the same way."""
    rng = np.random.default_rng(5)
    V = 37
    lens = rng.integers(0, 12, size=V)
    first = np.concatenate([[0], np.cumsum(lens + 1)[:-1]])  # rank k owns lens[k] + 1 slots
    slots = [(k, p) for k in range(V) for p in range(lens[k] + 1)]
    table_key = [first[k] + p for k, p in slots]
    arith_key = [(k << 16) | p for k, p in slots]
    assert np.array_equal(np.argsort(table_key, kind="stable"), np.argsort(arith_key, kind="stable"))
    assert len(set(arith_key)) == len(arith_key) and max(arith_key) < (1 << 26)  # below ORD_ARITH_BASE: intra-list ordinals (positions) stay smaller
