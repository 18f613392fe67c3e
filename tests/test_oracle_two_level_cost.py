"""CPU test of the oracle's assignment model with a SECOND keyed cross-join on another score level (oracle/sfo_models.hpp: make_assignment, cost2 /
cost2_level -- the checker of uni programs compiled onto two levels, round 6): the same CrossBiConstraint shape the reference pins in
constraint/tests/cross_bi_incr.rs:60-83,205-381, once per level.  Scores against a direct numpy evaluation; incremental == fresh along committed moves; every trial
score of the change / swap streams against numpy; the per-constraint rows."""
import numpy as np


def _numpy_score(values, cost, cost2):
    on = values >= 0
    idx = np.flatnonzero(on)
    return np.array([-(~on).sum() - cost2[idx, values[on]].sum(), -cost[idx, values[on]].sum()])


def test_two_level_value_costs_against_numpy(oracle):
    rng = np.random.default_rng(17)
    n, k = 40, 7
    values = rng.integers(-1, k, n)
    cost = rng.integers(0, 6, (n, k))
    cost[cost < 3] = 0
    cost2 = rng.integers(0, 4, (n, k))
    cost2[cost2 < 2] = 0
    o = oracle.Model.assignment(values, cost, k, cost_weight=1, ex_level=-1, cost2=cost2, cost2_level=0)
    bits = oracle.LEAF_SCALAR_CHANGE | oracle.LEAF_SCALAR_SWAP
    o.configure(leaves=bits, random_seed=3, la_size=5, limit=30)
    assert (o.score()[:2] == _numpy_score(values, cost, cost2)).all()
    assert (o.fresh_score()[:2] == o.score()[:2]).all()
    sc, cnt = o.evaluate_each()
    on = values >= 0
    assert sc[1, 1] == -cost[np.flatnonzero(on), values[on]].sum() and sc[2, 0] == -cost2[np.flatnonzero(on), values[on]].sum()
    assert cnt[1] == (cost[np.flatnonzero(on), values[on]] != 0).sum() and cnt[2] == (cost2[np.flatnonzero(on), values[on]] != 0).sum()
    cur = values.copy()
    for it in range(12):
        moves = o.enumerate(0, it, 50 + it, 3)
        scores, doable = o.evaluate_moves(moves)
        for mv, s, dbl in zip(moves, scores, doable):  # every trial against numpy
            if not dbl:
                continue
            trial = cur.copy()
            if mv["kind"] == 0:
                trial[mv["a"]] = mv["value"]
            else:
                trial[mv["a"]], trial[mv["b"]] = trial[mv["b"]], trial[mv["a"]]
            assert (s[:2] == _numpy_score(trial, cost, cost2)).all(), (it, mv)
        mv = moves[np.flatnonzero(doable)[rng.integers(int(doable.sum()))]]
        o.apply_move(mv)
        if mv["kind"] == 0:
            cur[mv["a"]] = mv["value"]
        else:
            cur[mv["a"]], cur[mv["b"]] = cur[mv["b"]], cur[mv["a"]]
        assert (o.get_vars(0, 0) == cur).all()
        assert (o.score()[:2] == _numpy_score(cur, cost, cost2)).all() and (o.fresh_score()[:2] == o.score()[:2]).all()
    o.phase_start()
    o.steps(30)
    assert (o.score()[:2] == _numpy_score(np.asarray(o.get_vars(0, 0)), cost, cost2)).all()
