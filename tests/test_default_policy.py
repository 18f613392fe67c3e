"""CPU test: the truth table of compile_default_local_search_components
(crates/solverforge-solver/src/runtime/compiler/default_local_search/policy.rs:21-82) through the C ABI's pure function
sf_default_local_search_components -- no device needed.  The expected column is written out from the reference source, row by row,
not computed by the code under test."""
import itertools

import pytest

from solverforge_amd import Acceptor, Forager, SolverConfig

LA, DLA, SA = Acceptor.LATE_ACCEPTANCE, Acceptor.DIVERSIFIED_LATE_ACCEPTANCE, Acceptor.SIMULATED_ANNEALING
AC, FLSI = Forager.ACCEPTED_COUNT, Forager.FIRST_LAST_STEP_SCORE_IMPROVING


def expected(has_lists, has_groups, has_precedence, has_nearby_scalar, has_conflict_repairs):
    # policy.rs:48-61
    if has_lists:
        acceptor = LA
    elif has_groups:
        acceptor = DLA
    else:
        acceptor = SA
    # policy.rs:63-79
    if has_groups and not has_lists:
        forager = (FLSI, 0)  # accepted_count_limit: None
    elif has_precedence:
        forager = (FLSI, 256)
    elif has_lists or has_nearby_scalar or has_conflict_repairs:
        forager = (AC, 256)
    else:
        forager = (AC, 1)
    return acceptor, forager


def test_truth_table_of_the_default_components():
    rows = 0
    for flags in itertools.product((False, True), repeat=5):
        cfg = SolverConfig.default_components(*flags, random_seed=11)
        acc, (forager, limit) = expected(*flags)
        assert (cfg.acceptor, cfg.forager, cfg.accepted_count_limit) == (acc, forager, limit), flags
        assert cfg.late_acceptance_size == 400 and cfg.random_ties and cfg.selection_order == 3 and cfg.random_seed == 11
        rows += 1
    assert rows == 32


@pytest.mark.parametrize("flags,want", [
    # the rows the reference's own models hit (policy.rs:48-79), spelled out
    ((True, False, False, False, False), (LA, AC, 256)),    # CVRP: lists -> LateAcceptance(400) + AcceptedCount(256)
    ((True, False, True, False, False), (LA, FLSI, 256)),   # job shop with precedence hooks -> FirstLastStepScoreImproving(256)
    ((True, True, True, False, False), (LA, FLSI, 256)),    # lists win over groups for the acceptor; precedence picks the forager
    ((False, True, False, False, False), (DLA, FLSI, 0)),   # grouped scalar-only -> DiversifiedLateAcceptance + no limit
    ((False, True, True, True, True), (DLA, FLSI, 0)),      # ... whatever else is declared
    ((False, False, False, False, False), (SA, AC, 1)),     # plain scalar -> SimulatedAnnealing + AcceptedCount(1)
    ((False, False, False, True, False), (SA, AC, 256)),    # nearby scalar leaves -> 256
    ((False, False, False, False, True), (SA, AC, 256)),    # conflict repairs -> 256
    ((False, False, True, False, False), (SA, FLSI, 256)),  # has_precedence is only ever true with lists; the function is total anyway
])
def test_named_rows(flags, want):
    cfg = SolverConfig.default_components(*flags)
    assert (cfg.acceptor, cfg.forager, cfg.accepted_count_limit) == want


@pytest.mark.gpu
def test_configure_default_derives_the_provider_flags_from_declarations():
    """sf_provider_declare: scalar groups and conflict repairs are host-side providers (planning/scalar/group.rs, planning/conflict_repair.rs:62-84);
    once declared, sf_solver_configure_default reads has_groups / has_conflict_repairs from the context (policy.rs:21-82) instead of taking them
    from the caller.  Scalar-only model: nothing declared -> SimulatedAnnealing + AcceptedCount(1); a conflict repair -> AcceptedCount(256); a group ->
    DiversifiedLateAcceptance(400) + FirstLastStepScoreImproving without a limit."""
    import numpy as np

    import solverforge_amd as sfa

    def model():
        return sfa.build_nqueens((np.arange(12) % 13).astype(np.int64) - 1)

    d = model()
    c = d.configure_default(random_seed=1)
    assert (c.acceptor, c.forager, c.accepted_count_limit) == (sfa.Acceptor.SIMULATED_ANNEALING, sfa.Forager.ACCEPTED_COUNT, 1)
    d.close()
    d = model()
    d.declare_provider(2, "Row conflict")
    c = d.configure_default(random_seed=1)
    assert (c.acceptor, c.forager, c.accepted_count_limit) == (sfa.Acceptor.SIMULATED_ANNEALING, sfa.Forager.ACCEPTED_COUNT, 256)
    assert d.configure_default(random_seed=1, has_conflict_repairs=False).accepted_count_limit == 1  # an explicit flag still wins
    d.close()
    d = model()
    d.declare_provider(1, "rows")
    c = d.configure_default(random_seed=1)
    assert (c.acceptor, c.forager, c.accepted_count_limit) == (sfa.Acceptor.DIVERSIFIED_LATE_ACCEPTANCE, sfa.Forager.FIRST_LAST_STEP_SCORE_IMPROVING, 0)
    with pytest.raises(sfa.SolverForgeError):
        d.declare_provider(7, "x")
    with pytest.raises(sfa.SolverForgeError):
        d.declare_provider(1, "")
    d.close()
