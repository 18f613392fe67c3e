"""solverforge_amd — MI355X-native SolverForge hot path (incremental scoring + neighbourhood sweep).

Host-side mirror of the reference's Director / MoveSelector / local-search surface on top of the
C ABI in include/solverforge_amd.h.  PyTorch is plumbing only (multi-GPU rendezvous); compute is
hand-written HIP for gfx950.
"""
from .director import (
    PairOp,  # noqa: F401
    Acceptor,
    AnnealingMode,
    Engine,
    Forager,
    GpuScoreDirector,
    MoveKind,
    SelectionOrder,
    SolverConfig,
    UniCmp,
    UniLhs,
)
from .models import build_assignment, build_balance, build_cvrp, build_graph_coloring, build_jobshop, build_nqueens, build_precedence_shop, build_shift_schedule  # noqa: F401
from ._lib import MOVE_DTYPE, SolverForgeError  # noqa: F401
