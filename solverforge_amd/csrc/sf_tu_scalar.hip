// One translation unit of the scalar engine: k_scalar_search_wave<SF_TU_L, *, VT>, traced and untraced -- built twice per (L, VT): SF_TU_IR = 0 without
// the interpreted pair-predicate joins (the models whose program matched a specialised loop), 1 with them (four partners side by side; eight spill:
// graph colouring interpreted 5.9 -> 4.8 G moves/s, profiles/r06k_pair_ir_ab.txt).
#ifndef SF_TU_IR
#define SF_TU_IR 1
#endif
#define SF_SCALAR_PAIR_IR SF_TU_IR
#if !SF_TU_IR && !defined(SF_CONFLICT_W)
#define SF_CONFLICT_W 8  // partner ids in flight per pass of the specialised join: 8 / 16 / 24 / 32 -> 10.0 / 9.8 / 9.3 / 9.4 G on graph colouring (profiles/r06k_pair_ir_ab.txt)
#endif
#define SF_TU_ENGINES 7
#include "sf_launch.h"

namespace sf {

template <>
hipError_t launch_tu_scalar<SF_TU_L, SF_TU_VTB, SF_TU_IR>(bool trace, const SearchLaunch& a) {
#if SF_TU_VTB == 1
    using VT = int8_t;
#else
    using VT = int16_t;
#endif
    if (trace) return launch_with_lds(k_scalar_search_wave<SF_TU_L, true, VT>, a, *a.sm, *a.p);
    return launch_with_lds(k_scalar_search_wave<SF_TU_L, false, VT>, a, *a.sm, *a.p);
}

}  // namespace sf
