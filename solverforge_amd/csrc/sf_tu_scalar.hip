// One translation unit of the scalar engine: k_scalar_search_wave<SF_TU_L, *, VT>, traced and untraced.
#define SF_TU_ENGINES 7
#include "sf_launch.h"

namespace sf {

template <>
hipError_t launch_tu_scalar<SF_TU_L, SF_TU_VTB>(bool trace, const SearchLaunch& a) {
#if SF_TU_VTB == 1
    using VT = int8_t;
#else
    using VT = int16_t;
#endif
    if (trace) return launch_with_lds(k_scalar_search_wave<SF_TU_L, true, VT>, a, *a.sm, *a.p);
    return launch_with_lds(k_scalar_search_wave<SF_TU_L, false, VT>, a, *a.sm, *a.p);
}

}  // namespace sf
