// One translation unit of the wave engine: k_list_search_wave<SF_TU_L, *, MODE> (general traced / untraced, FAST, FAST + SMALL).
#define SF_TU_ENGINES 3
#include "sf_launch.h"

namespace sf {

template <>
hipError_t launch_tu_list_wave<SF_TU_L>(bool trace, int mode, const SearchLaunch& a) {
    if (mode == 4) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true, 5>, a, *a.lm, *a.p, a.nb);  // ... compiled for 5 waves per SIMD
    if (mode == 3) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true>, a, *a.lm, *a.p, a.nb);  // FAST + SMALL, COMPACT slice
    if (mode == 2) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2>, a, *a.lm, *a.p, a.nb);
    if (mode == 1) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 1>, a, *a.lm, *a.p, a.nb);
    if (trace) return launch_with_lds(k_list_search_wave<SF_TU_L, true, 0>, a, *a.lm, *a.p, a.nb);
    return launch_with_lds(k_list_search_wave<SF_TU_L, false, 0>, a, *a.lm, *a.p, a.nb);
}

}  // namespace sf
