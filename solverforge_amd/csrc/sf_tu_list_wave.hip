// One translation unit of the wave engine: k_list_search_wave<SF_TU_L, *, MODE> (general traced / untraced, FAST, FAST + SMALL).
#define SF_TU_ENGINES 3
#include "sf_launch.h"

namespace sf {

template <>
hipError_t launch_tu_list_wave<SF_TU_L>(bool trace, int mode, const SearchLaunch& a) {
    if (mode == 6) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true, 4, true>, a, *a.lm, *a.p, a.nb);  // COMPACT slice, node -> slot table in HBM
    if (mode == 5) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true, 6>, a, *a.lm, *a.p, a.nb);  // ... compiled for 6 waves per SIMD
    if (mode == 4) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true, 5>, a, *a.lm, *a.p, a.nb);  // ... compiled for 5 waves per SIMD
    if (mode == 3) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2, true>, a, *a.lm, *a.p, a.nb);  // FAST + SMALL, COMPACT slice
    if (mode == 2) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 2>, a, *a.lm, *a.p, a.nb);
    if (mode == 1) return launch_with_lds(k_list_search_wave<SF_TU_L, false, 1>, a, *a.lm, *a.p, a.nb);
    if (trace) return launch_with_lds(k_list_search_wave<SF_TU_L, true, 0>, a, *a.lm, *a.p, a.nb);
    return launch_with_lds(k_list_search_wave<SF_TU_L, false, 0>, a, *a.lm, *a.p, a.nb);
}

}  // namespace sf

#if defined(SF_PHASE_PROFILE) || defined(SF_GEN_COUNT)  // diagnostic builds: this unit's copy of the phase counters (a __device__ variable is per translation unit)
#define SF_PH_NAME2(l) sf_debug_phases_wave_##l
#define SF_PH_NAME(l) SF_PH_NAME2(l)
extern "C" int32_t SF_PH_NAME(SF_TU_L)(uint64_t* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_phase), 64) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_phase), z, 64);
    return 0;
}
#endif
