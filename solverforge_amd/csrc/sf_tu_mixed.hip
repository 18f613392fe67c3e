// One translation unit of the generic N-leaf engine: k_mixed_search_wave<SF_TU_L, *, VT, SF_TU_RUIN, SF_TU_PREC>,
// traced and untraced.  Built once per (L, VT, RUIN, PREC) combination by csrc/Makefile.
#include "sf_launch.h"

namespace sf {

template <int VTB>
struct TuValueType {
    using type = int16_t;
};
template <>
struct TuValueType<1> {
    using type = int8_t;
};

template <>
hipError_t launch_tu_mixed<SF_TU_L, SF_TU_VTB, SF_TU_RUIN != 0, SF_TU_PREC != 0>(bool trace, int mode, const SearchLaunch& a) {
    using VT = TuValueType<SF_TU_VTB>::type;
#if SF_TU_VTB == 2 && SF_TU_PREC == 0
    if (mode == 1 && !trace)
        return launch_with_lds(k_mixed_search_wave<SF_TU_L, false, VT, SF_TU_RUIN != 0, false, 1>, a, *a.lm, *a.sm, *a.gl, *a.p, a.has_list,
                               a.has_scalar, a.nb);
#endif
#if SF_TU_PREC != 0
    if (mode == 2 && !trace)
        return launch_with_lds(k_mixed_search_wave<SF_TU_L, false, VT, SF_TU_RUIN != 0, true, 2>, a, *a.lm, *a.sm, *a.gl, *a.p, a.has_list, a.has_scalar,
                               a.nb);
#endif
    (void)mode;
    if (trace)
        return launch_with_lds(k_mixed_search_wave<SF_TU_L, true, VT, SF_TU_RUIN != 0, SF_TU_PREC != 0>, a, *a.lm, *a.sm, *a.gl, *a.p, a.has_list,
                               a.has_scalar, a.nb);
    return launch_with_lds(k_mixed_search_wave<SF_TU_L, false, VT, SF_TU_RUIN != 0, SF_TU_PREC != 0>, a, *a.lm, *a.sm, *a.gl, *a.p, a.has_list,
                           a.has_scalar, a.nb);
}

}  // namespace sf

#ifdef SF_PHASE_PROFILE  // diagnostic builds: this unit's copies of the phase counters (a __device__ variable is per translation unit)
#define SF_PH_NAME2(p, l, v, r, q) p##_##l##_##v##_##r##_##q
#define SF_PH_NAME(p, l, v, r, q) SF_PH_NAME2(p, l, v, r, q)
extern "C" int32_t SF_PH_NAME(sf_debug_phases_mixed, SF_TU_L, SF_TU_VTB, SF_TU_RUIN, SF_TU_PREC)(uint64_t* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_phase), 64) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_phase), z, 64);
    return 0;
}
extern "C" int32_t SF_PH_NAME(sf_debug_ruin2_phases_mixed, SF_TU_L, SF_TU_VTB, SF_TU_RUIN, SF_TU_PREC)(uint64_t* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_rphase2), 64) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_rphase2), z, 64);
    return 0;
}
extern "C" int32_t SF_PH_NAME(sf_debug_ruin_phases_mixed, SF_TU_L, SF_TU_VTB, SF_TU_RUIN, SF_TU_PREC)(uint64_t* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_rphase), 64) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_rphase), z, 64);
    return 0;
}
#endif

#if defined(SF_RUIN_V2_CHECK) && SF_TU_RUIN != 0
#define SF_RV2_NAME2(l, v, r, q) sf_debug_rv2_check_##l##_##v##_##r##_##q
#define SF_RV2_NAME(l, v, r, q) SF_RV2_NAME2(l, v, r, q)
extern "C" int32_t SF_RV2_NAME(SF_TU_L, SF_TU_VTB, SF_TU_RUIN, SF_TU_PREC)(uint64_t* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_rv2_check), 64) != hipSuccess) return -1;
    return 0;
}
#endif
