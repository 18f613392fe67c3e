// Seam between the C-ABI translation unit (sf_api.hip) and the search-kernel translation units (sf_tu_*.hip).
// The fused search kernels are large templates; each unit instantiates a few of them, so the library builds
// in parallel (csrc/Makefile) and an edit of one engine recompiles that engine only.  Every launcher sets the
// dynamic-LDS attribute of its instantiation, launches on the caller's stream and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>

// A unit names the engine sources it instantiates from with SF_TU_ENGINES (bit 0 block engine + plain list kernels, 1 wave engine,
// 2 scalar engine, 3 generic engine = all of them) before including this header, so that its object depends on those sources only
// (csrc/Makefile records the real include set per object); the C-ABI unit takes everything.
#ifndef SF_TU_ENGINES
#define SF_TU_ENGINES 15
#endif
#include "sf_list_model.h"
#if SF_TU_ENGINES & 1
#include "sf_list_kernels.hip"
#endif
#if SF_TU_ENGINES & 2
#include "sf_list_wave.hip"
#endif
#if SF_TU_ENGINES & 4
#include "sf_scalar_kernels.hip"
#endif
#if SF_TU_ENGINES & 8
#include "sf_mixed_wave.hip"
#endif

namespace sf {

struct ScalarModel;
struct GLeaves;

struct SearchLaunch {
    int grid, block;
    size_t lds;
    hipStream_t stream;
    const ListModel* lm;
    const ScalarModel* sm;
    const GLeaves* gl;
    const SearchParams* p;
    int has_list, has_scalar;
    NbrIndex nb;
};

// block engine: k_list_search<L, TRACE>
template <int L>
hipError_t launch_tu_list_block(bool trace, const SearchLaunch& a);
// wave engine: k_list_search_wave<L, TRACE, MODE, COMPACT> (mode 1 / 2 = the FAST instantiations, never traced; 3 = MODE 2 in the
// COMPACT LDS layout, 4 = the same compiled for 5 waves per SIMD)
template <int L>
hipError_t launch_tu_list_wave(bool trace, int mode, const SearchLaunch& a);
// scalar engine: k_scalar_search_wave<L, TRACE, VT, IR>, VT = int8_t (VTB 1) or int16_t (VTB 2); IR = the unit built with the interpreted
// pair-predicate joins (1) or without them (0: specialised joins only)
template <int L, int VTB, int IR>
hipError_t launch_tu_scalar(bool trace, const SearchLaunch& a);
// generic N-leaf engine: k_mixed_search_wave<L, TRACE, VT, RUIN, PREC, MODE> (mode 1 = the FAST instantiation of the default
// list policy: VTB 2, no precedence constraint, never traced; mode 2 = the PREC instantiations built for four workgroups per CU)
template <int L, int VTB, bool RUIN, bool PREC>
hipError_t launch_tu_mixed(bool trace, int mode, const SearchLaunch& a);

#define SF_TU_DECL_MIXED(L, VTB, RUIN, PREC) \
    template <>                              \
    hipError_t launch_tu_mixed<L, VTB, RUIN, PREC>(bool trace, int mode, const SearchLaunch& a);
SF_TU_DECL_MIXED(2, 1, false, false)
SF_TU_DECL_MIXED(4, 1, false, false)
SF_TU_DECL_MIXED(2, 2, false, false)
SF_TU_DECL_MIXED(4, 2, false, false)
SF_TU_DECL_MIXED(2, 2, true, false)
SF_TU_DECL_MIXED(4, 2, true, false)
SF_TU_DECL_MIXED(2, 1, false, true)
SF_TU_DECL_MIXED(4, 1, false, true)
SF_TU_DECL_MIXED(2, 2, false, true)
SF_TU_DECL_MIXED(4, 2, false, true)
SF_TU_DECL_MIXED(2, 2, true, true)
SF_TU_DECL_MIXED(4, 2, true, true)
#undef SF_TU_DECL_MIXED
#define SF_TU_DECL_SCALAR(L, VTB)                                                \
    template <>                                                                  \
    hipError_t launch_tu_scalar<L, VTB, 0>(bool trace, const SearchLaunch& a);   \
    template <>                                                                  \
    hipError_t launch_tu_scalar<L, VTB, 1>(bool trace, const SearchLaunch& a);
SF_TU_DECL_SCALAR(2, 1)
SF_TU_DECL_SCALAR(2, 2)
SF_TU_DECL_SCALAR(4, 1)
SF_TU_DECL_SCALAR(4, 2)
#undef SF_TU_DECL_SCALAR
template <>
hipError_t launch_tu_list_wave<2>(bool trace, int mode, const SearchLaunch& a);
template <>
hipError_t launch_tu_list_wave<4>(bool trace, int mode, const SearchLaunch& a);
template <>
hipError_t launch_tu_list_block<2>(bool trace, const SearchLaunch& a);
template <>
hipError_t launch_tu_list_block<4>(bool trace, const SearchLaunch& a);

template <class K, class... Args>
inline hipError_t launch_with_lds(K kern, const SearchLaunch& a, Args... args) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(a.block), a.lds, a.stream, args...);
    return hipGetLastError();
}

}  // namespace sf
