// Translation unit of the block engine: k_list_search<L, TRACE> for both level counts.
#define SF_TU_ENGINES 1
#include "sf_launch.h"

namespace sf {

template <>
hipError_t launch_tu_list_block<2>(bool trace, const SearchLaunch& a) {
    if (trace) return launch_with_lds(k_list_search<2, true>, a, *a.lm, *a.p);
    return launch_with_lds(k_list_search<2, false>, a, *a.lm, *a.p);
}
template <>
hipError_t launch_tu_list_block<4>(bool trace, const SearchLaunch& a) {
    if (trace) return launch_with_lds(k_list_search<4, true>, a, *a.lm, *a.p);
    return launch_with_lds(k_list_search<4, false>, a, *a.lm, *a.p);
}

}  // namespace sf
