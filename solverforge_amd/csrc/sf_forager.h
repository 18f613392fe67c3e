// Forager decisions of one 64-candidate replay chunk (lanes = pulls in union cursor order), shared by the four search
// engines.  Restates phase/localsearch/forager.rs:157-420 (AcceptedCount / FirstAccepted / BestScore) and
// forager/improving.rs:17-227 (FirstBestScoreImproving / FirstLastStepScoreImproving) for a whole chunk at once.
#pragma once
#include "sf_common.h"

namespace sf {

// forager kinds (sf_forager_kind)
constexpr int FORAGER_ACCEPTED_COUNT = 0, FORAGER_FIRST_ACCEPTED = 1, FORAGER_BEST_SCORE = 2, FORAGER_FIRST_BEST_IMPROVING = 3,
              FORAGER_FIRST_LAST_STEP_IMPROVING = 4;

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t forager_mbcnt64(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// How many lanes of the chunk the step consumes: the loop `while !forager.is_quit_early()` (candidates.rs:66) leaves
// after the accepted candidate that fills the AcceptedCount / FirstAccepted quota, or after the first accepted candidate
// that beats the improving forager's reference score (`thr` = best score ever seen for FirstBestScoreImproving, the last
// step score for FirstLastStepScoreImproving; limit == 0 = no accepted-count limit).  `improving_pick` is set when the
// step ends on such a candidate: BestCandidate::replace makes it the pick whatever came before (improving.rs:92-95,205-208).
template <int L>
__device__ __forceinline__ uint32_t forager_chunk_cut(int forager, uint32_t limit, uint32_t accepted, bool acc, const ScoreV<L>& sc,
                                                      const ScoreV<L>& thr, uint32_t nvalid, bool& improving_pick) {
    improving_pick = false;
    if (forager == FORAGER_BEST_SCORE) return nvalid;
    const uint64_t accmask = __ballot(acc);
    uint64_t cutmask = 0;
    if (forager <= FORAGER_FIRST_ACCEPTED || (forager == FORAGER_FIRST_LAST_STEP_IMPROVING && limit > 0)) {
        const uint32_t remaining = forager == FORAGER_FIRST_ACCEPTED ? 1u : limit - accepted;
        const uint32_t pre = forager_mbcnt64(accmask) + (acc ? 1u : 0u);
        cutmask = __ballot(acc && pre == remaining);
    }
    if (forager >= FORAGER_FIRST_BEST_IMPROVING) {
        const uint64_t imp = __ballot(acc && score_cmp<L>(sc, thr) > 0);
        if (imp) {
            const uint32_t first_imp = (uint32_t)__ffsll((unsigned long long)imp);
            improving_pick = !cutmask || first_imp <= (uint32_t)__ffsll((unsigned long long)cutmask);
            cutmask |= imp;
        }
    }
    return cutmask ? (uint32_t)__ffsll((unsigned long long)cutmask) : nvalid;
}

// is_quit_early() after a chunk whose accepted lanes have been added
__device__ __forceinline__ bool forager_quits(int forager, uint32_t limit, uint32_t accepted, int has_best, bool improving_pick) {
    if (forager == FORAGER_ACCEPTED_COUNT) return accepted >= limit;
    if (forager == FORAGER_FIRST_ACCEPTED) return has_best != 0;
    if (forager == FORAGER_BEST_SCORE) return false;
    return improving_pick || (forager == FORAGER_FIRST_LAST_STEP_IMPROVING && limit > 0 && accepted >= limit);
}
#endif

}  // namespace sf
