// Shared host/device primitives of the MI355X hot path: seeded stream context, salts,
// score vectors.  Integer code is bit-identical on host and device.
//
// Semantics restated from (paths under crates/solverforge-solver/src/):
//   heuristic/selector/move_selector/iter.rs:14-207   MoveStreamContext
//   phase/localsearch/forager.rs:143-155              reservoir_pick
// and crates/solverforge-core/src/score/{hard_soft,bendable,macros}.rs for the score algebra.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__ __forceinline__
#else
#define SF_HD inline
#endif

// Plain (non-template) kernels are compiled once, in the translation unit of the C ABI (sf_api.hip defines
// SF_TU_MAIN); the per-engine search units (sf_tu_*.hip, built in parallel) see them as uninstantiated templates.
#ifdef SF_TU_MAIN
#define SF_PLAIN_KERNEL
#else
#define SF_PLAIN_KERNEL template <int SF_TU_UNUSED_ = 0>
#endif

namespace sf {

constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ULL;
constexpr uint64_t OFFSET_MIX = 0xD1B54A32D192ED03ULL;     // iter.rs:122
constexpr uint64_t STRIDE_SALT_MIX = 0xA24BAED4963EE407ULL; // iter.rs:126,145

// leaf salts
constexpr uint64_t SALT_SCALAR_CHANGE_VALUE = 0xC4A46E0000000000ULL;   // cursor/change.rs:10
constexpr uint64_t SALT_SCALAR_CHANGE_ENTITY = 0xC4A46E0000000001ULL;  // cursor/change.rs:11
constexpr uint64_t SALT_SCALAR_SWAP_LEFT = 0x5A095CA1AA000001ULL;      // cursor/swap.rs:10
constexpr uint64_t SALT_SCALAR_SWAP_RIGHT = 0x5A095CA1AA000002ULL;     // cursor/swap.rs:11
constexpr uint64_t SALT_NEARBY_CHANGE_ENTITY = 0xA1EA2B17C4A40001ULL;  // nearby_change.rs:19
constexpr uint64_t SALT_NEARBY_CHANGE_SOURCE = 0xA1EA2B17C4A40002ULL;  // nearby_change.rs:20
constexpr uint64_t SALT_NEARBY_SWAP_ENTITY = 0xA1EA25A090000001ULL;    // nearby_swap.rs:19
constexpr uint64_t SALT_NEARBY_SWAP_SOURCE = 0xA1EA25A090000002ULL;    // nearby_swap.rs:20
constexpr uint64_t SALT_UNION_OFFSET = 0xA11CE5E1EC700001ULL;          // vec_union.rs:232
constexpr uint64_t SALT_UNION_STRIDE = 0xA11CE5E1EC700002ULL;          // vec_union.rs:240
constexpr uint64_t SALT_RESERVOIR = 0xF04A63E239B74D11ULL;             // forager.rs:145

constexpr int64_t UNREACHABLE = INT64_MAX;            // crates/solverforge-cvrp/src/problem_data.rs:6
constexpr int64_t MAX_SAFE_LEG_COST = INT64_MAX / 4;  // problem_data.rs:8

SF_HD uint64_t splitmix64(uint64_t v) {  // iter.rs:193-198
    v += GOLDEN;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ULL;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBULL;
    return v ^ (v >> 31);
}

SF_HD uint32_t gcd_u32(uint32_t a, uint32_t b) {  // iter.rs:200-207
    while (b != 0) {
        uint32_t r = a % b;
        a = b;
        b = r;
    }
    return a;
}

// Build-defined step-seed stream (the reference's rand 0.10.1 StdRng is not in the tree:
// parity unpinned).  draw k of replica seed s.
SF_HD uint64_t step_seed(uint64_t random_seed, uint64_t draw) { return splitmix64(random_seed + draw * GOLDEN); }

// x % n.  For n < 2^16 (entity counts, list lengths, leaf counts) three 32-bit remainders (high word, then the two
// 16-bit limbs of the low word) replace the 64-bit software division the GPU would otherwise expand to.
// SF_OUTLINE_MOD: an EXPERIMENT that was not adopted (no unit of csrc/Makefile defines it).  Inlined, the three software remainders are ~100
// instructions at each of ~90 call sites of the generic engine -- 12.7 K of its 65 K instructions -- and one out-of-line copy (a call costs a
// dozen) moved the step time by < 3 %: code size is not that kernel's bound.  The macro stays for diagnostic builds (EXTRA=-DSF_OUTLINE_MOD).
#if defined(SF_OUTLINE_MOD) && defined(__HIPCC__)
#define SF_MOD_ATTR __host__ __device__ inline __attribute__((noinline))
#else
#define SF_MOD_ATTR SF_HD
#endif
SF_MOD_ATTR uint32_t mod_u64(uint64_t x, uint32_t n) {
    if (n >= 65536u) return (uint32_t)(x % n);
    uint32_t r = (uint32_t)(x >> 32) % n;  // the high word fits a 32-bit remainder directly
    r = ((r << 16) | (uint32_t)((x >> 16) & 0xFFFFu)) % n;
    r = ((r << 16) | (uint32_t)(x & 0xFFFFu)) % n;
    return r;
}

// Exact remainder by a divisor that stays fixed for many draws (entity count, value count):
// Barrett reduction with M = floor((2^64 - 1) / n): q = hi64(x * M) is floor(x / n) or up to two
// less, so x - q * n needs at most two conditional subtractions; only its low 32 bits are needed
// (remainder < 3n).  One 64x64 high multiply instead of a software division.
struct FastMod {
    uint64_t M;
    uint32_t n;
};
SF_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
SF_HD FastMod make_fastmod(uint32_t n) {
    FastMod f;
    f.n = n;
    f.M = n ? ~0ULL / n : 0;
    return f;
}
SF_HD uint32_t fastmod_u64(uint64_t x, const FastMod& f) {
    if (f.n >= 0x40000000u) return (uint32_t)(x % f.n);  // keep 3n inside 32 bits
    const uint64_t q = mulhi64(x, f.M);
    uint32_t r = (uint32_t)x - (uint32_t)q * f.n;
    if (r >= f.n) r -= f.n;
    if (r >= f.n) r -= f.n;
    return r;
}

SF_HD int ctz_u64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}
// bit s (1 <= s < len <= 128) set when gcd(s, len) == 1
SF_HD void coprime_mask(uint32_t len, uint64_t& lo, uint64_t& hi) {
    lo = 0;
    hi = 0;
    for (uint32_t s = 1; s < len && s < 128; ++s)
        if (gcd_u32(s, len) == 1) (s < 64 ? lo : hi) |= 1ULL << (s & 63u);
}

struct StreamCtx {
    uint64_t step_index;
    uint64_t step_seed;
    int32_t order;  // sf_selection_order

    SF_HD bool canonical() const { return order <= 2; }
    SF_HD uint64_t mixed_seed(uint64_t salt) const {  // iter.rs:182-184
        return splitmix64(step_seed ^ (step_index * GOLDEN) ^ salt);
    }
    SF_HD uint32_t random_index(uint32_t len, uint64_t salt) const {  // iter.rs:90-95
        if (len <= 1) return 0;
        return mod_u64(mixed_seed(salt), len);
    }
    SF_HD uint32_t random_stride(uint32_t len, uint64_t salt) const {  // iter.rs:97-106
        if (len <= 1) return 1;
        uint32_t s = mod_u64(mixed_seed(salt), len - 1) + 1;
        while (gcd_u32(s, len) != 1) s = (s == len - 1) ? 1 : s + 1;
        return s;
    }
    // selection_index (iter.rs:112-128)
    SF_HD uint32_t selection_index(uint32_t offset, uint32_t len, uint64_t salt) const {
        if (order <= 2) return offset;
        if (order == 3) return random_index(len, salt ^ ((uint64_t)offset * OFFSET_MIX));
        uint32_t start = random_index(len, salt);
        uint32_t st = random_stride(len, salt ^ STRIDE_SALT_MIX);
        return mod_u64((uint64_t)start + (uint64_t)offset * st, len);
    }
    // selection_index for a length whose FastMod the caller keeps (same results, no division)
    SF_HD uint32_t selection_index_fm(uint32_t offset, const FastMod& f, uint64_t salt) const {
        if (order <= 2) return offset;
        if (f.n <= 1) return 0;
        if (order == 3) return fastmod_u64(mixed_seed(salt ^ ((uint64_t)offset * OFFSET_MIX)), f);
        return selection_index(offset, f.n, salt);
    }
    // random_stride with the coprimality of every candidate stride precomputed (CoprimeMask): the reference walks
    // s, s+1, .. (wrapping len-1 -> 1) until gcd(s, len) == 1 (iter.rs:97-106); the same walk is "first set bit at or
    // after s, else bit 1" on the mask.  Same result, no division loop per step.
    SF_HD uint32_t random_stride_cm(uint32_t len, uint64_t salt, uint64_t cm_lo, uint64_t cm_hi) const {
        if (len <= 1) return 1;
        const uint32_t s = mod_u64(mixed_seed(salt), len - 1) + 1;  // 1 ..= len-1 <= 127
        return first_coprime_from(s, cm_lo, cm_hi);
    }
    SF_HD static uint32_t first_coprime_from(uint32_t s, uint64_t cm_lo, uint64_t cm_hi) {
        uint64_t m;
        if (s < 64) {
            m = cm_lo >> s;
            if (m) return s + (uint32_t)ctz_u64(m);
            if (cm_hi) return 64u + (uint32_t)ctz_u64(cm_hi);
        } else {
            m = cm_hi >> (s - 64);
            if (m) return s + (uint32_t)ctz_u64(m);
        }
        return 1;  // wrapped: gcd(1, len) == 1
    }
    // same, with Barrett remainders by len (fm) and len - 1 (fm1) kept by the caller
    SF_HD void perm_params_fm(const FastMod& fm, const FastMod& fm1, uint64_t salt, uint32_t& start, uint32_t& stride, uint64_t cm_lo,
                              uint64_t cm_hi) const {
        if (order <= 2 || fm.n <= 1) {
            start = 0;
            stride = 1;
            return;
        }
        start = fastmod_u64(mixed_seed(salt), fm);
        stride = first_coprime_from(fastmod_u64(mixed_seed(salt ^ STRIDE_SALT_MIX), fm1) + 1, cm_lo, cm_hi);
    }
    SF_HD void perm_params_cm(uint32_t len, uint64_t salt, uint32_t& start, uint32_t& stride, uint64_t cm_lo, uint64_t cm_hi) const {
        if (order <= 2) {
            start = 0;
            stride = 1;
            return;
        }
        start = random_index(len, salt);
        stride = random_stride_cm(len, salt ^ STRIDE_SALT_MIX, cm_lo, cm_hi);
    }
    // permutation parameters of selection_index_without_replacement (iter.rs:133-150)
    SF_HD void perm_params(uint32_t len, uint64_t salt, uint32_t& start, uint32_t& stride) const {
        if (order <= 2) {
            start = 0;
            stride = 1;
            return;
        }
        start = random_index(len, salt);
        stride = random_stride(len, salt ^ STRIDE_SALT_MIX);
    }
};

SF_HD bool reservoir_pick(uint64_t seed, uint64_t equal_count) {  // forager.rs:143-148
    uint64_t mixed = splitmix64(seed ^ (equal_count * GOLDEN) ^ SALT_RESERVOIR);
    return (equal_count < 65536u ? (uint64_t)mod_u64(mixed, (uint32_t)equal_count) : mixed % equal_count) == 0;
}

SF_HD int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
SF_HD int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }

template <int L>
struct ScoreV {
    int64_t v[L];
};
template <int L>
SF_HD int score_cmp(const ScoreV<L>& a, const ScoreV<L>& b) {  // hard_soft.rs:130-137, bendable.rs:210-230
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (a.v[i] < b.v[i]) return -1;
        if (a.v[i] > b.v[i]) return 1;
    }
    return 0;
}

// DiversifiedLateAcceptance threshold (phase/localsearch/acceptor/diversified_late_acceptance.rs:139-146):
// best - |best|.multiply(tolerance), every level `(x as f64 * t).round() as i64` (score/macros.rs:61-71: round half away
// from zero, the cast saturates and maps NaN to 0).
SF_HD int64_t f64_as_i64(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
template <int L>
SF_HD ScoreV<L> dla_threshold(const ScoreV<L>& best, double tolerance) {
    ScoreV<L> t;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int64_t a = best.v[i] < 0 ? wsub(0, best.v[i]) : best.v[i];
        t.v[i] = wsub(best.v[i], f64_as_i64(__builtin_round((double)a * tolerance)));
    }
    return t;
}

}  // namespace sf
