// SimulatedAnnealingAcceptor on the wave replay
// (crates/solverforge-solver/src/phase/localsearch/acceptor/simulated_annealing.rs:11-430).
//
// The reference acceptor is sequential state: a SmallRng draw per worsening candidate that reaches
// the Boltzmann test (:372-374), and -- while auto-calibrating -- one recorded sample per worsening
// candidate (:357-364).  The replay evaluates 64 candidates of the pull order at once, so the
// decision of lane i is computed as "the state after lanes 0..i-1 were all consumed":
//   * sample rank / draw rank of a lane = v_mbcnt over the ballot of the lanes that record / draw,
//   * the k-th draw of the chunk = k+1 steps of xoshiro256++ from the committed state (a uniform
//     loop; every lane keeps the output whose rank is its own),
//   * the lane whose sample completes the calibration (CalibrationState::record returning true,
//     :66-70) and every lane after it use the freshly derived temperatures (:72-87,257-266).
// sa_commit() then advances the committed state by exactly the lanes the forager consumed
// (phase/candidates.rs:66,196,245: the loop stops pulling once the forager quits).
//
// rand 0.10.1 is not in the reference tree (Cargo.lock:314-338): SmallRng = xoshiro256++ seeded by
// splitmix64, f64 sample = (next_u64 >> 11) * 2^-53 are restated from the published algorithm and
// are "parity unpinned" (SURVEY.md §8c); the generator is checked against its reference vector in
// oracle/test_golden.cpp.  exp() is the device libm (<= 1 ulp from glibc's): a decision can differ
// from a CPU run only when a draw lands within one ulp of the acceptance probability.
#pragma once
#include <stdint.h>

#include "sf_common.h"

namespace sf {

// per-replica acceptor state, 32 u64 words (global [R][32]; one LDS copy per resident replica)
enum : int {
    SA_RNG = 0,        // 4: xoshiro256++ state
    SA_TEMP = 4,       // 4: current_temperatures (f64 bits)
    SA_CALIBRATING = 8,
    SA_SEEN = 9,       // CalibrationState::samples_seen
    SA_CNT = 10,       // 4: samples per level
    SA_SUM_LO = 14,    // 4: sum of |delta| per level, low / high 64 bits (i128 total, :80)
    SA_SUM_HI = 18,
    SA_TNEW = 22,      // 4: temperatures a calibration completing inside the current chunk installs
    SA_WORDS = 32
};

struct SaParams {
    double decay_rate;                 // DEFAULT_DECAY_RATE 0.999985 (:11)
    double hill_climbing_temperature;  // 1e-9 (:12)
    double denominator;                // -ln(target_acceptance_probability), computed on the host (:73)
    double fallback_temperature;       // 1.0 (:15)
    int32_t sample_size;               // 128 (:13)
    int32_t never_accept_hard;         // HardRegressionPolicy::NeverAcceptHardRegression (:18-21)
    int32_t hard_levels;               // levels labelled ScoreLevel::Hard
    int32_t levels;                    // Score::levels_count()
    uint64_t* state;                   // [R][SA_WORDS]
};

#if defined(__HIPCC__)

__device__ __forceinline__ void sa_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t sa_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint64_t sa_shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t sa_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
struct SaRng {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ uint64_t next() {
        const uint64_t result = sa_rotl(s0 + s3, 23) + s0;
        const uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = sa_rotl(s3, 45);
        return result;
    }
};

// (hi:lo) as an unsigned 128-bit integer -> f64, round to nearest even (Rust `i128 as f64`, :81)
__device__ __forceinline__ double sa_u128_to_f64(uint64_t hi, uint64_t lo) {
    if (hi == 0) return (double)lo;
    const int sh = 64 - __clzll((unsigned long long)hi);  // 1..64 bits above the low word
    uint64_t top = sh == 64 ? hi : ((hi << (64 - sh)) | (lo >> sh));
    const uint64_t lost = sh == 64 ? lo : (lo & ((1ULL << sh) - 1ULL));
    if (lost) top |= 1ULL;  // sticky bit: `top` keeps 64 significant bits, 11 more than the mantissa
    return ldexp((double)top, sh);
}

// CalibrationState::temperatures for one level (:72-87)
__device__ __forceinline__ double sa_level_temperature(const SaParams& sp, uint64_t cnt, uint64_t hi, uint64_t lo) {
    if (cnt == 0) return sp.fallback_temperature;
    const double mean = sa_u128_to_f64(hi, lo) / (double)cnt;
    const double t = mean / sp.denominator;
    return t > sp.fallback_temperature ? t : sp.fallback_temperature;  // f64::max (no NaN here)
}

// what sa_decide leaves for sa_commit (per lane + wave-uniform)
struct SaChunk {
    bool eff;          // worsening candidate that reaches the calibration / Boltzmann stage
    bool draw;         // consumes one rng draw
    uint32_t level;    // first differing level
    uint32_t rank;     // eff lanes before this one
    uint64_t absd;     // delta.saturating_abs()
    uint64_t drawmask;
    uint32_t need;     // samples still missing at chunk start (calibrating only)
    int fin;           // the calibration completes at lane `flane` of this chunk
    uint32_t flane;
    int calibrating;   // state at chunk start
};

__device__ __forceinline__ void sa_load(uint64_t* w, const SaParams& sp, int r, uint32_t lane) {
    if (lane < SA_WORDS) w[lane] = sp.state[(size_t)r * SA_WORDS + lane];
    sa_fence();
}
__device__ __forceinline__ void sa_store(const uint64_t* w, const SaParams& sp, int r, uint32_t lane) {
    sa_fence();
    if (lane < SA_WORDS) sp.state[(size_t)r * SA_WORDS + lane] = w[lane];
}

// add the samples of the lanes in `mask` (in lane order) to per-level (cnt, hi:lo) accumulators
template <int L>
__device__ __forceinline__ void sa_accumulate(uint64_t mask, const SaChunk& o, uint64_t* cnt, uint64_t* hi, uint64_t* lo) {
    while (mask) {  // wave-uniform loop: <= sample_size iterations per phase in total
        const int l = __ffsll((unsigned long long)mask) - 1;
        mask &= mask - 1;
        const uint64_t v = sa_shfl64(o.absd, l);
        const uint32_t k = (uint32_t)__shfl((int)o.level, l);
#pragma unroll
        for (int q = 0; q < L; ++q)
            if ((uint32_t)q == k) {
                const uint64_t nl = lo[q] + v;
                hi[q] += nl < lo[q] ? 1u : 0u;
                lo[q] = nl;
                cnt[q] += 1;
            }
    }
}

// is_accepted for the 64 candidates of one replay chunk (:338-375).  `doable` lanes carry the trial
// score `sc`; `cur` is last_step_score.  All 64 lanes must call.
template <int L>
__device__ __forceinline__ bool sa_decide(uint64_t* w, const SaParams& sp, bool doable, const ScoreV<L>& sc,
                                          const ScoreV<L>& cur, uint32_t lane, SaChunk& o) {
    const int c = doable ? score_cmp<L>(sc, cur) : 0;
    const bool ge = doable && c >= 0;  // improving or equal: accepted unconditionally (:344-346)
    uint32_t level = 0;
    int64_t delta = 0;
    bool found = false;
#pragma unroll
    for (int k = L - 1; k >= 0; --k)  // first differing level (:283-296)
        if (sc.v[k] != cur.v[k]) {
            level = (uint32_t)k;
            delta = (int64_t)((uint64_t)sc.v[k] - (uint64_t)cur.v[k]);
            found = true;
        }
    o.eff = doable && c < 0 && found && delta < 0 && !(sp.never_accept_hard && (int)level < sp.hard_levels);
    o.level = level;
    o.absd = delta == INT64_MIN ? (uint64_t)INT64_MAX : (uint64_t)(-delta);
    const uint64_t effmask = __ballot(o.eff);
    o.rank = sa_mbcnt(effmask);
    o.calibrating = (int)w[SA_CALIBRATING];
    o.fin = 0;
    o.flane = 0;
    o.need = 0;
    bool active = o.eff;
    if (o.calibrating) {
        o.need = (uint32_t)sp.sample_size - (uint32_t)w[SA_SEEN];
        o.fin = (uint32_t)__popcll(effmask) >= o.need;
        if (o.fin) {
            const uint64_t fm = __ballot(o.eff && o.rank == o.need - 1);
            o.flane = (uint32_t)(__ffsll((unsigned long long)fm) - 1);
            uint64_t cnt[L], hi[L], lo[L];
#pragma unroll
            for (int k = 0; k < L; ++k) {
                cnt[k] = w[SA_CNT + k];
                hi[k] = w[SA_SUM_HI + k];
                lo[k] = w[SA_SUM_LO + k];
            }
            const uint64_t upto = o.flane == 63 ? ~0ULL : ((1ULL << (o.flane + 1)) - 1ULL);
            sa_accumulate<L>(effmask & upto, o, cnt, hi, lo);
            sa_fence();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < L; ++k)
                    w[SA_TNEW + k] = k < sp.levels ? (uint64_t)__double_as_longlong(sa_level_temperature(sp, cnt[k], hi[k], lo[k])) : 0ULL;
            }
            sa_fence();
        }
        active = o.eff && o.fin && o.rank >= o.need - 1;  // earlier samples return false (:359-363)
    }
    const double T = __longlong_as_double((long long)w[(o.calibrating ? SA_TNEW : SA_TEMP) + (level < (uint32_t)L ? level : 0u)]);
    o.draw = active && T > sp.hill_climbing_temperature;  // temperature <= hill_climbing -> false (:367-370)
    o.drawmask = __ballot(o.draw);
    const uint32_t drank = sa_mbcnt(o.drawmask);
    const uint32_t nd = (uint32_t)__popcll(o.drawmask);
    SaRng g{w[SA_RNG], w[SA_RNG + 1], w[SA_RNG + 2], w[SA_RNG + 3]};
    uint64_t mine = 0;
    for (uint32_t i = 0; i < nd; ++i) {
        const uint64_t x = g.next();
        if (o.draw && drank == i) mine = x;
    }
    bool acc = ge;
    if (o.draw) {
        const double probability = exp((double)delta / T);
        const double u = (double)(mine >> 11) * (1.0 / 9007199254740992.0);
        acc = u < probability;
    }
    return acc;
}

// advance the committed acceptor state by the `nconsumed` leading lanes of the chunk
template <int L>
__device__ __forceinline__ void sa_commit(uint64_t* w, const SaParams& sp, const SaChunk& o, uint32_t nconsumed, uint32_t lane) {
    (void)sp;
    const uint64_t cons = nconsumed >= 64 ? ~0ULL : ((1ULL << nconsumed) - 1ULL);
    sa_fence();
    if (o.calibrating) {
        if (o.fin && o.flane < nconsumed) {
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < L; ++k) w[SA_TEMP + k] = w[SA_TNEW + k];  // install_temperatures (:251-255)
                w[SA_CALIBRATING] = 0;
            }
        } else {
            const uint64_t rec = __ballot(o.eff) & cons;
            if (rec) {
                uint64_t cnt[L], hi[L], lo[L];
#pragma unroll
                for (int k = 0; k < L; ++k) {
                    cnt[k] = w[SA_CNT + k];
                    hi[k] = w[SA_SUM_HI + k];
                    lo[k] = w[SA_SUM_LO + k];
                }
                const uint64_t seen = w[SA_SEEN];
                sa_accumulate<L>(rec, o, cnt, hi, lo);
                sa_fence();
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < L; ++k) {
                        w[SA_CNT + k] = cnt[k];
                        w[SA_SUM_HI + k] = hi[k];
                        w[SA_SUM_LO + k] = lo[k];
                    }
                    w[SA_SEEN] = seen + (uint64_t)__popcll(rec);
                }
            }
        }
    }
    const uint32_t nd = (uint32_t)__popcll(o.drawmask & cons);
    if (nd) {
        SaRng g{w[SA_RNG], w[SA_RNG + 1], w[SA_RNG + 2], w[SA_RNG + 3]};
        for (uint32_t i = 0; i < nd; ++i) (void)g.next();
        sa_fence();
        if (lane == 0) {
            w[SA_RNG] = g.s0;
            w[SA_RNG + 1] = g.s1;
            w[SA_RNG + 2] = g.s2;
            w[SA_RNG + 3] = g.s3;
        }
    }
    sa_fence();
}

// step_ended (:417-430): geometric cooling, floored at the hill-climbing temperature; frozen
// while the calibration is still sampling
__device__ __forceinline__ void sa_step_ended(uint64_t* w, const SaParams& sp, uint32_t lane) {
    sa_fence();
    if (lane < (uint32_t)sp.levels && !w[SA_CALIBRATING]) {
        double t = __longlong_as_double((long long)w[SA_TEMP + lane]) * sp.decay_rate;
        if (t < sp.hill_climbing_temperature) t = sp.hill_climbing_temperature;
        w[SA_TEMP + lane] = (uint64_t)__double_as_longlong(t);
    }
    sa_fence();
}

#endif  // __HIPCC__

}  // namespace sf
