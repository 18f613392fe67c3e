// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17) of SolverForge's score types and seeded stream
// context.  Nothing under oracle/ is part of the shipped product path: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Parity status: the reference is Rust and cannot be built in this image (no
// cargo/rustc, crates not vendored), so this restatement is pinned against the
// reference's own golden vectors / known-answer counts (see oracle/test_golden.cpp
// and tests/golden/*.json).  Step-seed provenance (rand 0.10.1 StdRng = ChaCha12,
// source not in the tree) is "parity unpinned": every parity test is defined
// given an explicit step_seed sequence.
//
// Follows:
//   crates/solverforge-core/src/score/hard_soft.rs:35-153   (HardSoftScore)
//   crates/solverforge-core/src/score/bendable.rs:35-308    (BendableScore<H,S>)
//   crates/solverforge-core/src/score/macros.rs:16-48       (+,-,neg per level)
//   crates/solverforge-solver/src/heuristic/selector/move_selector/iter.rs:14-207
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>

namespace sfo {

constexpr int MAX_LEVELS = 4;

// One score = up to MAX_LEVELS i64 levels, most significant first.  HardSoft =
// {hard, soft}; Bendable<H,S> = hard[0..H) then soft[0..S).  Unused levels stay
// zero, so comparing all MAX_LEVELS is the lexicographic Ord of the reference
// (hard_soft.rs:130-137, bendable.rs:210-230).  Arithmetic wraps like a Rust
// release build (macros.rs:16-48).
struct Score {
    int64_t v[MAX_LEVELS] = {0, 0, 0, 0};

    static Score zero() { return Score{}; }
    static Score of(int64_t hard, int64_t soft) {
        Score s;
        s.v[0] = hard;
        s.v[1] = soft;
        return s;
    }
    static Score level(int idx, int64_t value) {
        Score s;
        s.v[idx] = value;
        return s;
    }
    int64_t hard() const { return v[0]; }
    int64_t soft() const { return v[1]; }
};

inline int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
inline int64_t wrap_add(int64_t a, int64_t b) {
    return (int64_t)((uint64_t)a + (uint64_t)b);
}
inline int64_t wrap_sub(int64_t a, int64_t b) {
    return (int64_t)((uint64_t)a - (uint64_t)b);
}
inline int64_t wrap_neg(int64_t a) { return (int64_t)(0 - (uint64_t)a); }

inline Score operator+(const Score& a, const Score& b) {
    Score r;
    for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = wrap_add(a.v[i], b.v[i]);
    return r;
}
inline Score operator-(const Score& a, const Score& b) {
    Score r;
    for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = wrap_sub(a.v[i], b.v[i]);
    return r;
}
inline Score operator-(const Score& a) {
    Score r;
    for (int i = 0; i < MAX_LEVELS; ++i) r.v[i] = wrap_neg(a.v[i]);
    return r;
}
inline int cmp(const Score& a, const Score& b) {
    for (int i = 0; i < MAX_LEVELS; ++i) {
        if (a.v[i] < b.v[i]) return -1;
        if (a.v[i] > b.v[i]) return 1;
    }
    return 0;
}
inline bool operator==(const Score& a, const Score& b) { return cmp(a, b) == 0; }
inline bool operator!=(const Score& a, const Score& b) { return cmp(a, b) != 0; }
inline bool operator<(const Score& a, const Score& b) { return cmp(a, b) < 0; }
inline bool operator>(const Score& a, const Score& b) { return cmp(a, b) > 0; }
inline bool operator<=(const Score& a, const Score& b) { return cmp(a, b) <= 0; }
inline bool operator>=(const Score& a, const Score& b) { return cmp(a, b) >= 0; }

// is_feasible: every hard level >= 0 (hard_soft.rs:88-90, bendable.rs:105-107).
inline bool is_feasible(const Score& s, int hard_levels) {
    for (int i = 0; i < hard_levels; ++i)
        if (s.v[i] < 0) return false;
    return true;
}

// ---------------------------------------------------------------------------
// Seeded stream context (iter.rs:14-207).
// ---------------------------------------------------------------------------

inline uint64_t splitmix64(uint64_t value) {  // iter.rs:193-198
    value += 0x9E3779B97F4A7C15ULL;
    value = (value ^ (value >> 30)) * 0xBF58476D1CE4E5B9ULL;
    value = (value ^ (value >> 27)) * 0x94D049BB133111EBULL;
    return value ^ (value >> 31);
}

inline size_t gcd_usize(size_t left, size_t right) {  // iter.rs:200-207
    while (right != 0) {
        size_t rem = left % right;
        left = right;
        right = rem;
    }
    return left;
}

enum class SelectionOrder { Original, Sorted, Probabilistic, Random, Shuffled };

struct MoveStreamContext {
    uint64_t step_index = 0;
    uint64_t step_seed = 0;
    // accepted_count_limit: carried but not read by any in-scope cursor.
    int64_t accepted_count_limit = -1;
    SelectionOrder selection_order = SelectionOrder::Original;

    MoveStreamContext() = default;
    MoveStreamContext(uint64_t idx, uint64_t seed, int64_t limit = -1)
        : step_index(idx), step_seed(seed), accepted_count_limit(limit) {}
    MoveStreamContext with_selection_order(SelectionOrder order) const {
        MoveStreamContext c = *this;
        c.selection_order = order;
        return c;
    }

    bool is_canonical() const {  // iter.rs:175-180
        return selection_order == SelectionOrder::Original ||
               selection_order == SelectionOrder::Sorted ||
               selection_order == SelectionOrder::Probabilistic;
    }
    uint64_t mixed_seed(uint64_t salt) const {  // iter.rs:182-184
        return splitmix64(step_seed ^ (step_index * 0x9E3779B97F4A7C15ULL) ^ salt);
    }
    size_t start_offset(size_t len, uint64_t salt) const {  // iter.rs:59-67
        if (len <= 1) return 0;
        if (is_canonical()) return 0;
        return (size_t)(mixed_seed(salt) % len);
    }
    size_t stride(size_t len, uint64_t salt) const {  // iter.rs:69-81
        if (len <= 1) return 1;
        if (is_canonical()) return 1;
        return random_stride(len, salt);
    }
    size_t random_index(size_t len, uint64_t salt) const {  // iter.rs:90-95
        if (len <= 1) return 0;
        return (size_t)(mixed_seed(salt) % len);
    }
    size_t random_stride(size_t len, uint64_t salt) const {  // iter.rs:97-106
        if (len <= 1) return 1;
        size_t s = (size_t)(mixed_seed(salt) % (len - 1)) + 1;
        while (gcd_usize(s, len) != 1) s = (s == len - 1) ? 1 : s + 1;
        return s;
    }
    uint64_t random_seed(uint64_t salt) const { return mixed_seed(salt); }
    uint64_t offset_seed(uint64_t salt) const { return is_canonical() ? 0 : mixed_seed(salt); }  // iter.rs:80-85

    size_t selection_index(size_t offset, size_t len, uint64_t salt) const {  // iter.rs:112-128
        switch (selection_order) {
            case SelectionOrder::Original:
            case SelectionOrder::Sorted:
            case SelectionOrder::Probabilistic:
                return offset;
            case SelectionOrder::Random:
                return random_index(len, salt ^ ((uint64_t)offset * 0xD1B54A32D192ED03ULL));
            case SelectionOrder::Shuffled: {
                size_t start = random_index(len, salt);
                size_t st = random_stride(len, salt ^ 0xA24BAED4963EE407ULL);
                return (start + offset * st) % len;
            }
        }
        return offset;
    }
    size_t selection_index_without_replacement(size_t offset, size_t len,
                                               uint64_t salt) const {  // iter.rs:133-150
        if (is_canonical()) return offset;
        size_t start = random_index(len, salt);
        size_t st = random_stride(len, salt ^ 0xA24BAED4963EE407ULL);
        return (start + offset * st) % len;
    }
};

// rand 0.10.1 SmallRng on 64-bit targets = xoshiro256++ (Blackman/Vigna, public domain
// algorithm); `seed_from_u64` expands the seed with splitmix64 and `random::<f64>()` is the
// 53-bit multiply sample `(next_u64 >> 11) * 2^-53`.  The crate source is NOT under
// /root/reference (Cargo.lock:314-338) => the draw stream is "parity unpinned" (SURVEY §8c);
// the generator itself is checked against the published xoshiro256++ test vector.
struct SmallRng {
    uint64_t s[4] = {0, 0, 0, 0};
    static SmallRng seed_from_u64(uint64_t state) {
        SmallRng r;
        for (int i = 0; i < 4; ++i) {
            state += 0x9E3779B97F4A7C15ULL;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            r.s[i] = z ^ (z >> 31);
        }
        return r;
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next_u64() {
        const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    // rand `next_u32` of xoshiro256++: the upper half of next_u64 ("the lowest bits have some linear dependencies")
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    // `Rng::random_range` on usize (rand 0.9 / 0.10 `UniformUsize`: sampled as u32 whenever the range fits 32 bits, for
    // portability; Canon's method with one extra step, `UniformInt::sample_single_inclusive`).  The crate source is not in
    // the reference tree: PARITY UNPINNED against the reference, pinned between this oracle and the HIP path.
    uint64_t random_range_inclusive(uint64_t low, uint64_t high) {
        const uint64_t span = high - low;  // high >= low
        if (high <= 0xFFFFFFFFull) {  // UniformUsize: 32-bit sampling unless `high` itself needs 64 bits
            const uint32_t range = (uint32_t)span + 1u;  // 0 = the full 32-bit range
            if (range == 0) return low + next_u32();
            const uint64_t m = (uint64_t)next_u32() * range;
            uint32_t result = (uint32_t)(m >> 32);
            const uint32_t lo_order = (uint32_t)m;
            if (lo_order > (uint32_t)(0u - range)) {
                const uint32_t new_hi = (uint32_t)(((uint64_t)next_u32() * range) >> 32);
                if ((uint64_t)lo_order + new_hi > 0xFFFFFFFFull) result += 1;
            }
            return low + result;
        }
        const uint64_t range = span + 1;
        if (range == 0) return low + next_u64();
        const unsigned __int128 m = (unsigned __int128)next_u64() * range;
        uint64_t result = (uint64_t)(m >> 64);
        const uint64_t lo_order = (uint64_t)m;
        if (lo_order > (uint64_t)(0 - range)) {
            const uint64_t new_hi = (uint64_t)(((unsigned __int128)next_u64() * range) >> 64);
            if (lo_order + new_hi < lo_order) result += 1;
        }
        return low + result;
    }
    uint64_t random_range(uint64_t low, uint64_t high_exclusive) { return random_range_inclusive(low, high_exclusive - 1); }
};

// std::hash::DefaultHasher = SipHash-1-3 with a zero key (Rust std; published algorithm, Aumasson & Bernstein).
// `c` compression rounds per 8-byte block, `d` finalisation rounds.  siphash(2, 4, ...) reproduces the reference vectors
// of the SipHash paper (checked in test_golden.cpp), siphash(1, 3, ...) is what Rust's DefaultHasher computes.
inline uint64_t siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t* data, size_t len) {
    uint64_t v0 = 0x736f6d6570736575ULL ^ k0, v1 = 0x646f72616e646f6dULL ^ k1;
    uint64_t v2 = 0x6c7967656e657261ULL ^ k0, v3 = 0x7465646279746573ULL ^ k1;
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t m = 0;
        for (int b = 0; b < 8; ++b) m |= (uint64_t)data[i + b] << (8 * b);
        v3 ^= m;
        for (int r = 0; r < c_rounds; ++r) round();
        v0 ^= m;
    }
    uint64_t last = (uint64_t)(len & 0xFF) << 56;
    for (size_t b = 0; i + b < len; ++b) last |= (uint64_t)data[i + b] << (8 * b);
    v3 ^= last;
    for (int r = 0; r < c_rounds; ++r) round();
    v0 ^= last;
    v2 ^= 0xFF;
    for (int r = 0; r < d_rounds; ++r) round();
    return v0 ^ v1 ^ v2 ^ v3;
}
// hash_str (heuristic/move/metadata.rs:125-129): `str::hash` feeds the bytes and a 0xFF terminator to DefaultHasher
inline uint64_t hash_str(const char* sz) {
    std::vector<uint8_t> buf;
    for (const char* q = sz; *q; ++q) buf.push_back((uint8_t)*q);
    buf.push_back(0xFF);
    return siphash(1, 3, 0, 0, buf.data(), buf.size());
}
// scoped_seed (heuristic/selector/seed.rs:3-17)
inline uint64_t scoped_seed(uint64_t base_seed, size_t descriptor_index, const char* variable_name, const char* selector_kind) {
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    const uint64_t mixed = base_seed ^ ((uint64_t)descriptor_index * 0x9E3779B97F4A7C15ULL) ^ rotl(hash_str(variable_name), 17) ^
                           rotl(hash_str(selector_kind), 41);
    return splitmix64(mixed);
}

}  // namespace sfo
