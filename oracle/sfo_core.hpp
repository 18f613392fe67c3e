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

}  // namespace sfo
