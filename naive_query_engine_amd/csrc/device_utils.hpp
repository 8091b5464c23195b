// device_utils.hpp — shared __device__ helpers (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "nqe_internal.hpp"

namespace nqe {

constexpr int WAVE = 64;
constexpr int TILE_ROWS = 4096;  // compaction tile: 64 bitmap words of 64 rows
constexpr int TILE_WORDS = TILE_ROWS / 64;

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & 63; }
__device__ __forceinline__ uint64_t lanemask_lt() {
    // bits below this lane
    return (1ull << lane_id()) - 1ull;
}
__device__ __forceinline__ bool get_bit(const uint8_t *bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }

__device__ __forceinline__ double u2d(uint64_t w) { return __longlong_as_double((long long)w); }
__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }

// `val as f64` for the aggregates (sum.rs:44): Int64/UInt64/Float64 → f64
__device__ __forceinline__ double word_as_f64(uint64_t w, int dtype) {
    if (dtype == NQE_INT64) return double((long long)w);
    if (dtype == NQE_UINT64) return double(w);
    return u2d(w);
}

// The 64-bit software divide (and fmod) are ~200 instructions each; kept out of line so that kernels
// inlining apply_binary stay small (an inlined copy per unrolled row made the first aggregate kernel
// 30k instructions long and instruction-fetch bound).
static __device__ __noinline__ uint64_t divmod_general(int op, int dt, uint64_t a, uint64_t b) {
    if (dt == NQE_UINT64) return op == NQE_OP_DIVIDE ? a / b : a % b;
    if (dt == NQE_FLOAT64) {
        double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
        double z = op == NQE_OP_DIVIDE ? x / y : fmod(x, y);
        return (uint64_t)__double_as_longlong(z);
    }
    long long x = (long long)a, y = (long long)b;
    return (uint64_t)(op == NQE_OP_DIVIDE ? x / y : x % y);
}

// n / d for the host-prepared (magic, more) of a non-power-of-two d
__device__ __forceinline__ uint64_t udiv_magic(uint64_t n, const OpAux &aux) {
    uint64_t q = __umul64hi(aux.magic, n);
    uint64_t t = ((n - q) >> 1) + q;
    return t >> aux.more;
}

// One binary step on raw 64-bit words. `dt` is the OPERAND dtype (result dtype is Boolean for
// compares). Semantics follow arrow-rs 13 (see oracle/nqe_oracle.cpp): wrapping integer
// arithmetic, truncated remainder, IEEE float compares; zero divisor / MIN÷-1 raise device
// flags which the host turns into NQE_ERR_ARROW. Control flow is wave-uniform (op, dt come
// from kernel arguments).
// CHECKED = false: the caller guarantees an integer divisor other than 0 and -1 (literals vetted by the host), so the fault
// tests are compiled out.
template <bool CHECKED = true>
__device__ __forceinline__ uint64_t apply_binary(int op, int dt, uint64_t a, uint64_t b, const OpAux &aux,
                                                 bool valid, int *flags) {
    if (op <= NQE_OP_GT_EQ) {
        bool r;
        if (dt == NQE_INT64) {
            long long x = (long long)a, y = (long long)b;
            r = op == NQE_OP_EQ ? x == y : op == NQE_OP_NOT_EQ ? x != y : op == NQE_OP_LT ? x < y
                : op == NQE_OP_LT_EQ ? x <= y : op == NQE_OP_GT ? x > y : x >= y;
        } else if (dt == NQE_FLOAT64) {
            double x = u2d(a), y = u2d(b);
            r = op == NQE_OP_EQ ? x == y : op == NQE_OP_NOT_EQ ? x != y : op == NQE_OP_LT ? x < y
                : op == NQE_OP_LT_EQ ? x <= y : op == NQE_OP_GT ? x > y : x >= y;
        } else { // UInt64 and Boolean (0/1)
            r = op == NQE_OP_EQ ? a == b : op == NQE_OP_NOT_EQ ? a != b : op == NQE_OP_LT ? a < b
                : op == NQE_OP_LT_EQ ? a <= b : op == NQE_OP_GT ? a > b : a >= b;
        }
        return r ? 1ull : 0ull;
    }
    if (dt == NQE_FLOAT64) {
        double x = u2d(a), y = u2d(b), z;
        switch (op) {
        case NQE_OP_PLUS: z = x + y; break;
        case NQE_OP_MINUS: z = x - y; break;
        case NQE_OP_MULTIPLY: z = x * y; break;
        default:
            if (op == NQE_OP_DIVIDE && aux.more == -2) return d2u(x * u2d(aux.magic)); // literal divisor ±2^k: its exact reciprocal (make_aux)
            if (y == 0.0) {
                if (valid) atomicOr(&flags[NQE_FLAG_DIV_ZERO], 1);
                return 0;
            }
            return divmod_general(op, dt, a, b);
        }
        return d2u(z);
    }
    switch (op) {
    case NQE_OP_PLUS: return a + b;
    case NQE_OP_MINUS: return a - b;
    case NQE_OP_MULTIPLY: return a * b;
    default: break;
    }
    // divide / modulus on integers
    if (CHECKED) {
        if (b == 0) {
            if (valid) atomicOr(&flags[NQE_FLAG_DIV_ZERO], 1);
            return 0;
        }
        if (dt == NQE_INT64 && (long long)a == INT64_MIN && (long long)b == -1) {
            if (valid) atomicOr(&flags[NQE_FLAG_OVERFLOW], 1);
            return 0;
        }
    }
    if (aux.pow2_shift >= 0) {
        // divisor is a literal ±2^k: truncated division/remainder without the 64-bit divide
        if (dt == NQE_UINT64) return op == NQE_OP_DIVIDE ? a >> aux.pow2_shift : a & (aux.abs_lit - 1);
        long long x = (long long)a, y = (long long)b;
        uint64_t ux = x < 0 ? 0ull - a : a;
        if (op == NQE_OP_MODULOS) {
            uint64_t ur = ux & (aux.abs_lit - 1);
            return x < 0 ? 0ull - ur : ur;
        }
        uint64_t uq = ux >> aux.pow2_shift;
        bool neg = (x < 0) != (y < 0);
        return neg ? 0ull - uq : uq;
    }
    if (aux.more >= 0) {
        // literal divisor that is not a power of two: magic multiply instead of the 64-bit divide
        const bool sgn = dt == NQE_INT64;
        const bool xneg = sgn && (long long)a < 0;
        uint64_t ux = xneg ? 0ull - a : a;
        uint64_t uq = udiv_magic(ux, aux);
        if (op == NQE_OP_MODULOS) {
            uint64_t ur = ux - uq * aux.abs_lit;
            return xneg ? 0ull - ur : ur;
        }
        bool neg = sgn && (xneg != ((long long)b < 0));
        return neg ? 0ull - uq : uq;
    }
    return divmod_general(op, dt, a, b);
}

__device__ __forceinline__ OpAux no_aux() {
    OpAux a;
    a.pow2_shift = -1;
    a.more = -1;
    a.abs_lit = 0;
    a.magic = 0;
    return a;
}

// Evaluates a SimpleExpr on a source word that is already in a register.
template <bool CHECKED = true>
__device__ __forceinline__ uint64_t eval_simple(const SimpleExpr &e, uint64_t v, bool valid, int *flags) {
#pragma unroll
    for (int k = 0; k < SIMPLE_MAX_OPS; ++k) {
        if (k < e.nops) {
            uint64_t a = e.lit_left[k] ? e.lit[k] : v;
            uint64_t b = e.lit_left[k] ? v : e.lit[k];
            v = apply_binary<CHECKED>(e.op[k], e.op_dtype[k], a, b, e.lit_left[k] ? no_aux() : e.aux[k], valid, flags);
        }
    }
    return v;
}

// loads element i of a column as a 64-bit word (Boolean → 0/1)
__device__ __forceinline__ uint64_t load_word(const void *values, int dtype, int64_t i) {
    if (dtype == NQE_BOOLEAN) return get_bit(static_cast<const uint8_t *>(values), i) ? 1ull : 0ull;
    return static_cast<const uint64_t *>(values)[i];
}

// splitmix64 (SURVEY §8d synthetic data)
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
// hash for the device hash tables (any mixer works: the reference's XxHash64 values are
// unobservable in results, hash_join.rs:95 re-checks equality)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// broadcast lane `lane` (wave-uniform) of a 64-bit value through the scalar unit (v_readlane_b32),
// instead of an LDS-routed ds_bpermute
__device__ __forceinline__ uint64_t bcast64(uint64_t x, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
    return (uint64_t(hi) << 32) | lo;
}
__device__ __forceinline__ uint32_t bcast32(uint32_t x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, lane); }

// integer `x op lit` as a branch-free range test (host-prepared, see make_fast_pred)
// lo <= (x ^ flip) <= hi (signed) as ONE subtraction and ONE unsigned compare: flip is 0 or the sign bit, and x ^ signbit == x + 2^63,
// so with everything moved to the unsigned image the test is (x - base) < width, base and width wave-uniform (scalar arithmetic,
// hoisted out of the row loops).  width == 0 is the empty range — which the host always writes as lo = hi + 1 (make_fast_pred; any
// other lo > hi would wrap to a non-zero width) — or the full one, whose width 2^64 wraps.  Two vector
// instructions per row instead of four (two 32-bit xors + two 64-bit compares): the streaming kernels are issue-bound.
__device__ __forceinline__ bool range_pass(const FastPred &fp, uint64_t x) {
    const uint64_t f2 = fp.flip ^ 0x8000000000000000ull;                       // 2^63 for signed operands, 0 for unsigned ones
    const uint64_t nbase = f2 - (uint64_t(fp.lo) + 0x8000000000000000ull);     // -(LO - f2)
    const uint64_t width = uint64_t(fp.hi) - uint64_t(fp.lo) + 1ull;
    const bool full = width == 0 && fp.lo <= fp.hi;
    return (((x + nbase) < width) || full) != (fp.negate != 0);
}

// Float64 → order-preserving signed integer (fmask = 0 leaves integers alone).  Deliberately NOT part of range_pass: the
// headline kernel tests its integer key column with range_pass alone, and is VALU-sensitive — this mapping there cost
// 2.4 % as a wave-uniform branch and 9 % branch-free.  Float predicates go through pred_extract (the "other column"
// variants) or the F64 instance of the selection kernel.
__device__ __forceinline__ uint64_t f64_order_map(const FastPred &fp, uint64_t x) { return x ^ (uint64_t(int64_t(x) >> 63) & fp.fmask); }

// the word a FastPred tests for row `row`, from the loaded source word (a word column's element or a bitmap word)
__device__ __forceinline__ uint64_t pred_extract(const FastPred &fp, uint64_t w, int64_t row) {
    return f64_order_map(fp, (w >> (int(row) & fp.bit_mask)) & fp.val_mask);
}

// one leaf of a ConjPred on the row's word
__device__ __forceinline__ bool conj_test(const ConjTest &t, uint64_t w) {
    const uint64_t x = w ^ (uint64_t(int64_t(w) >> 63) & t.fmask);
    // (the one-subtraction form of range_pass)
    const uint64_t f2 = t.flip ^ 0x8000000000000000ull;
    const uint64_t nbase = f2 - (uint64_t(t.lo) + 0x8000000000000000ull);
    const uint64_t width = uint64_t(t.hi) - uint64_t(t.lo) + 1ull;
    const bool full = width == 0 && t.lo <= t.hi;
    return (((x + nbase) < width) || full) != (t.negate != 0);
}
// the arithmetic step of a test (wave-uniform dispatch; operands vetted by the host: no fault possible)
__device__ __forceinline__ uint64_t conj_pre(const ConjTest &t, uint64_t w) {
    const uint64_t a = t.pre_rev ? t.pre_lit : w, b = t.pre_rev ? w : t.pre_lit;
    if (t.pre_dt == NQE_FLOAT64) {
        const double x = u2d(a), y = u2d(b);
        switch (t.pre) {
        case NQE_OP_PLUS: return d2u(x + y);
        case NQE_OP_MINUS: return d2u(x - y);
        case NQE_OP_MULTIPLY: return d2u(x * y);
        default: return d2u(x / y); // (Float64 modulus is not admitted)
        }
    }
    switch (t.pre) {
    case NQE_OP_PLUS: return a + b;
    case NQE_OP_MINUS: return a - b;
    case NQE_OP_MULTIPLY: return a * b;
    default: break;
    }
    if (t.pre_dt == NQE_INT64) return apply_binary<false>(t.pre == NQE_OP_DIVIDE ? NQE_OP_DIVIDE : NQE_OP_MODULOS, NQE_INT64, a, b, t.pre_aux, false, nullptr);
    return apply_binary<false>(t.pre == NQE_OP_DIVIDE ? NQE_OP_DIVIDE : NQE_OP_MODULOS, NQE_UINT64, a, b, t.pre_aux, false, nullptr);
}

// NSRC: how many distinct words a test may name (the selects are VALU work the aggregate's streaming kernel feels: with the
// four-way select `id < N/2 and v > 10` ran at 4.5 TB/s, with the three words it actually has at 5.3)
// GENERAL = false: the caller guarantees c.general == 0 (the aggregate's lean PRED = 4 instances: the general form's registers made
// every one of them spill; it runs in the PRED = 5 instances instead)
template <int NSRC, bool GENERAL = true>
__device__ __forceinline__ bool conj_pass(const ConjPred &c, uint64_t w0, uint64_t w1 = 0, uint64_t w2 = 0, uint64_t w3 = 0) {
    // (constant test indices throughout: a helper taking the index at run time sent the whole ConjPred to scratch memory — 3.7x slower)
#define NQE_CONJ_WORD(T) ((NSRC > 3 && c.t[T].src == 3) ? w3 : (NSRC > 2 && c.t[T].src == 2) ? w2 : (NSRC > 1 && c.t[T].src == 1) ? w1 : w0)
    if (GENERAL && c.general) { // wave-uniform: tests with an arithmetic step, combined through the truth table
#define NQE_CONJ_GTEST(T) (conj_test(c.t[T], c.t[T].pre ? conj_pre(c.t[T], NQE_CONJ_WORD(T)) : NQE_CONJ_WORD(T)) ? (1u << T) : 0u)
        uint32_t idx = NQE_CONJ_GTEST(0);
        if (c.n > 1) idx |= NQE_CONJ_GTEST(1);
        if (c.n > 2) idx |= NQE_CONJ_GTEST(2);
        if (c.n > 3) idx |= NQE_CONJ_GTEST(3);
#undef NQE_CONJ_GTEST
        return (c.truth >> idx) & 1u;
    }
#define NQE_CONJ_TEST(T) conj_test(c.t[T], NQE_CONJ_WORD(T))
    const bool r0 = NQE_CONJ_TEST(0), r1 = NQE_CONJ_TEST(1);
    bool r = c.is_or ? (r0 || r1) : (r0 && r1);
    if (c.n > 2) { // wave-uniform; the two-test form stays straight-line code (a branch per test cost it 15 %)
        const bool r2 = NQE_CONJ_TEST(2);
        r = c.is_or ? (r || r2) : (r && r2);
        if (c.n > 3) {
            const bool r3 = NQE_CONJ_TEST(3);
            r = c.is_or ? (r || r3) : (r && r3);
        }
    }
#undef NQE_CONJ_TEST
#undef NQE_CONJ_WORD
    return r;
}

// wave-level exclusive prefix sum of a 32-bit value (wave64, DPP-free shuffle version)
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t &total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane_id() >= d) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

} // namespace nqe
