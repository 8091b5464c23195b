// Field → Int64 / Float64 / Boolean conversion of the CSV reader (host + device, header only).
//
// The reference reads CSV through arrow-rs 13's csv::Reader (src/datasource/csv.rs:60-69), whose primitive parsers are
// lexical-core's: Int64 = [+-]digits with overflow → error, Float64 = correctly rounded decimal → binary64.  arrow-rs and
// lexical-core are third-party dependencies that are not under /root/reference (Cargo.lock: arrow 13.0.0,
// lexical-core 0.8.x); this restates their published behaviour:
//   * grammar: [+-] (digits [. digits*] | . digits+) [(e|E) [+-] digits+]  |  [+-] (nan | inf | infinity), case-insensitive;
//   * result: the binary64 nearest to the exact decimal value, ties to even.
// Conversion: Clinger's exact fast path (≤ 19 significant digits, value < 2^53, |exponent| ≤ 22) and otherwise exact
// big-integer arithmetic on the first 40 significant digits (W·5^e, or the 65-bit quotient of W·2^k / 5^-e with a sticky
// remainder), rounded once.  When non-zero digits follow the 40th, W·10^E and (W+1)·10^E are both rounded; if they differ
// the input is compared digit by digit with their midpoint, whose decimal expansion is generated exactly (so the result
// is correctly rounded for any number of digits).  The same code runs in the CPU unit test
// (tests/cpp/test_csv_parse.cpp, against glibc strtod) and in the device kernel.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define NQE_HD __host__ __device__ inline
#define NQE_HD_COLD __host__ __device__ __noinline__ // big-integer paths: kept out of the kernels' hot code
#else
#define NQE_HD inline
#define NQE_HD_COLD inline
#endif

namespace nqe {
namespace csvp {

constexpr int BIG_LIMBS = 40; // 1280 bits: 40 digits (133 bits) · 5^400 (929 bits) + headroom

struct Big {
    uint32_t w[BIG_LIMBS];
    int n; // used limbs (w[n-1] != 0), 0 = zero
};

NQE_HD void big_set_u64(Big &b, uint64_t v) {
    b.n = 0;
    if (v & 0xffffffffu || v >> 32) {
        b.w[0] = uint32_t(v);
        b.w[1] = uint32_t(v >> 32);
        b.n = b.w[1] ? 2 : 1;
    }
}
NQE_HD void big_mul_small(Big &b, uint32_t m) {
    uint64_t carry = 0;
    for (int i = 0; i < b.n; ++i) {
        uint64_t t = uint64_t(b.w[i]) * m + carry;
        b.w[i] = uint32_t(t);
        carry = t >> 32;
    }
    if (carry && b.n < BIG_LIMBS) b.w[b.n++] = uint32_t(carry);
}
NQE_HD void big_add_small(Big &b, uint32_t a) {
    uint64_t carry = a;
    for (int i = 0; i < b.n && carry; ++i) {
        uint64_t t = uint64_t(b.w[i]) + carry;
        b.w[i] = uint32_t(t);
        carry = t >> 32;
    }
    if (carry && b.n < BIG_LIMBS) b.w[b.n++] = uint32_t(carry);
}
NQE_HD void big_mul_pow5(Big &b, int e) {
    while (e >= 13) {
        big_mul_small(b, 1220703125u); // 5^13
        e -= 13;
    }
    uint32_t m = 1;
    for (int i = 0; i < e; ++i) m *= 5;
    if (m > 1) big_mul_small(b, m);
}
NQE_HD int big_bitlen(const Big &b) {
    if (b.n == 0) return 0;
    uint32_t top = b.w[b.n - 1];
    int bits = 0;
    while (top) {
        ++bits;
        top >>= 1;
    }
    return (b.n - 1) * 32 + bits;
}
NQE_HD void big_shl(Big &b, int k) {
    if (b.n == 0 || k == 0) return;
    const int limbs = k / 32, bits = k % 32;
    int nn = b.n + limbs + 1;
    if (nn > BIG_LIMBS) nn = BIG_LIMBS;
    for (int i = nn - 1; i >= 0; --i) {
        const int s = i - limbs;
        uint32_t lo = (s >= 0 && s < b.n) ? b.w[s] : 0u;
        uint32_t lo2 = (s - 1 >= 0 && s - 1 < b.n) ? b.w[s - 1] : 0u;
        b.w[i] = bits ? ((lo << bits) | (lo2 >> (32 - bits))) : lo;
    }
    b.n = nn;
    while (b.n > 0 && b.w[b.n - 1] == 0) --b.n;
}
NQE_HD void big_shr1(Big &b) {
    for (int i = 0; i < b.n; ++i) b.w[i] = (b.w[i] >> 1) | (i + 1 < b.n ? (b.w[i + 1] << 31) : 0u);
    while (b.n > 0 && b.w[b.n - 1] == 0) --b.n;
}
NQE_HD int big_cmp(const Big &a, const Big &b) {
    if (a.n != b.n) return a.n < b.n ? -1 : 1;
    for (int i = a.n - 1; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}
NQE_HD void big_sub(Big &a, const Big &b) { // a -= b, a >= b
    int64_t borrow = 0;
    for (int i = 0; i < a.n; ++i) {
        int64_t t = int64_t(a.w[i]) - (i < b.n ? int64_t(b.w[i]) : 0) - borrow;
        borrow = t < 0;
        a.w[i] = uint32_t(t + (borrow ? (int64_t(1) << 32) : 0));
    }
    while (a.n > 0 && a.w[a.n - 1] == 0) --a.n;
}
// top 64 bits of b (bit length bl >= 1) left-aligned; *sticky |= any lower bit set
NQE_HD uint64_t big_top64(const Big &b, int bl, bool *sticky) {
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) {
        const int bit = bl - 1 - i;
        uint64_t v = 0;
        if (bit >= 0) v = (b.w[bit / 32] >> (bit % 32)) & 1u;
        m = (m << 1) | v;
    }
    const int low = bl - 64; // bits [0, low) are below the window
    for (int i = 0; i < b.n && !*sticky; ++i) {
        if ((i + 1) * 32 <= low) {
            if (b.w[i]) *sticky = true;
        } else if (i * 32 < low) {
            if (b.w[i] & ((1u << (low - i * 32)) - 1u)) *sticky = true;
        }
    }
    return m;
}

// value = m · 2^e (m has its top bit set, bit 63), plus a sticky "more non-zero bits below" → nearest binary64, ties to even
NQE_HD double make_double(uint64_t m, int e, bool sticky, bool neg) {
    uint64_t bits;
    const int ue = e + 63; // unbiased exponent of the leading bit
    if (m == 0) bits = 0;
    else if (ue > 1023) bits = 0x7ff0000000000000ull;
    else {
        int r = 11; // bits dropped below the 53-bit significand
        if (ue < -1022) r += -1022 - ue;
        uint64_t sig, guard;
        if (r >= 65) {
            sig = 0;
            guard = 0;
            sticky = true;
        } else if (r == 64) {
            sig = 0;
            guard = m >> 63;
            sticky = sticky || (m & 0x7fffffffffffffffull);
        } else {
            sig = m >> r;
            guard = (m >> (r - 1)) & 1u;
            sticky = sticky || (m & ((1ull << (r - 1)) - 1ull));
        }
        if (guard && (sticky || (sig & 1u))) ++sig;
        if (ue < -1022) bits = sig; // subnormal (a carry into bit 52 yields the smallest normal, as it must)
        else bits = (uint64_t(ue + 1022) << 52) + sig; // sig carries the hidden bit: + 1 on the exponent field
        if (bits >= 0x7ff0000000000000ull) bits = 0x7ff0000000000000ull;
    }
    if (neg) bits |= 0x8000000000000000ull;
    double d;
    __builtin_memcpy(&d, &bits, 8);
    return d;
}

NQE_HD bool is_digit(char c) { return c >= '0' && c <= '9'; }
NQE_HD char lower(char c) { return (c >= 'A' && c <= 'Z') ? char(c + 32) : c; }
NQE_HD bool ieq(const char *s, int len, const char *word, int wl) {
    if (len != wl) return false;
    for (int i = 0; i < len; ++i)
        if (lower(s[i]) != word[i]) return false;
    return true;
}

// lexical-core i64: [+-] digits+, no other characters; false on overflow
NQE_HD bool parse_i64(const char *s, int len, int64_t *out) {
    int i = 0;
    bool neg = false;
    if (i < len && (s[i] == '-' || s[i] == '+')) neg = s[i++] == '-';
    if (i >= len) return false;
    uint64_t v = 0;
    const uint64_t lim = neg ? 0x8000000000000000ull : 0x7fffffffffffffffull;
    for (; i < len; ++i) {
        if (!is_digit(s[i])) return false;
        const uint64_t d = uint64_t(s[i] - '0');
        if (v > (lim - d) / 10) return false;
        v = v * 10 + d;
    }
    *out = neg ? int64_t(0 - v) : int64_t(v);
    return true;
}

// arrow-rs 13 parse_bool: "true"/"false", ASCII case-insensitive
NQE_HD bool parse_bool(const char *s, int len, bool *out) {
    if (ieq(s, len, "true", 4)) { *out = true; return true; }
    if (ieq(s, len, "false", 5)) { *out = false; return true; }
    return false;
}

// nearest binary64 to W · 10^E, exactly (one rounding)
NQE_HD_COLD double scaled_to_double(Big W, int E, bool neg) {
    bool sticky = false;
    if (W.n == 0) return neg ? -0.0 : 0.0;
    if (E >= 0) {
        big_mul_pow5(W, E);
        const int bl = big_bitlen(W);
        const uint64_t m = big_top64(W, bl, &sticky);
        return make_double(m, bl - 64 + E, sticky, neg);
    }
    Big D;
    big_set_u64(D, 1);
    big_mul_pow5(D, -E);
    // scale so that the quotient has 64 or 65 bits: bitlen(W·2^k) = bitlen(D) + 64
    const int k = big_bitlen(D) + 64 - big_bitlen(W);
    if (k >= 0) big_shl(W, k);
    else big_shl(D, -k);
    // restoring division, quotient bits 64..0
    Big T = D;
    big_shl(T, 64);
    uint64_t q_hi = 0, q = 0; // q_hi = bit 64
    for (int bit = 64; bit >= 0; --bit) {
        if (big_cmp(W, T) >= 0) {
            big_sub(W, T);
            if (bit == 64) q_hi = 1;
            else q |= 1ull << bit;
        }
        big_shr1(T);
    }
    if (W.n != 0) sticky = true;
    // value = (q_hi·2^64 + q) · 2^(E - k)
    if (q_hi) {
        sticky = sticky || (q & 1ull);
        return make_double((1ull << 63) | (q >> 1), E - k + 1, sticky, neg);
    }
    int lz = 0;
    while (q && !(q >> 63)) { // q >= 2^63 by construction; kept for safety
        q <<= 1;
        ++lz;
    }
    return make_double(q, E - k - lz, sticky, neg);
}

// the exact conversion of a syntactically valid mantissa s[mant_begin, mant_end) (see parse_f64 for the arguments)
NQE_HD_COLD double parse_f64_exact(const char *s, int mant_begin, int mant_end, int nd, uint64_t w19, long long e10, bool dropped_nonzero, bool neg) {
    double result = 0.0;
    double *out = &result;
    // ---- exact path.  W = the first min(nd, 40) significant digits as an integer; value = W · 10^E (+ dropped digits)
    Big W;
    W.n = 0;
    int kept = 0;
    long long E = e10;
    if (nd <= 19) {
        big_set_u64(W, w19);
        kept = nd;
    } else {
        // re-scan the mantissa: the first scan treated digits 20.. as dropped; rebuild the scale for the ones kept here
        bool dot = false, lead = true;
        for (int k = mant_begin; k < mant_end; ++k) {
            const char c = s[k];
            if (c == '.') { dot = true; continue; }
            if (lead && c == '0') continue;
            lead = false;
            if (kept < 40) {
                big_mul_small(W, 10);
                if (W.n == 0) { if (c != '0') big_set_u64(W, uint64_t(c - '0')); }
                else big_add_small(W, uint32_t(c - '0'));
                if (kept >= 19) --E; // undo the ++e10 of a dropped integer digit / scale a kept fraction digit by 10^-1
                ++kept;
            }
            (void)dot;
        }
    }
    if (E > 400) {
        *out = make_double(1ull << 63, 2000, false, neg); // overflows to infinity
        return result;
    }
    if (E < -460) { // below half the smallest subnormal whatever W (< 10^40) is
        *out = neg ? -0.0 : 0.0;
        return result;
    }
    if (!dropped_nonzero) {
        *out = scaled_to_double(W, int(E), neg);
        return result;
    }
    // Non-zero digits beyond the 40th: the value lies strictly between W·10^E and (W+1)·10^E.  If both round to the same
    // double that is the answer; otherwise they are neighbours and the input is compared digit by digit with their
    // midpoint (whose decimal expansion is generated exactly, one digit at a time).
    const double lo = scaled_to_double(W, int(E), false);
    Big W1 = W;
    big_add_small(W1, 1);
    const double hi = scaled_to_double(W1, int(E), false);
    uint64_t lob, hib;
    __builtin_memcpy(&lob, &lo, 8);
    __builtin_memcpy(&hib, &hi, 8);
    double r = lo;
    if (lob != hib) {
        const uint64_t frac = lob & 0xfffffffffffffull;
        const int ef = int(lob >> 52) & 0x7ff;
        const uint64_t M = ef ? (frac | (1ull << 52)) : frac;
        const int q = ef ? ef - 1075 : -1074; // lo = M · 2^q; midpoint = (2M+1) · 2^(q-1)
        Big num, den;
        big_set_u64(num, 2 * M + 1);
        big_set_u64(den, 1);
        if (q - 1 >= 0) big_shl(num, q - 1);
        else big_shl(den, 1 - q);
        const long long K = E + kept; // input = 0.d1d2d3… · 10^K
        if (K >= 0) {
            big_mul_pow5(den, int(K));
            big_shl(den, int(K));
        } else {
            big_mul_pow5(num, int(-K));
            big_shl(num, int(-K));
        }
        int cmp = 0; // sign of (input - midpoint)
        bool lead = true;
        for (int k = mant_begin; k < mant_end && cmp == 0; ++k) {
            const char c = s[k];
            if (c == '.') continue;
            if (lead && c == '0') continue;
            lead = false;
            big_mul_small(num, 10);
            int dm = 0;
            while (big_cmp(num, den) >= 0) {
                big_sub(num, den);
                ++dm;
            }
            const int di = c - '0';
            if (di != dm) cmp = di < dm ? -1 : 1;
        }
        if (cmp == 0 && num.n != 0) cmp = -1; // the midpoint has further non-zero digits
        if (cmp > 0 || (cmp == 0 && (M & 1ull))) r = hi;
    }
    *out = neg ? -r : r;
    return result;
}

NQE_HD bool parse_f64(const char *s, int len, double *out) {
    int i = 0;
    bool neg = false;
    if (i < len && (s[i] == '-' || s[i] == '+')) neg = s[i++] == '-';
    if (i >= len) return false;
    if (ieq(s + i, len - i, "nan", 3)) {
        uint64_t bits = 0x7ff8000000000000ull | (neg ? 0x8000000000000000ull : 0);
        __builtin_memcpy(out, &bits, 8);
        return true;
    }
    if (ieq(s + i, len - i, "inf", 3) || ieq(s + i, len - i, "infinity", 8)) {
        uint64_t bits = 0x7ff0000000000000ull | (neg ? 0x8000000000000000ull : 0);
        __builtin_memcpy(out, &bits, 8);
        return true;
    }
    // ---- grammar + digit scan: w19 = first 19 significant digits, nd = significant digits seen,
    // e10 = power of ten that scales the integer formed by the kept digits
    const int mant_begin = i;
    uint64_t w19 = 0;
    int nd = 0, ndigits_any = 0;
    long long e10 = 0;
    bool seen_dot = false, dropped_nonzero = false;
    int p = i;
    while (p < len) {
        const char c = s[p];
        if (c == '.') {
            if (seen_dot) return false;
            seen_dot = true;
            ++p;
            continue;
        }
        if (c < '0' || c > '9') break;
        ++p;
        ++ndigits_any;
        if (nd == 0 && c == '0') { // leading zero: no significance, but a fraction zero shifts the scale
            if (seen_dot) --e10;
            continue;
        }
        if (nd < 19) {
            w19 = w19 * 10 + uint64_t(c - '0');
            if (seen_dot) --e10;
        } else {
            if (!seen_dot) ++e10; // an integer digit that is not kept scales the kept ones
            if (nd >= 40 && c != '0') dropped_nonzero = true;
        }
        ++nd;
    }
    const int mant_end = p;
    if (ndigits_any == 0) return false;
    if (p < len) {
        const char ec = s[p];
        if (!(ec == 'e' || ec == 'E')) return false;
        ++p;
        bool eneg = false;
        if (p < len) {
            const char sc = s[p];
            if (sc == '-') {
                eneg = true;
                ++p;
            } else if (sc == '+') ++p;
        }
        if (p >= len) return false;
        long long ex = 0;
        while (p < len) {
            const char c = s[p];
            if (c < '0' || c > '9') return false;
            if (ex < 100000) ex = ex * 10 + (c - '0');
            ++p;
        }
        e10 += eneg ? -ex : ex;
    }
    if (nd == 0) { // all digits zero
        *out = neg ? -0.0 : 0.0;
        return true;
    }
    // ---- Clinger fast path
    if (nd <= 19 && w19 < (1ull << 53) && e10 >= -22 && e10 <= 22) {
        const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        double d = double(w19);
        d = e10 < 0 ? d / p10[-e10] : d * p10[e10];
        *out = neg ? -d : d;
        return true;
    }
    *out = parse_f64_exact(s, mant_begin, mant_end, nd, w19, e10, dropped_nonzero, neg);
    return true;
}

} // namespace csvp
} // namespace nqe
