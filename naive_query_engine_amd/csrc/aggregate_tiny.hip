// aggregate_tiny.hip — `group by col % m`, m <= 4 (the reference's own aggregate query: `… group by id % 3`, src/main.rs:36-40,
// README.md:105-111), with the group state in REGISTERS.
//
// Three groups make every LDS-table design an atomics benchmark: rows of a thread are 1024 rows apart, so its key changes on every
// row (no run to cache), and the replicated direct-mapped tables of the streaming kernel (aggregate_fast_kernel.hpp) take 3-5 LDS
// atomics per row — `count(id), sum(age), sum(score), avg(score), max(score), min(score) … group by id % 3` ran at 0.67 of 8 TB/s on
// 24 B/row, LDS-atomic-bound.  With at most four keys a lane keeps ONE accumulator set per key — count, a sum per value column,
// min / max of the last column — and updates every set on every row under a select on `key == k`: ~12 vector instructions per key
// and row against a budget of ~130 at 24 B/row, no LDS traffic at all.  The sets are folded across the wave (shuffles), across the
// workgroup (LDS) and leave as per-workgroup partials in the layout of AggArgs::partials (agg_fold_partials_kernel folds them).
// Keys outside [0, m) — a negative Int64 — raise NQE_FLAG_OOB: the host redoes the pass with the streaming kernel (and remembers).
#include "aggregate_common.hpp"
#include "device_utils.hpp"
#include "nqe_internal.hpp"

#ifndef NQE_TINY_PAIR
#define NQE_TINY_PAIR 1 // rows in pairs, one 16-byte load per pair and column (round 6); 0: 8-byte loads, a lane's rows 1024 apart (round 5, A/B)
#endif
#ifndef NQE_TINY_U
#define NQE_TINY_U (NVT == 1 ? 4 : NVT == 2 ? (MM ? 2 : 4) : (TG_K == 3 && SH0) ? (MM ? (NQE_TINY_PAIR ? 2 : 3) : 4) : 2) // (beside min / max and several value columns more rows spill inside the loop: checked per instance in the ISA)
#endif
namespace nqe {
namespace agg {
namespace {

// (registers: K accumulator sets of 1 + 2 NVT (+ 4) words and two tiles of U x (1 + NVT) 8-byte words — three value columns with four sets and
// four rows per tile spilled 70-180 VGPRs: K = 3 for m <= 3, two rows per tile beside several value columns, no registers for a value
// column that IS the key column)

// x mod m for x >= 0, m in 1..4, without the 64-bit multiply-high of a general magic division: 4^i = 1 (mod 3), so x mod 3 is the sum of
// x's base-4 digits mod 3 = (2 * popcount(odd bits) + popcount(even bits)) mod 3
__device__ __forceinline__ uint32_t mod_small(uint64_t x, uint32_t m) {
    if (m == 3) {
        const uint32_t s = uint32_t(__popcll(x & 0x5555555555555555ull)) + 2u * uint32_t(__popcll(x & 0xAAAAAAAAAAAAAAAAull)); // <= 96
        return s - 3u * ((s * 43691u) >> 17); // s / 3 for s < 2^16
    }
    return uint32_t(x) & (m - 1u); // 1, 2, 4
}

// PRED: 0 none, 1 an integer range test on the key column.  NVT value columns, MM: min / max of the LAST one.  TG_K accumulator sets
// (>= m).  SH0: the FIRST value column is the key column itself (`count(id) … group by id % 3`): its words are the key words.
template <int PRED, int NVT, bool MM, int TG_K, bool SH0>
__global__ void __launch_bounds__(AGG_BLOCK) agg_tiny_groups_kernel(AggArgs a, FastPred fp, uint32_t m, uint32_t unpack_tiles, int *flags) {
    constexpr int TG_U = NQE_TINY_U; // rows per lane per register tile
    constexpr int NL = SH0 ? NVT - 1 : NVT; // value columns that are loaded
    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ valp[NVT];
    int vdt[NVT];
    bool nsum[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        valp[j] = static_cast<const uint64_t *>(a.val[j].values);
        vdt[j] = a.val[j].dtype;
        nsum[j] = a.need_sum[j] != 0;
    }
    const bool key_signed = a.key_src.dtype == NQE_INT64;
    uint32_t cnt[TG_K], nanm = 0;
    double sum[NVT][TG_K], mn[TG_K], mx[TG_K];
#pragma unroll
    for (int k = 0; k < TG_K; ++k) {
        cnt[k] = 0;
#pragma unroll
        for (int j = 0; j < NVT; ++j) sum[j][k] = 0.0;
        mn[k] = DBL_MAX;
        mx[k] = -DBL_MAX;
    }
    bool oob = false;
    struct Tile {
        uint64_t kw[TG_U], vw[NL > 0 ? NL : 1][TG_U];
    };
    const int64_t n = a.n, last = n - 1, step = int64_t(AGG_BLOCK) * TG_U, stride = int64_t(gridDim.x) * step;
#if NQE_TINY_PAIR
    // rows in pairs: lane t takes rows 2 t, 2 t + 1 of every 2048-row half-tile — ONE 16-byte load per pair and column instead of two
    // 8-byte loads 8 KB apart (three 8 GB streams: 4.19 -> 3.98 ms for the README query, 4.13 -> 3.97 for C1's list, A/B on one box;
    // the host takes this kernel only for 16-byte aligned columns: AggRun::tier_streaming).  Nothing here depends on which rows a lane
    // holds: the accumulators are per key, not per run.
    typedef unsigned long long tiny_u64x2 __attribute__((ext_vector_type(2)));
    auto row_of = [&](int64_t base, int u) { return base + int64_t(u / 2) * (2 * AGG_BLOCK) + 2 * int64_t(threadIdx.x) + (u & 1); };
#else
    auto row_of = [&](int64_t base, int u) { return base + int64_t(u) * AGG_BLOCK + threadIdx.x; };
#endif
    auto load = [&](Tile &t, int64_t base) {
        if (base + step <= n) { // a whole tile: scalar tile pointer + the lane's 32-bit offsets
#if NQE_TINY_PAIR
            static_assert(TG_U % 2 == 0, "pairs");
#pragma unroll
            for (int u = 0; u < TG_U; u += 2) {
                const uint32_t o = uint32_t(u / 2) * (2 * AGG_BLOCK) + 2 * threadIdx.x;
                const tiny_u64x2 k2 = __builtin_nontemporal_load(reinterpret_cast<const tiny_u64x2 *>(keyp + base + o));
                t.kw[u] = k2.x;
                t.kw[u + 1] = k2.y;
#pragma unroll
                for (int j = SH0 ? 1 : 0; j < NVT; ++j) {
                    const tiny_u64x2 v2 = __builtin_nontemporal_load(reinterpret_cast<const tiny_u64x2 *>(valp[j] + base + o));
                    t.vw[SH0 ? j - 1 : j][u] = v2.x;
                    t.vw[SH0 ? j - 1 : j][u + 1] = v2.y;
                }
            }
#else
#pragma unroll
            for (int u = 0; u < TG_U; ++u) {
                const uint32_t o = uint32_t(u) * AGG_BLOCK + threadIdx.x;
                t.kw[u] = __builtin_nontemporal_load(keyp + base + o);
#pragma unroll
                for (int j = SH0 ? 1 : 0; j < NVT; ++j) t.vw[SH0 ? j - 1 : j][u] = __builtin_nontemporal_load(valp[j] + base + o);
            }
#endif
        } else {
#pragma unroll
            for (int u = 0; u < TG_U; ++u) {
                int64_t row = row_of(base, u);
                row = row < last ? row : last;
                t.kw[u] = __builtin_nontemporal_load(keyp + row);
#pragma unroll
                for (int j = SH0 ? 1 : 0; j < NVT; ++j) t.vw[SH0 ? j - 1 : j][u] = __builtin_nontemporal_load(valp[j] + row);
            }
        }
    };
    // Per row: ONE packed count update (a 64-bit word of TG_K fields of 64 / TG_K bits: `1 << (bits * key)` — unpacked every unpack_tiles (4096) tiles, long
    // before a field can carry), per key and summed column a selected add (2 v_cndmask + v_add_f64), per key ONE select for both min and max: the
    // row's value with its high word forced to a quiet NaN when the key does not match — v_min_f64 / v_max_f64 return their other operand for a
    // NaN, which is also exactly what a NaN VALUE must do to min (max.rs:38-50; its mark for max is kept per key).
    uint64_t packed = 0;
    uint32_t tiles_packed = 0;
    constexpr int CBITS = 64 / TG_K;
    auto unpack = [&]() {
#pragma unroll
        for (int k = 0; k < TG_K; ++k) cnt[k] += uint32_t((packed >> (CBITS * k)) & ((1ull << CBITS) - 1ull));
        packed = 0;
        tiles_packed = 0;
    };
    auto process = [&](const Tile &t, int64_t base) {
#pragma unroll
        for (int u = 0; u < TG_U; ++u) {
            const int64_t row = row_of(base, u);
            bool ok = row < n;
            if (PRED == 1) ok = ok && range_pass(fp, t.kw[u]);
            oob = oob || (ok && key_signed && int64_t(t.kw[u]) < 0);
            const uint32_t key = ok ? mod_small(t.kw[u], m) : 0xFFu; // (a row that does not take part matches no key)
            packed += ok ? (1ull << (CBITS * key)) : 0ull;
            double x[NVT];
#pragma unroll
            for (int j = 0; j < NVT; ++j) x[j] = word_as_f64((SH0 && j == 0) ? t.kw[u] : t.vw[SH0 ? (j > 0 ? j - 1 : 0) : j][u], vdt[j]);
            if (MM) nanm |= (x[NVT - 1] != x[NVT - 1]) ? (1u << (key & 31u)) : 0u; // (key 0xFF: bit 31, never read)
            const uint64_t xlw = MM ? d2u(x[NVT - 1]) : 0ull;
#pragma unroll
            for (int k = 0; k < TG_K; ++k) {
                const bool sel = key == uint32_t(k);
#pragma unroll
                for (int j = 0; j < NVT; ++j)
                    if (nsum[j]) sum[j][k] += sel ? x[j] : 0.0; // (-0.0 + 0.0 = 0.0: the sum of a group of negative zeros loses its sign — equal under ==)
                if (MM) {
                    const double xk = u2d(sel ? xlw : (xlw | 0x7FF8000000000000ull));
                    mn[k] = fmin(mn[k], xk);
                    mx[k] = fmax(mx[k], xk);
                }
            }
        }
        if (++tiles_packed == unpack_tiles) unpack(); // (unpack_tiles <= 4096: 4096 tiles x TG_U rows < 2^16 — no field of the packed counter has carried; AggSwitches::tiny_unpack_tiles)
    };
    {
        Tile A, B;
        int64_t base = int64_t(blockIdx.x) * step;
        if (base < n) {
            load(A, base);
            for (;;) {
                const int64_t nb = base + stride;
                load(B, nb < n ? nb : base); // (past the end: the tile in hand again — an unconditional fetch keeps the registers out of a phi)
                process(A, base);
                if (nb >= n) break;
                base = nb + stride;
                load(A, base < n ? base : nb);
                process(B, nb);
                if (base >= n) break;
            }
        }
    }
    unpack();
    // ---- fold: lanes -> wave (shuffles) -> workgroup (LDS) -> this workgroup's partial
    __shared__ double s_sum[AGG_BLOCK / 64][NVT][TG_K], s_mn[AGG_BLOCK / 64][TG_K], s_mx[AGG_BLOCK / 64][TG_K];
    __shared__ uint32_t s_cnt[AGG_BLOCK / 64][TG_K], s_nan[AGG_BLOCK / 64], s_oob[AGG_BLOCK / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int k = 0; k < TG_K; ++k) {
            cnt[k] += uint32_t(__shfl_xor(int(cnt[k]), d, 64));
#pragma unroll
            for (int j = 0; j < NVT; ++j) sum[j][k] += __shfl_xor(sum[j][k], d, 64);
            if (MM) {
                mn[k] = fmin(mn[k], __shfl_xor(mn[k], d, 64));
                mx[k] = fmax(mx[k], __shfl_xor(mx[k], d, 64));
            }
        }
        nanm |= uint32_t(__shfl_xor(int(nanm), d, 64));
    }
    const uint64_t anyoob = __ballot(oob);
    const int wv = int(threadIdx.x) / 64;
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < TG_K; ++k) {
            s_cnt[wv][k] = cnt[k];
#pragma unroll
            for (int j = 0; j < NVT; ++j) s_sum[wv][j][k] = sum[j][k];
            s_mn[wv][k] = mn[k];
            s_mx[wv][k] = mx[k];
        }
        s_nan[wv] = nanm;
        s_oob[wv] = anyoob ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x < m) {
        const uint32_t k = threadIdx.x;
        uint32_t c = 0, nn = 0, ob = 0;
        double sm[NVT], lo = DBL_MAX, hi = -DBL_MAX;
#pragma unroll
        for (int j = 0; j < NVT; ++j) sm[j] = 0.0;
        for (int w = 0; w < AGG_BLOCK / 64; ++w) {
            c += s_cnt[w][k];
#pragma unroll
            for (int j = 0; j < NVT; ++j) sm[j] += s_sum[w][j][k];
            lo = fmin(lo, s_mn[w][k]);
            hi = fmax(hi, s_mx[w][k]);
            nn |= s_nan[w];
            ob |= s_oob[w];
        }
        if (ob && k == 0) atomicOr(&flags[NQE_FLAG_OOB], 1);
        const size_t cells = size_t(gridDim.x) * m;
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            double *__restrict__ ps = reinterpret_cast<double *>(a.partials) + size_t(j) * ((cells * 28 + 7) / 8);
            uint32_t *__restrict__ pc = reinterpret_cast<uint32_t *>(ps + 3 * cells);
            const size_t o = size_t(blockIdx.x) * m + k;
            const bool lastc = MM && j == NVT - 1;
            ps[o] = sm[j];
            ps[cells + o] = lastc ? lo : DBL_MAX;
            ps[2 * cells + o] = lastc ? hi : -DBL_MAX;
            pc[o] = c | ((lastc && ((nn >> k) & 1u)) ? NAN_BIT : 0u);
        }
    }
}

template <int PRED, int NVT, bool MM, int K> TinyGroupsKernel pick_tiny_sh(bool sh0) { return sh0 ? agg_tiny_groups_kernel<PRED, NVT, MM, K, true> : agg_tiny_groups_kernel<PRED, NVT, MM, K, false>; }
template <int PRED, int NVT, bool MM> TinyGroupsKernel pick_tiny_k(uint32_t m, bool sh0) { return m <= 3 ? pick_tiny_sh<PRED, NVT, MM, 3>(sh0) : pick_tiny_sh<PRED, NVT, MM, 4>(sh0); }
template <int PRED, int NVT> TinyGroupsKernel pick_tiny_mm(bool mm, uint32_t m, bool sh0) { return mm ? pick_tiny_k<PRED, NVT, true>(m, sh0) : pick_tiny_k<PRED, NVT, false>(m, sh0); }
template <int PRED> TinyGroupsKernel pick_tiny_nv(int nv, bool mm, uint32_t m, bool sh0) {
    return nv == 1 ? pick_tiny_mm<PRED, 1>(mm, m, sh0) : nv == 2 ? pick_tiny_mm<PRED, 2>(mm, m, sh0) : pick_tiny_mm<PRED, 3>(mm, m, sh0);
}

} // namespace

TinyGroupsKernel pick_tiny_groups_kernel(int pred, int nv, bool mm, uint32_t m, bool share0) {
    return pred == 0 ? pick_tiny_nv<0>(nv, mm, m, share0) : pick_tiny_nv<1>(nv, mm, m, share0);
}

} // namespace agg
} // namespace nqe

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE((nqe::agg::agg_tiny_groups_kernel<0, 1, true, 3, false>));
