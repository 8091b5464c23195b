// aggregate.hip — PhysicalAggregatePlan::execute (reference: src/physical_plan/aggregate/mod.rs:113-222,
// group split :54-102, operators aggregate/{sum,avg,count,max,min}.rs) fused with the SelectionPlan
// below it and the key expression group_expr[0].
//
// The reference concatenates the input, materialises the key column, builds
// HashMap<key, Vec<row>> and then calls a virtual update(batch, idx) per row per aggregate.
// Here ONE streaming kernel reads each referenced column exactly once (coalesced, all loads of
// an iteration issued before first use), evaluates predicate and key in registers, and
// accumulates {count, sum, min, max} per (group, value column) in a per-workgroup LDS hash
// table (open addressing, Fibonacci hash, 64-bit CAS on the key, LDS atomics on the state).
// Rows whose key does not fit the workgroup table go straight to the global table.  At the end
// every workgroup merges its LDS table into the global open-addressing table with device-scope
// atomics; a collect + sort-by-key + finalize tail emits one row per group.
//
// Algorithmic HBM bytes per row = 8 B per distinct referenced column (16 B/row for
// `... from t where id < K group by id % 1024` with count/sum/avg/min/max over v).
//
// Semantics kept from the reference (quirk Q10): all accumulation in f64 (`val as f64`), count =
// non-null values, max starts at f64::MIN, min at f64::MAX, OrderedFloat NaN ordering (max → NaN
// if any NaN, min ignores NaN), NULL keys dropped, NULL-predicate rows contribute nothing.
// Sum order differs from the reference's sequential row order (atomics): ≤1e-9 relative.
#include <algorithm>
#include <cmath>
#include <cfloat>

#include <cstdlib>

#include "aggregate_common.hpp"
#include "aggregate_tail.hpp"

namespace nqe {

namespace {

using namespace agg;

// ------------------------------------------------------------------ grouped kernel
// History: with the 64-bit software divide inlined for predicate and key in each of the 4 unrolled rows the first
// version of this kernel was 30k instructions long and instruction-fetch bound (2.2 TB/s).  Now literal divisors use
// shift/mask or a magic multiply and only column÷column reaches the out-of-line divmod_general, so the general
// SimpleExpr evaluator is inlined again; the common shapes are still specialised:
//   PRED: 0 none | 1 `col cmp lit` (literal on either side, normalised on the host) | 2 Boolean bitmap
//         | 3 any other SimpleExpr
//   KEY : 0 plain column | 1 `col % ±2^k` | 2 any other SimpleExpr
//   PLAIN: every streamed source is an 8-byte column without a validity bitmap (no bitmap loads)
__device__ __forceinline__ bool cmp_lit(int op, int dt, uint64_t a, uint64_t b) {
    bool lt, eq;
    if (dt == NQE_INT64) { lt = (long long)a < (long long)b; eq = a == b; }
    else if (dt == NQE_FLOAT64) { double x = u2d(a), y = u2d(b); lt = x < y; eq = x == y; if (x != x || y != y) return op == NQE_OP_NOT_EQ; }
    else { lt = a < b; eq = a == b; }
    return op == NQE_OP_EQ ? eq : op == NQE_OP_NOT_EQ ? !eq : op == NQE_OP_LT ? lt : op == NQE_OP_LT_EQ ? (lt || eq)
           : op == NQE_OP_GT ? !(lt || eq) : !lt;
}

template <int PRED, int KEY, bool PLAIN>
__global__ void __launch_bounds__(AGG_BLOCK) agg_grouped_kernel(AggArgs a, GroupTable g, int *flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t cap = uint32_t(a.lds_cap);
    const uint32_t slots = cap + 1;
    uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t nvl = a.nv > 0 ? uint32_t(a.nv) : 1u;                 // value columns of THIS pass
    double *lsum = reinterpret_cast<double *>(lkeys + slots);            // [nvl][slots]
    uint64_t *lmn = reinterpret_cast<uint64_t *>(lsum + nvl * slots);    // [nvl][slots]
    uint64_t *lmx = lmn + nvl * slots;                                   // [nvl][slots]
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(lmx + nvl * slots);    // [nvl][slots]
    const uint64_t ORD_MAX = f64_to_ord(DBL_MAX), ORD_MIN = f64_to_ord(-DBL_MAX);
    __shared__ int lds_full_flag;
    volatile int *lds_full = &lds_full_flag;
    if (threadIdx.x == 0) lds_full_flag = 0;

    for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
        lkeys[s] = EMPTY_KEY;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (uint32_t(j) >= nvl) continue;
            lsum[j * slots + s] = 0.0;
            lmn[j * slots + s] = ORD_MAX;
            lmx[j * slots + s] = ORD_MIN;
            lcnt[j * slots + s] = 0;
        }
    }
    __syncthreads();

    // Per-thread run cache.  All rows a thread visits are congruent modulo blockDim (row = base + u*blockDim +
    // tid with base a multiple of blockDim*AGG_U), so for clustered keys — and for `id % m` over a row-number id
    // whenever m divides blockDim — consecutive rows of a thread carry the SAME key.  They are accumulated in
    // registers and written to the workgroup table only when the key changes (one flush per run instead of four
    // LDS atomics per row).  Random keys flush every row.
    bool full = false; // register copy of lds_full_flag (set by own failures, refreshed per tile)
    bool run_live = false;
    uint64_t run_key = 0;
    uint32_t rcnt[NV];
    double rsum[NV];
    uint64_t rmn[NV], rmx[NV];
    bool rnan[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        rcnt[j] = 0; rsum[j] = 0.0; rmn[j] = ORD_MAX; rmx[j] = ORD_MIN; rnan[j] = false;
    }
    auto flush_run = [&]() {
        // once this workgroup's table has rejected a key, later keys skip it: any split of the updates between
        // the LDS table and the global table is correct (the merge is additive), and a full table costs 48 probes
        int slot = full ? -1 : lds_find_or_insert(lkeys, run_key, cap, a.lds_shift);
        if (slot < 0 && !full) {
            full = true;
            *lds_full = 1;
        }
        int64_t gslot = slot < 0 ? global_find_or_insert(g, run_key, flags) : 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (j >= a.nv) continue;
            if (slot >= 0) {
                uint32_t o = uint32_t(j) * slots + uint32_t(slot);
                if (rcnt[j]) atomicAdd(&lcnt[o], rcnt[j]); // < 2^31 rows per workgroup, bit 31 is the NaN flag
                if (rnan[j]) atomicOr(&lcnt[o], NAN_BIT);
                if (a.need_sum[j] && rcnt[j]) unsafeAtomicAdd(&lsum[o], rsum[j]);
                if (a.need_minmax[j]) {
                    // read-before-atomic: once a group holds a few rows almost no run improves its extremes, and
                    // an LDS read costs a small fraction of a 64-bit LDS atomic.  A stale read only causes a
                    // redundant (still correct) atomic.
                    if (rmn[j] < lmn[o]) atomicMin((unsigned long long *)&lmn[o], (unsigned long long)rmn[j]);
                    if (rmx[j] > lmx[o]) atomicMax((unsigned long long *)&lmx[o], (unsigned long long)rmx[j]);
                }
            } else if (gslot >= 0) {
                global_update(g, gslot, a.v0 + j, rcnt[j], rsum[j], a.need_sum[j] != 0, rmn[j], rmx[j], a.need_minmax[j] != 0,
                              rnan[j]);
            }
            rcnt[j] = 0; rsum[j] = 0.0; rmn[j] = ORD_MAX; rmx[j] = ORD_MIN; rnan[j] = false;
        }
    };

    const uint64_t *keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *predp = static_cast<const uint64_t *>(a.pred_src.values);
    const int pred_op = a.pred.op[0], pred_dt = a.pred.op_dtype[0];
    const uint64_t pred_lit = a.pred.lit[0];
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;

    const int64_t step = int64_t(blockDim.x) * AGG_U;
    for (int64_t base = int64_t(blockIdx.x) * step; base < a.n; base += int64_t(gridDim.x) * step) {
        if (__hip_atomic_load(&flags[NQE_FLAG_TABLE_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break; // host retries
        full = full || *lds_full != 0;
        uint64_t kw[AGG_U], pw[AGG_U], vw[NV][AGG_U];
        // ---- load phase: every referenced word of this iteration is requested before any use
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * blockDim.x + threadIdx.x;
            bool in = row < a.n;
            if (PLAIN) {
                kw[u] = in ? keyp[row] : 0;
                pw[u] = ((PRED == 1 || PRED == 3) && in && !a.pred_shares_key) ? predp[row] : 0;
            } else {
                kw[u] = in ? load_word(a.key_src.values, a.key_src.dtype, row) : 0;
                pw[u] = ((PRED == 1 || PRED == 3) && in && !a.pred_shares_key) ? load_word(a.pred_src.values, a.pred_src.dtype, row) : 0;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j)
                vw[j][u] = (in && j < a.nv && a.val[j].values && !a.val_shares_key[j])
                               ? static_cast<const uint64_t *>(a.val[j].values)[row]
                               : 0;
        }
        // ---- compute phase
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * blockDim.x + threadIdx.x;
            bool pass = row < a.n;
            if (PRED == 1 || PRED == 3) {
                bool ok = pass && (PLAIN || row_valid(a.pred_src, row));
                uint64_t w = a.pred_shares_key ? kw[u] : pw[u];
                if (PRED == 1) pass = ok && cmp_lit(pred_op, pred_dt, w, pred_lit);
                else pass = ok && eval_simple(a.pred, w, ok, flags) != 0;
            } else if (PRED == 2) {
                pass = pass && get_bit(static_cast<const uint8_t *>(a.pred_src.values), row) && row_valid(a.pred_src, row);
            }
            bool kok = pass && (PLAIN || row_valid(a.key_src, row));
            uint64_t key;
            if (KEY == 0) key = kw[u];
            else if (KEY == 1) {
                uint64_t x = kw[u];
                bool neg = key_signed && (long long)x < 0;
                uint64_t ur = (neg ? 0ull - x : x) & key_mask;
                key = neg ? 0ull - ur : ur;
            } else key = eval_simple(a.key, kw[u], kok, flags);
            pass = kok;
            if (!pass) continue;
            if (!run_live || key != run_key) {
                if (run_live) flush_run();
                run_key = key;
                run_live = true;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (j >= a.nv) continue;
                if (!PLAIN && !row_valid(a.val[j], row)) continue;
                rcnt[j] += 1;
                if (a.need_sum[j] || a.need_minmax[j]) {
                    double x = word_as_f64(a.val_shares_key[j] ? kw[u] : vw[j][u], a.val[j].dtype);
                    rsum[j] += x;
                    if (x != x) rnan[j] = true;
                    else {
                        uint64_t xo = f64_to_ord(x);
                        rmn[j] = xo < rmn[j] ? xo : rmn[j];
                        rmx[j] = xo > rmx[j] ? xo : rmx[j];
                    }
                }
            }
        }
    }
    if (run_live) flush_run();
    __syncthreads();
    // ---- merge this workgroup's table into the global one
    for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
        uint64_t k = lkeys[s];
        if (k == EMPTY_KEY) continue;
        uint64_t key = (s == cap) ? EMPTY_KEY : k;
        int64_t gslot = global_find_or_insert(g, key, flags);
        if (gslot < 0) continue;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (j >= a.nv) continue;
            uint32_t o = uint32_t(j) * slots + s;
            uint32_t c = lcnt[o];
            global_update(g, gslot, a.v0 + j, uint64_t(c & ~NAN_BIT), lsum[o], a.need_sum[j] != 0, lmn[o], lmx[o],
                          a.need_minmax[j] != 0, (c & NAN_BIT) != 0);
        }
    }
}

using GroupedKernel = void (*)(AggArgs, GroupTable, int *);
template <int PRED, int KEY> GroupedKernel pick_plain(bool plain) {
    return plain ? agg_grouped_kernel<PRED, KEY, true> : agg_grouped_kernel<PRED, KEY, false>;
}
template <int PRED> GroupedKernel pick_key(int key, bool plain) {
    switch (key) {
    case 0: return pick_plain<PRED, 0>(plain);
    case 1: return pick_plain<PRED, 1>(plain);
    default: return pick_plain<PRED, 2>(plain);
    }
}
GroupedKernel pick_grouped_kernel(int pred, int key, bool plain) {
    switch (pred) {
    case 0: return pick_key<0>(key, plain);
    case 1: return pick_key<1>(key, plain);
    case 2: return pick_key<2>(key, plain);
    default: return pick_key<3>(key, plain);
    }
}

// ------------------------------------------------------------------ un-grouped kernel
struct Partial {
    uint64_t cnt;
    double sum;
    double mn, mx;
    uint32_t nan;
    uint32_t pad;
};

__device__ __forceinline__ double shfl_down_f64(double v, int d) { return __shfl_down(v, d, 64); }

__global__ void __launch_bounds__(AGG_BLOCK) agg_ungrouped_kernel(AggArgs a, Partial *partials, int *flags) {
    uint64_t cnt[NV];
    double sum[NV], mn[NV], mx[NV];
    uint32_t nanf[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        cnt[j] = 0; sum[j] = 0.0; mn[j] = DBL_MAX; mx[j] = -DBL_MAX; nanf[j] = 0;
    }
    const int64_t step = int64_t(blockDim.x) * AGG_U;
    for (int64_t base = int64_t(blockIdx.x) * step; base < a.n; base += int64_t(gridDim.x) * step) {
        uint64_t pw[AGG_U], vw[NV][AGG_U];
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * blockDim.x + threadIdx.x;
            bool in = row < a.n;
            pw[u] = (in && a.pred_mode == 1) ? load_word(a.pred_src.values, a.pred_src.dtype, row) : 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                vw[j][u] = (in && j < a.nv && a.val[j].values) ? static_cast<const uint64_t *>(a.val[j].values)[row] : 0;
        }
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * blockDim.x + threadIdx.x;
            bool pass = row < a.n;
            if (a.pred_mode == 1) {
                bool ok = pass && row_valid(a.pred_src, row);
                pass = ok && eval_simple(a.pred, pw[u], ok, flags) != 0;
            } else if (a.pred_mode == 2) {
                pass = pass && get_bit(static_cast<const uint8_t *>(a.pred_src.values), row) && row_valid(a.pred_src, row);
            }
            if (!pass) continue;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (j >= a.nv || !row_valid(a.val[j], row)) continue;
                cnt[j] += 1;
                if (a.need_sum[j] || a.need_minmax[j]) {
                    double x = word_as_f64(vw[j][u], a.val[j].dtype);
                    sum[j] += x;
                    if (x != x) nanf[j] = 1;
                    else {
                        mn[j] = x < mn[j] ? x : mn[j];
                        mx[j] = x > mx[j] ? x : mx[j];
                    }
                }
            }
        }
    }
    __shared__ Partial wave_part[AGG_BLOCK / 64][NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        for (int d = 32; d > 0; d >>= 1) {
            cnt[j] += __shfl_down((unsigned long long)cnt[j], d, 64);
            sum[j] += shfl_down_f64(sum[j], d);
            double omn = shfl_down_f64(mn[j], d), omx = shfl_down_f64(mx[j], d);
            mn[j] = omn < mn[j] ? omn : mn[j];
            mx[j] = omx > mx[j] ? omx : mx[j];
            nanf[j] |= __shfl_down(nanf[j], d, 64);
        }
        if (lane_id() == 0) {
            Partial p{cnt[j], sum[j], mn[j], mx[j], nanf[j], 0};
            wave_part[threadIdx.x / 64][j] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        int j = threadIdx.x;
        Partial t{0, 0.0, DBL_MAX, -DBL_MAX, 0, 0};
        for (int w = 0; w < int(blockDim.x) / 64; ++w) { // fixed order: deterministic
            const Partial &p = wave_part[w][j];
            t.cnt += p.cnt; t.sum += p.sum;
            t.mn = p.mn < t.mn ? p.mn : t.mn;
            t.mx = p.mx > t.mx ? p.mx : t.mx;
            t.nan |= p.nan;
        }
        partials[size_t(blockIdx.x) * NV + j] = t;
    }
}

// un-grouped fast path: plain 8-byte columns, optional integer range predicate on one column.
// PRED: 0 none, 1 predicate column is value column 0 (one load serves both), 2 a separate column.
template <int PRED, int NVT, bool VF64, bool VNULL>
__global__ void __launch_bounds__(AGG_BLOCK) agg_ungrouped_fast_kernel(AggArgs a, FastPred fp, Partial *partials) {
    uint64_t cnt[NVT];
    double sum[NVT], mn[NVT], mx[NVT];
    bool nanf[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        cnt[j] = 0; sum[j] = 0.0; mn[j] = DBL_MAX; mx[j] = -DBL_MAX; nanf[j] = false;
    }
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(a.pred_src.values);
    const uint64_t *__restrict__ valp[NVT];
    const uint64_t *__restrict__ vvalid[NVT]; // VNULL: word-readable validity bitmaps, null = all valid
    const uint64_t *__restrict__ pvalid = reinterpret_cast<const uint64_t *>(PRED != 0 ? a.pred_src.valid : nullptr);
    int vdt[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        valp[j] = static_cast<const uint64_t *>(a.val[j].values);
        vvalid[j] = reinterpret_cast<const uint64_t *>(a.val[j].valid);
        vdt[j] = a.val[j].dtype;
    }
    const int64_t n = a.n, last = a.n - 1;
    struct Tile {
        uint64_t pw[AGG_U], vw[NVT][AGG_U];
        uint64_t vv[VNULL ? NVT : 1][AGG_U], pv[VNULL ? AGG_U : 1];
    };
    auto load_tile = [&](Tile &t, int64_t base) {
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            row = row < last ? row : last;
            if (PRED == 2) t.pw[u] = __builtin_nontemporal_load(&predp[row >> fp.row_shift]);
#pragma unroll
            for (int j = 0; j < NVT; ++j) t.vw[j][u] = __builtin_nontemporal_load(&valp[j][row]);
            if (VNULL) {
#pragma unroll
                for (int j = 0; j < NVT; ++j) t.vv[j][u] = vvalid[j] ? vvalid[j][row >> 6] : ~0ull;
                t.pv[u] = pvalid ? pvalid[row >> 6] : ~0ull;
            }
        }
    };
    auto process_tile = [&](const Tile &t, int64_t base) {
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            bool pass = row < n;
            if (PRED != 0) pass = pass && range_pass(fp, PRED == 1 ? t.vw[0][u] : pred_extract(fp, t.pw[u], row));
            if (VNULL) pass = pass && ((t.pv[u] >> (row & 63)) & 1ull); // a NULL predicate's row is all-NULL: contributes nothing
            if (!pass) continue;
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                double x = VF64 ? u2d(t.vw[j][u]) : word_as_f64(t.vw[j][u], vdt[j]);
                if (VNULL && !((t.vv[j][u] >> (row & 63)) & 1ull)) continue; // NULL value: not counted (Q10)
                cnt[j] += 1;
                sum[j] += x;
                nanf[j] = nanf[j] || (x != x);
                mn[j] = fmin(mn[j], x);
                mx[j] = fmax(mx[j], x);
            }
        }
    };
    const int64_t step = int64_t(AGG_BLOCK) * AGG_U;
    const int64_t stride = int64_t(gridDim.x) * step;
    int64_t base = int64_t(blockIdx.x) * step;
    if (base < n) {
        Tile A, B;
        load_tile(A, base);
        for (;;) {
            load_tile(B, base + stride);
            process_tile(A, base);
            base += stride;
            if (base >= n) break;
            load_tile(A, base + stride);
            process_tile(B, base);
            base += stride;
            if (base >= n) break;
        }
    }
    __shared__ Partial wave_part[AGG_BLOCK / 64][NV];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        uint32_t nf = nanf[j] ? 1u : 0u;
        for (int d = 32; d > 0; d >>= 1) {
            cnt[j] += __shfl_down((unsigned long long)cnt[j], d, 64);
            sum[j] += shfl_down_f64(sum[j], d);
            double omn = shfl_down_f64(mn[j], d), omx = shfl_down_f64(mx[j], d);
            mn[j] = omn < mn[j] ? omn : mn[j];
            mx[j] = omx > mx[j] ? omx : mx[j];
            nf |= __shfl_down(nf, d, 64);
        }
        if (lane_id() == 0) {
            Partial p{cnt[j], sum[j], mn[j], mx[j], nf, 0};
            wave_part[threadIdx.x / 64][j] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x < NVT) {
        int j = threadIdx.x;
        Partial t{0, 0.0, DBL_MAX, -DBL_MAX, 0, 0};
        for (int w = 0; w < AGG_BLOCK / 64; ++w) {
            const Partial &p = wave_part[w][j];
            t.cnt += p.cnt; t.sum += p.sum;
            t.mn = p.mn < t.mn ? p.mn : t.mn;
            t.mx = p.mx > t.mx ? p.mx : t.mx;
            t.nan |= p.nan;
        }
        partials[size_t(blockIdx.x) * NV + j] = t;
    }
}

using UngroupedFastKernel = void (*)(AggArgs, FastPred, Partial *);
template <int PRED, bool VNULL> UngroupedFastKernel pick_ungrouped_fast_nv(int nv, bool vf64) {
    if (nv == 1) return vf64 ? agg_ungrouped_fast_kernel<PRED, 1, true, VNULL> : agg_ungrouped_fast_kernel<PRED, 1, false, VNULL>;
    return vf64 ? agg_ungrouped_fast_kernel<PRED, 2, true, VNULL> : agg_ungrouped_fast_kernel<PRED, 2, false, VNULL>;
}
template <bool VNULL> UngroupedFastKernel pick_ungrouped_fast_pred(int pred, int nv, bool vf64) {
    return pred == 0 ? pick_ungrouped_fast_nv<0, VNULL>(nv, vf64) : pred == 1 ? pick_ungrouped_fast_nv<1, VNULL>(nv, vf64) : pick_ungrouped_fast_nv<2, VNULL>(nv, vf64);
}
UngroupedFastKernel pick_ungrouped_fast(int pred, int nv, bool vf64, bool vnull) {
    return vnull ? pick_ungrouped_fast_pred<true>(pred, nv, vf64) : pick_ungrouped_fast_pred<false>(pred, nv, vf64);
}

__global__ void agg_ungrouped_fold_kernel(const Partial *partials, int nblocks, int nv, int v0, GroupTable g) {
    int j = threadIdx.x;
    if (j >= nv) return;
    Partial t{0, 0.0, DBL_MAX, -DBL_MAX, 0, 0};
    for (int b = 0; b < nblocks; ++b) {
        const Partial &p = partials[size_t(b) * NV + j];
        t.cnt += p.cnt; t.sum += p.sum;
        t.mn = p.mn < t.mn ? p.mn : t.mn;
        t.mx = p.mx > t.mx ? p.mx : t.mx;
        t.nan |= p.nan;
    }
    size_t o = size_t(v0 + j) * (size_t(g.cap) + 1);
    g.cnt[o] += t.cnt;
    g.sum[o] += t.sum;
    uint64_t omn = f64_to_ord(t.mn), omx = f64_to_ord(t.mx);
    if (omn < g.mn[o]) g.mn[o] = omn;
    if (omx > g.mx[o]) g.mx[o] = omx;
    g.nan[o] |= t.nan;
}

// ------------------------------------------------------------------ host side
struct TableBufs {
    BufRef keys, cnt, sum, mn, mx, nan, dense_counter;
    GroupTable g{};
};

TableBufs make_table(nqe_ctx *ctx, uint32_t cap, int V, bool mark_slot0, bool dense = false) {
    TableBufs t;
    size_t slots = size_t(cap) + 1;
    if (V < 1) V = 1;
    t.keys = dev_alloc(ctx, slots * 8);
    t.cnt = dev_alloc(ctx, slots * size_t(V) * 8);
    t.sum = dev_alloc(ctx, slots * size_t(V) * 8);
    t.mn = dev_alloc(ctx, slots * size_t(V) * 8);
    t.mx = dev_alloc(ctx, slots * size_t(V) * 8);
    t.nan = dev_alloc(ctx, slots * size_t(V) * 4);
    t.g.keys = (uint64_t *)t.keys->ptr;
    t.g.cnt = (uint64_t *)t.cnt->ptr;
    t.g.sum = (double *)t.sum->ptr;
    t.g.mn = (uint64_t *)t.mn->ptr;
    t.g.mx = (uint64_t *)t.mx->ptr;
    t.g.nan = (uint32_t *)t.nan->ptr;
    t.g.cap = cap;
    int lg = 0;
    while ((1u << lg) < cap) ++lg;
    t.g.shift = cap == 1 ? 63 : 64 - lg;
    t.g.V = V;
    t.g.dense_count = nullptr;
    if (dense) { // every used slot is written in full by the producer: nothing to initialise but the counter
        t.dense_counter = dev_alloc_zero(ctx, 32); // [0] the counter; words [2..5]: the keys' range (dense_key_range_kernel)
        t.g.dense_count = (uint32_t *)t.dense_counter->ptr;
        return t;
    }
    launch(ctx, "agg_table_init", table_init_kernel, dim3(stream_grid(ctx, int64_t(slots) * V, 256)), dim3(256), 0, t.g,
           mark_slot0 ? 1 : 0);
    return t;
}

ColSrc src_of(const DevColumn &c) {
    ColSrc s;
    s.values = c.values ? c.values->ptr : nullptr;
    s.valid = c.valid();
    s.dtype = c.dtype;
    s.present = 1;
    return s;
}

SimpleExpr plain_column_expr(int dtype) {
    SimpleExpr s;
    std::memset(&s, 0, sizeof(s));
    s.src_dtype = s.out_dtype = dtype;
    for (int k = 0; k < SIMPLE_MAX_OPS; ++k) s.aux[k].pow2_shift = s.aux[k].more = -1;
    return s;
}

struct AggPlan {
    std::vector<int> val_cols;  // distinct value columns
    std::vector<int> vslot;     // aggregate -> value slot
    std::vector<int> need_sum, need_minmax;
};

AggPlan plan_aggs(const nqe_table *in, const nqe_aggregate *aggs, int naggs) {
    if (naggs < 0 || naggs > 16) fail(NQE_ERR_NOT_SUPPORTED, "at most 16 aggregates per call");
    AggPlan p;
    for (int i = 0; i < naggs; ++i) {
        const nqe_aggregate &a = aggs[i];
        if (a.func < NQE_AGG_COUNT || a.func > NQE_AGG_AVG) fail(NQE_ERR_NO_MATCH_FUNCTION, "unknown aggregate function");
        if (a.column < 0 || size_t(a.column) >= in->cols.size()) fail(NQE_ERR_NOT_SUPPORTED, "aggregate column index out of range");
        int dt = in->cols[size_t(a.column)].dtype;
        if (a.func != NQE_AGG_COUNT && !is_word_type(dt)) // sum.rs:93 / :109
            fail(NQE_ERR_NOT_SUPPORTED, "aggregate func for this column type is not supported");
        auto it = std::find(p.val_cols.begin(), p.val_cols.end(), a.column);
        int j;
        if (it == p.val_cols.end()) {
            p.val_cols.push_back(a.column);
            p.need_sum.push_back(0);
            p.need_minmax.push_back(0);
            j = int(p.val_cols.size()) - 1;
        } else {
            j = int(it - p.val_cols.begin());
        }
        p.vslot.push_back(j);
        if (a.func == NQE_AGG_SUM || a.func == NQE_AGG_AVG) p.need_sum[size_t(j)] = 1;
        if (a.func == NQE_AGG_MIN || a.func == NQE_AGG_MAX) p.need_minmax[size_t(j)] = 1;
    }
    return p;
}

// aggregate -> value slot from the aggregate list alone (distinct columns in order of first appearance: what plan_aggs assigns):
// the layout of the partial state, shared by the producer and the merges
std::vector<int> slots_of_aggs(const nqe_aggregate *aggs, int naggs, int *nslots) {
    std::vector<int> cols, vslot;
    for (int i = 0; i < naggs; ++i) {
        auto it = std::find(cols.begin(), cols.end(), aggs[i].column);
        if (it == cols.end()) {
            cols.push_back(aggs[i].column);
            vslot.push_back(int(cols.size()) - 1);
        } else
            vslot.push_back(int(it - cols.begin()));
    }
    *nslots = int(cols.size());
    return vslot;
}

struct AggResult {
    std::unique_ptr<nqe_table> out, keys;
};

// collect → sort by key → finalize
// occupied slots of a hashed table, collected ahead of the flag read-back so that the group count travels with the flags
// (one stream synchronisation per aggregate step instead of two)
struct Collected {
    BufRef keys, slots;
    int64_t G = -1; // -1: not collected yet
    // a densely written table (partitioned path): group count and key range in sort order, read back with the flags
    int64_t dense_G = -1;
    uint64_t ordmin = 0, ordmax = 0;
};

// output columns of an aggregate (or its partial state) with room for `rows` rows, and the kernel arguments that fill them
FinalizeArgs alloc_outputs(nqe_ctx *ctx, AggResult &r, int64_t rows, const nqe_aggregate *aggs, int naggs, const std::vector<int> &vslot,
                           bool partial) {
    r.out = std::make_unique<nqe_table>();
    r.out->ctx = ctx;
    r.out->rows = rows;
    FinalizeArgs f;
    std::memset(&f, 0, sizeof(f));
    f.naggs = naggs;
    f.partial = partial ? 1 : 0;
    for (int i = 0; i < naggs; ++i) {
        f.func[i] = aggs[i].func;
        f.vslot[i] = vslot[size_t(i)];
        f.nslots = std::max(f.nslots, f.vslot[i] + 1);
        if (!partial) {
            r.out->cols.push_back(make_word_column(ctx, aggs[i].func == NQE_AGG_COUNT ? NQE_UINT64 : NQE_FLOAT64, rows, false));
            f.out[i] = (uint64_t *)r.out->cols.back().values->ptr;
        }
    }
    if (partial) {
        const int dts[4] = {NQE_UINT64, NQE_FLOAT64, NQE_FLOAT64, NQE_FLOAT64};
        for (int v = 0; v < f.nslots; ++v)
            for (int k = 0; k < 4; ++k) {
                r.out->cols.push_back(make_word_column(ctx, dts[k], rows, false));
                f.out[4 * v + k] = (uint64_t *)r.out->cols.back().values->ptr;
            }
    }
    return f;
}

// small hashed table: one launch ranks the keys and writes the sorted outputs (rank_finalize_kernel); the columns are allocated
// for a full table and cut to the group count once it has travelled back with the flags (set_group_count)
AggResult emit_ranked(nqe_ctx *ctx, TableBufs &tb, int key_dtype, const nqe_aggregate *aggs, int naggs, const std::vector<int> &vslot,
                      bool partial) {
    const int64_t slots = int64_t(tb.g.cap) + 1;
    AggResult r;
    FinalizeArgs f = alloc_outputs(ctx, r, slots, aggs, naggs, vslot, partial);
    r.keys = std::make_unique<nqe_table>();
    r.keys->ctx = ctx;
    r.keys->rows = slots;
    r.keys->cols.push_back(make_word_column(ctx, key_dtype, slots, false));
    launch(ctx, "agg_rank_finalize", rank_finalize_kernel, dim3(unsigned((slots + RANK_SLOTS - 1) / RANK_SLOTS)), dim3(RANK_WAVES * 64), size_t(slots) * 12, tb.g,
           key_dtype == NQE_INT64 ? 1 : 0, f, (uint64_t *)r.keys->cols[0].values->ptr, (const int *)ctx->d_flags, ctx->h_flags_dev);
    return r;
}

void set_group_count(AggResult &r, int64_t G) {
    r.out->rows = G;
    for (auto &c : r.out->cols) c.length = G;
    r.keys->rows = G;
    for (auto &c : r.keys->cols) c.length = G;
}

AggResult emit(nqe_ctx *ctx, TableBufs &tb, bool grouped, int key_dtype, const nqe_aggregate *aggs, int naggs,
               const std::vector<int> &vslot, bool partial, const Collected *pre = nullptr) {
    int64_t G = 1;
    BufRef sorted_keys, sorted_slots;
    if (grouped) {
        size_t slots = size_t(tb.g.cap) + 1;
        BufRef ck, cs;
        if (pre && pre->G >= 0) {
            G = pre->G;
            ck = pre->keys;
            cs = pre->slots;
        } else if (tb.g.dense_count) { // groups already occupy slots [0, G)
            G = pre && pre->dense_G >= 0 ? pre->dense_G : int64_t(read_scalar(ctx, (const uint32_t *)tb.g.dense_count));
            const bool no_range_tail = getenv("NQE_NO_RANGE_TAIL") != nullptr; // A/B (read per call): the radix-sort tail
            if (pre && pre->dense_G > 4096 && !no_range_tail && pre->ordmax >= pre->ordmin && pre->ordmax - pre->ordmin < uint64_t(8) * uint64_t(G) + 65536) {
                // compact key range: rank by key - min (three launches, no further host wait) instead of sorting
                const uint64_t span = pre->ordmax - pre->ordmin + 1;
                const int items = 4;
                const uint64_t chunks = (span + uint64_t(DR_BLOCK) * items - 1) / (uint64_t(DR_BLOCK) * items);
                const size_t pos_bytes = size_t(chunks) * DR_BLOCK * items * 4, status_bytes = (size_t(chunks) + 1) * 8;
                BufRef work;
                try {
                    work = dev_alloc(ctx, pos_bytes + status_bytes);
                } catch (const Error &e) {
                    if (e.code != NQE_ERR_OUT_OF_MEMORY) throw;
                }
                if (work) {
                    const uint64_t flip = key_dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull;
                    NQE_HIP_CHECK(hipMemsetAsync(work->ptr, 0, pos_bytes + status_bytes, ctx->stream));
                    uint32_t *pos = (uint32_t *)work->ptr;
                    launch(ctx, "agg_dense_rank_mark", dense_rank_mark_kernel, dim3(stream_grid(ctx, G, 256)), dim3(256), 0, (const uint64_t *)tb.g.keys, uint32_t(G), flip,
                           pre->ordmin, pos);
                    AggResult r;
                    FinalizeArgs f = alloc_outputs(ctx, r, G, aggs, naggs, vslot, partial);
                    sorted_keys = dev_alloc(ctx, size_t(G) * 8 + 8);
                    launch(ctx, "agg_dense_rank_emit", dense_rank_emit_kernel<4>, dim3(unsigned(chunks)), dim3(DR_BLOCK), 0, tb.g, (const uint32_t *)pos, span, flip, pre->ordmin,
                           reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(work->ptr) + pos_bytes), (uint64_t *)sorted_keys->ptr, f);
                    r.keys = std::make_unique<nqe_table>();
                    r.keys->ctx = ctx;
                    r.keys->rows = G;
                    DevColumn kc;
                    kc.dtype = key_dtype;
                    kc.length = G;
                    kc.values = sorted_keys;
                    r.keys->cols.push_back(kc);
                    return r;
                }
            }
            ck = tb.keys;
            cs = dev_alloc(ctx, size_t(G) * 4 + 8);
            if (G) launch(ctx, "iota_u32", iota_slots_kernel, dim3(stream_grid(ctx, G, 256)), dim3(256), 0, (uint32_t *)cs->ptr, G);
        } else {
            BufRef counter = dev_alloc_zero(ctx, 4);
            ck = dev_alloc(ctx, slots * 8);
            cs = dev_alloc(ctx, slots * 4);
            launch(ctx, "agg_collect", collect_kernel, dim3(stream_grid(ctx, int64_t(slots), 256)), dim3(256), 0, tb.g, (uint64_t *)ck->ptr,
                   (uint32_t *)cs->ptr, (uint32_t *)counter->ptr);
            G = int64_t(read_scalar(ctx, (const uint32_t *)counter->ptr));
        }
        sorted_keys = dev_alloc(ctx, size_t(G) * 8 + 8);
        sorted_slots = dev_alloc(ctx, size_t(G) * 4 + 8);
        radix_sort_pairs_u64(ctx, (const uint64_t *)ck->ptr, (const uint32_t *)cs->ptr, (uint64_t *)sorted_keys->ptr,
                             (uint32_t *)sorted_slots->ptr, G, key_dtype == NQE_INT64);
    }
    AggResult r;
    FinalizeArgs f = alloc_outputs(ctx, r, G, aggs, naggs, vslot, partial);
    if (G > 0 && naggs > 0)
        launch(ctx, "agg_finalize", finalize_kernel, dim3(stream_grid(ctx, G, 256)), dim3(256), 0, tb.g,
               grouped ? (const uint32_t *)sorted_slots->ptr : (const uint32_t *)nullptr, G, f);
    if (grouped) {
        r.keys = std::make_unique<nqe_table>();
        r.keys->ctx = ctx;
        r.keys->rows = G;
        DevColumn kc;
        kc.dtype = key_dtype;
        kc.length = G;
        kc.values = sorted_keys;
        r.keys->cols.push_back(kc);
    }
    return r;
}

// one workgroup table of the streaming kernel addressed directly (`col % m`, a measured key range), one value column, no validity bitmaps:
// the table carries no key words (aggregate_fast_kernel.hpp: nokeys) — 28 bytes per slot, 5841 slots in 160 KB
constexpr uint64_t DIRECT_WIDE_SLOTS = 5840;
// … and when no aggregate asks for min / max (count / sum / avg: the MM = false instance, 12 bytes per slot): 13633 slots in 160 KB
constexpr uint64_t DIRECT_WIDE_SLOTS_NOMM = 13632;
constexpr uint64_t RANGE_TIER_MAX_SLOTS = 5120;             // slots of one LDS table of the range tier (28 bytes each)
constexpr uint64_t TINY_SALT = 0xC2B2AE3D27D4EB4Full;       // nqe_ctx::agg_key_ranges[hint ^ salt] present: the tiny-groups kernel met a key outside [0, m)
constexpr uint64_t PART_RANGE_SALT = 0x9E3779B97F4A7C15ull; // nqe_ctx::agg_key_ranges[hint ^ salt]: the key range of the query's groups (range partitions)

// ---- PhysicalAggregatePlan::execute (aggregate/mod.rs:113-222) on the device, one object per execution:
//   plan      prepare (key expression, predicate) -> size_tables -> load_hints (what this query shape did last time) -> sample_keys (the
//             first execution's key sample picks the starting tier) -> pick_key_range (a plain key column addressed by key - min);
//   execute   run: per attempt begin_attempt (the group table), then per pass of 1-3 value columns launch_pass = shape_pass (which
//             kernel variant the pass's columns, key and predicate admit) + ONE tier: tier_slab (partitioned, fixed-capacity slabs),
//             tier_exact (partitioned, exact sizes / two levels), tier_streaming (the single-pass LDS-table kernels and the run-time
//             specialised one), the general hashed kernel, or pass_ungrouped;
//   react     finish_attempt: the tail ahead of ONE flag read-back, then react_to_flags — an overflowed tier names the next one, the
//             attempt is redone there and the plan hint remembers it — or the result.
// Switches: the A/B diagnostics are read ONCE per context (nqe_ctx::agg_sw, AggSwitches in nqe_internal.hpp; listed in DESIGN.md §9);
// the ones tests flip between calls are read per call where they are used (NQE_NO_PLAN_HINTS, NQE_NO_KEY_SAMPLE, NQE_NO_RANGE_PARTITION,
// NQE_NO_RANGE_TAIL, NQE_NO_AGG_JIT, NQE_TEST_SLAB_OOM).
AggResult run_aggregate(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *group, int group_nodes,
                        const nqe_aggregate *aggs, int naggs, bool partial);

enum class PassStatus { Done, Abort }; // Abort: leave the pass loop — slab_oom / three_redo / dense_redo / jit_redo says why

struct AggRun {
    // ---- the call
    nqe_ctx *ctx;
    const nqe_table *in;
    const nqe_expr_node *pred;
    int pred_nodes;
    const nqe_expr_node *group;
    int group_nodes;
    const nqe_aggregate *aggs;
    int naggs;
    bool partial;
    const AggSwitches &sw;
    AggResult early; // prepare(): the query was answered by a selection + a predicate-free aggregate
    // ---- plan
    AggPlan plan;
    bool grouped = false, has_pred = false;
    AggArgs a;
    ExprInfo kinfo;
    int key_col = -1;
    bool utf8_key = false;
    DevColumn utf8_codes, utf8_src;
    DevColumn pred_col; // keeps a materialised predicate alive
    bool pred_may_fault = false;
    int conj_col[CONJ_MAX] = {-1, -1, -1, -1};
    TreePred tree;   // pred_mode 4
    BufRef tree_buf; // its program on the device
    bool jit_whole = false, jit_redo = false; // the predicate is evaluated by the run-time specialised streaming kernel (see prepare)
    DevColumn key_colbuf;
    bool kp_valid_words_ok = true, pred_bits_words_ok = true;
    int V = 0;
    uint32_t cap = 1, sized_cap = 1;
    bool partition_mode = false, level2 = false, dense_ok = true, slab_failed = false, key32_failed = false;
    bool range_part_ok = true, range_part_used = false, part_range_sampled = false;
    int64_t part_min = 0;
    uint64_t part_span = 0;
    bool three_on = true;
    bool range_on = false;
    int64_t range_min = 0;
    uint64_t range_span = 0;
    int subsets_log2 = 0, slab_parts_log2 = 8;
    uint64_t hint_key = 0;
    bool any_val_nullable = false, subsets_ok = false, plain_int_key = false, no_hints_env = false, range_sampled = false;
    uint64_t range_limit = 4096, key_flip = 0;
    uint64_t direct_pair_limit = 8192; // values of a key range the direct-mapped tables of TWO key subsets take (AggArgs::direct_sub_width)
    uint64_t direct_one_limit = 4096; // values of a key range ONE direct-mapped workgroup table takes (DIRECT_WIDE_SLOTS where the table has no key words)
    // ---- the attempt
    int attempt = 0;
    bool asked_partition = false, dense = false, flagless = false, slab_oom = false, three = false, three_redo = false, dense_redo = false, first_alone = false;
    int nv_step = 1, pass_nv = 0;
    TableBufs tb;
    bool tiny_ok = true, tiny_used = false; // the register-resident kernel for at most four groups (aggregate_tiny.hip)
    AggResult ranged;        // tier_range: outputs allocated for the whole key range, cut to the group count once it has travelled back with the flags
    uint64_t range_emit_key_min = 0; // ... the key the tail's offsets count from
    BufRef range_total, range_tab, range_status; // ... which agg_range_emit_kernel leaves in range_total; its inputs, kept until the attempt is over
    // ---- the pass (shape_pass)
    bool jit_launched = false, valid_words_ok = true, plain = false, bitmap_pred = false, vnull = false, range_pred = false, chain_pred = false, fast = false, vf64 = true;
    size_t shmem = 0;
    int blocks_per_cu = 1, grid = 1, kk = 2, fast_key = -1, pk = 0, fp = 0;
    AggArgs ka;
    FastPred fpred;

    AggRun(nqe_ctx *c, const nqe_table *t, const nqe_expr_node *p, int pn, const nqe_expr_node *g, int gn, const nqe_aggregate *ag, int na, bool part)
        : ctx(c), in(t), pred(p), pred_nodes(pn), group(g), group_nodes(gn), aggs(ag), naggs(na), partial(part), sw(c->agg_sw) {}

    void materialize_pred() { // the predicate tree as a Boolean column (expression machine), tested bit by bit
        pred_col = evaluate_expr(ctx, in, pred, pred_nodes);
        a.pred_mode = 2;
        a.pred_src = src_of(pred_col);
    }
    std::pair<int64_t, uint64_t> measure_key_range();
    bool prepare();
    void size_tables();
    void load_hints();
    void sample_keys();
    void pick_key_range();
    void begin_attempt();
    PassStatus launch_pass(int v0);
    PassStatus shape_pass();
    PassStatus tier_slab();
    void range_tail(const AggArgs &sa, const SlabArgs &sl, uint32_t rslots);
    void range_emit(int parts_log2, int Q, uint32_t rslots, uint64_t span, int64_t key_min);
    void tier_exact();
    PassStatus tier_streaming(int v0);
    void pass_ungrouped(int v0);
    bool finish_attempt(AggResult *out);
    void keys_to_strings(AggResult &res);
    // values of a measured key range the streaming tier addresses directly: one table's (direct_one_limit), or 2^subsets_log2 tables of range_limit
    uint64_t one_table_or_subsets_limit() const { return subsets_log2 == 1 ? direct_pair_limit : (subsets_log2 ? (range_limit << subsets_log2) : direct_one_limit); }
    bool react_to_flags(const int *f, const Collected &pre);
    AggResult run();
};

// key expression (group_expr[0] only, quirk Q8) and predicate.  true: `early` holds the result (a general key expression under a filter)
bool AggRun::prepare() {
    plan = plan_aggs(in, aggs, naggs);
    grouped = group && group_nodes > 0;
    has_pred = pred && pred_nodes > 0;

    std::memset(&a, 0, sizeof(a));
    a.n = in->rows;
    for (int k = 0; k < SIMPLE_MAX_OPS; ++k) a.pred.aux[k].pow2_shift = a.pred.aux[k].more = a.key.aux[k].pow2_shift = a.key.aux[k].more = -1;

    if (grouped) {
        kinfo = analyze_expr(in, group, group_nodes);
        if (kinfo.out_dtype == NQE_UTF8) {
            // group by a String column (aggregate/mod.rs:170-216): encode strings to representative-row codes,
            // then the Int64 path; keys_out returns the strings of the representatives
            utf8_key = true;
            Utf8Dict dict;
            utf8_src = in->cols[size_t(kinfo.s.col)];
            utf8_codes = utf8_encode_build(ctx, utf8_src, &dict);
            kinfo.out_dtype = NQE_INT64;
            kinfo.simple = true;
        }
        if (kinfo.out_dtype != NQE_INT64 && kinfo.out_dtype != NQE_UINT64) // aggregate/mod.rs:217
            fail(NQE_ERR_NOT_SUPPORTED, "group by only support by `Int64`, `UInt64`, `String`");
        if (!kinfo.simple && has_pred) {
            // A general key expression must only see rows that survive the filter (it may divide by a
            // value the filter excludes): run the selection first, then aggregate without a predicate.
            nqe_table *sel = nullptr;
            nqe_status st = nqe_selection_execute(ctx, in, pred, pred_nodes, &sel);
            if (st != NQE_OK) fail(st, ctx->last_error);
            std::unique_ptr<nqe_table> guard(sel);
            early = run_aggregate(ctx, sel, nullptr, 0, group, group_nodes, aggs, naggs, partial);
            return true;
        }
    }
    // ---- predicate
    if (has_pred) {
        ExprInfo pinfo = analyze_expr(in, pred, pred_nodes);
        pred_may_fault = pinfo.may_fault;
        if (pinfo.out_dtype != NQE_BOOLEAN)
            fail(NQE_ERR_NOT_SUPPORTED, "predicate is not a BooleanArray (selection.rs:61 unwrap panics)");
        if (pinfo.simple) {
            a.pred_mode = 1;
            a.pred = pinfo.s;
            a.pred_src = src_of(in->cols[size_t(pinfo.s.col)]);
        } else if (grouped && match_conj(in, pred, pred_nodes, &a.conj, conj_col)) {
            a.pred_mode = 3; // resolved (or materialised) per pass, see the launch section
        } else if (grouped && !pinfo.may_fault && match_tree_pred(in, pred, pred_nodes, &tree)) {
            a.pred_mode = 4; // likewise
        } else {
            // A tree the static kernels can only take as a materialised Boolean column (column-with-column compares, products of
            // columns, ...): one more pass over its columns plus the bitmap.  When the query has the shape of the lean specialised
            // streaming kernel (expr_jit.hpp: nqe_jit_agg — key `col % m` with 512-4096 table slots, value columns without NULLs)
            // and that kernel is compiled, the predicate is evaluated there, in the aggregation pass: the rest of this function then
            // sees a query without a predicate, and every pass is launched through the specialised kernel (jit_whole).
            const int klast = kinfo.s.nops - 1;
            bool cand = grouped && !utf8_key && kinfo.simple && !kinfo.may_fault && !pinfo.may_fault && klast >= 0 && kinfo.s.op[klast] == NQE_OP_MODULOS && !kinfo.s.lit_left[klast] &&
                        (kinfo.s.op_dtype[klast] == NQE_INT64 || kinfo.s.op_dtype[klast] == NQE_UINT64) && !in->cols[size_t(kinfo.s.col)].validity && !plan.val_cols.empty() &&
                        plan.val_cols.size() <= 2 && !getenv("NQE_NO_AGG_JIT");
            for (int c : plan.val_cols) cand = cand && is_word_type(in->cols[size_t(c)].dtype) && !in->cols[size_t(c)].validity;
            if (cand) {
                for (int c : plan.val_cols) {
                    uint32_t sp;
                    int64_t bi;
                    cand = cand && aggregate_tree_specialised(ctx, in, pred, pred_nodes, group, group_nodes, c, 1, nullptr, &sp, &bi, true);
                }
            }
            if (cand) jit_whole = true; // (a.pred_mode stays 0)
            else materialize_pred();
        }
    }
    if (grouped) {
        a.has_key = 1;
        if (utf8_key) {
            a.key = plain_column_expr(NQE_INT64);
            a.key_src = src_of(utf8_codes);
        } else if (kinfo.simple) {
            a.key = kinfo.s;
            key_col = kinfo.s.col;
            a.key_src = src_of(in->cols[size_t(key_col)]);
        } else {
            key_colbuf = evaluate_expr(ctx, in, group, group_nodes);
            a.key = plain_column_expr(kinfo.out_dtype);
            a.key_src = src_of(key_colbuf);
        }
        a.pred_shares_key = (a.pred_mode == 1 && key_col >= 0 && a.pred.col == key_col) ? 1 : 0;
    }
    // validity bitmaps the fast kernels read as whole 64-bit words: library-owned buffers are padded, a borrowed one only
    // if its length is a multiple of 64
    auto words_ok = [](const DevColumn &c) { return !c.validity || c.validity->owned || (c.length % 64) == 0; };
    kp_valid_words_ok = true;
    if (grouped && key_col >= 0 && !utf8_key) kp_valid_words_ok = kp_valid_words_ok && words_ok(in->cols[size_t(key_col)]);
    if (grouped && utf8_key) kp_valid_words_ok = kp_valid_words_ok && words_ok(utf8_src); // the codes share the strings' validity buffer
    if (a.pred_mode == 1) kp_valid_words_ok = kp_valid_words_ok && words_ok(in->cols[size_t(a.pred.col)]);
    // a Boolean INPUT column used as the predicate is read by the fast kernels as whole 64-bit words of its VALUES bitmap too: a
    // borrowed one ends at ceil(n/8) bytes, so unless n is a multiple of 64 (and the pointer 8-byte aligned) take the general kernel
    pred_bits_words_ok = true;
    if (a.pred_mode == 1 && a.pred.nops == 0 && a.pred_src.dtype == NQE_BOOLEAN) {
        const DevColumn &pc = in->cols[size_t(a.pred.col)];
        if (pc.values && !pc.values->owned && ((pc.length % 64) != 0 || (reinterpret_cast<uintptr_t>(pc.values->ptr) & 7) != 0)) pred_bits_words_ok = false;
    }

    return false;
}

void AggRun::size_tables() {
    V = int(plan.val_cols.size());
    // Global table: the first attempt is SMALL (8192 slots) whatever the input size — every workgroup merges at most one LDS
    // table's worth of groups, and unless the workgroups see different key sets their union fits, so that initialisation is
    // 0.4 MB instead of 92 MB and the whole tail is one launch (rank_finalize_kernel).  When the union does not fit (TABLE_FULL:
    // keys correlated with the tile→workgroup assignment, or a small input of mostly distinct keys) the retry is sized for the
    // worst case (every workgroup inserting its own ≤4096 groups, at most one per row), as is the partitioned path.
    cap = 1, sized_cap = 1;
    if (grouped) {
        int64_t guess = std::min<int64_t>(std::max<int64_t>(in->rows, 1), int64_t(1) << 20);
        sized_cap = 4096;
        while (int64_t(sized_cap) < 2 * guess) sized_cap <<= 1;
        cap = std::min(sized_cap, RANK_MAX_CAP);
    }
    // KEY-RANGE partitions of the slab form (aggregate_common.hpp: SlabArgs::range_span): the range they work from — the exact one an
    // earlier execution's dense tail measured (remembered under a salted hint key), or this execution's key sample (`col % m`: what the
    // modulus allows; a plain column: the sample's range padded by 1/256).  span 0: none; a key outside it sends the attempt back to hashed
    // partitions (a sampled range is then replaced by the measured one, a remembered one by (0, 0): never again).
    // NQE_NO_RANGE_PARTITION=1: hashed partitions only (A/B)
    range_part_ok = getenv("NQE_NO_RANGE_PARTITION") == nullptr; // (read per call: tests switch it)
    three_on = !sw.no_three_column_pass;
    // a plain integer key column whose value RANGE fits a workgroup table (nqe_ctx::agg_key_ranges)
    // Between one LDS table's worth of groups and the partitioned path: the fast kernel with two key subsets (see
    // AggArgs::subsets_log2) — every row is read by two workgroups, each of which keeps its half of the keys.  Rows of the other
    // half cost a wave as many issue slots as its own (lanes are masked, instructions are not skipped), so the kernel time doubles:
    // per 10^8 rows 0.80-0.91 ms at 4096-6000 groups against 1.29 ms partitioned; with four subsets (1.5-1.7 ms) partitioning wins.
    // slab form of the partitioned path: 256 partitions first, PARTS when one of them holds more distinct keys than a workgroup table
    slab_parts_log2 = std::min(std::max(sw.slab_parts_first, 6), PARTS_LOG2);
}

void AggRun::load_hints() {
    // plan hint (see nqe_ctx::agg_hints): FNV-1a over everything that decides which kernels the query takes — the key column's
    // buffer and expression, the predicate and its column, the value columns (buffers, validity, types) and the row count — so
    // that a hint is only ever applied to the very query shape that recorded it
    hint_key = 0;
    if (grouped && a.key_src.values && in->rows >= (int64_t(1) << 18)) {
        hint_key = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t nbytes) {
            const unsigned char *b = static_cast<const unsigned char *>(p);
            for (size_t i = 0; i < nbytes; ++i) hint_key = (hint_key ^ b[i]) * 1099511628211ull;
        };
        const void *kp = a.key_src.values;
        mix(&in->uid, sizeof(in->uid)); // (the table handle's identity: nqe_internal.hpp)
        mix(&kp, sizeof(kp));
        mix(&in->rows, sizeof(in->rows));
        mix(&a.key, sizeof(a.key));
        mix(&a.key_src, sizeof(a.key_src));
        mix(&a.pred_mode, sizeof(a.pred_mode));
        if (a.pred_mode == 3) {
            mix(&a.conj, sizeof(a.conj));
            mix(conj_col, sizeof(conj_col));
        } else if (a.pred_mode == 4) {
            mix(&tree, sizeof(tree));
        } else if (a.pred_mode) {
            mix(&a.pred, sizeof(a.pred));
            mix(&a.pred_src, sizeof(a.pred_src));
        }
        if (jit_whole) // the predicate lives in the specialised kernel only (pred_mode 0): its nodes tell this shape from the unfiltered query
            for (int i = 0; i < pred_nodes; ++i) {
                mix(&pred[i], offsetof(nqe_expr_node, value));
                if (!(pred[i].kind == NQE_EXPR_LITERAL && pred[i].dtype == NQE_UTF8)) mix(&pred[i].value, sizeof(pred[i].value));
            }
        for (int c : plan.val_cols) {
            const DevColumn &dc = in->cols[size_t(c)];
            const void *vp = dc.values ? dc.values->ptr : nullptr, *vv = dc.valid();
            mix(&vp, sizeof(vp));
            mix(&vv, sizeof(vv));
            mix(&dc.dtype, sizeof(dc.dtype));
        }
        if (hint_key == 0) hint_key = 1;
        const bool no_hints = getenv("NQE_NO_PLAN_HINTS") != nullptr; // diagnostics (A/B runs; read per call: tests switch it)
        if (!no_hints) {
            auto pr = ctx->agg_key_ranges.find(hint_key ^ PART_RANGE_SALT);
            if (pr != ctx->agg_key_ranges.end()) {
                if (pr->second.second == 0) range_part_ok = false;
                else {
                    part_min = pr->second.first;
                    part_span = pr->second.second;
                }
            }
        }
        if (!no_hints && ctx->agg_key_ranges.find(hint_key ^ TINY_SALT) != ctx->agg_key_ranges.end()) tiny_ok = false; // a key outside [0, m) was met before
        auto it = ctx->agg_hints.find(hint_key);
        if (!no_hints && it != ctx->agg_hints.end()) {
            const uint8_t hv = it->second & 0x3f;
            if (it->second & 0x40) key32_failed = true; // keys beyond int32: 16-byte tuples
            if (it->second & 0x80) three_on = false;    // more groups than the three-column instance holds
            if (hv == 1 || hv == 16 || hv == 17) { // 1: PARTS partitions, 16: the smaller first count was enough
                partition_mode = true;             // 17: the exact form (a slab overflowed or did not fit)
                if (hv == 1) slab_parts_log2 = PARTS_LOG2;
                if (hv == 17) slab_failed = true;
                cap = std::max(cap, sized_cap);
            } else if (hv >= 2 && hv - 1 <= sw.subsets_max) {
                subsets_log2 = hv - 1;
                cap = std::max(cap, std::min(sized_cap, RANK_MAX_CAP << subsets_log2));
            }
        }
    }
    any_val_nullable = false;
    for (int c : plan.val_cols) any_val_nullable = any_val_nullable || in->cols[size_t(c)].validity != nullptr;
    // the two-subset instances exist for one value column and sources without validity bitmaps
    subsets_ok = V == 1 && !any_val_nullable && !a.key_src.valid && !(a.pred_mode != 0 && a.pred_src.valid);
    range_limit = V <= 1 ? 4096 : 2048; // the smallest workgroup table among the passes
    direct_one_limit = (subsets_ok && getenv("NQE_NO_WIDE_DIRECT") == nullptr) ? DIRECT_WIDE_SLOTS : range_limit; // (subsets_ok: one value column, no validity anywhere; the switch: A/B, read per call)
    // no predicate, no min / max, a plain key column or `col % m`: the 12-byte slots of the MM = false instance (tier_streaming: nomm1 — the
    // same conditions, so that a range the planner turns on is one the pass can address)
    if (direct_one_limit == DIRECT_WIDE_SLOTS && a.pred_mode == 0 && V == 1 && !plan.need_minmax[0] && a.key.nops <= 1) direct_one_limit = DIRECT_WIDE_SLOTS_NOMM;
    direct_pair_limit = 2 * direct_one_limit; // two key subsets: the two halves of the range, each in a table like that
    key_flip = a.key_src.dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull;
    plain_int_key = key_col >= 0 && !utf8_key && a.key.nops == 0 && !a.key_src.valid && (a.key_src.dtype == NQE_INT64 || a.key_src.dtype == NQE_UINT64);
}

// exact min / max of a plain integer key column: one more read of the column (the fallback behind a sampled range)
std::pair<int64_t, uint64_t> AggRun::measure_key_range() {
    BufRef mm = dev_alloc(ctx, 16);
    NQE_HIP_CHECK(hipMemsetAsync(mm->ptr, 0xFF, 8, ctx->stream));
    NQE_HIP_CHECK(hipMemsetAsync(static_cast<char *>(mm->ptr) + 8, 0, 8, ctx->stream));
    launch(ctx, "agg_key_range", key_range_kernel, dim3(stream_grid(ctx, in->rows, 256)), dim3(256), 0, (const uint64_t *)a.key_src.values, in->rows, key_flip,
           (unsigned long long *)mm->ptr);
    uint64_t h[2];
    NQE_HIP_CHECK(hipMemcpyAsync(h, mm->ptr, 16, hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    // (span 0: the whole 64-bit range, or no rows)
    return std::make_pair(int64_t(h[0] ^ key_flip), h[1] >= h[0] ? h[1] - h[0] + 1 : 0ull);
}
void AggRun::sample_keys() {
    // ---- the first execution of a query shape (nothing remembered, or NQE_NO_PLAN_HINTS): a SAMPLE of the keys picks the starting
    // tier instead of falling through abandoned ones — the reference's run_sql is one-shot (db.rs:24-37), so the first execution is
    // the one that counts.  65536 keys (key_sample_kernel, ~20 us): their distinct count is a lower bound of the groups, so a tier it
    // rules out would certainly have overflowed; their min / max stand in for the full pass over a plain key column (the streaming
    // kernel checks every key against the range, and a key outside it asks for the exact measurement).  Only without a predicate: a
    // filter may leave far fewer groups than the table holds.
    no_hints_env = getenv("NQE_NO_PLAN_HINTS") != nullptr; // (both read per call: tests switch them)
    const bool no_sample = getenv("NQE_NO_KEY_SAMPLE") != nullptr;
    const bool simple_mod_key = a.key.nops == 1 && a.key.op[0] == NQE_OP_MODULOS && !a.key.lit_left[0] && (a.key.op_dtype[0] == NQE_INT64 || a.key.op_dtype[0] == NQE_UINT64) &&
                                a.key.aux[0].abs_lit > 1;
    // (the tiers the sample may start in exist for the streaming kernel's shapes only: at least one value column, every one of them
    // 8-byte words — `select k from t group by k` and count() over a Utf8 / Boolean column go through the general kernel, which takes
    // neither key subsets nor a densely written table)
    bool sample_shape = V >= 1;
    for (int c : plan.val_cols) sample_shape = sample_shape && is_word_type(in->cols[size_t(c)].dtype) && in->cols[size_t(c)].values;
    if (grouped && hint_key && !no_sample && sample_shape && in->rows >= KEY_SAMPLE_MIN_ROWS && a.pred_mode == 0 && !jit_whole && a.key_src.values && !a.key_src.valid && !utf8_key &&
        (a.key_src.dtype == NQE_INT64 || a.key_src.dtype == NQE_UINT64) &&
        (a.key.nops == 0 || (simple_mod_key && (key_flip ? 2 * a.key.aux[0].abs_lit - 1 : a.key.aux[0].abs_lit) > direct_one_limit && a.key.aux[0].abs_lit > range_limit)) && // (`col % m`, its keys within a workgroup table: nothing to find out)
        (no_hints_env || (ctx->agg_hints.find(hint_key) == ctx->agg_hints.end() && ctx->agg_key_ranges.find(hint_key) == ctx->agg_key_ranges.end()))) {
        BufRef set = dev_alloc(ctx, (size_t(1) << KEY_SAMPLE_SLOTS_LOG2) * 8), so = dev_alloc(ctx, 32);
        NQE_HIP_CHECK(hipMemsetAsync(set->ptr, 0xFF, (size_t(1) << KEY_SAMPLE_SLOTS_LOG2) * 8, ctx->stream));
        NQE_HIP_CHECK(hipMemsetAsync(so->ptr, 0xFF, 8, ctx->stream));
        NQE_HIP_CHECK(hipMemsetAsync(static_cast<char *>(so->ptr) + 8, 0, 24, ctx->stream));
        launch(ctx, "agg_key_sample", key_sample_kernel, dim3(KEY_SAMPLE / 256), dim3(256), 0, (const uint64_t *)a.key_src.values, in->rows, a.key, key_flip,
               (unsigned long long *)set->ptr, (unsigned long long *)so->ptr);
        uint64_t h[4];
        NQE_HIP_CHECK(hipMemcpyAsync(h, so->ptr, 32, hipMemcpyDeviceToHost, ctx->stream));
        sync(ctx);
        const uint64_t D = h[2] + (h[3] ? 1 : 0);
        // groups of the whole table from the sample's distinct count: D = G (1 - exp(-S / G)) (uniform keys; skew only lowers it)
        double G = double(D);
        if (D > uint64_t(KEY_SAMPLE) / 2) {
            double lo = double(D), hi = 1e13;
            for (int it = 0; it < 60; ++it) {
                const double mid = std::sqrt(lo * hi);
                (mid * (1.0 - std::exp(-double(KEY_SAMPLE) / mid)) < double(D) ? lo : hi) = mid;
            }
            G = lo;
        }
        if (ctx->agg_hints.size() >= 256) ctx->agg_hints.clear();
        // between one and two workgroup tables' worth of groups: two key subsets over a direct-mapped table when the sampled range fits two tables
        // (pick_key_range); otherwise the range tier of the partitioned path when it fits THAT (10^8 rows, 4500-6000 keys spread over 7x their
        // number: 0.89-0.90 ms per execution against 1.05-1.15 for two HASHED subsets, tools/probe_sparse_groups.py); hashed subsets for the rest
        const uint64_t sample_span = h[1] >= h[0] ? h[1] - h[0] + 1 : 0;
        // groups ONE workgroup table takes: every key of a range it addresses directly; three quarters of its slots when it hashes (the kernel's own
        // limit, AggArgs::lds_limit: linear probing beyond that load costs more than the next tier)
        const bool one_direct = plain_int_key && !sw.no_key_range && sample_span != 0 && sample_span <= direct_one_limit;
        const uint64_t one_limit = one_direct ? direct_one_limit : (!sw.lds_load_limit ? range_limit : range_limit * 3 / 4);
        const bool sub_direct = plain_int_key && !sw.no_key_range && V == 1 && sw.direct_subsets && subsets_ok && sw.subsets_max >= 1 && sample_span != 0 &&
                                sample_span <= direct_pair_limit;
        const bool tier_instead = !sub_direct && sw.direct_subsets && range_part_ok && V == 1 && sw.range_tier && plain_int_key && sample_span != 0 &&
                                  sample_span + sample_span / 128 + 32 < uint64_t(256) * RANGE_TIER_MAX_SLOTS;
        if (!sub_direct && (D > 2 * one_limit || (D > one_limit && (!subsets_ok || sw.subsets_max < 1 || tier_instead)))) {
            partition_mode = true; // more distinct keys in the sample than the workgroup tables of the streaming tiers hold
            cap = std::max(cap, sized_cap);
            if (G > 800e3) slab_parts_log2 = PARTS_LOG2; // … and more than 256 partitions of one table each
            ctx->agg_hints[hint_key] = uint8_t(slab_parts_log2 < PARTS_LOG2 ? 16 : 1);
            if (range_part_ok && part_span == 0 && h[1] >= h[0]) { // the range the partitions are cut from (see part_min / part_span)
                uint64_t lo = h[0], hi = h[1]; // sort order: key ^ key_flip
                if (simple_mod_key) {
                    const uint64_t m1 = a.key.aux[0].abs_lit - 1;
                    const bool sgn = key_flip != 0;
                    lo = (sgn && lo < key_flip) ? (uint64_t(0) - m1) ^ key_flip : key_flip; // a negative key in the sample: -(m - 1), else 0
                    hi = m1 ^ key_flip;
                } else {
                    const uint64_t pad = (hi - lo) / 256 + 16;
                    lo = lo > pad ? lo - pad : 0;
                    hi = hi < ~uint64_t(0) - pad ? hi + pad : ~uint64_t(0);
                }
                if (hi - lo < (uint64_t(PARTS) << 12)) {
                    part_min = int64_t(lo ^ key_flip);
                    part_span = hi - lo + 1;
                    part_range_sampled = true;
                }
            }
        } else if (D > one_limit) {
            subsets_log2 = 1;
            cap = std::max(cap, std::min(sized_cap, RANK_MAX_CAP << subsets_log2));
            ctx->agg_hints[hint_key] = uint8_t(2);
            if (sub_direct) { // the sample's range (pick_key_range: a direct-mapped table over the two subsets)
                if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
                ctx->agg_key_ranges[hint_key] = std::make_pair(int64_t(h[0] ^ key_flip), h[1] >= h[0] ? h[1] - h[0] + 1 : 0ull);
                range_sampled = true;
            }
        } else {
            ctx->agg_hints.emplace(hint_key, uint8_t(0)); // sampled: the single-pass tier (a later overflow overwrites this)
            if (plain_int_key && !sw.no_key_range) {
                // the sample's range: the whole column's when it is as narrow as a workgroup table (checked row by row by the kernel)
                if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
                ctx->agg_key_ranges[hint_key] = std::make_pair(int64_t(h[0] ^ key_flip), h[1] >= h[0] ? h[1] - h[0] + 1 : 0ull);
                range_sampled = true;
            }
        }
        if (sw.debug)
            fprintf(stderr, "[nqe] aggregate key sample: distinct %llu of %d -> ~%.3g groups; start partition %d subsets_log2 %d slab_parts_log2 %d\n",
                    (unsigned long long)D, KEY_SAMPLE, G, int(partition_mode), subsets_log2, slab_parts_log2);
    }
}

void AggRun::pick_key_range() {
    // (round 5: also under key subsets — two workgroups per row range, each holding one half of a range of up to 2 x 4096 values in a
    // direct-mapped table (AggArgs::direct_sub_width): one value column)
    const bool sub_range = subsets_log2 == 1 && subsets_ok && V == 1 && sw.direct_subsets;
    if (hint_key && !sw.no_key_range && !partition_mode && (subsets_log2 == 0 || sub_range) && (!no_hints_env || range_sampled) && plain_int_key) {
        // `group by k`, k a plain integer column (dictionary codes, small ids): a value range that fits a workgroup table makes the
        // streaming kernel address it by key - min (no hash, no probe sequence, replicas for a handful of groups).  The range comes
        // from the first execution's key sample, or — tables too small to sample, queries with a predicate — from one pass over the
        // column, and is remembered (nqe_ctx::agg_key_ranges)
        auto rt = ctx->agg_key_ranges.find(hint_key);
        if (rt == ctx->agg_key_ranges.end()) {
            if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
            rt = ctx->agg_key_ranges.emplace(hint_key, measure_key_range()).first;
        }
        if (rt->second.second != 0 && rt->second.second <= one_table_or_subsets_limit()) {
            range_on = true;
            range_min = rt->second.first;
            range_span = rt->second.second;
            if (range_span > range_limit && subsets_log2 == 0) cap = std::max(cap, std::min(sized_cap, RANK_MAX_CAP << 1)); // (room in the group table should the fold go through it)
        }
    }
    if (!subsets_ok) subsets_log2 = 0;
}

void AggRun::begin_attempt() {
    ranged = AggResult();
    range_total = range_tab = range_status = BufRef();
    tiny_used = false;
    range_part_used = false;
    asked_partition = false; // a streaming pass of this attempt ran with allow_partition (see the TABLE_FULL handler below)
    // The partitioned path (entered after the fast kernel asked for it) with a single pass over the value columns writes
    // its groups densely: at most one LDS table's worth per (sub-)partition, never more than the input rows.
    dense = grouped && partition_mode && dense_ok && V <= NV && !any_val_nullable && !a.key_src.valid &&
                       !(a.pred_mode != 0 && a.pred_src.valid);
    uint32_t tcap = cap;
    if (dense) tcap = uint32_t(std::min<int64_t>(int64_t(PARTS) * (level2 ? SUB : 1) * 4097, std::max<int64_t>(in->rows, 1)));
    tb = make_table(ctx, tcap, V, !grouped, dense);
    // an un-grouped aggregate whose passes all took the fast kernel (no flag argument) under a predicate that cannot fault
    flagless = !grouped && !pred_may_fault;
    // nullable sources: one value column per pass — the two-column VNULL variants of the fast kernel spill 60-135 VGPRs
    // (3.1-3.4 TB/s); two passes of the one-column variant (5.3 TB/s each over key + one value column) are faster
    nv_step = (jit_whole || any_val_nullable || a.key_src.valid || (a.pred_mode != 0 && a.pred_src.valid)) ? 1 : NV;
    slab_oom = false; // the slab allocation failed: redo the attempt in the exact form
    // three value columns, none asking for min / max, over a key and predicate the streaming kernel computes itself (C1's
    // `count(id), sum(age), avg(score) … group by id % 3`): ONE pass of the three-column instance (2048-slot workgroup table without min / max arrays)
    // instead of two passes — 32 B/row read once instead of 16 + 24.  A pass that turns out not to fit (another kernel
    // variant, more groups than that table holds) clears `three_on` and the attempt is redone in passes of one and two.
    three = three_on && V == 3 && nv_step == NV && grouped && !partition_mode && subsets_log2 == 0 && (a.pred_mode == 0 || a.pred_mode == 1);
    // (min / max of the LAST column only: its instance carries one pair of min / max arrays — count(id), sum(age), …, max(score), min(score))
    for (int j = 0; three && j < V; ++j) three = !plan.need_minmax[size_t(j)] || j == V - 1;
    three_redo = dense_redo = false;
    // an odd number of value columns in passes of two leaves one pass with a single column: let it be the FIRST column when that
    // one is the key column itself — its pass then reads 8 B/row through the single-load instance instead of 16
    first_alone = !three && nv_step == 2 && V >= 3 && (V % 2) == 1 && grouped && key_col >= 0 && !utf8_key && plan.val_cols[0] == key_col &&
                             is_word_type(in->cols[size_t(key_col)].dtype);
    pass_nv = 0;
}

// which kernel variant the pass's key, predicate and value columns admit (the members the tier functions read)
PassStatus AggRun::shape_pass() {
    a.lds_cap = 2048; // 1 value column: 72 KB (2 WG/CU); 2: 128 KB (1 WG/CU); 3 (no min / max arrays): 136 KB
    int lg = 0;
    while ((1 << lg) < a.lds_cap) ++lg;
    a.lds_shift = 64 - lg;
    size_t slots = size_t(a.lds_cap) + 1;
    shmem = slots * 8 + size_t(std::max(a.nv, 1)) * slots * (a.nv == NVMAX ? 8 + 4 : 8 + 8 + 8 + 4);
    if (a.nv == NVMAX && a.need_minmax[NVMAX - 1]) shmem += slots * 16; // the three-column instance with min / max on its last column
    shmem = (shmem + 15) / 16 * 16;
    blocks_per_cu = shmem <= 80 * 1024 ? 2 : 1;
    grid = int(std::min<int64_t>(int64_t(ctx->num_cus) * blocks_per_cu,
                                     (in->rows + int64_t(AGG_BLOCK) * AGG_U - 1) / (int64_t(AGG_BLOCK) * AGG_U)));
    kk = 2, fast_key = -1; // kk: general-kernel key kind; fast_key: fast-kernel key kind (-1 = not covered)
    if (a.key.nops == 0) kk = 0, fast_key = 0;
    else if (a.key.nops == 1 && a.key.op[0] == NQE_OP_MODULOS && !a.key.lit_left[0] &&
             (a.key.op_dtype[0] == NQE_INT64 || a.key.op_dtype[0] == NQE_UINT64)) {
        if (a.key.aux[0].pow2_shift >= 0) kk = 1, fast_key = 1;
        else if (a.key.aux[0].more >= 0) fast_key = 2; // `col % d`, d not a power of two: magic multiply
    }
    if (fast_key < 0 && a.key.nops >= 1) {
        // any other chain of integer arithmetic with literals that cannot fault (divisors: literals other than 0
        // and -1): the fast kernels evaluate it with the generic interpreter (KEY = 3)
        bool ok = true;
        for (int k = 0; k < a.key.nops; ++k) {
            const int op = a.key.op[k];
            ok = ok && op >= NQE_OP_PLUS && op <= NQE_OP_MODULOS && (a.key.op_dtype[k] == NQE_INT64 || a.key.op_dtype[k] == NQE_UINT64);
            if (op == NQE_OP_DIVIDE || op == NQE_OP_MODULOS)
                ok = ok && !a.key.lit_left[k] && a.key.lit[k] != 0 && a.key.lit[k] != ~0ull;
        }
        if (ok) fast_key = 3;
    }
    // ---- `A and B [and …]` / `A or B [or …]` of up to four range tests (pred_mode 3): inside the single-pass streaming kernel when everything it
    // reads is a plain 8-byte column; everywhere else (more groups than one workgroup table, validity bitmaps, a key the
    // kernel does not compute) the predicate is materialised as a Boolean column first, as any other tree is
    if (a.pred_mode == 1 && a.nv > 1 && a.pred.nops > 1) {
        // a chain with Float64 steps is interpreted by the one-value-column instances only
        bool f64_steps = false;
        for (int k = 0; k < a.pred.nops; ++k) f64_steps = f64_steps || a.pred.op_dtype[k] == NQE_FLOAT64;
        if (f64_steps) materialize_pred();
    }
    if (a.pred_mode == 3) {
        bool ok = !partition_mode && subsets_log2 == 0 && fast_key >= 0 && a.nv >= 1 && is_word_type(a.key_src.dtype) && !a.key_src.valid;
        // (the general form — tests with an arithmetic step, nested and/or — runs in the PRED = 5 instances: one value
        // column per pass, built-in key shapes)
        if (a.conj.general) ok = ok && a.nv == 1 && fast_key != 3;
        for (int j = 0; j < a.nv; ++j) ok = ok && a.val[j].values && !a.val[j].valid;
        const void *other = nullptr; // the one column the kernel would load for the predicate alone
        if (ok) {
            a.conj.need_pw = 0;
            for (int t = 0; t < a.conj.n; ++t) {
                const DevColumn &lc = in->cols[size_t(conj_col[t])];
                const void *lp = lc.values->ptr;
                if (lp == a.key_src.values) a.conj.t[t].src = 0;
                else if (lp == a.val[0].values) a.conj.t[t].src = 1;
                else {
                    if (other && other != lp) ok = false; // two such columns: not this kernel's shape
                    other = lp;
                    a.conj.t[t].src = 2;
                    a.conj.need_pw = 1;
                    a.pred_src = src_of(lc);
                }
            }
        }
        if (!ok) materialize_pred();
    }
    // ---- any other fault-free tree over the same columns (pred_mode 4): the stack machine inside the streaming kernel
    if (a.pred_mode == 4) {
        // (instances: one value column per pass, built-in key shapes — the stack machine's registers)
        bool ok = !partition_mode && subsets_log2 == 0 && fast_key >= 0 && fast_key != 3 && a.nv == 1 && is_word_type(a.key_src.dtype) && !a.key_src.valid;
        for (int j = 0; j < a.nv; ++j) ok = ok && a.val[j].values && !a.val[j].valid;
        TreePred tp = tree;
        const void *other = nullptr;
        int slot_of[TREE_MAX_COLS] = {0, 0, 0};
        a.tree_need_pw = 0;
        for (int c = 0; ok && c < tree.ncols; ++c) {
            const DevColumn &lc = in->cols[size_t(tree.col[c])];
            const void *lp = lc.values->ptr;
            if (lp == a.key_src.values) slot_of[c] = 0;
            else if (lp == a.val[0].values) slot_of[c] = 1;
            else {
                if (other && other != lp) ok = false; // two such columns: not this kernel's shape
                other = lp;
                slot_of[c] = 2;
                a.tree_need_pw = 1;
                a.pred_src = src_of(lc);
            }
        }
        if (ok) {
            for (int i = 0; i < tp.n; ++i) {
                if (tp.ins[i].a_src >= TS_W0) tp.ins[i].a_src = TS_W0 + slot_of[tp.ins[i].a_src - TS_W0];
                if (tp.ins[i].b_src >= TS_W0) tp.ins[i].b_src = TS_W0 + slot_of[tp.ins[i].b_src - TS_W0];
            }
            tree_buf = dev_alloc(ctx, sizeof(TreeInstr) * TREE_MAX_INSTR);
            launch(ctx, "agg_store_tree", store_tree_kernel, dim3(1), dim3(64), 0, tp, (TreeInstr *)tree_buf->ptr);
            a.tree_prog = reinterpret_cast<uint64_t>(tree_buf->ptr);
            a.tree_n = tp.n;
        } else
            materialize_pred();
    }
    // ---- kernel variant (see the template comment)
    pk = 0;
    ka = a;
    if (a.pred_mode == 2) pk = 2;
    else if (a.pred_mode == 3) pk = a.conj.general ? 5 : 4;
    else if (a.pred_mode == 4) pk = 6;
    else if (a.pred_mode == 1) {
        const SimpleExpr &pe = a.pred;
        pk = 3;
        if (pe.nops == 1 && pe.op[0] <= NQE_OP_GT_EQ && is_word_type(pe.src_dtype)) {
            pk = 1;
            if (pe.lit_left[0]) { // lit op x  ≡  x op' lit
                static const int flip[6] = {NQE_OP_EQ, NQE_OP_NOT_EQ, NQE_OP_GT, NQE_OP_GT_EQ, NQE_OP_LT, NQE_OP_LT_EQ};
                ka.pred.op[0] = flip[pe.op[0]];
                ka.pred.lit_left[0] = 0;
            }
        }
    }
    if (pk == 5) ka.tree_need_pw = a.conj.need_pw; // (the interpreted-predicate instances load the third word on this flag)
    plain = is_word_type(a.key_src.dtype);
    // a Boolean predicate column without nulls (a Boolean input column, or any predicate tree evaluated by the
    // expression machine) is tested by the same variants as a separate integer predicate column: the word of a
    // row is its bit
    // NULL keys are dropped and a NULL predicate filters the row out (the NULL row a selection would emit has a NULL
    // key / NULL values, Q4 + Q8): the VNULL variants AND both validity bits into the row's pass flag
    const bool kp_nullable = a.key_src.valid != nullptr || (a.pred_mode != 0 && a.pred_src.valid != nullptr);
    bitmap_pred = a.pred_src.dtype == NQE_BOOLEAN && (a.pred_mode == 2 || (a.pred_mode == 1 && a.pred.nops == 0));
    if (a.pred_mode == 1 && !bitmap_pred) plain = plain && is_word_type(a.pred_src.dtype);
    // value columns may carry validity bitmaps (VNULL variants of the fast kernel); the partitioned path and
    // everything else nullable stays with the general kernel
    vnull = kp_nullable;
    for (int j = 0; j < a.nv; ++j) {
        plain = plain && a.val[j].values;
        vnull = vnull || a.val[j].valid != nullptr;
    }
    if (vnull && (partition_mode || !valid_words_ok || !kp_valid_words_ok)) plain = false;
    if (bitmap_pred && !pred_bits_words_ok) plain = false;
    fpred = FastPred{};
    if (bitmap_pred) fpred = bitmap_fast_pred();
    range_pred = pk == 1 && make_fast_pred(a.pred, &fpred);
    // any other fault-free integer chain `col op lit [op lit]` ending in a comparison: interpreted inside the fast kernel
    chain_pred = false;
    if (a.pred_mode == 1 && !bitmap_pred && !range_pred && a.pred.nops >= 1 &&
        a.pred.op[a.pred.nops - 1] <= NQE_OP_GT_EQ) {
        chain_pred = true;
        for (int k = 0; k < a.pred.nops; ++k) {
            const int op = a.pred.op[k], dt = a.pred.op_dtype[k];
            chain_pred = chain_pred && op <= NQE_OP_MODULOS && (dt == NQE_INT64 || dt == NQE_UINT64 || (dt == NQE_FLOAT64 && op != NQE_OP_MODULOS));
            if (op <= NQE_OP_GT_EQ && k != a.pred.nops - 1) chain_pred = false; // a comparison feeds nothing but the result
            if (op == NQE_OP_DIVIDE || op == NQE_OP_MODULOS) {
                if (dt == NQE_FLOAT64) { // x / lit, lit != +-0 (a zero divisor is arrow's DivideByZero)
                    double dl;
                    std::memcpy(&dl, &a.pred.lit[k], 8);
                    chain_pred = chain_pred && !a.pred.lit_left[k] && dl != 0.0;
                } else
                    chain_pred = chain_pred && !a.pred.lit_left[k] && a.pred.lit[k] != 0 && a.pred.lit[k] != ~0ull;
            }
        }
    }
    fast = plain && a.nv >= 1 && fast_key >= 0 && (pk == 0 || pk >= 4 || bitmap_pred || range_pred || chain_pred);
    if (pk >= 4 && (!fast || vnull)) fail(NQE_ERR_NOT_SUPPORTED, "internal: a tree predicate reached a kernel that cannot evaluate it");
    if (a.nv == NVMAX) {
        // the three-column instances: no predicate or a range test on the key column, the built-in key shapes, no validity
        const bool key_range = pk == 1 && range_pred && a.pred_shares_key && !bitmap_pred && !fpred.fmask;
        if (!(fast && (pk == 0 || key_range) && fast_key != 3 && !vnull && !partition_mode)) {
            three_on = false;
            three_redo = true;
            return PassStatus::Abort;
        }
    }
    return PassStatus::Done;
}

// ---- partitioned path, slab form
PassStatus AggRun::tier_slab() {
    // ---- partitioned path, slab form: ONE pass scatters (key, values) tuples into per-workgroup slabs of fixed
    // capacity (no count pass, no scan, no read-back), then one workgroup per partition aggregates its slabs
    // one value column, integer keys: 12-byte tuples {value, int32 key} unless a key was seen not to fit
    // key-range partitions (aggregate_common.hpp: SlabArgs::range_span): 256 tables (512 beyond 2^20 values) of ceil(span / parts) <= 4096 slots;
    // their tuples hold key - range_min, which fits 32 bits whatever the keys' magnitude
    const bool range_part = a.nv == 1 && dense && range_part_ok && part_span != 0 && part_span <= (uint64_t(PARTS) << 12);
    const bool k32 = a.nv == 1 && (!key32_failed || range_part);
    const int rpt = k32 ? slab_scatter_soa_rows_per_thread() : slab_scatter_rows_per_thread(fp, fast_key, a.nv);
    // the two-stream form runs 512-thread workgroups, two per CU: their barrier phases overlap
    const int sc_threads = k32 ? sw.soa_threads : AGG_BLOCK, sc_per_cu = k32 ? 1024 / sw.soa_threads : slab_scatter_wg_per_cu();
    const int64_t tile_rows = int64_t(sc_threads) * rpt;
    int W = int(std::min<int64_t>(int64_t(ctx->num_cus) * sc_per_cu, (in->rows + tile_rows - 1) / tile_rows));
    int64_t chunk = ((in->rows + W - 1) / W + tile_rows - 1) / tile_rows * tile_rows;
    W = int((in->rows + chunk - 1) / chunk);
    int sparts_log2 = slab_parts_log2;
    // (a table of the range tier may hold up to 5120 slots — 28 bytes each, 140 KB of LDS: a SAMPLED range of 2^20 keys is padded by 1/256
    // and would otherwise need 512 partitions on the first execution: 1.58 instead of ~1.0 ms per 10^8 rows)
    const bool tier_wanted = range_part && V == 1 && sw.range_tier;
    if (range_part) sparts_log2 = part_span <= (uint64_t(256) * (tier_wanted ? RANGE_TIER_MAX_SLOTS : 4096)) ? 8 : PARTS_LOG2;
    // the range tier (aggregate_common.hpp: RangeRec): as many partitions as the RANGE needs at 2^range_slots_log2 slots per table (16 .. 256;
    // 512 beyond 2^20 values), several workgroups per partition in the second kernel, the transposing tail.  One value column (V == 1).
    const bool range_tier = tier_wanted;
    if (range_tier && sparts_log2 == 8) {
        int need = 4;
        while (need < 8 && (uint64_t(1) << (need + sw.range_slots_log2)) < part_span) ++need;
        sparts_log2 = need;
    }
    const int used_parts = 1 << sparts_log2;
    const uint64_t rslots = range_part ? (part_span + (uint64_t(1) << sparts_log2) - 1) >> sparts_log2 : 0;
    range_part_used = range_part;
    if (sw.debug)
        fprintf(stderr, "[nqe] slab partitions (hint %llx, k32 %d dense %d ok %d): range %d min %lld span %llu: %d tables of %llu slots (sampled %d)\n",
                (unsigned long long)hint_key, int(k32), int(dense), int(range_part_ok), int(range_part), (long long)part_min, (unsigned long long)part_span, used_parts,
                (unsigned long long)rslots, int(part_range_sampled));
    const int sparts = 1 << sparts_log2;
    const int64_t mean = chunk / used_parts;
    // an ODD number of 256-byte units per slab: with a power-of-two slab stride (16 KB at 10^8 rows) the 512 write
    // streams of a workgroup — and those of every other workgroup — start on the same HBM channel and move in
    // step (the scatter took 0.78 or 0.97 ms depending on where the buffer happened to land)
    const int64_t capt = ((mean + mean / 4 + 64 + 15) / 16 | 1) * 16;
    const size_t tw = size_t(1 + a.nv);
    const size_t tuple_bytes = k32 ? 12 : tw * 8;
    // the slabs take 1.25x the tuple volume (+ padding) on top of the group table: when that does not fit, the exact
    // form (count → scan → scatter into exactly sized partitions) still may — fall back instead of failing
    BufRef slabs, fill;
    try {
        if (getenv("NQE_TEST_SLAB_OOM")) fail(NQE_ERR_OUT_OF_MEMORY, "slab allocation (NQE_TEST_SLAB_OOM)"); // tests: as if the allocation had failed
        slabs = dev_alloc(ctx, size_t(sparts) * size_t(W) * size_t(capt) * tuple_bytes + 16);
        fill = dev_alloc(ctx, size_t(sparts) * size_t(W) * 4);
    } catch (const Error &e) {
        if (e.code != NQE_ERR_OUT_OF_MEMORY) throw;
        slab_failed = true;
        if (hint_key) ctx->agg_hints[hint_key] = uint8_t(17 | (key32_failed ? 0x40 : 0)); // partitioned, exact form: do not try the slabs again
        flags_reset(ctx);
        slab_oom = true;
        return PassStatus::Abort;
    }
    SlabArgs sl;
    sl.slabs = (uint64_t *)slabs->ptr;
    sl.fill = (uint32_t *)fill->ptr;
    sl.chunk = chunk;
    sl.W = W;
    sl.cap = int32_t(capt);
    sl.parts_log2 = sparts_log2;
    sl.range_min = part_min;
    sl.range_span = range_part ? part_span : 0;
    // (K32: the SoA scatter — stage and carry buffers of 12 bytes per tuple, five counters per partition, the block owner map)
    const size_t sc_carry = size_t(sparts) << (sparts_log2 <= 8 ? 4 : 3);
    const size_t sc_shmem = k32 ? (size_t(tile_rows) + sc_carry) * 12 + size_t(sparts) * 20 + (size_t(tile_rows) / 8 + size_t(sparts)) * 2 + 16 : size_t(tile_rows) * 8 * tw + size_t(PARTS) * 12;
    launch(ctx, "agg_partition_scatter", pick_slab_scatter_kernel(fp, fast_key, a.nv, k32, sc_threads), dim3(W), dim3(sc_threads), sc_shmem, ka, fpred, sl,
           ctx->d_flags);
    AggArgs sa = ka;
    size_t sshmem = shmem;
    int sblocks = blocks_per_cu;
    if (a.nv == 1) { // one value column: a 4096-slot table (147 KB, one workgroup per CU) doubles the distinct keys a partition may hold
        sa.lds_cap = 4096;
        sa.lds_shift = 64 - 12;
        sshmem = ((size_t(4097) * (8 + 28)) + 15) / 16 * 16;
        sblocks = 1;
    }
    if (range_tier) {
        range_tail(sa, sl, uint32_t(rslots));
        // (the slabs go back to the pool when this scope ends: whatever takes them next runs behind these kernels on the same stream)
        return PassStatus::Done;
    }
    if (range_part) { // tables addressed by key - base: 28 bytes per key of the partition's interval
        const size_t dshmem = size_t(28) * size_t(rslots) + 16;
        const int dblocks = int(std::max<size_t>(1, std::min<size_t>(4, (size_t(144) << 10) / dshmem)));
        launch(ctx, "agg_segments_direct", pick_slab_segments_direct_kernel(vf64), dim3(std::min(used_parts, ctx->num_cus * dblocks)), dim3(AGG_BLOCK), dshmem, sa,
               sl, tb.g, ctx->d_flags);
    } else
        launch(ctx, "agg_segments", pick_slab_segments_kernel(a.nv, vf64, k32), dim3(std::min(sparts, ctx->num_cus * sblocks)), dim3(AGG_BLOCK), sshmem,
               sa, sl, tb.g, ctx->d_flags);
    sync(ctx); // the slabs are released at the end of this scope
    return PassStatus::Done;
}

// the range tier behind its scatter: Q workgroups per partition aggregate the slabs into whole tables, the transposing tail ranks and
// writes the groups.  Nothing here waits for the device: the group count travels back with the flags (finish_attempt).
void AggRun::range_tail(const AggArgs &sa, const SlabArgs &sl, uint32_t rslots) {
    const int parts = 1 << sl.parts_log2;
    const size_t dshmem = size_t(28) * size_t(rslots) + 16;
    const int per_cu = dshmem <= (size_t(72) << 10) ? 2 : 1; // 1024-thread workgroups: two per CU when their tables fit side by side
    const int Q = std::max(1, std::min(ctx->num_cus * per_cu / parts, sl.W));
    range_tab = dev_alloc(ctx, size_t(parts) * size_t(Q) * size_t(rslots) * sizeof(RangeRec) + 64);
    const BufRef &tab = range_tab;
    launch(ctx, "agg_segments_direct", pick_range_segments_kernel(vf64), dim3(unsigned(parts * Q)), dim3(AGG_BLOCK), dshmem, sa, sl, Q, (RangeRec *)tab->ptr);
    range_emit(sl.parts_log2, Q, rslots, part_span, part_min);
}

// the tail of the range tier over range_tab = [2^parts_log2][Q][rslots] records: ranks the occupied keys of [key_min, key_min + span) and writes
// keys and aggregates in key order (`ranged`; the group count and the exact key range travel back with the flags)
void AggRun::range_emit(int parts_log2, int Q, uint32_t rslots, uint64_t span, int64_t key_min) {
    const BufRef &tab = range_tab;
    // outputs for the whole range (an upper bound of the groups, and never more than the rows); cut to the group count in finish_attempt
    const int64_t room = int64_t(std::min<uint64_t>(span, uint64_t(std::max<int64_t>(in->rows, 1))));
    ranged = AggResult();
    FinalizeArgs f = alloc_outputs(ctx, ranged, room, aggs, naggs, plan.vslot, partial);
    ranged.keys = std::make_unique<nqe_table>();
    ranged.keys->ctx = ctx;
    ranged.keys->rows = room;
    ranged.keys->cols.push_back(make_word_column(ctx, kinfo.out_dtype, room, false));
    // keys per thread of the tail: 4096-key blocks for wide ranges, 1024-key blocks to keep narrow ones parallel
    const int items = sw.range_emit_items ? sw.range_emit_items : (span >= (uint64_t(1) << 19) ? 4 : 1);
    const uint32_t kb = uint32_t(RE_BLOCK * items), sb = std::max<uint32_t>(1u, kb >> parts_log2), nblocks = (rslots + sb - 1) / sb;
    // block statuses (ticket + one word per block), then three zeroed words for the host: the group count, ~(first key - key_min), last key - key_min
    range_status = dev_alloc(ctx, (size_t(nblocks) + 2 + 4) * 8);
    const BufRef &status = range_status;
    NQE_HIP_CHECK(hipMemsetAsync(status->ptr, 0, (size_t(nblocks) + 2 + 4) * 8, ctx->stream));
    range_total = range_status;
    range_emit_key_min = uint64_t(key_min);
    auto *st = (unsigned long long *)status->ptr;
    auto *tot = st + nblocks + 2;
    auto *keys_out = (uint64_t *)ranged.keys->cols[0].values->ptr;
    const size_t eshmem = size_t(kb) * 32;
    if (items == 4)
        launch(ctx, "agg_range_emit", agg_range_emit_kernel<4>, dim3(nblocks), dim3(RE_BLOCK), eshmem, (const RangeRec *)tab->ptr, parts_log2, Q, rslots, span, uint64_t(key_min), st, keys_out,
               f, tot);
    else
        launch(ctx, "agg_range_emit", agg_range_emit_kernel<1>, dim3(nblocks), dim3(RE_BLOCK), eshmem, (const RangeRec *)tab->ptr, parts_log2, Q, rslots, span, uint64_t(key_min), st, keys_out,
               f, tot);
    NQE_HIP_CHECK(hipMemcpyAsync(ctx->h_flags + NQE_NUM_FLAGS, tot, 24, hipMemcpyDeviceToHost, ctx->stream));
}

void AggRun::tier_exact() {
    // ---- partitioned path, exact form: count → scan → scatter → one workgroup per partition (skewed keys whose
    // partitions overflow a slab, and the two-level form for more distinct keys than PARTS tables hold)
    const int64_t stepr = int64_t(AGG_BLOCK) * 8; // multiple of the count tile (4096) and the scatter tile (8192/4096)
    int nblk = int(std::min<int64_t>(512, (in->rows + stepr - 1) / stepr));
    int64_t chunk = ((in->rows + nblk - 1) / nblk + stepr - 1) / stepr * stepr;
    nblk = int((in->rows + chunk - 1) / chunk);
    const int64_t ncnt = int64_t(PARTS) * nblk;
    BufRef counts = dev_alloc(ctx, size_t(ncnt) * 4), offs = dev_alloc(ctx, size_t(ncnt + 1) * 8);
    PartArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.counts = (uint32_t *)counts->ptr;
    pa.offsets = (const uint64_t *)offs->ptr;
    pa.chunk = chunk;
    launch(ctx, "agg_partition_count", pick_part_kernel(fp, fast_key, a.nv, false), dim3(nblk), dim3(AGG_BLOCK), 0, ka, fpred, pa);
    exclusive_scan_u32_to_u64(ctx, (const uint32_t *)counts->ptr, (uint64_t *)offs->ptr, ncnt);
    const int64_t R = int64_t(read_scalar(ctx, (const uint64_t *)offs->ptr + ncnt));
    if (R > 0) {
        BufRef okey = dev_alloc(ctx, size_t(R) * 8 + 8), ov0 = dev_alloc(ctx, size_t(R) * 8 + 8), ov1;
        if (a.nv > 1) ov1 = dev_alloc(ctx, size_t(R) * 8 + 8);
        pa.out_key = (uint64_t *)okey->ptr;
        pa.out_val[0] = (uint64_t *)ov0->ptr;
        pa.out_val[1] = ov1 ? (uint64_t *)ov1->ptr : nullptr;
        const size_t sc_rows = size_t(AGG_BLOCK) * (a.nv == 1 ? 8 : 4);
        const size_t sc_shmem = sc_rows * 8 * size_t(1 + a.nv) + size_t(PARTS) * (8 + 4 + 4);
        launch(ctx, "agg_partition_scatter", pick_scatter_kernel(fp, fast_key, a.nv), dim3(nblk), dim3(AGG_BLOCK), sc_shmem, ka, fpred,
               pa);
        // one value column: a 4096-slot table (147 KB, one workgroup per CU) doubles the distinct keys a partition may hold
        AggArgs sa = ka;
        size_t sshmem = shmem;
        int sblocks = blocks_per_cu;
        if (a.nv == 1) {
            sa.lds_cap = 4096;
            sa.lds_shift = 64 - 12;
            sshmem = ((size_t(4097) * (8 + 28)) + 15) / 16 * 16;
            sblocks = 1;
        }
        int sgrid = std::min(PARTS, ctx->num_cus * sblocks);
        auto segk = pick_segments_kernel(a.nv, vf64);
        if (!level2) {
            launch(ctx, "agg_segments", segk, dim3(sgrid), dim3(AGG_BLOCK), sshmem, sa, (const uint64_t *)offs->ptr, int64_t(nblk), PARTS,
                   PARTS_LOG2, 1, (const uint64_t *)okey->ptr, (const uint64_t *)ov0->ptr,
                   ov1 ? (const uint64_t *)ov1->ptr : (const uint64_t *)nullptr, tb.g, ctx->d_flags);
        } else {
            // partitions were overfull: split each into 64 sub-partitions, then one workgroup per sub-partition
            BufRef k2 = dev_alloc(ctx, size_t(R) * 8 + 8), v02 = dev_alloc(ctx, size_t(R) * 8 + 8), v12;
            if (a.nv > 1) v12 = dev_alloc(ctx, size_t(R) * 8 + 8);
            BufRef suboff = dev_alloc(ctx, size_t(PARTS) * SUB * 8 + 16);
            auto subk = pick_subpartition_kernel(a.nv);
            launch(ctx, "agg_subpartition", subk, dim3(std::min(PARTS, ctx->num_cus)), dim3(AGG_BLOCK), sc_rows * 8 * size_t(1 + a.nv),
                   (const uint64_t *)offs->ptr, int64_t(nblk), (const uint64_t *)okey->ptr, (const uint64_t *)ov0->ptr,
                   ov1 ? (const uint64_t *)ov1->ptr : (const uint64_t *)nullptr, (uint64_t *)k2->ptr, (uint64_t *)v02->ptr,
                   v12 ? (uint64_t *)v12->ptr : (uint64_t *)nullptr, (uint64_t *)suboff->ptr);
            launch(ctx, "agg_segments", segk, dim3(std::min(PARTS * SUB, ctx->num_cus * sblocks)), dim3(AGG_BLOCK), sshmem, sa,
                   (const uint64_t *)suboff->ptr, int64_t(1), PARTS * SUB, PARTS_LOG2 + SUB_LOG2, 0, (const uint64_t *)k2->ptr,
                   (const uint64_t *)v02->ptr, v12 ? (const uint64_t *)v12->ptr : (const uint64_t *)nullptr, tb.g, ctx->d_flags);
            sync(ctx);
        }
        sync(ctx); // the partition buffers are released at the end of this scope
    }
}

PassStatus AggRun::tier_streaming(int v0) {
    // (only kernels that have a partitioned counterpart may ask for it — the fuzzer found an interpreted predicate
    // asking before the partition kernels had that variant: a densely laid out table went to the hashed general kernel)
    ka.allow_partition = in->rows >= (int64_t(1) << 18) ? 1 : 0;
    asked_partition = asked_partition || ka.allow_partition != 0;
    ka.flag_check_mask = sw.flag_check_mask; // how often a wave looks at the overflow flags (aggregate_common.hpp): every 8th iteration
    // ONE 1024-thread workgroup per CU: fewer concurrent streams read HBM faster (A/B on one box: headline
    // 2.44 -> 2.39 ms, C3 2.56 -> 2.41 ms, random keys 3.63 -> 3.54 ms, 1 % nulls 0.81 -> 0.69 ms per 2e8 rows;
    // tools/stream_bench.hip shows the same for a bare read kernel)
    int fgrid = std::min(grid, ctx->num_cus);
    ka.subsets_log2 = subsets_log2;
    if (subsets_log2) fgrid = std::max(1, ctx->num_cus / (8 << subsets_log2)) * (8 << subsets_log2);
    size_t fshmem = shmem;
    if (a.nv == 1) { // the CU's LDS is this workgroup's alone: a 4096-slot table (147 KB) keeps up to ~3500 groups on this path
        ka.lds_cap = 4096;
        ka.lds_shift = 64 - 12;
        fshmem = ((size_t(4097) * (8 + 28)) + 15) / 16 * 16;
    }
    // (round 6) a directly addressed table without validity bitmaps has no key words: up to DIRECT_WIDE_SLOTS keys in ONE workgroup table
    // nomm1: one value column nobody asks min / max of, through the instance without those arrays (12 instead of 28 bytes per slot)
    const bool nomm_ok = a.nv == 1 && !a.need_minmax[0] && !vnull && (fp == 0 || fp == 1) && fast_key != 3;
    const bool nomm1 = nomm_ok && subsets_log2 == 0;
    const bool pair_nomm = nomm_ok && subsets_log2 == 1 && fast_key == 0; // … and the two halves of a measured range, each in such a table
    auto widen = [&](uint64_t span) {
        if (a.nv != 1 || vnull || subsets_log2 != 0 || span > direct_one_limit || span > (nomm1 ? DIRECT_WIDE_SLOTS_NOMM : DIRECT_WIDE_SLOTS)) return false;
        ka.lds_cap = int32_t((span + 15) & ~uint64_t(15));
        return true;
    };
    const int lastop = a.key.nops - 1;
    if (fast_key == 1 || fast_key == 2 ||
        (fast_key == 3 && a.key.op[lastop] == NQE_OP_MODULOS && !a.key.lit_left[lastop])) {
        // `… % m`: keys lie in (-m, m) (signed) or [0, m) — direct-mapped LDS table when that span fits
        const bool sgn = a.key.op_dtype[lastop] == NQE_INT64;
        const uint64_t m = a.key.aux[lastop].abs_lit;
        const uint64_t span = sgn ? 2 * m - 1 : m;
        if (m > 0 && (span <= uint64_t(ka.lds_cap) || widen(span))) {
            ka.direct = 1;
            ka.direct_bias = sgn ? int64_t(m) - 1 : 0;
            // few groups: replicate the table so that the lanes of a wave do not all update the same words
            // (at most 64 replicas, at most 1024 slots in all: the merge walks them)
            while (ka.direct_rep < 6 && (span << (ka.direct_rep + 1)) <= 1024) ++ka.direct_rep;
        }
    }
    ka.direct_sub_width = 0;
    ka.lds_limit = (ka.allow_partition && sw.lds_load_limit) ? uint32_t(ka.lds_cap) * 3u / 4u : 0u; // (hashed tables only look at it)
    // two subsets over a measured range: its two halves (whole 16-slot units), each a table without key words
    const uint64_t half_span = ((range_span + 1) / 2 + 15) & ~uint64_t(15);
    const bool pair_fits = subsets_log2 == 1 && a.nv == 1 && !vnull && range_span <= direct_pair_limit && half_span <= (pair_nomm ? DIRECT_WIDE_SLOTS_NOMM : DIRECT_WIDE_SLOTS);
    if (range_on && fast_key == 0 && !ka.direct && (subsets_log2 ? pair_fits : (range_span <= uint64_t(ka.lds_cap) || widen(range_span)))) {
        // the key column's measured range fits the table: slot = key - min, every key checked against the range
        ka.direct = 2;
        ka.direct_bias = int64_t(0ull - uint64_t(range_min));
        ka.direct_span = range_span;
        while (ka.direct_rep < 6 && (range_span << (ka.direct_rep + 1)) <= 1024) ++ka.direct_rep;
        // ... or the tables of the two workgroups that share their rows, each holding one half of the range (pick_key_range turns the range on
        // under subsets for one value column and two subsets only)
        if (subsets_log2) {
            ka.lds_cap = int32_t(half_span);
            ka.direct_sub_width = int32_t(half_span);
        }
    }
    const bool nomm_sub = pair_nomm && ka.direct_sub_width != 0;
    const size_t slot_bytes = (nomm1 || nomm_sub) ? 12 : 28;
    if (ka.direct && !vnull) { // no key words in the table (aggregate_fast_kernel.hpp: nokeys); a widened table: its own slot count
        const size_t fslots = size_t(ka.lds_cap) + 1;
        fshmem = a.nv == 1 ? fslots * slot_bytes : fshmem - fslots * 8;
        fshmem = (fshmem + 15) / 16 * 16;
    }
    ka.subset_shift = ka.lds_shift - 3; // the bits below the table's slot bits (subsets_log2 <= 3)
    // the value column is the key column itself (and the predicate, if any, tests it too): the single-load instance
    // (three columns: the instance whose tile leaves the first value column out because it IS the key column)
    const bool share = (a.nv == 1 || a.nv == NVMAX) && a.val_shares_key[0] && a.val[0].values == a.key_src.values && (fp == 0 || fp == 1) && fast_key != 3 &&
                       !vnull && (a.nv == NVMAX || !vf64) && subsets_log2 == 0;
    // no aggregate of the pass asks for min / max: instances without those LDS arrays (two and three columns, and the
    // single-load one — `count(id) … group by id % 3` updates one LDS word per row instead of reading two and updating four)
    bool nomm = a.nv >= 2 || share || nomm1 || nomm_sub;
    for (int j = 0; j < a.nv; ++j) nomm = nomm && !a.need_minmax[j];
    // a predicate tree the static kernel would interpret (PRED 5 / 6) over `col % m` keys and one value column: the lean
    // run-time specialised kernel, once it has been compiled — its workgroup tables are folded into the group table here
    // interpreted chain predicates (PRED 3: `id % 10 < 5`, `id * 3 >= K`) take the specialised kernel too: 0.70 -> 0.555 ms per 2x10^8 rows
    // ---- `col % m`, m <= 4 (the reference's own `group by id % 3`): the register-resident kernel (aggregate_tiny.hip) — no LDS table at all
    {
        const uint64_t tm = a.key.nops == 1 ? a.key.aux[0].abs_lit : 0;
        bool tiny = sw.tiny_groups && tiny_ok && !jit_whole && ka.direct == 1 && (fast_key == 1 || fast_key == 2) && tm >= 1 && tm <= 4 && (fp == 0 || fp == 1) && !vnull &&
                    subsets_log2 == 0 && key_col >= 0 && a.nv >= 1 && a.nv <= 3 && in->rows >= (int64_t(1) << 20);
        for (int j = 0; tiny && j < a.nv; ++j) tiny = a.val[j].values != nullptr && (j == a.nv - 1 || !a.need_minmax[j]);
        // (its tiles are read in 16-byte loads: a column that starts on an odd word — a slice of a borrowed buffer — takes the streaming kernel)
        tiny = tiny && (reinterpret_cast<uintptr_t>(a.key_src.values) & 15u) == 0;
        for (int j = 0; tiny && j < a.nv; ++j) tiny = (reinterpret_cast<uintptr_t>(a.val[j].values) & 15u) == 0;
        if (tiny) {
            const size_t cells = size_t(fgrid) * size_t(tm), col_words = (cells * 28 + 7) / 8;
            BufRef partials = dev_alloc(ctx, col_words * 8 * size_t(a.nv) + 64);
            ka.partials = reinterpret_cast<uint64_t>(partials->ptr);
            ka.partial_span = uint32_t(tm);
            const bool mm = a.need_minmax[a.nv - 1] != 0;
            launch(ctx, "agg_grouped_tiny", pick_tiny_groups_kernel(fp, a.nv, mm, uint32_t(tm), a.val[0].values == a.key_src.values), dim3(fgrid), dim3(AGG_BLOCK), 0, ka, fpred, uint32_t(tm), uint32_t(sw.tiny_unpack_tiles), ctx->d_flags);
            for (int j = 0; j < a.nv; ++j) {
                const double *ps = reinterpret_cast<const double *>(partials->ptr) + size_t(j) * col_words;
                launch(ctx, "agg_fold_partials", agg_fold_partials_kernel, dim3((unsigned(tm) + 15) / 16), dim3(256), 0, ps, ps + cells, ps + 2 * cells, reinterpret_cast<const uint32_t *>(ps + 3 * cells),
                       fgrid, uint32_t(tm), int64_t(0), (mm && j == a.nv - 1) ? 1 : 0, tb.g, a.v0 + j, ctx->d_flags, 0, (RangeRec *)nullptr);
            }
            tiny_used = true;
            return PassStatus::Done;
        }
    }
    const bool jit_chains = !sw.no_agg_jit_chains;
    BufRef jit_partials;
    uint32_t jit_span = 0;
    int64_t jit_bias = 0;
    // … and so do interpreted chain KEYS (`(id + 1) % 1000`, KEY 3) whatever the predicate: the kernel bakes the whole key program
    // agg_jit_all = 1: also `col % m` by magic multiply (KEY 2: the literal modulus baked in — `id % 1000` 0.584 -> 0.548 ms, `id % 2000`
    // 0.643 -> 0.560 per 2x10^8 rows), 2 = every `% m` key (A/B: the headline's power-of-two key 2.53 vs 2.47 ms — within noise, it stays static), 0 = neither
    const int jit_all = sw.agg_jit_all;
    const bool jit_try = jit_whole || (has_pred && (fp >= 5 || (fp == 3 && jit_chains))) || (fast_key == 3 && jit_chains) || (jit_all >= 1 && fast_key == 2) || jit_all >= 2;
    if (jit_try && (fast_key == 1 || fast_key == 2 || fast_key == 3) && ka.direct == 1 && ka.direct_rep == 0 && a.nv == 1 && !vnull && subsets_log2 == 0 &&
        key_col >= 0 && a.val[0].values &&
        aggregate_tree_specialised(ctx, in, has_pred ? pred : nullptr, pred_nodes, group, group_nodes, plan.val_cols[size_t(v0)], fgrid,
                                   &jit_partials, &jit_span, &jit_bias)) {
        const size_t cells = size_t(fgrid) * jit_span;
        const double *ps = (const double *)jit_partials->ptr;
        jit_launched = true;
        launch(ctx, "agg_merge_partials", agg_merge_partials_kernel, dim3((jit_span + 3) / 4), dim3(256), 0, ps, ps + cells, ps + 2 * cells,
               reinterpret_cast<const uint32_t *>(ps + 3 * cells), fgrid, jit_span, jit_bias, tb.g, a.v0, ctx->d_flags);
    } else if (jit_whole) {
        jit_redo = true; // (cannot happen once the dry run said yes — but a static kernel must never run without the predicate)
        return PassStatus::Abort;
    } else {
    FastKernel fk = pick_fast_kernel(fp, fast_key, a.nv, vf64, vnull, subsets_log2 != 0, nomm, share);
    if (!fk) fail(NQE_ERR_NOT_SUPPORTED, "internal: no such variant of the streaming aggregate kernel");
    // a direct-mapped table without validity bitmaps leaves the kernel whole (AggArgs::partials) and is folded by agg_fold_partials_kernel
    BufRef direct_partials;
    uint32_t pspan = 0;
    size_t pcol_words = 0;
    if (ka.direct && !vnull && (subsets_log2 == 0 || ka.direct_sub_width) && sw.direct_partials) {
        pspan = ka.direct_sub_width ? uint32_t(ka.lds_cap) : uint32_t(ka.direct == 2 ? range_span : (a.key.op_dtype[a.key.nops - 1] == NQE_INT64 ? 2 * a.key.aux[a.key.nops - 1].abs_lit - 1 : a.key.aux[a.key.nops - 1].abs_lit));
        const size_t cells = size_t(fgrid) * pspan;
        pcol_words = (cells * 28 + 7) / 8;
        // ... when the atomics would matter: workgroups x groups x 4 of them at ~2.4 x 10^10 / s against the rows' streaming time.  Measured
        // (10^8 rows, random keys): 4096 groups 0.467 -> 0.266 ms, 1024 groups 0.290 -> 0.262; the headline (10^9 rows, 1024 groups in runs)
        // 2.50 -> 2.56 — there the fold's launch and the table stores cost more than a million atomics spread over the pass
        const bool worth = uint64_t(cells) * 512 > uint64_t(in->rows);
        try {
            if (worth) direct_partials = dev_alloc(ctx, pcol_words * 8 * size_t(a.nv) + 64);
        } catch (const Error &e) {
            if (e.code != NQE_ERR_OUT_OF_MEMORY) throw; // (no room: the atomics)
        }
        if (direct_partials) {
            // counts zeroed: an attempt the kernel abandons (NEED_PARTITION) writes no table, and the fold must then find nothing
            for (int j = 0; j < a.nv; ++j)
                NQE_HIP_CHECK(hipMemsetAsync(reinterpret_cast<double *>(direct_partials->ptr) + size_t(j) * pcol_words + 3 * cells, 0, cells * 4, ctx->stream));
            ka.partials = reinterpret_cast<uint64_t>(direct_partials->ptr);
            ka.partial_span = pspan;
        }
    }
    launch(ctx, "agg_grouped_fast", fk, dim3(fgrid), dim3(AGG_BLOCK), fshmem, ka, fpred, tb.g, ctx->d_flags);
    if (direct_partials) {
        const size_t cells = size_t(fgrid) * pspan;
        for (int j = 0; j < a.nv; ++j) {
            const double *ps = reinterpret_cast<const double *>(direct_partials->ptr) + size_t(j) * pcol_words;
            const bool mmj = a.need_minmax[j] != 0 && !(nomm);
            const int fsub = ka.direct_sub_width ? subsets_log2 : 0;
            // the subsets' tables cover the key range in order: folded into ONE table of records the range tier's tail ranks and writes (no global
            // hash table of 16384 slots, no collect / sort / finalize behind a host round trip: ~0.12 ms of a 10^8-row step)
            // (round 6: so does ONE table of more than 4096 keys — its groups would crowd the first attempt's 8192-slot group table)
            const bool to_tail = (fsub || pspan > 4096) && V == 1 && sw.range_tier;
            if (to_tail) range_tab = dev_alloc(ctx, (size_t(pspan) << fsub) * sizeof(RangeRec) + 64);
            launch(ctx, "agg_fold_partials", agg_fold_partials_kernel, dim3(((pspan << fsub) + 15) / 16), dim3(256), 0, ps, ps + cells, ps + 2 * cells,
                   reinterpret_cast<const uint32_t *>(ps + 3 * cells), fgrid, pspan, ka.direct_bias, mmj ? 1 : 0, tb.g, a.v0 + j, ctx->d_flags, fsub,
                   to_tail ? (RangeRec *)range_tab->ptr : (RangeRec *)nullptr);
            if (to_tail) range_emit(0, 1, pspan << fsub, fsub ? range_span : uint64_t(pspan), fsub ? range_min : -ka.direct_bias); // (one table: slot s holds the key s - direct_bias)
        }
    }
    }
    return PassStatus::Done;
}

void AggRun::pass_ungrouped(int v0) {
    const int grid = int(std::min<int64_t>(int64_t(ctx->num_cus), // one 1024-thread workgroup per CU (A/B: 0.82 -> 0.57 ms per 2e8 rows)
                                     (in->rows + int64_t(AGG_BLOCK) * AGG_U - 1) / (int64_t(AGG_BLOCK) * AGG_U)));
    BufRef partials = dev_alloc(ctx, size_t(grid) * NV * sizeof(Partial));
    // fast path: plain 8-byte value columns, predicate none or an integer `col cmp lit`
    FastPred ufp{};
    bool uplain = a.nv >= 1;
    bool uvnull = a.pred_mode != 0 && a.pred_src.valid != nullptr;
    for (int j = 0; j < a.nv; ++j) {
        uplain = uplain && a.val[j].values;
        uvnull = uvnull || a.val[j].valid != nullptr;
    }
    if (uvnull && (!valid_words_ok || !kp_valid_words_ok)) uplain = false; // bitmaps must be readable as whole words
    if (a.pred_src.dtype == NQE_BOOLEAN && a.pred_mode == 1 && a.pred.nops == 0 && !pred_bits_words_ok) uplain = false;
    const bool ubitmap = a.pred_src.dtype == NQE_BOOLEAN && (a.pred_mode == 2 || (a.pred_mode == 1 && a.pred.nops == 0));
    if (ubitmap) ufp = bitmap_fast_pred();
    bool upred_ok = a.pred_mode == 0 || ubitmap ||
                    (a.pred_mode == 1 && is_word_type(a.pred_src.dtype) && make_fast_pred(a.pred, &ufp));
    if (uplain && upred_ok) {
        int up = a.pred_mode == 0 ? 0 : ((a.pred_src.values == a.val[0].values && !ubitmap && !ufp.fmask) ? 1 : 2);
        bool vf64 = true;
        for (int j = 0; j < a.nv; ++j) vf64 = vf64 && a.val[j].dtype == NQE_FLOAT64;
        launch(ctx, "agg_ungrouped_fast", pick_ungrouped_fast(up, a.nv, vf64, uvnull), dim3(grid), dim3(AGG_BLOCK), 0, a, ufp,
               (Partial *)partials->ptr);
    } else {
        flagless = false;
        launch(ctx, "agg_ungrouped", agg_ungrouped_kernel, dim3(grid), dim3(AGG_BLOCK), 0, a, (Partial *)partials->ptr,
               ctx->d_flags);
    }
    launch(ctx, "agg_ungrouped_fold", agg_ungrouped_fold_kernel, dim3(1), dim3(64), 0,
           (const Partial *)partials->ptr, grid, a.nv, v0, tb.g);
}

PassStatus AggRun::launch_pass(int v0) {
    jit_launched = false;
    pass_nv = three ? NVMAX : (first_alone && v0 == 0) ? 1 : nv_step;
    a.nv = std::min(pass_nv, V - v0);
    if (a.nv < 0) a.nv = 0;
    a.v0 = v0;
    valid_words_ok = true; // validity bitmaps readable as whole 64-bit words (owned buffers are padded)
    for (int j = 0; j < NVMAX; ++j) {
        std::memset(&a.val[j], 0, sizeof(ColSrc));
        a.val_shares_key[j] = a.need_sum[j] = a.need_minmax[j] = 0;
        if (j < a.nv) {
            int c = plan.val_cols[size_t(v0 + j)];
            const DevColumn &dc = in->cols[size_t(c)];
            if (dc.validity && !dc.validity->owned && (dc.length % 64) != 0) valid_words_ok = false;
            a.val[j] = src_of(dc);
            if (!is_word_type(dc.dtype)) a.val[j].values = nullptr; // count-only over Boolean/Utf8: validity only
            a.need_sum[j] = plan.need_sum[size_t(v0 + j)];
            a.need_minmax[j] = plan.need_minmax[size_t(v0 + j)];
            a.val_shares_key[j] = (grouped && c == key_col && is_word_type(dc.dtype)) ? 1 : 0;
        }
    }
    if (in->rows == 0) return PassStatus::Done;
    if (!grouped) {
        pass_ungrouped(v0);
    } else {
        if (shape_pass() == PassStatus::Abort) return PassStatus::Abort;
        if (fast) {
            // variant 1 tests the key word with the integer range test alone; Float64 predicates and bitmaps use the
            // "other column" variant, whose extraction step applies the order mapping
            fp = pk == 0 ? 0 : pk >= 4 ? pk : (chain_pred ? 3 : ((a.pred_shares_key && !bitmap_pred && !fpred.fmask) ? 1 : 2));
            vf64 = true;
            for (int j = 0; j < a.nv; ++j) vf64 = vf64 && a.val[j].dtype == NQE_FLOAT64;
            if (partition_mode && !level2 && !slab_failed) {
                if (tier_slab() == PassStatus::Abort) return PassStatus::Abort;
            } else if (partition_mode) {
                tier_exact();
            } else if (tier_streaming(v0) == PassStatus::Abort)
                return PassStatus::Abort;
        } else {
            // the general kernel's PLAIN variant reads predicate and values as bare 8-byte words: not for a Boolean
            // predicate column (bits) nor nullable values (found by the differential fuzzer: a bitmap predicate with
            // a key shape the fast kernel does not cover was read as words)
            if (dense) { // a pass the partition kernels do not cover under a densely written table: the hashed table, the attempt again
                dense_ok = false;
                dense_redo = true;
                return PassStatus::Abort;
            }
            launch(ctx, "agg_grouped", pick_grouped_kernel(pk, kk, plain && !vnull && !bitmap_pred), dim3(std::min(grid, ctx->num_cus)), dim3(AGG_BLOCK), shmem, ka, tb.g,
                   ctx->d_flags);
        }
    }
    if (jit_whole && !jit_launched) { // a pass went through a static kernel, i.e. WITHOUT the predicate: this attempt's table is discarded
        jit_redo = true;
        return PassStatus::Abort;
    }
    return PassStatus::Done;
}

// what the flags of a finished attempt ask for.  true: the attempt is redone (the members say where)
bool AggRun::react_to_flags(const int *f, const Collected &pre) {
    (void)pre;
    if (f[NQE_FLAG_DIV_ZERO]) fail(NQE_ERR_ARROW, "Divide by zero");
    if (f[NQE_FLAG_OVERFLOW]) fail(NQE_ERR_ARROW, "attempt to divide with overflow");
    if (f[NQE_FLAG_OOB] && tiny_used) {
        // a key outside [0, m) — a negative Int64 — under the tiny-groups kernel: the streaming kernel, now and for this query shape's later executions
        tiny_ok = false;
        if (hint_key) {
            if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
            ctx->agg_key_ranges[hint_key ^ TINY_SALT] = std::make_pair(int64_t(0), uint64_t(0));
        }
        flags_reset(ctx);
        return true;
    }
    if ((f[NQE_FLAG_OOB] || f[NQE_FLAG_SLAB_OVERFLOW]) && partition_mode && range_part_used) {
        // a key outside the range the partitions were cut from (the sample missed it, the column changed), or key intervals of very
        // unequal weight: hashed partitions, now and for this query shape's later executions
        // (a range that came from the SAMPLE and missed a key: the hashed attempt's dense tail measures the exact one for the next
        // execution; a remembered range that no longer holds, or lopsided intervals: never again for this query shape)
        const bool remeasure = part_range_sampled && f[NQE_FLAG_OOB] && !f[NQE_FLAG_SLAB_OVERFLOW];
        range_part_used = false;
        part_span = 0;
        part_range_sampled = false;
        if (!remeasure) range_part_ok = false;
        if (hint_key && !remeasure) {
            if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
            ctx->agg_key_ranges[hint_key ^ PART_RANGE_SALT] = std::make_pair(int64_t(0), uint64_t(0));
        }
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_KEY32_OVERFLOW] && partition_mode && !key32_failed) {
        key32_failed = true; // a group key outside int32: the 16-byte tuple form
        if (hint_key) ctx->agg_hints[hint_key] = uint8_t(0x40 | (slab_parts_log2 < PARTS_LOG2 ? 16 : 1));
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_SLAB_OVERFLOW] && partition_mode && !slab_failed) {
        slab_failed = true; // a partition outgrew its slab (skewed keys): exact partition sizes instead
        if (hint_key) ctx->agg_hints[hint_key] = uint8_t(17 | (key32_failed ? 0x40 : 0)); // … and the next execution of this query shape starts there
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_NEED_LEVEL2] && partition_mode && !level2 && !slab_failed && slab_parts_log2 < PARTS_LOG2) {
        slab_parts_log2 = PARTS_LOG2; // a partition outgrew a workgroup table: the full partition count, still in slab form
        if (hint_key) ctx->agg_hints[hint_key] = uint8_t(1 | (key32_failed ? 0x40 : 0));
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_NEED_LEVEL2] && partition_mode && !level2) {
        level2 = true; // partitions hold more distinct keys than a workgroup table: one more partitioning level
        cap = std::max<uint32_t>(cap, 1u << 24);
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_NEED_PARTITION] && !partition_mode && range_on) {
        // a key outside the remembered range (the column's contents changed, or the range came from a sample that missed the
        // column's extremes), or a pass whose table is smaller than the range
        range_on = false;
        if (hint_key) ctx->agg_key_ranges.erase(hint_key);
        flags_reset(ctx);
        if (range_sampled && hint_key) {
            range_sampled = false;
            const auto exact = measure_key_range(); // the exact range: addressed by key - min after all, or remembered as too wide
            ctx->agg_key_ranges[hint_key] = exact;
            if (exact.second != 0 && exact.second <= one_table_or_subsets_limit()) {
                range_on = true;
                range_min = exact.first;
                range_span = exact.second;
            }
        }
        return true;
    }
    if (f[NQE_FLAG_NEED_PARTITION] && !partition_mode && three) {
        // more groups than the three-column instance's table holds: passes of one and two columns (2048 / 4096 slots) may still do
        three_on = false;
        if (hint_key) {
            if (ctx->agg_hints.size() >= 256) ctx->agg_hints.clear();
            ctx->agg_hints[hint_key] = 0x80;
        }
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_NEED_PARTITION] && !partition_mode) {
        // a workgroup table overflowed: two key subsets, and beyond those hash-partitioned rows
        bool to_subsets = subsets_ok && subsets_log2 < sw.subsets_max;
        if (to_subsets && subsets_log2 == 0 && plain_int_key && !sw.no_key_range && V == 1 && sw.direct_subsets && hint_key) {
            // a plain integer key column: its exact range (one pass, remembered) decides as the key sample does for queries without a predicate —
            // up to two tables' worth of values: the two subsets address their tables directly; a range the partitioned path's range tier takes:
            // that tier; anything wider: hashed subsets
            auto rt = ctx->agg_key_ranges.find(hint_key);
            if (rt == ctx->agg_key_ranges.end() || range_sampled) {
                if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
                ctx->agg_key_ranges[hint_key] = measure_key_range();
                rt = ctx->agg_key_ranges.find(hint_key);
                range_sampled = false;
            }
            const uint64_t span = rt->second.second;
            if (span != 0 && span <= direct_one_limit && attempt < 6) { // ONE table addressed by key - min takes it after all (the hashed attempt gave up at three quarters of its slots)
                range_on = true;
                range_min = rt->second.first;
                range_span = span;
                cap = std::max(cap, std::min(sized_cap, RANK_MAX_CAP << 1));
                ctx->agg_hints[hint_key] = uint8_t(key32_failed ? 0x40 : 0);
                flags_reset(ctx);
                return true;
            }
            if (span != 0 && span <= direct_pair_limit) {
                range_on = true;
                range_min = rt->second.first;
                range_span = span;
            } else if (range_part_ok && sw.range_tier && span != 0 && span < uint64_t(256) * RANGE_TIER_MAX_SLOTS) {
                to_subsets = false;
                if (span < (uint64_t(PARTS) << 12)) { // the measured (exact) range is what the range tier cuts its partitions from, in this execution already
                    part_min = rt->second.first;
                    part_span = span;
                    part_range_sampled = false;
                }
            }
        }
        if (to_subsets) {
            ++subsets_log2;
            cap = std::max(cap, std::min(sized_cap, RANK_MAX_CAP << subsets_log2));
        } else {
            subsets_log2 = 0;
            partition_mode = true;
            cap = std::max(cap, sized_cap);
        }
        if (hint_key) {
            if (ctx->agg_hints.size() >= 256) ctx->agg_hints.clear();
            ctx->agg_hints[hint_key] = uint8_t((partition_mode ? (slab_parts_log2 < PARTS_LOG2 ? 16 : 1) : 1 + subsets_log2) | (key32_failed ? 0x40 : 0));
        }
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_DENSE_OVERFLOW]) { // a sub-partition with more distinct keys than an LDS table: hashed global table instead
        dense_ok = false;
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_TABLE_FULL] && !partition_mode && asked_partition) {
        // More groups than the global table of the streaming tiers holds, while no workgroup's LDS table overflowed: a table of
        // 2^18 .. a few million rows with many groups (every workgroup sees fewer distinct keys than its table holds).  Growing
        // the global table would leave each workgroup folding its LDS table into it through device-scope atomics — as many of
        // them as rows (500 000 rows, 90 000 groups: 1.33 ms in the streaming kernel); the partitioned path has none.
        subsets_log2 = 0;
        partition_mode = true;
        cap = std::max(cap, sized_cap);
        if (hint_key) {
            if (ctx->agg_hints.size() >= 256) ctx->agg_hints.clear();
            ctx->agg_hints[hint_key] = uint8_t((slab_parts_log2 < PARTS_LOG2 ? 16 : 1) | (key32_failed ? 0x40 : 0));
        }
        flags_reset(ctx);
        return true;
    }
    if (f[NQE_FLAG_TABLE_FULL]) {
        // grow in 64 bits: 2^30 << 3 wraps a uint32_t to 0 (a zero-capacity table, then a spurious overflow error)
        const uint64_t next = cap < sized_cap ? uint64_t(sized_cap) : uint64_t(cap) << 3;
        if (cap >= (1u << 31) || attempt > 8) fail(NQE_ERR_OUT_OF_MEMORY, "group table overflow");
        cap = uint32_t(std::min<uint64_t>(next, uint64_t(1) << 31));
        flags_reset(ctx);
        return true;
    }
    return false;
}

// keys_out of a `group by <Utf8 column>`: every tier aggregates the Int64 codes (representative row of each distinct string,
// utf8_encode_build); the result carries the strings of those rows.  Every return of finish_attempt passes through here.
void AggRun::keys_to_strings(AggResult &res) {
    if (!utf8_key || !res.keys) return;
    DevColumn codes = res.keys->cols[0];
    res.keys->cols[0] = take_utf8(ctx, utf8_src, (const int64_t *)codes.words(), codes.length, false);
    sync(ctx);
}

// the tail ahead of the flag read-back, the reactions, the result.  false: redo the attempt
bool AggRun::finish_attempt(AggResult *out) {
    Collected pre;
    AggResult ranked;
    if (ranged.out) {
        // the range tier wrote keys and aggregates already; its group count rides along with the flags
    } else if (grouped && !tb.g.dense_count && tb.g.cap <= RANK_MAX_CAP) {
        // first-attempt table: the whole tail (collect, sort, finalize) runs ahead of the read-back
        ranked = emit_ranked(ctx, tb, kinfo.out_dtype, aggs, naggs, plan.vslot, partial);
    } else if (grouped && !tb.g.dense_count && tb.g.cap <= (1u << 16)) {
        // small table: collect speculatively; the count lands in the spare flag slot and is read with the flags
        const size_t slots = size_t(tb.g.cap) + 1;
        pre.keys = dev_alloc(ctx, slots * 8);
        pre.slots = dev_alloc(ctx, slots * 4);
        launch(ctx, "agg_collect", collect_kernel, dim3(stream_grid(ctx, int64_t(slots), 256)), dim3(256), 0, tb.g, (uint64_t *)pre.keys->ptr,
               (uint32_t *)pre.slots->ptr, reinterpret_cast<uint32_t *>(ctx->d_flags + NQE_FLAG_GROUP_COUNT));
    }
    const bool dense_tail = grouped && tb.g.dense_count != nullptr && !ranged.out;
    if (dense_tail) { // group count and key range of the densely written table ride along with the flags (emit: the ranked tail)
        launch(ctx, "agg_dense_key_range", dense_key_range_kernel, dim3(unsigned(std::min<int64_t>(128, (int64_t(tb.g.cap) + 4095) / 4096))), dim3(256), 0, (const uint64_t *)tb.g.keys, tb.g.cap,
               kinfo.out_dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull, tb.g.dense_count);
        NQE_HIP_CHECK(hipMemcpyAsync(ctx->h_flags + NQE_NUM_FLAGS, tb.g.dense_count, 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    int f[NQE_NUM_FLAGS];
    if (flagless) std::memset(f, 0, sizeof(f)); // nothing on this path raises a flag and the result has exactly one row: no read-back
    else if (ranked.out) flags_read_mirrored(ctx, f);
    else flags_read(ctx, f);
    if (pre.keys) pre.G = int64_t(uint32_t(f[NQE_FLAG_GROUP_COUNT]));
    if (dense_tail && !flagless) {
        const volatile int *x = ctx->h_flags + NQE_NUM_FLAGS;
        const uint64_t inv_min = uint64_t(uint32_t(x[2])) | (uint64_t(uint32_t(x[3])) << 32), mx = uint64_t(uint32_t(x[4])) | (uint64_t(uint32_t(x[5])) << 32);
        pre.dense_G = int64_t(std::min<uint32_t>(uint32_t(x[0]), tb.g.cap));
        pre.ordmin = ~inv_min;
        pre.ordmax = mx;
    }
    if (sw.debug)
        fprintf(stderr, "[nqe] aggregate attempt %d: partition %d subsets_log2 %d cap %u flags need_partition %d slab_overflow %d level2 %d table_full %d dense_overflow %d\n",
                attempt, int(partition_mode), subsets_log2, cap, f[NQE_FLAG_NEED_PARTITION], f[NQE_FLAG_SLAB_OVERFLOW], f[NQE_FLAG_NEED_LEVEL2],
                f[NQE_FLAG_TABLE_FULL], f[NQE_FLAG_DENSE_OVERFLOW]);
    if (react_to_flags(f, pre)) return false;
    if (ranged.out) {
        uint64_t sum3[3];
        std::memcpy(sum3, (const void *)(ctx->h_flags + NQE_NUM_FLAGS), sizeof(sum3));
        const int64_t G = int64_t(sum3[0]);
        sum3[1] = range_emit_key_min + ~sum3[1]; // (the tail leaves offsets from key_min: the first one complemented)
        sum3[2] = range_emit_key_min + sum3[2];
        set_group_count(ranged, G);
        if (hint_key && G > 0) { // the exact range of the groups (the keys come out in order): the next execution cuts its partitions from it
            const uint64_t flip = kinfo.out_dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull;
            const uint64_t lo = sum3[1] ^ flip, hi = sum3[2] ^ flip;
            if (hi >= lo && hi - lo < (uint64_t(PARTS) << 12)) {
                if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
                ctx->agg_key_ranges[hint_key ^ PART_RANGE_SALT] = std::make_pair(int64_t(sum3[1]), hi - lo + 1);
            }
        }
        keys_to_strings(ranged);
        *out = std::move(ranged);
        return true;
    }
    if (pre.dense_G > 0 && hint_key && range_part_ok && pre.ordmax >= pre.ordmin && pre.ordmax - pre.ordmin < (uint64_t(PARTS) << 12)) {
        // the exact key range of this query's groups: the next execution cuts its partitions from it
        const uint64_t flip = kinfo.out_dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull;
        if (ctx->agg_key_ranges.size() >= 256) ctx->agg_key_ranges.clear();
        ctx->agg_key_ranges[hint_key ^ PART_RANGE_SALT] = std::make_pair(int64_t(pre.ordmin ^ flip), pre.ordmax - pre.ordmin + 1);
        if (sw.debug)
            fprintf(stderr, "[nqe] aggregate: remembered key range %lld + %llu (hint %llx)\n", (long long)int64_t(pre.ordmin ^ flip), (unsigned long long)(pre.ordmax - pre.ordmin + 1),
                    (unsigned long long)hint_key);
    }
    AggResult res;
    if (ranked.out) {
        set_group_count(ranked, int64_t(uint32_t(f[NQE_FLAG_GROUP_COUNT])));
        res = std::move(ranked);
    } else
        res = emit(ctx, tb, grouped, grouped ? kinfo.out_dtype : NQE_INT64, aggs, naggs, plan.vslot, partial, &pre);
    keys_to_strings(res);
    *out = std::move(res);
    return true;
}

AggResult AggRun::run() {
    for (attempt = 0;; ++attempt) {
        begin_attempt();
        for (int v0 = 0; v0 < std::max(V, 1); v0 += pass_nv)
            if (launch_pass(v0) == PassStatus::Abort) break;
        if (jit_redo) { // the specialised kernel was not to be had after all: the predicate as a Boolean column, the attempt again
            jit_redo = jit_whole = false;
            materialize_pred();
            flags_reset(ctx);
            continue;
        }
        if (dense_redo) flags_reset(ctx);
        if (slab_oom || three_redo || dense_redo) continue;
        AggResult res;
        if (finish_attempt(&res)) return res;
    }
}

AggResult run_aggregate(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *group, int group_nodes,
                        const nqe_aggregate *aggs, int naggs, bool partial) {
    AggRun r(ctx, in, pred, pred_nodes, group, group_nodes, aggs, naggs, partial);
    if (r.prepare()) return std::move(r.early);
    r.size_tables();
    r.load_hints();
    r.sample_keys();
    r.pick_key_range();
    return r.run();
}

} // namespace

} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_aggregate_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int32_t pred_nodes,
                                 const nqe_expr_node *group, int32_t group_nodes, const nqe_aggregate *aggs,
                                 int32_t num_aggs, nqe_table **out, nqe_table **keys_out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !out || (num_aggs > 0 && !aggs)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    flags_reset(ctx);
    AggResult r = run_aggregate(ctx, in, pred, pred_nodes, group, group_nodes, aggs, num_aggs, false);
    *out = r.out.release();
    if (keys_out) *keys_out = r.keys.release();
    NQE_API_END()
}

nqe_status nqe_aggregate_partial(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int32_t pred_nodes,
                                 const nqe_expr_node *group, int32_t group_nodes, const nqe_aggregate *aggs,
                                 int32_t num_aggs, nqe_table **state_out, nqe_table **keys_out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !state_out || (num_aggs > 0 && !aggs)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (group && group_nodes > 0 && !keys_out) fail(NQE_ERR_INVALID_ARGUMENT, "keys_out is required for grouped partials");
    flags_reset(ctx);
    AggResult r = run_aggregate(ctx, in, pred, pred_nodes, group, group_nodes, aggs, num_aggs, true);
    *state_out = r.out.release();
    if (keys_out) *keys_out = r.keys.release();
    NQE_API_END()
}

nqe_status nqe_aggregate_merge(nqe_ctx *ctx, const nqe_table *const *states, const nqe_table *const *keys, int32_t n,
                               const nqe_aggregate *aggs, int32_t num_aggs, nqe_table **out, nqe_table **keys_out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !states || n <= 0 || !out || num_aggs < 0 || num_aggs > 16 || (num_aggs > 0 && !aggs))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    const bool grouped = keys != nullptr && keys[0] != nullptr;
    int key_dtype = NQE_INT64;
    int64_t total = 0;
    int V = 0;
    const std::vector<int> vslot = slots_of_aggs(aggs, num_aggs, &V);
    for (int k = 0; k < n; ++k) {
        if (!states[k] || int(states[k]->cols.size()) != 4 * V) fail(NQE_ERR_INVALID_ARGUMENT, "state table shape mismatch");
        if (grouped) {
            if (!keys[k] || keys[k]->cols.size() != 1 || keys[k]->rows != states[k]->rows)
                fail(NQE_ERR_INVALID_ARGUMENT, "keys table shape mismatch");
            key_dtype = keys[k]->cols[0].dtype;
        }
        total += states[k]->rows;
    }
    flags_reset(ctx);
    uint32_t cap = 1;
    if (grouped) {
        cap = 4096;
        while (int64_t(cap) < 2 * total) cap <<= 1;
    }
    TableBufs tb = make_table(ctx, cap, V, !grouped);
    BufRef ptrs = dev_alloc(ctx, size_t(4 * std::max(V, 1)) * sizeof(void *));
    for (int k = 0; k < n; ++k) {
        int64_t rows = states[k]->rows;
        if (rows == 0 || V == 0) {
            if (rows && grouped) { /* keys still need inserting */ } else continue;
        }
        std::vector<const uint64_t *> h(size_t(4 * std::max(V, 1)), nullptr);
        for (int c = 0; c < 4 * V; ++c) h[size_t(c)] = states[k]->cols[size_t(c)].words();
        NQE_HIP_CHECK(hipMemcpyAsync(ptrs->ptr, h.data(), h.size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
        launch(ctx, "agg_merge_states", merge_states_kernel, dim3(stream_grid(ctx, rows, 256)), dim3(256), 0, tb.g,
               grouped ? keys[k]->cols[0].words() : (const uint64_t *)nullptr, rows, V,
               (const uint64_t *const *)ptrs->ptr, ctx->d_flags);
        sync(ctx); // `h` / ptrs are reused by the next partial
    }
    AggResult r;
    if (grouped && cap <= RANK_MAX_CAP) {
        // the exchanged states of a small group set (the sharded headline: world x 1024 rows): tail ahead of the read-back
        r = emit_ranked(ctx, tb, key_dtype, aggs, num_aggs, vslot, false);
        int f[NQE_NUM_FLAGS];
        flags_read_mirrored(ctx, f);
        if (f[NQE_FLAG_TABLE_FULL]) fail(NQE_ERR_OUT_OF_MEMORY, "device hash table overflow");
        set_group_count(r, int64_t(uint32_t(f[NQE_FLAG_GROUP_COUNT])));
    } else {
        throw_on_flags(ctx);
        r = emit(ctx, tb, grouped, key_dtype, aggs, num_aggs, vslot, false);
    }
    *out = r.out.release();
    if (keys_out) *keys_out = r.keys.release();
    NQE_API_END()
}

nqe_status nqe_aggregate_merge_packed(nqe_ctx *ctx, const void *gathered_device, int32_t num_parts, int64_t stride_rows, int32_t grouped,
                                      int32_t key_dtype, const nqe_aggregate *aggs, int32_t num_aggs, nqe_table **out, nqe_table **keys_out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !gathered_device || num_parts <= 0 || stride_rows <= 0 || !out || num_aggs < 0 || num_aggs > 16 || (num_aggs > 0 && !aggs) ||
        (grouped && !keys_out))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    if (keys_out) *keys_out = nullptr;
    int V = 0;
    const std::vector<int> vslot = slots_of_aggs(aggs, num_aggs, &V);
    const int64_t bound = int64_t(num_parts) * stride_rows;
    uint32_t sized_cap = 1;
    if (grouped) {
        sized_cap = 4096;
        while (int64_t(sized_cap) < 2 * bound) sized_cap <<= 1;
    }
    // small table first (the sharded headline merges world x 1024 rows of the same 1024 keys): single-launch tail, one read-back
    uint32_t cap = grouped ? std::min(sized_cap, RANK_MAX_CAP) : 1u;
    for (;;) {
        flags_reset(ctx);
        TableBufs tb = make_table(ctx, cap, V, !grouped);
        launch(ctx, "agg_merge_packed", merge_packed_kernel, dim3(stream_grid(ctx, std::min<int64_t>(bound, int64_t(1) << 16), 256)), dim3(256), 0, tb.g,
               (const uint64_t *)gathered_device, int(num_parts), stride_rows, grouped ? 1 : 0, V, ctx->d_flags);
        AggResult r;
        int f[NQE_NUM_FLAGS];
        const bool ranked = grouped && cap <= RANK_MAX_CAP;
        if (ranked) {
            r = emit_ranked(ctx, tb, key_dtype, aggs, num_aggs, vslot, false);
            flags_read_mirrored(ctx, f);
        } else
            flags_read(ctx, f);
        if (f[NQE_FLAG_OOB]) break; // some part was sent header-only: the caller takes the exact-size path (*out stays NULL)
        if (f[NQE_FLAG_TABLE_FULL]) {
            if (cap >= sized_cap) fail(NQE_ERR_OUT_OF_MEMORY, "device hash table overflow");
            cap = sized_cap;
            continue;
        }
        if (ranked) set_group_count(r, int64_t(uint32_t(f[NQE_FLAG_GROUP_COUNT])));
        else r = emit(ctx, tb, grouped != 0, key_dtype, aggs, num_aggs, vslot, false);
        *out = r.out.release();
        if (keys_out) *keys_out = r.keys.release();
        break;
    }
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::agg_ungrouped_fold_kernel);
