// aggregate_partition.hip — partitioned aggregation for more distinct keys than a workgroup's LDS table holds.
// More distinct keys than a workgroup's LDS table holds would turn every row into device-scope atomics, and
// those run at a flat ≈2.4e10 ops/s on MI355X whatever the scope, table size or layout (tools/atomics_bench.hip):
// 100 M rows took 17-20 ms at 4 K…1 M groups versus 0.6 ms at 1 K.  Instead the passing rows are hash-partitioned
// (count → scan → scatter of (key, values) tuples, PARTS partitions so that a workgroup's open write lines stay L2
// resident) and each partition — whose distinct keys now fit an LDS table — is aggregated by one workgroup.
#include "aggregate_common.hpp"

namespace nqe {
namespace agg {
namespace {

template <int PRED, int KEY, int NVT, bool SCATTER>
__global__ void __launch_bounds__(AGG_BLOCK) agg_partition_kernel(AggArgs a, FastPred fp, PartArgs pa) {
    __shared__ uint32_t cnt[PARTS];
    __shared__ uint64_t basep[PARTS];
    for (int p = threadIdx.x; p < PARTS; p += blockDim.x) {
        cnt[p] = 0;
        if (SCATTER) basep[p] = pa.offsets[size_t(p) * gridDim.x + blockIdx.x];
    }
    __syncthreads();
    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(PRED >= 2 ? a.pred_src.values : a.key_src.values);
    const uint64_t *__restrict__ valp[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) valp[j] = static_cast<const uint64_t *>(a.val[j].values);
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const OpAux key_aux = a.key.aux[0];
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;
    const int64_t lo = int64_t(blockIdx.x) * pa.chunk;
    const int64_t hi = lo + pa.chunk < a.n ? lo + pa.chunk : a.n;
    const int64_t last = a.n - 1;
    for (int64_t base = lo; base < hi; base += int64_t(AGG_BLOCK) * AGG_U) {
        uint64_t kw[AGG_U], pw[AGG_U], vw[NVT][AGG_U];
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            row = row < last ? row : last;
            kw[u] = __builtin_nontemporal_load(&keyp[row]);
            if (PRED == 2) pw[u] = __builtin_nontemporal_load(&predp[row >> fp.row_shift]);
            if (PRED == 3) pw[u] = __builtin_nontemporal_load(&predp[row]);
            if (SCATTER) {
#pragma unroll
                for (int j = 0; j < NVT; ++j) vw[j][u] = __builtin_nontemporal_load(&valp[j][row]);
            }
        }
#pragma unroll
        for (int u = 0; u < AGG_U; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            bool pass = row < hi;
            if (PRED == 3) pass = pass && eval_simple<false>(a.pred, pw[u], false, nullptr) != 0; // interpreted chain (host-vetted: cannot fault)
            else if (PRED != 0) pass = pass && range_pass(fp, PRED == 1 ? kw[u] : pred_extract(fp, pw[u], row));
            const uint64_t key = inline_key<KEY>(a.key, kw[u], key_mask, key_aux, key_signed);
            if (!pass) continue;
            uint32_t p = uint32_t((key * GOLD) >> (64 - PARTS_LOG2));
            uint32_t r = atomicAdd(&cnt[p], 1u);
            if (SCATTER) {
                uint64_t pos = basep[p] + r;
                pa.out_key[pos] = key;
#pragma unroll
                for (int j = 0; j < NVT; ++j) pa.out_val[j][pos] = vw[j][u];
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int p = threadIdx.x; p < PARTS; p += blockDim.x) pa.counts[size_t(p) * gridDim.x + blockIdx.x] = cnt[p];
    }
}

// Scatter pass with LDS write-combining: a tile of SC_ROWS rows is counting-sorted by partition inside LDS
// (rank = LDS atomic on a per-tile counter, tile-local exclusive scan), then copied out so that consecutive lanes
// write consecutive tuples of the same partition (runs of SC_ROWS/PARTS tuples → full 128-B lines instead of
// 8-byte stores sprayed over 512 streams: 2.4 ms → see DESIGN.md for the measured effect).
template <int PRED, int KEY, int NVT>
__global__ void __launch_bounds__(AGG_BLOCK) agg_partition_scatter_kernel(AggArgs a, FastPred fp, PartArgs pa) {
    constexpr int RPT = NVT == 1 ? 8 : 4;            // rows per thread per tile
    constexpr int SC_ROWS = AGG_BLOCK * RPT;         // 8192 (one value column) / 4096 (two)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem);     // [SC_ROWS]
    uint64_t *sval = skey + SC_ROWS;                          // [NVT][SC_ROWS]
    uint64_t *gcur = sval + NVT * SC_ROWS;                    // [PARTS] global write cursor of this workgroup
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(gcur + PARTS); // [PARTS] tuples of this tile per partition
    uint32_t *tstart = tcnt + PARTS;                          // [PARTS] tile-local exclusive scan
    __shared__ uint32_t wave_tot[AGG_BLOCK / 64];
    for (int p = threadIdx.x; p < PARTS; p += blockDim.x) {
        gcur[p] = pa.offsets[size_t(p) * gridDim.x + blockIdx.x];
        tcnt[p] = 0;
    }
    __syncthreads();
    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(PRED >= 2 ? a.pred_src.values : a.key_src.values);
    const uint64_t *__restrict__ valp[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) valp[j] = static_cast<const uint64_t *>(a.val[j].values);
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const OpAux key_aux = a.key.aux[0];
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;
    const int64_t lo = int64_t(blockIdx.x) * pa.chunk;
    const int64_t hi = lo + pa.chunk < a.n ? lo + pa.chunk : a.n;
    const int64_t last = a.n - 1;
    for (int64_t base = lo; base < hi; base += SC_ROWS) {
        uint64_t key[RPT], vw[NVT][RPT];
        uint32_t part[RPT], rank[RPT];
        bool pass[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            int64_t rc = row < last ? row : last;
            uint64_t kw = __builtin_nontemporal_load(&keyp[rc]);
            uint64_t pw = PRED == 2 ? pred_extract(fp, __builtin_nontemporal_load(&predp[rc >> fp.row_shift]), rc)
                                    : (PRED == 3 ? __builtin_nontemporal_load(&predp[rc]) : kw);
#pragma unroll
            for (int j = 0; j < NVT; ++j) vw[j][u] = __builtin_nontemporal_load(&valp[j][rc]);
            bool ok = row < hi;
            if (PRED == 3) ok = ok && eval_simple<false>(a.pred, pw, false, nullptr) != 0;
            else if (PRED != 0) ok = ok && range_pass(fp, pw);
            key[u] = inline_key<KEY>(a.key, kw, key_mask, key_aux, key_signed);
            pass[u] = ok;
            part[u] = uint32_t((key[u] * GOLD) >> (64 - PARTS_LOG2));
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) rank[u] = pass[u] ? atomicAdd(&tcnt[part[u]], 1u) : 0u;
        __syncthreads();
        // tile-local exclusive scan of the PARTS counters (threads 0..PARTS-1)
        uint32_t c = threadIdx.x < PARTS ? tcnt[threadIdx.x] : 0u, wt;
        uint32_t ex = wave_exclusive_scan(c, wt);
        if (lane_id() == 63) wave_tot[threadIdx.x / 64] = wt;
        __syncthreads();
        if (threadIdx.x < PARTS) {
            uint32_t pre = 0;
            for (int w = 0; w < int(threadIdx.x) / 64; ++w) pre += wave_tot[w];
            tstart[threadIdx.x] = pre + ex;
        }
        uint32_t tile_total = 0;
        for (int w = 0; w < PARTS / 64; ++w) tile_total += wave_tot[w];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (!pass[u]) continue;
            uint32_t i = tstart[part[u]] + rank[u];
            skey[i] = key[u];
#pragma unroll
            for (int j = 0; j < NVT; ++j) sval[j * SC_ROWS + i] = vw[j][u];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tile_total; i += blockDim.x) {
            uint64_t k = skey[i];
            uint32_t p = uint32_t((k * GOLD) >> (64 - PARTS_LOG2));
            uint64_t dest = gcur[p] + (i - tstart[p]);
            pa.out_key[dest] = k;
#pragma unroll
            for (int j = 0; j < NVT; ++j) pa.out_val[j][dest] = sval[j * SC_ROWS + i];
        }
        __syncthreads();
        if (threadIdx.x < PARTS) {
            gcur[threadIdx.x] += tcnt[threadIdx.x];
            tcnt[threadIdx.x] = 0;
        }
        __syncthreads();
    }
}

// Second partitioning level: workgroup p splits parent partition p (a contiguous tuple range) into SUB sub-partitions
// by the next SUB_LOG2 hash bits — count, tile-local scan, then the same LDS-sorted scatter as level 1.  The output
// occupies the same global range as the input partition, so no cross-workgroup scan is needed.
template <int NVT>
__global__ void __launch_bounds__(AGG_BLOCK) agg_subpartition_kernel(const uint64_t *offsets, int64_t off_stride, const uint64_t *in_key,
                                                                     const uint64_t *in_v0, const uint64_t *in_v1, uint64_t *out_key,
                                                                     uint64_t *out_v0, uint64_t *out_v1, uint64_t *sub_offsets) {
    constexpr int RPT = NVT == 1 ? 8 : 4;
    constexpr int SC_ROWS = AGG_BLOCK * RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem);
    uint64_t *sval = skey + SC_ROWS;
    __shared__ uint64_t gcur[SUB];
    __shared__ uint32_t tcnt[SUB], tstart[SUB], total_cnt[SUB];
    const uint64_t *__restrict__ inv[2] = {in_v0, in_v1};
    uint64_t *outv[2] = {out_v0, out_v1};
    for (int p = blockIdx.x; p < PARTS; p += gridDim.x) {
        const int64_t lo = int64_t(offsets[int64_t(p) * off_stride]), hi = int64_t(offsets[int64_t(p + 1) * off_stride]);
        __syncthreads();
        if (threadIdx.x < SUB) total_cnt[threadIdx.x] = 0, tcnt[threadIdx.x] = 0;
        __syncthreads();
        // ---- count
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            uint32_t sp = uint32_t(((in_key[i] * GOLD) << PARTS_LOG2) >> (64 - SUB_LOG2));
            atomicAdd(&total_cnt[sp], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t run = uint64_t(lo);
            for (int sp = 0; sp < SUB; ++sp) {
                gcur[sp] = run;
                sub_offsets[int64_t(p) * SUB + sp] = run;
                run += total_cnt[sp];
            }
            if (p == PARTS - 1) sub_offsets[int64_t(PARTS) * SUB] = run;
        }
        __syncthreads();
        // ---- scatter, tile by tile, sorted in LDS first
        for (int64_t base = lo; base < hi; base += SC_ROWS) {
            uint64_t key[RPT], vw[NVT][RPT];
            uint32_t part[RPT], rank[RPT];
            bool pass[RPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
                pass[u] = row < hi;
                int64_t rc = pass[u] ? row : hi - 1;
                key[u] = in_key[rc];
#pragma unroll
                for (int j = 0; j < NVT; ++j) vw[j][u] = inv[j][rc];
                part[u] = uint32_t(((key[u] * GOLD) << PARTS_LOG2) >> (64 - SUB_LOG2));
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) rank[u] = pass[u] ? atomicAdd(&tcnt[part[u]], 1u) : 0u;
            __syncthreads();
            if (threadIdx.x < 64) { // SUB == 64: one wave scans the tile counters
                uint32_t c = tcnt[threadIdx.x], wt;
                tstart[threadIdx.x] = wave_exclusive_scan(c, wt);
            }
            __syncthreads();
            uint32_t tile_total = tstart[SUB - 1] + tcnt[SUB - 1];
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (!pass[u]) continue;
                uint32_t i = tstart[part[u]] + rank[u];
                skey[i] = key[u];
#pragma unroll
                for (int j = 0; j < NVT; ++j) sval[j * SC_ROWS + i] = vw[j][u];
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < tile_total; i += blockDim.x) {
                uint64_t k = skey[i];
                uint32_t sp = uint32_t(((k * GOLD) << PARTS_LOG2) >> (64 - SUB_LOG2));
                uint64_t dest = gcur[sp] + (i - tstart[sp]);
                out_key[dest] = k;
#pragma unroll
                for (int j = 0; j < NVT; ++j) outv[j][dest] = sval[j * SC_ROWS + i];
            }
            __syncthreads();
            if (threadIdx.x < SUB) {
                gcur[threadIdx.x] += tcnt[threadIdx.x];
                tcnt[threadIdx.x] = 0;
            }
            __syncthreads();
        }
    }
}

// one workgroup per partition (grid-stride over partitions): plain (key, values) tuples → LDS table → global table.
// The LDS slot uses the hash bits BELOW the partition bits (all keys of a partition share the top PARTS_LOG2 bits).
template <int NVT, bool VF64>
__global__ void __launch_bounds__(AGG_BLOCK) agg_segments_kernel(AggArgs a, const uint64_t *seg_offsets, int64_t seg_stride, int nsegs, int part_bits,
                                                                 int signal_level2, const uint64_t *keys,
                                                                 const uint64_t *v0, const uint64_t *v1, GroupTable g, int *flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t cap = uint32_t(a.lds_cap);
    const uint32_t slots = cap + 1;
    uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
    double *lsum = reinterpret_cast<double *>(lkeys + slots);
    uint64_t *lmn = reinterpret_cast<uint64_t *>(lsum + NVT * slots);
    uint64_t *lmx = lmn + NVT * slots;
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(lmx + NVT * slots);
    const uint64_t ORD_MAX = f64_to_ord(DBL_MAX), ORD_MIN = f64_to_ord(-DBL_MAX);
    const uint64_t *__restrict__ valp[2] = {v0, v1};
    int vdt[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) vdt[j] = a.val[j].dtype;
    // read and written as an LDS word (ds_read / ds_write): through a generic `volatile int *` the accesses were FLAT loads, each
    // followed by s_waitcnt vmcnt(0) — on every first-probe miss the wave waited for the tuples it had just prefetched
    __shared__ int seg_full_flag;
#define SEG_FULL() __hip_atomic_load(&seg_full_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SEG_FULL_SET() __hip_atomic_store(&seg_full_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    for (int seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        __syncthreads();
        if (signal_level2 && __hip_atomic_load(&flags[NQE_FLAG_NEED_LEVEL2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (threadIdx.x == 0) seg_full_flag = 0;
        for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
            lkeys[s] = EMPTY_KEY;
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                lsum[j * slots + s] = 0.0;
                lmn[j * slots + s] = ORD_MAX;
                lmx[j * slots + s] = ORD_MIN;
                lcnt[j * slots + s] = 0;
            }
        }
        __syncthreads();
        const int64_t lo = int64_t(seg_offsets[int64_t(seg) * seg_stride]), hi = int64_t(seg_offsets[int64_t(seg + 1) * seg_stride]);
        for (int64_t base = lo; base < hi; base += int64_t(AGG_BLOCK) * AGG_U) {
            uint64_t kw[AGG_U], vw[NVT][AGG_U];
#pragma unroll
            for (int u = 0; u < AGG_U; ++u) {
                int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
                row = row < hi - 1 ? row : hi - 1;
                kw[u] = keys[row];
#pragma unroll
                for (int j = 0; j < NVT; ++j) vw[j][u] = valp[j][row];
            }
#pragma unroll
            for (int u = 0; u < AGG_U; ++u) {
                int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
                if (row >= hi) continue;
                const uint64_t key = kw[u];
                // tuples of a partition arrive in no particular order: no run cache, one table update per row
                int slot;
                if (key == EMPTY_KEY) { lkeys[cap] = 0; slot = int(cap); }
                else if (SEG_FULL()) slot = -1; // this partition has more distinct keys than the table: spill the rest
                else {
                    uint32_t sl = uint32_t(((key * GOLD) << part_bits) >> a.lds_shift);
                    slot = -1;
                    for (int probe = 0; probe < 32; ++probe) {
                        uint64_t k = lkeys[sl];
                        if (k == key) { slot = int(sl); break; }
                        if (k == EMPTY_KEY) {
                            uint64_t old = atomicCAS((unsigned long long *)&lkeys[sl], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                            if (old == EMPTY_KEY || old == key) { slot = int(sl); break; }
                        }
                        sl = (sl + 1) & (cap - 1);
                    }
                }
                if (slot < 0 && !SEG_FULL()) {
                    SEG_FULL_SET();
                    if (signal_level2) atomicOr(&flags[NQE_FLAG_NEED_LEVEL2], 1); // the host re-partitions one level deeper
                }
                if (slot < 0 && signal_level2) continue;                              // result will be discarded
                if (slot < 0 && g.dense_count) {                                      // no hash table to spill to: the host falls back
                    atomicOr(&flags[NQE_FLAG_DENSE_OVERFLOW], 1);
                    continue;
                }
                int64_t gslot = slot < 0 ? global_find_or_insert(g, key, flags) : 0; // partition larger than the table: spill
#pragma unroll
                for (int j = 0; j < NVT; ++j) {
                    double x = VF64 ? u2d(vw[j][u]) : word_as_f64(vw[j][u], vdt[j]);
                    bool isn = x != x;
                    uint64_t xo = f64_to_ord(x);
                    if (slot >= 0) {
                        uint32_t o = uint32_t(j) * slots + uint32_t(slot);
                        atomicAdd(&lcnt[o], 1u);
                        if (isn) atomicOr(&lcnt[o], NAN_BIT);
                        unsafeAtomicAdd(&lsum[o], x);
                        if (!isn) {
                            if (xo < lmn[o]) atomicMin((unsigned long long *)&lmn[o], (unsigned long long)xo);
                            if (xo > lmx[o]) atomicMax((unsigned long long *)&lmx[o], (unsigned long long)xo);
                        }
                    } else if (gslot >= 0) {
                        global_update(g, gslot, a.v0 + j, 1, x, true, xo, xo, !isn, isn);
                    }
                }
            }
        }
        __syncthreads();
        if (g.dense_count) {
            // ---- dense output: count this partition's groups, reserve [base, base + n) with one atomic, write them there
            __shared__ uint32_t wave_tot[AGG_BLOCK / 64];
            __shared__ uint32_t dense_base;
            uint32_t mine = 0;
            for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) mine += lkeys[s] != EMPTY_KEY ? 1u : 0u;
            uint32_t wtot;
            const uint32_t wexcl = wave_exclusive_scan(mine, wtot);
            if (lane_id() == 0) wave_tot[threadIdx.x / 64] = wtot;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tot = 0;
                for (int w = 0; w < AGG_BLOCK / 64; ++w) { uint32_t c = wave_tot[w]; wave_tot[w] = tot; tot += c; }
                dense_base = tot ? atomicAdd(g.dense_count, tot) : 0u;
            }
            __syncthreads();
            uint32_t pos = dense_base + wave_tot[threadIdx.x / 64] + wexcl;
            const size_t gstride = size_t(g.cap) + 1;
            for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
                uint64_t k = lkeys[s];
                if (k == EMPTY_KEY) continue;
                if (pos < g.cap) {
                    g.keys[pos] = (s == cap) ? EMPTY_KEY : k;
#pragma unroll
                    for (int j = 0; j < NVT; ++j) {
                        const uint32_t o = uint32_t(j) * slots + s;
                        const uint32_t c = lcnt[o];
                        const size_t go = size_t(a.v0 + j) * gstride + pos;
                        g.cnt[go] = uint64_t(c & ~NAN_BIT);
                        g.sum[go] = lsum[o];
                        g.mn[go] = lmn[o];
                        g.mx[go] = lmx[o];
                        g.nan[go] = (c & NAN_BIT) ? 1u : 0u;
                    }
                } else atomicOr(&flags[NQE_FLAG_DENSE_OVERFLOW], 1);
                ++pos;
            }
            continue;
        }
        for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
            uint64_t k = lkeys[s];
            if (k == EMPTY_KEY) continue;
            uint64_t key = (s == cap) ? EMPTY_KEY : k;
            int64_t gslot = global_find_or_insert(g, key, flags);
            if (gslot < 0) continue;
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                uint32_t o = uint32_t(j) * slots + s;
                uint32_t c = lcnt[o];
                global_update(g, gslot, a.v0 + j, uint64_t(c & ~NAN_BIT), lsum[o], true, lmn[o], lmx[o], true, (c & NAN_BIT) != 0);
            }
        }
    }
}

// ------------------------------------------------------------------ slab form: no count pass, software-pipelined scatter
// Rows per thread per tile: 8 where the registers allow two tiles in flight (one value column, predicate on the key column or
// none), else 4.
#ifndef NQE_SLAB_KEYMOD_RPT
#define NQE_SLAB_KEYMOD_RPT 8 // rows per thread of the `col % m` key variants: 8 spill 14-16 VGPRs and are still faster than 4 without (scatter 0.85 vs 0.93 ms)
#endif
#ifndef NQE_SLAB_WG_PER_CU
#define NQE_SLAB_WG_PER_CU 1 // scatter workgroups per CU (A/B): 2 = half tiles (4 rows per thread, 70 KB of LDS each) whose barrier phases overlap —
                             // measured slower: kernels 1.06 -> 1.12 ms at 65536 groups, 1.17 -> 1.29 at 2^20 (shorter runs per partition per tile)
#endif
template <int PRED, int KEY, int NVT> struct SlabShape {
    static constexpr int RPT = NQE_SLAB_WG_PER_CU > 1 ? 4 : ((NVT == 1 && PRED <= 1 && (KEY == 0 || (NQE_SLAB_KEYMOD_RPT == 8 && KEY != 3))) ? 8 : 4);
};

// One workgroup per chunk of rows.  Per tile: fused predicate + key → partition → rank (LDS atomic on the tile's counter) →
// tile-local scan → tuples written to LDS at their sorted position → copied out so that consecutive lanes write consecutive
// tuples of one partition into THIS workgroup's slab of it.  The next tile's words are requested before the copy-out, so the
// loads overlap the LDS phases and the stores (one workgroup per CU fits — 136 KB of LDS — and the unpipelined form spent
// 22 µs per 8192-row tile where the CU's share of HBM needs 13).
// (Tuples of 16 or 24 bytes — a 64-bit key, or two value columns.  One value column under a key that fits 32 bits takes the two-stream
// whole-block form: agg_slab_scatter_soa_kernel below.)
#ifdef NQE_SLAB_PROFILE
// diagnostic build (tools/probe_slab_phases.py): shader-clock time of thread 0 of every scatter workgroup per phase of a tile
__device__ unsigned long long nqe_slab_prof[8];
#define SLAB_STAMP(i)                                   \
    do {                                                \
        if (threadIdx.x == 0) {                         \
            const unsigned long long now_ = clock64();  \
            prof_acc[i] += now_ - prof_t;               \
            prof_t = now_;                              \
        }                                               \
    } while (0)
#else
#define SLAB_STAMP(i) do { } while (0)
#endif
template <int PRED, int KEY, int NVT>
__global__ void __launch_bounds__(AGG_BLOCK) agg_slab_scatter_kernel(AggArgs a, FastPred fp, SlabArgs sa, int *flags) {
    constexpr int RPT = SlabShape<PRED, KEY, NVT>::RPT;
    constexpr int SC_ROWS = AGG_BLOCK * RPT;
    constexpr int TW = 1 + NVT; // words per tuple
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *stup = reinterpret_cast<uint64_t *>(smem);                    // [SC_ROWS][TW]
    uint32_t *gcur = reinterpret_cast<uint32_t *>(stup + size_t(SC_ROWS) * TW); // [PARTS] tuples this workgroup has written per partition
    uint32_t *tcnt = gcur + PARTS;                                          // [PARTS] tuples of this tile per partition
    uint32_t *tstart = tcnt + PARTS;                                        // [PARTS] tile-local exclusive scan
    __shared__ uint32_t wave_tot[AGG_BLOCK / 64];
    const int parts_log2 = sa.parts_log2, parts = 1 << parts_log2; // <= PARTS (the LDS counters are sized for PARTS)
    for (int p = threadIdx.x; p < PARTS; p += blockDim.x) {
        gcur[p] = 0;
        tcnt[p] = 0;
    }
    __syncthreads();
    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(PRED >= 2 ? a.pred_src.values : a.key_src.values);
    const uint64_t *__restrict__ valp[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) valp[j] = static_cast<const uint64_t *>(a.val[j].values);
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const OpAux key_aux = a.key.aux[0];
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;
    const int64_t lo = int64_t(blockIdx.x) * sa.chunk;
    const int64_t hi = lo + sa.chunk < a.n ? lo + sa.chunk : a.n;
    const int64_t last = a.n - 1;
    const uint32_t cap = uint32_t(sa.cap);
    struct Regs {
        uint64_t kw[RPT], pw[PRED >= 2 ? RPT : 1], vw[NVT][RPT];
    };
    auto load = [&](Regs &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            row = row < last ? row : last; // clamp: unconditional, in-bounds
            r.kw[u] = __builtin_nontemporal_load(&keyp[row]);
            if (PRED == 2) r.pw[PRED >= 2 ? u : 0] = __builtin_nontemporal_load(&predp[row >> fp.row_shift]);
            if (PRED == 3) r.pw[PRED >= 2 ? u : 0] = __builtin_nontemporal_load(&predp[row]);
#pragma unroll
            for (int j = 0; j < NVT; ++j) r.vw[j][u] = __builtin_nontemporal_load(&valp[j][row]);
        }
    };
#ifdef NQE_SLAB_PROFILE
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = clock64();
#endif
    auto tile = [&](const Regs &r, Regs &next, int64_t base) {
        uint64_t key[RPT];
        uint32_t part[RPT], rank[RPT];
        bool pass[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            bool ok = row < hi;
            if (PRED == 3) ok = ok && eval_simple<false>(a.pred, r.pw[PRED >= 2 ? u : 0], false, nullptr) != 0; // host-vetted chain: cannot fault
            else if (PRED == 2) ok = ok && range_pass(fp, pred_extract(fp, r.pw[PRED >= 2 ? u : 0], row < last ? row : last));
            else if (PRED == 1) ok = ok && range_pass(fp, r.kw[u]);
            key[u] = inline_key<KEY>(a.key, r.kw[u], key_mask, key_aux, key_signed);
            part[u] = uint32_t((key[u] * GOLD) >> (64 - parts_log2));
            pass[u] = ok;
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) rank[u] = pass[u] ? atomicAdd(&tcnt[part[u]], 1u) : 0u;
        __syncthreads();
        SLAB_STAMP(0); // wait for the tile's words, key / partition, rank atomics
        // tile-local exclusive scan of the PARTS counters (threads 0..PARTS-1)
        uint32_t c = int(threadIdx.x) < parts ? tcnt[threadIdx.x] : 0u, wt;
        uint32_t ex = wave_exclusive_scan(c, wt);
        if (lane_id() == 63) wave_tot[threadIdx.x / 64] = wt;
        __syncthreads();
        if (int(threadIdx.x) < parts) {
            uint32_t pre = 0;
            for (int w = 0; w < int(threadIdx.x) / 64; ++w) pre += wave_tot[w];
            tstart[threadIdx.x] = pre + ex;
        }
        uint32_t tile_total = 0;
        for (int w = 0; w < PARTS / 64; ++w) tile_total += wave_tot[w];
        __syncthreads();
        SLAB_STAMP(1); // scan of the tile's counters
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (!pass[u]) continue;
            const uint32_t i = tstart[part[u]] + rank[u];
            if (TW == 2) {
                *reinterpret_cast<ulonglong2 *>(&stup[size_t(i) * 2]) = make_ulonglong2(key[u], r.vw[0][u]);
            } else {
                stup[size_t(i) * TW] = key[u];
#pragma unroll
                for (int j = 0; j < NVT; ++j) stup[size_t(i) * TW + 1 + j] = r.vw[j][u];
            }
        }
        if (base + SC_ROWS < hi) load(next, base + SC_ROWS); // workgroup-uniform: the next tile's words fly during the copy-out
        __syncthreads();
        SLAB_STAMP(2); // tuples to LDS, next tile's loads issued
        for (uint32_t i = threadIdx.x; i < tile_total; i += blockDim.x) {
            uint64_t k, v0 = 0, v1 = 0;
            if (TW == 2) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(&stup[size_t(i) * 2]);
                k = t.x;
                v0 = t.y;
            } else {
                k = stup[size_t(i) * TW];
                v0 = stup[size_t(i) * TW + 1];
                if (NVT > 1) v1 = stup[size_t(i) * TW + 2];
            }
            const uint32_t p = uint32_t((k * GOLD) >> (64 - parts_log2));
            const uint32_t at = gcur[p] + (i - tstart[p]);
            if (at < cap) {
                uint64_t *dst = sa.slabs + ((size_t(blockIdx.x) * size_t(parts) + p) * size_t(cap) + at) * TW;
                if (TW == 2) {
                    *reinterpret_cast<ulonglong2 *>(dst) = make_ulonglong2(k, v0);
                } else {
                    dst[0] = k;
                    dst[1] = v0;
                    if (NVT > 1) dst[2] = v1;
                }
            } else {
                atomicOr(&flags[NQE_FLAG_SLAB_OVERFLOW], 1); // skewed keys: the host redoes the query with exact partition sizes
            }
        }
        __syncthreads();
        SLAB_STAMP(3); // copy-out
        if (int(threadIdx.x) < parts) {
            gcur[threadIdx.x] += tcnt[threadIdx.x];
            tcnt[threadIdx.x] = 0;
        }
        __syncthreads();
        SLAB_STAMP(4); // cursors
    };
    if (lo < hi) {
        Regs A, B;
        load(A, lo);
        for (int64_t base = lo; base < hi; base += 2 * int64_t(SC_ROWS)) {
            tile(A, B, base);
            if (base + SC_ROWS >= hi) break;
            tile(B, A, base + SC_ROWS);
        }
    }
    for (int p = threadIdx.x; p < parts; p += blockDim.x) sa.fill[size_t(p) * size_t(sa.W) + blockIdx.x] = gcur[p] < cap ? gcur[p] : cap;
#ifdef NQE_SLAB_PROFILE
    if (threadIdx.x == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&nqe_slab_prof[i], prof_acc[i]);
        atomicAdd(&nqe_slab_prof[7], 1ull);
    }
#endif
}

// ------------------------------------------------------------------ the K32 scatter with WHOLE-LINE stores (round 5)
// tools/scatter_bench.hip: what the 12-byte-record scatter above pays for is not its 256 streams but the shape of its stores — runs
// of ~32 records that start and end anywhere, plain stores (a non-temporal store of a PARTIAL line is far worse: 1.1 ms), 0.70-0.83 ms
// per 10^8 rows at 256 partitions where separate value / key streams written in whole aligned lines with non-temporal stores take
// 0.57.  So the slab of (workgroup, partition) is two arrays — cap 8-byte values, then (behind every slab's values) cap 4-byte keys —
// and a slab only ever receives whole BLOCKS of 16 tuples (8 with 512 partitions): one 128-byte line of values + 64 bytes of keys per
// block, written by 16 consecutive lanes.  What a tile leaves over per partition (< 16 tuples) waits in an LDS carry buffer and leads
// the partition's next block.  Per tile: rank the rows per partition (LDS atomic), scan the counts and the block counts, stage the
// tuples by partition, copy out whole blocks — a block's tuples come from the carry buffer first, then from the stage — move the
// leftovers to the carry buffer, advance the cursors.  The last, partial block of every partition is written when the chunk ends.
// (Run length no longer matters — every store is a whole block — so neither does the tile size; the tile stays 8 rows per thread where
// the registers allow two tiles in flight.)  Key as stored: key - range_min under key-range partitions, else the key's low 32 bits
// (a key outside int32 raises NQE_FLAG_KEY32_OVERFLOW as before).
#ifndef NQE_SOA_RPT
#define NQE_SOA_RPT 4 // rows per thread per tile of the SoA scatter (every store is a whole block: the tile size no longer decides the store shape)
#endif
// THREADS: 512 (two workgroups per CU: their barrier phases overlap, as the staged compaction's do) or 1024.  The LDS layout follows the
// partition count of the run: stage [THREADS x RPT], carry [parts x block], five counters per partition, the block owner map.
template <int PRED, int KEY, int THREADS>
__global__ void __launch_bounds__(THREADS) agg_slab_scatter_soa_kernel(AggArgs a, FastPred fp, SlabArgs sa, int *flags) {
    constexpr int RPT = NQE_SOA_RPT;
    constexpr int SC_ROWS = THREADS * RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int parts_log2 = sa.parts_log2, parts = 1 << parts_log2;
    const int blk_log2 = parts_log2 <= 8 ? 4 : 3, blk = 1 << blk_log2;
    const int carry = parts << blk_log2;
    uint64_t *sval = reinterpret_cast<uint64_t *>(smem);      // [SC_ROWS] the tile's values by partition
    uint64_t *cval = sval + SC_ROWS;                          // [carry]
    uint32_t *skey = reinterpret_cast<uint32_t *>(cval + carry); // [SC_ROWS]
    uint32_t *ckey = skey + SC_ROWS;                          // [carry]
    uint32_t *gblk = ckey + carry;                            // [parts] blocks this workgroup has written per partition
    uint32_t *ccnt = gblk + parts;                            // [parts] tuples in the carry buffer
    uint32_t *tcnt = ccnt + parts;                            // [parts] tuples of this tile
    uint32_t *tstart = tcnt + parts;                          // [parts] tile-local exclusive scan of tcnt
    uint32_t *bstart = tstart + parts;                        // [parts] exclusive scan of the blocks this tile completes
    uint16_t *bown = reinterpret_cast<uint16_t *>(bstart + parts); // [SC_ROWS / 8 + parts] partition of each such block
    __shared__ uint32_t wave_tot[2][THREADS / 64];
    const bool range_part = sa.range_span != 0;
    for (int p = threadIdx.x; p < parts; p += blockDim.x) gblk[p] = ccnt[p] = tcnt[p] = 0;
    __syncthreads();
    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(PRED >= 2 ? a.pred_src.values : a.key_src.values);
    const uint64_t *__restrict__ valp = static_cast<const uint64_t *>(a.val[0].values);
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const OpAux key_aux = a.key.aux[0];
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;
    const int64_t lo = int64_t(blockIdx.x) * sa.chunk;
    const int64_t hi = lo + sa.chunk < a.n ? lo + sa.chunk : a.n;
    const int64_t last = a.n - 1;
    const uint32_t cap = uint32_t(sa.cap);
    // this workgroup's slabs: values of (w, p) at vbase + p * cap, keys at kbase + p * cap
    uint64_t *__restrict__ vbase = sa.slabs + size_t(blockIdx.x) * size_t(parts) * size_t(cap);
    uint32_t *__restrict__ kbase = reinterpret_cast<uint32_t *>(sa.slabs + size_t(sa.W) * size_t(parts) * size_t(cap)) + size_t(blockIdx.x) * size_t(parts) * size_t(cap);
    struct Regs {
        uint64_t kw[RPT], pw[PRED >= 2 ? RPT : 1], vw[RPT];
    };
    auto load = [&](Regs &r, int64_t base) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            int64_t row = base + int64_t(u) * THREADS + threadIdx.x;
            row = row < last ? row : last; // clamp: unconditional, in-bounds
            r.kw[u] = __builtin_nontemporal_load(&keyp[row]);
            if (PRED == 2) r.pw[PRED >= 2 ? u : 0] = __builtin_nontemporal_load(&predp[row >> fp.row_shift]);
            if (PRED == 3) r.pw[PRED >= 2 ? u : 0] = __builtin_nontemporal_load(&predp[row]);
            r.vw[u] = __builtin_nontemporal_load(&valp[row]);
        }
    };
    // tuple j of partition p's pending sequence: the carried tuples first, then the tile's
    // (one index into sval / skey — the carry buffers lie right behind the stages: a selected POINTER sends the pointers to scratch memory)
    auto pending = [&](uint32_t p, uint32_t j, uint32_t cc, uint64_t &v, uint32_t &k) {
        const uint32_t i = j < cc ? uint32_t(SC_ROWS) + (p << blk_log2) + j : tstart[p] + (j - cc);
        v = sval[i];
        k = skey[i];
    };
    auto tile = [&](const Regs &r, Regs &next, int64_t base) {
        uint32_t k32[RPT], part[RPT], rank[RPT];
        bool pass[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = base + int64_t(u) * THREADS + threadIdx.x;
            bool ok = row < hi;
            if (PRED == 3) ok = ok && eval_simple<false>(a.pred, r.pw[PRED >= 2 ? u : 0], false, nullptr) != 0; // host-vetted chain: cannot fault
            else if (PRED == 2) ok = ok && range_pass(fp, pred_extract(fp, r.pw[PRED >= 2 ? u : 0], row < last ? row : last));
            else if (PRED == 1) ok = ok && range_pass(fp, r.kw[u]);
            const uint64_t key = inline_key<KEY>(a.key, r.kw[u], key_mask, key_aux, key_signed);
            if (range_part) { // (wave-uniform choice) a key outside the range: the host redoes the query hashed
                const uint64_t d = key - uint64_t(sa.range_min);
                if (ok && d >= sa.range_span) {
                    atomicOr(&flags[NQE_FLAG_OOB], 1);
                    ok = false;
                }
                part[u] = range_partition(d, parts_log2);
                k32[u] = uint32_t(d);
            } else {
                part[u] = uint32_t((key * GOLD) >> (64 - parts_log2));
                k32[u] = uint32_t(key);
                if (ok && int64_t(int32_t(uint32_t(key))) != int64_t(key)) atomicOr(&flags[NQE_FLAG_KEY32_OVERFLOW], 1);
            }
            pass[u] = ok;
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) rank[u] = pass[u] ? atomicAdd(&tcnt[part[u]], 1u) : 0u;
        __syncthreads();
        // threads 0..parts-1: exclusive scans of the tile's counts and of the blocks the tile completes
        uint32_t c = 0, nb = 0;
        if (int(threadIdx.x) < parts) {
            c = tcnt[threadIdx.x];
            nb = (ccnt[threadIdx.x] + c) >> blk_log2;
        }
        uint32_t wt0, wt1;
        const uint32_t ex0 = wave_exclusive_scan(c, wt0), ex1 = wave_exclusive_scan(nb, wt1);
        if (lane_id() == 63) {
            wave_tot[0][threadIdx.x / 64] = wt0;
            wave_tot[1][threadIdx.x / 64] = wt1;
        }
        __syncthreads();
        uint32_t nblocks = 0;
        {
            uint32_t pre0 = 0, pre1 = 0;
            for (int w = 0; w < THREADS / 64; ++w) {
                if (w < int(threadIdx.x) / 64) {
                    pre0 += wave_tot[0][w];
                    pre1 += wave_tot[1][w];
                }
                nblocks += wave_tot[1][w];
            }
            if (int(threadIdx.x) < parts) {
                tstart[threadIdx.x] = pre0 + ex0;
                bstart[threadIdx.x] = pre1 + ex1;
                for (uint32_t b = 0; b < nb; ++b) bown[pre1 + ex1 + b] = uint16_t(threadIdx.x);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (!pass[u]) continue;
            const uint32_t i = tstart[part[u]] + rank[u];
            sval[i] = r.vw[u];
            skey[i] = k32[u];
        }
        if (base + SC_ROWS < hi) load(next, base + SC_ROWS); // workgroup-uniform: the next tile's words fly during the copy-out
        __syncthreads();
        // whole blocks: 2^blk_log2 consecutive lanes write one block — a 128-byte line of values and 64 bytes of keys
        for (uint32_t t = threadIdx.x; t < (nblocks << blk_log2); t += blockDim.x) {
            const uint32_t b = t >> blk_log2, l = t & uint32_t(blk - 1), p = bown[b];
            const uint32_t j = ((b - bstart[p]) << blk_log2) + l, at = (gblk[p] << blk_log2) + j;
            uint64_t v;
            uint32_t k;
            pending(p, j, ccnt[p], v, k);
            if (at < cap) {
                __builtin_nontemporal_store(v, vbase + size_t(p) * cap + at);
                __builtin_nontemporal_store(k, kbase + size_t(p) * cap + at);
            } else
                atomicOr(&flags[NQE_FLAG_SLAB_OVERFLOW], 1); // skewed keys: the host redoes the query with exact partition sizes
        }
        __syncthreads();
        // leftovers to the carry buffer (a partition that completed a block consumed its carry: the leftovers are the tile's own)
        for (uint32_t t = threadIdx.x; t < uint32_t(parts << blk_log2); t += blockDim.x) {
            const uint32_t p = t >> blk_log2, l = t & uint32_t(blk - 1);
            const uint32_t cc = ccnt[p], tot = cc + tcnt[p], nbp = tot >> blk_log2, rem = tot & uint32_t(blk - 1);
            const uint32_t j = (nbp << blk_log2) + l;
            if (l < rem && j >= cc) {
                uint64_t v;
                uint32_t k;
                pending(p, j, cc, v, k);
                cval[(p << blk_log2) + l] = v;
                ckey[(p << blk_log2) + l] = k;
            }
        }
        __syncthreads();
        if (int(threadIdx.x) < parts) {
            const uint32_t tot = ccnt[threadIdx.x] + tcnt[threadIdx.x];
            gblk[threadIdx.x] += tot >> blk_log2;
            ccnt[threadIdx.x] = tot & uint32_t(blk - 1);
            tcnt[threadIdx.x] = 0;
        }
        __syncthreads();
    };
    if (lo < hi) {
        Regs A, B;
        load(A, lo);
        for (int64_t base = lo; base < hi; base += 2 * int64_t(SC_ROWS)) {
            tile(A, B, base);
            if (base + SC_ROWS >= hi) break;
            tile(B, A, base + SC_ROWS);
        }
    }
    // the last, partial block of every partition; the fill counts
    for (uint32_t t = threadIdx.x; t < uint32_t(parts << blk_log2); t += blockDim.x) {
        const uint32_t p = t >> blk_log2, l = t & uint32_t(blk - 1), at = (gblk[p] << blk_log2) + l;
        if (l < ccnt[p]) {
            if (at < cap) {
                vbase[size_t(p) * cap + at] = cval[(p << blk_log2) + l];
                kbase[size_t(p) * cap + at] = ckey[(p << blk_log2) + l];
            } else
                atomicOr(&flags[NQE_FLAG_SLAB_OVERFLOW], 1);
        }
    }
    for (int p = threadIdx.x; p < parts; p += blockDim.x) {
        const uint32_t n = (gblk[p] << blk_log2) + ccnt[p];
        sa.fill[size_t(p) * size_t(sa.W) + blockIdx.x] = n < cap ? n : cap;
    }
}

// one workgroup per partition (grid-stride); its waves take the partition's slabs round-robin and stream their tuples, four
// per lane per step, into the workgroup's LDS table with the batched update of the fast kernel (all first probes, then all
// min/max reads of the step in flight together — tuples of a partition arrive in no particular order, every row is an update).
template <int NVT, bool VF64, bool K32 = false>
__global__ void __launch_bounds__(AGG_BLOCK) agg_slab_segments_kernel(AggArgs a, SlabArgs sa, GroupTable g, int *flags) {
    constexpr int TW = 1 + NVT;
    constexpr int SU = 4; // tuples per lane per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t cap = uint32_t(a.lds_cap);
    const uint32_t slots = cap + 1;
    uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
    double *lsum = reinterpret_cast<double *>(lkeys + slots);
    uint64_t *lmn = reinterpret_cast<uint64_t *>(lsum + NVT * slots);
    uint64_t *lmx = lmn + NVT * slots;
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(lmx + NVT * slots);
    const uint64_t ORD_MAX = f64_to_ord(DBL_MAX), ORD_MIN = f64_to_ord(-DBL_MAX);
    int vdt[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) vdt[j] = a.val[j].dtype;
    // read and written as an LDS word (ds_read / ds_write): through a generic `volatile int *` the accesses were FLAT loads, each
    // followed by s_waitcnt vmcnt(0) — on every first-probe miss the wave waited for the tuples it had just prefetched
    __shared__ int seg_full_flag;
    const int wave = int(threadIdx.x) / 64, nwaves = AGG_BLOCK / 64;
    const int parts_log2 = sa.parts_log2, parts = 1 << parts_log2;
    for (int p = blockIdx.x; p < parts; p += gridDim.x) {
        __syncthreads();
        if (__hip_atomic_load(&flags[NQE_FLAG_NEED_LEVEL2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (threadIdx.x == 0) seg_full_flag = 0;
        for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
            lkeys[s] = EMPTY_KEY;
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                lsum[j * slots + s] = 0.0;
                lmn[j * slots + s] = ORD_MAX;
                lmx[j * slots + s] = ORD_MIN;
                lcnt[j * slots + s] = 0;
            }
        }
        __syncthreads();
        // The wave's slabs w = wave, wave + 16, ... as ONE flat sequence of 256-tuple steps, the next step's tuples requested
        // before the current step is processed (a slab is ~750 tuples = 3 steps; walking the slabs one at a time exposed the
        // fill-count load and the first tuple load of every slab: 16 dependent round trips per wave per partition).
        const int nl = (sa.W - wave + nwaves - 1) / nwaves; // slabs of this wave (<= 64: W <= 1024)
        const uint32_t myfill = lane_id() < nl ? sa.fill[size_t(p) * size_t(sa.W) + size_t(wave + lane_id() * nwaves)] : 0u;
        const uint64_t *__restrict__ pbase = sa.slabs + size_t(p) * size_t(sa.cap) * TW; // slab (w, p) = pbase + w * PARTS * cap tuples
        struct Step {
            uint64_t key[SU], vw[NVT][SU];
            int32_t k32[K32 ? SU : 1]; // the 12-byte tuple's key AS LOADED: widening it here would make fetch() wait for its own loads
            bool live[SU];
        };
        int cl = 0;          // current slab (index into the wave's list), wave-uniform
        uint32_t ci0 = 0;    // first tuple of the current step
        auto seek = [&](int &l, uint32_t &i0) { // first position at or after (l, i0) that holds tuples; l == nl: none
            while (l < nl && i0 >= uint32_t(__builtin_amdgcn_readlane(int(myfill), l))) {
                ++l;
                i0 = 0;
            }
        };
        // `on` false: a dummy step (every tuple dead) over a position known to hold tuples — the prefetch of the step past the last is
        // issued unconditionally, because a conditional fetch makes the step's registers a phi and the copies that resolve it sit
        // right behind the loads (s_waitcnt vmcnt(0) before the CURRENT step is processed: no prefetch at all; seen in the ISA)
        auto fetch = [&](Step &st, int l, uint32_t i0, bool on) {
            const uint32_t f = uint32_t(__builtin_amdgcn_readlane(int(myfill), l));
            const uint64_t *__restrict__ slab = pbase + size_t(wave + l * nwaves) * size_t(parts) * size_t(sa.cap) * TW;
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const uint32_t i = i0 + uint32_t(u) * 64 + uint32_t(lane_id());
                st.live[u] = on && i < f;
                const uint32_t ic = st.live[u] ? i : f - 1;
                if (K32) { // (two arrays: agg_slab_scatter_soa_kernel)
                    const size_t at = (size_t(wave + l * nwaves) * size_t(parts) + size_t(p)) * size_t(sa.cap) + ic;
                    st.k32[K32 ? u : 0] = int32_t(__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(sa.slabs + size_t(sa.W) * size_t(parts) * size_t(sa.cap)) + at));
                    st.vw[0][u] = __builtin_nontemporal_load(sa.slabs + at);
                } else if (TW == 2) {
                    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
                    const v2u64 t = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(&slab[size_t(ic) * 2]));
                    st.key[u] = t.x;
                    st.vw[0][u] = t.y;
                } else {
                    st.key[u] = slab[size_t(ic) * TW];
#pragma unroll
                    for (int j = 0; j < NVT; ++j) st.vw[j][u] = slab[size_t(ic) * TW + 1 + j];
                }
            }
        };
        auto update = [&](const Step &st) {
            uint64_t skey[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) skey[u] = K32 ? uint64_t(int64_t(st.k32[K32 ? u : 0])) : st.key[u];
            // ---- slots: every first probe issued before any is examined
            uint32_t s0[SU];
            uint64_t k0[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                s0[u] = uint32_t(((skey[u] * GOLD) << parts_log2) >> a.lds_shift);
                k0[u] = lkeys[s0[u]];
            }
            int slot[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                slot[u] = -1;
                if (!st.live[u]) continue;
                const uint64_t key = skey[u];
                if (key == EMPTY_KEY) {
                    lkeys[cap] = 0;
                    slot[u] = int(cap);
                } else if (k0[u] == key) {
                    slot[u] = int(s0[u]);
                } else if (!SEG_FULL()) {
                    uint32_t sl = s0[u];
                    for (int probe = 0; probe < 32; ++probe) {
                        uint64_t k = lkeys[sl];
                        if (k == key) { slot[u] = int(sl); break; }
                        if (k == EMPTY_KEY) {
                            uint64_t old = atomicCAS((unsigned long long *)&lkeys[sl], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                            if (old == EMPTY_KEY || old == key) { slot[u] = int(sl); break; }
                        }
                        sl = (sl + 1) & (cap - 1);
                    }
                }
                if (slot[u] < 0 && !SEG_FULL()) { // more distinct keys than the table: the host partitions one level deeper (exact form)
                    SEG_FULL_SET();
                    atomicOr(&flags[NQE_FLAG_NEED_LEVEL2], 1);
                }
            }
            // ---- read-before-atomic, the step's rows in flight together
            uint64_t cmn[NVT][SU], cmx[NVT][SU];
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const uint32_t o = uint32_t(j) * slots + uint32_t(slot[u] < 0 ? 0 : slot[u]);
                    cmn[j][u] = lmn[o];
                    cmx[j][u] = lmx[o];
                }
            }
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    if (slot[u] < 0) continue;
                    const double x = VF64 ? u2d(st.vw[j][u]) : word_as_f64(st.vw[j][u], vdt[j]);
                    const bool isn = x != x;
                    const uint64_t xo = f64_to_ord(x);
                    const uint32_t o = uint32_t(j) * slots + uint32_t(slot[u]);
                    atomicAdd(&lcnt[o], 1u);
                    unsafeAtomicAdd(&lsum[o], x);
                    if (isn) atomicOr(&lcnt[o], NAN_BIT);
                    else {
                        if (xo < cmn[j][u]) atomicMin((unsigned long long *)&lmn[o], (unsigned long long)xo);
                        if (xo > cmx[j][u]) atomicMax((unsigned long long *)&lmx[o], (unsigned long long)xo);
                    }
                }
            }
        };
        seek(cl, ci0);
        if (cl < nl) {
            const int fl = cl;
            const uint32_t fi0 = ci0;
            Step A, B;
            fetch(A, cl, ci0, true);
            for (;;) {
                int nlx = cl;
                uint32_t ni0 = ci0 + 64 * SU;
                seek(nlx, ni0);
                const bool more_b = nlx < nl;
                fetch(B, more_b ? nlx : fl, more_b ? ni0 : fi0, more_b);
                update(A);
                if (!more_b) break;
                cl = nlx;
                ci0 = ni0 + 64 * SU;
                seek(cl, ci0);
                const bool more_a = cl < nl;
                fetch(A, more_a ? cl : fl, more_a ? ci0 : fi0, more_a);
                update(B);
                if (!more_a) break;
            }
        }
        __syncthreads();
        if (SEG_FULL()) break; // result discarded
        if (g.dense_count) {
            // ---- dense output: count this partition's groups, reserve [base, base + n) with one atomic, write them there
            __shared__ uint32_t wave_tot[AGG_BLOCK / 64];
            __shared__ uint32_t dense_base;
            uint32_t mine = 0;
            for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) mine += lkeys[s] != EMPTY_KEY ? 1u : 0u;
            uint32_t wtot;
            const uint32_t wexcl = wave_exclusive_scan(mine, wtot);
            if (lane_id() == 0) wave_tot[threadIdx.x / 64] = wtot;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tot = 0;
                for (int w = 0; w < AGG_BLOCK / 64; ++w) { uint32_t c = wave_tot[w]; wave_tot[w] = tot; tot += c; }
                dense_base = tot ? atomicAdd(g.dense_count, tot) : 0u;
            }
            __syncthreads();
            uint32_t pos = dense_base + wave_tot[threadIdx.x / 64] + wexcl;
            const size_t gstride = size_t(g.cap) + 1;
            for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
                uint64_t k = lkeys[s];
                if (k == EMPTY_KEY) continue;
                if (pos < g.cap) {
                    g.keys[pos] = (s == cap) ? EMPTY_KEY : k;
#pragma unroll
                    for (int j = 0; j < NVT; ++j) {
                        const uint32_t o = uint32_t(j) * slots + s;
                        const uint32_t c = lcnt[o];
                        const size_t go = size_t(a.v0 + j) * gstride + pos;
                        g.cnt[go] = uint64_t(c & ~NAN_BIT);
                        g.sum[go] = lsum[o];
                        g.mn[go] = lmn[o];
                        g.mx[go] = lmx[o];
                        g.nan[go] = (c & NAN_BIT) ? 1u : 0u;
                    }
                } else atomicOr(&flags[NQE_FLAG_DENSE_OVERFLOW], 1);
                ++pos;
            }
            continue;
        }
        for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
            uint64_t k = lkeys[s];
            if (k == EMPTY_KEY) continue;
            uint64_t key = (s == cap) ? EMPTY_KEY : k;
            int64_t gslot = global_find_or_insert(g, key, flags);
            if (gslot < 0) continue;
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                uint32_t o = uint32_t(j) * slots + s;
                uint32_t c = lcnt[o];
                global_update(g, gslot, a.v0 + j, uint64_t(c & ~NAN_BIT), lsum[o], true, lmn[o], lmx[o], true, (c & NAN_BIT) != 0);
            }
        }
    }
}

// Key-range partitions (SlabArgs::range_span != 0; one value column, 12-byte tuples): the table of partition p is addressed by
// (key - range_min) >> parts_log2 — no hash, no probe sequence, no key words, no overflow; the key of a slot is rebuilt from (p, slot);
// min / max are doubles behind ordered compares (native LDS f64 atomics).  The hashed kernel above spends ~75 instructions
// per tuple (64-bit multiply, probe, compare, ordered-integer min / max); this one a dozen.  Same tuple stream (a wave's slabs as one
// sequence of 256-tuple steps, the next step requested before the current one is processed), same dense output.
template <bool VF64>
__global__ void __launch_bounds__(AGG_BLOCK) agg_slab_segments_direct_kernel(AggArgs a, SlabArgs sa, GroupTable g, int *flags) {
#ifndef NQE_DIRECT_SU
#define NQE_DIRECT_SU 4 // (8 measured the same: 0.32-0.33 ms per 10^8 tuples either way — the kernel is bound by its LDS atomics, not by loads in flight)
#endif
    constexpr int SU = NQE_DIRECT_SU; // tuples per lane per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int parts_log2 = sa.parts_log2;
    const uint32_t W = uint32_t((sa.range_span + (uint64_t(1) << parts_log2) - 1) >> parts_log2); // slots per partition (<= 4096: the host checks)
    double *lsum = reinterpret_cast<double *>(smem);
    double *lmn = lsum + W;
    double *lmx = lmn + W;
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(lmx + W);
    const int vdt = a.val[0].dtype;
    const int wave = int(threadIdx.x) / 64, nwaves = AGG_BLOCK / 64;
    const int parts = 1 << parts_log2;
    for (int p = blockIdx.x; p < parts; p += gridDim.x) {
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < W; s += blockDim.x) {
            lsum[s] = 0.0;
            lmn[s] = DBL_MAX;
            lmx[s] = -DBL_MAX;
            lcnt[s] = 0;
        }
        __syncthreads();
        const int nl = (sa.W - wave + nwaves - 1) / nwaves; // slabs of this wave (<= 64: W <= 1024)
        const uint32_t myfill = lane_id() < nl ? sa.fill[size_t(p) * size_t(sa.W) + size_t(wave + lane_id() * nwaves)] : 0u;
        struct Step {
            uint64_t vw[SU];
            int32_t k32[SU]; // as loaded (widened where it is used: see agg_slab_segments_kernel)
            bool live[SU];
        };
        int cl = 0;       // current slab (index into the wave's list), wave-uniform
        uint32_t ci0 = 0; // first tuple of the current step
        auto seek = [&](int &l, uint32_t &i0) { // first position at or after (l, i0) that holds tuples; l == nl: none
            while (l < nl && i0 >= uint32_t(__builtin_amdgcn_readlane(int(myfill), l))) {
                ++l;
                i0 = 0;
            }
        };
        auto fetch = [&](Step &st, int l, uint32_t i0, bool on) { // (`on` false: a dummy step over a position known to hold tuples)
            const uint32_t f = uint32_t(__builtin_amdgcn_readlane(int(myfill), l));
            const size_t sbase = (size_t(wave + l * nwaves) * size_t(parts) + size_t(p)) * size_t(sa.cap);
            const uint64_t *__restrict__ svals = sa.slabs + sbase;
            const uint32_t *__restrict__ skeys = reinterpret_cast<const uint32_t *>(sa.slabs + size_t(sa.W) * size_t(parts) * size_t(sa.cap)) + sbase;
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const uint32_t i = i0 + uint32_t(u) * 64 + uint32_t(lane_id());
                st.live[u] = on && i < f;
                const uint32_t ic = i < f ? i : f - 1;
                st.k32[u] = int32_t(__builtin_nontemporal_load(skeys + ic));
                st.vw[u] = __builtin_nontemporal_load(svals + ic);
            }
        };
        auto update = [&](const Step &st) {
            uint32_t slot[SU];
            double x[SU], cmn[SU], cmx[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                slot[u] = st.live[u] ? uint32_t(st.k32[u]) >> parts_log2 : 0u; // the tuple holds key - range_min (< span: the scatter checked the range, so slot < W)
                x[u] = VF64 ? u2d(st.vw[u]) : word_as_f64(st.vw[u], vdt);
                cmn[u] = lmn[slot[u]];
                cmx[u] = lmx[slot[u]];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                if (!st.live[u]) continue;
                atomicAdd(&lcnt[slot[u]], 1u);
                unsafeAtomicAdd(&lsum[slot[u]], x[u]);
                if (x[u] != x[u]) atomicOr(&lcnt[slot[u]], NAN_BIT);
                else {
                    if (x[u] < cmn[u]) unsafeAtomicMin(&lmn[slot[u]], x[u]);
                    if (x[u] > cmx[u]) unsafeAtomicMax(&lmx[slot[u]], x[u]);
                }
            }
        };
        seek(cl, ci0);
        if (cl < nl) {
            const int fl = cl;
            const uint32_t fi0 = ci0;
            Step A, B;
            fetch(A, cl, ci0, true);
            for (;;) {
                int nlx = cl;
                uint32_t ni0 = ci0 + 64 * SU;
                seek(nlx, ni0);
                const bool more_b = nlx < nl;
                fetch(B, more_b ? nlx : fl, more_b ? ni0 : fi0, more_b);
                update(A);
                if (!more_b) break;
                cl = nlx;
                ci0 = ni0 + 64 * SU;
                seek(cl, ci0);
                const bool more_a = cl < nl;
                fetch(A, more_a ? cl : fl, more_a ? ci0 : fi0, more_a);
                update(B);
                if (!more_a) break;
            }
        }
        __syncthreads();
        // ---- dense output: count this partition's groups, reserve [base, base + n) with one atomic, write them there
        __shared__ uint32_t wave_tot[AGG_BLOCK / 64];
        __shared__ uint32_t dense_base;
        uint32_t mine = 0;
        for (uint32_t s = threadIdx.x; s < W; s += blockDim.x) mine += lcnt[s] != 0 ? 1u : 0u;
        uint32_t wtot;
        const uint32_t wexcl = wave_exclusive_scan(mine, wtot);
        if (lane_id() == 0) wave_tot[threadIdx.x / 64] = wtot;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < AGG_BLOCK / 64; ++w) {
                const uint32_t c = wave_tot[w];
                wave_tot[w] = tot;
                tot += c;
            }
            dense_base = tot ? atomicAdd(g.dense_count, tot) : 0u;
        }
        __syncthreads();
        uint32_t pos = dense_base + wave_tot[threadIdx.x / 64] + wexcl;
        const size_t gstride = size_t(g.cap) + 1;
        for (uint32_t s = threadIdx.x; s < W; s += blockDim.x) {
            const uint32_t c = lcnt[s];
            if (c == 0) continue;
            if (pos < g.cap) {
                const size_t go = size_t(a.v0) * gstride + pos;
                g.keys[pos] = uint64_t(sa.range_min + int64_t((uint64_t(s) << parts_log2) | uint64_t((uint32_t(p) ^ range_scramble(s, parts_log2)) & uint32_t(parts - 1))));
                g.cnt[go] = uint64_t(c & ~NAN_BIT);
                g.sum[go] = lsum[s];
                g.mn[go] = f64_to_ord(lmn[s]);
                g.mx[go] = f64_to_ord(lmx[s]);
                g.nan[go] = (c & NAN_BIT) ? 1u : 0u;
            } else atomicOr(&flags[NQE_FLAG_DENSE_OVERFLOW], 1);
            ++pos;
        }
    }
}

// The range tier's second kernel (aggregate_common.hpp: RangeRec): workgroup (p, q) = blockIdx / Q, blockIdx % Q streams the slabs
// w = q, q + Q, q + 2Q, ... of partition p into its LDS table (agg_slab_segments_direct_kernel's tuple stream and update) and writes
// the whole table to tab[(p * Q + q) * W + slot].
template <bool VF64>
__global__ void __launch_bounds__(AGG_BLOCK) agg_range_segments_kernel(AggArgs a, SlabArgs sa, int Q, RangeRec *__restrict__ tab) {
    constexpr int SU = NQE_DIRECT_SU; // tuples per lane per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int parts_log2 = sa.parts_log2;
    const uint32_t W = uint32_t((sa.range_span + (uint64_t(1) << parts_log2) - 1) >> parts_log2); // slots per partition (<= 4096: the host checks)
    double *lsum = reinterpret_cast<double *>(smem);
    double *lmn = lsum + W;
    double *lmx = lmn + W;
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(lmx + W);
    const int vdt = a.val[0].dtype;
    const int wave = int(threadIdx.x) / 64, nwaves = AGG_BLOCK / 64;
    const int parts = 1 << parts_log2;
    const int p = int(blockIdx.x) / Q, q = int(blockIdx.x) % Q;
    for (uint32_t s = threadIdx.x; s < W; s += blockDim.x) {
        lsum[s] = 0.0;
        lmn[s] = DBL_MAX;
        lmx[s] = -DBL_MAX;
        lcnt[s] = 0;
    }
    __syncthreads();
    const int nq = (sa.W - q + Q - 1) / Q;               // slabs of this workgroup: w = q + Q * j, j < nq
    const int nl = nq > wave ? (nq - wave + nwaves - 1) / nwaves : 0; // ... of this wave: j = wave + l * nwaves (<= 64: W <= 1024)
    auto slab_of = [&](int l) { return q + Q * (wave + l * nwaves); };
    const uint32_t myfill = lane_id() < nl ? sa.fill[size_t(p) * size_t(sa.W) + size_t(slab_of(lane_id()))] : 0u;
    struct Step {
        uint64_t vw[SU];
        int32_t k32[SU];
        bool live[SU];
    };
    int cl = 0;
    uint32_t ci0 = 0;
    auto seek = [&](int &l, uint32_t &i0) {
        while (l < nl && i0 >= uint32_t(__builtin_amdgcn_readlane(int(myfill), l))) {
            ++l;
            i0 = 0;
        }
    };
    auto fetch = [&](Step &st, int l, uint32_t i0, bool on) { // (`on` false: a dummy step over a position known to hold tuples)
        const uint32_t f = uint32_t(__builtin_amdgcn_readlane(int(myfill), l));
        const size_t sbase = (size_t(slab_of(l)) * size_t(parts) + size_t(p)) * size_t(sa.cap);
        const uint64_t *__restrict__ svals = sa.slabs + sbase;
        const uint32_t *__restrict__ skeys = reinterpret_cast<const uint32_t *>(sa.slabs + size_t(sa.W) * size_t(parts) * size_t(sa.cap)) + sbase;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const uint32_t i = i0 + uint32_t(u) * 64 + uint32_t(lane_id());
            st.live[u] = on && i < f;
            const uint32_t ic = i < f ? i : f - 1;
            st.k32[u] = int32_t(__builtin_nontemporal_load(skeys + ic));
            st.vw[u] = __builtin_nontemporal_load(svals + ic);
        }
    };
    auto update = [&](const Step &st) {
        uint32_t slot[SU];
        double x[SU], cmn[SU], cmx[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            slot[u] = st.live[u] ? uint32_t(st.k32[u]) >> parts_log2 : 0u; // the tuple holds key - range_min (< span: the scatter checked the range, so slot < W)
            x[u] = VF64 ? u2d(st.vw[u]) : word_as_f64(st.vw[u], vdt);
            cmn[u] = lmn[slot[u]];
            cmx[u] = lmx[slot[u]];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (!st.live[u]) continue;
            atomicAdd(&lcnt[slot[u]], 1u);
            unsafeAtomicAdd(&lsum[slot[u]], x[u]);
            if (x[u] != x[u]) atomicOr(&lcnt[slot[u]], NAN_BIT);
            else {
                if (x[u] < cmn[u]) unsafeAtomicMin(&lmn[slot[u]], x[u]);
                if (x[u] > cmx[u]) unsafeAtomicMax(&lmx[slot[u]], x[u]);
            }
        }
    };
    seek(cl, ci0);
    if (cl < nl) {
        const int fl = cl;
        const uint32_t fi0 = ci0;
        Step A, B;
        fetch(A, cl, ci0, true);
        for (;;) {
            int nlx = cl;
            uint32_t ni0 = ci0 + 64 * SU;
            seek(nlx, ni0);
            const bool more_b = nlx < nl;
            fetch(B, more_b ? nlx : fl, more_b ? ni0 : fi0, more_b);
            update(A);
            if (!more_b) break;
            cl = nlx;
            ci0 = ni0 + 64 * SU;
            seek(cl, ci0);
            const bool more_a = cl < nl;
            fetch(A, more_a ? cl : fl, more_a ? ci0 : fi0, more_a);
            update(B);
            if (!more_a) break;
        }
    }
    __syncthreads();
    RangeRec *__restrict__ mine = tab + size_t(blockIdx.x) * size_t(W);
    for (uint32_t s = threadIdx.x; s < W; s += blockDim.x) {
        RangeRec r;
        r.sum = lsum[s];
        r.mn = lmn[s];
        r.mx = lmx[s];
        r.cnt = uint64_t(lcnt[s] & ~NAN_BIT) | ((lcnt[s] & NAN_BIT) ? NAN_BIT64 : 0ull); // (one workgroup's rows of a pass: below 2^31)
        mine[s] = r;
    }
}

template <int PRED, int KEY> SlabScatterKernel pick_slab_scatter_nv(int nv, bool k32, int soa_threads) {
    if (nv == 1 && k32) return agg_slab_scatter_soa_kernel<PRED, KEY, 512>; // (soa_threads: AggSwitches — 512, two workgroups per CU)
    if (nv == 1) return agg_slab_scatter_kernel<PRED, KEY, 1>;
    return agg_slab_scatter_kernel<PRED, KEY, 2>;
}
template <int PRED> SlabScatterKernel pick_slab_scatter_key(int key, int nv, bool k32, int soa_threads) {
    switch (key) {
    case 0: return pick_slab_scatter_nv<PRED, 0>(nv, k32, soa_threads);
    case 1: return pick_slab_scatter_nv<PRED, 1>(nv, k32, soa_threads);
    case 2: return pick_slab_scatter_nv<PRED, 2>(nv, k32, soa_threads);
    default: return pick_slab_scatter_nv<PRED, 3>(nv, k32, soa_threads);
    }
}

template <int PRED, int KEY> PartKernel pick_scatter_nv(int nv) {
    return nv == 1 ? agg_partition_scatter_kernel<PRED, KEY, 1> : agg_partition_scatter_kernel<PRED, KEY, 2>;
}
template <int PRED> PartKernel pick_scatter_key(int key, int nv) {
    switch (key) {
    case 0: return pick_scatter_nv<PRED, 0>(nv);
    case 1: return pick_scatter_nv<PRED, 1>(nv);
    case 2: return pick_scatter_nv<PRED, 2>(nv);
    default: return pick_scatter_nv<PRED, 3>(nv);
    }
}
template <int PRED, int KEY, int NVT> PartKernel pick_part_sc(bool scatter) {
    return scatter ? agg_partition_kernel<PRED, KEY, NVT, true> : agg_partition_kernel<PRED, KEY, NVT, false>;
}
template <int PRED, int KEY> PartKernel pick_part_nv(int nv, bool scatter) {
    return nv == 1 ? pick_part_sc<PRED, KEY, 1>(scatter) : pick_part_sc<PRED, KEY, 2>(scatter);
}
template <int PRED> PartKernel pick_part_key(int key, int nv, bool scatter) {
    switch (key) {
    case 0: return pick_part_nv<PRED, 0>(nv, scatter);
    case 1: return pick_part_nv<PRED, 1>(nv, scatter);
    case 2: return pick_part_nv<PRED, 2>(nv, scatter);
    default: return pick_part_nv<PRED, 3>(nv, scatter);
    }
}

} // namespace

PartKernel pick_scatter_kernel(int pred, int key, int nv) {
    switch (pred) {
    case 0: return pick_scatter_key<0>(key, nv);
    case 1: return pick_scatter_key<1>(key, nv);
    case 2: return pick_scatter_key<2>(key, nv);
    default: return pick_scatter_key<3>(key, nv);
    }
}
PartKernel pick_part_kernel(int pred, int key, int nv, bool scatter) {
    switch (pred) {
    case 0: return pick_part_key<0>(key, nv, scatter);
    case 1: return pick_part_key<1>(key, nv, scatter);
    case 2: return pick_part_key<2>(key, nv, scatter);
    default: return pick_part_key<3>(key, nv, scatter);
    }
}
SlabScatterKernel pick_slab_scatter_kernel(int pred, int key, int nv, bool k32, int soa_threads) {
    switch (pred) {
    case 0: return pick_slab_scatter_key<0>(key, nv, k32, soa_threads);
    case 1: return pick_slab_scatter_key<1>(key, nv, k32, soa_threads);
    case 2: return pick_slab_scatter_key<2>(key, nv, k32, soa_threads);
    default: return pick_slab_scatter_key<3>(key, nv, k32, soa_threads);
    }
}
int slab_scatter_soa_rows_per_thread() { return NQE_SOA_RPT; }
int slab_scatter_rows_per_thread(int pred, int key, int nv) { return NQE_SLAB_WG_PER_CU > 1 ? 4 : ((nv == 1 && pred <= 1 && (key == 0 || (NQE_SLAB_KEYMOD_RPT == 8 && key != 3))) ? 8 : 4); }
int slab_scatter_wg_per_cu() { return NQE_SLAB_WG_PER_CU; }
SlabSegmentsKernel pick_slab_segments_kernel(int nv, bool vf64, bool k32) {
    if (nv == 1 && k32) return vf64 ? agg_slab_segments_kernel<1, true, true> : agg_slab_segments_kernel<1, false, true>;
    return nv == 1 ? (vf64 ? agg_slab_segments_kernel<1, true> : agg_slab_segments_kernel<1, false>)
                   : (vf64 ? agg_slab_segments_kernel<2, true> : agg_slab_segments_kernel<2, false>);
}
RangeSegmentsKernel pick_range_segments_kernel(bool vf64) { return vf64 ? agg_range_segments_kernel<true> : agg_range_segments_kernel<false>; }
SlabSegmentsKernel pick_slab_segments_direct_kernel(bool vf64) { return vf64 ? agg_slab_segments_direct_kernel<true> : agg_slab_segments_direct_kernel<false>; }
SubpartitionKernel pick_subpartition_kernel(int nv) { return nv == 1 ? agg_subpartition_kernel<1> : agg_subpartition_kernel<2>; }
SegmentsKernel pick_segments_kernel(int nv, bool vf64) {
    return nv == 1 ? (vf64 ? agg_segments_kernel<1, true> : agg_segments_kernel<1, false>) : (vf64 ? agg_segments_kernel<2, true> : agg_segments_kernel<2, false>);
}

} // namespace agg
} // namespace nqe

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE((nqe::agg::agg_slab_segments_kernel<1, true, false>));

#ifdef NQE_SLAB_PROFILE
extern "C" void nqe_debug_slab_profile(unsigned long long *out) { // reads and clears the phase clocks
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(nqe::agg::nqe_slab_prof), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(nqe::agg::nqe_slab_prof), z, sizeof(z));
}
#endif
