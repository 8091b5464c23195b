// aggregate_tail.hip — the kernels of the hash aggregate behind its streaming pass (split off aggregate.hip in round 6; see
// aggregate_tail.hpp): table set-up, collect / finalize, the tails, the merges of exchanged partial states, key range / key sample,
// the folds of per-workgroup tables.  Reference: aggregate/mod.rs:113-222 (evaluate per group → one output row), avg.rs:121, max.rs:38-50.
#include <algorithm>
#include <cmath>
#include <cfloat>

#include "aggregate_tail.hpp"

namespace nqe {
namespace agg {

// ------------------------------------------------------------------ table init / collect / finalize
__global__ void table_init_kernel(GroupTable g, int mark_slot0) {
    size_t slots = size_t(g.cap) + 1;
    size_t total = slots * size_t(g.V);
    size_t stride = size_t(gridDim.x) * blockDim.x;
    const uint64_t ORD_MAX = f64_to_ord(DBL_MAX), ORD_MIN = f64_to_ord(-DBL_MAX);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (i < slots) g.keys[i] = (mark_slot0 && i == 0) ? 0ull : EMPTY_KEY;
        g.cnt[i] = 0;
        g.sum[i] = 0.0;
        g.mn[i] = ORD_MAX;
        g.mx[i] = ORD_MIN;
        g.nan[i] = 0;
    }
}

__global__ void __launch_bounds__(256) collect_kernel(GroupTable g, uint64_t *out_keys, uint32_t *out_slots, uint32_t *counter) {
    // One returning atomic per WORKGROUP: a single hot word sustains only ≈88 M returning atomics/s on MI355X
    // (one per wave-iteration cost 23.8 ms for a 2^27-slot table).  Pass 1 counts the occupied slots of the
    // workgroup's contiguous chunk, one atomicAdd reserves its output range, pass 2 writes (order is irrelevant:
    // the entries are sorted by key afterwards).
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t block_base;
    const size_t slots = size_t(g.cap) + 1;
    const size_t chunk = ((slots + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
    const size_t lo = size_t(blockIdx.x) * chunk, hi = lo + chunk < slots ? lo + chunk : slots;
    const int wv = threadIdx.x / 64;
    uint32_t mine = 0;
    for (size_t s0 = lo + size_t(wv) * 64; s0 < hi; s0 += 256) {
        size_t s = s0 + lane_id();
        bool used = s < hi && g.keys[s] != EMPTY_KEY;
        mine += uint32_t(__popcll(__ballot(used))); // wave-uniform
    }
    if (lane_id() == 0) wave_cnt[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t run = block_base;
    for (int w = 0; w < wv; ++w) run += wave_cnt[w];
    for (size_t s0 = lo + size_t(wv) * 64; s0 < hi; s0 += 256) {
        size_t s = s0 + lane_id();
        uint64_t k = s < hi ? g.keys[s] : EMPTY_KEY;
        bool used = k != EMPTY_KEY;
        uint64_t m = __ballot(used);
        if (used) {
            uint32_t idx = run + uint32_t(__popcll(m & lanemask_lt()));
            out_keys[idx] = (s == g.cap) ? EMPTY_KEY : k;
            out_slots[idx] = uint32_t(s);
        }
        run += uint32_t(__popcll(m));
    }
}


// output row r ← the state of table slot s.  The state of a value column is gathered once for all aggregates over it (count,
// sum, avg, min, max of one column are five outputs of ONE random access per array, not of five).
__device__ __forceinline__ void finalize_row(const GroupTable &g, uint32_t s, int64_t r, const FinalizeArgs &f) {
    const size_t slots = size_t(g.cap) + 1;
    if (f.partial) { // the raw state of every value column, once (five aggregates over one column exchange 4 words, not 20)
        for (int v = 0; v < f.nslots; ++v) {
            const size_t o = size_t(v) * slots + s;
            f.out[4 * v + 0][r] = g.cnt[o];
            f.out[4 * v + 1][r] = d2u(g.sum[o]);
            f.out[4 * v + 2][r] = d2u(ord_to_f64(g.mn[o]));
            f.out[4 * v + 3][r] = d2u(g.nan[o] ? __longlong_as_double(0x7FF8000000000000ll) : ord_to_f64(g.mx[o]));
        }
        return;
    }
    int cached = -1;
    uint64_t cnt = 0;
    double sum = 0, mn = 0, mx = 0;
    for (int i = 0; i < f.naggs; ++i) {
        if (f.vslot[i] != cached) {
            cached = f.vslot[i];
            size_t o = size_t(cached) * slots + s;
            cnt = g.cnt[o];
            sum = g.sum[o];
            mn = ord_to_f64(g.mn[o]);
            mx = g.nan[o] ? __longlong_as_double(0x7FF8000000000000ll) : ord_to_f64(g.mx[o]);
        }
        {
            uint64_t w;
            switch (f.func[i]) {
            case NQE_AGG_COUNT: w = cnt; break;                                   // count.rs:76
            case NQE_AGG_SUM: w = d2u(sum); break;                                // sum.rs:115
            case NQE_AGG_AVG: w = d2u(sum / double(uint32_t(cnt))); break;        // avg.rs:121 (cnt is u32)
            case NQE_AGG_MIN: w = d2u(mn); break;
            default: w = d2u(mx); break;
            }
            f.out[i][r] = w;
        }
    }
}

// N output rows of one thread: the state words of all N slots are requested before any is used (finalize_row N times in a row
// is N dependent round trips — the stores of one row may alias the loads of the next as far as the compiler can tell)
template <int N>
__device__ __forceinline__ void finalize_rows(const GroupTable &g, const uint32_t (&s)[N], const bool (&live)[N], const int64_t (&r)[N], const FinalizeArgs &f) {
    const size_t slots = size_t(g.cap) + 1;
    for (int v = 0; v < f.nslots; ++v) {
        uint64_t cnt[N], mnw[N], mxw[N];
        double sum[N];
        uint32_t nan[N];
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const size_t o = size_t(v) * slots + (live[u] ? s[u] : 0u);
            cnt[u] = g.cnt[o];
            sum[u] = g.sum[o];
            mnw[u] = g.mn[o];
            mxw[u] = g.mx[o];
            nan[u] = g.nan[o];
        }
#pragma unroll
        for (int u = 0; u < N; ++u) {
            if (!live[u]) continue;
            const double mn = ord_to_f64(mnw[u]), mx = nan[u] ? __longlong_as_double(0x7FF8000000000000ll) : ord_to_f64(mxw[u]);
            if (f.partial) {
                f.out[4 * v + 0][r[u]] = cnt[u];
                f.out[4 * v + 1][r[u]] = d2u(sum[u]);
                f.out[4 * v + 2][r[u]] = d2u(mn);
                f.out[4 * v + 3][r[u]] = d2u(mx);
                continue;
            }
            for (int i = 0; i < f.naggs; ++i) {
                if (f.vslot[i] != v) continue;
                uint64_t w;
                switch (f.func[i]) {
                case NQE_AGG_COUNT: w = cnt[u]; break;
                case NQE_AGG_SUM: w = d2u(sum[u]); break;
                case NQE_AGG_AVG: w = d2u(sum[u] / double(uint32_t(cnt[u]))); break; // avg.rs:121 (cnt is u32)
                case NQE_AGG_MIN: w = d2u(mn); break;
                default: w = d2u(mx); break;
                }
                f.out[i][r[u]] = w;
            }
        }
    }
}

__global__ void finalize_kernel(GroupTable g, const uint32_t *sorted_slots, int64_t G, FinalizeArgs f) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < G; r += stride)
        finalize_row(g, sorted_slots ? sorted_slots[r] : 0, r, f);
}

// ------------------------------------------------------------------ tail of a DENSELY written table whose keys lie in a compact range
// The partitioned path leaves its groups in slots [0, G) in no order.  Sorting them by key was a histogram, a host read of it,
// two or three counting passes and the finalize gather — twelve launches and three host waits, 0.2 ms of a 1.3 ms step at 65536
// groups, 0.44 of 1.6 ms at 2^20.  Keys of the shapes that reach this path are mostly compact — `col % m`, ids, dates — so:
//   dense_key_range_kernel   min / max of the keys in sort order, ahead of the flag read-back; its words and the group count
//                            travel to the host with the flags (ONE wait for the whole tail);
//   dense_rank_mark_kernel   pos[key − min] = slot + 1 over a zeroed array of the range's length;
//   dense_rank_emit_kernel   one pass over pos: the non-zero entries, counted across workgroups by a decoupled look-back, are the
//                            groups in key order — key and aggregates are written at their rank (finalize_row fused).
// A range wider than 8 G + 65536 entries keeps the radix sort.
__global__ void __launch_bounds__(256) dense_key_range_kernel(const uint64_t *__restrict__ keys, uint32_t cap, uint64_t flip, uint32_t *dense) {
    // dense: [0] group count; words [2,3] = max of ~ord(key), [4,5] = max of ord(key) as uint64 (zero-initialised with the counter).
    // One pair of atomics per WORKGROUP and few workgroups: same-address device atomics retire one at a time (a pair per wave of
    // 512 workgroups took 37 us for 90000 keys)
    __shared__ unsigned long long wlo[4], whi[4];
    const uint32_t G = dense[0] < cap ? dense[0] : cap;
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) {
        const uint64_t o = keys[i] ^ flip;
        lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t l2 = __shfl_xor((unsigned long long)lo, d), h2 = __shfl_xor((unsigned long long)hi, d);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane_id() == 0) {
        wlo[threadIdx.x / 64] = lo;
        whi[threadIdx.x / 64] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            lo = wlo[w] < lo ? wlo[w] : lo;
            hi = whi[w] > hi ? whi[w] : hi;
        }
        if (lo <= hi) {
            atomicMax(reinterpret_cast<unsigned long long *>(dense + 2), (unsigned long long)~lo);
            atomicMax(reinterpret_cast<unsigned long long *>(dense + 4), (unsigned long long)hi);
        }
    }
}

__global__ void dense_rank_mark_kernel(const uint64_t *__restrict__ keys, uint32_t G, uint64_t flip, uint64_t ordmin, uint32_t *pos) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) pos[(keys[i] ^ flip) - ordmin] = i + 1;
}

// consecutive entries of pos per thread (one 16-byte load; their groups' state is gathered together: finalize_rows).  One entry per
// thread was slower — 33 against 18 us for 90000 entries: four times the chunks for the look-back to walk
// status[0]: ticket counter; status[1 + c]: (value << 2) | 1 = chunk c's own count, | 2 = the count of chunks 0..c.  A chunk is taken
// by ticket, so every chunk below a waiting one has been started — the look-back cannot wait on a workgroup that is not running.
template <int DR_ITEMS>
__global__ void __launch_bounds__(DR_BLOCK) dense_rank_emit_kernel(GroupTable g, const uint32_t *__restrict__ pos, uint64_t span, uint64_t flip, uint64_t ordmin,
                                                                   unsigned long long *status, uint64_t *out_keys, FinalizeArgs f) {
    __shared__ uint32_t s_chunk;
    __shared__ uint32_t wcnt[DR_BLOCK / 64];
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_chunk = uint32_t(atomicAdd(&status[0], 1ull));
    __syncthreads();
    const uint64_t c = s_chunk;
    const uint64_t j0 = (c * DR_BLOCK + threadIdx.x) * DR_ITEMS;
    uint32_t p[DR_ITEMS];
    if (DR_ITEMS == 4 && j0 + DR_ITEMS <= span) { // (pos is allocated in whole chunks)
        const uint4 q = *reinterpret_cast<const uint4 *>(pos + j0);
        p[0] = q.x;
        p[DR_ITEMS > 1 ? 1 : 0] = q.y;
        p[DR_ITEMS > 2 ? 2 : 0] = q.z;
        p[DR_ITEMS > 3 ? 3 : 0] = q.w;
    } else {
#pragma unroll
        for (int u = 0; u < DR_ITEMS; ++u) p[u] = j0 + u < span ? pos[j0 + u] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < DR_ITEMS; ++u) mine += p[u] ? 1u : 0u;
    uint32_t wtot;
    const uint32_t wexcl = wave_exclusive_scan(mine, wtot);
    const int wv = int(threadIdx.x) / 64;
    if (lane_id() == 0) wcnt[wv] = wtot;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < DR_BLOCK / 64; ++w) {
        before += w < wv ? wcnt[w] : 0u;
        total += wcnt[w];
    }
    if (wv == 0) {
        if (lane_id() == 0 && c > 0) __hip_atomic_store(&status[1 + c], ((unsigned long long)total << 2) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0;
        for (int64_t look = int64_t(c) - 1; look >= 0; look -= 64) {
            const int64_t idx = look - lane_id();
            unsigned long long v = 2; // below chunk 0: an inclusive count of zero
            if (idx >= 0) {
                do v = __hip_atomic_load(&status[1 + idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                while ((v & 3ull) == 0);
            }
            const uint64_t incl = __ballot((v & 3ull) == 2ull);
            const int stop = incl ? __ffsll((unsigned long long)incl) - 1 : 64; // the nearest chunk whose inclusive count is known
            unsigned long long part = lane_id() <= stop ? (v >> 2) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
            excl += part;
            if (incl) break;
        }
        if (lane_id() == 0) {
            __hip_atomic_store(&status[1 + c], ((excl + total) << 2) | 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    int64_t r = int64_t(s_excl) + before + wexcl;
    uint32_t slot[DR_ITEMS];
    bool live[DR_ITEMS];
    int64_t row[DR_ITEMS];
#pragma unroll
    for (int u = 0; u < DR_ITEMS; ++u) {
        live[u] = p[u] != 0;
        slot[u] = p[u] - 1;
        row[u] = r;
        if (live[u]) out_keys[r++] = (ordmin + j0 + u) ^ flip;
    }
    if (f.naggs > 0) finalize_rows<DR_ITEMS>(g, slot, live, row, f);
}

// ------------------------------------------------------------------ tail of the RANGE TIER (aggregate_common.hpp: RangeRec)
// tab[(p * Q + q) * W + s]: the table of partition p as workgroup q of Q saw it.  A workgroup (taken by ticket, so that every block
// below a waiting one has been started) owns the slots [s0, s0 + SB) of EVERY partition = the RE_BLOCK x ITEMS consecutive keys from
// key_min + s0 * parts on: it adds the Q partials of each (partition, slot) — SB consecutive 32-byte records per partition: coalesced
// — into LDS at the key's place (d = slot * parts + ((p ^ scramble(slot)) & (parts - 1)): the inverse of range_partition), then walks
// the keys in order, ITEMS per thread: occupied ones are counted across workgroups by the decoupled look-back of
// dense_rank_emit_kernel and written — key and aggregates — at their rank.  The last block leaves the group count in *total.
template <int ITEMS>
__global__ void __launch_bounds__(RE_BLOCK) agg_range_emit_kernel(const RangeRec *__restrict__ tab, int parts_log2, int Q, uint32_t W, uint64_t span, uint64_t key_min,
                                                                  unsigned long long *status, uint64_t *out_keys, FinalizeArgs f, unsigned long long *total) {
    constexpr int KB = RE_BLOCK * ITEMS; // keys per block
    extern __shared__ __attribute__((aligned(16))) unsigned char re_smem[];
    double *lsum = reinterpret_cast<double *>(re_smem);
    double *lmn = lsum + KB;
    double *lmx = lmn + KB;
    uint64_t *lcnt = reinterpret_cast<uint64_t *>(lmx + KB);
    __shared__ uint32_t s_chunk;
    __shared__ uint32_t wcnt[RE_BLOCK / 64];
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_chunk = uint32_t(atomicAdd(&status[0], 1ull));
    __syncthreads();
    const uint64_t c = s_chunk;
    const uint32_t parts = 1u << parts_log2, SB = uint32_t(KB) >> parts_log2, s0 = uint32_t(c) * SB; // (KB >= parts: SB >= 1)
    const uint32_t nblocks = (W + SB - 1) / SB;
    int sb_log2 = 0;
    while ((1u << sb_log2) < SB) ++sb_log2;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t i = uint32_t(k) * RE_BLOCK + threadIdx.x, p = i >> sb_log2, sl = i & (SB - 1), sg = s0 + sl;
        double sum = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
        uint64_t cnt = 0;
        if (sg < W) {
            const RangeRec *__restrict__ r = tab + size_t(p) * size_t(Q) * size_t(W) + sg;
            for (int q = 0; q < Q; ++q) {
                const RangeRec x = r[size_t(q) * size_t(W)];
                sum += x.sum;
                mn = x.mn < mn ? x.mn : mn;
                mx = x.mx > mx ? x.mx : mx;
                cnt = ((cnt & ~NAN_BIT64) + (x.cnt & ~NAN_BIT64)) | ((cnt | x.cnt) & NAN_BIT64);
            }
        }
        const uint32_t low = (p ^ range_scramble(sg, parts_log2)) & (parts - 1u), dl = (sl << parts_log2) | low;
        if (((uint64_t(sg) << parts_log2) | low) >= span) cnt = 0; // (beyond the range: no tuple can have named this slot)
        lsum[dl] = sum;
        lmn[dl] = mn;
        lmx[dl] = mx;
        lcnt[dl] = cnt;
    }
    __syncthreads();
    const uint32_t j0 = threadIdx.x * ITEMS;
    uint64_t cnts[ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        cnts[u] = lcnt[j0 + u];
        mine += cnts[u] ? 1u : 0u;
    }
    uint32_t wtot;
    const uint32_t wexcl = wave_exclusive_scan(mine, wtot);
    const int wv = int(threadIdx.x) / 64;
    if (lane_id() == 0) wcnt[wv] = wtot;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RE_BLOCK / 64; ++w) {
        before += w < wv ? wcnt[w] : 0u;
        tot += wcnt[w];
    }
    if (wv == 0) {
        if (lane_id() == 0 && c > 0) __hip_atomic_store(&status[1 + c], ((unsigned long long)tot << 2) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long excl = 0;
        for (int64_t look = int64_t(c) - 1; look >= 0; look -= 64) {
            const int64_t idx = look - lane_id();
            unsigned long long v = 2; // below block 0: an inclusive count of zero
            if (idx >= 0) {
                do v = __hip_atomic_load(&status[1 + idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                while ((v & 3ull) == 0);
            }
            const uint64_t incl = __ballot((v & 3ull) == 2ull);
            const int stop = incl ? __ffsll((unsigned long long)incl) - 1 : 64; // the nearest block whose inclusive count is known
            unsigned long long part = lane_id() <= stop ? (v >> 2) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
            excl += part;
            if (incl) break;
        }
        if (lane_id() == 0) {
            __hip_atomic_store(&status[1 + c], ((excl + tot) << 2) | 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if (c + 1 == nblocks) total[0] = excl + tot; // the group count; total[1] / total[2]: ~(first key - key_min) / last key - key_min (below)
        }
    }
    __syncthreads();
    int64_t r = int64_t(s_excl) + before + wexcl;
    const uint64_t key0 = key_min + (uint64_t(s0) << parts_log2) + j0;
    // the exact range of the keys, for the host (the next execution cuts its partitions from it): the block's first and last occupied key, two
    // atomics per block on zeroed words (the minimum as the maximum of the complement)
    if (mine) {
        const uint64_t d0 = (uint64_t(s0) << parts_log2) + j0;
        if (before + wexcl == 0) {
            int u0 = 0;
#pragma unroll
            for (int u = ITEMS - 1; u >= 0; --u) u0 = cnts[u] ? u : u0;
            atomicMax(&total[1], ~(unsigned long long)(d0 + uint64_t(u0)));
        }
        if (before + wexcl + mine == tot) {
            int u1 = 0;
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) u1 = cnts[u] ? u : u1;
            atomicMax(&total[2], (unsigned long long)(d0 + uint64_t(u1)));
        }
    }
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        if (!cnts[u]) continue;
        out_keys[r] = key0 + uint64_t(u);
        const uint64_t cnt = cnts[u] & ~NAN_BIT64;
        const double sum = lsum[j0 + u], mn = lmn[j0 + u], mx = (cnts[u] & NAN_BIT64) ? __longlong_as_double(0x7FF8000000000000ll) : lmx[j0 + u];
        if (f.partial) { // (one value column: table slot 0)
            f.out[0][r] = cnt;
            f.out[1][r] = d2u(sum);
            f.out[2][r] = d2u(mn);
            f.out[3][r] = d2u(mx);
        } else {
            for (int i = 0; i < f.naggs; ++i) {
                uint64_t w;
                switch (f.func[i]) {
                case NQE_AGG_COUNT: w = cnt; break;
                case NQE_AGG_SUM: w = d2u(sum); break;
                case NQE_AGG_AVG: w = d2u(sum / double(uint32_t(cnt))); break; // avg.rs:121 (cnt is u32)
                case NQE_AGG_MIN: w = d2u(mn); break;
                default: w = d2u(mx); break;
                }
                f.out[i][r] = w;
            }
        }
        ++r;
    }
}

// Tail of a SMALL hashed table (the first-attempt 8192-slot table: the headline's 1024 groups): collect + sort + finalize in one
// launch, enqueued ahead of the flag read-back.  Every workgroup compacts the occupied slots into LDS (keys in sort order + slot
// numbers; the same deterministic order in every workgroup), owns 64 of the G entries, and ranks each by counting the keys below
// it — keys in the table are distinct, so the ranks are exactly the permutation 0..G-1 of the sorted output.  Wave w of 16 scans
// one sixteenth of the entries with broadcast LDS reads (lane = entry), the partial counts meet in LDS, wave 0 writes key and
// aggregates of its entries at their ranks.  Work is G²/64 broadcast reads spread over G/64 workgroups (workgroups past the last
// entry leave after the compaction); it replaces collect (20 µs) + single-workgroup bitonic sort (21 µs) + finalize (7.5 µs)
// and the host round trip between them.

__global__ void __launch_bounds__(RANK_WAVES * 64) rank_finalize_kernel(GroupTable g, int signed_order, FinalizeArgs f, uint64_t *out_keys,
                                                                        const int *flags, int *mirror) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rank_smem[];
    const uint32_t slots = g.cap + 1;
    uint64_t *ok = reinterpret_cast<uint64_t *>(rank_smem);        // [slots] ordered keys of the occupied slots, compacted
    uint32_t *oslot = reinterpret_cast<uint32_t *>(ok + slots);    // [slots] their slot numbers
    __shared__ uint32_t part[RANK_WAVES][RANK_SLOTS];
    __shared__ uint32_t wbase[RANK_PASSES * RANK_WAVES + 1];
    const uint64_t flip = signed_order ? 0x8000000000000000ull : 0ull;
    const int wv = threadIdx.x / 64;
    const int passes = int((slots + blockDim.x - 1) / blockDim.x);
    // ---- compaction, pass 1: occupied slots per (pass, wave).  The ≤ 9 key words of a thread are requested back to back and
    // kept in registers for pass 2 (one memory round trip for the whole table instead of one per pass)
    uint64_t kreg[RANK_PASSES];
#pragma unroll
    for (int p = 0; p < RANK_PASSES; ++p) {
        const uint32_t s = uint32_t(p) * (RANK_WAVES * 64) + threadIdx.x;
        kreg[p] = s < slots ? g.keys[s] : EMPTY_KEY;
    }
#pragma unroll
    for (int p = 0; p < RANK_PASSES; ++p) {
        const uint64_t m = __ballot(kreg[p] != EMPTY_KEY);
        if (lane_id() == 0 && p < passes) wbase[p * RANK_WAVES + wv] = uint32_t(__popcll(m));
    }
    __syncthreads();
    if (wv == 0) { // exclusive scan of the ≤ 144 counts
        uint32_t run = 0;
        const int n = passes * RANK_WAVES;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane_id();
            const uint32_t c = i < n ? wbase[i] : 0u;
            uint32_t tot;
            const uint32_t ex = wave_exclusive_scan(c, tot);
            if (i < n) wbase[i] = run + ex;
            run += tot;
        }
        if (lane_id() == 0) wbase[n] = run;
    }
    __syncthreads();
    const uint32_t G = wbase[passes * RANK_WAVES];
    if (blockIdx.x == 0 && threadIdx.x < NQE_NUM_FLAGS) {
        // this is the last kernel before the read-back: the flags of the kernels before it and the group count go straight into
        // the pinned host mirror (no device-to-host copy command between the kernel and the host's wait)
        mirror[threadIdx.x] = threadIdx.x == NQE_FLAG_GROUP_COUNT ? int(G) : flags[threadIdx.x];
        __threadfence_system();
    }
    if (blockIdx.x * RANK_SLOTS >= G) return; // no entries for this workgroup
    // ---- compaction, pass 2
#pragma unroll
    for (int p = 0; p < RANK_PASSES; ++p) {
        const uint32_t s = uint32_t(p) * (RANK_WAVES * 64) + threadIdx.x;
        const bool used = kreg[p] != EMPTY_KEY;
        const uint64_t m = __ballot(used);
        if (used) {
            const uint32_t pos = wbase[p * RANK_WAVES + wv] + uint32_t(__popcll(m & lanemask_lt()));
            ok[pos] = (s == g.cap ? EMPTY_KEY : kreg[p]) ^ flip;
            oslot[pos] = s;
        }
    }
    __syncthreads();
    // ---- rank
    const uint32_t e = blockIdx.x * RANK_SLOTS + lane_id();
    const uint64_t mine = e < G ? ok[e] : ~0ull;
    const uint32_t seg = (G + RANK_WAVES - 1) / RANK_WAVES;
    const uint32_t lo = uint32_t(wv) * seg < G ? uint32_t(wv) * seg : G, hi = lo + seg < G ? lo + seg : G;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t j = lo;
    for (; j + 4 <= hi; j += 4) {
        c0 += ok[j] < mine;
        c1 += ok[j + 1] < mine;
        c2 += ok[j + 2] < mine;
        c3 += ok[j + 3] < mine;
    }
    for (; j < hi; ++j) c0 += ok[j] < mine;
    part[wv][lane_id()] = c0 + c1 + c2 + c3;
    __syncthreads();
    if (wv == 0 && e < G) {
        uint32_t rank = 0;
#pragma unroll
        for (int w = 0; w < RANK_WAVES; ++w) rank += part[w][lane_id()];
        out_keys[rank] = mine ^ flip;
        finalize_row(g, oslot[e], int64_t(rank), f);
    }
}

// merges partial-state rows (one row per (rank, group)) into the table; V = naggs
__global__ void merge_states_kernel(GroupTable g, const uint64_t *keys, int64_t n, int naggs, const uint64_t *const *state_cols,
                                    int *flags) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        int64_t slot = keys ? global_find_or_insert(g, keys[r], flags) : 0;
        if (slot < 0) continue;
        for (int i = 0; i < naggs; ++i) {
            uint64_t cnt = state_cols[4 * i + 0][r];
            double sum = u2d(state_cols[4 * i + 1][r]);
            double mn = u2d(state_cols[4 * i + 2][r]);
            double mx = u2d(state_cols[4 * i + 3][r]);
            bool nan = mx != mx;
            global_update(g, slot, i, cnt, sum, true, f64_to_ord(mn), nan ? f64_to_ord(-DBL_MAX) : f64_to_ord(mx), true, nan);
        }
    }
}

// the same merge straight from an all-gathered nqe_table_pack_words buffer: part p = `ncols` column segments of `stride` words
// (key first when nk = 1, then {count, sum, min, max} per aggregate) + one header word = its row count, which is read HERE — the
// host never needs the counts, so the exchange costs no read-back of its own.  A header beyond the stride (the sender took the
// exact-size path) raises NQE_FLAG_OOB.
__global__ void merge_packed_kernel(GroupTable g, const uint64_t *src, int nparts, int64_t stride, int nk, int naggs, int *flags) {
    const int ncols = nk + 4 * naggs;
    const int64_t part_words = int64_t(ncols) * stride + 1;
    const int64_t nthreads = int64_t(gridDim.x) * blockDim.x;
    for (int p = 0; p < nparts; ++p) {
        const uint64_t *base = src + int64_t(p) * part_words;
        const uint64_t rows = base[int64_t(ncols) * stride];
        if (rows > uint64_t(stride)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&flags[NQE_FLAG_OOB], 1);
            continue;
        }
        for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < int64_t(rows); r += nthreads) {
            int64_t slot = nk ? global_find_or_insert(g, base[r], flags) : 0;
            if (slot < 0) continue;
            for (int i = 0; i < naggs; ++i) {
                const uint64_t *st = base + int64_t(nk + 4 * i) * stride + r;
                uint64_t cnt = st[0];
                double sum = u2d(st[stride]);
                double mn = u2d(st[2 * stride]);
                double mx = u2d(st[3 * stride]);
                bool nan = mx != mx;
                global_update(g, slot, i, cnt, sum, true, f64_to_ord(mn), nan ? f64_to_ord(-DBL_MAX) : f64_to_ord(mx), true, nan);
            }
        }
    }
}

__global__ void iota_slots_kernel(uint32_t *out, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) out[i] = uint32_t(i);
}

// the program of a predicate tree, from the kernel arguments into the device buffer the streaming kernel reads it from
// min / max of a plain 8-byte integer key column in its own order (flip = the sign bit for Int64): out[0] = min, out[1] = max, both
// in the flipped (unsigned-comparable) form; the caller starts them at ~0 / 0
__global__ void __launch_bounds__(256) key_range_kernel(const uint64_t *keys, int64_t n, uint64_t flip, unsigned long long *out) {
    uint64_t mn = ~0ull, mx = 0ull;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const uint64_t k = __builtin_nontemporal_load(&keys[i]) ^ flip;
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = (uint64_t)__shfl_down((unsigned long long)mn, o, 64), b = (uint64_t)__shfl_down((unsigned long long)mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], (unsigned long long)mn);
        atomicMax(&out[1], (unsigned long long)mx);
    }
}

// A SAMPLE of the group keys, taken by the first execution of a query shape (no predicate): KEY_SAMPLE rows spread evenly over the
// table — one per thread, at a pseudo-random offset inside its stride, so that keys in arithmetic progression (row numbers under a
// modulus) do not alias with the stride — their keys' min / max in the flipped (unsigned-comparable) form and the EXACT number of
// distinct keys among them (open-addressing set of KEY_SAMPLE_SLOTS words, all ones = empty; a key of all ones is counted through
// out[3]).  The distinct count is a LOWER BOUND of the query's groups: a tier it rules out would certainly have overflowed.
// tables sampled: from 2^18 rows (a quarter of the keys then) — below, an overfull workgroup table spills to the global one instead of
// asking for another tier.  (2^22 until the end of round 4: half a million rows over 90 000 groups went single pass -> two key subsets
// -> partitioned, 2.0 ms for a 0.13 ms query)
__global__ void __launch_bounds__(256) key_sample_kernel(const uint64_t *keys, int64_t n, SimpleExpr ke, uint64_t flip, unsigned long long *set,
                                                         unsigned long long *out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t stride = n / KEY_SAMPLE; // the host samples tables of KEY_SAMPLE_MIN_ROWS rows and more (stride >= 4)
    uint64_t h = uint64_t(i) + 0x9E3779B97F4A7C15ull;
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
    h ^= h >> 31;
    const int64_t row = i * stride + int64_t(h % uint64_t(stride));
    const uint64_t x = keys[row];
    const uint64_t key = ke.nops ? eval_simple<false>(ke, x, false, nullptr) : x;
    uint64_t mn = key ^ flip, mx = mn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = (uint64_t)__shfl_down((unsigned long long)mn, o, 64), b = (uint64_t)__shfl_down((unsigned long long)mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], (unsigned long long)mn);
        atomicMax(&out[1], (unsigned long long)mx);
    }
    if (key == ~0ull) {
        atomicOr(&out[3], 1ull);
        return;
    }
    constexpr uint32_t MASK = (1u << KEY_SAMPLE_SLOTS_LOG2) - 1u;
    uint32_t slot = uint32_t((key * GOLD) >> (64 - KEY_SAMPLE_SLOTS_LOG2));
    for (;;) {
        const unsigned long long prev = atomicCAS(&set[slot], ~0ull, (unsigned long long)key);
        if (prev == ~0ull) {
            atomicAdd(&out[2], 1ull);
            return;
        }
        if (prev == key) return;
        slot = (slot + 1) & MASK;
    }
}

// folds the per-workgroup direct-mapped tables the run-time specialised streaming kernel wrote (expr_jit.hpp: nqe_jit_agg;
// [grid][span] sums | mins | maxs, then counts with the NaN mark in their top bit) into the group table: one thread per slot,
// key = slot - bias
__global__ void __launch_bounds__(256) agg_merge_partials_kernel(const double *psum, const double *pmn, const double *pmx, const uint32_t *pcnt, int grid, uint32_t span,
                                                                 int64_t bias, GroupTable g, int v, int *flags) {
    // one WAVE per slot, its lanes over the workgroups (a thread per slot walked the 256 tables one dependent load after the other:
    // 0.15 ms for 2047 slots — a quarter of the streaming kernel's time at 2x10^8 rows)
    const uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= span) return;
    uint64_t c = 0;
    uint32_t nanm = 0;
    double sum = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
    for (int b = lane_id(); b < grid; b += 64) {
        const size_t o = size_t(b) * span + s;
        const uint32_t cc = pcnt[o];
        if (cc == 0) continue;
        c += cc & ~NAN_BIT;
        nanm |= cc & NAN_BIT;
        sum += psum[o];
        mn = fmin(mn, pmn[o]);
        mx = fmax(mx, pmx[o]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        c += (uint64_t)__shfl_xor((unsigned long long)c, d, 64);
        nanm |= uint32_t(__shfl_xor(int(nanm), d, 64));
        sum += __shfl_xor(sum, d, 64);
        mn = fmin(mn, __shfl_xor(mn, d, 64));
        mx = fmax(mx, __shfl_xor(mx, d, 64));
    }
    if (lane_id() != 0 || c == 0) return; // (c == 0: no row of this key passed the predicate)
    const int64_t gslot = global_find_or_insert(g, uint64_t(int64_t(s) - bias), flags);
    if (gslot < 0) return;
    global_update(g, gslot, v, c, sum, true, f64_to_ord(mn), f64_to_ord(mx), true, nanm != 0);
}

// the same fold for the STATIC streaming kernel's direct-mapped tables (AggArgs::partials), shaped for up to 4096 slots x 256 workgroups:
// a 256-thread block takes 16 consecutive slots; thread (slot, g) = (tid & 15, tid >> 4) walks the workgroups g, g + 16, ... — 16
// consecutive lanes read 128 contiguous bytes of one workgroup's table per array — the sixteen partial results of a slot meet in LDS.
// sub_log2 (a direct-mapped table in 2^sub_log2 key-range subsets, AggArgs::direct_sub_width): slot S of the whole range is slot S % span of
// the tables of subset S / span — the workgroups whose index has that subset in bits [3, 3 + sub_log2)
__global__ void __launch_bounds__(256) agg_fold_partials_kernel(const double *psum, const double *pmn, const double *pmx, const uint32_t *pcnt, int grid, uint32_t span,
                                                                int64_t bias, int need_minmax, GroupTable g, int v, int *flags, int sub_log2, RangeRec *tab) {
    __shared__ double ssum[16][16], smn[16][16], smx[16][16];
    __shared__ unsigned long long scnt[16][16];
    __shared__ uint32_t snan[16][16];
    const uint32_t sl = threadIdx.x & 15, gq = threadIdx.x >> 4, S = blockIdx.x * 16 + sl, total = span << sub_log2;
    const uint32_t subset = sub_log2 ? S / span : 0u, s = sub_log2 ? S % span : S;
    uint64_t c = 0;
    uint32_t nanm = 0;
    double sum = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
    if (S < total) {
#pragma unroll 4
        for (int i = int(gq); i < (grid >> sub_log2); i += 16) {
            const int b = sub_log2 ? int(((uint32_t(i) >> 3) << (3 + sub_log2)) | (subset << 3) | (uint32_t(i) & 7u)) : i;
            const size_t o = size_t(b) * span + s;
            const uint32_t cc = pcnt[o];
            const double x = psum[o], lo = need_minmax ? pmn[o] : DBL_MAX, hi = need_minmax ? pmx[o] : -DBL_MAX;
            if (cc == 0) continue;
            c += cc & ~NAN_BIT;
            nanm |= cc & NAN_BIT;
            sum += x;
            mn = fmin(mn, lo);
            mx = fmax(mx, hi);
        }
    }
    ssum[gq][sl] = sum;
    smn[gq][sl] = mn;
    smx[gq][sl] = mx;
    scnt[gq][sl] = c;
    snan[gq][sl] = nanm;
    __syncthreads();
    if (gq != 0 || S >= total) return;
    for (int q = 1; q < 16; ++q) {
        c += scnt[q][sl];
        nanm |= snan[q][sl];
        sum += ssum[q][sl];
        mn = fmin(mn, smn[q][sl]);
        mx = fmax(mx, smx[q][sl]);
    }
    if (tab) { // the range tier's tail takes it from here (agg_range_emit_kernel: one partition, one table of `total` slots in key order)
        RangeRec r;
        r.sum = sum;
        r.mn = mn;
        r.mx = mx;
        r.cnt = c | (nanm ? NAN_BIT64 : 0ull);
        tab[S] = r;
        return;
    }
    if (c == 0) return;
    const int64_t gslot = global_find_or_insert(g, uint64_t(int64_t(S) - bias), flags);
    if (gslot < 0) return;
    global_update(g, gslot, v, c, sum, true, f64_to_ord(mn), f64_to_ord(mx), need_minmax != 0, nanm != 0);
}

__global__ void store_tree_kernel(TreePred p, TreeInstr *dst) {
    if (int(threadIdx.x) < p.n) dst[threadIdx.x] = p.ins[threadIdx.x];
}


template __global__ void dense_rank_emit_kernel<4>(GroupTable, const uint32_t *, uint64_t, uint64_t, uint64_t, unsigned long long *, uint64_t *, FinalizeArgs);
template __global__ void agg_range_emit_kernel<1>(const RangeRec *, int, int, uint32_t, uint64_t, uint64_t, unsigned long long *, uint64_t *, FinalizeArgs, unsigned long long *);
template __global__ void agg_range_emit_kernel<4>(const RangeRec *, int, int, uint32_t, uint64_t, uint64_t, unsigned long long *, uint64_t *, FinalizeArgs, unsigned long long *);

} // namespace agg
} // namespace nqe

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::agg::table_init_kernel);
