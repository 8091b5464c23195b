// What can a partitioning pass reach on one MI355X?  (round 5: the partitioned aggregate's scatter moves its 16 B/row in + 12 B/row out
// at 4.1 TB/s; a plain copy reaches 5.5-6.)  A SYNTHETIC scatter with the product kernel's memory behaviour and none of its ranking
// logic: every workgroup streams its chunk of rows tile by tile (16 B/row, non-temporal), stages the tile's tuples in LDS, and appends
// run = tile / S consecutive tuples to each of its S private output streams (uniform keys: every partition receives tile / S tuples of
// every tile).  Swept: S (partitions), tile rows, workgroup shape, tuple layout (12-byte {f64, u32} records; 16-byte records; two
// streams f64 + u16 = 10 bytes), and whether a stream only ever receives whole 128-byte lines (what a per-partition carry buffer in
// LDS — software write combining — would produce).  hipcc -O3 --offload-arch=gfx950 -o scatter_bench scatter_bench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) Tuple12 {
    uint64_t val;
    uint32_t key;
};
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

// LAYOUT 0: 12-byte records, 1: 16-byte records, 2: SoA (8-byte values, 2-byte slots).  NTS: non-temporal stores.
// Every stream (workgroup w, partition p) owns cap tuples at (w * S + p) * cap.
template <int THREADS, int RPT, int LAYOUT, int NTS>
__global__ void __launch_bounds__(THREADS) synth_scatter(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, unsigned char *__restrict__ out, int64_t n, int64_t chunk, int S,
                                                         int64_t cap, int skew /* tuples: every stream starts `skew` tuples off a line boundary */) {
    constexpr int TILE = THREADS * RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *sval = reinterpret_cast<uint64_t *>(smem);            // [TILE]
    uint32_t *skey = reinterpret_cast<uint32_t *>(sval + TILE);     // [TILE]
    const int64_t lo = int64_t(blockIdx.x) * chunk, hi = lo + chunk < n ? lo + chunk : n, last = n - 1;
    const int run = TILE / S; // tuples per partition per tile
    int64_t written = skew;   // tuples already in each of this workgroup's streams
    for (int64_t base = lo; base < hi; base += TILE) {
        uint64_t kw[RPT], vw[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            int64_t row = base + int64_t(u) * THREADS + threadIdx.x;
            row = row < last ? row : last;
            kw[u] = __builtin_nontemporal_load(&a[row]);
            vw[u] = __builtin_nontemporal_load(&b[row]);
        }
        __syncthreads(); // the previous tile's copy-out is done
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            // "sorted position": a permutation of the tile that keeps consecutive lanes apart, like the rank of a random key does
            const int i = (u * THREADS + int(threadIdx.x)) ;
            const int pos = int((uint32_t(i) * 2654435761u) % uint32_t(TILE));
            sval[pos] = vw[u];
            skey[pos] = uint32_t(kw[u]);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < TILE; j += THREADS) {
            const int p = j / run, r = j - p * run;
            const int64_t at = (int64_t(blockIdx.x) * S + p) * cap + written + r;
            const uint64_t v = sval[j];
            const uint32_t k = skey[j];
            if (LAYOUT == 0) {
                Tuple12 t;
                t.val = v;
                t.key = k;
                reinterpret_cast<Tuple12 *>(out)[at] = t; // (12-byte stores: plain, as the product's)
            } else if (LAYOUT == 1) {
                v2u64 t;
                t.x = v;
                t.y = k;
                if (NTS) __builtin_nontemporal_store(t, reinterpret_cast<v2u64 *>(out) + at);
                else reinterpret_cast<v2u64 *>(out)[at] = t;
            } else if (LAYOUT == 3) { // SoA: 8-byte values, 4-byte keys (what whole-block flushes of a carry buffer would write)
                uint64_t *ov = reinterpret_cast<uint64_t *>(out);
                uint32_t *ok = reinterpret_cast<uint32_t *>(out + size_t(gridDim.x) * size_t(S) * size_t(cap) * 8);
                if (NTS) {
                    __builtin_nontemporal_store(v, ov + at);
                    __builtin_nontemporal_store(k, ok + at);
                } else {
                    ov[at] = v;
                    ok[at] = k;
                }
            } else {
                uint64_t *ov = reinterpret_cast<uint64_t *>(out);
                uint16_t *ok = reinterpret_cast<uint16_t *>(out + size_t(gridDim.x) * size_t(S) * size_t(cap) * 8);
                if (NTS) {
                    __builtin_nontemporal_store(v, ov + at);
                    __builtin_nontemporal_store(uint16_t(k), ok + at);
                } else {
                    ov[at] = v;
                    ok[at] = uint16_t(k);
                }
            }
        }
        written += run;
    }
}

template <int THREADS, int RPT, int LAYOUT, int NTS>
void run(const char *name, const uint64_t *a, const uint64_t *b, unsigned char *out, int64_t n, int wgs_per_cu, int S, int skew) {
    constexpr int TILE = THREADS * RPT;
    const int W = 256 * wgs_per_cu;
    int64_t chunk = ((n + W - 1) / W + TILE - 1) / TILE * TILE;
    const int64_t cap = ((chunk / S + 64 + 15) / 16 | 1) * 16; // an odd number of 256-byte units, as the product's slabs
    const size_t shmem = size_t(TILE) * 12;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)synth_scatter<THREADS, RPT, LAYOUT, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, int(shmem)));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0));
        synth_scatter<THREADS, RPT, LAYOUT, NTS><<<W, THREADS, shmem>>>(a, b, out, n, chunk, S, cap, skew);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double wb = LAYOUT == 0 ? 12.0 : LAYOUT == 1 ? 16.0 : LAYOUT == 3 ? 12.0 : 10.0;
    printf("%-22s thr %4d x %d/CU tile %5d S %3d run %4d tuples skew %2d: %.3f ms  %.0f GB/s\n", name, THREADS, wgs_per_cu, TILE, S, TILE / S, skew, best, (16.0 + wb) * n / best / 1e6);
}

int main() {
    const int64_t n = 100000000;
    uint64_t *a, *b;
    unsigned char *out;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMalloc(&out, size_t(n) * 16 * 2));
    CK(hipMemset(a, 1, n * 8));
    CK(hipMemset(b, 2, n * 8));
    for (int S : {64, 256, 512}) {
        for (int skew : {0, 5}) {
            run<1024, 8, 0, 0>("rec12 plain", a, b, out, n, 1, S, skew);
            run<1024, 8, 3, 0>("soa12 plain", a, b, out, n, 1, S, skew);
            run<1024, 8, 3, 1>("soa12 nt", a, b, out, n, 1, S, skew);
            run<1024, 4, 3, 0>("soa12 plain", a, b, out, n, 1, S, skew);
            run<1024, 4, 3, 1>("soa12 nt", a, b, out, n, 1, S, skew);
            run<512, 8, 3, 1>("soa12 nt", a, b, out, n, 2, S, skew);
            run<1024, 8, 1, 1>("rec16 nt", a, b, out, n, 1, S, skew);
        }
    }
    return 0;
}
