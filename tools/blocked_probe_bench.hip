// A BLOCKED probe against a table beyond L2, measured (VERDICT r04 task 5b; DESIGN §3.7 had only priced it).
// Baseline: out[i] = T[key[i]] — 10^8 random 8-byte reads from a table of M entries (M = 10^7: 80 MB), every one a 128-byte line
// from beyond the 4 MB per-XCD L2.  Blocked: the probe rows are taken in tiles of 2^22; per tile
//   K1  partitions {key, row in tile} by key SLICE (slices of 2^18 entries = 2 MB of table by default) — per 4096-row chunk an LDS
//       counting sort, one reservation per (chunk, slice) in the slice's buffer, the chunk's tuples of a slice leave as one run (64
//       tuples = 512 B on average);
//   K2  takes slice after slice (the workgroups of one XCD share a slice: its part of the table is fetched into that L2 once), looks
//       the payload up and writes it back BY ROW POSITION within the tile: out[tile + row] — an 8-byte store somewhere in the tile's
//       33 MB of output, the step the reference's output order (probe-row order, hash_join.rs:86-101) forces.
// The question: do the Infinity Cache / L2 absorb the tile-local scatter of K2 (then this beats the line-fetch floor), or does every
// 8-byte store cost a line (then it does not).  Also timed: K1 + K2 without the output scatter (payloads XOR-reduced), i.e. what a
// second partition pass by row range would have to fit into.
//   hipcc -O3 --offload-arch=gfx950 -o blocked_probe_bench blocked_probe_bench.hip ;  ./blocked_probe_bench [M] [log2 entries per slice]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int TILE_LOG2 = 22;
constexpr int64_t TILE = int64_t(1) << TILE_LOG2;
constexpr int CHUNK = 4096; // rows per K1 workgroup: 256 threads x 16

__global__ void fill_keys(uint64_t *k, int64_t n, uint64_t m, uint64_t seed) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        uint64_t x = (uint64_t(i) + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        k[i] = x % m;
    }
}
__global__ void fill_table(uint64_t *t, int64_t m) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < m; i += int64_t(gridDim.x) * blockDim.x) t[i] = uint64_t(i) * 3 + 1;
}

// ---- baseline: the direct gather
__global__ void __launch_bounds__(256) gather_direct(const uint64_t *__restrict__ key, const uint64_t *__restrict__ T, uint64_t *__restrict__ out, int64_t n) {
    constexpr int U = 4;
    const int64_t step = int64_t(blockDim.x) * U;
    for (int64_t base = int64_t(blockIdx.x) * step; base < n; base += int64_t(gridDim.x) * step) {
        uint64_t k[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
            k[u] = __builtin_nontemporal_load(key + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = T[k[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
            if (i < n) __builtin_nontemporal_store(v[u], out + i);
        }
    }
}

// ---- K1: one workgroup per 4096-row chunk of the tile
// tuple = (key within the slice) << TILE_LOG2 | row in tile
__global__ void __launch_bounds__(256) k1_partition(const uint64_t *__restrict__ key, int64_t tile_base, int64_t tile_rows, int slice_shift, int S, uint32_t *__restrict__ slice_fill,
                                                     uint64_t *__restrict__ buf, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *cnt = sm;              // [S]
    uint32_t *start = cnt + S;       // [S] exclusive scan within the chunk
    uint32_t *gbase = start + S;     // [S] reserved position in the slice's buffer
    uint64_t *stage = reinterpret_cast<uint64_t *>(gbase + S + (S & 1)); // [CHUNK]
    for (int s = threadIdx.x; s < S; s += blockDim.x) cnt[s] = 0;
    __syncthreads();
    const int64_t c0 = int64_t(blockIdx.x) * CHUNK;
    uint64_t k[16];
    uint32_t sl[16], rk[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int64_t r = c0 + u * 256 + threadIdx.x;
        k[u] = r < tile_rows ? __builtin_nontemporal_load(key + tile_base + r) : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        sl[u] = k[u] == ~0ull ? 0u : uint32_t(k[u] >> slice_shift);
        rk[u] = k[u] == ~0ull ? 0u : atomicAdd(&cnt[sl[u]], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int s = 0; s < S; ++s) {
            start[s] = run;
            run += cnt[s];
        }
    }
    if (int(threadIdx.x) < S && cnt[threadIdx.x]) gbase[threadIdx.x] = atomicAdd(&slice_fill[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (k[u] == ~0ull) continue;
        const uint64_t row = uint64_t(c0 + u * 256 + threadIdx.x);
        stage[start[sl[u]] + rk[u]] = ((k[u] & ((1ull << slice_shift) - 1ull)) << TILE_LOG2) | row;
    }
    __syncthreads();
    // copy-out: the staged tuples are in slice order; a lane finds its slice by binary search over `start`
    const uint32_t total = start[S - 1] + cnt[S - 1];
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        int lo = 0, hi = S - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (start[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        const int64_t at = int64_t(gbase[lo]) + (i - start[lo]);
        if (at < cap) __builtin_nontemporal_store(stage[i], buf + int64_t(lo) * cap + at);
    }
}

// ---- K2: workgroup (slice s, part q of Q); blockIdx = ((s / 8) * Q + q) * 8 + s % 8 — the Q workgroups of a slice on ONE XCD
template <int SCATTER>
__global__ void __launch_bounds__(256) k2_lookup(const uint64_t *__restrict__ buf, int64_t cap, const uint32_t *__restrict__ slice_fill, const uint64_t *__restrict__ T, int slice_shift,
                                                  int Q, uint64_t *__restrict__ out, int64_t tile_base, uint64_t *sink) {
    const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3, q = g % Q, s = (g / Q) * 8 + xcd;
    const int64_t n = min(int64_t(slice_fill[s]), cap);
    const uint64_t *__restrict__ b = buf + int64_t(s) * cap;
    const uint64_t *__restrict__ Ts = T + (uint64_t(s) << slice_shift);
    uint64_t acc = 0;
    constexpr int U = 4;
    for (int64_t base = int64_t(q) * 256 * U; base < n; base += int64_t(Q) * 256 * U) {
        uint64_t t[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256 + threadIdx.x;
            t[u] = __builtin_nontemporal_load(b + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = Ts[t[u] >> TILE_LOG2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i >= n) continue;
            if (SCATTER) out[tile_base + int64_t(t[u] & (TILE - 1))] = v[u];
            else acc ^= v[u];
        }
    }
    if (!SCATTER && acc == 0x123456789abcdefull) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int64_t n = 100000000;
    const uint64_t M = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const int slice_shift = argc > 2 ? atoi(argv[2]) : 18; // log2 of the entries per slice: 2^18 x 8 B = 2 MB of table (the per-XCD L2 holds 4 MB)
    const uint64_t slice_len = 1ull << slice_shift;
    const int S = int((((M + slice_len - 1) >> slice_shift) + 7) / 8 * 8); // slices, padded to a multiple of 8 (one XCD per slice; the padding stays empty)
    const int64_t cap = int64_t(double(TILE) * double(slice_len) / double(M) * 1.25) + 4096;
    uint64_t *key, *T, *out, *out2, *buf, *sink;
    uint32_t *fill;
    CK(hipMalloc(&key, n * 8));
    CK(hipMalloc(&out, n * 8));
    CK(hipMalloc(&out2, n * 8));
    CK(hipMalloc(&T, (uint64_t(S) << slice_shift) * 8));
    CK(hipMalloc(&buf, size_t(S) * cap * 8));
    CK(hipMalloc(&fill, S * 4));
    CK(hipMalloc(&sink, 8));
    fill_keys<<<4096, 256>>>(key, n, M, 12345);
    fill_table<<<4096, 256>>>(T, int64_t(M));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    printf("table %llu entries (%.1f MB), %d slices of %.2f MB, tiles of 2^%d rows, 10^8 probe rows\n", (unsigned long long)M, M * 8 / 1e6, S, slice_len * 8 / 1e6, TILE_LOG2);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        gather_direct<<<256 * 8, 256>>>(key, T, out, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("direct gather:                                   %.3f ms\n", ms);
    }
    const size_t k1_lds = size_t(3 * S + (S & 1)) * 4 + size_t(CHUNK) * 8;
    for (int Q : {4, 8, 16}) {
        for (int scatter = 1; scatter >= 0; --scatter) {
            for (int rep = 0; rep < 2; ++rep) {
                float k1ms = 0, k2ms = 0;
                CK(hipEventRecord(e0));
                for (int64_t tb = 0; tb < n; tb += TILE) {
                    const int64_t rows = n - tb < TILE ? n - tb : TILE;
                    CK(hipMemsetAsync(fill, 0, S * 4));
                    k1_partition<<<unsigned((rows + CHUNK - 1) / CHUNK), 256, k1_lds>>>(key, tb, rows, slice_shift, S, fill, buf, cap);
                    if (scatter) k2_lookup<1><<<S * Q, 256>>>(buf, cap, fill, T, slice_shift, Q, out2, tb, sink);
                    else k2_lookup<0><<<S * Q, 256>>>(buf, cap, fill, T, slice_shift, Q, out2, tb, sink);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                (void)k1ms; (void)k2ms;
                printf("blocked, Q = %2d workgroups per slice, %s: %.3f ms\n", Q, scatter ? "payload written by row position" : "payload not written (XOR-reduced)  ", ms);
            }
        }
    }
    // K1 alone
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int64_t tb = 0; tb < n; tb += TILE) {
            const int64_t rows = n - tb < TILE ? n - tb : TILE;
            CK(hipMemsetAsync(fill, 0, S * 4));
            k1_partition<<<unsigned((rows + CHUNK - 1) / CHUNK), 256, k1_lds>>>(key, tb, rows, slice_shift, S, fill, buf, cap);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("K1 (partition by slice) alone:                   %.3f ms\n", ms);
    }
    // correctness of the blocked form against the direct gather
    {
        for (int64_t tb = 0; tb < n; tb += TILE) {
            const int64_t rows = n - tb < TILE ? n - tb : TILE;
            CK(hipMemsetAsync(fill, 0, S * 4));
            k1_partition<<<unsigned((rows + CHUNK - 1) / CHUNK), 256, k1_lds>>>(key, tb, rows, slice_shift, S, fill, buf, cap);
            k2_lookup<1><<<S * 8, 256>>>(buf, cap, fill, T, slice_shift, 8, out2, tb, sink);
        }
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> a(1 << 20), b(1 << 20);
        int64_t bad = 0;
        for (int64_t off : {int64_t(0), int64_t(50000000), n - (1 << 20)}) {
            CK(hipMemcpy(a.data(), out + off, a.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), out2 + off, b.size() * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
        }
        printf("blocked == direct on 3 x 2^20 sampled output rows: %s\n", bad ? "NO" : "yes");
    }
    return 0;
}
