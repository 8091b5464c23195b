// Read-bandwidth ceiling of one MI355X for the access pattern of the headline kernel: two 8 GB columns streamed once with
// non-temporal 8-byte loads per lane, a trivial reduction as the only compute.  hipcc -O3 --offload-arch=gfx950 -o stream_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, int WIDE>
__global__ void __launch_bounds__(1024) read2(const uint64_t *a, const uint64_t *b, int64_t n, uint64_t *out) {
    uint64_t acc = 0;
    const int64_t step = int64_t(blockDim.x) * U * (WIDE ? 2 : 1);
    for (int64_t base = int64_t(blockIdx.x) * step; base < n; base += int64_t(gridDim.x) * step) {
        if (WIDE) {
            typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
            v2u64 x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t i = base / 2 + int64_t(u) * blockDim.x + threadIdx.x;
                i = i < n / 2 - 1 ? i : n / 2 - 1;
                x[u] = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(a) + i);
                y[u] = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(b) + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= x[u].x + x[u].y + y[u].x + y[u].y;
        } else {
            uint64_t x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
                i = i < n - 1 ? i : n - 1;
                x[u] = __builtin_nontemporal_load(a + i);
                y[u] = __builtin_nontemporal_load(b + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= x[u] + y[u];
        }
    }
    if (acc == 0x123456789abcdefull) out[0] = acc; // keep the loads alive
}

// Write-side ceilings for the kernels that emit as much as they read (compaction, the join's fused write): RD input columns
// streamed once, WR output columns written once (WR = 0: the read2 case above; RD = 0: a pure fill), NT picks non-temporal
// stores.  One 8-byte element per lane per column per unrolled step, like the product kernels.
template <int U, int RD, int WR, int NT>
__global__ void __launch_bounds__(1024) copy_rw(const uint64_t *in, uint64_t *outp, int64_t n) {
    const int64_t step = int64_t(blockDim.x) * U;
    for (int64_t base = int64_t(blockIdx.x) * step; base < n; base += int64_t(gridDim.x) * step) {
        uint64_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
            i = i < n - 1 ? i : n - 1;
            x[u] = uint64_t(i);
#pragma unroll
            for (int r = 0; r < RD; ++r) x[u] += __builtin_nontemporal_load(in + r * n + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
            if (WR == 0 && x[u] == 0x123456789abcdefull) outp[0] = x[u]; // read-only variant: keep the loads alive
            if (i < n) {
#pragma unroll
                for (int w = 0; w < WR; ++w) {
                    if (NT) __builtin_nontemporal_store(x[u] + w, outp + w * n + i);
                    else outp[w * n + i] = x[u] + w;
                }
            }
        }
    }
}

template <int U, int RD, int WR, int NT>
int run_rw(const uint64_t *in, uint64_t *outp, int64_t n, int blocks_per_cu, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; ++w) copy_rw<U, RD, WR, NT><<<grid, threads>>>(in, outp, n);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) copy_rw<U, RD, WR, NT><<<grid, threads>>>(in, outp, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("copy U=%d read=%d write=%d nt=%d blocks/CU=%d threads=%d: %.3f ms = %.0f GB/s\n", U, RD, WR, NT, blocks_per_cu, threads, ms,
           8.0 * (RD + WR) * n / ms / 1e6);
    return 0;
}

template <int U, int WIDE>
int run(const uint64_t *a, const uint64_t *b, int64_t n, uint64_t *out, int blocks_per_cu, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; ++w) read2<U, WIDE><<<grid, threads>>>(a, b, n, out);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) read2<U, WIDE><<<grid, threads>>>(a, b, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("U=%d wide=%d blocks/CU=%d threads=%d: %.3f ms = %.0f GB/s\n", U, WIDE, blocks_per_cu, threads, ms, 16.0 * n / ms / 1e6);
    return 0;
}

int main(int argc, char **argv) {
    const int64_t n = 1000000000;
    if (argc > 1 && std::string(argv[1]) == "calib") {
        // known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS access pattern (8-byte non-temporal loads
        // and stores, like the product kernels): read2 streams 2 x 8e9 B, the fill writes 4 x 1.6e9 B, the copy reads and writes 1.6e9 B
        uint64_t *a, *b, *out;
        CK(hipMalloc(&a, n * 8));
        CK(hipMalloc(&b, n * 8));
        CK(hipMalloc(&out, 8));
        CK(hipMemset(a, 1, n * 8));
        CK(hipMemset(b, 2, n * 8));
        for (int r = 0; r < 3; ++r) read2<4, 0><<<256, 1024>>>(a, b, n, out);
        for (int r = 0; r < 3; ++r) copy_rw<4, 0, 4, 1><<<256 * 2, 1024>>>(b, a, 200000000);
        for (int r = 0; r < 3; ++r) copy_rw<4, 1, 1, 1><<<256 * 2, 1024>>>(b, a, 200000000);
        CK(hipDeviceSynchronize());
        printf("calib: read2<4,0> 16e9 B read; copy_rw<4,0,4,1> 6.4e9 B written; copy_rw<4,1,1,1> 1.6e9 B read + 1.6e9 B written\n");
        return 0;
    }
    uint64_t *a, *b, *out;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, n * 8));
    CK(hipMemset(b, 2, n * 8));
    for (int bpc : {1, 2, 4, 8}) {
        for (int th : {256, 1024}) {
            if (run<4, 0>(a, b, n, out, bpc, th)) return 1;
            if (run<8, 0>(a, b, n, out, bpc, th)) return 1;
            if (run<4, 1>(a, b, n, out, bpc, th)) return 1;
        }
    }
    // one 0.8 GB column (the predicate pass of a 10^8-row selection): how much of the ceiling a 0.1 ms launch can reach at all
    // (back-to-back repetitions: up to a third of the column may still sit in the 256 MB Infinity Cache, so this is an upper bound)
    for (int bpc : {1, 2, 8}) {
        for (int th : {256, 1024}) {
            if (run_rw<8, 1, 0, 0>(b, a, 100000000, bpc, th)) return 1;
            if (run_rw<16, 1, 0, 0>(b, a, 100000000, bpc, th)) return 1;
        }
    }
    // read:write mixes at 2e8 rows per column (1.6 GB each): fill, 1:1 copy, the join's 2:4, the selection's 2:2
    const int64_t m = 200000000;
    uint64_t *wout = a; // a holds up to 5 columns of m rows (8 GB); b supplies the inputs
    for (int bpc : {1, 2, 8}) {
        for (int th : {256, 1024}) {
            if (run_rw<4, 0, 4, 0>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 0, 4, 1>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 1, 1, 0>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 1, 1, 1>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 2, 4, 0>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 2, 4, 1>(b, wout, m, bpc, th)) return 1;
            if (run_rw<16, 2, 4, 1>(b, wout, m, bpc, th)) return 1;
            if (run_rw<4, 2, 2, 1>(b, wout, m, bpc, th)) return 1;
        }
    }
    return 0;
}
