// micro_bench.hip — two hardware rates that decide the round-2 designs ("Round 2 microbenchmarks"):
//   gather : random reads of ELEM bytes out of a table of T bytes, index taken from a streamed key column (the hash-join
//            probe's access pattern), by table size, element width and load flavour — where is the per-XCD L2 cliff and what
//            does an access beyond it cost?
//   lds    : LDS atomic throughput per CU by instruction (ds_add_u32, ds_add_f64, ds_min_u64, returning ds_add_u32) on random
//            slots — what bounds the hash aggregate when every row flushes (random keys)?
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o micro_bench micro_bench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}

__global__ void fill_keys(uint64_t *k, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) k[i] = splitmix64(uint64_t(i));
}

// FLAVOUR 0 plain, 1 nontemporal, 2 relaxed agent-scope atomic load (sc1)
template <typename T, int FLAVOUR> __device__ __forceinline__ T ld(const T *p) {
    if (FLAVOUR == 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <> __device__ __forceinline__ uint64_t ld<uint64_t, 2>(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <> __device__ __forceinline__ uint32_t ld<uint32_t, 2>(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

template <typename T, int FLAVOUR, int U>
__global__ void __launch_bounds__(256) gather(const uint64_t *keys, int64_t n, const T *table, uint64_t slots_mask, uint64_t *out) {
    uint64_t acc = 0;
    const int64_t step = int64_t(blockDim.x) * U;
    for (int64_t base = int64_t(blockIdx.x) * step; base < n; base += int64_t(gridDim.x) * step) {
        uint64_t k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + int64_t(u) * blockDim.x + threadIdx.x;
            i = i < n - 1 ? i : n - 1;
            k[u] = __builtin_nontemporal_load(keys + i);
        }
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<T, FLAVOUR>(table + (k[u] & slots_mask));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (sizeof(T) == 16) acc += v[u].x + v[u].y;
            else acc += uint64_t(v[u]);
        }
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}
template <> __device__ __forceinline__ v2u64 ld<v2u64, 2>(const v2u64 *p) { return *p; }

template <typename T, int FLAVOUR, int U>
int run_gather(const uint64_t *keys, int64_t n, void *table, size_t table_bytes, uint64_t *out, const char *name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t mask = table_bytes / sizeof(T) - 1;
    const int grid = 256 * 8;
    for (int w = 0; w < 2; ++w) gather<T, FLAVOUR, U><<<grid, 256>>>(keys, n, (const T *)table, mask, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) gather<T, FLAVOUR, U><<<grid, 256>>>(keys, n, (const T *)table, mask, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("gather %-28s table %7.2f MB elem %2zu B U=%2d: %7.3f ms per %.0e accesses = %6.2f G/s\n", name, table_bytes / 1048576.0, sizeof(T), U, ms,
           double(n), n / (ms * 1e-3) / 1e9);
    return 0;
}

// ---- LDS atomics: 1024 threads, one block per CU; each thread issues ITERS x 4 atomics on pseudo-random slots
// OP 0 ds_add_u32, 1 ds_add_f64, 2 ds_min_u64 (unconditional), 3 ds_add_rtn_u32 (value used), 4 add_u32 + add_f64 (the flush of a
// random-key row), 5 read-before min/max (two ds_read_b64, no atomic) + add_u32 + add_f64 (the full flush)
template <int OP>
__global__ void __launch_bounds__(1024) lds_atomics(int slots, int iters, uint64_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *c32 = reinterpret_cast<uint32_t *>(smem);                     // [slots]
    double *f64 = reinterpret_cast<double *>(smem + size_t(slots) * 4);     // [slots]
    uint64_t *m64 = reinterpret_cast<uint64_t *>(smem + size_t(slots) * 12); // [slots]
    uint64_t *x64 = reinterpret_cast<uint64_t *>(smem + size_t(slots) * 20); // [slots]
    for (int s = threadIdx.x; s < slots; s += blockDim.x) { c32[s] = 0; f64[s] = 0; m64[s] = ~0ull; x64[s] = 0; }
    __syncthreads();
    uint64_t h = splitmix64(uint64_t(blockIdx.x) * 1024 + threadIdx.x);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t s = uint32_t(h >> 33) % uint32_t(slots);
            if (OP == 0) atomicAdd(&c32[s], 1u);
            if (OP == 1) unsafeAtomicAdd(&f64[s], 1.5);
            if (OP == 2) atomicMin((unsigned long long *)&m64[s], (unsigned long long)(h >> 8));
            if (OP == 3) acc += atomicAdd(&c32[s], 1u);
            if (OP == 4 || OP == 5) { atomicAdd(&c32[s], 1u); unsafeAtomicAdd(&f64[s], 1.5); }
            if (OP == 5) {
                const uint64_t v = h >> 8;
                if (v < m64[s]) atomicMin((unsigned long long *)&m64[s], (unsigned long long)v);
                if (v > x64[s]) atomicMax((unsigned long long *)&x64[s], (unsigned long long)v);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) acc += c32[threadIdx.x] + uint64_t(f64[threadIdx.x]) + m64[threadIdx.x];
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

template <int OP> int run_lds(int slots, uint64_t *out, const char *name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000, grid = 256;
    const size_t shm = size_t(slots) * 28;
    CK(hipFuncSetAttribute((const void *)lds_atomics<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, int(shm)));
    lds_atomics<OP><<<grid, 1024, shm>>>(slots, 10, out);
    CK(hipEventRecord(e0));
    lds_atomics<OP><<<grid, 1024, shm>>>(slots, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = double(iters) * 4 * 1024 * grid;
    printf("lds %-44s slots %5d: %7.3f ms for %.2e rows = %6.2f Grows/s chip-wide (%5.2f rows/clk/CU at 2.4 GHz)\n", name, slots, ms, rows, rows / (ms * 1e-3) / 1e9,
           rows / (ms * 1e-3) / 256 / 2.4e9);
    return 0;
}

int main(int argc, char **argv) {
    const char *what = argc > 1 ? argv[1] : "all";
    uint64_t *out;
    CK(hipMalloc(&out, 64));
    if (!strcmp(what, "all") || !strcmp(what, "lds")) {
        for (int slots : {1024, 4096}) {
            if (run_lds<0>(slots, out, "ds_add_u32")) return 1;
            if (run_lds<1>(slots, out, "ds_add_f64")) return 1;
            if (run_lds<2>(slots, out, "ds_min_u64")) return 1;
            if (run_lds<3>(slots, out, "ds_add_rtn_u32")) return 1;
            if (run_lds<4>(slots, out, "add_u32 + add_f64")) return 1;
            if (run_lds<5>(slots, out, "add_u32 + add_f64 + read-before min/max")) return 1;
        }
    }
    if (!strcmp(what, "gather1")) { // one shape for the PMC passes: how many bytes does a random 8-byte read beyond the L2 move?
        const int64_t n = 100000000;
        uint64_t *keys;
        CK(hipMalloc(&keys, size_t(n) * 8));
        fill_keys<<<2048, 256>>>(keys, n);
        void *table;
        CK(hipMalloc(&table, size_t(1) << 30));
        CK(hipMemset(table, 1, size_t(1) << 30));
        CK(hipDeviceSynchronize());
        if (run_gather<uint64_t, 0, 8>(keys, n, table, size_t(1) << 30, out, "u64 plain 1 GB")) return 1;
        if (run_gather<uint64_t, 0, 8>(keys, n, table, size_t(32) << 20, out, "u64 plain 32 MB")) return 1;
        return 0;
    }
    if (!strcmp(what, "all") || !strcmp(what, "gather")) {
        const int64_t n = 100000000;
        uint64_t *keys;
        CK(hipMalloc(&keys, size_t(n) * 8));
        fill_keys<<<2048, 256>>>(keys, n);
        void *table;
        const size_t max_bytes = size_t(1) << 30;
        CK(hipMalloc(&table, max_bytes));
        CK(hipMemset(table, 1, max_bytes));
        CK(hipDeviceSynchronize());
        for (size_t mb : {1, 2, 3, 4, 6, 8, 16, 32, 128, 1024}) {
            size_t bytes = mb << 20;
            if (mb == 3) bytes = size_t(2) << 20; // masks need powers of two: 3/6 stand for "repeat" rows, skipped
            if (mb == 3 || mb == 6) continue;
            if (run_gather<uint32_t, 0, 8>(keys, n, table, bytes, out, "u32 plain")) return 1;
            if (run_gather<uint64_t, 0, 8>(keys, n, table, bytes, out, "u64 plain")) return 1;
            if (run_gather<uint64_t, 1, 8>(keys, n, table, bytes, out, "u64 nontemporal")) return 1;
            if (run_gather<uint64_t, 2, 8>(keys, n, table, bytes, out, "u64 sc1 (agent atomic load)")) return 1;
            if (run_gather<v2u64, 0, 8>(keys, n, table, bytes, out, "16 B plain")) return 1;
            if (run_gather<uint64_t, 0, 16>(keys, n, table, bytes, out, "u64 plain, 16 in flight")) return 1;
        }
    }
    return 0;
}
