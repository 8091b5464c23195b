// atomics_bench.hip — how fast are global atomics on MI355X by scope, table size and layout?
// (decides the design of the >LDS-capacity group-by path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }

// MODE 0: agent-scope atomics, one shared table, SoA (cnt,sum,mn,mx in 4 arrays)
// MODE 1: agent-scope, AoS 64-byte slots (4 atomics hit one line)
// MODE 2: workgroup-scope atomics into a per-XCD private copy, AoS
// MODE 3: agent-scope, single u64 add only (SoA cnt) — per-op cost
// MODE 4: workgroup-scope per-XCD, single u64 add
template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t *tab, uint64_t slots, int64_t n, uint32_t *xcc_hist) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    uint32_t x = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[x], 1u);
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t s = splitmix64(uint64_t(i)) % slots;
        double v = double(i & 1023);
        if (MODE == 0) {
            __hip_atomic_fetch_add(&tab[s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add((double *)&tab[slots + s], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(&tab[2 * slots + s], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&tab[3 * slots + s], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {
            uint64_t *p = tab + s * 8;
            __hip_atomic_fetch_add(&p[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add((double *)&p[1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(&p[2], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&p[3], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {
            uint64_t *p = tab + (uint64_t(x) * slots + s) * 8;
            __hip_atomic_fetch_add(&p[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add((double *)&p[1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&p[2], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&p[3], uint64_t(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {
            __hip_atomic_fetch_add(&tab[s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(&tab[uint64_t(x) * slots + s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main() {
    const int64_t n = 100000000;
    uint64_t *tab; uint32_t *hist;
    const uint64_t maxslots = 1ull << 24;
    CK(hipMalloc(&tab, maxslots * 8 * 8 * 8)); // AoS x 8 XCD copies
    CK(hipMalloc(&hist, 32));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[5] = {"agent SoA x4", "agent AoS x4", "wg-scope perXCD AoS x4", "agent 1 add", "wg-scope perXCD 1 add"};
    for (uint64_t slots : {1ull << 12, 1ull << 16, 1ull << 20, 1ull << 24}) {
        for (int mode = 0; mode < 5; ++mode) {
            CK(hipMemset(tab, 0, slots * 8 * 8 * 8)); CK(hipMemset(hist, 0, 32));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            switch (mode) {
            case 0: k<0><<<2048, 256>>>(tab, slots, n, hist); break;
            case 1: k<1><<<2048, 256>>>(tab, slots, n, hist); break;
            case 2: k<2><<<2048, 256>>>(tab, slots, n, hist); break;
            case 3: k<3><<<2048, 256>>>(tab, slots, n, hist); break;
            default: k<4><<<2048, 256>>>(tab, slots, n, hist); break;
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            // checksum: total count over all copies must be n
            std::vector<uint64_t> h; uint64_t total = 0;
            size_t words = (mode == 0 || mode == 3) ? slots : (mode == 4 ? slots * 8 : slots * 8 * (mode == 2 ? 8 : 1));
            h.resize(words);
            CK(hipMemcpy(h.data(), tab, words * 8, hipMemcpyDeviceToHost));
            if (mode == 0 || mode == 3) for (uint64_t s = 0; s < slots; ++s) total += h[s];
            else if (mode == 4) for (uint64_t s = 0; s < slots * 8; ++s) total += h[s];
            else for (uint64_t s = 0; s < words / 8; ++s) total += h[s * 8];
            printf("slots %8llu  %-24s %8.3f ms  %6.2f Grows/s  count_ok=%d\n", (unsigned long long)slots, names[mode], ms, n / ms / 1e6, total == (uint64_t)n);
        }
    }
    uint32_t hh[8]; CK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
    printf("blocks per XCC id:"); for (int i = 0; i < 8; ++i) printf(" %u", hh[i]); printf("\n");
    return 0;
}
