// Can the second reader of a row range be served by the XCD's L2?  (round 6: the two-key-subset aggregate reads every row twice — two
// workgroups of one XCD walk the same tiles, each keeping its half of the keys — and runs at exactly twice the one-reader time: the
// pair drifts apart and the second read goes to memory again, PMC traffic 1.73x.)  A SYNTHETIC pair: workgroups b and b + 8 (same XCD
// under round-robin dispatch) stream the same 16 B/row tiles with the product kernel's loop — 1024 threads, TU rows per lane per tile,
// a prefetched second register tile, non-temporal or plain loads, one workgroup per CU (LDS-sized like the product's tables) — and one
// of them does `extra` dependent ALU steps per row more (the subsets' work differs).  MODE 0: free-running.  MODE 1: the LEADER is
// throttled — every wave publishes its tile count, reads its counterpart wave's (the same rows of the same tiles) BEFORE the prefetch
// is issued, and after the tile spins (bounded) while it is more than `lag` tiles ahead.  READERS 1: one workgroup per range (the floor).
//   hipcc -O3 --offload-arch=gfx950 -o pair_bench pair_bench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BLOCK = 1024;

template <int TU, int MODE, bool NT>
__global__ void __launch_bounds__(BLOCK) pair_kernel(const uint64_t *__restrict__ ka, const uint64_t *__restrict__ va, int64_t n, int readers_log2, int extra, int lag, uint32_t *prog,
                                                     unsigned long long *out, unsigned long long *waited) {
    extern __shared__ uint32_t lds[]; // (only its size matters: one workgroup per CU)
    const uint32_t sub_mask = (1u << readers_log2) - 1u;
    const uint32_t subset = (blockIdx.x >> 3) & sub_mask;
    const uint32_t lane_wg = (blockIdx.x & 7u) | ((blockIdx.x >> (3 + readers_log2)) << 3);
    const uint32_t lanes = gridDim.x >> readers_log2;
    const int wave = threadIdx.x >> 6;
    const int64_t step = int64_t(BLOCK) * TU, stride = int64_t(lanes) * step, last = n - 1;
    // progress words: [range][subset][wave]
    uint32_t *mine = prog + (size_t(lane_wg) * 2 + subset) * 16 + wave;
    const uint32_t *theirs = prog + (size_t(lane_wg) * 2 + (subset ^ 1u)) * 16 + wave;
    struct Tile {
        uint64_t k[TU], v[TU];
    };
    auto load = [&](Tile &t, int64_t base) {
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            int64_t row = base + int64_t(u) * BLOCK + threadIdx.x;
            row = row < last ? row : last;
            if (NT) {
                t.k[u] = __builtin_nontemporal_load(&ka[row]);
                t.v[u] = __builtin_nontemporal_load(&va[row]);
            } else {
                t.k[u] = ka[row];
                t.v[u] = va[row];
            }
        }
    };
    uint64_t acc = 0, spins = 0;
    const int work = subset ? extra : 0;
    auto process = [&](const Tile &t) {
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            uint64_t x = t.k[u] ^ t.v[u];
            for (int i = 0; i < work; ++i) x = x * 0x9E3779B97F4A7C15ull + 1; // dependent chain
            acc += x;
        }
    };
    uint32_t it = 0;
    auto throttle = [&](uint32_t seen) {
        if (MODE == 0 || readers_log2 == 0) return;
        // (wave-uniform: every lane read the same word)
        seen = __builtin_amdgcn_readfirstlane(seen);
        int budget = 4096;
        while (seen + uint32_t(lag) < it && budget-- > 0) {
            __builtin_amdgcn_s_sleep(8);
            seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            ++spins;
        }
    };
    int64_t base = int64_t(lane_wg) * step;
    if (base < n) {
        Tile A, B;
        load(A, base);
        while (true) {
            uint32_t seen = 0;
            if (MODE == 1 && readers_log2) seen = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // issued ahead of the prefetch: back before it
            load(B, base + stride);
            process(A);
            ++it;
            if (MODE == 1 && readers_log2 && (threadIdx.x & 63) == 0) __hip_atomic_store(mine, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            throttle(seen);
            base += stride;
            if (base >= n) break;
            if (MODE == 1 && readers_log2) seen = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            load(A, base + stride);
            process(B);
            ++it;
            if (MODE == 1 && readers_log2 && (threadIdx.x & 63) == 0) __hip_atomic_store(mine, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            throttle(seen);
            base += stride;
            if (base >= n) break;
        }
    }
    if (MODE == 1 && readers_log2 && (threadIdx.x & 63) == 0) __hip_atomic_store(mine, 0xffffffffu - 65536u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // done: nobody waits for this wave
    if (acc == 0x1234567u) out[0] = acc;
    if ((threadIdx.x & 63) == 0 && spins) atomicAdd(waited, (unsigned long long)spins);
    if (threadIdx.x == 0 && lds[0] == 0x7fffffff) out[1] = 1;
}

template <int TU, int MODE, bool NT>
void run(const char *name, const uint64_t *k, const uint64_t *v, int64_t n, int readers_log2, int extra, int lag, uint32_t *prog, unsigned long long *out) {
    const int grid = 256 << readers_log2;
    const size_t shmem = 120 * 1024;
    CK(hipFuncSetAttribute((const void *)pair_kernel<TU, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, int(shmem)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned long long w = 0;
    for (int r = 0; r < 6; ++r) {
        CK(hipMemsetAsync(prog, 0, size_t(256) * 2 * 16 * 4));
        CK(hipMemsetAsync(out + 2, 0, 8));
        CK(hipEventRecord(e0));
        pair_kernel<TU, MODE, NT><<<grid, BLOCK, shmem>>>(k, v, n, readers_log2, extra, lag, prog, out, out + 2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) {
            best = ms;
            CK(hipMemcpy(&w, out + 2, 8, hipMemcpyDeviceToHost));
        }
    }
    printf("%-10s TU %d %s readers %d extra %2d lag %d: %.4f ms  %.0f GB/s of the 16 B/row read once   (spins %llu)\n", name, TU, NT ? "nt   " : "plain", 1 << readers_log2, extra, lag, best,
           double(n) * 16 / best / 1e6, w);
}

int main() {
    const int64_t n = 100000000;
    uint64_t *k, *v;
    uint32_t *prog;
    unsigned long long *out;
    CK(hipMalloc(&k, n * 8));
    CK(hipMalloc(&v, n * 8));
    CK(hipMalloc(&prog, size_t(256) * 2 * 16 * 4));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(k, 1, n * 8));
    CK(hipMemset(v, 2, n * 8));
    CK(hipMemset(out, 0, 64));
    run<8, 0, true>("one", k, v, n, 0, 0, 0, prog, out);
    run<4, 0, true>("one", k, v, n, 0, 0, 0, prog, out);
    for (int extra : {0, 4, 16}) {
        run<8, 0, true>("free", k, v, n, 1, extra, 0, prog, out);
        run<8, 0, false>("free", k, v, n, 1, extra, 0, prog, out);
        run<4, 0, true>("free", k, v, n, 1, extra, 0, prog, out);
        for (int lag : {0, 1, 2, 4}) {
            run<8, 1, true>("throttled", k, v, n, 1, extra, lag, prog, out);
            run<8, 1, false>("throttled", k, v, n, 1, extra, lag, prog, out);
            run<4, 1, true>("throttled", k, v, n, 1, extra, lag, prog, out);
            run<4, 1, false>("throttled", k, v, n, 1, extra, lag, prog, out);
        }
    }
    return 0;
}
