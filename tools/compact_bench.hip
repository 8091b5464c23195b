// Variants of C2's two kernels (keep mask from `id < K`, stable compaction of `age + 100` by the mask) timed the way the product runs
// them: keep(id) and compact(age -> out) alternate, each bracketed by HIP events, so every launch reads data the caches do not hold.
// 10^8 rows; ids = row numbers ("sorted": the first half is kept) or a hash ("random": every tile keeps ~half).
// hipcc -O3 --offload-arch=gfx950 -o compact_bench compact_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int TILE_ROWS = 4096, TILE_WORDS = 64;
__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & 63; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint64_t bcast64(uint64_t x, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
    return (uint64_t(hi) << 32) | lo;
}
__device__ __forceinline__ uint32_t bcast32(uint32_t x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, lane); }
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t &total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane_id() >= d) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void fill_kernel(uint64_t *ids, uint64_t *age, int64_t n, int random_ids) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        ids[i] = random_ids ? splitmix64(uint64_t(i) + 1) % uint64_t(n) : uint64_t(i);
        age[i] = 18 + splitmix64(uint64_t(i) + 2) % 60;
    }
}

// ---------------------------------------------------------------- keep variants
// K0: the product's keep_from_range_strided_kernel (R loads per lane per chunk, one atomic per chunk)
template <int R, int NT, int ATOMIC>
__global__ void __launch_bounds__(256) keep_strided(const uint64_t *__restrict__ words, uint64_t limit, int64_t n, uint64_t *keep, uint32_t *tile_counts) {
    const int lane = lane_id();
    const int64_t n_chunks = (n + 64 * R - 1) / (64 * R), last = n - 1;
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t row0 = chunk * (64 * R) + lane;
        uint64_t v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * 64;
            v[r] = NT ? __builtin_nontemporal_load(&words[row < last ? row : last]) : words[row < last ? row : last];
        }
        uint32_t total = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * 64;
            const uint64_t kw = __ballot(row < n && v[r] < limit);
            if (row - lane < n && lane == 0) keep[chunk * R + r] = kw;
            total += __popcll(kw);
        }
        if (ATOMIC && lane == 0 && total) atomicAdd(&tile_counts[(chunk * (64 * R)) / TILE_ROWS], total);
    }
}
// K1: a 256-thread workgroup per 4096-row tile (16 rows per lane, all in flight), the tile's count written once (no atomics, no memset)
template <int NT>
__global__ void __launch_bounds__(256) keep_tile(const uint64_t *__restrict__ words, uint64_t limit, int64_t n, int64_t ntiles, uint64_t *keep, uint32_t *tile_counts) {
    __shared__ uint32_t wtot[4];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t last = n - 1;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TILE_ROWS + wave * 1024 + lane;
        uint64_t v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + r * 64;
            v[r] = NT ? __builtin_nontemporal_load(&words[row < last ? row : last]) : words[row < last ? row : last];
        }
        uint32_t total = 0;
        uint64_t mine = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + r * 64;
            const uint64_t kw = __ballot(row < n && v[r] < limit);
            if (lane == r) mine = kw;
            total += __popcll(kw);
        }
        if (lane < 16 && (tile * TILE_WORDS + wave * 16 + lane) * 64 < n) keep[tile * TILE_WORDS + wave * 16 + lane] = mine;
        if (lane == 0) wtot[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- compaction variants
// C0: the product's compact_strided_kernel<EXPR> (chunk of CW keep words per wave, direct stores)
template <int CW, int NTS>
__global__ void __launch_bounds__(256) compact_strided(const uint64_t *__restrict__ words, const uint64_t *keep, const uint64_t *tile_offsets, int64_t n, int64_t ntiles, uint64_t *out) {
    constexpr int CPT = TILE_WORDS / CW, SEL_B = 8;
    const int64_t nwords = (n + 63) / 64, last = n - 1, n_chunks = ntiles * CPT;
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t tile = chunk / CPT;
        const int c0 = int(chunk % CPT) * CW;
        const uint64_t base = tile_offsets[tile];
        if (tile_offsets[tile + 1] == base) continue;
        const int64_t w = tile * TILE_WORDS + lane_id();
        const uint64_t my_word = w < nwords ? keep[w] : 0;
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        for (int k0 = c0; k0 < c0 + CW; k0 += SEL_B) {
            uint64_t kw[SEL_B];
            bool any = false;
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                kw[k] = bcast64(my_word, k0 + k);
                any = any || kw[k] != 0;
            }
            if (!any) continue;
            uint64_t v[SEL_B];
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                const int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]);
            }
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                const uint32_t off = bcast32(my_off, k0 + k);
                if ((kw[k] >> lane_id()) & 1) {
                    const uint64_t x = v[k] + 100;
                    uint64_t *dst = &out[base + off + __popcll(kw[k] & lanemask_lt())];
                    if (NTS) __builtin_nontemporal_store(x, dst);
                    else *dst = x;
                }
            }
        }
    }
}
// C1: staged through LDS, THREADS-thread workgroup per 4096-row tile, aligned whole-wave stores; PF: next tile's loads before the copy-out
template <int THREADS, int PF, int NTS>
__global__ void __launch_bounds__(THREADS) compact_staged(const uint64_t *__restrict__ words, const uint64_t *__restrict__ keep, const uint64_t *__restrict__ tile_offsets, int64_t n,
                                                          int64_t ntiles, uint64_t *__restrict__ out) {
    constexpr int WAVES = THREADS / 64, R = TILE_WORDS / WAVES;
    __shared__ uint64_t stage[TILE_ROWS];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int64_t nwords = (n + 63) / 64, last = n - 1;
    struct Regs {
        uint64_t v[R];
        uint64_t my_word;
    };
    auto next_tile = [&](int64_t t) {
        while (t < ntiles && tile_offsets[t + 1] == tile_offsets[t]) t += gridDim.x;
        return t;
    };
    auto load = [&](Regs &r, int64_t tile) {
        const int64_t w = tile * TILE_WORDS + lane;
        r.my_word = w < nwords ? keep[w] : 0;
        const int64_t row0 = (tile * TILE_WORDS + int64_t(wave) * R) * 64 + lane;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t row = row0 + int64_t(k) * 64;
            r.v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]);
        }
    };
    auto process = [&](const Regs &r, Regs &nxt, int64_t tile, int64_t ntile) {
        const uint64_t base = tile_offsets[tile];
        const uint32_t T = uint32_t(tile_offsets[tile + 1] - base);
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(r.my_word)), tot);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint64_t kw = bcast64(r.my_word, wave * R + k);
            const uint32_t off = bcast32(my_off, wave * R + k);
            if ((kw >> lane) & 1) stage[off + __popcll(kw & lanemask_lt())] = r.v[k] + 100;
        }
        __syncthreads();
        if (PF && ntile < ntiles) load(nxt, ntile);
        const int head = int(base & 15);
        for (int j = int(threadIdx.x) - head; j < int(T); j += THREADS)
            if (j >= 0) {
                if (NTS) __builtin_nontemporal_store(stage[j], &out[base + uint64_t(j)]);
                else out[base + uint64_t(j)] = stage[j];
            }
        __syncthreads();
        if (!PF && ntile < ntiles) load(nxt, ntile);
    };
    int64_t tile = next_tile(blockIdx.x);
    if (tile >= ntiles) return;
    Regs A, B;
    load(A, tile);
    for (;;) {
        int64_t nt = next_tile(tile + gridDim.x);
        process(A, B, tile, nt);
        if (nt >= ntiles) break;
        tile = next_tile(nt + gridDim.x);
        process(B, A, nt, tile);
        if (tile >= ntiles) break;
    }
}
// C2: one WAVE per 512-row chunk, staged through the wave's own 4 KB of LDS (no workgroup barrier): whole-wave stores of the chunk's run
template <int NTS>
__global__ void __launch_bounds__(256) compact_wave_staged(const uint64_t *__restrict__ words, const uint64_t *keep, const uint64_t *tile_offsets, int64_t n, int64_t ntiles, uint64_t *out) {
    constexpr int CW = 8, CPT = TILE_WORDS / CW;
    __shared__ uint64_t stage_all[4][512];
    uint64_t *stage = stage_all[threadIdx.x >> 6];
    const int lane = lane_id();
    const int64_t nwords = (n + 63) / 64, last = n - 1, n_chunks = ntiles * CPT;
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t tile = chunk / CPT;
        const int c0 = int(chunk % CPT) * CW;
        const uint64_t base = tile_offsets[tile];
        if (tile_offsets[tile + 1] == base) continue;
        const int64_t w = tile * TILE_WORDS + lane;
        const uint64_t my_word = w < nwords ? keep[w] : 0;
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        const uint32_t first = bcast32(my_off, c0);
        const uint32_t endo = c0 + CW < TILE_WORDS ? bcast32(my_off, c0 + CW) : tot;
        const uint32_t T = endo - first;
        if (T == 0) continue;
        uint64_t v[CW];
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int64_t row = (tile * TILE_WORDS + c0 + k) * 64 + lane;
            v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]);
        }
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const uint64_t kw = bcast64(my_word, c0 + k);
            const uint32_t off = bcast32(my_off, c0 + k) - first;
            if ((kw >> lane) & 1) stage[off + __popcll(kw & lanemask_lt())] = v[k] + 100;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the wave's own LDS writes
        __builtin_amdgcn_wave_barrier();
        const uint64_t ob = base + first;
        const int head = int(ob & 15);
        for (int j = lane - head; j < int(T); j += 64)
            if (j >= 0) {
                if (NTS) __builtin_nontemporal_store(stage[j], &out[ob + uint64_t(j)]);
                else out[ob + uint64_t(j)] = stage[j];
            }
        __builtin_amdgcn_wave_barrier();
    }
}

struct Timer {
    hipEvent_t a, b;
    double total = 0;
    int count = 0;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
};

int main(int argc, char **argv) {
    const int64_t n = 100000000, ntiles = (n + TILE_ROWS - 1) / TILE_ROWS, nwords = (n + 63) / 64;
    uint64_t *ids, *age, *out, *keep, *toff;
    uint32_t *tcounts;
    CK(hipMalloc(&ids, n * 8));
    CK(hipMalloc(&age, n * 8));
    CK(hipMalloc(&out, n * 8));
    CK(hipMalloc(&keep, nwords * 8 + 64));
    CK(hipMalloc(&tcounts, (ntiles + 1) * 4));
    CK(hipMalloc(&toff, (ntiles + 1) * 8));
    const uint64_t limit = n / 2;
    for (int random_ids = 0; random_ids < 2; ++random_ids) {
        fill_kernel<<<2048, 256>>>(ids, age, n, random_ids);
        CK(hipMemset(tcounts, 0, (ntiles + 1) * 4));
        keep_strided<8, 1, 1><<<2048, 256>>>(ids, limit, n, keep, tcounts);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> hc(ntiles + 1);
        CK(hipMemcpy(hc.data(), tcounts, (ntiles + 1) * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> ho(ntiles + 1);
        uint64_t acc = 0;
        for (int64_t t = 0; t < ntiles; ++t) { ho[t] = acc; acc += hc[t]; }
        ho[ntiles] = acc;
        CK(hipMemcpy(toff, ho.data(), (ntiles + 1) * 8, hipMemcpyHostToDevice));
        printf("---- ids %s: %llu rows kept\n", random_ids ? "random" : "sorted", (unsigned long long)acc);
        // reference output of the product form, to check every variant against
        std::vector<uint64_t> ref(acc), got(acc);
        compact_strided<8, 1><<<2048, 256>>>(age, keep, toff, n, ntiles, out);
        CK(hipMemcpy(ref.data(), out, acc * 8, hipMemcpyDeviceToHost));
        struct Var { std::string name; std::function<void()> run; bool is_keep; };
        std::vector<Var> vars;
        auto addk = [&](std::string nm, std::function<void()> f) { vars.push_back({nm, f, true}); };
        auto addc = [&](std::string nm, std::function<void()> f) { vars.push_back({nm, f, false}); };
        for (int bpc : {4, 8, 16}) {
            addk("keep_strided R=8 nt atomic bpc=" + std::to_string(bpc), [=] { hipMemsetAsync(tcounts, 0, (ntiles + 1) * 4, 0); keep_strided<8, 1, 1><<<256 * bpc, 256>>>(ids, limit, n, keep, tcounts); });
            addk("keep_strided R=8 nt NOatomic bpc=" + std::to_string(bpc), [=] { keep_strided<8, 1, 0><<<256 * bpc, 256>>>(ids, limit, n, keep, tcounts); });
            addk("keep_strided R=8 plain atomic bpc=" + std::to_string(bpc), [=] { hipMemsetAsync(tcounts, 0, (ntiles + 1) * 4, 0); keep_strided<8, 0, 1><<<256 * bpc, 256>>>(ids, limit, n, keep, tcounts); });
            addk("keep_strided R=16 nt atomic bpc=" + std::to_string(bpc), [=] { hipMemsetAsync(tcounts, 0, (ntiles + 1) * 4, 0); keep_strided<16, 1, 1><<<256 * bpc, 256>>>(ids, limit, n, keep, tcounts); });
            addk("keep_tile nt bpc=" + std::to_string(bpc), [=] { keep_tile<1><<<256 * bpc, 256>>>(ids, limit, n, ntiles, keep, tcounts); });
            addk("keep_tile plain bpc=" + std::to_string(bpc), [=] { keep_tile<0><<<256 * bpc, 256>>>(ids, limit, n, ntiles, keep, tcounts); });
        }
        for (int bpc : {4, 8, 16}) {
            addc("compact_strided CW=8 nts bpc=" + std::to_string(bpc), [=] { compact_strided<8, 1><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_strided CW=8 plain-st bpc=" + std::to_string(bpc), [=] { compact_strided<8, 0><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_strided CW=16 nts bpc=" + std::to_string(bpc), [=] { compact_strided<16, 1><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_strided CW=64 nts bpc=" + std::to_string(bpc), [=] { compact_strided<64, 1><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_wave_staged nts bpc=" + std::to_string(bpc), [=] { compact_wave_staged<1><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_wave_staged plain-st bpc=" + std::to_string(bpc), [=] { compact_wave_staged<0><<<256 * bpc, 256>>>(age, keep, toff, n, ntiles, out); });
        }
        for (int wgs : {2, 4, 6, 8}) {
            addc("compact_staged 256thr pf nts wgs=" + std::to_string(wgs), [=] { compact_staged<256, 1, 1><<<256 * wgs, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_staged 256thr pf plain-st wgs=" + std::to_string(wgs), [=] { compact_staged<256, 1, 0><<<256 * wgs, 256>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_staged 256thr nopf nts wgs=" + std::to_string(wgs), [=] { compact_staged<256, 0, 1><<<256 * wgs, 256>>>(age, keep, toff, n, ntiles, out); });
        }
        for (int wgs : {2, 3, 4}) {
            addc("compact_staged 512thr pf nts wgs=" + std::to_string(wgs), [=] { compact_staged<512, 1, 1><<<256 * wgs, 512>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_staged 512thr pf plain-st wgs=" + std::to_string(wgs), [=] { compact_staged<512, 1, 0><<<256 * wgs, 512>>>(age, keep, toff, n, ntiles, out); });
            addc("compact_staged 512thr nopf nts wgs=" + std::to_string(wgs), [=] { compact_staged<512, 0, 1><<<256 * wgs, 512>>>(age, keep, toff, n, ntiles, out); });
        }
        for (int wgs : {1, 2}) addc("compact_staged 1024thr pf nts wgs=" + std::to_string(wgs), [=] { compact_staged<1024, 1, 1><<<256 * wgs, 1024>>>(age, keep, toff, n, ntiles, out); });
        // every compaction variant: correct?
        for (auto &v : vars) {
            if (v.is_keep) continue;
            CK(hipMemset(out, 0xEE, acc * 8));
            v.run();
            CK(hipMemcpy(got.data(), out, acc * 8, hipMemcpyDeviceToHost));
            if (got != ref) printf("MISMATCH: %s\n", v.name.c_str());
        }
        // timing: pairs (keep variant i, compaction variant j) alternate so that neither finds its input in the caches; each variant
        // is timed against a fixed partner (the product's)
        std::vector<Timer> tm(vars.size());
        const int reps = 12;
        auto partner_keep = [&] { hipMemsetAsync(tcounts, 0, (ntiles + 1) * 4, 0); keep_strided<8, 1, 1><<<2048, 256>>>(ids, limit, n, keep, tcounts); };
        auto partner_compact = [&] { compact_strided<8, 1><<<2048, 256>>>(age, keep, toff, n, ntiles, out); };
        for (size_t i = 0; i < vars.size(); ++i) {
            for (int r = 0; r < reps + 2; ++r) {
                if (vars[i].is_keep) {
                    CK(hipEventRecord(tm[i].a));
                    vars[i].run();
                    CK(hipEventRecord(tm[i].b));
                    partner_compact();
                } else {
                    partner_keep();
                    CK(hipEventRecord(tm[i].a));
                    vars[i].run();
                    CK(hipEventRecord(tm[i].b));
                }
                CK(hipEventSynchronize(tm[i].b));
                float ms;
                CK(hipEventElapsedTime(&ms, tm[i].a, tm[i].b));
                if (r >= 2) { tm[i].total += ms; tm[i].count++; }
            }
            // the keep variants must leave a correct mask behind for the next compaction
            if (vars[i].is_keep) { hipMemsetAsync(tcounts, 0, (ntiles + 1) * 4, 0); keep_strided<8, 1, 1><<<2048, 256>>>(ids, limit, n, keep, tcounts); }
            const double ms = tm[i].total / tm[i].count;
            const double bytes = vars[i].is_keep ? 8.0 * n : (random_ids ? 8.0 * n : 4.0 * n) + 8.0 * acc;
            printf("%-52s %.4f ms  %.0f GB/s\n", vars[i].name.c_str(), ms, bytes / ms / 1e6);
        }
    }
    return 0;
}
