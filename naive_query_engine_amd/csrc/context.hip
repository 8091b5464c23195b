// context.hip — contexts, the pooled device allocator, tables (device RecordBatches) and the
// small table-level C-ABI entry points (create / release / download / project / slice / concat /
// take / synth).  No operator logic here.
#include <algorithm>
#include <mutex>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

void AggSwitches::read_environment() {
    const char *e = getenv("NQE_TINY_UNPACK_TILES");
    tiny_unpack_tiles = std::min(4096, std::max(1, e ? atoi(e) : 4096));
    debug = getenv("NQE_DEBUG") != nullptr;
}

namespace nqe {


static std::mutex g_err_mu;
static std::string g_global_error;
void set_global_error(const std::string &msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_global_error = msg;
}

// ---------------------------------------------------------------- code objects
static std::vector<const void *> &module_probes() {
    static std::vector<const void *> v; // constructed on first use: registrations run during static initialisation, in any order
    return v;
}
void register_module_probe(const void *kernel) {
    if (kernel) module_probes().push_back(kernel);
}
// loads every translation unit's code object for the current device (see nqe_internal.hpp)
static void load_modules() {
    if (getenv("NQE_LAZY_MODULES")) return;
    for (const void *k : module_probes()) {
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, k) != hipSuccess) (void)hipGetLastError(); // the first launch reports a real problem
    }
}

// ---------------------------------------------------------------- allocator
static size_t round_capacity(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= (1u << 20)) { // next power of two up to 1 MiB
        size_t c = 256;
        while (c < bytes) c <<= 1;
        return c;
    }
    const size_t g = size_t(2) << 20; // 2 MiB granules above that
    return (bytes + g - 1) / g * g;
}

void pool_trim(nqe_ctx *ctx) {
    for (auto &kv : ctx->pool) (void)hipFree(kv.second);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}

// best fit in the reserved block; null when nothing fits
static void *arena_alloc(nqe_ctx *ctx, size_t cap) {
    if (!ctx->arena_base || cap > ctx->arena_free_bytes) return nullptr;
    auto best = ctx->arena_free.end();
    for (auto it = ctx->arena_free.begin(); it != ctx->arena_free.end(); ++it)
        if (it->second >= cap && (best == ctx->arena_free.end() || it->second < best->second)) best = it;
    if (best == ctx->arena_free.end()) return nullptr;
    const size_t off = best->first, len = best->second;
    ctx->arena_free.erase(best);
    if (len > cap) ctx->arena_free.emplace(off + cap, len - cap);
    ctx->arena_free_bytes -= cap;
    return ctx->arena_base + off;
}
static void arena_release(nqe_ctx *ctx, void *ptr, size_t cap) {
    size_t off = size_t(static_cast<char *>(ptr) - ctx->arena_base), len = cap;
    auto next = ctx->arena_free.lower_bound(off);
    if (next != ctx->arena_free.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) { // coalesce with the free range below
            off = prev->first;
            len += prev->second;
            ctx->arena_free.erase(prev);
        }
    }
    if (next != ctx->arena_free.end() && off + len == next->first) { // … and above
        len += next->second;
        ctx->arena_free.erase(next);
    }
    ctx->arena_free.emplace(off, len);
    ctx->arena_free_bytes += cap;
}

BufRef dev_alloc(nqe_ctx *ctx, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    b->ctx = ctx;
    b->bytes = bytes;
    b->owned = true;
    b->lib_memory = true;
    size_t cap = round_capacity(bytes);
    auto it = ctx->pool.lower_bound(cap);
    if (it != ctx->pool.end() && it->first <= cap + cap / 4) {
        b->ptr = it->second;
        b->capacity = it->first;
        ctx->pool_bytes -= it->first;
        ctx->pool.erase(it);
    } else if (void *ap = arena_alloc(ctx, cap)) {
        b->ptr = ap;
        b->capacity = cap;
        b->in_arena = true;
    } else {
        hipError_t e = hipMalloc(&b->ptr, cap);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            NQE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            pool_trim(ctx);
            e = hipMalloc(&b->ptr, cap);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(e == hipErrorOutOfMemory ? NQE_ERR_OUT_OF_MEMORY : NQE_ERR_HIP,
                 std::string("hipMalloc(") + std::to_string(cap) + "): " + hipGetErrorString(e));
        }
        b->capacity = cap;
    }
    ctx->live_bytes += b->capacity;
    return b;
}

BufRef dev_alloc_zero(nqe_ctx *ctx, size_t bytes) {
    BufRef b = dev_alloc(ctx, bytes);
    if (bytes) NQE_HIP_CHECK(hipMemsetAsync(b->ptr, 0, bytes, ctx->stream));
    return b;
}

BufRef dev_borrow(nqe_ctx *ctx, const void *ptr, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    b->ctx = ctx;
    b->ptr = const_cast<void *>(ptr);
    b->bytes = bytes;
    b->capacity = bytes;
    b->owned = false;
    return b;
}

BufRef dev_view(const BufRef &parent, size_t offset, size_t bytes) {
    DevBuf *v = new DevBuf();
    v->ctx = parent->ctx;
    v->ptr = static_cast<char *>(parent->ptr) + offset;
    v->bytes = bytes;
    v->capacity = bytes;
    v->owned = false;
    v->lib_memory = parent->lib_memory;
    v->caller_immutable = parent->caller_immutable;
    return BufRef(v, [parent](DevBuf *p) { delete p; });
}

DevBuf::~DevBuf() {
    if (owned && ptr && ctx) {
        // stream-ordered reuse: every consumer of this block was enqueued on ctx->stream before
        // any later allocation's first use, so handing it back to the pool is safe.
        ctx->live_bytes -= capacity;
        if (in_arena) {
            arena_release(ctx, ptr, capacity);
            return;
        }
        ctx->pool.emplace(capacity, ptr);
        ctx->pool_bytes += capacity;
    }
}

// ---------------------------------------------------------------- flags
void flags_reset(nqe_ctx *ctx) {
    if (ctx->flags_clean) return;
    NQE_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, sizeof(int) * NQE_NUM_FLAGS, ctx->stream));
    ctx->flags_clean = true;
}
static void flags_note(nqe_ctx *ctx, const int *f, bool count_on_device) {
    bool zero = true;
    for (int i = 0; i < NQE_NUM_FLAGS; ++i) zero = zero && (f[i] == 0 || (!count_on_device && i == NQE_FLAG_GROUP_COUNT));
    ctx->flags_clean = zero; // every kernel launched so far has completed
}
void flags_read(nqe_ctx *ctx, int out[NQE_NUM_FLAGS]) {
    NQE_HIP_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, sizeof(int) * NQE_NUM_FLAGS, hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    for (int i = 0; i < NQE_NUM_FLAGS; ++i) out[i] = ctx->h_flags[i];
    flags_note(ctx, out, true);
}
void flags_read_mirrored(nqe_ctx *ctx, int out[NQE_NUM_FLAGS]) {
    sync(ctx);
    for (int i = 0; i < NQE_NUM_FLAGS; ++i) out[i] = ctx->h_flags[i];
    flags_note(ctx, out, false); // the group count went to the mirror only; its device slot is still zero
}
void throw_on_flags(nqe_ctx *ctx) {
    int f[NQE_NUM_FLAGS];
    flags_read(ctx, f);
    if (f[NQE_FLAG_DIV_ZERO]) fail(NQE_ERR_ARROW, "Divide by zero");
    if (f[NQE_FLAG_OVERFLOW]) fail(NQE_ERR_ARROW, "attempt to divide with overflow");
    if (f[NQE_FLAG_OOB]) fail(NQE_ERR_ARROW, "take index out of bounds");
    if (f[NQE_FLAG_TABLE_FULL]) fail(NQE_ERR_OUT_OF_MEMORY, "device hash table overflow");
}

// ---------------------------------------------------------------- small kernels
__global__ void synth_kernel(int kind, uint64_t seed, int64_t first_row, int64_t n, uint64_t modulus, int64_t base,
                             uint64_t *out) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        uint64_t i = uint64_t(first_row + r);
        uint64_t v;
        if (kind == NQE_SYNTH_ROWID) v = i;
        else if (kind == NQE_SYNTH_UNIFORM) v = uint64_t(int64_t(splitmix64(seed + i) % modulus) + base);
        else v = d2u(double(splitmix64(seed + i) >> 11) * 0x1.0p-53 * 100.0);
        out[r] = v;
    }
}

// out row j <- bit of src row (j + src_off); one u64 word per wave via ballot
__global__ void bitmap_slice_kernel(const uint8_t *src, int64_t src_off, uint64_t *dst, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t n_pad = (n + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
        bool b = j < n ? (src ? get_bit(src, j + src_off) : true) : false;
        uint64_t w = __ballot(b);
        if (lane_id() == 0) dst[j >> 6] = w;
    }
}

// dst bits [dst_off, dst_off+n) |= src bits [0, n) (src null ⇒ ones); dst pre-zeroed, word padded
__global__ void bitmap_place_kernel(const uint8_t *src, uint64_t *dst, int64_t dst_off, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t n_pad = (n + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
        bool b = j < n ? (src ? get_bit(src, j) : true) : false;
        uint64_t w = __ballot(b);
        if (lane_id() == 0 && w) {
            int64_t p = dst_off + (j & ~63ll);
            int sh = int(p & 63);
            atomicOr((unsigned long long *)&dst[p >> 6], (unsigned long long)(w << sh));
            if (sh) atomicOr((unsigned long long *)&dst[(p >> 6) + 1], (unsigned long long)(w >> (64 - sh)));
        }
    }
}

__global__ void take_words_kernel(const uint64_t *src, const uint8_t *src_valid, int64_t src_len, const int64_t *idx,
                                  int64_t m, uint64_t *out, uint64_t *out_valid, int *flags) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t m_pad = (m + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < m_pad; j += stride) {
        bool ok = false;
        if (j < m) {
            int64_t i = idx[j];
            if (i < 0 || i >= src_len) {
                atomicOr(&flags[NQE_FLAG_OOB], 1);
                out[j] = 0;
            } else {
                ok = src_valid ? get_bit(src_valid, i) : true;
                out[j] = ok ? src[i] : 0;
            }
        }
        if (out_valid) {
            uint64_t w = __ballot(ok);
            if (lane_id() == 0) out_valid[j >> 6] = w;
        }
    }
}

__global__ void take_bits_kernel(const uint8_t *src, const uint8_t *src_valid, int64_t src_len, const int64_t *idx,
                                 int64_t m, uint64_t *out, uint64_t *out_valid, int *flags) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t m_pad = (m + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < m_pad; j += stride) {
        bool ok = false, bit = false;
        if (j < m) {
            int64_t i = idx[j];
            if (i < 0 || i >= src_len) atomicOr(&flags[NQE_FLAG_OOB], 1);
            else {
                ok = src_valid ? get_bit(src_valid, i) : true;
                bit = ok && get_bit(src, i);
            }
        }
        uint64_t w = __ballot(bit);
        uint64_t v = __ballot(ok);
        if (lane_id() == 0) {
            out[j >> 6] = w;
            if (out_valid) out_valid[j >> 6] = v;
        }
    }
}

// ---- Utf8 take: lengths → exclusive scan (= int32 offsets) → byte copy.  idx < 0 emits NULL when allowed
// (rows a NULL predicate emits, quirk Q4), otherwise it is out of bounds like any idx >= src_len.
__global__ void utf8_take_lengths_kernel(const int32_t *src_off, const uint8_t *src_valid, int64_t src_len, const int64_t *idx, int64_t m,
                                         int allow_null_idx, uint32_t *lens, uint64_t *out_valid, int *flags) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t m_pad = (m + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < m_pad; j += stride) {
        bool ok = false;
        uint32_t len = 0;
        if (j < m) {
            int64_t i = idx[j];
            if (i < 0 && allow_null_idx) ok = false;
            else if (i < 0 || i >= src_len) atomicOr(&flags[NQE_FLAG_OOB], 1);
            else {
                ok = src_valid ? get_bit(src_valid, i) : true;
                len = ok ? uint32_t(src_off[i + 1] - src_off[i]) : 0u;
            }
            lens[j] = len;
        }
        if (j == m) lens[m] = 0; // the scan leaves the total here
        if (out_valid) {
            uint64_t w = __ballot(ok);
            if (lane_id() == 0) out_valid[j >> 6] = w;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (m & 63) == 0) lens[m] = 0;
}

// 2^LG lanes per output string, 64 >> LG strings per wave and step: a whole wave per string left 50-60 of its lanes idle on the 5-20
// byte strings of names and keys (10^7 strings per column took 1.6 ms); the host picks LG from the mean output length.  The lanes of
// a group read the same index / offsets (one broadcast load) and stride over the string's bytes.
template <int LG>
__global__ void __launch_bounds__(256) utf8_take_copy_kernel(const int32_t *src_off, const uint8_t *src_data, const uint8_t *src_valid, int64_t src_len,
                                                             const int64_t *idx, int64_t m, const uint32_t *out_off, uint8_t *out_data) {
    constexpr int G = 1 << LG, PER_WAVE = 64 >> LG;
    const int64_t waves = (int64_t(gridDim.x) * blockDim.x) >> 6, wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint32_t sub = uint32_t(lane_id()) & uint32_t(G - 1), grp = uint32_t(lane_id()) >> LG;
    for (int64_t j0 = wave * PER_WAVE; j0 < m; j0 += waves * PER_WAVE) {
        const int64_t j = j0 + grp;
        if (j >= m) continue;
        const int64_t i = idx[j];
        if (i < 0 || i >= src_len) continue;
        if (src_valid && !get_bit(src_valid, i)) continue;
        const int32_t so = src_off[i];
        const uint32_t d = out_off[j], len = out_off[j + 1] - d;
        // whole 8-byte words first (unaligned 8-byte loads and stores: fine for global memory on gfx9 and later), then the tail's bytes
        typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
        const uint32_t nw = len >> 3;
        for (uint32_t w = sub; w < nw; w += G)
            *reinterpret_cast<u64_unaligned *>(out_data + d + 8 * w) = *reinterpret_cast<const u64_unaligned *>(src_data + so + 8 * w);
        for (uint32_t b = (nw << 3) + sub; b < len; b += G) out_data[d + b] = src_data[so + b];
    }
}

__global__ void iota_i64_kernel(int64_t *out, int64_t first, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = first + i;
}

__global__ void rebase_offsets_kernel(const int32_t *src_off, int64_t n, int32_t delta, int32_t *dst_off) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i <= n; i += stride) dst_off[i] = src_off[i] - src_off[0] + delta;
}

__global__ void pack_bytes_kernel(const uint8_t *bytes, int64_t n, uint64_t *words) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int64_t n_pad = (n + 63) / 64 * 64;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
        bool b = j < n && bytes[j] != 0;
        uint64_t w = __ballot(b);
        if (lane_id() == 0) words[j >> 6] = w;
    }
}

void pack_bytes_to_bits(nqe_ctx *ctx, const uint8_t *bytes, int64_t n, uint64_t *bitmap_words) {
    if (n == 0) return;
    launch(ctx, "pack_bytes", pack_bytes_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, bytes, n, bitmap_words);
}

// ---------------------------------------------------------------- column helpers
DevColumn make_word_column(nqe_ctx *ctx, int dtype, int64_t n, bool with_validity) {
    DevColumn c;
    c.dtype = dtype;
    c.length = n;
    c.values = dev_alloc(ctx, size_t(n) * 8);
    if (with_validity) {
        c.validity = dev_alloc(ctx, bitmap_alloc_bytes(n));
        c.null_count = -1;
    }
    return c;
}
DevColumn make_bool_column(nqe_ctx *ctx, int64_t n, bool with_validity) {
    DevColumn c;
    c.dtype = NQE_BOOLEAN;
    c.length = n;
    c.values = dev_alloc(ctx, bitmap_alloc_bytes(n));
    if (with_validity) {
        c.validity = dev_alloc(ctx, bitmap_alloc_bytes(n));
        c.null_count = -1;
    }
    return c;
}

DevColumn take_utf8(nqe_ctx *ctx, const DevColumn &src, const int64_t *idx, int64_t m, bool allow_null_idx) {
    DevColumn out;
    out.dtype = NQE_UTF8;
    out.length = m;
    const bool need_valid = src.validity != nullptr || allow_null_idx;
    out.values = dev_alloc(ctx, size_t(m + 1) * 4 + 8); // lengths, then (after the scan) int32 offsets
    if (need_valid) {
        out.validity = dev_alloc(ctx, bitmap_alloc_bytes(m));
        out.null_count = -1;
    }
    launch(ctx, "utf8_take_lengths", utf8_take_lengths_kernel, dim3(stream_grid(ctx, m + 1, 256)), dim3(256), 0,
           (const int32_t *)src.values->ptr, src.valid(), src.length, idx, m, allow_null_idx ? 1 : 0, (uint32_t *)out.values->ptr,
           need_valid ? (uint64_t *)out.validity->ptr : nullptr, ctx->d_flags);
    exclusive_scan_u32_inplace(ctx, (uint32_t *)out.values->ptr, m + 1);
    uint32_t total = read_scalar(ctx, (const uint32_t *)out.values->ptr + m);
    if (total >= 0x80000000u) fail(NQE_ERR_ARROW, "Utf8 take: offsets overflow int32");
    out.data_length = total;
    out.data = dev_alloc(ctx, size_t(total) + 8);
    if (m && total) {
        const uint64_t mean = uint64_t(total) / uint64_t(m);
        const int lg = mean <= 32 ? 2 : mean <= 96 ? 3 : mean <= 256 ? 4 : mean <= 1024 ? 5 : 6; // lanes per string (a lane moves 8 bytes per step)
        auto k = lg == 2 ? utf8_take_copy_kernel<2> : lg == 3 ? utf8_take_copy_kernel<3> : lg == 4 ? utf8_take_copy_kernel<4> : lg == 5 ? utf8_take_copy_kernel<5>
                                                                                                                                     : utf8_take_copy_kernel<6>;
        launch(ctx, "utf8_take_copy", k, dim3(stream_grid(ctx, (m + (64 >> lg) - 1) / (64 >> lg), 4)), dim3(256), 0, (const int32_t *)src.values->ptr,
               src.data ? (const uint8_t *)src.data->ptr : nullptr, src.valid(), src.length, idx, m, (const uint32_t *)out.values->ptr,
               (uint8_t *)out.data->ptr);
    }
    return out;
}

BufRef iota_i64(nqe_ctx *ctx, int64_t first, int64_t n) {
    BufRef b = dev_alloc(ctx, size_t(n) * 8 + 8);
    if (n) launch(ctx, "iota_i64", iota_i64_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, (int64_t *)b->ptr, first, n);
    return b;
}

DevColumn take_column(nqe_ctx *ctx, const DevColumn &src, const int64_t *idx, int64_t m) {
    bool v = src.validity != nullptr;
    if (is_word_type(src.dtype)) {
        DevColumn out = make_word_column(ctx, src.dtype, m, v);
        if (m)
            launch(ctx, "take_words", take_words_kernel, dim3(stream_grid(ctx, m, 256)), dim3(256), 0, src.words(),
                   src.valid(), src.length, idx, m, (uint64_t *)out.values->ptr,
                   v ? (uint64_t *)out.validity->ptr : nullptr, ctx->d_flags);
        return out;
    }
    if (src.dtype == NQE_BOOLEAN) {
        DevColumn out = make_bool_column(ctx, m, v);
        if (m)
            launch(ctx, "take_bits", take_bits_kernel, dim3(stream_grid(ctx, m, 256)), dim3(256), 0, src.bits(),
                   src.valid(), src.length, idx, m, (uint64_t *)out.values->ptr,
                   v ? (uint64_t *)out.validity->ptr : nullptr, ctx->d_flags);
        return out;
    }
    if (src.dtype == NQE_UTF8) return take_utf8(ctx, src, idx, m, false);
    fail(NQE_ERR_NOT_SUPPORTED, "take: unsupported column type");
}

static BufRef slice_bitmap(nqe_ctx *ctx, const uint8_t *src, int64_t off, int64_t len) {
    BufRef b = dev_alloc(ctx, bitmap_alloc_bytes(len));
    if (len)
        launch(ctx, "bitmap_slice", bitmap_slice_kernel, dim3(stream_grid(ctx, len, 256)), dim3(256), 0, src, off,
               (uint64_t *)b->ptr, len);
    return b;
}

DevColumn slice_column(nqe_ctx *ctx, const DevColumn &src, int64_t off, int64_t len) {
    DevColumn out;
    out.dtype = src.dtype;
    out.length = len;
    if (is_word_type(src.dtype)) {
        out.values = dev_alloc(ctx, size_t(len) * 8);
        if (len)
            NQE_HIP_CHECK(hipMemcpyAsync(out.values->ptr, src.words() + off, size_t(len) * 8, hipMemcpyDeviceToDevice,
                                         ctx->stream));
    } else if (src.dtype == NQE_BOOLEAN) {
        out.values = slice_bitmap(ctx, src.bits(), off, len);
    } else {
        if (src.dtype != NQE_UTF8) fail(NQE_ERR_NOT_SUPPORTED, "slice: unsupported column type");
        BufRef idx = iota_i64(ctx, off, len);
        return take_utf8(ctx, src, (const int64_t *)idx->ptr, len, false);
    }
    if (src.validity) {
        out.validity = slice_bitmap(ctx, src.valid(), off, len);
        out.null_count = -1;
    }
    return out;
}

void bitmap_place(nqe_ctx *ctx, const uint8_t *src, uint64_t *dst, int64_t dst_off, int64_t n) {
    if (n > 0) launch(ctx, "bitmap_place", bitmap_place_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, src, dst, dst_off, n);
}

void utf8_rebase_offsets(nqe_ctx *ctx, const int32_t *src, int64_t n, int32_t delta, int32_t *dst) {
    launch(ctx, "utf8_rebase", rebase_offsets_kernel, dim3(stream_grid(ctx, n + 1, 256)), dim3(256), 0, src, n, delta, dst);
}

DevColumn concat_columns(nqe_ctx *ctx, const std::vector<const DevColumn *> &parts) {
    DevColumn out;
    out.dtype = parts[0]->dtype;
    int64_t total = 0;
    bool any_valid = false;
    for (auto *p : parts) {
        if (p->dtype != out.dtype) fail(NQE_ERR_ARROW, "concat: column types differ");
        total += p->length;
        any_valid |= p->validity != nullptr;
    }
    out.length = total;
    if (is_word_type(out.dtype)) {
        out.values = dev_alloc(ctx, size_t(total) * 8);
        int64_t off = 0;
        for (auto *p : parts) {
            if (p->length)
                NQE_HIP_CHECK(hipMemcpyAsync((uint64_t *)out.values->ptr + off, p->words(), size_t(p->length) * 8,
                                             hipMemcpyDeviceToDevice, ctx->stream));
            off += p->length;
        }
    } else if (out.dtype == NQE_BOOLEAN) {
        out.values = dev_alloc_zero(ctx, bitmap_alloc_bytes(total) + 8);
        int64_t off = 0;
        for (auto *p : parts) {
            if (p->length)
                launch(ctx, "bitmap_place", bitmap_place_kernel, dim3(stream_grid(ctx, p->length, 256)), dim3(256), 0,
                       p->bits(), (uint64_t *)out.values->ptr, off, p->length);
            off += p->length;
        }
    } else {
        if (out.dtype != NQE_UTF8) fail(NQE_ERR_NOT_SUPPORTED, "concat: unsupported column type");
        // offsets are rebased part by part; the byte ranges are copied back to back
        std::vector<int32_t> first(parts.size()), lastv(parts.size());
        for (size_t k = 0; k < parts.size(); ++k) {
            first[k] = lastv[k] = 0;
            if (parts[k]->length) {
                first[k] = read_scalar(ctx, (const int32_t *)parts[k]->values->ptr);
                lastv[k] = read_scalar(ctx, (const int32_t *)parts[k]->values->ptr + parts[k]->length);
            }
        }
        int64_t bytes = 0;
        for (size_t k = 0; k < parts.size(); ++k) bytes += lastv[k] - first[k];
        if (bytes >= (int64_t(1) << 31)) fail(NQE_ERR_ARROW, "Utf8 concat: offsets overflow int32");
        out.values = dev_alloc_zero(ctx, size_t(total + 1) * 4 + 8);
        out.data = dev_alloc(ctx, size_t(bytes) + 8);
        out.data_length = bytes;
        int64_t row = 0, pos = 0;
        for (size_t k = 0; k < parts.size(); ++k) {
            const DevColumn *p = parts[k];
            if (p->length == 0) continue;
            launch(ctx, "utf8_rebase", rebase_offsets_kernel, dim3(stream_grid(ctx, p->length + 1, 256)), dim3(256), 0,
                   (const int32_t *)p->values->ptr, p->length, int32_t(pos), (int32_t *)out.values->ptr + row);
            int64_t nb = lastv[k] - first[k];
            if (nb)
                NQE_HIP_CHECK(hipMemcpyAsync((uint8_t *)out.data->ptr + pos, (const uint8_t *)p->data->ptr + first[k], size_t(nb),
                                             hipMemcpyDeviceToDevice, ctx->stream));
            row += p->length;
            pos += nb;
        }
    }
    if (any_valid) {
        out.validity = dev_alloc_zero(ctx, bitmap_alloc_bytes(total) + 8);
        out.null_count = -1;
        int64_t off = 0;
        for (auto *p : parts) {
            if (p->length)
                launch(ctx, "bitmap_place", bitmap_place_kernel, dim3(stream_grid(ctx, p->length, 256)), dim3(256), 0,
                       p->valid(), (uint64_t *)out.validity->ptr, off, p->length);
            off += p->length;
        }
    }
    return out;
}

static size_t values_bytes(int dtype, int64_t n) {
    if (is_word_type(dtype)) return size_t(n) * 8;
    if (dtype == NQE_BOOLEAN) return bitmap_bytes(n);
    if (dtype == NQE_UTF8) return size_t(n + 1) * 4;
    return 0;
}

} // namespace nqe

using namespace nqe;

extern "C" {

uint32_t nqe_abi_version(void) { return NQE_ABI_VERSION; }

const char *nqe_last_global_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_global_error;
    return copy.c_str();
}

nqe_status nqe_ctx_create(int32_t device, void *stream, nqe_ctx **out) {
    NQE_API_BEGIN(nullptr)
    if (!out) fail(NQE_ERR_INVALID_ARGUMENT, "out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        fail(NQE_ERR_HIP, "no HIP device available (libnqe_hip needs an MI355X; there is no CPU fallback)");
    }
    if (device < 0 || device >= count) fail(NQE_ERR_INVALID_ARGUMENT, "device index out of range");
    NQE_HIP_CHECK(hipSetDevice(device));
    auto ctx = std::make_unique<nqe_ctx>();
    ctx->device = device;
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        NQE_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    NQE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.sharedMemPerBlock > 0) ctx->lds_per_block = size_t(prop.sharedMemPerBlock);
    if (prop.gcnArchName[0]) { // "gfx950:sramecc+:xnack-" -> "gfx950"
        ctx->arch = prop.gcnArchName;
        ctx->arch = ctx->arch.substr(0, ctx->arch.find(':'));
    }
    NQE_HIP_CHECK(hipMalloc(&ctx->d_flags, sizeof(int) * NQE_NUM_FLAGS));
    NQE_HIP_CHECK(hipHostMalloc(&ctx->h_flags, sizeof(int) * (NQE_NUM_FLAGS + NQE_FLAG_MIRROR_EXTRA), hipHostMallocMapped | hipHostMallocCoherent));
    NQE_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_flags_dev), ctx->h_flags, 0));
    NQE_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, sizeof(int) * NQE_NUM_FLAGS, ctx->stream));
    ctx->agg_sw.read_environment();
    load_modules();
    *out = ctx.release();
    if (const char *mb = getenv("NQE_RESERVE_MB")) {
        const long long n = atoll(mb);
        if (n > 0 && nqe_ctx_reserve(*out, size_t(n) << 20) != NQE_OK) { // (a failed reservation is not a failed context: the pool takes over)
        }
    }
    NQE_API_END()
}

nqe_status nqe_ctx_destroy(nqe_ctx *ctx) {
    if (!ctx) return NQE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->timings) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    pool_trim(ctx);
    if (ctx->arena_base) (void)hipFree(ctx->arena_base);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NQE_OK;
}

nqe_status nqe_ctx_synchronize(nqe_ctx *ctx) {
    NQE_API_BEGIN(ctx)
    sync(ctx);
    NQE_API_END()
}

nqe_status nqe_ctx_memory_stats(nqe_ctx *ctx, int64_t *live_bytes, int64_t *pooled_bytes) {
    NQE_API_BEGIN(ctx)
    if (!ctx) fail(NQE_ERR_INVALID_ARGUMENT, "null context");
    if (live_bytes) *live_bytes = int64_t(ctx->live_bytes);
    if (pooled_bytes) *pooled_bytes = int64_t(ctx->pool_bytes + ctx->arena_free_bytes);
    NQE_API_END()
}

nqe_status nqe_ctx_reserve(nqe_ctx *ctx, size_t bytes) {
    NQE_API_BEGIN(ctx)
    if (!ctx) fail(NQE_ERR_INVALID_ARGUMENT, "null context");
    if (ctx->arena_base) fail(NQE_ERR_INVALID_ARGUMENT, "the context has a reserved block already");
    if (bytes == 0) return NQE_OK;
    const size_t g = size_t(2) << 20;
    bytes = (bytes + g - 1) / g * g;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(e == hipErrorOutOfMemory ? NQE_ERR_OUT_OF_MEMORY : NQE_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    // touch every page once (the driver maps lazily: the first kernel to write a fresh block would pay for it)
    NQE_HIP_CHECK(hipMemsetAsync(p, 0, bytes, ctx->stream));
    sync(ctx);
    ctx->arena_base = static_cast<char *>(p);
    ctx->arena_size = ctx->arena_free_bytes = bytes;
    ctx->arena_free.emplace(size_t(0), bytes);
    NQE_API_END()
}

nqe_status nqe_ctx_trim(nqe_ctx *ctx) {
    NQE_API_BEGIN(ctx)
    if (!ctx) fail(NQE_ERR_INVALID_ARGUMENT, "null context");
    sync(ctx);
    pool_trim(ctx);
    NQE_API_END()
}

const char *nqe_last_error(const nqe_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

nqe_status nqe_ctx_timing_enable(nqe_ctx *ctx, int32_t enable) {
    NQE_API_BEGIN(ctx)
    ctx->timing = enable != 0;
    NQE_API_END()
}
nqe_status nqe_ctx_timing_reset(nqe_ctx *ctx) {
    NQE_API_BEGIN(ctx)
    sync(ctx);
    for (auto &t : ctx->timings) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    ctx->timings.clear();
    NQE_API_END()
}
nqe_status nqe_ctx_timing_query(nqe_ctx *ctx, const char *name_substr, double *total_ms, int64_t *launches) {
    NQE_API_BEGIN(ctx)
    sync(ctx);
    double tot = 0;
    int64_t cnt = 0;
    for (auto &t : ctx->timings) {
        if (name_substr && *name_substr && t.name.find(name_substr) == std::string::npos) continue;
        float ms = 0;
        NQE_HIP_CHECK(hipEventElapsedTime(&ms, t.start, t.stop));
        tot += ms;
        ++cnt;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    NQE_API_END()
}

// every kernel family launched since the last reset, by exact name: "name\tms\tlaunches\n" per line (sorted by name)
nqe_status nqe_ctx_jit_wait(nqe_ctx *ctx) {
    NQE_API_BEGIN(ctx)
    if (!ctx) fail(NQE_ERR_INVALID_ARGUMENT, "null context");
    jit_wait(ctx);
    NQE_API_END()
}

nqe_status nqe_ctx_timing_report(nqe_ctx *ctx, char *buf, int64_t capacity, int64_t *needed) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !needed || capacity < 0 || (capacity > 0 && !buf)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    sync(ctx);
    std::map<std::string, std::pair<double, int64_t>> acc;
    for (auto &t : ctx->timings) {
        float ms = 0;
        NQE_HIP_CHECK(hipEventElapsedTime(&ms, t.start, t.stop));
        auto &a = acc[t.name];
        a.first += ms;
        a.second += 1;
    }
    std::string out;
    char line[256];
    for (auto &kv : acc) {
        snprintf(line, sizeof(line), "%s\t%.6f\t%lld\n", kv.first.c_str(), kv.second.first, (long long)kv.second.second);
        out += line;
    }
    *needed = int64_t(out.size()) + 1;
    if (capacity >= *needed) std::memcpy(buf, out.c_str(), out.size() + 1);
    NQE_API_END()
}

// ---------------------------------------------------------------- tables
namespace {
constexpr int PACK_MAX_COLS = 48, UNPACK_MAX_PARTS = 64;
struct PackArgs {
    const uint64_t *src[PACK_MAX_COLS];
    int64_t rows, stride;
    int32_t ncols, pad;
};
__global__ void __launch_bounds__(256) pack_words_kernel(PackArgs pa, uint64_t *dst) {
    const int64_t total = pa.rows * pa.ncols;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int c = int(i / pa.rows);
        const int64_t r = i - int64_t(c) * pa.rows;
        dst[int64_t(c) * pa.stride + r] = pa.src[c][r];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dst[int64_t(pa.ncols) * pa.stride] = uint64_t(pa.rows);
}
struct UnpackArgs {
    int64_t offset[UNPACK_MAX_PARTS + 1]; // output row of part p's first row; [nparts] = total
    int64_t stride;
    int32_t nparts, ncols;
};
__global__ void __launch_bounds__(256) unpack_words_kernel(UnpackArgs ua, const uint64_t *src, uint64_t *dst) {
    const int64_t total = ua.offset[ua.nparts];
    const int64_t part_words = int64_t(ua.ncols) * ua.stride + 1;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total * ua.ncols; i += int64_t(gridDim.x) * blockDim.x) {
        const int c = int(i / total);
        const int64_t r = i - int64_t(c) * total;
        int p = 0;
        while (r >= ua.offset[p + 1]) ++p;
        dst[i] = src[int64_t(p) * part_words + int64_t(c) * ua.stride + (r - ua.offset[p])];
    }
}
__global__ void __launch_bounds__(256) count_valid_kernel(const uint64_t *words, int64_t n_rows, unsigned long long *out) {
    const int64_t nfull = n_rows / 64; // whole words; the tail is read byte-wise (a borrowed bitmap ends at ceil(n/8) bytes)
    unsigned long long c = 0;
    for (int64_t w = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; w < nfull; w += int64_t(gridDim.x) * blockDim.x)
        c += (unsigned long long)__popcll(words[w]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words + nfull);
        for (int64_t r = 0; r < (n_rows & 63); ++r) c += (bytes[r >> 3] >> (r & 7)) & 1;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (nqe::lane_id() == 0 && c) atomicAdd(out, c);
}
} // namespace

// number of set bits among the first n_rows bits of a (word-padded) device bitmap
static int64_t count_valid(nqe_ctx *ctx, const void *bitmap, int64_t n_rows) {
    if (n_rows == 0) return 0;
    BufRef cnt = dev_alloc_zero(ctx, 8);
    launch(ctx, "count_valid", count_valid_kernel, dim3(stream_grid(ctx, n_rows / 64 + 1, 256)), dim3(256), 0, (const uint64_t *)bitmap, n_rows,
           (unsigned long long *)cnt->ptr);
    return int64_t(read_scalar(ctx, (const unsigned long long *)cnt->ptr));
}

nqe_status nqe_table_create(nqe_ctx *ctx, const nqe_column *columns, int32_t num_columns, nqe_table **out) {
    return nqe_table_create_flags(ctx, columns, num_columns, 0u, out);
}

nqe_status nqe_table_create_flags(nqe_ctx *ctx, const nqe_column *columns, int32_t num_columns, uint32_t flags, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !out || num_columns < 0 || (num_columns > 0 && !columns)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (flags & ~uint32_t(NQE_TABLE_IMMUTABLE)) fail(NQE_ERR_INVALID_ARGUMENT, "unknown table flags");
    const bool immutable = (flags & NQE_TABLE_IMMUTABLE) != 0;
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    for (int i = 0; i < num_columns; ++i) {
        const nqe_column &c = columns[i];
        if (c.length < 0) fail(NQE_ERR_INVALID_ARGUMENT, "negative column length");
        if (i == 0) t->rows = c.length;
        else if (c.length != t->rows) fail(NQE_ERR_ARROW, "all columns in a record batch must have the same length");
        if (!(is_word_type(c.dtype) || c.dtype == NQE_BOOLEAN || c.dtype == NQE_UTF8))
            fail(NQE_ERR_NOT_SUPPORTED, "unsupported column dtype");
        DevColumn d;
        d.dtype = c.dtype;
        d.length = c.length;
        d.null_count = c.validity ? c.null_count : 0;
        size_t vb = values_bytes(c.dtype, c.length);
        if (c.location == NQE_DEVICE) {
            d.values = dev_borrow(ctx, c.values, vb);
            if (c.validity) d.validity = dev_borrow(ctx, c.validity, bitmap_bytes(c.length));
            if (c.dtype == NQE_UTF8) {
                d.data = dev_borrow(ctx, c.data, size_t(c.data_length));
                d.data_length = c.data_length;
            }
            for (const BufRef *b : {&d.values, &d.validity, &d.data})
                if (*b) (*b)->caller_immutable = immutable;
        } else {
            // pad bitmaps to whole words so word-wise kernels may read them
            size_t ab = c.dtype == NQE_BOOLEAN ? bitmap_alloc_bytes(c.length) : vb;
            d.values = c.dtype == NQE_BOOLEAN ? dev_alloc_zero(ctx, ab) : dev_alloc(ctx, ab);
            if (vb) NQE_HIP_CHECK(hipMemcpyAsync(d.values->ptr, c.values, vb, hipMemcpyHostToDevice, ctx->stream));
            if (c.validity) {
                d.validity = dev_alloc_zero(ctx, bitmap_alloc_bytes(c.length));
                if (c.length)
                    NQE_HIP_CHECK(hipMemcpyAsync(d.validity->ptr, c.validity, bitmap_bytes(c.length), hipMemcpyHostToDevice,
                                                 ctx->stream));
            }
            if (c.dtype == NQE_UTF8) {
                d.data = dev_alloc(ctx, size_t(c.data_length));
                d.data_length = c.data_length;
                if (c.data_length)
                    NQE_HIP_CHECK(hipMemcpyAsync(d.data->ptr, c.data, size_t(c.data_length), hipMemcpyHostToDevice, ctx->stream));
            }
        }
        // A bitmap that marks every row valid is dropped (null_count 0 from the caller, or counted here when unknown): such
        // columns then take the kernels specialised for non-null inputs.  Arrow semantics are unchanged (an absent bitmap
        // == all valid).
        if (d.validity) {
            if (d.null_count < 0) d.null_count = c.length - count_valid(ctx, d.validity->ptr, c.length);
            if (d.null_count == 0) d.validity = nullptr;
        }
        t->cols.push_back(std::move(d));
    }
    sync(ctx); // host buffers may be released by the caller on return
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_table_release(nqe_table *table) {
    delete table;
    return NQE_OK;
}

int64_t nqe_table_num_rows(const nqe_table *table) { return table ? table->rows : 0; }
int32_t nqe_table_num_columns(const nqe_table *table) { return table ? int32_t(table->cols.size()) : 0; }

nqe_status nqe_table_column(const nqe_table *table, int32_t i, nqe_column *out) {
    NQE_API_BEGIN(table ? table->ctx : nullptr)
    if (!table || !out || i < 0 || size_t(i) >= table->cols.size()) fail(NQE_ERR_INVALID_ARGUMENT, "column index out of range");
    const DevColumn &c = table->cols[size_t(i)];
    std::memset(out, 0, sizeof(*out));
    out->dtype = c.dtype;
    out->location = NQE_DEVICE;
    out->length = c.length;
    out->null_count = c.null_count;
    out->values = c.values ? c.values->ptr : nullptr;
    out->validity = c.valid();
    out->data = c.data ? c.data->ptr : nullptr;
    out->data_length = c.data_length;
    NQE_API_END()
}

nqe_status nqe_table_download_column(const nqe_table *table, int32_t i, void *values_out, uint8_t *validity_out,
                                     void *data_out) {
    NQE_API_BEGIN(table ? table->ctx : nullptr)
    if (!table || i < 0 || size_t(i) >= table->cols.size()) fail(NQE_ERR_INVALID_ARGUMENT, "column index out of range");
    nqe_ctx *ctx = table->ctx;
    const DevColumn &c = table->cols[size_t(i)];
    size_t vb = values_bytes(c.dtype, c.length);
    if (vb && values_out) NQE_HIP_CHECK(hipMemcpyAsync(values_out, c.values->ptr, vb, hipMemcpyDeviceToHost, ctx->stream));
    if (c.validity && validity_out && c.length)
        NQE_HIP_CHECK(hipMemcpyAsync(validity_out, c.validity->ptr, bitmap_bytes(c.length), hipMemcpyDeviceToHost, ctx->stream));
    if (c.dtype == NQE_UTF8 && data_out && c.data_length)
        NQE_HIP_CHECK(hipMemcpyAsync(data_out, c.data->ptr, size_t(c.data_length), hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    NQE_API_END()
}

nqe_status nqe_table_project(nqe_ctx *ctx, const nqe_table *in, const int32_t *indices, int32_t n, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!in || !out || n < 0) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = in->rows;
    for (int k = 0; k < n; ++k) {
        if (indices[k] < 0 || size_t(indices[k]) >= in->cols.size()) fail(NQE_ERR_ARROW, "project index out of bounds");
        t->cols.push_back(in->cols[size_t(indices[k])]); // shares the buffers (Arc clone)
    }
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_table_slice(nqe_ctx *ctx, const nqe_table *in, int64_t offset, int64_t length, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!in || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (offset < 0 || length < 0 || offset + length > in->rows) fail(NQE_ERR_ARROW, "slice out of bounds");
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = length;
    for (auto &c : in->cols) t->cols.push_back(slice_column(ctx, c, offset, length));
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_table_concat(nqe_ctx *ctx, const nqe_table *const *tables, int32_t n, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!tables || !out || n <= 0) fail(NQE_ERR_INVALID_ARGUMENT, "concat needs at least one table");
    size_t nc = tables[0]->cols.size();
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    for (int k = 0; k < n; ++k) {
        if (tables[k]->cols.size() != nc) fail(NQE_ERR_ARROW, "concat: schemas differ");
        t->rows += tables[k]->rows;
    }
    if (n == 1) {
        // one part: the library's own buffers are shared (tables are immutable); borrowed ones are copied like any output
        for (auto &c : tables[0]->cols) {
            if (c.shareable()) t->cols.push_back(c);
            else {
                t->cols.push_back(slice_column(ctx, c, 0, c.length));
                t->cols.back().null_count = c.null_count;
            }
        }
    } else {
        for (size_t c = 0; c < nc; ++c) {
            std::vector<const DevColumn *> parts;
            for (int k = 0; k < n; ++k) parts.push_back(&tables[k]->cols[c]);
            t->cols.push_back(concat_columns(ctx, parts));
        }
    }
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_table_pack_words(nqe_ctx *ctx, const nqe_table *const *tables, int32_t num_tables, int64_t stride_rows, void *dst_device) {
    NQE_API_BEGIN(ctx)
    if (!tables || num_tables <= 0 || stride_rows < 0 || !dst_device) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    PackArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    const int64_t rows = tables[0]->rows;
    for (int k = 0; k < num_tables; ++k) {
        if (tables[k]->rows != rows) fail(NQE_ERR_INVALID_ARGUMENT, "pack: tables differ in rows");
        for (auto &c : tables[k]->cols) {
            if (!is_word_type(c.dtype) || c.validity) fail(NQE_ERR_NOT_SUPPORTED, "pack: only 8-byte columns without validity");
            if (pa.ncols == PACK_MAX_COLS) fail(NQE_ERR_NOT_SUPPORTED, "pack: too many columns");
            pa.src[pa.ncols++] = (const uint64_t *)c.words();
        }
    }
    if (rows > stride_rows) fail(NQE_ERR_INVALID_ARGUMENT, "pack: rows exceed the stride");
    pa.rows = rows;
    pa.stride = stride_rows;
    launch(ctx, "pack_words", pack_words_kernel, dim3(unsigned(std::max<int64_t>(1, std::min<int64_t>(1024, (rows * pa.ncols + 255) / 256)))), dim3(256), 0, pa,
           (uint64_t *)dst_device);
    NQE_API_END()
}

nqe_status nqe_table_unpack_words(nqe_ctx *ctx, const void *src_device, int32_t num_parts, int32_t num_columns, int64_t stride_rows,
                                  const int64_t *counts, const int32_t *dtypes, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!out || num_parts <= 0 || num_parts > UNPACK_MAX_PARTS || num_columns <= 0 || stride_rows < 0 || !counts || !dtypes || (stride_rows && !src_device))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    UnpackArgs ua;
    std::memset(&ua, 0, sizeof(ua));
    int64_t total = 0;
    for (int p = 0; p < num_parts; ++p) {
        if (counts[p] < 0 || counts[p] > stride_rows) fail(NQE_ERR_INVALID_ARGUMENT, "unpack: count exceeds the stride");
        ua.offset[p] = total;
        total += counts[p];
    }
    ua.offset[num_parts] = total;
    ua.nparts = num_parts;
    ua.ncols = num_columns;
    ua.stride = stride_rows;
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = total;
    BufRef all = dev_alloc(ctx, size_t(total) * size_t(num_columns) * 8 + 8); // one allocation, columns are views
    for (int c = 0; c < num_columns; ++c) {
        if (!is_word_type(dtypes[c])) fail(NQE_ERR_NOT_SUPPORTED, "unpack: only 8-byte columns");
        DevColumn d;
        d.dtype = dtypes[c];
        d.length = total;
        d.null_count = 0;
        d.values = dev_view(all, size_t(c) * size_t(total) * 8, size_t(total) * 8);
        t->cols.push_back(std::move(d));
    }
    if (total)
        launch(ctx, "unpack_words", unpack_words_kernel, dim3(unsigned(std::min<int64_t>(1024, (total * num_columns + 255) / 256))), dim3(256), 0, ua,
               (const uint64_t *)src_device, (uint64_t *)all->ptr);
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_take(nqe_ctx *ctx, const nqe_table *in, const nqe_table *idx_table, int32_t idx_column, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!in || !idx_table || !out || idx_column < 0 || size_t(idx_column) >= idx_table->cols.size())
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    const DevColumn &ix = idx_table->cols[size_t(idx_column)];
    if (ix.dtype != NQE_INT64) fail(NQE_ERR_INVALID_ARGUMENT, "take indices must be Int64");
    flags_reset(ctx);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = ix.length;
    for (auto &c : in->cols) t->cols.push_back(take_column(ctx, c, (const int64_t *)ix.words(), ix.length));
    throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_synth_fill(nqe_ctx *ctx, int32_t kind, uint64_t seed, int64_t first_row, int64_t n, uint64_t modulus,
                          int64_t base, void *out_device) {
    NQE_API_BEGIN(ctx)
    if (n < 0 || (n > 0 && !out_device)) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (kind < NQE_SYNTH_ROWID || kind > NQE_SYNTH_F64_0_100) fail(NQE_ERR_INVALID_ARGUMENT, "unknown synth kind");
    if (kind == NQE_SYNTH_UNIFORM && modulus == 0) fail(NQE_ERR_INVALID_ARGUMENT, "modulus must be > 0");
    if (n)
        launch(ctx, "synth_fill", synth_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kind, seed, first_row, n,
               modulus, base, (uint64_t *)out_device);
    NQE_API_END()
}

nqe_status nqe_device_alloc(nqe_ctx *ctx, size_t bytes, void **out) {
    NQE_API_BEGIN(ctx)
    if (!out) fail(NQE_ERR_INVALID_ARGUMENT, "out is NULL");
    NQE_HIP_CHECK(hipSetDevice(ctx->device));
    NQE_HIP_CHECK(hipMalloc(out, bytes ? bytes : 8));
    NQE_API_END()
}
nqe_status nqe_device_free(nqe_ctx *ctx, void *ptr) {
    NQE_API_BEGIN(ctx)
    sync(ctx);
    NQE_HIP_CHECK(hipFree(ptr));
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::synth_kernel);
