// strings.hip — Utf8 KEYS for group-by and join (reference: aggregate/mod.rs:170-216 groups by String,
// hash_join.rs:146-160 / :203-224 joins on Utf8 with XxHash64 over the bytes + an equality re-check).
//
// Device design: a content-addressed table maps every distinct string to a REPRESENTATIVE ROW
// (open addressing on a 64-bit FNV-1a of the bytes; a slot holds a row number; on a hash-slot hit the
// bytes are compared, so the encoding is exact, never probabilistic).  Equal strings ⇔ equal codes, so
// the Int64 group-by / join machinery runs unchanged on the code column; the actual strings only come
// back through `take` (join payload, optional keys_out).  Like the reference, key validity is ignored by
// the join (bytes of a NULL slot = its offsets' span) and NULL keys are dropped by the group-by (the code
// column carries the string column's validity).
#include <cstring>
#include <string>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {

namespace {

constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
constexpr int64_t EMPTY = -1;

// Hash and equality of short strings, eight bytes per step (unaligned 8-byte loads: supported for global memory on gfx9 and later;
// the tail of fewer than eight bytes is read byte by byte, so nothing beyond the string is touched).  The hash is unobservable in
// results (as the reference's XxHash64 is, hash_join.rs:68-70): only its spread matters.  Byte-at-a-time FNV-1a made the dictionary
// encode of 10^7 sixteen-byte keys 1.3 ms.
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p) {
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    return *reinterpret_cast<const u64_unaligned *>(p);
}
__device__ __forceinline__ uint64_t tail_word(const uint8_t *p, int32_t n) { // n < 8 bytes, little-endian
    uint64_t w = 0;
    for (int32_t i = 0; i < n; ++i) w |= uint64_t(p[i]) << (8 * i);
    return w;
}
__device__ __forceinline__ uint64_t fnv1a64(const uint8_t *p, int32_t len) {
    uint64_t h = 0xcbf29ce484222325ull ^ uint64_t(uint32_t(len));
    int32_t i = 0;
    for (; i + 8 <= len; i += 8) {
        h ^= load_u64_unaligned(p + i);
        h *= 0x9FB21C651E98DF25ull;
        h ^= h >> 29;
    }
    if (i < len) {
        h ^= tail_word(p + i, len - i);
        h *= 0x9FB21C651E98DF25ull;
        h ^= h >> 29;
    }
    return h * 0x100000001b3ull;
}

__device__ __forceinline__ bool bytes_equal(const uint8_t *a, const uint8_t *b, int32_t len) {
    int32_t i = 0;
    for (; i + 8 <= len; i += 8)
        if (load_u64_unaligned(a + i) != load_u64_unaligned(b + i)) return false;
    return i == len || tail_word(a + i, len - i) == tail_word(b + i, len - i);
}

// INSERT: codes[i] = representative row of string i inside `col` itself.
// LOOKUP (build_* != null): codes[i] = representative BUILD row of the equal build string, or -(i+2) (matches nothing).
template <bool INSERT>
__global__ void utf8_encode_kernel(const int32_t *offs, const uint8_t *data, int64_t n, const int32_t *build_offs, const uint8_t *build_data,
                                   long long *slots, uint32_t cap, int shift, int64_t *codes) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t o = offs[i], len = offs[i + 1] - o;
        const uint8_t *s = data + o;
        uint32_t slot = uint32_t((fnv1a64(s, len) * GOLD) >> shift);
        int64_t code = INSERT ? i : -(i + 2);
        for (uint32_t probe = 0; probe < cap; ++probe) {
            long long cur = __hip_atomic_load(&slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == EMPTY) {
                if (!INSERT) break; // absent
                long long old = atomicCAS((unsigned long long *)&slots[slot], (unsigned long long)EMPTY, (unsigned long long)i);
                if (old == EMPTY) break; // this row is the representative
                cur = old;
            }
            const int32_t co = build_offs[cur], clen = build_offs[cur + 1] - co;
            if (clen == len && bytes_equal(build_data + co, s, len)) {
                code = cur;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
        codes[i] = code;
    }
}

__global__ void fill_i64_kernel(long long *p, long long v, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

} // namespace

DevColumn utf8_encode_build(nqe_ctx *ctx, const DevColumn &col, Utf8Dict *dict) {
    const int64_t n = col.length;
    uint32_t cap = 64;
    while (uint64_t(cap) < 2ull * uint64_t(n)) cap <<= 1;
    int lg = 0;
    while ((1u << lg) < cap) ++lg;
    dict->cap = cap;
    dict->shift = 64 - lg;
    dict->slots = dev_alloc(ctx, size_t(cap) * 8);
    dict->build = col;
    launch(ctx, "utf8_dict_init", fill_i64_kernel, dim3(stream_grid(ctx, cap, 256)), dim3(256), 0, (long long *)dict->slots->ptr, (long long)EMPTY,
           int64_t(cap));
    DevColumn codes = make_word_column(ctx, NQE_INT64, n, false);
    codes.validity = col.validity; // NULL strings stay NULL keys
    codes.null_count = col.null_count;
    if (n)
        launch(ctx, "utf8_encode_insert", utf8_encode_kernel<true>, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, (const int32_t *)col.values->ptr,
               col.data ? (const uint8_t *)col.data->ptr : nullptr, n, (const int32_t *)col.values->ptr,
               col.data ? (const uint8_t *)col.data->ptr : nullptr, (long long *)dict->slots->ptr, cap, dict->shift, (int64_t *)codes.values->ptr);
    return codes;
}

DevColumn utf8_encode_probe(nqe_ctx *ctx, const DevColumn &col, const Utf8Dict &dict) {
    const int64_t n = col.length;
    DevColumn codes = make_word_column(ctx, NQE_INT64, n, false);
    codes.validity = col.validity;
    codes.null_count = col.null_count;
    if (n)
        launch(ctx, "utf8_encode_lookup", utf8_encode_kernel<false>, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, (const int32_t *)col.values->ptr,
               col.data ? (const uint8_t *)col.data->ptr : nullptr, n, (const int32_t *)dict.build.values->ptr,
               dict.build.data ? (const uint8_t *)dict.build.data->ptr : nullptr, (long long *)dict.slots->ptr, dict.cap, dict.shift,
               (int64_t *)codes.values->ptr);
    return codes;
}


// ---------------------------------------------------------------- Utf8 comparisons (binary.rs:127-132 → arrow *_dyn)
// eq/neq/lt/lt_eq/gt/gt_eq on StringArrays compare the UTF-8 bytes lexicographically (str ordering).  Either side may be
// a literal (ScalarValue::Utf8; the reference materialises it as an n-row StringArray, here it stays one device string).
namespace {
struct Utf8Side {
    const int32_t *offs; // null: scalar
    const uint8_t *data; // column bytes, or the scalar's bytes
    const uint8_t *valid;
    int32_t scalar_len;
    int32_t scalar_null;
};
__global__ void __launch_bounds__(256) utf8_compare_kernel(Utf8Side a, Utf8Side b, int op, int64_t n, uint64_t *out_bits, uint64_t *out_valid) {
    const int64_t n_pad = (n + 63) / 64 * 64;
    for (int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; row < n_pad; row += int64_t(gridDim.x) * blockDim.x) {
        bool ok = false, r = false;
        if (row < n) {
            const bool av = a.offs ? (a.valid ? get_bit(a.valid, row) : true) : !a.scalar_null;
            const bool bv = b.offs ? (b.valid ? get_bit(b.valid, row) : true) : !b.scalar_null;
            ok = av && bv;
            if (ok) {
                const uint8_t *pa = a.offs ? a.data + a.offs[row] : a.data;
                const uint8_t *pb = b.offs ? b.data + b.offs[row] : b.data;
                const int32_t la = a.offs ? a.offs[row + 1] - a.offs[row] : a.scalar_len;
                const int32_t lb = b.offs ? b.offs[row + 1] - b.offs[row] : b.scalar_len;
                const int32_t m = la < lb ? la : lb;
                int c = 0;
                for (int32_t i = 0; i < m; ++i) {
                    if (pa[i] != pb[i]) {
                        c = pa[i] < pb[i] ? -1 : 1;
                        break;
                    }
                }
                if (c == 0) c = la < lb ? -1 : la > lb ? 1 : 0;
                r = op == NQE_OP_EQ ? c == 0 : op == NQE_OP_NOT_EQ ? c != 0 : op == NQE_OP_LT ? c < 0 : op == NQE_OP_LT_EQ ? c <= 0 : op == NQE_OP_GT ? c > 0 : c >= 0;
            }
        }
        const uint64_t wb = __ballot(ok && r), wv = __ballot(ok);
        if (lane_id() == 0) {
            out_bits[row >> 6] = wb;
            if (out_valid) out_valid[row >> 6] = wv;
        }
    }
}
__global__ void __launch_bounds__(256) utf8_repeat_kernel(const uint8_t *lit, int32_t len, int64_t n, int32_t *offs, uint8_t *data) {
    const int64_t total = n * len;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total || i <= n; i += int64_t(gridDim.x) * blockDim.x) {
        if (i < total) data[i] = lit[i % len];
        if (i <= n) offs[i] = int32_t(i * len);
    }
}
BufRef upload_bytes(nqe_ctx *ctx, const std::string &sv) {
    BufRef b = dev_alloc(ctx, sv.size() + 8);
    if (!sv.empty()) NQE_HIP_CHECK(hipMemcpyAsync(b->ptr, sv.data(), sv.size(), hipMemcpyHostToDevice, ctx->stream));
    sync(ctx); // `sv` may be a temporary of the caller
    return b;
}
} // namespace

DevColumn utf8_compare(nqe_ctx *ctx, int op, const DevColumn *lcol, const std::string &llit, bool llit_null, const DevColumn *rcol,
                       const std::string &rlit, bool rlit_null, int64_t n) {
    BufRef lbuf, rbuf;
    auto side = [&](const DevColumn *c, const std::string &lit, bool lit_null, BufRef &buf) {
        Utf8Side sd;
        std::memset(&sd, 0, sizeof(sd));
        if (c) {
            sd.offs = static_cast<const int32_t *>(c->values->ptr);
            sd.data = c->data ? static_cast<const uint8_t *>(c->data->ptr) : nullptr;
            sd.valid = c->valid();
        } else {
            buf = upload_bytes(ctx, lit);
            sd.data = static_cast<const uint8_t *>(buf->ptr);
            sd.scalar_len = int32_t(lit.size());
            sd.scalar_null = lit_null ? 1 : 0;
        }
        return sd;
    };
    const Utf8Side a = side(lcol, llit, llit_null, lbuf), b = side(rcol, rlit, rlit_null, rbuf);
    const bool need_valid = (lcol ? lcol->validity != nullptr : llit_null) || (rcol ? rcol->validity != nullptr : rlit_null);
    DevColumn out = make_bool_column(ctx, n, need_valid);
    if (n)
        launch(ctx, "utf8_compare", utf8_compare_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, a, b, op, n, (uint64_t *)out.values->ptr,
               need_valid ? (uint64_t *)out.validity->ptr : nullptr);
    sync(ctx); // the literal staging buffers are released here
    return out;
}

// ScalarValue::Utf8(Some(s)).into_array(n): StringArray of n copies (None: all-null)
DevColumn utf8_literal_column(nqe_ctx *ctx, const std::string &lit, bool lit_null, int64_t n) {
    DevColumn c;
    c.dtype = NQE_UTF8;
    c.length = n;
    const int64_t len = lit_null ? 0 : int64_t(lit.size());
    if (n * len > int64_t(INT32_MAX)) fail(NQE_ERR_ARROW, "Utf8 literal column exceeds 2 GiB (i32 offsets)");
    c.values = dev_alloc_zero(ctx, size_t(n + 1) * 4 + 8);
    c.data = dev_alloc(ctx, size_t(n * len) + 8);
    c.data_length = n * len;
    if (len && n) {
        BufRef b = upload_bytes(ctx, lit);
        launch(ctx, "utf8_repeat", utf8_repeat_kernel, dim3(stream_grid(ctx, n * len, 256)), dim3(256), 0, (const uint8_t *)b->ptr, int32_t(len), n,
               (int32_t *)c.values->ptr, (uint8_t *)c.data->ptr);
        sync(ctx);
    }
    if (lit_null) {
        c.validity = dev_alloc_zero(ctx, bitmap_alloc_bytes(n));
        c.null_count = n;
    }
    return c;
}

} // namespace nqe

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::fill_i64_kernel);
