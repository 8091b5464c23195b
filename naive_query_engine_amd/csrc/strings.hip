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
#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {

namespace {

constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
constexpr int64_t EMPTY = -1;

__device__ __forceinline__ uint64_t fnv1a64(const uint8_t *p, int32_t len) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (int32_t i = 0; i < len; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

__device__ __forceinline__ bool bytes_equal(const uint8_t *a, const uint8_t *b, int32_t len) {
    for (int32_t i = 0; i < len; ++i)
        if (a[i] != b[i]) return false;
    return true;
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

} // namespace nqe
