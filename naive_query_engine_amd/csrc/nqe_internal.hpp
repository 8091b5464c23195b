// nqe_internal.hpp — host-side internals of libnqe_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/nqe.h"

namespace nqe {

struct Error {
    int code;
    std::string msg;
};

[[noreturn]] inline void fail(int code, const std::string &msg) { throw Error{code, msg}; }

#define NQE_HIP_CHECK(expr)                                                                                  \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            ::nqe::fail(_e == hipErrorOutOfMemory ? NQE_ERR_OUT_OF_MEMORY : NQE_ERR_HIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                  \
    } while (0)

inline size_t bitmap_bytes(int64_t n) { return size_t((n + 7) / 8); }
// bitmaps we allocate are padded to whole 64-bit words so kernels may store u64 words
inline size_t bitmap_alloc_bytes(int64_t n) { return size_t((n + 63) / 64) * 8; }
inline bool is_word_type(int dt) { return dt == NQE_INT64 || dt == NQE_UINT64 || dt == NQE_FLOAT64; }

} // namespace nqe

// Device error flag slots (ctx->d_flags[i])
enum { NQE_FLAG_DIV_ZERO = 0, NQE_FLAG_OVERFLOW = 1, NQE_FLAG_TABLE_FULL = 2, NQE_FLAG_OOB = 3, NQE_FLAG_NEED_PARTITION = 4, NQE_FLAG_NEED_LEVEL2 = 5, NQE_FLAG_DENSE_OVERFLOW = 6,
       NQE_FLAG_GROUP_COUNT = 7 /* not an error: the aggregate's group count rides along with the flag read-back */,
       NQE_FLAG_SLAB_OVERFLOW = 8 /* a partition slab of the count-free scatter is full: the host takes the exact form */,
       NQE_FLAG_KEY32_OVERFLOW = 9 /* a group key outside the int32 range met the 12-byte-tuple slab form: the host takes 16-byte tuples */, NQE_NUM_FLAGS = 10 };
// words of the pinned flag mirror past the flags: results a tail kernel sends along with the flag read-back (aggregate.hip: the dense table's
// group count and key range)
constexpr int NQE_FLAG_MIRROR_EXTRA = 8;

// Settings of the aggregate operator (aggregate.hip).  Rounds 2-5 read every one of them from the environment for A/B runs; what those
// runs decided is a constant now (the measurements: profiles/r02 … r05, DESIGN.md §9 "retired switches"), and the forms that lost on
// every measured shape are gone.  Still read when a context is created: the test hook and the tier trace.  (The switches the tests flip
// between calls — NQE_NO_PLAN_HINTS, NQE_NO_KEY_SAMPLE, NQE_NO_RANGE_PARTITION, NQE_NO_RANGE_TAIL, NQE_NO_AGG_JIT, NQE_NO_WIDE_DIRECT,
// NQE_TEST_SLAB_OOM — are read per call where they are used.)
struct AggSwitches {
    static constexpr bool no_three_column_pass = false; // three value columns without min / max go through ONE pass of the three-column instance
    static constexpr bool no_key_range = false;         // a plain key column whose measured range fits a workgroup table is addressed by key - min
    static constexpr int subsets_max = 1;               // log2 of the key subsets of the streaming tier (two subsets; four measured slower than partitioning)
    static constexpr int slab_parts_first = 8;          // log2 of the first partition count of the hashed slab form
    static constexpr int flag_check_mask = 7;           // a wave of the one-tile streaming loop looks at the overflow flags every 8th iteration
    static constexpr bool no_agg_jit_chains = false;    // chain predicates / chain keys take the run-time specialised kernel
    static constexpr int agg_jit_all = 1;               // `col % m` by magic multiply goes through the specialised kernel (power-of-two moduli stay static)
    static constexpr bool tiny_groups = true;           // `col % m`, m <= 4: group state in registers (aggregate_tiny.hip)
    static constexpr bool direct_partials = true;       // direct-mapped workgroup tables leave whole and are folded by a kernel (no device-scope atomics)
    static constexpr bool range_tier = true;            // key-range partitions: partition count from the range, Q workgroups per partition, transposing tail
    static constexpr int range_slots_log2 = 12;         // log2 of the slots per table the range tier sizes its partition count for (9: 1.11, 10: 1.03, 11: 0.99, 12: 0.97 ms per step at 65536 groups)
    static constexpr int soa_threads = 512;             // workgroup size of the two-stream scatter: two per CU (1024 x 1 and 256 x 4 measured slower: profiles/r06/ab_soa_threads.txt)
    static constexpr bool direct_subsets = true;        // two key subsets over a measured key range address their tables directly
    static constexpr bool lds_load_limit = true;        // a hashed workgroup table hands over at three quarters of its slots
    static constexpr int range_emit_items = 0;          // keys per thread of the range tier's tail: by the range (1 below 2^19 keys, else 4)
    int tiny_unpack_tiles = 4096;      // NQE_TINY_UNPACK_TILES: tiles between two unpackings of the register kernel's packed row counters (1..4096; a TEST hook: at 4096 the branch first runs beyond ~2 x 10^9 rows)
    bool debug = false;                // NQE_DEBUG=1: the tier decisions on stderr
    void read_environment();
};

struct nqe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::string last_error;

    // caching device allocator: freed blocks are reused by later (stream-ordered) work
    std::multimap<size_t, void *> pool;
    size_t pool_bytes = 0;
    // nqe_ctx_reserve: one block taken from the driver up front and sub-allocated (best fit, free neighbours coalesced): the scratch
    // and output buffers of a FIRST query then cost no hipMalloc (a first hipMalloc of a gigabyte block is a millisecond and more —
    // the reference's run_sql is one-shot, so the first execution is the one that counts).  Reuse is stream-ordered like the pool's.
    char *arena_base = nullptr;
    size_t arena_size = 0, arena_free_bytes = 0;
    std::map<size_t, size_t> arena_free; // offset -> bytes, address-ordered
    size_t live_bytes = 0;

    // per-kernel timing (bench.py roofline leg)
    bool timing = false;
    struct TimingRec {
        std::string name;
        hipEvent_t start, stop;
    };
    std::vector<TimingRec> timings;

    // Grouped aggregates that had to be redone hash-partitioned (more distinct keys than a workgroup's LDS table), remembered by
    // (key column buffer, rows, key expression): the next execution of the same query over the same table starts partitioned
    // instead of paying for an abandoned single-pass attempt and its read-back first.  Only a starting point — a partitioned run
    // is correct for any number of groups, and a single-pass run still falls back when its tables overflow.
    AggSwitches agg_sw;
    std::map<uint64_t, uint8_t> agg_hints;
    // … and, by the same key: the value range {min, span} of a plain integer key column (`group by k`: dictionary codes, small ids),
    // measured by the first execution of the query shape.  A range that fits a workgroup table makes the streaming kernel address
    // the table by key - min.  The kernel checks every key against the range (the column's contents may have changed under the
    // entry): a key outside it asks for the other paths, and the entry is dropped.
    std::map<uint64_t, std::pair<int64_t, uint64_t>> agg_key_ranges;
    // PK-FK joins whose optimistic one-pass probe failed (some foreign key without its primary key), by (build key column, build
    // rows, probe key column, probe rows): HashJoin::execute builds a fresh join table per call (nqe_hash_join_execute), so the
    // verdict has to outlive the table for the next execution of the same join to go straight to the two-pass form
    std::map<uint64_t, uint8_t> join_hints;
    // run-time specialised expression kernels (expr_jit.hpp: JitCache — worker threads, code objects), created on first use
    std::shared_ptr<void> jit;

    size_t lds_per_block = 160 * 1024; // LDS a workgroup may ask for on this device (hipDeviceProp_t::sharedMemPerBlock)
    std::string arch = "gfx950";      // the device's ISA name (hipRTC target of the run-time specialised kernels)
    int *d_flags = nullptr; // NQE_NUM_FLAGS ints on the device
    int *h_flags = nullptr; // pinned host mirror
    int *h_flags_dev = nullptr; // the mirror's address as seen by kernels (a tail kernel may write it directly)
    // d_flags known to be all zero at the current stream position: set by a read-back that saw only zeros (or a reset), cleared
    // by every kernel launch — lets the next operator skip its reset (one fill launch per operator)
    bool flags_clean = false;
};

namespace nqe {

// A device allocation (owned → returned to the ctx pool) or a borrowed device pointer.
struct DevBuf {
    nqe_ctx *ctx = nullptr;
    void *ptr = nullptr;
    size_t bytes = 0;    // usable size requested
    size_t capacity = 0; // pooled block size
    bool owned = false;
    bool in_arena = false; // a range of the context's reserved block (returned to its free list)
    // the memory belongs to the context's pool (an allocation of ours, or a view into one): its reuse is ordered on the context's
    // stream, so work enqueued on that stream may still be reading it when the last reference goes away.  false: the caller's.
    bool lib_memory = false;
    // borrowed memory the caller has promised to keep alive and unmodified while any table derived from it lives
    // (NQE_TABLE_IMMUTABLE at nqe_table_create_flags): an output may then alias it, as it may alias the library's own buffers
    bool caller_immutable = false;
    ~DevBuf();
};
using BufRef = std::shared_ptr<DevBuf>;
// may an OUTPUT table reference this buffer instead of a copy?  The library's own memory is reference-counted, so an alias keeps
// it alive; borrowed memory is the caller's to free or overwrite the moment the producing call returns — unless promised otherwise.
inline bool buf_shareable(const BufRef &b) { return !b || b->lib_memory || b->caller_immutable; }

BufRef dev_alloc(nqe_ctx *ctx, size_t bytes);
BufRef dev_alloc_zero(nqe_ctx *ctx, size_t bytes);
BufRef dev_borrow(nqe_ctx *ctx, const void *ptr, size_t bytes);
// a sub-range of `parent` (keeps the parent alive)
BufRef dev_view(const BufRef &parent, size_t offset, size_t bytes);
void pool_trim(nqe_ctx *ctx);

// One Arrow array resident in HBM.
struct DevColumn {
    int dtype = NQE_NULLTYPE;
    int64_t length = 0;
    int64_t null_count = 0; // -1 unknown; 0 ⇒ validity may still be present (all ones)
    BufRef values;          // 8 B/row words, packed bits (Boolean), or int32 offsets (Utf8)
    BufRef validity;        // packed bits or null
    BufRef data;            // Utf8 bytes
    int64_t data_length = 0;

    const uint64_t *words() const { return values ? static_cast<const uint64_t *>(values->ptr) : nullptr; }
    const uint8_t *bits() const { return values ? static_cast<const uint8_t *>(values->ptr) : nullptr; }
    const uint8_t *valid() const { return validity ? static_cast<const uint8_t *>(validity->ptr) : nullptr; }
    bool shareable() const { return buf_shareable(values) && buf_shareable(validity) && buf_shareable(data); }
};

} // namespace nqe

inline uint64_t nqe_next_table_uid() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1, std::memory_order_relaxed);
}
struct nqe_table {
    nqe_ctx *ctx = nullptr;
    std::vector<nqe::DevColumn> cols;
    int64_t rows = 0;
    // identity of this handle: what a context remembers about a query (aggregate plan hints, key ranges) is keyed by it as well as by
    // the buffers — a NEW table over the same device memory (a caller refilling its buffers, the pool handing a block out again)
    // holds other data and starts from nothing instead of from the previous table's plan
    uint64_t uid = nqe_next_table_uid();
};

namespace nqe {

// ---- code objects ------------------------------------------------------------------------
// HIP loads a translation unit's code object (1-5 MB each here) on the FIRST launch of one of its kernels: 2-5 ms that the first
// query of a process would pay per operator family (the reference's run_sql is one-shot, db.rs:24-37).  Every translation unit
// registers one of its kernels at static-initialisation time; nqe_ctx_create asks the runtime for that kernel's attributes, which
// loads the unit for the context's device (NQE_LAZY_MODULES=1 leaves the loading to the first launch).
void register_module_probe(const void *kernel);
#define NQE_MODULE_PROBE(kernel) static const bool nqe_module_probe_ = (nqe::register_module_probe(reinterpret_cast<const void *>(&kernel)), true)

// ---- launch helpers ---------------------------------------------------------------------
struct TimerScope {
    nqe_ctx *ctx;
    size_t idx = size_t(-1);
    TimerScope(nqe_ctx *c, const char *name) : ctx(c) {
        if (ctx->timing) {
            nqe_ctx::TimingRec r;
            r.name = name;
            NQE_HIP_CHECK(hipEventCreate(&r.start));
            NQE_HIP_CHECK(hipEventCreate(&r.stop));
            NQE_HIP_CHECK(hipEventRecord(r.start, ctx->stream));
            ctx->timings.push_back(r);
            idx = ctx->timings.size() - 1;
        }
    }
    ~TimerScope() {
        if (idx != size_t(-1)) (void)hipEventRecord(ctx->timings[idx].stop, ctx->stream);
    }
};

template <typename K, typename... Args>
inline void launch(nqe_ctx *ctx, const char *name, K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    TimerScope t(ctx, name);
    ctx->flags_clean = false;
    hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, args...);
    NQE_HIP_CHECK(hipGetLastError());
}

// grid for a streaming kernel: enough blocks to fill the chip, grid-stride the rest
inline int stream_grid(nqe_ctx *ctx, int64_t work_items, int per_block, int blocks_per_cu = 8) {
    int64_t need = (work_items + per_block - 1) / per_block;
    int64_t cap = int64_t(ctx->num_cus) * blocks_per_cu;
    if (need < 1) need = 1;
    return int(need < cap ? need : cap);
}

void flags_reset(nqe_ctx *ctx);
// expr.hip: waits for the run-time specialisations being compiled for this context
void jit_wait(nqe_ctx *ctx);
// synchronises the stream, returns flag values
void flags_read(nqe_ctx *ctx, int out[NQE_NUM_FLAGS]);
// the same when the last kernel of the stream copied the flags into the pinned mirror itself (no device-to-host copy)
void flags_read_mirrored(nqe_ctx *ctx, int out[NQE_NUM_FLAGS]);
void throw_on_flags(nqe_ctx *ctx);

// Host wait for the stream.  The operators wait for a handful of bytes (flags, a row count) right behind kernels that run for
// microseconds to a few milliseconds, and an interrupt-driven hipStreamSynchronize adds its wake-up latency to every such step:
// poll the stream for the first milliseconds, then block.
inline void sync(nqe_ctx *ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) NQE_HIP_CHECK(e);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
    NQE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

template <typename T> inline T read_scalar(nqe_ctx *ctx, const T *dptr) {
    T v;
    NQE_HIP_CHECK(hipMemcpyAsync(&v, dptr, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    return v;
}

// ---- column utilities (columns.hip) -----------------------------------------------------
DevColumn make_word_column(nqe_ctx *ctx, int dtype, int64_t n, bool with_validity);
DevColumn make_bool_column(nqe_ctx *ctx, int64_t n, bool with_validity);
// out[j] = in[idx[j]] for every column (arrow take); idx are int64 row numbers on the device
DevColumn take_column(nqe_ctx *ctx, const DevColumn &src, const int64_t *idx, int64_t m);
// Utf8 gather; idx < 0 emits NULL when allow_null_idx (else out of bounds)
DevColumn take_utf8(nqe_ctx *ctx, const DevColumn &src, const int64_t *idx, int64_t m, bool allow_null_idx);
BufRef iota_i64(nqe_ctx *ctx, int64_t first, int64_t n);
DevColumn slice_column(nqe_ctx *ctx, const DevColumn &src, int64_t off, int64_t len);
DevColumn concat_columns(nqe_ctx *ctx, const std::vector<const DevColumn *> &parts);
// dst bits [dst_off, dst_off + n) |= src bits [0, n) (src == nullptr: ones); dst zero-initialised and padded to whole words + 8 bytes
void bitmap_place(nqe_ctx *ctx, const uint8_t *src, uint64_t *dst, int64_t dst_off, int64_t n);
// dst[i] = src[i] - src[0] + delta for i in [0, n]: the offsets of a Utf8 part moved behind `delta` bytes of earlier parts
void utf8_rebase_offsets(nqe_ctx *ctx, const int32_t *src, int64_t n, int32_t delta, int32_t *dst);
// exclusive prefix sum of n uint32 counts into uint64 offsets (offsets[n] = total); device arrays
void exclusive_scan_u32_to_u64(nqe_ctx *ctx, const uint32_t *counts, uint64_t *offsets, int64_t n);
void exclusive_scan_u32_inplace(nqe_ctx *ctx, uint32_t *data, int64_t n);
// packs one byte per row (0/1) into an LSB-first bitmap
void pack_bytes_to_bits(nqe_ctx *ctx, const uint8_t *bytes, int64_t n, uint64_t *bitmap_words);
// stable LSD radix sort of (key, value) pairs by 64-bit key; keys_out/vals_out are device arrays of n
void radix_sort_pairs_u64(nqe_ctx *ctx, const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                          uint32_t *vals_out, int64_t n, bool signed_order);

// ---- Utf8 keys (strings.hip): exact string → representative-row encoding
struct Utf8Dict {
    BufRef slots;      // int64 representative build row per slot, -1 = empty
    uint32_t cap = 0;
    int shift = 0;
    DevColumn build;   // the encoded (build) column: its strings back the table
};
// codes[i] = representative row of string i (Int64 column sharing the strings' validity)
DevColumn utf8_encode_build(nqe_ctx *ctx, const DevColumn &col, Utf8Dict *dict);
// codes[i] = representative BUILD row of the equal build string, or a negative value that matches nothing
DevColumn utf8_encode_probe(nqe_ctx *ctx, const DevColumn &col, const Utf8Dict &dict);

// ---- expressions (expr.hip) -------------------------------------------------------------
struct OpAux {          // host-precomputed helpers for `x / lit`, `x % lit`
    int32_t pow2_shift; // >= 0: |lit| is 2^shift
    int32_t more;       // >= 0: |lit| is not a power of two: q = (((n - mulhi(magic,n)) >> 1) + mulhi(magic,n)) >> more
    uint64_t abs_lit;
    uint64_t magic;
};

// col [op lit]{0,4}: the expression shapes fused into the consumer kernels (literal on either side of every step)
constexpr int SIMPLE_MAX_OPS = 4;
struct SimpleExpr {
    int32_t col;       // index into the INPUT table
    int32_t src_dtype; // dtype of the column
    int32_t out_dtype;
    int32_t nops;
    int32_t op[SIMPLE_MAX_OPS];
    int32_t lit_left[SIMPLE_MAX_OPS]; // 1: lit op v
    int32_t op_dtype[SIMPLE_MAX_OPS]; // operand dtype of step k
    uint64_t lit[SIMPLE_MAX_OPS];
    OpAux aux[SIMPLE_MAX_OPS];
};

// `x op lit` over an Int64/UInt64 column rewritten as  lo <= (x ^ flip) <= hi  (xor negate)
struct FastPred {
    int64_t lo, hi;
    uint64_t flip;
    int32_t negate;
    // where the tested word of row r comes from: word column → src[r]; Boolean bitmap → (src[r >> 6] >> (r & 63)) & 1
    int32_t row_shift; // 0 | 6
    int32_t bit_mask;  // 0 | 63
    int32_t pad;
    uint64_t val_mask; // ~0 | 1
    // Float64 operands: x ^= (x >> 63 arithmetic) & fmask with fmask = 0x7fff…f maps IEEE doubles to signed integers in
    // the same order (negative values reversed); NaNs land beyond ±inf and are excluded by [lo, hi].  0 for integers.
    uint64_t fmask;
};
// `A and B [and C [and D]]` / the same with `or`: up to four range tests `col cmp lit` over non-null 8-byte columns — a WHERE
// clause's usual shape — tested per row inside the consuming kernel (the aggregate's streaming kernel when the columns are its key
// column, its first value column and at most one more; the selection's keep-mask kernel over up to four columns) instead of
// through a materialised Boolean column (one more pass over the predicate's columns)
constexpr int CONJ_MAX = 4;
struct ConjTest {
    int64_t lo, hi;
    uint64_t flip;  // sign bit for UInt64 operands
    uint64_t fmask; // Float64 operands: order map (see FastPred::fmask), 0 for integers
    int32_t negate;
    int32_t src;    // which loaded word of the row (aggregate: 0 key column, 1 first value column, 2 the predicate column;
                    // selection: the column's slot among the distinct tested columns)
    // one fault-free arithmetic step on the word ahead of the range test — `id % 3 = 0`, `v * 2.0 > 100.0`, `100 - w >= 7`:
    // pre = 0: none; else the nqe_operator (PLUS … MODULOS) over operands of type pre_dt, the literal on the right unless pre_rev
    int32_t pre, pre_dt, pre_rev, pad;
    uint64_t pre_lit;
    OpAux pre_aux;
};
struct ConjPred {
    ConjTest t[CONJ_MAX];
    int32_t n;       // tests
    int32_t is_or;   // (lists only)
    int32_t need_pw; // aggregate: some test reads the predicate column (src == 2)
    // general = 0: an and-list / or-list of plain range tests (the straight-line form).  general = 1: ANY nesting of and / or over
    // the tests, some of them with an arithmetic step: the tests' outcomes index the truth table (bit i of `truth`: the predicate's
    // value when test k's outcome is bit k of i)
    int32_t general;
    uint32_t truth;
    int32_t pad;
};
// Any other fault-free predicate tree over at most three non-null 8-byte columns (`v < 20 or id % 3 = 0`, `a + b > c`, …): a
// register stack machine runs it per row INSIDE the consuming kernel (the aggregate's streaming kernel: the tested columns are
// its key column, its first value column and at most one more) instead of a pass that materialises a Boolean column.  One
// instruction per BINARY node, post-order; operands: the stack (depth <= 2), a literal, or one of the row's loaded words.
constexpr int TREE_MAX_INSTR = 12, TREE_MAX_COLS = 3;
enum TreeSrc : int32_t { TS_STACK = 0, TS_LIT = 1, TS_W0 = 4 /* + word slot */ };
struct TreeInstr {
    int32_t op, dt;       // operator, operand dtype
    int32_t a_src, b_src; // TreeSrc
    uint64_t lit_a, lit_b;
    OpAux aux;            // host-prepared divisor constants when b is a literal
};
struct TreePred {
    int32_t n, ncols;
    TreeInstr ins[TREE_MAX_INSTR];
    int32_t col[TREE_MAX_COLS]; // table column behind word slot k of the program as built (the consumer renumbers the slots)
};
// false: the tree does not fit (nullable / non-word columns, NULL or Utf8 literals, more columns or instructions, a deeper stack, a
// divisor that is not a literal other than 0 and -1 — anything that could raise a device flag)
bool match_tree_pred(const nqe_table *in, const nqe_expr_node *nodes, int n, TreePred *out);
// recognises the shape (see expr.hip); cols[CONJ_MAX] receive the tested column of every test
bool match_conj(const nqe_table *in, const nqe_expr_node *nodes, int n, ConjPred *out, int *cols);
// FastPred "bit r of a non-null Boolean bitmap is set"
FastPred bitmap_fast_pred();
// returns false when the SimpleExpr is not a single Int64/UInt64/Float64 compare against a literal
bool make_fast_pred(const SimpleExpr &pe, FastPred *fp);

struct ExprInfo {
    int out_dtype = NQE_NULLTYPE;
    bool simple = false;
    SimpleExpr s{};
    // true when evaluating it can raise a device error flag (a divide/modulus whose divisor is not a literal other than
    // 0 and -1): only then does an operator need the flag read-back, which is a stream synchronisation
    bool may_fault = false;
};
// type-checks (IntervalError / NotSupported exactly where the reference raises them) and
// recognises the fusable shape
ExprInfo analyze_expr(const nqe_table *in, const nqe_expr_node *nodes, int n);
// general evaluator: one streaming kernel per binary node, literals kept scalar
DevColumn evaluate_expr(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, int n);

// ---- filter (selection.hip) ---------------------------------------------------------------
struct KeepMask {
    BufRef keep;          // bit i = row i is emitted (pred true OR pred NULL)
    BufRef pvalid;        // predicate validity (null if predicate has no nulls)
    BufRef tile_offsets;  // uint64 per 4096-row tile (exclusive), [ntiles] = total
    int64_t n = 0;        // rows covered
    int64_t ntiles = 0;
    int64_t total = 0;    // emitted rows
    mutable BufRef kept_idx; // int64 source row per emitted row (-1: NULL row from a NULL predicate); built on demand for Utf8
};
// source-row list of the emitted rows (cached in the mask)
const int64_t *kept_rows(nqe_ctx *ctx, const KeepMask &km);
KeepMask build_keep_mask(nqe_ctx *ctx, const DevColumn &pred, int64_t n_rows);
// predicate given as a fused SimpleExpr over `in`
KeepMask build_keep_mask_simple(nqe_ctx *ctx, const nqe_table *in, const SimpleExpr &pred);
DevColumn compact_column(nqe_ctx *ctx, const DevColumn &src, const KeepMask &km);
// out row r (r-th kept row i) = src[gidx[i]]
DevColumn compact_gather_column(nqe_ctx *ctx, const DevColumn &src, const uint32_t *gidx, const KeepMask &km);
// scans the per-tile counts into km.tile_offsets and reads back km.total
KeepMask finish_mask(nqe_ctx *ctx, KeepMask km, BufRef tile_counts);
// Utf8 comparison; a null column pointer means that side is the literal (lit, lit_null)
DevColumn utf8_compare(nqe_ctx *ctx, int op, const DevColumn *lcol, const std::string &llit, bool llit_null, const DevColumn *rcol,
                       const std::string &rlit, bool rlit_null, int64_t n);
DevColumn utf8_literal_column(nqe_ctx *ctx, const std::string &lit, bool lit_null, int64_t n);
// general tree `nodes` over the rows of `km`, compacted in the same pass; false = does not fit the stack machine
bool evaluate_expr_compacted(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, int n, const KeepMask &km, DevColumn *result);
// the whole projection list behind a selection in ONE run-time specialised kernel (expr_jit.hpp); false when the list does not fit it
// or its kernel is not compiled yet (the caller then takes the per-expression path)
bool project_specialised(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, const int32_t *expr_offsets, int num_exprs, const KeepMask &km,
                         std::vector<DevColumn> *out);
// selection + projection in one run-time specialised pass (tree predicate, inputs without NULLs, word-typed outputs); false: take the mask + compaction path
bool select_project_fused(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *nodes, const int32_t *expr_offsets,
                          int num_exprs, std::vector<DevColumn> *result, int64_t *total_out);
// the streaming aggregate of `val_col` grouped by a key `… % m` (any fault-free integer chain or tree ending in a literal modulus), under an
// optional predicate, through the lean run-time specialised kernel (expr.hip); on success *partials holds every workgroup's table for
// aggregate.hip's merge kernel
bool aggregate_tree_specialised(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *group, int group_nodes,
                                int val_col, int grid, BufRef *partials, uint32_t *span_out, int64_t *bias_out, bool dry_run = false);
// evaluates `e` over `in` and compacts the result in the same pass
DevColumn compact_simple_expr(nqe_ctx *ctx, const nqe_table *in, const SimpleExpr &e, const KeepMask &km);

} // namespace nqe

// wraps a C entry point: exceptions → status + ctx->last_error
#define NQE_API_BEGIN(ctxptr)                                                                                \
    nqe_ctx *_api_ctx = (ctxptr);                                                                            \
    try {                                                                                                    \
        if (_api_ctx) (void)hipSetDevice(_api_ctx->device);
#define NQE_API_END()                                                                                        \
    }                                                                                                        \
    catch (const ::nqe::Error &e) {                                                                          \
        if (_api_ctx) _api_ctx->last_error = e.msg;                                                          \
        ::nqe::set_global_error(e.msg);                                                                      \
        return (nqe_status)e.code;                                                                           \
    }                                                                                                        \
    catch (const std::bad_alloc &) {                                                                         \
        if (_api_ctx) _api_ctx->last_error = "host out of memory";                                           \
        return NQE_ERR_OUT_OF_MEMORY;                                                                        \
    }                                                                                                        \
    catch (const std::exception &e) {                                                                        \
        if (_api_ctx) _api_ctx->last_error = e.what();                                                       \
        return NQE_ERR_OTHERS;                                                                               \
    }                                                                                                        \
    return NQE_OK;

namespace nqe {
void set_global_error(const std::string &msg);
}
