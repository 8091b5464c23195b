/*
 * nqe.h — C ABI of the MI355X-native physical execution layer for
 * naive-query-engine's hot operators (filter, projection, hash group-by
 * aggregate, inner hash join over Arrow-layout Int64/UInt64/Float64/Boolean/Utf8
 * columns) and of the CSV ingest that feeds them.
 *
 * The reference (Veeupup/naive-query-engine, Rust) has NO FFI; its operator
 * boundary is the in-crate trait
 *
 *     trait PhysicalPlan { schema(); execute() -> Result<Vec<RecordBatch>>; children() }
 *                                              (src/physical_plan/plan.rs:14-21)
 *
 * and the arrow-rs compute kernels its operators call.  Every entry point
 * below replaces one `execute()` body (or one arrow-rs call site inside it)
 * and cites it.  A Rust `impl PhysicalPlan for Gpu*Plan` binds these through
 * `extern "C"` (see INTEGRATION.md for the shim).
 *
 * Conventions
 *   - plain C, no C++/torch types; all structs are POD and passed by pointer.
 *   - every function returns nqe_status (0 = OK); the numeric codes 1..13
 *     are 1:1 with the reference's `ErrorCode` variants (src/error.rs:13-40);
 *     nqe_last_error(ctx) returns a message for the last failure on ctx.
 *   - no call aborts or throws across the ABI: the reference's panics
 *     (`unimplemented!()` selection.rs:98, binary.rs:85) map to
 *     NQE_ERR_NOT_SUPPORTED.
 *   - columns use the Arrow columnar layout (what `RecordBatch` columns are):
 *     64-bit little-endian values, LSB-first validity bitmap, Boolean values
 *     bit-packed LSB-first, Utf8 = int32 offsets[length+1] + bytes.
 *   - inputs are borrowed for the duration of the call and never mutated;
 *     outputs are nqe_table handles owned by the caller (nqe_table_release).
 *     Tables are immutable and their columns may share device buffers the
 *     LIBRARY owns (reference-counted: a projection of a column, the two key
 *     columns of an equi-join on an integer key are one buffer; the probe-side
 *     columns of a join on unique build keys in which every probe row matched
 *     are the probe table's own buffers): never write through nqe_table_column.
 *     Borrowed NQE_DEVICE memory (nqe_table_create) is never aliased by an
 *     operator's output — it is the caller's to free or overwrite as soon as
 *     the operator has returned and the stream has been synchronised — unless
 *     the table was created with NQE_TABLE_IMMUTABLE.  The explicit zero-copy
 *     views (nqe_table_project) borrow what their input borrows.
 *   - a context owns one HIP stream on one device and is used by one host
 *     thread at a time (the reference is single-threaded).
 *   - calls are stream-ordered: an operator may return while its last kernels
 *     still run on the context's stream; every later call on the same context,
 *     every download and nqe_ctx_synchronize order after them.  Errors that
 *     depend on device data (DivideByZero, overflow) are still reported by the
 *     call that causes them: a call whose expressions can raise one reads the
 *     device flags back (and thereby synchronises) before returning.  Device
 *     pointers from nqe_table_column may be handed to another stream only after
 *     nqe_ctx_synchronize.
 *   - when a call has to wait for the device (a row count, the flags) the host
 *     thread polls the stream for up to 5 ms before it blocks: the waits sit
 *     behind kernels of microseconds to a few milliseconds, and an
 *     interrupt-driven wake-up would add its latency to every operator.
 */
#ifndef NQE_H
#define NQE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NQE_ABI_VERSION 1

/* ---- status codes: 1..13 mirror `enum ErrorCode` (src/error.rs:13-40) ---- */
typedef enum nqe_status {
    NQE_OK = 0,
    NQE_ERR_ARROW = 1,            /* ErrorCode::ArrowError (DivideByZero, length mismatch, overflow) */
    NQE_ERR_IO = 2,               /* ErrorCode::IoError */
    NQE_ERR_NO_SUCH_FIELD = 3,    /* ErrorCode::NoSuchField */
    NQE_ERR_COLUMN_NOT_EXISTS = 4,/* ErrorCode::ColumnNotExists */
    NQE_ERR_LOGICAL = 5,          /* ErrorCode::LogicalError (column.rs:26,54) */
    NQE_ERR_NO_SUCH_TABLE = 6,    /* ErrorCode::NoSuchTable */
    NQE_ERR_PARSER = 7,           /* ErrorCode::ParserError */
    NQE_ERR_INTERVAL = 8,         /* ErrorCode::IntervalError (binary.rs:115: operand type mismatch) */
    NQE_ERR_PLAN = 9,             /* ErrorCode::PlanError (hash_join.rs:126: empty `on`) */
    NQE_ERR_NO_MATCH_FUNCTION = 10,/* ErrorCode::NoMatchFunction */
    NQE_ERR_NOT_SUPPORTED = 11,   /* ErrorCode::NotSupported (aggregate/mod.rs:217, sum.rs:93) + reference panics */
    NQE_ERR_NOT_IMPLEMENTED = 12, /* ErrorCode::NotImplemented (hash_join.rs:161) */
    NQE_ERR_OTHERS = 13,          /* ErrorCode::Others */
    /* codes with no reference analogue */
    NQE_ERR_HIP = 100,            /* a HIP runtime call failed */
    NQE_ERR_RCCL = 101,           /* a collective / the exchange transport failed (nqe_comm_*, nqe_sharded_*) */
    NQE_ERR_INVALID_ARGUMENT = 102,
    NQE_ERR_OUT_OF_MEMORY = 103
} nqe_status;

/* ---- Arrow data types the hot path accepts (selection.rs:69-99, binary.rs:48-86) ---- */
typedef enum nqe_dtype {
    NQE_NULLTYPE = 0, /* DataType::Null (ScalarValue::Null) */
    NQE_BOOLEAN = 1,
    NQE_INT64 = 2,
    NQE_UINT64 = 3,
    NQE_FLOAT64 = 4,
    NQE_UTF8 = 5
} nqe_dtype;

typedef enum nqe_location { NQE_HOST = 0, NQE_DEVICE = 1 } nqe_location;

/* One Arrow array (a RecordBatch column). */
typedef struct nqe_column {
    int32_t dtype;           /* nqe_dtype */
    int32_t location;        /* nqe_location: where values/validity/data point */
    int64_t length;          /* rows */
    int64_t null_count;      /* -1 = unknown (count validity) */
    const void *values;      /* Int64/UInt64/Float64: length*8 B; Boolean: ceil(length/8) B, LSB-first;
                                Utf8: int32 offsets[length+1] */
    const uint8_t *validity; /* ceil(length/8) B LSB-first bitmap, NULL = no nulls */
    const void *data;        /* Utf8 bytes, else NULL */
    int64_t data_length;     /* Utf8 byte count, else 0 */
} nqe_column;

/* ---- expressions: flat post-order encoding of a PhysicalExpr tree ----
 * mirrors ColumnExpr (expression/column.rs:18-29, index form — the planner
 * resolves names to the first matching index, planner/mod.rs:190-200),
 * PhysicalLiteralExpr(ScalarValue) (expression/literal.rs:17-26,
 * logical_plan/expression.rs:174-187) and PhysicalBinaryExpr(l, Operator, r)
 * (expression/binary.rs:91-101). A tree is valid when evaluating the nodes
 * left-to-right on a stack leaves exactly one value. */
typedef enum nqe_expr_kind { NQE_EXPR_COLUMN = 0, NQE_EXPR_LITERAL = 1, NQE_EXPR_BINARY = 2 } nqe_expr_kind;

/* same order as `enum Operator` (logical_plan/expression.rs:335-362) */
typedef enum nqe_operator {
    NQE_OP_EQ = 0,
    NQE_OP_NOT_EQ = 1,
    NQE_OP_LT = 2,
    NQE_OP_LT_EQ = 3,
    NQE_OP_GT = 4,
    NQE_OP_GT_EQ = 5,
    NQE_OP_PLUS = 6,
    NQE_OP_MINUS = 7,
    NQE_OP_MULTIPLY = 8,
    NQE_OP_DIVIDE = 9,
    NQE_OP_MODULOS = 10,
    NQE_OP_AND = 11,
    NQE_OP_OR = 12
} nqe_operator;

typedef struct nqe_expr_node {
    int32_t kind;    /* nqe_expr_kind */
    int32_t op;      /* BINARY: nqe_operator */
    int32_t column;  /* COLUMN: index into the input batch */
    int32_t dtype;   /* LITERAL: nqe_dtype of the ScalarValue */
    int32_t is_null; /* LITERAL: 1 = ScalarValue::X(None) */
    int32_t utf8_length; /* LITERAL of dtype NQE_UTF8: byte length of value.utf8 (the field was `reserved` before Utf8
                            literals existed; the struct layout is unchanged) */
    union {
        int64_t i64;
        uint64_t u64;
        double f64;
        int64_t boolean;  /* 0 / 1 */
        const char *utf8; /* ScalarValue::Utf8(Some(s)): borrowed for the duration of the call, not NUL-terminated */
    } value;
} nqe_expr_node;

/* same order as `enum AggregateFunc` (logical_plan/expression.rs:491-502) */
typedef enum nqe_agg_func {
    NQE_AGG_COUNT = 0,
    NQE_AGG_SUM = 1,
    NQE_AGG_MIN = 2,
    NQE_AGG_MAX = 3,
    NQE_AGG_AVG = 4
} nqe_agg_func;

/* Sum/Avg/Count/Min/Max::create(ColumnExpr) (aggregate/{sum,avg,count,min,max}.rs): the
 * argument is always a bare column (planner/mod.rs:107-114). */
typedef struct nqe_aggregate {
    int32_t func;   /* nqe_agg_func */
    int32_t column; /* input column index */
} nqe_aggregate;

typedef struct nqe_ctx nqe_ctx;     /* one device + one HIP stream + scratch pool */
typedef struct nqe_table nqe_table; /* one device-resident RecordBatch */

/* ------------------------------------------------------------------ context */
uint32_t nqe_abi_version(void);
/* `stream` = an existing hipStream_t to launch on (e.g. torch's current stream), or NULL to
 * create a private one. */
nqe_status nqe_ctx_create(int32_t device, void *stream, nqe_ctx **out);
nqe_status nqe_ctx_destroy(nqe_ctx *ctx);
nqe_status nqe_ctx_synchronize(nqe_ctx *ctx);
/* Device memory of the context: bytes held by live tables/handles, and bytes cached in the context's block pool
 * (released blocks are reused stream-ordered; a failed hipMalloc trims the pool and retries).  nqe_ctx_trim returns
 * the cached blocks to the driver (after a stream synchronisation). */
nqe_status nqe_ctx_memory_stats(nqe_ctx *ctx, int64_t *live_bytes, int64_t *pooled_bytes);
nqe_status nqe_ctx_trim(nqe_ctx *ctx);
/* Takes `bytes` of device memory from the driver NOW (one block, pages touched) and serves later allocations of the context —
 * operator outputs, scratch — from it (best fit, freed ranges coalesced; what does not fit falls back to the pool / the driver).
 * A first query then pays no hipMalloc: the reference's run_sql is one-shot (db.rs:24-37), and a first hipMalloc of a gigabyte
 * output costs as much as the query.  Once per context; released by nqe_ctx_destroy (nqe_ctx_trim leaves it alone).  The free part
 * counts as `pooled_bytes`.  NQE_RESERVE_MB=<n> in the environment reserves at nqe_ctx_create. */
nqe_status nqe_ctx_reserve(nqe_ctx *ctx, size_t bytes);
const char *nqe_last_error(const nqe_ctx *ctx);
/* global message for failures with no ctx (nqe_ctx_create itself) */
const char *nqe_last_global_error(void);

/* per-kernel timing with HIP events on the ctx stream (bench.py's roofline leg).
 * While enabled every launch of the named kernel family is bracketed by events. */
nqe_status nqe_ctx_timing_enable(nqe_ctx *ctx, int32_t enable);
/* sum of durations (ms) and launch count since the last reset for kernels whose name
 * contains `name_substr`; resets nothing. */
nqe_status nqe_ctx_timing_query(nqe_ctx *ctx, const char *name_substr, double *total_ms, int64_t *launches);
nqe_status nqe_ctx_timing_reset(nqe_ctx *ctx);
/* every kernel family launched since the last reset by exact name, one "name\tms\tlaunches\n" line each, NUL-terminated;
 * *needed = bytes required (call with capacity 0 to size the buffer). */
nqe_status nqe_ctx_timing_report(nqe_ctx *ctx, char *buf, int64_t capacity, int64_t *needed);
/* Expression trees of three or more operators over large inputs (binary.rs:108-155 nested) are, besides being interpreted by the
 * stack machine, specialised at run time: the tree becomes straight-line HIP source compiled with hipRTC on a worker thread, and
 * executions of the same tree shape switch to the compiled kernel once it is ready (results are identical; without libhiprtc the
 * interpreter simply stays).  This call blocks until every compilation in flight for the context has finished — for tests and
 * benchmarks that want the steady state.  NQE_NO_JIT=1 disables the specialisation, NQE_JIT_SYNC=1 compiles before the first
 * execution.  Code objects are kept on disk (NQE_JIT_CACHE_DIR, default ~/.cache/nqe_jit; NQE_NO_JIT_DISK_CACHE=1: not) with the
 * source they were compiled from, so that a new process takes a known tree's kernel on its first execution. */
nqe_status nqe_ctx_jit_wait(nqe_ctx *ctx);

/* ------------------------------------------------------------------ tables
 * MemTable::try_create + ScanPlan::execute (datasource/memory.rs:21-41, scan.rs:34-36):
 * columns enter HBM once; NQE_HOST columns are copied to the device, NQE_DEVICE
 * columns are borrowed (zero-copy; the caller keeps them alive). All columns must
 * have the same length. */
nqe_status nqe_table_create(nqe_ctx *ctx, const nqe_column *columns, int32_t num_columns, nqe_table **out);
/* The same with flags.  NQE_TABLE_IMMUTABLE: the caller promises that the borrowed NQE_DEVICE buffers stay alive and
 * unmodified for as long as ANY table derived from this one lives (what `Arc`-shared Arrow buffers give the reference for
 * free): operator outputs may then alias them instead of copying — the probe-side columns of a join in which every probe
 * row matched, a projected bare column.  Without it every output of an operator is memory the library owns. */
#define NQE_TABLE_IMMUTABLE 1u
nqe_status nqe_table_create_flags(nqe_ctx *ctx, const nqe_column *columns, int32_t num_columns, uint32_t flags, nqe_table **out);
nqe_status nqe_table_release(nqe_table *table);
int64_t nqe_table_num_rows(const nqe_table *table);
int32_t nqe_table_num_columns(const nqe_table *table);
/* describes column i with DEVICE pointers (valid until the table is released) */
nqe_status nqe_table_column(const nqe_table *table, int32_t i, nqe_column *out);
/* copies column i to host buffers sized per the layout above; validity_out may be NULL when
 * the column has no validity bitmap (see nqe_table_column); data_out only for Utf8. Blocking. */
nqe_status nqe_table_download_column(const nqe_table *table, int32_t i, void *values_out,
                                     uint8_t *validity_out, void *data_out);
/* MemTable::scan(Some(projection)) → RecordBatch::project (memory.rs:31-41): zero-copy */
nqe_status nqe_table_project(nqe_ctx *ctx, const nqe_table *in, const int32_t *indices, int32_t n,
                             nqe_table **out);
/* RecordBatch::slice as used by PhysicalLimitPlan/PhysicalOffsetPlan (limit.rs:32-49,
 * offset.rs:30-51). Copies (device-to-device) so that bitmaps stay offset-free. */
nqe_status nqe_table_slice(nqe_ctx *ctx, const nqe_table *in, int64_t offset, int64_t length,
                           nqe_table **out);
/* concat_batches (hash_join.rs:258-273): column-wise concatenation; n == 0 is
 * NQE_ERR_INVALID_ARGUMENT here (the host mirror builds the empty batch itself). */
nqe_status nqe_table_concat(nqe_ctx *ctx, const nqe_table *const *tables, int32_t n, nqe_table **out);

/* ------------------------------------------------------------------ Arrow C Data Interface
 * MemTable::try_create(schema, batches) (datasource/memory.rs:21-29; Catalog::add_memory_table, catalog.rs:40-49) for hosts that hold
 * real Arrow data, and the way back for results (the Vec<RecordBatch> of plan.rs:18).  The structs are the standard ones of the
 * Arrow C Data Interface (arrow-rs `arrow::ffi`, pyarrow `_export_to_c` / `_import_from_c`); include Arrow's own header first if
 * you have it. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif
/* One RecordBatch = one struct array (format "+s") whose children are the columns: "l" Int64, "L" UInt64, "g" Float64, "b" Boolean,
 * "u" Utf8; anything else is NQE_ERR_NOT_SUPPORTED (selection.rs:98).  Sliced arrays (offset != 0) are honoured.  The buffers are
 * host memory; they are copied to HBM and `array` is then RELEASED by this call (the interface's move semantics — on failure it is
 * left untouched); `schema` is only read. */
nqe_status nqe_table_import_arrow(nqe_ctx *ctx, struct ArrowArray *array, const struct ArrowSchema *schema, nqe_table **out);
/* The table as one struct array with host copies of its buffers, owned by the returned structs until their `release` callbacks
 * run.  names: one per column, or NULL ("c0", "c1", ...).  Blocking. */
nqe_status nqe_table_export_arrow(const nqe_table *table, const char *const *names, struct ArrowArray *out_array,
                                  struct ArrowSchema *out_schema);

/* ------------------------------------------------------------------ CSV ingest
 * CsvTable::try_create (datasource/csv.rs:53-86) = infer_schema_from_csv (csv.rs:76-85, arrow-rs 13
 * csv::reader::infer_reader_schema over the first max_read_records records) + csv::Reader::next() — only the FIRST
 * batch of batch_size rows is kept (quirk Q1, csv.rs:71-73).  CsvConfig (csv.rs:23-43): has_header = true,
 * delimiter = ',', max_read_records = Some(3), batch_size = 1_000_000; file_projection / datetime_format are not mirrored. */
typedef struct nqe_csv_options {
    int32_t has_header;       /* first record = column names; otherwise "column_1", "column_2", ... */
    int32_t delimiter;        /* one byte */
    int64_t max_read_records; /* records sampled by the inference; < 0 = all */
    int64_t batch_size;       /* rows kept; < 0 = all */
} nqe_csv_options;
/* Schema inference (host side, a handful of records): per column Boolean / Int64 / Float64 / Utf8 (Int64+Float64 →
 * Float64, any other mix → Utf8; Date-like columns → NQE_ERR_NOT_SUPPORTED), nullable[c] = an empty field was seen.
 * names: the column names joined with '\0' terminators (names_bytes = bytes needed). */
nqe_status nqe_csv_infer_schema(nqe_ctx *ctx, const void *bytes_host, int64_t nbytes, const nqe_csv_options *opt,
                                int32_t max_columns, int32_t *num_columns, int32_t *dtypes, int32_t *nullable,
                                char *names, int64_t names_capacity, int64_t *names_bytes);
/* Parses the file image (host or device memory, `location` = nqe_location) into one device table with the given column
 * types: records split on the GPU (quotes with "" escapes, \r / \n / \r\n, empty lines skipped), fields converted on
 * the GPU (lexical-core semantics: Int64 overflow and malformed numbers are NQE_ERR_ARROW, Float64 correctly rounded,
 * empty numeric/Boolean fields are NULL, Utf8 fields are never NULL); a record with a different number of fields is
 * NQE_ERR_ARROW. */
nqe_status nqe_csv_read(nqe_ctx *ctx, const void *bytes, int32_t location, int64_t nbytes, const nqe_csv_options *opt,
                        const int32_t *dtypes, int32_t num_columns, nqe_table **out);

/* ------------------------------------------------------------------ exchange plumbing (multi-GPU)
 * No reference analogue (the reference is single-process).  A rank's partial aggregate (keys + state tables: 8-byte
 * columns without validity) is packed into ONE device buffer so that the exchange is a single RCCL all-gather, and the
 * gathered buffer is unpacked into one concatenated table for nqe_aggregate_merge.
 *
 * pack:   dst[c * stride_rows + r] = word r of column c (columns of `tables` in order, c over all tables);
 *         dst[num_columns * stride_rows] = number of rows (the tables must agree) — the header the peers read.
 *         Every table must have rows <= stride_rows; dst holds num_columns * stride_rows + 1 words.
 * unpack: src = num_parts such buffers back to back; out column c = for p in 0..num_parts: src_p[c][0 .. counts[p]).
 *         counts[p] is what the caller read from the headers (host memory). */
nqe_status nqe_table_pack_words(nqe_ctx *ctx, const nqe_table *const *tables, int32_t num_tables, int64_t stride_rows,
                                void *dst_device);
nqe_status nqe_table_unpack_words(nqe_ctx *ctx, const void *src_device, int32_t num_parts, int32_t num_columns,
                                  int64_t stride_rows, const int64_t *counts, const int32_t *dtypes, nqe_table **out);

/* ------------------------------------------------------------------ sharded operators (multi-GPU)
 * No reference analogue.  One process (or thread) per GPU, each with its own context and a communicator over it; every rank
 * calls the same sharded entry point with ITS row range.  Collectives are enqueued on the context's stream (no host
 * synchronisation between an operator's kernels and its exchange); failures of the transport are NQE_ERR_RCCL.
 *
 * The default transport is RCCL over xGMI, bound at run time (librccl.so.1 is dlopen'ed by the first nqe_comm_* call, never
 * by single-GPU use): rank 0 calls nqe_comm_get_unique_id, the host distributes the 128 bytes by whatever it has (MPI, a
 * store, torch.distributed), every rank calls nqe_comm_create.  nqe_comm_create_custom plugs in the host's own collectives
 * (all_gather / all_gather_v), nqe_comm_create_p2p the host's own point-to-point primitives (send / recv / group brackets —
 * MPI, a test fabric): the library then builds its collectives from them with the very code it runs over RCCL's ncclSend /
 * ncclRecv.
 *
 * Nobody blocks on a failed rank: every sharded entry point runs its fallible local work (the partial aggregate, the probe, the
 * selection: data-dependent DivideByZero, out of memory, ...) first, and a rank that fails there STILL enters the operator's
 * exchange, with its status in the header every exchange starts with — so that every rank returns an error together (the
 * failing rank its own, the others NQE_ERR_RCCL naming the rank and its status) instead of blocking in a collective the failed
 * rank never joined.  A failure AFTER the exchange (the merge of the gathered partials, a Utf8 re-encode: out of memory on one
 * rank) is that rank's alone: the others have their result and return NQE_OK — statuses may then differ between ranks, but no
 * rank waits for another. */
#define NQE_COMM_ID_BYTES 128
/* rows of the fixed-size buffer a partial aggregate travels in (one collective, counts read on the device); larger partials take
 * an exact-size two-step exchange */
#define NQE_EXCHANGE_ROWS 4096
typedef struct nqe_comm nqe_comm;
typedef struct nqe_transport {
    void *user;
    /* every rank contributes `bytes` device bytes; recv (device) receives world*bytes ordered by rank.  Enqueue on `stream`
     * (a hipStream_t) or complete before returning.  0 = success. */
    int32_t (*all_gather)(void *user, const void *send, void *recv, size_t bytes, void *stream);
    /* variable-length form: rank r's contribution (recv_bytes[r] bytes) lands at recv + recv_offsets[r]; send_bytes ==
     * recv_bytes[own rank].  Arrays are host memory, valid for the duration of the call. */
    int32_t (*all_gather_v)(void *user, const void *send, size_t send_bytes, void *recv, const size_t *recv_offsets,
                            const size_t *recv_bytes, void *stream);
    /* optional (may be NULL): brackets around the all_gather_v calls of one table (RCCL: ncclGroupStart / ncclGroupEnd) */
    int32_t (*group_begin)(void *user);
    int32_t (*group_end)(void *user);
    /* optional: called by nqe_comm_destroy */
    void (*destroy)(void *user);
} nqe_transport;
nqe_status nqe_comm_get_unique_id(void *id_out /* NQE_COMM_ID_BYTES */);
nqe_status nqe_comm_rccl_version(int32_t *version_out);
nqe_status nqe_comm_create(nqe_ctx *ctx, const void *unique_id, int32_t rank, int32_t world, nqe_comm **out);
nqe_status nqe_comm_create_custom(nqe_ctx *ctx, const nqe_transport *transport, int32_t rank, int32_t world, nqe_comm **out);
/* Point-to-point primitives with RCCL's semantics: send/recv are enqueued on `stream` (or complete before group_end returns),
 * a send is matched by the peer's recv of the SAME byte count in the same order per (sender, receiver) pair, calls between
 * group_begin and group_end (which may nest) are issued together, so that a rank may post its sends and receives in any order
 * without deadlock.  0 = success.  destroy is optional (called by nqe_comm_destroy). */
typedef struct nqe_p2p {
    void *user;
    int32_t (*send)(void *user, const void *buf, size_t bytes, int32_t peer, void *stream);
    int32_t (*recv)(void *user, void *buf, size_t bytes, int32_t peer, void *stream);
    int32_t (*group_begin)(void *user);
    int32_t (*group_end)(void *user);
    void (*destroy)(void *user);
} nqe_p2p;
nqe_status nqe_comm_create_p2p(nqe_ctx *ctx, const nqe_p2p *p2p, int32_t rank, int32_t world, nqe_comm **out);
nqe_status nqe_comm_destroy(nqe_comm *comm);
int32_t nqe_comm_rank(const nqe_comm *comm);
int32_t nqe_comm_world(const nqe_comm *comm);
/* Ordered variable-length all-gather of a per-rank result table: out = the ranks' tables concatenated in rank order (= row
 * order for row-range shards), on every rank — what `concat_batches` (hash_join.rs:258-273) would make of them.  Every column
 * type of the path travels: 8-byte words move peer to peer straight from the local table into their place in `out` (columns
 * that share a buffer locally are moved once and share it in `out`); validity bitmaps and Boolean values are received per rank
 * and shifted to their bit offset (a column is nullable in `out` when it is on any rank); Utf8 bytes land in place and the
 * offsets are rebased.  A fixed-size header (status, rows, per-column flags and byte counts) travels first: the one host wait. */
nqe_status nqe_table_all_gather(nqe_comm *comm, const nqe_table *local, nqe_table **out);
/* PhysicalAggregatePlan::execute (aggregate/mod.rs:113-222) over the union of every rank's `in`: per-rank partial state
 * {count,sum,min,max} → ONE all-gather → merge on every rank; avg is finalised after the merge.  Same output contract as
 * nqe_aggregate_execute on the concatenated input (f64 sums in a different order: 1e-9 relative).  Utf8 group keys
 * (aggregate/mod.rs:170-216) travel as strings: the partials' (key string, state) rows are all-gathered and merged by string;
 * the output is ordered by first appearance in rank order. */
nqe_status nqe_sharded_aggregate_execute(nqe_comm *comm, const nqe_table *in, const nqe_expr_node *pred, int32_t pred_nodes,
                                         const nqe_expr_node *group, int32_t group_nodes, const nqe_aggregate *aggs,
                                         int32_t num_aggs, nqe_table **out, nqe_table **keys_out);
/* HashJoin::probe (hash_join.rs:168-254) with the build side replicated (every rank built `build` from the whole left table)
 * and the probe side range-split: `right_local` is this rank's contiguous row range.  gather = 0: out = this rank's output
 * rows (rank order == probe row order); gather = 1: out = nqe_table_all_gather of them (the single-GPU result on every rank). */
typedef struct nqe_join_table nqe_join_table;
nqe_status nqe_sharded_hash_join_probe(nqe_comm *comm, const nqe_join_table *build, const nqe_table *right_local,
                                       int32_t right_key, int32_t gather, nqe_table **out);
/* Fused ProjectionPlan(SelectionPlan(input)) over a row-range shard, optionally gathered like the join. */
nqe_status nqe_sharded_selection_projection_execute(nqe_comm *comm, const nqe_table *in_local, const nqe_expr_node *pred,
                                                    int32_t pred_nodes, const nqe_expr_node *nodes, const int32_t *expr_offsets,
                                                    int32_t num_exprs, int32_t gather, nqe_table **out);

/* ------------------------------------------------------------------ expressions
 * PhysicalExpr::evaluate(batch).into_array() (expression/mod.rs:25-29, binary.rs:108-155,
 * datatype.rs:27-34): evaluates one expression over `in`, returns a 1-column table.
 * Literals are kept as scalars in registers (never materialised, cf. binary.rs:121 TODO),
 * except a root literal, which is expanded to `in.num_rows` rows as into_array does.
 * Compares (= != < <= > >=) work on every type incl. Utf8 (byte-wise lexicographic, as arrow's
 * *_dyn kernels); and/or are Kleene; arithmetic is wrapping on Int64/UInt64, IEEE on Float64.
 * A tree of binary nodes is evaluated in one pass over the columns it references.
 * Errors: operand dtype mismatch → NQE_ERR_INTERVAL; divide/modulus with a valid zero
 * divisor → NQE_ERR_ARROW; arithmetic on Boolean/Utf8 → NQE_ERR_NOT_SUPPORTED. */
nqe_status nqe_expr_evaluate(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes,
                             int32_t num_nodes, nqe_table **out);

/* ------------------------------------------------------------------ filter
 * build_array_by_predicate! over every column (selection.rs:34-51, :65-100): stable
 * compaction of all columns of `in` by a Boolean predicate column: true → keep the row,
 * false → drop, NULL → emit a NULL row (quirk Q4). Rows = min(pred.length, in.num_rows)
 * (iterator zip, quirk Q3). `predicate` is column `pred_column` of `pred_table`. */
nqe_status nqe_filter(nqe_ctx *ctx, const nqe_table *in, const nqe_table *pred_table,
                      int32_t pred_column, nqe_table **out);
/* SelectionPlan::execute for a single input batch (selection.rs:58-107) = evaluate + filter. */
nqe_status nqe_selection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred,
                                 int32_t pred_nodes, nqe_table **out);

/* ------------------------------------------------------------------ projection
 * ProjectionPlan::execute for one batch (projection.rs:43-70): one output column per
 * expression. `nodes` holds the expressions back to back; expression e occupies
 * nodes[expr_offsets[e] .. expr_offsets[e+1]). */
nqe_status nqe_projection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes,
                                  const int32_t *expr_offsets, int32_t num_exprs, nqe_table **out);
/* Fused ProjectionPlan(SelectionPlan(input)) for one batch: identical result to
 * nqe_selection_execute followed by nqe_projection_execute, but only the columns the
 * projection references are compacted and the expressions are evaluated in the
 * compaction kernel (C2 of BASELINE.json). */
nqe_status nqe_selection_projection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred,
                                            int32_t pred_nodes, const nqe_expr_node *nodes,
                                            const int32_t *expr_offsets, int32_t num_exprs,
                                            nqe_table **out);

/* ------------------------------------------------------------------ hash aggregate
 * PhysicalAggregatePlan::execute (aggregate/mod.rs:113-222) fused with an optional
 * SelectionPlan below it (`pred`, may be NULL/0) and the key expression group_expr[0]
 * (`group`, may be NULL/0 = un-grouped path :123-139).
 * Output `out`: one row per group (one row when un-grouped), one column per aggregate,
 * Float64 for sum/avg/min/max and UInt64 for count (sum.rs:29,115, count.rs:23,76), NO
 * key column (quirk Q8). Row order = ascending first-occurrence order of the key is NOT
 * promised: rows are sorted by key (the reference's order is HashMap-random; compare as
 * multisets). If `keys_out` != NULL it receives a 1-column table with the group keys in
 * the same row order (debug/merge aid, not part of the reference output).
 * NULL keys are dropped (:64); NULL values are skipped by every aggregate; a NULL
 * predicate emits a NULL row from the selection, which then has a NULL key/value (Q4).
 * Accumulation is f64 for every input type (`val as f64`, quirk Q10); max starts at
 * f64::MIN, min at f64::MAX, NaN ordering follows OrderedFloat (max.rs:30,48).
 * Key dtype must be Int64, UInt64 or Utf8 (a bare Utf8 column; keys_out then holds the strings); anything else is
 * NQE_ERR_NOT_SUPPORTED (aggregate/mod.rs:217). */
nqe_status nqe_aggregate_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred,
                                 int32_t pred_nodes, const nqe_expr_node *group, int32_t group_nodes,
                                 const nqe_aggregate *aggs, int32_t num_aggs, nqe_table **out,
                                 nqe_table **keys_out);
/* Partial-state form used to merge per-GPU / per-batch partials:
 * the output columns are the raw state of every DISTINCT aggregate column (slot v = the v-th
 * distinct `column` of `aggs` in order of first appearance; count,sum,avg,min,max over one
 * column exchange four words per group, not twenty), in this order for every slot v:
 *   4*v+0 count (UInt64, non-null values), 4*v+1 sum (Float64), 4*v+2 min (Float64),
 *   4*v+3 max (Float64; NaN if any NaN was seen),
 * plus keys in `keys_out` (required when grouped).  The merges take the same `aggs`. */
nqe_status nqe_aggregate_partial(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred,
                                 int32_t pred_nodes, const nqe_expr_node *group, int32_t group_nodes,
                                 const nqe_aggregate *aggs, int32_t num_aggs, nqe_table **state_out,
                                 nqe_table **keys_out);
/* Merges `n` partial (state, keys) pairs (e.g. all-gathered from all ranks, concatenated or
 * not) and finalises to the nqe_aggregate_execute output format. keys may be NULL for the
 * un-grouped form. */
nqe_status nqe_aggregate_merge(nqe_ctx *ctx, const nqe_table *const *states, const nqe_table *const *keys,
                               int32_t n, const nqe_aggregate *aggs, int32_t num_aggs, nqe_table **out,
                               nqe_table **keys_out);
/* The same merge straight from an all-gathered buffer of nqe_table_pack_words parts (part p = [key column when `grouped`] +
 * {count,sum,min,max} per distinct aggregate column, `stride_rows` words per column, + 1 header word = the part's row count): the counts
 * are read on the device, so the whole exchange + merge costs the host one wait.  When some part's header exceeds
 * `stride_rows` (its sender shipped the header only) the call returns NQE_OK with *out = NULL and the caller exchanges
 * exact-size tables instead (nqe_aggregate_merge).  Replaces the same merge loop as nqe_aggregate_merge. */
nqe_status nqe_aggregate_merge_packed(nqe_ctx *ctx, const void *gathered_device, int32_t num_parts, int64_t stride_rows,
                                      int32_t grouped, int32_t key_dtype, const nqe_aggregate *aggs, int32_t num_aggs,
                                      nqe_table **out, nqe_table **keys_out);

/* ------------------------------------------------------------------ hash join
 * HashJoin::execute = build() + probe() (hash_join.rs:124-254, :280-284) for one left
 * (build) batch and one right (probe) batch: inner equi-join on
 * left[left_key] == right[right_key]; output = all left columns gathered by build index
 * followed by all right columns gathered by probe index (`take`, :237-246); row order is
 * probe-row-major and, for duplicate build keys, ascending build row index (:86-101).
 * Key validity is ignored (quirk Q11): the raw 8-byte slot value is compared.
 * Key dtypes: both Int64, both UInt64 or both Utf8 (other types NQE_ERR_NOT_IMPLEMENTED, :161; differing types
 * NQE_ERR_NOT_SUPPORTED — the reference's downcast unwrap panics). */
nqe_status nqe_hash_join_execute(nqe_ctx *ctx, const nqe_table *left, const nqe_table *right,
                                 int32_t left_key, int32_t right_key, nqe_table **out);
/* Two-phase form: build once (replicated per GPU), probe many right batches / shards (nqe_join_table is declared with the
 * sharded operators above). */
nqe_status nqe_hash_join_build(nqe_ctx *ctx, const nqe_table *left, int32_t left_key, nqe_join_table **out);
nqe_status nqe_hash_join_probe(nqe_ctx *ctx, const nqe_join_table *build, const nqe_table *right,
                               int32_t right_key, nqe_table **out);
nqe_status nqe_join_table_release(nqe_join_table *jt);

/* ------------------------------------------------------------------ take
 * arrow::compute::take(array, &Int64Array indices, None) over every column
 * (hash_join.rs:239,245): out[j] = in[indices[j]]; `indices` = Int64 column `idx_column` of
 * `idx_table`, no nulls. */
nqe_status nqe_take(nqe_ctx *ctx, const nqe_table *in, const nqe_table *idx_table, int32_t idx_column,
                    nqe_table **out);

/* ------------------------------------------------------------------ synthetic columns
 * Deterministic generators of SURVEY §8d / BASELINE.md §3 so that 10^9-row tables never
 * cross PCIe: u(i,s) = splitmix64(s + i), i = first_row + row.
 *   NQE_SYNTH_ROWID      Int64   i
 *   NQE_SYNTH_UNIFORM    Int64   u(i,seed) mod `modulus`  (+ `base`)
 *   NQE_SYNTH_F64_0_100  Float64 (u(i,seed) >> 11) * 2^-53 * 100.0
 * Writes `n` 8-byte values to the DEVICE buffer `out`. */
typedef enum nqe_synth_kind { NQE_SYNTH_ROWID = 0, NQE_SYNTH_UNIFORM = 1, NQE_SYNTH_F64_0_100 = 2 } nqe_synth_kind;
nqe_status nqe_synth_fill(nqe_ctx *ctx, int32_t kind, uint64_t seed, int64_t first_row, int64_t n,
                          uint64_t modulus, int64_t base, void *out_device);
/* device allocation helpers for hosts that have no device allocator of their own */
nqe_status nqe_device_alloc(nqe_ctx *ctx, size_t bytes, void **out);
nqe_status nqe_device_free(nqe_ctx *ctx, void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* NQE_H */
