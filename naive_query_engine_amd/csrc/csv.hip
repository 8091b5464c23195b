// CSV → columnar ingest on the GPU: CsvTable::try_create (src/datasource/csv.rs:53-86).
//
// The reference reads the file through arrow-rs 13's csv::Reader (csv crate state machine: '"' quoting with "" escapes,
// '\r' / '\n' / "\r\n" terminators, empty lines skipped, every record must have the schema's number of fields), takes the
// schema from arrow::csv::reader::infer_reader_schema over the first max_read_records records (csv.rs:76-85, default 3),
// and keeps only the FIRST batch of batch_size rows (quirk Q1: `for record in reader.next()`, csv.rs:71-73).
//
// Device pipeline (the file bytes are uploaded once):
//   1. record splitting with a parallel DFA: every thread runs the 5-state quoting automaton over its 64-byte chunk for ALL
//      start states at once (a transition vector, 5 x 3 bits), vectors are composed by a workgroup scan + a scan over
//      workgroup totals, and a second walk with the now known start state counts, then writes, the record start / end
//      positions.  Quote parity tricks are not enough: a '"' is special only at the start of a field.
//   2. one thread per record walks its bytes with the same automaton, splits fields and converts them in place:
//      Int64 / Float64 / Boolean through csv_parse.hpp (lexical-core semantics, correctly rounded), Utf8 lengths;
//      empty numeric fields are NULL; a scan of the lengths gives the Utf8 offsets and a second walk copies the bytes
//      (with "" unescaped).
// Schema inference touches max_read_records records only and runs on the host (it decides the kernels' column types).
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "csv_parse.hpp"
#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {
namespace {

constexpr int CSV_CHUNK = 64;   // bytes per thread in the splitting passes
constexpr int CSV_BLOCK = 256;
constexpr int CSV_MAX_COLS = 64;

// quoting automaton of one record stream
enum : uint32_t { ST_R = 0 /* record start */, ST_F = 1 /* field start */, ST_U = 2 /* unquoted */, ST_Q = 3 /* quoted */, ST_E = 4 /* quote in quoted */ };
constexpr uint32_t VEC_ID = ST_R | (ST_F << 3) | (ST_U << 6) | (ST_Q << 9) | (ST_E << 12);

__host__ __device__ inline bool is_nl(uint8_t c) { return c == '\n' || c == '\r'; }

// next state; *rec_start: this byte opens a record; *rec_end: this byte (a terminator) closes one
__host__ __device__ inline uint32_t csv_step(uint32_t st, uint8_t c, uint8_t delim, bool *rec_start, bool *rec_end) {
    const bool nl = is_nl(c);
    *rec_start = st == ST_R && !nl;
    *rec_end = nl && (st == ST_F || st == ST_U || st == ST_E);
    switch (st) {
    case ST_R:
    case ST_F: return nl ? ST_R : c == '"' ? ST_Q : c == delim ? ST_F : ST_U;
    case ST_U: return nl ? ST_R : c == delim ? ST_F : ST_U;
    case ST_Q: return c == '"' ? ST_E : ST_Q;
    default: return c == '"' ? ST_Q : nl ? ST_R : c == delim ? ST_F : ST_U; // ST_E
    }
}
__host__ __device__ inline uint32_t vec_get(uint32_t v, uint32_t s) { return (v >> (3 * s)) & 7u; }
// (a then b)
__host__ __device__ inline uint32_t vec_compose(uint32_t a, uint32_t b) {
    uint32_t c = 0;
#pragma unroll
    for (uint32_t s = 0; s < 5; ++s) c |= vec_get(b, vec_get(a, s)) << (3 * s);
    return c;
}

// workgroup inclusive scan of transition vectors (Hillis-Steele in LDS); returns the exclusive prefix of the thread
__device__ inline uint32_t block_scan_vec(uint32_t v, uint32_t *lds, uint32_t *total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < CSV_BLOCK; d <<= 1) {
        uint32_t prev = t >= d ? lds[t - d] : VEC_ID;
        __syncthreads();
        lds[t] = vec_compose(prev, lds[t]);
        __syncthreads();
    }
    const uint32_t excl = t ? lds[t - 1] : VEC_ID;
    *total = lds[CSV_BLOCK - 1];
    __syncthreads();
    return excl;
}

__global__ void __launch_bounds__(CSV_BLOCK) csv_vec_kernel(const uint8_t *bytes, int64_t n, uint8_t delim, uint32_t *tvec, uint32_t *bvec) {
    __shared__ uint32_t lds[CSV_BLOCK];
    const int64_t t = int64_t(blockIdx.x) * CSV_BLOCK + threadIdx.x;
    const int64_t lo = t * CSV_CHUNK, hi = min(lo + CSV_CHUNK, n);
    uint32_t st[5] = {ST_R, ST_F, ST_U, ST_Q, ST_E};
    for (int64_t i = lo; i < hi; ++i) {
        const uint8_t c = bytes[i];
        bool a, b;
#pragma unroll
        for (int s = 0; s < 5; ++s) st[s] = csv_step(st[s], c, delim, &a, &b);
    }
    const uint32_t v = st[0] | (st[1] << 3) | (st[2] << 6) | (st[3] << 9) | (st[4] << 12);
    uint32_t total;
    const uint32_t excl = block_scan_vec(v, lds, &total);
    tvec[t] = excl;
    if (threadIdx.x == 0) bvec[blockIdx.x] = total;
}

// exclusive scan over the workgroup totals (one workgroup, tiles of CSV_BLOCK)
__global__ void __launch_bounds__(CSV_BLOCK) csv_block_scan_kernel(uint32_t *bvec, int64_t nblocks) {
    __shared__ uint32_t lds[CSV_BLOCK];
    uint32_t carry = VEC_ID;
    for (int64_t base = 0; base < nblocks; base += CSV_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? bvec[i] : VEC_ID;
        uint32_t total;
        const uint32_t excl = block_scan_vec(v, lds, &total);
        if (i < nblocks) bvec[i] = vec_compose(carry, excl);
        carry = vec_compose(carry, total);
    }
}

// WRITE = false: per-thread counts of record starts / ends; WRITE = true: positions at the scanned offsets
template <bool WRITE>
__global__ void __launch_bounds__(CSV_BLOCK) csv_mark_kernel(const uint8_t *bytes, int64_t n, uint8_t delim, const uint32_t *tvec, const uint32_t *bpre,
                                                             uint32_t *cnt_start, uint32_t *cnt_end, int64_t *rec_start, int64_t *rec_end) {
    const int64_t t = int64_t(blockIdx.x) * CSV_BLOCK + threadIdx.x;
    const int64_t lo = t * CSV_CHUNK, hi = min(lo + CSV_CHUNK, n);
    if (lo >= n) return; // (the count arrays are zero-initialised)
    uint32_t st = vec_get(vec_compose(bpre[blockIdx.x], tvec[t]), ST_R); // the stream starts at a record start
    uint32_t ns = 0, ne = 0;
    const uint32_t os = WRITE ? cnt_start[t] : 0, oe = WRITE ? cnt_end[t] : 0;
    for (int64_t i = lo; i < hi; ++i) {
        bool a, b;
        st = csv_step(st, bytes[i], delim, &a, &b);
        if (a) {
            if (WRITE) rec_start[os + ns] = i;
            ++ns;
        }
        if (b) {
            if (WRITE) rec_end[oe + ne] = i;
            ++ne;
        }
    }
    if (hi == n && st != ST_R) { // the last record has no terminator
        if (WRITE) rec_end[oe + ne] = n;
        ++ne;
    }
    if (!WRITE) {
        cnt_start[t] = ns;
        cnt_end[t] = ne;
    }
}

struct CsvCols {
    int32_t ncols;
    int32_t dtype[CSV_MAX_COLS];
    uint64_t *words[CSV_MAX_COLS];      // Int64 / Float64 values
    uint8_t *bool_bytes[CSV_MAX_COLS];  // Boolean values (1 byte per row, packed afterwards)
    uint8_t *valid_bytes[CSV_MAX_COLS]; // non-Utf8 columns: 1 = value present
    uint32_t *str_len;                  // [n_utf8][rows + 1] lengths (pass 0) → offsets after the scan
    uint8_t *str_data[CSV_MAX_COLS];    // pass 1
    int32_t str_slot[CSV_MAX_COLS];     // Utf8 column → row of str_len
};
enum { CSV_FLAG_FIELDS = 0, CSV_FLAG_PARSE = 1, CSV_FLAG_NULLS = 2 /* + column */ };

// One thread per record.  PASS 0: split + convert (+ Utf8 lengths); PASS 1: copy Utf8 bytes to their offsets.
template <int PASS>
__global__ void __launch_bounds__(256) csv_fields_kernel(const uint8_t *bytes, const int64_t *rec_start, const int64_t *rec_end, int64_t first_rec, int64_t rows,
                                                         uint8_t delim, CsvCols cc, int *flags, int64_t *bad_row) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t lo = rec_start[first_rec + r], hi = rec_end[first_rec + r];
    int col = 0;
    int64_t i = lo;
    while (true) {
        // ---- one field: [fs, fe) raw, content = the bytes the csv crate would hand out
        const int64_t fs = i;
        bool quoted = false, messy = false; // messy: escapes or text after the closing quote (content is not one raw slice)
        int64_t cs = fs, ce = fs;           // content slice when !messy
        uint32_t clen = 0;                  // content length (always)
        uint32_t st = ST_F;
        uint8_t *dst = nullptr;
        if (PASS == 1 && col < cc.ncols && cc.dtype[col] == NQE_UTF8) dst = cc.str_data[col] + cc.str_len[int64_t(cc.str_slot[col]) * (rows + 1) + r];
        for (; i < hi; ++i) {
            const uint8_t c = bytes[i];
            if (st == ST_F) {
                if (c == '"') { st = ST_Q; quoted = true; cs = ce = i + 1; continue; }
                if (c == delim) break;
                st = ST_U;
                cs = i;
            } else if (st == ST_U) {
                if (c == delim) break;
            } else if (st == ST_Q) {
                if (c == '"') { st = ST_E; continue; }
            } else { // ST_E
                if (c == delim) break;
                if (c == '"') st = ST_Q; // escaped quote: one '"' of content
                else st = ST_U;          // text after the closing quote is appended
                messy = true;
            }
            if (dst) dst[clen] = c;
            ++clen;
            if (!messy) ce = i + 1;
        }
        // ---- convert
        if (col < cc.ncols) {
            const int dt = cc.dtype[col];
            if (dt == NQE_UTF8) {
                if (PASS == 0) cc.str_len[int64_t(cc.str_slot[col]) * (rows + 1) + r] = clen;
            } else if (PASS == 0) {
                bool ok = true, present = clen != 0;
                uint64_t w = 0;
                if (present) {
                    const char *p = reinterpret_cast<const char *>(bytes + cs);
                    const int len = int(ce - cs);
                    if (messy) ok = false;
                    else if (dt == NQE_INT64) { int64_t v; ok = csvp::parse_i64(p, len, &v); w = uint64_t(v); }
                    else if (dt == NQE_FLOAT64) { double v; ok = csvp::parse_f64(p, len, &v); __builtin_memcpy(&w, &v, 8); }
                    else { bool v; ok = csvp::parse_bool(p, len, &v); w = v ? 1 : 0; }
                    if (!ok) {
                        atomicOr(&flags[CSV_FLAG_PARSE], 1);
                        atomicMin((unsigned long long *)bad_row, (unsigned long long)r);
                    }
                } else atomicOr(&flags[CSV_FLAG_NULLS + col], 1);
                if (dt == NQE_BOOLEAN) cc.bool_bytes[col][r] = uint8_t(w);
                else cc.words[col][r] = w;
                cc.valid_bytes[col][r] = present ? 1 : 0;
            }
        }
        (void)quoted;
        ++col;
        if (i >= hi) break;
        ++i; // the delimiter
    }
    if (PASS == 0 && col != cc.ncols) {
        atomicOr(&flags[CSV_FLAG_FIELDS], 1);
        atomicMin((unsigned long long *)bad_row, (unsigned long long)r);
    }
}

// ---------------------------------------------------------------- host: records of a byte range (schema inference)
struct HostRecord {
    std::vector<std::string> fields;
};
// reads up to `limit` records with the same automaton (sequentially)
std::vector<HostRecord> host_records(const uint8_t *b, int64_t n, uint8_t delim, int64_t limit) {
    std::vector<HostRecord> out;
    uint32_t st = ST_R;
    HostRecord cur;
    std::string field;
    auto end_field = [&]() { cur.fields.push_back(field); field.clear(); };
    for (int64_t i = 0; i <= n && int64_t(out.size()) < limit; ++i) {
        if (i == n) {
            if (st != ST_R) { end_field(); out.push_back(cur); }
            break;
        }
        const uint8_t c = b[i];
        bool a, e;
        const uint32_t nx = csv_step(st, c, delim, &a, &e);
        if (e) {
            end_field();
            out.push_back(cur);
            cur = HostRecord();
        } else if (st == ST_R && is_nl(c)) {
            // empty line
        } else if ((st == ST_R || st == ST_F) && c == '"') {
            // opening quote
        } else if (st == ST_Q && c == '"') {
            // closing quote or first of an escaped pair
        } else if (c == delim && st != ST_Q) {
            end_field();
        } else field.push_back(char(c));
        st = nx;
    }
    return out;
}

bool all_digits(const std::string &s, size_t a, size_t b) {
    if (a >= b) return false;
    for (size_t i = a; i < b; ++i)
        if (s[i] < '0' || s[i] > '9') return false;
    return true;
}
enum Inferred { INF_UTF8 = 1, INF_BOOL = 2, INF_F64 = 4, INF_I64 = 8, INF_DATE = 16 };
// arrow-rs 13 infer_field_schema: leading '"' → Utf8; true/false → Boolean; ^-?(\d+\.\d+)$ → Float64; ^-?(\d+)$ → Int64;
// ISO date / datetime → Date32 / Date64; else Utf8
int infer_field(const std::string &s) {
    if (!s.empty() && s[0] == '"') return INF_UTF8;
    bool bv;
    if (csvp::parse_bool(s.data(), int(s.size()), &bv)) return INF_BOOL;
    const size_t b = (!s.empty() && s[0] == '-') ? 1 : 0;
    const size_t dot = s.find('.');
    if (dot != std::string::npos && all_digits(s, b, dot) && all_digits(s, dot + 1, s.size())) return INF_F64;
    if (all_digits(s, b, s.size())) return INF_I64;
    auto shape = [&](const char *pat) {
        if (s.size() != strlen(pat)) return false;
        for (size_t i = 0; i < s.size(); ++i)
            if (pat[i] == 'd' ? !(s[i] >= '0' && s[i] <= '9') : s[i] != pat[i]) return false;
        return true;
    };
    if (shape("dddd-dd-ddTdd:dd:dd") || shape("dddd-dd-dd")) return INF_DATE;
    return INF_UTF8;
}

} // namespace
} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_csv_infer_schema(nqe_ctx *ctx, const void *bytes_host, int64_t nbytes, const nqe_csv_options *opt, int32_t max_columns,
                                int32_t *num_columns, int32_t *dtypes, int32_t *nullable, char *names, int64_t names_capacity, int64_t *names_bytes) {
    NQE_API_BEGIN(ctx)
    if (!opt || nbytes < 0 || (nbytes && !bytes_host) || !num_columns) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    const uint8_t *b = static_cast<const uint8_t *>(bytes_host);
    const int64_t want = (opt->max_read_records < 0 ? INT64_MAX / 2 : opt->max_read_records) + (opt->has_header ? 1 : 0);
    std::vector<HostRecord> recs = host_records(b, nbytes, uint8_t(opt->delimiter), want);
    if (recs.empty()) fail(NQE_ERR_ARROW, "csv: empty file");
    const size_t nc = recs[0].fields.size();
    std::vector<std::string> hdr;
    for (size_t c = 0; c < nc; ++c) hdr.push_back(opt->has_header ? recs[0].fields[c] : "column_" + std::to_string(c + 1));
    std::vector<int> poss(nc, 0), nul(nc, 0);
    for (size_t r = opt->has_header ? 1 : 0; r < recs.size(); ++r) {
        if (recs[r].fields.size() != nc) fail(NQE_ERR_ARROW, "csv: record with a different number of fields");
        for (size_t c = 0; c < nc; ++c) {
            const std::string &s = recs[r].fields[c];
            if (s.empty()) nul[c] = 1;
            else poss[c] |= infer_field(s);
        }
    }
    *num_columns = int32_t(nc);
    std::string joined;
    for (auto &h : hdr) { joined += h; joined.push_back('\0'); }
    if (names_bytes) *names_bytes = int64_t(joined.size());
    if (int32_t(nc) > max_columns || (names && int64_t(joined.size()) > names_capacity)) fail(NQE_ERR_INVALID_ARGUMENT, "csv: output buffers too small");
    for (size_t c = 0; c < nc; ++c) {
        int dt;
        const int p = poss[c];
        if (p == INF_BOOL) dt = NQE_BOOLEAN;
        else if (p == INF_I64) dt = NQE_INT64;
        else if (p == INF_F64 || p == (INF_F64 | INF_I64)) dt = NQE_FLOAT64;
        else if (p == INF_DATE) fail(NQE_ERR_NOT_SUPPORTED, "csv: Date32/Date64 columns are outside the hot path's types");
        else dt = NQE_UTF8; // nothing seen, or mixed
        if (dtypes) dtypes[c] = dt;
        if (nullable) nullable[c] = nul[c];
    }
    if (names && !joined.empty()) std::memcpy(names, joined.data(), joined.size());
    NQE_API_END()
}

nqe_status nqe_csv_read(nqe_ctx *ctx, const void *bytes, int32_t location, int64_t nbytes, const nqe_csv_options *opt, const int32_t *dtypes,
                        int32_t num_columns, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !opt || !out || !dtypes || num_columns <= 0 || num_columns > CSV_MAX_COLS || nbytes < 0 || (nbytes && !bytes))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    for (int c = 0; c < num_columns; ++c)
        if (!(dtypes[c] == NQE_INT64 || dtypes[c] == NQE_FLOAT64 || dtypes[c] == NQE_BOOLEAN || dtypes[c] == NQE_UTF8))
            fail(NQE_ERR_NOT_SUPPORTED, "csv: column type");
    const uint8_t delim = uint8_t(opt->delimiter);
    BufRef dbytes;
    if (location == NQE_DEVICE) dbytes = dev_borrow(ctx, bytes, size_t(nbytes));
    else {
        dbytes = dev_alloc(ctx, size_t(nbytes) + 8);
        if (nbytes) NQE_HIP_CHECK(hipMemcpyAsync(dbytes->ptr, bytes, size_t(nbytes), hipMemcpyHostToDevice, ctx->stream));
    }
    const uint8_t *db = static_cast<const uint8_t *>(dbytes->ptr);
    // ---- 1. record boundaries
    const int64_t nthreads = (nbytes + CSV_CHUNK - 1) / CSV_CHUNK;
    const int64_t nblocks = std::max<int64_t>(1, (nthreads + CSV_BLOCK - 1) / CSV_BLOCK);
    const int64_t tpad = nblocks * CSV_BLOCK;
    BufRef tvec = dev_alloc(ctx, size_t(tpad) * 4), bvec = dev_alloc(ctx, size_t(nblocks) * 4);
    BufRef cs = dev_alloc_zero(ctx, size_t(tpad + 1) * 4), ce = dev_alloc_zero(ctx, size_t(tpad + 1) * 4);
    int64_t nrec = 0;
    BufRef rstart, rend;
    if (nbytes) {
        launch(ctx, "csv_vec", csv_vec_kernel, dim3(unsigned(nblocks)), dim3(CSV_BLOCK), 0, db, nbytes, delim, (uint32_t *)tvec->ptr, (uint32_t *)bvec->ptr);
        launch(ctx, "csv_block_scan", csv_block_scan_kernel, dim3(1), dim3(CSV_BLOCK), 0, (uint32_t *)bvec->ptr, nblocks);
        launch(ctx, "csv_mark_count", csv_mark_kernel<false>, dim3(unsigned(nblocks)), dim3(CSV_BLOCK), 0, db, nbytes, delim, (const uint32_t *)tvec->ptr,
               (const uint32_t *)bvec->ptr, (uint32_t *)cs->ptr, (uint32_t *)ce->ptr, (int64_t *)nullptr, (int64_t *)nullptr);
        exclusive_scan_u32_inplace(ctx, (uint32_t *)cs->ptr, tpad + 1);
        exclusive_scan_u32_inplace(ctx, (uint32_t *)ce->ptr, tpad + 1);
        const uint32_t ns = read_scalar(ctx, (const uint32_t *)cs->ptr + tpad), ne = read_scalar(ctx, (const uint32_t *)ce->ptr + tpad);
        if (ns != ne) fail(NQE_ERR_OTHERS, "csv: record start/end counts differ");
        nrec = ns;
        rstart = dev_alloc(ctx, size_t(nrec) * 8 + 8);
        rend = dev_alloc(ctx, size_t(nrec) * 8 + 8);
        if (nrec)
            launch(ctx, "csv_mark_write", csv_mark_kernel<true>, dim3(unsigned(nblocks)), dim3(CSV_BLOCK), 0, db, nbytes, delim, (const uint32_t *)tvec->ptr,
                   (const uint32_t *)bvec->ptr, (uint32_t *)cs->ptr, (uint32_t *)ce->ptr, (int64_t *)rstart->ptr, (int64_t *)rend->ptr);
    }
    const int64_t first = opt->has_header ? 1 : 0;
    int64_t rows = std::max<int64_t>(0, nrec - first);
    if (opt->batch_size >= 0) rows = std::min(rows, opt->batch_size); // only the first batch is kept (quirk Q1)
    // ---- 2. fields
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = rows;
    CsvCols cc;
    std::memset(&cc, 0, sizeof(cc));
    cc.ncols = num_columns;
    std::vector<BufRef> vbytes{static_cast<size_t>(num_columns)}, bbytes{static_cast<size_t>(num_columns)};
    int n_utf8 = 0;
    for (int c = 0; c < num_columns; ++c) {
        cc.dtype[c] = dtypes[c];
        DevColumn d;
        if (dtypes[c] == NQE_UTF8) {
            cc.str_slot[c] = n_utf8++;
            d.dtype = NQE_UTF8;
            d.length = rows;
        } else {
            d = dtypes[c] == NQE_BOOLEAN ? make_bool_column(ctx, rows, true) : make_word_column(ctx, dtypes[c], rows, true);
            vbytes[size_t(c)] = dev_alloc(ctx, size_t(rows) + 8);
            cc.valid_bytes[c] = (uint8_t *)vbytes[size_t(c)]->ptr;
            if (dtypes[c] == NQE_BOOLEAN) {
                bbytes[size_t(c)] = dev_alloc(ctx, size_t(rows) + 8);
                cc.bool_bytes[c] = (uint8_t *)bbytes[size_t(c)]->ptr;
            } else cc.words[c] = (uint64_t *)d.values->ptr;
        }
        t->cols.push_back(std::move(d));
    }
    BufRef slen = dev_alloc_zero(ctx, size_t(std::max(n_utf8, 1)) * size_t(rows + 1) * 4);
    cc.str_len = (uint32_t *)slen->ptr;
    BufRef flags = dev_alloc_zero(ctx, sizeof(int) * (CSV_FLAG_NULLS + CSV_MAX_COLS) + 8);
    BufRef bad = dev_alloc(ctx, 8);
    NQE_HIP_CHECK(hipMemsetAsync(bad->ptr, 0xff, 8, ctx->stream));
    if (rows) {
        const dim3 grid(unsigned((rows + 255) / 256));
        launch(ctx, "csv_fields", csv_fields_kernel<0>, grid, dim3(256), 0, db, (const int64_t *)rstart->ptr, (const int64_t *)rend->ptr, first, rows, delim, cc,
               (int *)flags->ptr, (int64_t *)bad->ptr);
    }
    std::vector<int> hf(static_cast<size_t>(CSV_FLAG_NULLS + CSV_MAX_COLS), 0);
    NQE_HIP_CHECK(hipMemcpyAsync(hf.data(), flags->ptr, hf.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    const int64_t bad_row = int64_t(read_scalar(ctx, (const uint64_t *)bad->ptr));
    if (hf[CSV_FLAG_FIELDS]) fail(NQE_ERR_ARROW, "csv: record " + std::to_string(bad_row + first + 1) + " has a different number of fields than the schema");
    if (hf[CSV_FLAG_PARSE]) fail(NQE_ERR_ARROW, "csv: error while parsing a value at line " + std::to_string(bad_row + first + 1));
    for (int c = 0; c < num_columns; ++c) {
        DevColumn &d = t->cols[size_t(c)];
        if (dtypes[c] == NQE_UTF8) {
            uint32_t *len = cc.str_len + int64_t(cc.str_slot[c]) * (rows + 1);
            exclusive_scan_u32_inplace(ctx, len, rows + 1);
            const uint32_t total = read_scalar(ctx, (const uint32_t *)len + rows);
            if (total > uint32_t(INT32_MAX)) fail(NQE_ERR_ARROW, "csv: Utf8 column exceeds 2 GiB (i32 offsets)");
            d.data = dev_alloc(ctx, size_t(total) + 8);
            d.data_length = total;
            cc.str_data[c] = (uint8_t *)d.data->ptr;
            d.values = dev_view(slen, size_t(cc.str_slot[c]) * size_t(rows + 1) * 4, size_t(rows + 1) * 4);
            d.null_count = 0;
        } else {
            if (dtypes[c] == NQE_BOOLEAN) pack_bytes_to_bits(ctx, cc.bool_bytes[c], rows, (uint64_t *)d.values->ptr);
            if (hf[size_t(CSV_FLAG_NULLS + c)]) {
                pack_bytes_to_bits(ctx, cc.valid_bytes[c], rows, (uint64_t *)d.validity->ptr);
                d.null_count = -1;
            } else {
                d.validity = nullptr;
                d.null_count = 0;
            }
        }
    }
    if (n_utf8 && rows) {
        const dim3 grid(unsigned((rows + 255) / 256));
        launch(ctx, "csv_fields_copy", csv_fields_kernel<1>, grid, dim3(256), 0, db, (const int64_t *)rstart->ptr, (const int64_t *)rend->ptr, first, rows, delim, cc,
               (int *)flags->ptr, (int64_t *)bad->ptr);
    }
    sync(ctx); // the staging buffers are released here; a host input may be freed by the caller on return
    *out = t.release();
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::csv_block_scan_kernel);
