// arrow_c.hip — Arrow C Data Interface at the boundary: real RecordBatches enter HBM and results leave it as
// `ArrowArray` / `ArrowSchema` pairs with release callbacks — what arrow-rs (`arrow::ffi`), pyarrow (`_export_to_c` /
// `_import_from_c`) and every other Arrow implementation exchange.  Replaces the hand-marshalling a host would otherwise do around
// MemTable::try_create (datasource/memory.rs:21-29) and around the Vec<RecordBatch> an operator returns (plan.rs:18).
// A RecordBatch travels as ONE struct array (format "+s") whose children are the columns.  Supported column formats: "l" Int64,
// "L" UInt64, "g" Float64, "b" Boolean, "u" Utf8 (32-bit offsets) — the types the hot path accepts (selection.rs:69-99).
#include <cstdlib>
#include <cstring>

#include "nqe_internal.hpp"

namespace nqe {
namespace {

int dtype_of_format(const char *f) {
    if (!f) return -1;
    if (!std::strcmp(f, "l")) return NQE_INT64;
    if (!std::strcmp(f, "L")) return NQE_UINT64;
    if (!std::strcmp(f, "g")) return NQE_FLOAT64;
    if (!std::strcmp(f, "b")) return NQE_BOOLEAN;
    if (!std::strcmp(f, "u")) return NQE_UTF8;
    return -1;
}
const char *format_of_dtype(int dt) {
    switch (dt) {
    case NQE_INT64: return "l";
    case NQE_UINT64: return "L";
    case NQE_FLOAT64: return "g";
    case NQE_BOOLEAN: return "b";
    case NQE_UTF8: return "u";
    default: return nullptr;
    }
}

// bits [off, off + n) of an LSB-first bitmap, re-based to bit 0
std::vector<uint8_t> rebase_bits(const uint8_t *src, int64_t off, int64_t n) {
    std::vector<uint8_t> out(size_t((n + 7) / 8), 0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = off + i;
        if ((src[s >> 3] >> (s & 7)) & 1) out[size_t(i >> 3)] |= uint8_t(1u << (i & 7));
    }
    return out;
}

// ---- export: the producer side owns host copies of the buffers until the consumer calls release
struct ExportedArray {
    std::vector<std::vector<uint8_t>> bufs; // owned storage
    std::vector<const void *> buffer_ptrs;
    std::vector<ArrowArray *> children;
    std::vector<std::unique_ptr<ArrowArray>> child_store;
    // releases the children it still owns: also on the export's error path, where the half-built parent is simply destroyed
    // (a child's private data is not reachable from the unique_ptr that holds its struct)
    ~ExportedArray() {
        for (auto &c : child_store)
            if (c && c->release) c->release(c.get());
    }
};
void release_array(ArrowArray *a) {
    if (!a || !a->release) return;
    delete static_cast<ExportedArray *>(a->private_data); // children first (destructor)
    a->release = nullptr;
}
struct ExportedSchema {
    std::string format, name;
    std::vector<ArrowSchema *> children;
    std::vector<std::unique_ptr<ArrowSchema>> child_store;
    ~ExportedSchema() {
        for (auto &c : child_store)
            if (c && c->release) c->release(c.get());
    }
};
void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    delete static_cast<ExportedSchema *>(s->private_data);
    s->release = nullptr;
}

} // namespace
} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_table_import_arrow(nqe_ctx *ctx, struct ArrowArray *array, const struct ArrowSchema *schema, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !array || !schema || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (!array->release) fail(NQE_ERR_INVALID_ARGUMENT, "import: the ArrowArray has already been released");
    if (!schema->format || std::strcmp(schema->format, "+s") != 0) fail(NQE_ERR_NOT_SUPPORTED, "import: expected a struct array (\"+s\": a RecordBatch)");
    if (array->n_children != schema->n_children) fail(NQE_ERR_ARROW, "import: array and schema disagree on the number of columns");
    if (array->n_buffers >= 1 && array->buffers && array->buffers[0] && array->null_count != 0)
        fail(NQE_ERR_NOT_SUPPORTED, "import: a RecordBatch's struct array carries no nulls of its own");
    if (array->dictionary) fail(NQE_ERR_NOT_SUPPORTED, "import: dictionary-encoded data");
    const int64_t rows = array->length, poff = array->offset;
    std::vector<nqe_column> cols(size_t(array->n_children));
    std::vector<std::vector<uint8_t>> keep;      // re-based bitmaps
    std::vector<std::vector<int32_t>> keep_offs; // re-based Utf8 offsets
    for (int64_t i = 0; i < array->n_children; ++i) {
        const ArrowArray *c = array->children[i];
        const ArrowSchema *cs = schema->children[i];
        const int dt = dtype_of_format(cs->format);
        if (dt < 0) fail(NQE_ERR_NOT_SUPPORTED, std::string("import: unsupported column format \"") + (cs->format ? cs->format : "") + "\" (selection.rs:98 unimplemented!())");
        if (c->dictionary) fail(NQE_ERR_NOT_SUPPORTED, "import: dictionary-encoded column");
        const int64_t off = c->offset + poff; // a sliced batch: the parent's offset applies to every child
        if (c->length < poff + rows) fail(NQE_ERR_ARROW, "import: a column is shorter than its batch");
        // buffers as the format prescribes: validity + values (+ bytes for Utf8); a producer may leave the data pointers of a
        // zero-length array NULL, never those of one with rows
        const int64_t need = dt == NQE_UTF8 ? 3 : 2;
        if (c->n_buffers < need || !c->buffers) fail(NQE_ERR_ARROW, "import: a column has fewer buffers than its format prescribes");
        if (rows > 0 && !c->buffers[1]) fail(NQE_ERR_ARROW, "import: a column of a non-empty batch has no values buffer");
        if (rows > 0 && dt == NQE_UTF8 && !c->buffers[2] && static_cast<const int32_t *>(c->buffers[1])[off + rows] != static_cast<const int32_t *>(c->buffers[1])[off])
            fail(NQE_ERR_ARROW, "import: a Utf8 column with bytes has no data buffer");
        nqe_column &d = cols[size_t(i)];
        std::memset(&d, 0, sizeof(d));
        d.dtype = dt;
        d.location = NQE_HOST;
        d.length = rows;
        d.null_count = c->null_count;
        const uint8_t *valid = c->n_buffers > 0 ? static_cast<const uint8_t *>(c->buffers[0]) : nullptr;
        if (valid && c->null_count != 0) {
            if (off & 7) {
                keep.push_back(rebase_bits(valid, off, rows));
                d.validity = keep.back().data();
            } else
                d.validity = valid + (off >> 3);
            // the producer counted the nulls of ITS array: whenever this batch is a different row range of it, recount
            if (off != 0 || c->length != rows) d.null_count = -1;
        } else
            d.null_count = 0;
        if (rows == 0) { // nothing to read: the producer's pointers may be NULL
            static const uint64_t zero_words[2] = {0, 0};
            d.values = zero_words;
            d.validity = nullptr;
            d.null_count = 0;
            d.data = dt == NQE_UTF8 ? reinterpret_cast<const char *>(zero_words) : nullptr;
        } else if (dt == NQE_BOOLEAN) {
            const uint8_t *bits = static_cast<const uint8_t *>(c->buffers[1]);
            if (off & 7) {
                keep.push_back(rebase_bits(bits, off, rows));
                d.values = keep.back().data();
            } else
                d.values = bits + (off >> 3);
        } else if (dt == NQE_UTF8) {
            const int32_t *offs = static_cast<const int32_t *>(c->buffers[1]) + off;
            const char *data = static_cast<const char *>(c->buffers[2]);
            const int32_t first = rows ? offs[0] : 0;
            if (first != 0) { // offsets must start at 0 in the table's own buffers
                keep_offs.emplace_back(size_t(rows) + 1);
                for (int64_t r = 0; r <= rows; ++r) keep_offs.back()[size_t(r)] = offs[r] - first;
                d.values = keep_offs.back().data();
            } else
                d.values = offs;
            d.data = data ? data + first : nullptr;
            d.data_length = rows ? int64_t(offs[rows] - first) : 0;
        } else {
            d.values = static_cast<const uint64_t *>(c->buffers[1]) + off;
        }
    }
    nqe_table *t = nullptr;
    nqe_status st = nqe_table_create(ctx, cols.data(), int32_t(cols.size()), &t); // copies to HBM, synchronises
    if (st != NQE_OK) fail(st, ctx->last_error);
    if (cols.empty()) t->rows = rows;
    // the data has been copied: the consumer is done with the producer's buffers (the interface's "move" semantics)
    array->release(array);
    *out = t;
    NQE_API_END()
}

nqe_status nqe_table_export_arrow(const nqe_table *table, const char *const *names, struct ArrowArray *out_array, struct ArrowSchema *out_schema) {
    NQE_API_BEGIN(table ? table->ctx : nullptr)
    if (!table || !out_array || !out_schema) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    const size_t nc = table->cols.size();
    auto ea = std::make_unique<ExportedArray>();
    auto es = std::make_unique<ExportedSchema>();
    es->format = "+s";
    for (size_t i = 0; i < nc; ++i) {
        const DevColumn &c = table->cols[i];
        const char *fmt = format_of_dtype(c.dtype);
        if (!fmt) fail(NQE_ERR_NOT_SUPPORTED, "export: unsupported column type");
        // ---- host copies of the column's buffers
        auto ca = std::make_unique<ArrowArray>();
        auto cp = std::make_unique<ExportedArray>();
        std::memset(ca.get(), 0, sizeof(ArrowArray));
        nqe_column info;
        nqe_status st = nqe_table_column(table, int32_t(i), &info);
        if (st != NQE_OK) fail(st, table->ctx->last_error);
        const int64_t n = c.length;
        std::vector<uint8_t> valid(c.validity ? size_t((n + 7) / 8) : 0), values, data;
        if (c.dtype == NQE_BOOLEAN) values.resize(size_t((n + 7) / 8));
        else if (c.dtype == NQE_UTF8) {
            values.resize(size_t(n + 1) * 4);
            data.resize(size_t(c.data_length));
        } else values.resize(size_t(n) * 8);
        st = nqe_table_download_column(table, int32_t(i), values.empty() ? nullptr : values.data(), valid.empty() ? nullptr : valid.data(),
                                       data.empty() ? nullptr : data.data());
        if (st != NQE_OK) fail(st, table->ctx->last_error);
        int64_t nulls = 0;
        if (c.validity) {
            for (int64_t r = 0; r < n; ++r) nulls += !((valid[size_t(r >> 3)] >> (r & 7)) & 1);
        }
        cp->bufs.push_back(std::move(valid));
        cp->bufs.push_back(std::move(values));
        if (c.dtype == NQE_UTF8) cp->bufs.push_back(std::move(data));
        for (size_t b = 0; b < cp->bufs.size(); ++b) {
            static const uint64_t empty_word = 0; // a zero-length buffer still needs a non-null pointer for some consumers
            cp->buffer_ptrs.push_back(b == 0 && !c.validity ? nullptr : (cp->bufs[b].empty() ? static_cast<const void *>(&empty_word) : cp->bufs[b].data()));
        }
        ca->length = n;
        ca->null_count = nulls;
        ca->offset = 0;
        ca->n_buffers = int64_t(cp->buffer_ptrs.size());
        ca->buffers = cp->buffer_ptrs.data();
        ca->release = release_array;
        ca->private_data = cp.release();
        ea->children.push_back(ca.get());
        ea->child_store.push_back(std::move(ca));
        // ---- schema child
        auto cs = std::make_unique<ArrowSchema>();
        auto csp = std::make_unique<ExportedSchema>();
        std::memset(cs.get(), 0, sizeof(ArrowSchema));
        csp->format = fmt;
        csp->name = names && names[i] ? names[i] : ("c" + std::to_string(i));
        cs->format = csp->format.c_str();
        cs->name = csp->name.c_str();
        cs->flags = 2; // ARROW_FLAG_NULLABLE
        cs->release = release_schema;
        cs->private_data = csp.release();
        es->children.push_back(cs.get());
        es->child_store.push_back(std::move(cs));
    }
    std::memset(out_array, 0, sizeof(ArrowArray));
    std::memset(out_schema, 0, sizeof(ArrowSchema));
    ea->buffer_ptrs.push_back(nullptr); // a struct array's validity buffer: absent
    out_array->length = table->rows;
    out_array->null_count = 0;
    out_array->n_buffers = 1;
    out_array->buffers = ea->buffer_ptrs.data();
    out_array->n_children = int64_t(nc);
    out_array->children = ea->children.data();
    out_array->release = release_array;
    out_array->private_data = ea.release();
    out_schema->format = es->format.c_str();
    es->name = "";
    out_schema->name = es->name.c_str();
    out_schema->n_children = int64_t(nc);
    out_schema->children = es->children.data();
    out_schema->release = release_schema;
    out_schema->private_data = es.release();
    NQE_API_END()
}

} // extern "C"
