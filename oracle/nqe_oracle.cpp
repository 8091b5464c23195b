// nqe_oracle.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A single-threaded C++ restatement of the reference's physical operators
// (Veeupup/naive-query-engine, Rust, src/physical_plan/**) and of the arrow-rs 13
// compute kernels they call.  It exists to CHECK the HIP path; nothing in the
// product (naive_query_engine_amd/, include/) may link, import or call it.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// Parity pinning: validated against every asserting reference test on this path
// (selection.rs:166-172, projection.rs:113-118, sql/planner.rs:655-679,
// planner/mod.rs:261-270, scan.rs:63-74) and the README known-answer tables
// (README.md:69-112) in tests/test_oracle_golden.py.  arrow-rs 13.0.0 and
// twox-hash 1.6.3 sources are NOT in /root/reference (Cargo.lock:15-18, :582-585);
// their published semantics are restated here: wrapping integer add/sub/mul,
// DivideByZero for any valid zero divisor (ints and floats), truncated `%`, IEEE float
// compares, Kleene and/or, validity = AND of operand validities; XXH64 seed 0 over the
// 8 native-endian key bytes (checked against the python `xxhash` package in the tests).
// Operators with no asserting reference test (every aggregate, hash join, all binary
// operators except + and >) are pinned only by the README tables: "parity unpinned"
// beyond those vectors, see DESIGN.md.
//
// Structure deliberately follows the reference, including its costs (it is also the
// `cpu_baseline` "port"): literals are materialised as full columns, every column is
// compacted row-at-a-time through a builder, group-by builds HashMap<key, Vec<row>>
// and calls a virtual `update(batch, idx)` per row per aggregate, the join hashes keys
// with XxHash64 into HashMap<u64, Vec<row>> chains and gathers with `take`.

#include <cmath>
#include <cstdint>
#include <cerrno>
#include <cctype>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <cfloat>
#include <unordered_map>
#include <vector>

#include "../include/nqe.h"

namespace {

// ------------------------------------------------------------------ errors
// enum ErrorCode (src/error.rs:13-40) → nqe_status numeric codes.
struct OracleError {
    int code;
    std::string msg;
};
thread_local std::string g_last_error;

[[noreturn]] void fail(int code, const std::string &msg) { throw OracleError{code, msg}; }

// ------------------------------------------------------------------ arrays
// One Arrow array.  8-byte types live in `v` as raw 64-bit words.
struct Arr {
    int dtype = NQE_NULLTYPE;
    int64_t len = 0;
    std::vector<uint64_t> v;      // Int64/UInt64/Float64 raw words
    std::vector<uint8_t> bits;    // Boolean values, LSB-first
    std::vector<uint8_t> valid;   // validity bitmap, empty = no null buffer
    std::vector<int32_t> offs;    // Utf8 offsets (len+1)
    std::string data;             // Utf8 bytes

    bool is_valid(int64_t i) const { return valid.empty() || ((valid[i >> 3] >> (i & 7)) & 1); }
    bool is_null(int64_t i) const { return !is_valid(i); }
    bool bit(int64_t i) const { return (bits[i >> 3] >> (i & 7)) & 1; }
    int64_t null_count() const {
        if (valid.empty()) return 0;
        int64_t n = 0;
        for (int64_t i = 0; i < len; ++i) n += !is_valid(i);
        return n;
    }
    std::string str(int64_t i) const { return data.substr(offs[i], offs[i + 1] - offs[i]); }
};
using ArrRef = std::shared_ptr<const Arr>;

struct Batch {
    std::vector<ArrRef> cols;
    int64_t rows = 0;
};
using Batches = std::vector<Batch>;

inline void set_bit(std::vector<uint8_t> &b, int64_t i) { b[i >> 3] |= uint8_t(1u << (i & 7)); }
inline size_t bm_bytes(int64_t n) { return size_t((n + 7) / 8); }

bool is_word_type(int dt) { return dt == NQE_INT64 || dt == NQE_UINT64 || dt == NQE_FLOAT64; }

inline double as_f64(uint64_t w) { double d; std::memcpy(&d, &w, 8); return d; }
inline uint64_t f64_bits(double d) { uint64_t w; std::memcpy(&w, &d, 8); return w; }

// arrow `PrimitiveBuilder` / `BooleanBuilder` / `StringBuilder` with append_option
// (selection.rs:37-49).  The null bitmap is materialised on the first null, so an
// all-valid result has no null buffer (as `finish()` does in arrow-rs 13).
struct Builder {
    std::shared_ptr<Arr> a;
    explicit Builder(int dtype) : a(std::make_shared<Arr>()) {
        a->dtype = dtype;
        if (dtype == NQE_UTF8) a->offs.push_back(0);
    }
    void grow_valid(bool valid_bit) {
        int64_t i = a->len;
        if (!valid_bit && a->valid.empty()) {
            // materialise: all previous rows valid
            a->valid.assign(bm_bytes(i + 1), 0);
            for (int64_t k = 0; k < i; ++k) set_bit(a->valid, k);
        }
        if (!a->valid.empty()) {
            if (a->valid.size() < bm_bytes(i + 1)) a->valid.push_back(0);
            if (valid_bit) set_bit(a->valid, i);
        }
    }
    void append_word(bool valid_bit, uint64_t w) {
        grow_valid(valid_bit);
        a->v.push_back(valid_bit ? w : 0);
        a->len++;
    }
    void append_bool(bool valid_bit, bool b) {
        grow_valid(valid_bit);
        if (a->bits.size() < bm_bytes(a->len + 1)) a->bits.push_back(0);
        if (valid_bit && b) set_bit(a->bits, a->len);
        a->len++;
    }
    void append_str(bool valid_bit, const Arr &src, int64_t i) {
        grow_valid(valid_bit);
        if (valid_bit) a->data.append(src.data, src.offs[i], src.offs[i + 1] - src.offs[i]);
        a->offs.push_back(int32_t(a->data.size()));
        a->len++;
    }
    // append_option(array.iter() item i)
    void append_from(const Arr &src, int64_t i) {
        bool ok = src.is_valid(i);
        switch (src.dtype) {
        case NQE_BOOLEAN: append_bool(ok, ok && src.bit(i)); break;
        case NQE_UTF8: append_str(ok, src, i); break;
        default: append_word(ok, src.v[i]); break;
        }
    }
    void append_null_like(const Arr &src) {
        switch (src.dtype) {
        case NQE_BOOLEAN: append_bool(false, false); break;
        case NQE_UTF8: append_str(false, src, 0); break;
        default: append_word(false, 0); break;
        }
    }
    ArrRef finish() { return a; }
};

// ------------------------------------------------------------------ ScalarValue
// logical_plan/expression.rs:174-187
struct Scalar {
    int dtype = NQE_NULLTYPE;
    bool is_null = true;
    uint64_t word = 0; // i64/u64/f64 bits, or 0/1 for Boolean
    std::string str;   // Utf8
};

ArrRef new_null_array(int dtype, int64_t n) {
    auto a = std::make_shared<Arr>();
    a->dtype = dtype;
    a->len = n;
    a->valid.assign(bm_bytes(n), 0); // n == 0: no buffer, nothing to be null
    if (is_word_type(dtype)) a->v.assign(size_t(n), 0);
    if (dtype == NQE_BOOLEAN) a->bits.assign(bm_bytes(n), 0);
    if (dtype == NQE_UTF8) a->offs.assign(size_t(n) + 1, 0);
    return a;
}

// ScalarValue::into_array (logical_plan/expression.rs:210-222): the literal becomes a
// full n-row column (`from_value`), None becomes an all-null array.
ArrRef scalar_into_array(const Scalar &s, int64_t n) {
    if (s.dtype == NQE_NULLTYPE) return new_null_array(NQE_NULLTYPE, n);
    if (s.is_null) return new_null_array(s.dtype, n);
    auto a = std::make_shared<Arr>();
    a->dtype = s.dtype;
    a->len = n;
    if (s.dtype == NQE_BOOLEAN) {
        a->bits.assign(bm_bytes(n), s.word ? 0xFF : 0x00);
    } else if (is_word_type(s.dtype)) {
        a->v.assign(size_t(n), s.word);
    } else if (s.dtype == NQE_UTF8) { // StringArray::from_iter_values(repeat(value).take(size))
        a->offs.push_back(0);
        for (int64_t i = 0; i < n; ++i) {
            a->data += s.str;
            a->offs.push_back(int32_t(a->data.size()));
        }
    } else {
        fail(NQE_ERR_NOT_SUPPORTED, "literal type");
    }
    return a;
}

// ColumnValue (src/datatype.rs:15-35)
struct ColumnValue {
    bool is_const = false;
    ArrRef array;
    Scalar scalar;
    int64_t n = 0;
    int data_type() const { return is_const ? scalar.dtype : array->dtype; }
    ArrRef into_array() const { return is_const ? scalar_into_array(scalar, n) : array; }
};

// ------------------------------------------------------------------ PhysicalExpr
// expression/mod.rs:25-29
struct PhysicalExpr {
    virtual ~PhysicalExpr() = default;
    virtual ColumnValue evaluate(const Batch &input) const = 0;
    virtual bool is_column() const { return false; }
};
using ExprRef = std::shared_ptr<PhysicalExpr>;

// expression/column.rs:39-57 (index form; names are resolved by the host mirror)
struct ColumnExpr : PhysicalExpr {
    int idx;
    explicit ColumnExpr(int i) : idx(i) {}
    ColumnValue evaluate(const Batch &input) const override {
        if (idx < 0 || size_t(idx) >= input.cols.size())
            fail(NQE_ERR_LOGICAL, "ColumnExpr index out of range");
        ColumnValue cv;
        cv.array = input.cols[size_t(idx)];
        cv.n = input.rows;
        return cv;
    }
    bool is_column() const override { return true; }
};

// expression/literal.rs:32-34
struct LiteralExpr : PhysicalExpr {
    Scalar literal;
    explicit LiteralExpr(Scalar s) : literal(s) {}
    ColumnValue evaluate(const Batch &input) const override {
        ColumnValue cv;
        cv.is_const = true;
        cv.scalar = literal;
        cv.n = input.rows;
        return cv;
    }
};

// combine_option_bitmap: validity = AND of both, absent when neither has one
std::vector<uint8_t> combine_validity(const Arr &l, const Arr &r, int64_t n) {
    if (l.valid.empty() && r.valid.empty()) return {};
    std::vector<uint8_t> out(bm_bytes(n), 0);
    for (int64_t i = 0; i < n; ++i)
        if (l.is_valid(i) && r.is_valid(i)) set_bit(out, i);
    return out;
}

template <typename T> inline T word_as(uint64_t w) { T t; std::memcpy(&t, &w, 8); return t; }

template <typename T> bool cmp_apply(int op, T a, T b) {
    switch (op) {
    case NQE_OP_EQ: return a == b;
    case NQE_OP_NOT_EQ: return a != b;
    case NQE_OP_LT: return a < b;
    case NQE_OP_LT_EQ: return a <= b;
    case NQE_OP_GT: return a > b;
    default: return a >= b;
    }
}

// eq_dyn / neq_dyn / lt_dyn / lt_eq_dyn / gt_dyn / gt_eq_dyn (binary.rs:127-132)
ArrRef compare_dyn(int op, const Arr &l, const Arr &r) {
    if (l.len != r.len) fail(NQE_ERR_ARROW, "Cannot perform comparison operation on arrays of different length");
    if (l.dtype == NQE_NULLTYPE) fail(NQE_ERR_ARROW, "comparison on Null arrays is not supported");
    int64_t n = l.len;
    auto out = std::make_shared<Arr>();
    out->dtype = NQE_BOOLEAN;
    out->len = n;
    out->bits.assign(bm_bytes(n), 0);
    out->valid = combine_validity(l, r, n);
    for (int64_t i = 0; i < n; ++i) {
        bool b;
        switch (l.dtype) {
        case NQE_INT64: b = cmp_apply<int64_t>(op, int64_t(l.v[i]), int64_t(r.v[i])); break;
        case NQE_UINT64: b = cmp_apply<uint64_t>(op, l.v[i], r.v[i]); break;
        case NQE_FLOAT64: b = cmp_apply<double>(op, as_f64(l.v[i]), as_f64(r.v[i])); break;
        case NQE_BOOLEAN: b = cmp_apply<int>(op, int(l.bit(i)), int(r.bit(i))); break;
        case NQE_UTF8: b = cmp_apply<std::string>(op, l.str(i), r.str(i)); break;
        default: fail(NQE_ERR_ARROW, "comparison: unsupported type");
        }
        if (b) set_bit(out->bits, i);
    }
    return out;
}

// and_kleene / or_kleene (binary.rs:133-148)
ArrRef kleene(int op, const Arr &l, const Arr &r) {
    if (l.len != r.len) fail(NQE_ERR_ARROW, "Cannot perform bitwise operation on arrays of different length");
    int64_t n = l.len;
    auto out = std::make_shared<Arr>();
    out->dtype = NQE_BOOLEAN;
    out->len = n;
    out->bits.assign(bm_bytes(n), 0);
    bool any_valid_buf = !l.valid.empty() || !r.valid.empty();
    if (any_valid_buf) out->valid.assign(bm_bytes(n), 0);
    for (int64_t i = 0; i < n; ++i) {
        bool lv = l.is_valid(i), rv = r.is_valid(i);
        bool lb = lv && l.bit(i), rb = rv && r.bit(i);
        bool val, ok;
        if (op == NQE_OP_AND) {
            // false AND x = false; true AND null = null
            bool lf = lv && !lb, rf = rv && !rb;
            ok = (lv && rv) || lf || rf;
            val = ok && lb && rb;
        } else {
            // true OR x = true; false OR null = null
            bool lt = lv && lb, rt = rv && rb;
            ok = (lv && rv) || lt || rt;
            val = ok && (lb || rb);
        }
        if (val) set_bit(out->bits, i);
        if (any_valid_buf && ok) set_bit(out->valid, i);
    }
    return out;
}

// kernels::arithmetic::{add, subtract, multiply, divide, modulus} (binary.rs:149-153)
// arrow-rs 13 math_op: plain wrapping `a op b` on every slot; math_checked_divide_op:
// DivideByZero when a VALID slot's divisor is zero (floats too: `right.is_zero()`),
// null slots produce the default value.  i64::MIN / -1 and i64::MIN % -1 panic in Rust
// (overflow check is unconditional for division) → reported as NQE_ERR_ARROW here.
ArrRef arithmetic(int op, const Arr &l, const Arr &r) {
    if (l.len != r.len) fail(NQE_ERR_ARROW, "Cannot perform math operation on arrays of different length");
    int64_t n = l.len;
    auto out = std::make_shared<Arr>();
    out->dtype = l.dtype;
    out->len = n;
    out->v.assign(size_t(n), 0);
    out->valid = combine_validity(l, r, n);
    const bool checked = (op == NQE_OP_DIVIDE || op == NQE_OP_MODULOS);
    for (int64_t i = 0; i < n; ++i) {
        if (checked && !out->is_valid(i)) { out->v[i] = 0; continue; }
        uint64_t a = l.v[i], b = r.v[i], o = 0;
        switch (l.dtype) {
        case NQE_INT64: {
            int64_t x = int64_t(a), y = int64_t(b);
            switch (op) {
            case NQE_OP_PLUS: o = a + b; break;
            case NQE_OP_MINUS: o = a - b; break;
            case NQE_OP_MULTIPLY: o = a * b; break;
            default:
                if (y == 0) fail(NQE_ERR_ARROW, "Divide by zero");
                if (x == std::numeric_limits<int64_t>::min() && y == -1)
                    fail(NQE_ERR_ARROW, "attempt to divide with overflow");
                o = uint64_t(op == NQE_OP_DIVIDE ? x / y : x % y);
            }
            break;
        }
        case NQE_UINT64:
            switch (op) {
            case NQE_OP_PLUS: o = a + b; break;
            case NQE_OP_MINUS: o = a - b; break;
            case NQE_OP_MULTIPLY: o = a * b; break;
            default:
                if (b == 0) fail(NQE_ERR_ARROW, "Divide by zero");
                o = (op == NQE_OP_DIVIDE ? a / b : a % b);
            }
            break;
        case NQE_FLOAT64: {
            double x = as_f64(a), y = as_f64(b), z;
            switch (op) {
            case NQE_OP_PLUS: z = x + y; break;
            case NQE_OP_MINUS: z = x - y; break;
            case NQE_OP_MULTIPLY: z = x * y; break;
            default:
                if (y == 0.0) fail(NQE_ERR_ARROW, "Divide by zero");
                z = (op == NQE_OP_DIVIDE ? x / y : std::fmod(x, y)); // Rust f64 % = fmod
            }
            o = f64_bits(z);
            break;
        }
        default:
            fail(NQE_ERR_NOT_SUPPORTED, "arithmetic on this type is unimplemented!() (binary.rs:85)");
        }
        out->v[i] = o;
    }
    return out;
}

// expression/binary.rs:108-155
struct BinaryExpr : PhysicalExpr {
    ExprRef left, right;
    int op;
    BinaryExpr(ExprRef l, int o, ExprRef r) : left(std::move(l)), right(std::move(r)), op(o) {}
    ColumnValue evaluate(const Batch &input) const override {
        ColumnValue lv = left->evaluate(input);
        ColumnValue rv = right->evaluate(input);
        int ldt = lv.data_type(), rdt = rv.data_type();
        if (ldt != rdt)
            fail(NQE_ERR_INTERVAL, "Cannot evaluate binary expression with types " + std::to_string(ldt) +
                                       " and " + std::to_string(rdt));
        // TODO in the reference (binary.rs:121): scalars are materialised
        ArrRef la = lv.into_array();
        ArrRef ra = rv.into_array();
        ColumnValue out;
        out.n = input.rows;
        switch (op) {
        case NQE_OP_EQ: case NQE_OP_NOT_EQ: case NQE_OP_LT: case NQE_OP_LT_EQ: case NQE_OP_GT: case NQE_OP_GT_EQ:
            out.array = compare_dyn(op, *la, *ra);
            break;
        case NQE_OP_AND: case NQE_OP_OR:
            if (ldt != NQE_BOOLEAN)
                fail(NQE_ERR_INTERVAL, "Cannot evaluate binary expression And/Or with non-Boolean types");
            out.array = kleene(op, *la, *ra);
            break;
        case NQE_OP_PLUS: case NQE_OP_MINUS: case NQE_OP_MULTIPLY: case NQE_OP_DIVIDE: case NQE_OP_MODULOS:
            if (!is_word_type(ldt))
                fail(NQE_ERR_NOT_SUPPORTED, "arithmetic on this type is unimplemented!() (binary.rs:85)");
            out.array = arithmetic(op, *la, *ra);
            break;
        default:
            fail(NQE_ERR_INVALID_ARGUMENT, "unknown operator");
        }
        return out;
    }
};

ExprRef build_expr(const nqe_expr_node *nodes, int n) {
    std::vector<ExprRef> st;
    for (int i = 0; i < n; ++i) {
        const nqe_expr_node &nd = nodes[i];
        if (nd.kind == NQE_EXPR_COLUMN) {
            st.push_back(std::make_shared<ColumnExpr>(nd.column));
        } else if (nd.kind == NQE_EXPR_LITERAL) {
            Scalar s;
            s.dtype = nd.dtype;
            s.is_null = nd.is_null != 0;
            if (nd.dtype == NQE_UTF8) {
                if (!s.is_null && nd.value.utf8) s.str.assign(nd.value.utf8, size_t(nd.utf8_length));
            } else s.word = nd.dtype == NQE_BOOLEAN ? uint64_t(nd.value.boolean != 0) : nd.value.u64;
            st.push_back(std::make_shared<LiteralExpr>(s));
        } else if (nd.kind == NQE_EXPR_BINARY) {
            if (st.size() < 2) fail(NQE_ERR_INVALID_ARGUMENT, "malformed expression");
            ExprRef r = st.back(); st.pop_back();
            ExprRef l = st.back(); st.pop_back();
            st.push_back(std::make_shared<BinaryExpr>(l, nd.op, r));
        } else {
            fail(NQE_ERR_INVALID_ARGUMENT, "unknown expression node kind");
        }
    }
    if (st.size() != 1) fail(NQE_ERR_INVALID_ARGUMENT, "malformed expression");
    return st[0];
}

// ------------------------------------------------------------------ concat / take
// arrow::compute::concat per column (hash_join.rs:258-273).  `proto` gives the schema
// (dtypes) for the empty case (RecordBatch::new_empty).
Batch concat_batches(const std::vector<int> &schema, const Batches &batches) {
    Batch out;
    if (batches.empty()) {
        for (int dt : schema) out.cols.push_back(Builder(dt).finish());
        return out;
    }
    size_t nc = batches[0].cols.size();
    for (size_t c = 0; c < nc; ++c) {
        Builder b(batches[0].cols[c]->dtype);
        for (const Batch &bt : batches)
            for (int64_t i = 0; i < bt.rows; ++i) b.append_from(*bt.cols[c], i);
        out.cols.push_back(b.finish());
    }
    for (const Batch &bt : batches) out.rows += bt.rows;
    return out;
}

std::vector<int> schema_of(const Batch &b) {
    std::vector<int> s;
    for (auto &c : b.cols) s.push_back(c->dtype);
    return s;
}

// compute::take(array, &Int64Array, None) (hash_join.rs:239,245)
ArrRef take(const Arr &src, const std::vector<int64_t> &idx) {
    Builder b(src.dtype);
    for (int64_t i : idx) {
        if (i < 0 || i >= src.len) fail(NQE_ERR_ARROW, "take index out of bounds");
        b.append_from(src, i);
    }
    return b.finish();
}

// ------------------------------------------------------------------ PhysicalPlan
// physical_plan/plan.rs:14-21
struct PhysicalPlan {
    virtual ~PhysicalPlan() = default;
    virtual Batches execute() = 0;
    virtual std::vector<int> schema() const = 0; // dtypes only (names live in the host mirror)
};
using PlanRef = std::shared_ptr<PhysicalPlan>;

// scan.rs:34-36 + MemTable::scan (datasource/memory.rs:31-41)
struct ScanPlan : PhysicalPlan {
    Batches table;
    std::vector<int> dtypes;
    bool has_projection = false;
    std::vector<int> projection;
    Batches execute() override {
        if (!has_projection) return table; // Arc clones
        Batches out;
        for (const Batch &b : table) {
            Batch p;
            p.rows = b.rows;
            for (int i : projection) {
                if (i < 0 || size_t(i) >= b.cols.size()) fail(NQE_ERR_ARROW, "project index out of bounds");
                p.cols.push_back(b.cols[size_t(i)]);
            }
            out.push_back(p);
        }
        return out;
    }
    std::vector<int> schema() const override {
        if (!has_projection) return dtypes;
        std::vector<int> s;
        for (int i : projection) s.push_back(dtypes[size_t(i)]);
        return s;
    }
};

// selection.rs:58-107
struct SelectionPlan : PhysicalPlan {
    PlanRef input;
    ExprRef expr;
    Batches execute() override {
        Batches in = input->execute();
        if (in.empty()) fail(NQE_ERR_NOT_SUPPORTED, "index out of bounds: input[0] (selection.rs:60 panics)");
        ArrRef predicate = expr->evaluate(in[0]).into_array(); // batch 0 only (quirk Q3)
        if (predicate->dtype != NQE_BOOLEAN)
            fail(NQE_ERR_NOT_SUPPORTED, "predicate is not a BooleanArray (selection.rs:61 unwrap panics)");
        Batches out;
        for (const Batch &batch : in) {
            Batch ob;
            for (const ArrRef &col : batch.cols) {
                switch (col->dtype) {
                case NQE_BOOLEAN: case NQE_UINT64: case NQE_INT64: case NQE_FLOAT64: case NQE_UTF8: break;
                default: fail(NQE_ERR_NOT_SUPPORTED, "unimplemented!() column type in selection (selection.rs:98)");
                }
                // build_array_by_predicate! (selection.rs:34-51): zip truncates to the shorter side
                Builder b(col->dtype);
                int64_t n = std::min(predicate->len, col->len);
                for (int64_t i = 0; i < n; ++i) {
                    if (predicate->is_valid(i)) {
                        if (predicate->bit(i)) b.append_from(*col, i);
                    } else {
                        b.append_null_like(*col); // quirk Q4: NULL predicate emits a NULL row
                    }
                }
                ob.cols.push_back(b.finish());
            }
            ob.rows = ob.cols.empty() ? 0 : ob.cols[0]->len;
            out.push_back(ob);
        }
        return out;
    }
    std::vector<int> schema() const override { return input->schema(); }
};

// projection.rs:43-70
struct ProjectionPlan : PhysicalPlan {
    PlanRef input;
    std::vector<ExprRef> exprs;
    bool empty_schema = false; // projection.rs:47-48: pass-through above an aggregate
    Batches execute() override {
        Batches in = input->execute();
        if (empty_schema) return in;
        Batches out;
        for (const Batch &batch : in) {
            Batch ob;
            ob.rows = batch.rows;
            for (const ExprRef &e : exprs) ob.cols.push_back(e->evaluate(batch).into_array());
            for (auto &c : ob.cols)
                if (c->len != ob.rows) fail(NQE_ERR_ARROW, "all columns in a record batch must have the same length");
            out.push_back(ob);
        }
        return out;
    }
    std::vector<int> schema() const override { return {}; }
};

// limit.rs:32-49 / offset.rs:30-51 (RecordBatch::slice restated as a copy)
Batch slice_batch(const Batch &b, int64_t off, int64_t len) {
    Batch o;
    o.rows = len;
    for (auto &c : b.cols) {
        Builder bl(c->dtype);
        for (int64_t i = off; i < off + len; ++i) bl.append_from(*c, i);
        o.cols.push_back(bl.finish());
    }
    return o;
}
struct LimitPlan : PhysicalPlan {
    PlanRef input;
    int64_t n = 0;
    Batches execute() override {
        Batches in = input->execute(), ret;
        int64_t k = n;
        for (const Batch &b : in) {
            if (k == 0) break;
            if (b.rows <= k) { ret.push_back(b); k -= b.rows; }
            else { ret.push_back(slice_batch(b, 0, k)); k = 0; }
        }
        return ret;
    }
    std::vector<int> schema() const override { return input->schema(); }
};
struct OffsetPlan : PhysicalPlan {
    PlanRef input;
    int64_t n = 0;
    Batches execute() override {
        Batches in = input->execute(), ret;
        int64_t k = n;
        for (const Batch &b : in) {
            if (k == 0) { ret.push_back(b); continue; }
            if (k >= b.rows) { k -= b.rows; continue; }
            ret.push_back(slice_batch(b, k, b.rows - k));
            k = 0;
        }
        return ret;
    }
    std::vector<int> schema() const override { return input->schema(); }
};

// ------------------------------------------------------------------ aggregates
// trait AggregateOperator (aggregate/mod.rs:225-235)
struct AggregateOperator {
    int col;
    explicit AggregateOperator(int c) : col(c) {}
    virtual ~AggregateOperator() = default;
    virtual int out_dtype() const { return NQE_FLOAT64; }
    virtual void update_batch(const Batch &data) = 0;
    virtual void update(const Batch &data, int64_t idx) = 0;
    virtual Scalar evaluate() const = 0;
    virtual void clear_state() = 0;

  protected:
    // self.col_expr.evaluate(data)?.into_array() — done per row per op in the reference
    ArrRef column(const Batch &data) const { return ColumnExpr(col).evaluate(data).into_array(); }
    static double val_as_f64(const Arr &c, int64_t i) {
        switch (c.dtype) {
        case NQE_INT64: return double(int64_t(c.v[i]));
        case NQE_UINT64: return double(c.v[i]);
        default: return as_f64(c.v[i]);
        }
    }
    static void check_batch_type(const Arr &c, const char *fn) {
        if (!is_word_type(c.dtype)) fail(NQE_ERR_NOT_SUPPORTED, std::string(fn) + " func for this type is not supported");
    }
    static void check_row_type(const Arr &c) {
        if (!is_word_type(c.dtype)) fail(NQE_ERR_NOT_SUPPORTED, "unimplemented!() aggregate input type (sum.rs:109)");
    }
    static Scalar f64_scalar(double d) { Scalar s; s.dtype = NQE_FLOAT64; s.is_null = false; s.word = f64_bits(d); return s; }
};

// sum.rs:27-121
struct Sum : AggregateOperator {
    double sum = 0.0;
    using AggregateOperator::AggregateOperator;
    void update_batch(const Batch &data) override {
        ArrRef c = column(data);
        check_batch_type(*c, "Sum");
        for (int64_t i = 0; i < c->len; ++i)
            if (c->is_valid(i)) sum += val_as_f64(*c, i);
    }
    void update(const Batch &data, int64_t idx) override {
        ArrRef c = column(data);
        check_row_type(*c);
        if (!c->is_null(idx)) sum += val_as_f64(*c, idx);
    }
    Scalar evaluate() const override { return f64_scalar(sum); }
    void clear_state() override { sum = 0.0; }
};

// avg.rs:27-128 (cnt is u32)
struct Avg : AggregateOperator {
    double sum = 0.0;
    uint32_t cnt = 0;
    using AggregateOperator::AggregateOperator;
    void update_batch(const Batch &data) override {
        ArrRef c = column(data);
        check_batch_type(*c, "Avg");
        for (int64_t i = 0; i < c->len; ++i)
            if (c->is_valid(i)) { sum += val_as_f64(*c, i); cnt += 1; }
    }
    void update(const Batch &data, int64_t idx) override {
        ArrRef c = column(data);
        check_row_type(*c);
        if (!c->is_null(idx)) { sum += val_as_f64(*c, idx); cnt += 1; }
    }
    Scalar evaluate() const override { return f64_scalar(sum / double(cnt)); }
    void clear_state() override { sum = 0.0; cnt = 0; }
};

// count.rs:22-82
struct Count : AggregateOperator {
    uint64_t cnt = 0;
    using AggregateOperator::AggregateOperator;
    int out_dtype() const override { return NQE_UINT64; }
    void update_batch(const Batch &data) override {
        ArrRef c = column(data);
        cnt += uint64_t(c->len - c->null_count());
    }
    void update(const Batch &data, int64_t idx) override {
        ArrRef c = column(data);
        if (!c->is_null(idx)) cnt += 1;
    }
    Scalar evaluate() const override { Scalar s; s.dtype = NQE_UINT64; s.is_null = false; s.word = cnt; return s; }
    void clear_state() override { cnt = 0; }
};

// OrderedFloat<f64> ordering (ordered-float 3.0.0): NaN is greater than everything and
// equal to itself; otherwise partial_cmp.
inline bool of_gt(double a, double b) {
    if (std::isnan(a)) return !std::isnan(b);
    if (std::isnan(b)) return false;
    return a > b;
}
inline bool of_lt(double a, double b) {
    if (std::isnan(b)) return !std::isnan(a);
    if (std::isnan(a)) return false;
    return a < b;
}

// max.rs:28-131 / min.rs (identical modulo `<` and f64::MAX)
struct MinMax : AggregateOperator {
    bool is_max;
    double val;
    MinMax(int c, bool mx) : AggregateOperator(c), is_max(mx) { clear_state(); }
    void step(double x) {
        if (is_max ? of_gt(x, val) : of_lt(x, val)) val = x;
    }
    void update_batch(const Batch &data) override {
        ArrRef c = column(data);
        check_batch_type(*c, is_max ? "Max" : "min");
        for (int64_t i = 0; i < c->len; ++i)
            if (c->is_valid(i)) step(val_as_f64(*c, i));
    }
    void update(const Batch &data, int64_t idx) override {
        ArrRef c = column(data);
        check_row_type(*c);
        if (!c->is_null(idx)) step(val_as_f64(*c, idx));
    }
    Scalar evaluate() const override { return f64_scalar(val); }
    void clear_state() override {
        // f64::MIN (= -f64::MAX) / f64::MAX, not ±infinity (quirk Q10)
        val = is_max ? -std::numeric_limits<double>::max() : std::numeric_limits<double>::max();
    }
};

std::unique_ptr<AggregateOperator> make_agg(const nqe_aggregate &a) {
    switch (a.func) {
    case NQE_AGG_COUNT: return std::make_unique<Count>(a.column);
    case NQE_AGG_SUM: return std::make_unique<Sum>(a.column);
    case NQE_AGG_AVG: return std::make_unique<Avg>(a.column);
    case NQE_AGG_MIN: return std::make_unique<MinMax>(a.column, false);
    case NQE_AGG_MAX: return std::make_unique<MinMax>(a.column, true);
    default: fail(NQE_ERR_NO_MATCH_FUNCTION, "unknown aggregate function");
    }
}

// aggregate/mod.rs:113-222
struct AggregatePlan : PhysicalPlan {
    PlanRef input;
    std::vector<ExprRef> group_expr;
    std::vector<std::unique_ptr<AggregateOperator>> aggr_ops; // state survives execute() (quirk Q9)

    std::vector<int> out_schema() const {
        std::vector<int> s;
        for (auto &op : aggr_ops) s.push_back(op->out_dtype());
        return s;
    }
    Batch evaluate_row() const {
        Batch b;
        b.rows = 1;
        for (auto &op : aggr_ops) b.cols.push_back(scalar_into_array(op->evaluate(), 1));
        return b;
    }
    template <typename K> Batches group_by(const Arr &group_val, const Batch &single_batch) {
        // group val -> Vec<index> (aggregate/mod.rs:60-71)
        std::unordered_map<K, std::vector<size_t>> group_idxs;
        for (int64_t idx = 0; idx < group_val.len; ++idx) {
            if (!group_val.is_valid(idx)) continue; // NULL keys dropped (:64)
            K key = K(group_val.v[idx]);
            auto it = group_idxs.find(key);
            if (it != group_idxs.end()) it->second.push_back(size_t(idx));
            else group_idxs.emplace(key, std::vector<size_t>{size_t(idx)});
        }
        Batches batches;
        for (auto &kv : group_idxs) { // HashMap::values() order (arbitrary)
            for (size_t idx : kv.second)
                for (auto &op : aggr_ops) op->update(single_batch, int64_t(idx)); // :76-81
            batches.push_back(evaluate_row());
            for (auto &op : aggr_ops) op->clear_state();
        }
        return {concat_batches(out_schema(), batches)};
    }
    Batches group_by_utf8(const Arr &group_val, const Batch &single_batch) {
        std::unordered_map<std::string, std::vector<size_t>> group_idxs;
        for (int64_t idx = 0; idx < group_val.len; ++idx) {
            if (!group_val.is_valid(idx)) continue;
            group_idxs[group_val.str(idx)].push_back(size_t(idx));
        }
        Batches batches;
        for (auto &kv : group_idxs) {
            for (size_t idx : kv.second)
                for (auto &op : aggr_ops) op->update(single_batch, int64_t(idx));
            batches.push_back(evaluate_row());
            for (auto &op : aggr_ops) op->clear_state();
        }
        return {concat_batches(out_schema(), batches)};
    }
    Batches execute() override {
        if (group_expr.empty()) { // :123-139
            Batches batches = input->execute();
            for (const Batch &b : batches)
                for (auto &op : aggr_ops) op->update_batch(b);
            return {evaluate_row()};
        }
        Batches batches = input->execute();
        Batch single_batch = concat_batches(input->schema(), batches); // :143-144
        ArrRef val = group_expr[0]->evaluate(single_batch).into_array(); // only group_expr[0] (Q8)
        switch (val->dtype) {
        case NQE_INT64: return group_by<int64_t>(*val, single_batch);
        case NQE_UINT64: return group_by<uint64_t>(*val, single_batch);
        case NQE_UTF8: return group_by_utf8(*val, single_batch);
        default: fail(NQE_ERR_NOT_SUPPORTED, "group by only support by `Int64`, `UInt64`, `String`");
        }
    }
    std::vector<int> schema() const override { return input->schema(); } // quirk Q8/Q13
};

// ------------------------------------------------------------------ XxHash64
// twox-hash 1.6.3 XxHash64::default() (seed 0), Hasher::write_i64 = write(&to_ne_bytes()).
constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t xx_round(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t xx_merge(uint64_t acc, uint64_t val) { return (acc ^ xx_round(0, val)) * P1 + P4; }
uint64_t xxh64(const uint8_t *p, size_t len, uint64_t seed) {
    const uint8_t *end = p + len;
    uint64_t h;
    auto rd64 = [](const uint8_t *q) { uint64_t v; std::memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t *q) { uint32_t v; std::memcpy(&v, q, 4); return uint64_t(v); };
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xx_round(v1, rd64(p)); p += 8;
            v2 = xx_round(v2, rd64(p)); p += 8;
            v3 = xx_round(v3, rd64(p)); p += 8;
            v4 = xx_round(v4, rd64(p)); p += 8;
        } while (p + 32 <= end);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = xx_merge(h, v1); h = xx_merge(h, v2); h = xx_merge(h, v3); h = xx_merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += uint64_t(len);
    while (p + 8 <= end) { h ^= xx_round(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= uint64_t(*p) * P5; h = rotl(h, 11) * P1; ++p; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
inline uint64_t hash_word(uint64_t w) { return xxh64(reinterpret_cast<const uint8_t *>(&w), 8, 0); }

// hash_join.rs:44-289
struct HashJoin : PhysicalPlan {
    PlanRef left, right;
    bool has_on = true;
    int left_key = 0, right_key = 0; // on[0] resolved to column indices by the host mirror
    std::unordered_map<uint64_t, std::vector<size_t>> hashtable; // never cleared (quirk Q11)
    Batch data;

    Batches execute() override {
        // build() :124-166
        if (!has_on) fail(NQE_ERR_PLAN, "Inner Join on Conditions can't not be empty");
        Batches lb = left->execute();
        Batch single = concat_batches(left->schema(), lb);
        if (left_key < 0 || size_t(left_key) >= single.cols.size()) fail(NQE_ERR_LOGICAL, "ColumnExpr must has name or idx");
        ArrRef left_col = single.cols[size_t(left_key)];
        switch (left_col->dtype) {
        case NQE_INT64: case NQE_UINT64:
            for (int64_t i = 0; i < single.rows; ++i) // validity ignored: value(i)
                hashtable[hash_word(left_col->v[i])].push_back(size_t(i));
            break;
        case NQE_UTF8:
            for (int64_t i = 0; i < single.rows; ++i) {
                std::string s = left_col->str(i);
                hashtable[xxh64(reinterpret_cast<const uint8_t *>(s.data()), s.size(), 0)].push_back(size_t(i));
            }
            break;
        default: fail(NQE_ERR_NOT_IMPLEMENTED, "NotImplemented: join key type (hash_join.rs:161)");
        }
        data = single;
        // probe() :168-254
        Batches rb = right->execute();
        Batches out;
        for (const Batch &rbatch : rb) {
            if (right_key < 0 || size_t(right_key) >= rbatch.cols.size()) fail(NQE_ERR_LOGICAL, "ColumnExpr must has name or idx");
            ArrRef right_col = rbatch.cols[size_t(right_key)];
            std::vector<int64_t> outer_pos, inner_pos;
            switch (right_col->dtype) {
            case NQE_INT64: case NQE_UINT64:
                if (left_col->dtype != right_col->dtype)
                    fail(NQE_ERR_NOT_SUPPORTED, "join key types differ (downcast unwrap panics, hash_join.rs:83)");
                for (int64_t i = 0; i < rbatch.rows; ++i) {
                    uint64_t rv = right_col->v[i];
                    auto it = hashtable.find(hash_word(rv));
                    if (it == hashtable.end()) continue;
                    for (size_t idx : it->second)
                        if (left_col->v[idx] == rv) { outer_pos.push_back(int64_t(idx)); inner_pos.push_back(i); }
                }
                break;
            case NQE_UTF8:
                if (left_col->dtype != NQE_UTF8)
                    fail(NQE_ERR_NOT_SUPPORTED, "join key types differ (downcast unwrap panics)");
                for (int64_t i = 0; i < rbatch.rows; ++i) {
                    std::string s = right_col->str(i);
                    auto it = hashtable.find(xxh64(reinterpret_cast<const uint8_t *>(s.data()), s.size(), 0));
                    if (it == hashtable.end()) continue;
                    for (size_t idx : it->second)
                        if (left_col->str(int64_t(idx)) == s) { outer_pos.push_back(int64_t(idx)); inner_pos.push_back(i); }
                }
                break;
            default: fail(NQE_ERR_NOT_IMPLEMENTED, "NotImplemented: join key type");
            }
            Batch ob;
            ob.rows = int64_t(outer_pos.size());
            for (auto &c : data.cols) ob.cols.push_back(take(*c, outer_pos));
            for (auto &c : rbatch.cols) ob.cols.push_back(take(*c, inner_pos));
            out.push_back(ob);
        }
        return out;
    }
    std::vector<int> schema() const override {
        std::vector<int> s = left->schema();
        for (int d : right->schema()) s.push_back(d);
        return s;
    }
};

// ------------------------------------------------------------------ C API glue
Batch batch_from_columns(const nqe_column *cols, int ncols) {
    Batch b;
    for (int c = 0; c < ncols; ++c) {
        const nqe_column &nc = cols[c];
        if (nc.location != NQE_HOST) fail(NQE_ERR_INVALID_ARGUMENT, "oracle takes host columns only");
        auto a = std::make_shared<Arr>();
        a->dtype = nc.dtype;
        a->len = nc.length;
        size_t n = size_t(nc.length);
        if (is_word_type(nc.dtype)) {
            a->v.resize(n);
            if (n) std::memcpy(a->v.data(), nc.values, n * 8);
        } else if (nc.dtype == NQE_BOOLEAN) {
            a->bits.assign(bm_bytes(nc.length), 0);
            if (n) std::memcpy(a->bits.data(), nc.values, bm_bytes(nc.length));
        } else if (nc.dtype == NQE_UTF8) {
            a->offs.resize(n + 1);
            std::memcpy(a->offs.data(), nc.values, (n + 1) * 4);
            a->data.assign(static_cast<const char *>(nc.data), size_t(nc.data_length));
        } else {
            fail(NQE_ERR_INVALID_ARGUMENT, "unsupported column dtype");
        }
        if (nc.validity && n) a->valid.assign(nc.validity, nc.validity + bm_bytes(nc.length));
        if (c == 0) b.rows = nc.length;
        else if (b.rows != nc.length) fail(NQE_ERR_ARROW, "all columns in a record batch must have the same length");
        b.cols.push_back(a);
    }
    return b;
}

} // namespace

// A Vec<RecordBatch> (an operator input/output) plus its schema dtypes.
struct orc_batches {
    Batches batches;
    std::vector<int> dtypes;
};

namespace {
PlanRef scan_of(const orc_batches *t) {
    auto s = std::make_shared<ScanPlan>();
    s->table = t->batches;
    s->dtypes = t->dtypes;
    return s;
}
orc_batches *wrap(Batches b, std::vector<int> dtypes) {
    auto *o = new orc_batches;
    o->batches = std::move(b);
    if (!o->batches.empty()) o->dtypes = schema_of(o->batches[0]);
    else o->dtypes = std::move(dtypes);
    return o;
}
template <typename F> int guarded(F &&f) {
    try {
        f();
        return NQE_OK;
    } catch (const OracleError &e) {
        g_last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        g_last_error = "out of memory";
        return NQE_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return NQE_ERR_OTHERS;
    }
}
} // namespace

extern "C" {

const char *orc_last_error(void) { return g_last_error.c_str(); }

orc_batches *orc_batches_new(const int32_t *dtypes, int32_t ncols) {
    auto *b = new orc_batches;
    for (int i = 0; i < ncols; ++i) b->dtypes.push_back(dtypes[i]);
    return b;
}
void orc_batches_free(orc_batches *b) { delete b; }
int orc_batches_push(orc_batches *b, const nqe_column *cols, int32_t ncols) {
    return guarded([&] {
        Batch bt = batch_from_columns(cols, ncols);
        if (b->dtypes.empty()) b->dtypes = schema_of(bt);
        b->batches.push_back(std::move(bt));
    });
}
int32_t orc_batches_count(const orc_batches *b) { return int32_t(b->batches.size()); }
int32_t orc_batches_num_columns(const orc_batches *b) { return int32_t(b->dtypes.size()); }
int64_t orc_batch_num_rows(const orc_batches *b, int32_t i) { return b->batches[size_t(i)].rows; }
int orc_batch_column(const orc_batches *b, int32_t i, int32_t c, nqe_column *out) {
    return guarded([&] {
        if (i < 0 || size_t(i) >= b->batches.size() || c < 0 || size_t(c) >= b->batches[size_t(i)].cols.size())
            fail(NQE_ERR_INVALID_ARGUMENT, "batch/column index out of range");
        const Arr &a = *b->batches[size_t(i)].cols[size_t(c)];
        std::memset(out, 0, sizeof(*out));
        out->dtype = a.dtype;
        out->location = NQE_HOST;
        out->length = a.len;
        out->null_count = a.null_count();
        out->validity = a.valid.empty() ? nullptr : a.valid.data();
        if (a.dtype == NQE_BOOLEAN) out->values = a.bits.data();
        else if (a.dtype == NQE_UTF8) { out->values = a.offs.data(); out->data = a.data.data(); out->data_length = int64_t(a.data.size()); }
        else out->values = a.v.data();
    });
}

// ScanPlan over a MemTable with an optional projection (n < 0 → None)
int orc_scan(const orc_batches *table, const int32_t *projection, int32_t n, orc_batches **out) {
    return guarded([&] {
        auto s = std::make_shared<ScanPlan>();
        s->table = table->batches;
        s->dtypes = table->dtypes;
        if (n >= 0) { s->has_projection = true; s->projection.assign(projection, projection + n); }
        *out = wrap(s->execute(), s->schema());
    });
}

int orc_expr_evaluate(const orc_batches *in, int32_t batch, const nqe_expr_node *nodes, int32_t n, orc_batches **out) {
    return guarded([&] {
        ExprRef e = build_expr(nodes, n);
        Batch ob;
        const Batch &ib = in->batches.at(size_t(batch));
        ob.cols.push_back(e->evaluate(ib).into_array());
        ob.rows = ob.cols[0]->len;
        *out = wrap({ob}, {});
    });
}

int orc_selection(const orc_batches *in, const nqe_expr_node *pred, int32_t n, orc_batches **out) {
    return guarded([&] {
        SelectionPlan p;
        p.input = scan_of(in);
        p.expr = build_expr(pred, n);
        *out = wrap(p.execute(), in->dtypes);
    });
}

int orc_projection(const orc_batches *in, const nqe_expr_node *nodes, const int32_t *offsets, int32_t nexprs,
                   orc_batches **out) {
    return guarded([&] {
        ProjectionPlan p;
        p.input = scan_of(in);
        p.empty_schema = (nexprs == 0);
        for (int e = 0; e < nexprs; ++e) p.exprs.push_back(build_expr(nodes + offsets[e], offsets[e + 1] - offsets[e]));
        *out = wrap(p.execute(), {});
    });
}

int orc_limit(const orc_batches *in, int64_t n, orc_batches **out) {
    return guarded([&] { LimitPlan p; p.input = scan_of(in); p.n = n; *out = wrap(p.execute(), in->dtypes); });
}
int orc_offset(const orc_batches *in, int64_t n, orc_batches **out) {
    return guarded([&] { OffsetPlan p; p.input = scan_of(in); p.n = n; *out = wrap(p.execute(), in->dtypes); });
}

// PhysicalAggregatePlan over (optional SelectionPlan over) a scan; `executions` > 1 re-runs
// execute() on the same plan object to expose quirk Q9.
int orc_aggregate(const orc_batches *in, const nqe_expr_node *pred, int32_t pred_n, const nqe_expr_node *group,
                  int32_t group_n, const nqe_aggregate *aggs, int32_t naggs, int32_t executions, orc_batches **out) {
    return guarded([&] {
        AggregatePlan p;
        PlanRef src = scan_of(in);
        if (pred_n > 0) {
            auto sel = std::make_shared<SelectionPlan>();
            sel->input = src;
            sel->expr = build_expr(pred, pred_n);
            src = sel;
        }
        p.input = src;
        if (group_n > 0) p.group_expr.push_back(build_expr(group, group_n));
        for (int i = 0; i < naggs; ++i) p.aggr_ops.push_back(make_agg(aggs[i]));
        Batches res;
        for (int k = 0; k < (executions < 1 ? 1 : executions); ++k) res = p.execute();
        *out = wrap(std::move(res), p.out_schema());
    });
}

int orc_hash_join(const orc_batches *left, const orc_batches *right, int32_t left_key, int32_t right_key,
                  orc_batches **out) {
    return guarded([&] {
        HashJoin j;
        j.left = scan_of(left);
        j.right = scan_of(right);
        j.has_on = left_key >= 0 && right_key >= 0;
        j.left_key = left_key;
        j.right_key = right_key;
        std::vector<int> sch = j.schema();
        *out = wrap(j.execute(), sch);
    });
}

// the same plan object executed `executions` times: the hash table is never cleared (quirk Q11), so the k-th execute()
// emits every match k times (Vec<idx> = [first build..., second build..., ...], hash_join.rs:58-78)
int orc_hash_join_n(const orc_batches *left, const orc_batches *right, int32_t left_key, int32_t right_key, int32_t executions,
                    orc_batches **out) {
    return guarded([&] {
        HashJoin j;
        j.left = scan_of(left);
        j.right = scan_of(right);
        j.has_on = left_key >= 0 && right_key >= 0;
        j.left_key = left_key;
        j.right_key = right_key;
        std::vector<int> sch = j.schema();
        Batches last;
        for (int e = 0; e < std::max(1, executions); ++e) last = j.execute();
        *out = wrap(last, sch);
    });
}

uint64_t orc_xxhash64_word(uint64_t w) { return hash_word(w); }
uint64_t orc_xxhash64(const uint8_t *p, size_t n, uint64_t seed) { return xxh64(p, n, seed); }

// Synthetic generators of SURVEY §8d (host twin of nqe_synth_fill): splitmix64(seed + i).
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
// ------------------------------------------------------------------ CsvTable::try_create (datasource/csv.rs:53-86)
// Sequential restatement of what the reference gets from arrow-rs 13 + the csv crate (neither is under /root/reference:
// arrow 13.0.0 / csv 1.1 / lexical-core 0.8 per Cargo.lock), written independently of the device code:
//   * records: '"'-quoted fields with "" escapes (a quote is special only at the start of a field; text after a closing
//     quote is appended literally), terminators \r, \n, \r\n, empty lines skipped, a last record without terminator is kept;
//   * schema: infer_reader_schema over the first max_read_records records (non-empty fields only): leading '"' → Utf8,
//     true/false → Boolean, ^-?\d+\.\d+$ → Float64, ^-?\d+$ → Int64, ISO date(-time) → Date (unsupported here), else
//     Utf8; {Int64, Float64} → Float64, other mixes → Utf8; nullable = an empty field was seen;
//   * values: only the first batch of batch_size rows (quirk Q1); empty numeric/Boolean → NULL; Int64/Float64 via the
//     C library after a lexical-core grammar check (strtod is correctly rounded); Utf8 never NULL.
namespace {
struct CsvReaderState {
    const uint8_t *p;
    int64_t n, i = 0;
    uint8_t delim;
    // next record; false at end of input
    bool next(std::vector<std::string> &fields) {
        fields.clear();
        while (i < n && (p[i] == '\n' || p[i] == '\r')) ++i; // empty lines
        if (i >= n) return false;
        std::string cur;
        for (;;) {
            // one field
            cur.clear();
            if (i < n && p[i] == '"') {
                ++i;
                for (;;) {
                    if (i >= n) break;
                    if (p[i] == '"') {
                        if (i + 1 < n && p[i + 1] == '"') { cur.push_back('"'); i += 2; continue; }
                        ++i; // closing quote
                        break;
                    }
                    cur.push_back(char(p[i++]));
                }
                // csv-core: after the closing quote the field continues as an unquoted one (quotes are then literal)
                while (i < n && p[i] != delim && p[i] != '\n' && p[i] != '\r') cur.push_back(char(p[i++]));
            } else {
                while (i < n && p[i] != delim && p[i] != '\n' && p[i] != '\r') cur.push_back(char(p[i++]));
            }
            fields.push_back(cur);
            if (i < n && p[i] == delim) { ++i; continue; }
            break;
        }
        if (i < n) ++i; // one terminator byte (\r\n leaves an empty line, skipped above)
        return true;
    }
};

bool csv_all_digits(const std::string &s, size_t a, size_t b) {
    if (a >= b) return false;
    for (size_t k = a; k < b; ++k)
        if (!isdigit((unsigned char)s[k])) return false;
    return true;
}
bool csv_ieq(const std::string &s, const char *w) {
    if (s.size() != strlen(w)) return false;
    for (size_t k = 0; k < s.size(); ++k)
        if (tolower((unsigned char)s[k]) != w[k]) return false;
    return true;
}
enum { CI_UTF8 = 1, CI_BOOL = 2, CI_F64 = 4, CI_I64 = 8, CI_DATE = 16 };
int csv_infer(const std::string &s) {
    if (s[0] == '"') return CI_UTF8;
    if (csv_ieq(s, "true") || csv_ieq(s, "false")) return CI_BOOL;
    size_t b = s[0] == '-' ? 1 : 0, dot = s.find('.');
    if (dot != std::string::npos && csv_all_digits(s, b, dot) && csv_all_digits(s, dot + 1, s.size())) return CI_F64;
    if (csv_all_digits(s, b, s.size())) return CI_I64;
    auto shape = [&](const char *pat) {
        if (s.size() != strlen(pat)) return false;
        for (size_t k = 0; k < s.size(); ++k)
            if (pat[k] == 'd' ? !isdigit((unsigned char)s[k]) : s[k] != pat[k]) return false;
        return true;
    };
    if (shape("dddd-dd-ddTdd:dd:dd") || shape("dddd-dd-dd")) return CI_DATE;
    return CI_UTF8;
}
// lexical-core float grammar (no whitespace, no hex, digits required, exponent digits required)
bool csv_float_grammar(const std::string &s) {
    size_t k = 0;
    if (k < s.size() && (s[k] == '+' || s[k] == '-')) ++k;
    std::string rest = s.substr(k);
    if (csv_ieq(rest, "nan") || csv_ieq(rest, "inf") || csv_ieq(rest, "infinity")) return true;
    size_t digits = 0;
    while (k < s.size() && isdigit((unsigned char)s[k])) { ++k; ++digits; }
    if (k < s.size() && s[k] == '.') {
        ++k;
        while (k < s.size() && isdigit((unsigned char)s[k])) { ++k; ++digits; }
    }
    if (!digits) return false;
    if (k < s.size() && (s[k] == 'e' || s[k] == 'E')) {
        ++k;
        if (k < s.size() && (s[k] == '+' || s[k] == '-')) ++k;
        size_t ed = 0;
        while (k < s.size() && isdigit((unsigned char)s[k])) { ++k; ++ed; }
        if (!ed) return false;
    }
    return k == s.size();
}
} // namespace

// names_out: '\0'-joined column names (caller buffer)
int orc_csv_read(const uint8_t *bytes, int64_t nbytes, int32_t has_header, int32_t delimiter, int64_t max_read_records, int64_t batch_size,
                 orc_batches **out, char *names_out, int64_t names_cap, int32_t *nullable_out, int32_t nullable_cap) {
    return guarded([&] {
        // ---- schema
        CsvReaderState rd{bytes, nbytes, 0, uint8_t(delimiter)};
        std::vector<std::string> rec, names;
        if (!rd.next(rec)) fail(NQE_ERR_ARROW, "csv: empty file");
        const size_t nc = rec.size();
        if (has_header) names = rec;
        else {
            for (size_t c = 0; c < nc; ++c) names.push_back("column_" + std::to_string(c + 1));
            rd.i = 0;
        }
        std::vector<int> poss(nc, 0), nul(nc, 0);
        for (int64_t r = 0; max_read_records < 0 || r < max_read_records; ++r) {
            if (!rd.next(rec)) break;
            if (rec.size() != nc) fail(NQE_ERR_ARROW, "csv: record with a different number of fields");
            for (size_t c = 0; c < nc; ++c) {
                if (rec[c].empty()) nul[c] = 1;
                else poss[c] |= csv_infer(rec[c]);
            }
        }
        std::vector<int> dts(nc);
        for (size_t c = 0; c < nc; ++c) {
            const int p = poss[c];
            if (p == CI_BOOL) dts[c] = NQE_BOOLEAN;
            else if (p == CI_I64) dts[c] = NQE_INT64;
            else if (p == CI_F64 || p == (CI_F64 | CI_I64)) dts[c] = NQE_FLOAT64;
            else if (p == CI_DATE) fail(NQE_ERR_NOT_SUPPORTED, "csv: Date32/Date64 columns are outside the hot path's types");
            else dts[c] = NQE_UTF8;
        }
        // ---- first batch
        CsvReaderState rd2{bytes, nbytes, 0, uint8_t(delimiter)};
        if (has_header) rd2.next(rec);
        std::vector<Builder> bs;
        for (size_t c = 0; c < nc; ++c) bs.emplace_back(dts[c]);
        int64_t rows = 0;
        while ((batch_size < 0 || rows < batch_size) && rd2.next(rec)) {
            if (rec.size() != nc) fail(NQE_ERR_ARROW, "csv: record with a different number of fields than the schema");
            for (size_t c = 0; c < nc; ++c) {
                const std::string &f = rec[c];
                if (dts[c] == NQE_UTF8) {
                    Arr tmp;
                    tmp.dtype = NQE_UTF8;
                    tmp.offs = {0, int32_t(f.size())};
                    tmp.data = f;
                    bs[c].append_str(true, tmp, 0);
                } else if (f.empty()) {
                    if (dts[c] == NQE_BOOLEAN) bs[c].append_bool(false, false);
                    else bs[c].append_word(false, 0);
                } else if (dts[c] == NQE_BOOLEAN) {
                    if (csv_ieq(f, "true")) bs[c].append_bool(true, true);
                    else if (csv_ieq(f, "false")) bs[c].append_bool(true, false);
                    else fail(NQE_ERR_ARROW, "csv: error while parsing a Boolean value");
                } else if (dts[c] == NQE_INT64) {
                    size_t k = (f[0] == '+' || f[0] == '-') ? 1 : 0;
                    if (!csv_all_digits(f, k, f.size())) fail(NQE_ERR_ARROW, "csv: error while parsing an Int64 value");
                    errno = 0;
                    long long v = strtoll(f.c_str(), nullptr, 10);
                    if (errno) fail(NQE_ERR_ARROW, "csv: Int64 overflow");
                    bs[c].append_word(true, uint64_t(v));
                } else {
                    if (!csv_float_grammar(f)) fail(NQE_ERR_ARROW, "csv: error while parsing a Float64 value");
                    bs[c].append_word(true, f64_bits(strtod(f.c_str(), nullptr)));
                }
            }
            ++rows;
        }
        auto *ob = new orc_batches;
        Batch bt;
        bt.rows = rows;
        for (size_t c = 0; c < nc; ++c) {
            ob->dtypes.push_back(dts[c]);
            bt.cols.push_back(bs[c].a);
        }
        ob->batches.push_back(std::move(bt));
        *out = ob;
        std::string joined;
        for (auto &nm : names) { joined += nm; joined.push_back('\0'); }
        if (int64_t(joined.size()) > names_cap || int32_t(nc) > nullable_cap) { delete ob; fail(NQE_ERR_INVALID_ARGUMENT, "csv: output buffers too small"); }
        std::memcpy(names_out, joined.data(), joined.size());
        for (size_t c = 0; c < nc; ++c) nullable_out[c] = nul[c];
    });
}

int orc_synth_fill(int32_t kind, uint64_t seed, int64_t first_row, int64_t n, uint64_t modulus, int64_t base, void *out) {
    return guarded([&] {
        uint64_t *o = static_cast<uint64_t *>(out);
        for (int64_t r = 0; r < n; ++r) {
            uint64_t i = uint64_t(first_row + r);
            switch (kind) {
            case NQE_SYNTH_ROWID: o[r] = i; break;
            case NQE_SYNTH_UNIFORM:
                if (modulus == 0) fail(NQE_ERR_INVALID_ARGUMENT, "modulus must be > 0");
                o[r] = uint64_t(int64_t(splitmix64(seed + i) % modulus) + base);
                break;
            case NQE_SYNTH_F64_0_100: o[r] = f64_bits(double(splitmix64(seed + i) >> 11) * 0x1.0p-53 * 100.0); break;
            default: fail(NQE_ERR_INVALID_ARGUMENT, "unknown synth kind");
            }
        }
    });
}

// ---- an OPTIMISED multi-core CPU form of the headline query, for bench.py's optional second CPU number (last row): NOT
// the reference's algorithm (which is single-threaded by construction, aggregate/mod.rs:33) — what a tuned CPU engine would do for
// `select count(v), sum(v), min(v), max(v) from t where id < limit group by id % modulus` over plain columns: `threads` workers over
// contiguous row ranges, each with its own direct-mapped table of `modulus` groups (modulus <= 65536, id >= 0), merged at the end.
// out: modulus x {count, sum, min, max} as doubles.  Checked against orc_aggregate by tests/test_oracle_golden.py.
// `v_dtype`: NQE_FLOAT64, or NQE_INT64 / NQE_UINT64 values accumulated `as f64` (sum.rs:86-101); `has_limit` 0: no predicate (C3);
// modulus <= 2^21 (the many-group configs: key = a column in [0, G) and modulus = G).  ids must be >= 0 (then the reference's truncated
// signed `%` is the unsigned one).  NaN-free values (the synthetic columns): min / max are plain compares from -DBL_MAX / DBL_MAX.
// The masked form adds validity (Arrow semantics as the reference's operators read them): `id_valid` / `v_valid` are byte masks (one
// uint8_t per row, 0 = NULL; nullptr = no NULLs).  A NULL id makes predicate and key NULL: SelectionPlan emits the row as a NULL row
// (selection.rs:46, quirk Q4) and the aggregate drops NULL group keys (aggregate/mod.rs:64) — the row counts nowhere.  A NULL value
// in a row that survives: the group exists, count / sum / min / max skip it (count.rs:63, sum.rs:86-101, max.rs:38).  out: modulus x
// `stride` doubles = count, sum, min, max and (stride >= 5) the ROWS of the group whatever their value's validity — rows > 0 is what
// makes a key a group (a group of NULL values only has count 0, sum 0.0, min f64::MAX, max f64::MIN: quirk Q10).
int orc_grouped_parallel_masked(const int64_t *ids, const uint8_t *id_valid, const void *v, const uint8_t *v_valid, int32_t v_dtype, int64_t n, int32_t has_limit,
                                int64_t limit, int64_t modulus, int32_t threads, int32_t stride, double *out) {
    return guarded([&] {
        if (modulus <= 0 || modulus > (int64_t(1) << 21) || threads < 1 || n < 0 || stride < 4) fail(NQE_ERR_INVALID_ARGUMENT, "orc_grouped_parallel: bad arguments");
        if (v_dtype != NQE_FLOAT64 && v_dtype != NQE_INT64 && v_dtype != NQE_UINT64) fail(NQE_ERR_INVALID_ARGUMENT, "orc_grouped_parallel: value type");
        struct Acc { uint64_t cnt, rows; double sum, mn, mx; };
        std::vector<std::vector<Acc>> part(size_t(threads), std::vector<Acc>(size_t(modulus), Acc{0, 0, 0.0, DBL_MAX, -DBL_MAX}));
        std::vector<int> negative(size_t(threads), 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                Acc *a = part[size_t(t)].data();
                const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
                const double *vf = static_cast<const double *>(v);
                const int64_t *vi = static_cast<const int64_t *>(v);
                const uint64_t *vu = static_cast<const uint64_t *>(v);
                for (int64_t r = lo; r < hi; ++r) {
                    if (id_valid && !id_valid[r]) continue;
                    const int64_t id = ids[r];
                    if (id < 0) { negative[size_t(t)] = 1; continue; }
                    if (has_limit && id >= limit) continue;
                    Acc &g = a[uint64_t(id) % uint64_t(modulus)];
                    g.rows += 1;
                    if (v_valid && !v_valid[r]) continue;
                    const double x = v_dtype == NQE_FLOAT64 ? vf[r] : v_dtype == NQE_INT64 ? double(vi[r]) : double(vu[r]);
                    g.cnt += 1;
                    g.sum += x;
                    g.mn = x < g.mn ? x : g.mn;
                    g.mx = x > g.mx ? x : g.mx;
                }
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < threads; ++t)
            if (negative[size_t(t)]) fail(NQE_ERR_INVALID_ARGUMENT, "orc_grouped_parallel: negative id");
        for (int64_t k = 0; k < modulus; ++k) {
            Acc m{0, 0, 0.0, DBL_MAX, -DBL_MAX};
            for (int t = 0; t < threads; ++t) {
                const Acc &g = part[size_t(t)][size_t(k)];
                m.cnt += g.cnt;
                m.rows += g.rows;
                m.sum += g.sum;
                m.mn = g.mn < m.mn ? g.mn : m.mn;
                m.mx = g.mx > m.mx ? g.mx : m.mx;
            }
            double *o = out + size_t(stride) * size_t(k);
            o[0] = double(m.cnt); o[1] = m.sum; o[2] = m.mn; o[3] = m.mx;
            if (stride >= 5) o[4] = double(m.rows);
        }
    });
}

int orc_grouped_parallel(const int64_t *ids, const void *v, int32_t v_dtype, int64_t n, int32_t has_limit, int64_t limit, int64_t modulus, int32_t threads,
                         double *out) {
    return orc_grouped_parallel_masked(ids, nullptr, v, nullptr, v_dtype, n, has_limit, limit, modulus, threads, 4, out);
}

int orc_headline_parallel(const int64_t *ids, const double *v, int64_t n, int64_t limit, int64_t modulus, int32_t threads, double *out) {
    if (modulus > 65536) return guarded([&] { fail(NQE_ERR_INVALID_ARGUMENT, "orc_headline_parallel: bad arguments"); });
    return orc_grouped_parallel(ids, v, NQE_FLOAT64, n, 1, limit, modulus, threads, out);
}

// orc_synth_fill on `threads` threads (bench.py / tests: host copies of 10^8..10^9-row synthetic columns in a second, not ten)
int orc_synth_fill_mt(int32_t kind, uint64_t seed, int64_t first_row, int64_t n, uint64_t modulus, int64_t base, void *out, int32_t threads) {
    if (threads < 1) threads = 1;
    std::vector<int> rc(size_t(threads), 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
            rc[size_t(t)] = orc_synth_fill(kind, seed, first_row + lo, hi - lo, modulus, base, static_cast<uint64_t *>(out) + lo);
        });
    for (auto &th : pool) th.join();
    for (int t = 0; t < threads; ++t)
        if (rc[size_t(t)]) return rc[size_t(t)];
    return 0;
}

} // extern "C"
