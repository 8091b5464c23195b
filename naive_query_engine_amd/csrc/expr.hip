// expr.hip — PhysicalExpr::evaluate on the device (reference: src/physical_plan/expression/
// binary.rs:108-155, column.rs:39-57, literal.rs:32-34; arrow-rs 13 compare / kleene /
// arithmetic kernels at the call sites binary.rs:127-153).
//
// Two evaluation forms:
//   * general: one streaming kernel per binary node (what arrow does), except that literals
//     stay scalars in registers instead of being materialised as n-row columns
//     (logical_plan/expression.rs:210-222, the TODO at binary.rs:121);
//   * fused:   `col [op lit]{0,2}` shapes (SimpleExpr) are evaluated inside the consumer
//     kernel (compaction, aggregation) from the streamed word — no temporary at all.
#include <cmath>
#include <cstring>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {

namespace {

struct Node {
    int kind = 0, op = 0, column = 0, dtype = 0;
    bool lit_null = false;
    uint64_t lit = 0;
    std::string lit_str;       // Utf8 literal
    int left = -1, right = -1; // children (indices into the node vector)
    int out_dtype = NQE_NULLTYPE;
};

bool is_compare(int op) { return op >= NQE_OP_EQ && op <= NQE_OP_GT_EQ; }
bool is_logic(int op) { return op == NQE_OP_AND || op == NQE_OP_OR; }
bool is_arith(int op) { return op >= NQE_OP_PLUS && op <= NQE_OP_MODULOS; }

// builds the tree and type-checks it exactly where binary.rs does
std::vector<Node> parse(const nqe_table *in, const nqe_expr_node *nodes, int n, int *root) {
    if (!nodes || n <= 0) fail(NQE_ERR_INVALID_ARGUMENT, "empty expression");
    std::vector<Node> t;
    std::vector<int> st;
    for (int i = 0; i < n; ++i) {
        const nqe_expr_node &nd = nodes[i];
        Node x;
        x.kind = nd.kind;
        if (nd.kind == NQE_EXPR_COLUMN) {
            if (nd.column < 0 || size_t(nd.column) >= in->cols.size())
                fail(NQE_ERR_NOT_SUPPORTED, "column index out of range (RecordBatch::column panics)");
            x.column = nd.column;
            x.out_dtype = in->cols[size_t(nd.column)].dtype;
        } else if (nd.kind == NQE_EXPR_LITERAL) {
            x.dtype = nd.dtype;
            x.lit_null = nd.is_null != 0 || nd.dtype == NQE_NULLTYPE;
            x.lit = nd.dtype == NQE_BOOLEAN ? uint64_t(nd.value.boolean != 0) : nd.value.u64;
            x.out_dtype = nd.dtype;
            if (nd.dtype == NQE_UTF8) {
                x.lit = 0;
                if (!x.lit_null) {
                    if (nd.utf8_length < 0 || (nd.utf8_length > 0 && !nd.value.utf8)) fail(NQE_ERR_INVALID_ARGUMENT, "Utf8 literal without bytes");
                    x.lit_str.assign(nd.value.utf8 ? nd.value.utf8 : "", size_t(nd.utf8_length));
                }
            }
        } else if (nd.kind == NQE_EXPR_BINARY) {
            if (st.size() < 2) fail(NQE_ERR_INVALID_ARGUMENT, "malformed expression");
            x.right = st.back(); st.pop_back();
            x.left = st.back(); st.pop_back();
            x.op = nd.op;
            int ldt = t[size_t(x.left)].out_dtype, rdt = t[size_t(x.right)].out_dtype;
            if (ldt != rdt) // binary.rs:114-119
                fail(NQE_ERR_INTERVAL, "Cannot evaluate binary expression with types " + std::to_string(ldt) + " and " +
                                           std::to_string(rdt));
            if (is_compare(x.op)) {
                if (ldt == NQE_NULLTYPE) fail(NQE_ERR_ARROW, "comparison on Null arrays is not supported");
                x.out_dtype = NQE_BOOLEAN;
            } else if (is_logic(x.op)) {
                if (ldt != NQE_BOOLEAN) // binary_op! (binary.rs:32-42)
                    fail(NQE_ERR_INTERVAL, "Cannot evaluate binary expression And/Or with non-Boolean types");
                x.out_dtype = NQE_BOOLEAN;
            } else if (is_arith(x.op)) {
                if (!is_word_type(ldt)) // arithemic_op! `_ => unimplemented!()` (binary.rs:85)
                    fail(NQE_ERR_NOT_SUPPORTED, "arithmetic on this type is unimplemented!() (binary.rs:85)");
                x.out_dtype = ldt;
            } else {
                fail(NQE_ERR_INVALID_ARGUMENT, "unknown operator");
            }
        } else {
            fail(NQE_ERR_INVALID_ARGUMENT, "unknown expression node kind");
        }
        t.push_back(x);
        st.push_back(int(t.size()) - 1);
    }
    if (st.size() != 1) fail(NQE_ERR_INVALID_ARGUMENT, "malformed expression");
    *root = st[0];
    return t;
}

OpAux make_aux(int op, int dt, uint64_t lit) {
    OpAux a;
    a.pow2_shift = -1;
    a.more = -1;
    a.abs_lit = 0;
    a.magic = 0;
    if (op == NQE_OP_DIVIDE && dt == NQE_FLOAT64) {
        // x / ±2^k  ==  x * ±2^-k bit for bit (scaling by a power of two is exact; where the quotient is subnormal both round the same
        // real number): a multiplication instead of the ~40-instruction Float64 division — when 2^k and 2^-k are both normal
        const uint64_t mant = lit & 0x000fffffffffffffull, ex = (lit >> 52) & 0x7ff;
        if (mant == 0 && ex >= 2 && ex <= 2044) {
            a.more = -2;
            a.magic = (lit & 0x8000000000000000ull) | ((2046 - ex) << 52);
        }
        return a;
    }
    if ((op == NQE_OP_DIVIDE || op == NQE_OP_MODULOS) && (dt == NQE_INT64 || dt == NQE_UINT64) && lit != 0) {
        uint64_t ab = lit;
        if (dt == NQE_INT64 && int64_t(lit) < 0) ab = 0ull - lit;
        a.abs_lit = ab;
        if ((ab & (ab - 1)) == 0) {
            int s = 0;
            while ((ab >> s) != 1) ++s;
            a.pow2_shift = s;
        } else {
            // unsigned 64-bit division by an invariant divisor (Granlund–Montgomery, the branch-free "add"
            // form): magic = floor(2^(64+L) / d) * 2 + adjustment + 1 with L = floor(log2 d)
            int L = 63;
            while (!((ab >> L) & 1)) --L;
            unsigned __int128 num = (unsigned __int128)1 << (64 + L);
            uint64_t pm = uint64_t(num / ab);
            uint64_t rem = uint64_t(num % ab);
            pm += pm;
            uint64_t twice = rem + rem;
            if (twice >= ab || twice < rem) pm += 1;
            a.magic = pm + 1;
            a.more = L;
        }
    }
    return a;
}

// col [op lit]{0,SIMPLE_MAX_OPS}
bool match_simple(const std::vector<Node> &t, int i, SimpleExpr *s) {
    const Node &x = t[size_t(i)];
    if (x.kind == NQE_EXPR_COLUMN) {
        std::memset(s, 0, sizeof(*s));
        s->col = x.column;
        s->src_dtype = x.out_dtype;
        s->out_dtype = x.out_dtype;
        for (int k = 0; k < SIMPLE_MAX_OPS; ++k) s->aux[k].pow2_shift = s->aux[k].more = -1;
        return true; // a bare column of any type (Utf8 included) passes through
    }
    if (x.kind != NQE_EXPR_BINARY || is_logic(x.op)) return false;
    const Node &l = t[size_t(x.left)], &r = t[size_t(x.right)];
    bool lit_left;
    int sub;
    const Node *litn;
    if (r.kind == NQE_EXPR_LITERAL && !r.lit_null && l.kind != NQE_EXPR_LITERAL) {
        lit_left = false; sub = x.left; litn = &r;
    } else if (l.kind == NQE_EXPR_LITERAL && !l.lit_null && r.kind != NQE_EXPR_LITERAL) {
        lit_left = true; sub = x.right; litn = &l;
    } else {
        return false;
    }
    if (litn->dtype == NQE_UTF8) return false; // string compares have their own kernel
    if (!match_simple(t, sub, s) || s->nops >= SIMPLE_MAX_OPS) return false;
    int k = s->nops++;
    s->op[k] = x.op;
    s->lit_left[k] = lit_left ? 1 : 0;
    s->op_dtype[k] = litn->dtype;
    s->lit[k] = litn->lit;
    s->aux[k] = lit_left ? make_aux(0, 0, 0) : make_aux(x.op, litn->dtype, litn->lit);
    s->out_dtype = x.out_dtype;
    return true;
}

// ------------------------------------------------------------------ kernels
struct Operand {
    const void *values;   // words or packed bits
    const uint8_t *valid; // or null
    uint64_t lit;
    int32_t is_lit;
    int32_t lit_null;
};

// out = a op b, 64 consecutive rows per wave so that ballots form the packed result words.
// bool_out: result is Boolean (compare / and / or) → packed into out_bits.
__global__ void __launch_bounds__(256) binary_kernel(Operand a, Operand b, int op, int dt, OpAux aux, int64_t n,
                                                     uint64_t *out_words, uint64_t *out_bits, uint64_t *out_valid,
                                                     int *flags) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t n_pad = (n + 63) / 64 * 64;
    const bool logic = op == NQE_OP_AND || op == NQE_OP_OR;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < n_pad; j += stride) {
        const bool in = j < n;
        bool av = in && (a.is_lit ? !a.lit_null : (a.valid ? get_bit(a.valid, j) : true));
        bool bv = in && (b.is_lit ? !b.lit_null : (b.valid ? get_bit(b.valid, j) : true));
        uint64_t x = a.is_lit ? a.lit : (in ? load_word(a.values, dt, j) : 0);
        uint64_t y = b.is_lit ? b.lit : (in ? load_word(b.values, dt, j) : 0);
        bool ok;
        uint64_t r;
        if (logic) {
            // and_kleene / or_kleene
            bool lb = av && x, rb = bv && y;
            if (op == NQE_OP_AND) {
                ok = (av && bv) || (av && !lb) || (bv && !rb);
                r = ok && lb && rb;
            } else {
                ok = (av && bv) || lb || rb;
                r = ok && (lb || rb);
            }
        } else {
            ok = av && bv;
            r = in ? apply_binary(op, dt, x, y, aux, ok, flags) : 0;
        }
        if (out_words) {
            if (in) out_words[j] = ok ? r : 0;
        } else {
            uint64_t w = __ballot(ok && r);
            if (lane_id() == 0) out_bits[j >> 6] = w;
        }
        if (out_valid) {
            uint64_t v = __ballot(ok);
            if (lane_id() == 0) out_valid[j >> 6] = v;
        }
    }
}

__global__ void fill_words_kernel(uint64_t *out, uint64_t v, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < n; j += stride) out[j] = v;
}

// ------------------------------------------------------------------ fused whole-tree evaluation
// A tree of binary nodes is evaluated in ONE pass by a small stack machine: one instruction per BINARY node
// (post-order), whose operands are a literal (SGPR broadcast), a pre-loaded column word, or the top of a register-
// resident stack of intermediate results.  Control flow is wave-uniform (the program lives in the kernel arguments).
// Reads each referenced column once and writes the result once — no temporaries (the reference / arrow materialise one
// full column per node plus one per literal).
//
// Each wave walks 256-row chunks; a lane owns EX_ROWS rows (chunk + r*64 + lane: every access is a coalesced 512-byte
// wave access and a ballot is one bitmap word).  All column loads of a chunk are issued back to back before anything is
// consumed; the interpretive overhead (scalar instruction fetch, op/dtype branch chain) is paid once per EX_ROWS rows and
// the next instruction is fetched while the current one executes.  The stack keeps its top at level 0 by register moves:
// `stack op x` (the common left-deep shape) moves nothing.
constexpr int EX_MAX_INSTR = 16, EX_MAX_COLS = 4, EX_MAX_DEPTH = 3, EX_ROWS = 4;
enum ExSrc : int32_t { EX_STACK = 0, EX_LIT = 1, EX_LIT_NULL = 2, EX_COL = 4 /* + slot */ };
struct ExInstr {
    int32_t op, dt;       // operator, operand dtype
    int32_t a_src, b_src; // ExSrc
    uint64_t lit_a, lit_b;
    OpAux aux;            // host-prepared divisor constants when b is a literal
};
struct ExProgram {
    int32_t n, ncols;
    ExInstr ins[EX_MAX_INSTR];
    const void *col_values[EX_MAX_COLS];
    const uint8_t *col_valid[EX_MAX_COLS];
    int32_t col_dtype[EX_MAX_COLS];
};

} // namespace
} // namespace nqe
#include "expr_jit.hpp" // the same programs as straight-line source, compiled at run time (hipRTC)
namespace nqe {
void jit_wait(nqe_ctx *ctx) { jit_wait_all(ctx); }
namespace {

template <int OP, int DT> struct OpTag { static constexpr int op = OP, dt = DT; };
// wave-uniform (op, dtype) → compile-time constants.  Boolean operands compare like UInt64 words (0/1).
template <class F> __device__ __forceinline__ void dispatch_binary(int op, int dt, F &&f) {
#define NQE_DISPATCH_OP(O)                                                                                                       \
    case O:                                                                                                                      \
        if (dt == NQE_INT64) f(OpTag<O, NQE_INT64>{});                                                                           \
        else if (dt == NQE_FLOAT64) f(OpTag<O, NQE_FLOAT64>{});                                                                  \
        else f(OpTag<O, NQE_UINT64>{});                                                                                          \
        break;
    switch (op) {
        NQE_DISPATCH_OP(NQE_OP_EQ) NQE_DISPATCH_OP(NQE_OP_NOT_EQ) NQE_DISPATCH_OP(NQE_OP_LT) NQE_DISPATCH_OP(NQE_OP_LT_EQ)
        NQE_DISPATCH_OP(NQE_OP_GT) NQE_DISPATCH_OP(NQE_OP_GT_EQ) NQE_DISPATCH_OP(NQE_OP_PLUS) NQE_DISPATCH_OP(NQE_OP_MINUS)
        NQE_DISPATCH_OP(NQE_OP_MULTIPLY) NQE_DISPATCH_OP(NQE_OP_DIVIDE)
    default: // NQE_OP_MODULOS
        if (dt == NQE_INT64) f(OpTag<NQE_OP_MODULOS, NQE_INT64>{});
        else if (dt == NQE_FLOAT64) f(OpTag<NQE_OP_MODULOS, NQE_FLOAT64>{});
        else f(OpTag<NQE_OP_MODULOS, NQE_UINT64>{});
        break;
    }
#undef NQE_DISPATCH_OP
}

__device__ __forceinline__ void ex_combine(const ExInstr &in, uint64_t &a, bool &av, uint64_t b, bool bv, int *flags) {
    if (in.op == NQE_OP_AND || in.op == NQE_OP_OR) { // and_kleene / or_kleene
        bool lb = av && a, rb = bv && b, ok, r;
        if (in.op == NQE_OP_AND) { ok = (av && bv) || (av && !lb) || (bv && !rb); r = ok && lb && rb; }
        else { ok = (av && bv) || lb || rb; r = ok && (lb || rb); }
        a = r ? 1ull : 0ull;
        av = ok;
    } else {
        bool ok = av && bv;
        a = apply_binary(in.op, in.dt, a, b, in.aux, ok, flags);
        av = ok;
    }
}

// Loads the EX_ROWS rows a lane owns (row0 + r*64) of every program column; `inm` = rows that exist / are wanted.
// NULLS = false: no column has a validity bitmap and no literal is NULL, so every mask equals `inm` and none is computed
// (the kernel is VALU-issue bound once the program has a few instructions; mask bookkeeping is ~40% of it).
template <bool NULLS, int NC, int R = EX_ROWS>
__device__ __forceinline__ void ex_load(const ExProgram &P, int64_t row0, int64_t n, uint32_t inm, uint64_t (&cw)[NC][R], uint32_t (&cvm)[NC]) {
    // issue every load of the chunk (rows clamped to n-1 so that no load is predicated), then consume
    int64_t rc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rc[r] = min(row0 + r * 64, n - 1);
    uint32_t vbyte[NC][R];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 0; r < R; ++r) { cw[c][r] = 0; vbyte[c][r] = 0xffu; }
        if (c < P.ncols) {
            if (P.col_dtype[c] == NQE_BOOLEAN) {
#pragma unroll
                for (int r = 0; r < R; ++r) cw[c][r] = static_cast<const uint8_t *>(P.col_values[c])[rc[r] >> 3];
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) cw[c][r] = __builtin_nontemporal_load(static_cast<const uint64_t *>(P.col_values[c]) + rc[r]);
            }
            if (NULLS && P.col_valid[c]) {
#pragma unroll
                for (int r = 0; r < R; ++r) vbyte[c][r] = P.col_valid[c][rc[r] >> 3];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cvm[c] = inm;
        if (c < P.ncols) {
            if (P.col_dtype[c] == NQE_BOOLEAN) {
#pragma unroll
                for (int r = 0; r < R; ++r) cw[c][r] = (cw[c][r] >> (int(rc[r]) & 7)) & 1ull;
            }
            if (NULLS && P.col_valid[c]) {
                uint32_t m = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) m |= ((vbyte[c][r] >> (int(rc[r]) & 7)) & 1u) << r;
                cvm[c] = m & inm;
            }
        }
    }
}

// Runs the program on the loaded rows; the result words are left in res[], the returned mask holds their validity.
template <bool NULLS, int NC, int R = EX_ROWS>
__device__ __forceinline__ uint32_t ex_run(const ExProgram &P, const uint64_t (&cw)[NC][R], const uint32_t (&cvm)[NC],
                                           uint32_t inm, uint32_t litm, uint64_t (&res)[R], int *flags) {
    // ---- run the program
    uint64_t s[EX_MAX_DEPTH][R];
    uint32_t vm[EX_MAX_DEPTH];
#pragma unroll
    for (int d = 0; d < EX_MAX_DEPTH; ++d) {
        vm[d] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) s[d][r] = 0;
    }
    ExInstr cur = P.ins[0];
    for (int pc = 0; pc < P.n; ++pc) {
        const ExInstr nxt = P.ins[pc + 1 < P.n ? pc + 1 : pc]; // in flight while `cur` executes
        const bool a_st = cur.a_src == EX_STACK, b_st = cur.b_src == EX_STACK;
        const int ac = cur.a_src - EX_COL, bc = cur.b_src - EX_COL;
        // Operand fetch and stack update are wave-uniform BRANCHES around plain register moves: a select costs VALU
        // issue slots per row, a scalar branch does not, and the budget to stay HBM-bound is ~130 VALU instructions per
        // 64 rows for the whole program.  Everything is copied by value with constant indices (a conditional over array
        // lvalues would turn the stack into a dynamically indexed private array, i.e. scratch memory).
        uint64_t a[R], b[R];
        uint32_t am, bm;
        if (a_st) {
            if (b_st) {
                am = vm[1];
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = s[1][r];
            } else {
                am = vm[0];
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = s[0][r];
            }
        } else if (ac < 0) {
            am = cur.a_src == EX_LIT ? litm : 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cur.lit_a;
        } else if (ac == 0) {
            am = cvm[0];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cw[0][r];
        } else if (NC <= 2 || ac == 1) {
            am = cvm[1];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cw[1][r];
        } else if (ac == 2) {
            am = cvm[NC > 2 ? 2 : 0];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cw[NC > 2 ? 2 : 0][r];
        } else {
            am = cvm[NC > 2 ? 3 : 0];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cw[NC > 2 ? 3 : 0][r];
        }
        if (b_st) {
            bm = vm[0];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = s[0][r];
        } else if (bc < 0) {
            bm = cur.b_src == EX_LIT ? litm : 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = cur.lit_b;
        } else if (bc == 0) {
            bm = cvm[0];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = cw[0][r];
        } else if (NC <= 2 || bc == 1) {
            bm = cvm[1];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = cw[1][r];
        } else if (bc == 2) {
            bm = cvm[NC > 2 ? 2 : 0];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = cw[NC > 2 ? 2 : 0][r];
        } else {
            bm = cvm[NC > 2 ? 3 : 0];
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = cw[NC > 2 ? 3 : 0][r];
        }
        uint32_t m;
        if (cur.op == NQE_OP_AND || cur.op == NQE_OP_OR) {
            if (NULLS) { // and_kleene / or_kleene
                m = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bool av = (am >> r) & 1u;
                    ex_combine(cur, a[r], av, b[r], (bm >> r) & 1u, flags);
                    m |= (av ? 1u : 0u) << r;
                }
            } else {
                m = inm;
                if (cur.op == NQE_OP_AND) {
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] &= b[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] |= b[r];
                }
            }
        } else {
            m = NULLS ? (am & bm) : inm;
            // one uniform op/dtype decision per instruction (not per row): the body is instantiated with constants
            dispatch_binary(cur.op, cur.dt, [&](auto tag) {
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = apply_binary(tag.op, tag.dt, a[r], b[r], cur.aux, (m >> r) & 1u, flags);
            });
        }
        if (a_st && b_st) { // pop 2, push 1
            vm[1] = vm[2];
#pragma unroll
            for (int r = 0; r < R; ++r) s[1][r] = s[2][r];
        } else if (!a_st && !b_st) { // push
            vm[2] = vm[1];
            vm[1] = vm[0];
#pragma unroll
            for (int r = 0; r < R; ++r) { s[2][r] = s[1][r]; s[1][r] = s[0][r]; }
        }
        vm[0] = m;
#pragma unroll
        for (int r = 0; r < R; ++r) s[0][r] = a[r];
        cur = nxt;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) res[r] = s[0][r];
    return vm[0];
}

// R rows per lane: the dispatch of an instruction (scalar work) is paid once per R x 64 rows; 4 by default (8 halves the scalar work
// but takes 176 VGPRs — see the launch site)
template <bool NULLS, int NC, int R = EX_ROWS>
__global__ void __launch_bounds__(256) expr_tree_kernel(ExProgram P, int64_t n, uint64_t *out_words, uint64_t *out_bits, uint64_t *out_valid,
                                                        int *flags) {
    const int lane = lane_id();
    const int64_t n_chunks = (n + 64 * R - 1) / (64 * R);
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t row0 = chunk * (64 * R) + lane;
        uint32_t inm = 0; // one bit per owned row
#pragma unroll
        for (int r = 0; r < R; ++r) inm |= (row0 + r * 64 < n ? 1u : 0u) << r;
        uint64_t cw[NC][R], res[R];
        uint32_t cvm[NC];
        ex_load<NULLS, NC, R>(P, row0, n, inm, cw, cvm);
        const uint32_t vm = ex_run<NULLS, NC, R>(P, cw, cvm, inm, inm, res, flags);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * 64;
            const bool ok = (vm >> r) & 1u;
            if (row - lane >= n) break; // wave-uniform: this 64-row word is past the end
            if (out_words) {
                if (row < n) __builtin_nontemporal_store(ok ? res[r] : 0ull, out_words + row);
            } else {
                uint64_t w = __ballot(ok && res[r]);
                if (lane == 0) out_bits[row >> 6] = w;
            }
            if (out_valid) {
                uint64_t w = __ballot(ok);
                if (lane == 0) out_valid[row >> 6] = w;
            }
        }
    }
}

// The same machine behind a selection: one wave per 4096-row tile of the keep bitmap (word k of the tile in lane k, as in
// compact_kernel); only the rows the filter emits are evaluated as valid (a dropped row can never raise DivideByZero,
// as in the reference where the projection runs on the filtered batch), 256-row chunks without any kept row are not even
// loaded, and results go straight to their compacted position.  A NULL predicate emits a NULL row (quirk Q4).
template <bool NULLS, int NC>
__global__ void __launch_bounds__(256) expr_tree_compact_kernel(ExProgram P, const uint64_t *keep, const uint64_t *pvalid,
                                                                const uint64_t *tile_offsets, int64_t n, int64_t ntiles, uint64_t *out_words,
                                                                uint8_t *out_bool_bytes, uint8_t *out_valid_bytes, int *flags) {
    constexpr int R = EX_ROWS;
    const int lane = lane_id();
    const int waves_per_block = blockDim.x / 64;
    const int64_t nwords = (n + 63) / 64;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles; tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t w = tile * TILE_WORDS + lane;
        const uint64_t my_word = w < nwords ? keep[w] : 0;
        const uint64_t my_pv = (pvalid && w < nwords) ? pvalid[w] : ~0ull;
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        if (tot == 0) continue;
        const uint64_t base = tile_offsets[tile];
        for (int k0 = 0; k0 < TILE_WORDS; k0 += R) {
            uint64_t kw[R];
            // inm: emitted rows whose predicate was valid (column values count); litm: every emitted row — a row emitted for
            // a NULL predicate is all-NULL in the reference's filtered batch, but literals are still valid there
            // (NULL OR true = true), found by the differential fuzzer
            uint32_t inm = 0, litm = 0, anyk = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                kw[r] = bcast64(my_word, k0 + r);
                anyk |= kw[r] != 0 ? 1u : 0u;
                litm |= uint32_t((kw[r] >> lane) & 1ull) << r;
                inm |= uint32_t(((kw[r] & bcast64(my_pv, k0 + r)) >> lane) & 1ull) << r;
            }
            if (!anyk) continue; // wave-uniform
            const int64_t row0 = (tile * TILE_WORDS + k0) * 64 + lane;
            uint64_t cw[NC][R], res[R];
            uint32_t cvm[NC];
            ex_load<NULLS, NC>(P, row0, n, inm, cw, cvm);
            const uint32_t vm = ex_run<NULLS, NC>(P, cw, cvm, inm, litm, res, flags);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((kw[r] >> lane) & 1ull) {
                    const bool ok = (vm >> r) & 1u;
                    const uint64_t pos = base + bcast32(my_off, k0 + r) + __popcll(kw[r] & lanemask_lt());
                    if (out_words) __builtin_nontemporal_store(ok ? res[r] : 0ull, out_words + pos);
                    if (out_bool_bytes) out_bool_bytes[pos] = (ok && res[r]) ? 1 : 0;
                    if (out_valid_bytes) out_valid_bytes[pos] = ok ? 1 : 0;
                }
            }
        }
    }
}

// builds the stack program; false when the tree does not fit the machine (then: node-at-a-time)
bool build_program(const nqe_table *in, const std::vector<Node> &t, int root, ExProgram *P, bool *needs_valid) {
    std::memset(P, 0, sizeof(*P));
    std::vector<int> order; // BINARY nodes, post-order
    std::vector<std::pair<int, bool>> st = {{root, false}};
    while (!st.empty()) {
        auto [i, done] = st.back();
        st.pop_back();
        const Node &x = t[size_t(i)];
        if (x.kind != NQE_EXPR_BINARY) continue;
        if (done) { order.push_back(i); continue; }
        st.push_back({i, true});
        st.push_back({x.right, false});
        st.push_back({x.left, false});
    }
    if (int(order.size()) > EX_MAX_INSTR || order.empty()) return false;
    *needs_valid = false;
    bool fits = true;
    auto operand = [&](int idx, int32_t *src, uint64_t *lit) {
        const Node &x = t[size_t(idx)];
        if (x.kind == NQE_EXPR_BINARY) { *src = EX_STACK; return; }
        if (x.kind == NQE_EXPR_LITERAL) {
            if (x.dtype == NQE_UTF8) { fits = false; return; }
            *src = x.lit_null ? EX_LIT_NULL : EX_LIT;
            *lit = x.lit;
            *needs_valid |= x.lit_null;
            return;
        }
        const DevColumn &c = in->cols[size_t(x.column)];
        if (!(is_word_type(c.dtype) || c.dtype == NQE_BOOLEAN)) { fits = false; return; }
        int slot = -1;
        for (int k = 0; k < P->ncols; ++k)
            if (P->col_values[k] == c.values->ptr && P->col_dtype[k] == c.dtype && P->col_valid[k] == c.valid()) slot = k;
        if (slot < 0) {
            if (P->ncols == EX_MAX_COLS) { fits = false; return; }
            slot = P->ncols++;
            P->col_values[slot] = c.values->ptr;
            P->col_valid[slot] = c.valid();
            P->col_dtype[slot] = c.dtype;
        }
        *src = EX_COL + slot;
        *needs_valid |= c.validity != nullptr;
    };
    int depth = 0;
    for (int i : order) {
        const Node &x = t[size_t(i)];
        ExInstr &I = P->ins[P->n++];
        I.op = x.op;
        I.dt = t[size_t(x.left)].out_dtype;
        I.aux.pow2_shift = I.aux.more = -1;
        operand(x.left, &I.a_src, &I.lit_a);
        operand(x.right, &I.b_src, &I.lit_b);
        if (!fits) return false;
        if (I.b_src == EX_LIT) I.aux = make_aux(x.op, I.dt, I.lit_b);
        depth += 1 - int(I.a_src == EX_STACK) - int(I.b_src == EX_STACK);
        if (depth > EX_MAX_DEPTH) return false;
    }
    return true;
}

struct Value {
    bool is_lit = false;
    bool lit_null = false;
    uint64_t lit = 0;
    std::string lit_str;
    int dtype = NQE_NULLTYPE;
    DevColumn col;
};

DevColumn materialise_literal(nqe_ctx *ctx, int dtype, uint64_t lit, bool lit_null, int64_t n) {
    // ScalarValue::into_array (logical_plan/expression.rs:210-222)
    if (dtype == NQE_NULLTYPE) fail(NQE_ERR_NOT_SUPPORTED, "Null-typed arrays are not supported on the device path");
    DevColumn c;
    if (dtype == NQE_BOOLEAN) {
        c = make_bool_column(ctx, n, lit_null);
        NQE_HIP_CHECK(hipMemsetAsync(c.values->ptr, (!lit_null && lit) ? 0xFF : 0x00, bitmap_alloc_bytes(n), ctx->stream));
    } else {
        c = make_word_column(ctx, dtype, n, lit_null);
        if (n)
            launch(ctx, "fill_words", fill_words_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                   (uint64_t *)c.values->ptr, lit_null ? 0ull : lit, n);
    }
    if (lit_null) {
        NQE_HIP_CHECK(hipMemsetAsync(c.validity->ptr, 0, bitmap_alloc_bytes(n), ctx->stream));
        c.null_count = n;
    }
    return c;
}

Operand operand_of(const Value &v) {
    Operand o;
    o.values = v.is_lit ? nullptr : v.col.values->ptr;
    o.valid = v.is_lit ? nullptr : v.col.valid();
    o.lit = v.lit;
    o.is_lit = v.is_lit;
    o.lit_null = v.lit_null;
    return o;
}

Value eval_node(nqe_ctx *ctx, const nqe_table *in, const std::vector<Node> &t, int i) {
    const Node &x = t[size_t(i)];
    Value v;
    v.dtype = x.out_dtype;
    if (x.kind == NQE_EXPR_COLUMN) {
        v.col = in->cols[size_t(x.column)]; // Arc clone (column.rs:41-43)
        return v;
    }
    if (x.kind == NQE_EXPR_LITERAL) {
        v.is_lit = true;
        v.lit = x.lit;
        v.lit_str = x.lit_str;
        v.lit_null = x.lit_null;
        return v;
    }
    Value l = eval_node(ctx, in, t, x.left);
    Value r = eval_node(ctx, in, t, x.right);
    const int64_t n = in->rows;
    if (l.dtype == NQE_UTF8) { // only compares reach here (parse)
        v.col = utf8_compare(ctx, x.op, l.is_lit ? nullptr : &l.col, l.lit_str, l.lit_null, r.is_lit ? nullptr : &r.col, r.lit_str, r.lit_null, n);
        return v;
    }
    if (l.is_lit && r.is_lit) { // lit op lit: materialise one side, as into_array would
        l.col = materialise_literal(ctx, l.dtype, l.lit, l.lit_null, n);
        l.is_lit = false;
    }
    const int dt = l.dtype;
    bool need_valid = (l.is_lit ? l.lit_null : l.col.validity != nullptr) || (r.is_lit ? r.lit_null : r.col.validity != nullptr);
    bool bool_out = x.out_dtype == NQE_BOOLEAN;
    v.col = bool_out ? make_bool_column(ctx, n, need_valid) : make_word_column(ctx, x.out_dtype, n, need_valid);
    OpAux aux = r.is_lit && !r.lit_null ? make_aux(x.op, dt, r.lit) : make_aux(0, 0, 0);
    if (n)
        launch(ctx, "expr_binary", binary_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, operand_of(l), operand_of(r),
               x.op, dt, aux, n, bool_out ? nullptr : (uint64_t *)v.col.values->ptr,
               bool_out ? (uint64_t *)v.col.values->ptr : nullptr, need_valid ? (uint64_t *)v.col.validity->ptr : nullptr,
               ctx->d_flags);
    return v;
}

} // namespace

// `x op lit` (x Int64/UInt64) → range test. Returns false if the shape is not covered.
bool make_fast_pred(const SimpleExpr &pe, FastPred *fp) {
    if (pe.nops != 1 || pe.op[0] > NQE_OP_GT_EQ) return false;
    if (pe.src_dtype != NQE_INT64 && pe.src_dtype != NQE_UINT64 && pe.src_dtype != NQE_FLOAT64) return false;
    static const int flip_op[6] = {NQE_OP_EQ, NQE_OP_NOT_EQ, NQE_OP_GT, NQE_OP_GT_EQ, NQE_OP_LT, NQE_OP_LT_EQ};
    int op = pe.lit_left[0] ? flip_op[pe.op[0]] : pe.op[0]; // lit op x  ≡  x op' lit
    const int64_t MIN = INT64_MIN, MAX = INT64_MAX;
    fp->negate = 0;
    fp->pad = 0;
    fp->row_shift = 0;
    fp->bit_mask = 0;
    fp->val_mask = ~0ull;
    fp->fmask = 0;
    if (pe.src_dtype == NQE_FLOAT64) {
        // IEEE compares as an integer range over the order-preserving image ord(x) = x ^ ((x >> 63) & 0x7fff…f) (signed):
        // every NaN maps beyond ord(±inf), so a range inside [ord(-inf), ord(+inf)] is false for NaN, and the negated
        // range (!=) is true for NaN — exactly arrow's lt/gt/eq/neq on Float64.  ±0 compare equal: the bound uses
        // whichever zero makes the range include / exclude both.
        fp->flip = 0;
        fp->fmask = 0x7fffffffffffffffull;
        auto ord = [](double d) {
            uint64_t b;
            std::memcpy(&b, &d, 8);
            return int64_t(b ^ (uint64_t(int64_t(b) >> 63) & 0x7fffffffffffffffull));
        };
        double c;
        std::memcpy(&c, &pe.lit[0], 8);
        const int64_t NINF = ord(-HUGE_VAL), PINF = ord(HUGE_VAL);
        if (c != c) { // NaN literal: every compare is false, != is true
            fp->lo = 1; fp->hi = 0;
            fp->negate = op == NQE_OP_NOT_EQ ? 1 : 0;
            return true;
        }
        const int64_t c_lo = ord(c == 0.0 ? -0.0 : c), c_hi = ord(c == 0.0 ? 0.0 : c); // image of {x : x == c}
        switch (op) {
        case NQE_OP_EQ: fp->lo = c_lo; fp->hi = c_hi; break;
        case NQE_OP_NOT_EQ: fp->lo = c_lo; fp->hi = c_hi; fp->negate = 1; break;
        case NQE_OP_LT: fp->lo = NINF; fp->hi = c_lo - 1; break;   // c = -inf: empty (hi < lo)
        case NQE_OP_LT_EQ: fp->lo = NINF; fp->hi = c_hi; break;
        case NQE_OP_GT: fp->lo = c_hi + 1; fp->hi = PINF; break;   // c = +inf: empty
        default: fp->lo = c_lo; fp->hi = PINF; break;
        }
        return true;
    }
    fp->flip = pe.src_dtype == NQE_UINT64 ? 0x8000000000000000ull : 0ull;
    const int64_t L = int64_t(pe.lit[0] ^ fp->flip);
    switch (op) {
    case NQE_OP_EQ: fp->lo = L; fp->hi = L; break;
    case NQE_OP_NOT_EQ: fp->lo = L; fp->hi = L; fp->negate = 1; break;
    case NQE_OP_LT: fp->lo = MIN; fp->hi = L - 1; if (L == MIN) { fp->lo = 1; fp->hi = 0; } break; // empty
    case NQE_OP_LT_EQ: fp->lo = MIN; fp->hi = L; break;
    case NQE_OP_GT: fp->lo = L + 1; fp->hi = MAX; if (L == MAX) { fp->lo = 1; fp->hi = 0; } break;
    default: fp->lo = L; fp->hi = MAX; break;
    }
    return true;
}

// Any nesting of `and` / `or` over ONE to CONJ_MAX tests, a test being `col cmp lit` (either side) or `(col arith lit) cmp lit` with a
// fault-free arithmetic step (`id % 3 = 0`, `v * 2.0 > 100.0`, `100 - w >= 7`), over non-null Int64/UInt64/Float64 columns (cols[]
// names the tested column of every test; which loaded word of a row serves a test — ConjTest::src — is the consumer's business).
// A pure and-list / or-list of plain range tests keeps the straight-line form (general = 0); everything else is evaluated through
// the truth table of the and/or structure.  Nodes are in postfix order.
bool match_conj(const nqe_table *in, const nqe_expr_node *nodes, int n, ConjPred *out, int *cols) {
    constexpr int MAXN = 8 * CONJ_MAX;
    if (n < 3 || n > MAXN || nodes[n - 1].kind != NQE_EXPR_BINARY) return false;
    // first node of the subtree that ends at node i
    int start[MAXN], stack[MAXN], sp = 0;
    for (int i = 0; i < n; ++i) {
        if (nodes[i].kind == NQE_EXPR_BINARY) {
            if (sp < 2) return false;
            sp -= 2;
            start[i] = stack[sp];
        } else
            start[i] = i;
        stack[sp++] = start[i];
    }
    if (sp != 1) return false;
    std::memset(out, 0, sizeof(*out));
    // leaves = maximal subtrees that are not and/or nodes, left to right
    int leaf_of[MAXN]; // node -> leaf number when the node is a leaf's root
    int leaves[CONJ_MAX], nl = 0;
    bool plain_list = true;
    const int root_op = nodes[n - 1].op;
    {
        int todo[MAXN], nt = 0;
        todo[nt++] = n - 1;
        int rev[CONJ_MAX], nr = 0;
        while (nt) {
            const int i = todo[--nt];
            if (nodes[i].kind == NQE_EXPR_BINARY && (nodes[i].op == NQE_OP_AND || nodes[i].op == NQE_OP_OR)) {
                if (nodes[i].op != root_op) plain_list = false;
                todo[nt++] = start[i - 1] - 1; // left operand's root (examined after the right one: leaves come out right to left)
                todo[nt++] = i - 1;
            } else {
                if (nr == CONJ_MAX) return false;
                rev[nr++] = i;
            }
        }
        for (int k = 0; k < nr; ++k) leaves[nl++] = rev[nr - 1 - k];
    }
    if (nl < 1) return false;
    if (root_op != NQE_OP_AND && root_op != NQE_OP_OR) plain_list = false; // a single test (with an arithmetic step, or it would be a SimpleExpr)
    bool any_pre = false;
    for (int t = 0; t < nl; ++t) {
        const int i = leaves[t];
        leaf_of[i] = t;
        const nqe_expr_node *leaf = nodes + start[i];
        const int len = i - start[i] + 1;
        if (len != 3 && len != 5) return false;
        ExprInfo li;
        try {
            li = analyze_expr(in, leaf, len);
        } catch (...) {
            return false; // whatever the leaf's problem is, the tree as a whole reports it
        }
        if (!li.simple || li.out_dtype != NQE_BOOLEAN || li.may_fault) return false;
        const DevColumn &c = in->cols[size_t(li.s.col)];
        if (!is_word_type(c.dtype) || c.validity || !c.values) return false;
        ConjTest &T = out->t[t];
        SimpleExpr cmp = li.s; // the comparison alone, over the type it compares
        if (li.s.nops == 2) {
            const int op = li.s.op[0], dt = li.s.op_dtype[0];
            if (op < NQE_OP_PLUS || op > NQE_OP_MODULOS) return false;
            if (dt == NQE_FLOAT64 && op == NQE_OP_MODULOS) return false;
            if ((op == NQE_OP_DIVIDE || op == NQE_OP_MODULOS) && li.s.lit_left[0]) return false; // (analyze_expr: may_fault — kept explicit)
            T.pre = op;
            T.pre_dt = dt;
            T.pre_rev = li.s.lit_left[0];
            T.pre_lit = li.s.lit[0];
            T.pre_aux = li.s.aux[0];
            any_pre = true;
            cmp.nops = 1;
            cmp.op[0] = li.s.op[1];
            cmp.lit_left[0] = li.s.lit_left[1];
            cmp.op_dtype[0] = li.s.op_dtype[1];
            cmp.lit[0] = li.s.lit[1];
            cmp.src_dtype = li.s.op_dtype[1];
        } else if (li.s.nops != 1)
            return false;
        FastPred fp{};
        if (!make_fast_pred(cmp, &fp)) return false;
        cols[t] = li.s.col;
        T.lo = fp.lo;
        T.hi = fp.hi;
        T.flip = fp.flip;
        T.fmask = fp.fmask;
        T.negate = fp.negate;
    }
    out->n = nl;
    if (plain_list && !any_pre && nl >= 2) {
        out->is_or = root_op == NQE_OP_OR ? 1 : 0;
        return true;
    }
    if (nl == 1 && !any_pre) return false; // a bare compare: the SimpleExpr paths are leaner
    // truth table: evaluate the and/or structure for every assignment of the tests
    out->general = 1;
    for (uint32_t asg = 0; asg < (1u << nl); ++asg) {
        bool val[MAXN];
        int vs = 0;
        // postfix walk over the and/or skeleton: a leaf's subtree contributes its assigned value at its root
        for (int i = 0; i < n; ++i) {
            bool is_leaf_root = false;
            for (int t = 0; t < nl; ++t) is_leaf_root = is_leaf_root || leaves[t] == i;
            if (is_leaf_root) {
                val[vs++] = (asg >> leaf_of[i]) & 1u;
                continue;
            }
            bool inside = false; // a node strictly inside some leaf's subtree
            for (int t = 0; t < nl; ++t) inside = inside || (i >= start[leaves[t]] && i < leaves[t]);
            if (inside) continue;
            // an and/or node of the skeleton
            const bool rv = val[--vs], lv = val[--vs];
            val[vs++] = nodes[i].op == NQE_OP_AND ? (lv && rv) : (lv || rv);
        }
        if (val[0]) out->truth |= 1u << asg;
    }
    return true;
}

bool match_tree_pred(const nqe_table *in, const nqe_expr_node *nodes, int n, TreePred *out) {
    std::memset(out, 0, sizeof(*out));
    int root;
    std::vector<Node> t;
    try {
        t = parse(in, nodes, n, &root);
    } catch (...) {
        return false; // the operator's own analysis reports the problem
    }
    if (t[size_t(root)].out_dtype != NQE_BOOLEAN || t[size_t(root)].kind != NQE_EXPR_BINARY) return false;
    ExProgram P;
    bool needs_valid = false;
    if (!build_program(in, t, root, &P, &needs_valid) || needs_valid) return false;
    if (P.n > TREE_MAX_INSTR || P.ncols > TREE_MAX_COLS) return false;
    // typed stacks (see tree_pred_eval): one VALUE register, three BOOLEAN levels; normal forms x ∈ {stack, word}, y ∈ {literal,
    // stack, word}, lit_a = 1: the operands are swapped back before the operation
    int vdepth = 0, bdepth = 0;
    for (int i = 0; i < P.n; ++i) {
        const ExInstr &I = P.ins[i];
        if (I.a_src == EX_LIT_NULL || I.b_src == EX_LIT_NULL) return false;
        TreeInstr &T = out->ins[i];
        T.op = I.op;
        T.dt = I.dt;
        T.lit_a = 0;
        T.lit_b = 0;
        T.aux = I.aux;
        auto src = [](int s) { return s >= EX_COL ? int(TS_W0) + (s - EX_COL) : (s == EX_LIT ? int(TS_LIT) : int(TS_STACK)); };
        if (I.op == NQE_OP_AND || I.op == NQE_OP_OR) {
            if (I.a_src != EX_STACK || I.b_src != EX_STACK) return false; // a Boolean literal or column as an operand: not this machine's
            if (bdepth < 2) return false;
            --bdepth;
            T.a_src = T.b_src = TS_STACK;
            continue;
        }
        if (I.dt == NQE_BOOLEAN) return false; // comparing Booleans
        if (I.a_src == EX_LIT && I.b_src == EX_LIT) return false;
        int xs = src(I.a_src), ys = src(I.b_src);
        uint64_t lit = I.lit_b;
        bool rev = false;
        if (xs == TS_LIT) { // literal on the left: swap, and remember it for the operators that are not commutative
            xs = ys;
            ys = TS_LIT;
            lit = I.lit_a;
            rev = I.op != NQE_OP_PLUS && I.op != NQE_OP_MULTIPLY && I.op != NQE_OP_EQ && I.op != NQE_OP_NOT_EQ;
            T.aux.pow2_shift = T.aux.more = -1;
        }
        if (I.op == NQE_OP_DIVIDE || I.op == NQE_OP_MODULOS) {
            // no fault may be possible: the kernel has no flag path for the predicate
            if (ys != TS_LIT || rev) return false;
            if (I.dt == NQE_FLOAT64 ? (lit == 0 || lit == 0x8000000000000000ull) : (lit == 0 || lit == ~0ull)) return false;
        }
        const int pops = int(xs == TS_STACK) + int(ys == TS_STACK);
        if (pops > vdepth) return false;
        vdepth -= pops;
        if (I.op <= NQE_OP_GT_EQ) {
            if (++bdepth > 3) return false;
        } else if (++vdepth > 1)
            return false; // two arithmetic subtrees alive at once
        T.a_src = xs;
        T.b_src = ys;
        T.lit_a = rev ? 1 : 0;
        T.lit_b = lit;
    }
    if (vdepth != 0 || bdepth != 1) return false;
    for (int c = 0; c < P.ncols; ++c) {
        if (!is_word_type(P.col_dtype[c])) return false;
        out->col[c] = -1;
        for (size_t k = 0; k < in->cols.size(); ++k)
            if (in->cols[k].values && in->cols[k].values->ptr == P.col_values[c] && in->cols[k].dtype == P.col_dtype[c] && !in->cols[k].validity) out->col[c] = int(k);
        if (out->col[c] < 0) return false;
    }
    out->n = P.n;
    out->ncols = P.ncols;
    return true;
}

FastPred bitmap_fast_pred() {
    FastPred fp{};
    fp.lo = fp.hi = 1;
    fp.row_shift = 6;
    fp.bit_mask = 63;
    fp.val_mask = 1;
    fp.fmask = 0;
    return fp;
}

ExprInfo analyze_expr(const nqe_table *in, const nqe_expr_node *nodes, int n) {
    int root;
    std::vector<Node> t = parse(in, nodes, n, &root);
    ExprInfo info;
    info.out_dtype = t[size_t(root)].out_dtype;
    info.simple = match_simple(t, root, &info.s);
    for (const Node &x : t) {
        if (x.kind != NQE_EXPR_BINARY || (x.op != NQE_OP_DIVIDE && x.op != NQE_OP_MODULOS)) continue;
        const Node &r = t[size_t(x.right)];
        const bool safe_literal = r.kind == NQE_EXPR_LITERAL && !r.lit_null &&
                                  (r.dtype == NQE_FLOAT64 ? r.lit != 0 && r.lit != 0x8000000000000000ull : r.lit != 0 && r.lit != ~0ull);
        if (!safe_literal) info.may_fault = true;
    }
    return info;
}

DevColumn evaluate_expr(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, int n) {
    int root;
    std::vector<Node> t = parse(in, nodes, n, &root);
    ExProgram P;
    bool needs_valid = false;
    if (t[size_t(root)].kind == NQE_EXPR_BINARY && build_program(in, t, root, &P, &needs_valid)) {
        const int64_t rows = in->rows;
        const int odt = t[size_t(root)].out_dtype;
        const bool bool_out = odt == NQE_BOOLEAN;
        DevColumn out = bool_out ? make_bool_column(ctx, rows, needs_valid) : make_word_column(ctx, odt, rows, needs_valid);
        if (rows) {
            dim3 grid(stream_grid(ctx, (rows + EX_ROWS - 1) / EX_ROWS, 256));
            uint64_t *ow = bool_out ? nullptr : (uint64_t *)out.values->ptr, *ob = bool_out ? (uint64_t *)out.values->ptr : nullptr;
            uint64_t *ov = needs_valid ? (uint64_t *)out.validity->ptr : nullptr;
            // instantiated per (nullable, <=2 / <=4 columns): the column registers of a lane are the largest block of VGPRs
#define NQE_TREE(NU, NC) launch(ctx, "expr_tree", expr_tree_kernel<NU, NC>, grid, dim3(256), 0, P, rows, ow, ob, ov, ctx->d_flags)
            // (8 rows per lane measured 176 VGPRs = 2 waves per SIMD — an 8-operator chain 1.19 -> 1.13 ms per 2x10^8 rows, but
            // `(id % 1000) * 3 + id / 7` 1.11 -> 1.36 and `v > 50 and id % 3 = 0` 0.98 -> 1.19: four rows per lane it is)
            // three or more steps over a large input: the run-time specialised form of this very program, once it has been compiled
            if (jit_expr_tree(ctx, P, needs_valid, rows, ow, ob, ov)) {
            } else if (needs_valid) { if (P.ncols <= 2) NQE_TREE(true, 2); else NQE_TREE(true, 4); }
            else { if (P.ncols <= 2) NQE_TREE(false, 2); else NQE_TREE(false, 4); }
#undef NQE_TREE
        }
        return out;
    }
    Value v = eval_node(ctx, in, t, root);
    if (v.is_lit && v.dtype == NQE_UTF8) return utf8_literal_column(ctx, v.lit_str, v.lit_null, in->rows);
    if (v.is_lit) return materialise_literal(ctx, v.dtype, v.lit, v.lit_null, in->rows);
    return v.col;
}

// `e` over the rows a selection emits, written compacted (one pass over the referenced columns).  Returns false when the
// tree does not fit the stack machine (caller: compact the inputs, then evaluate_expr).
bool evaluate_expr_compacted(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, int n, const KeepMask &km, DevColumn *result) {
    int root;
    std::vector<Node> t = parse(in, nodes, n, &root);
    ExProgram P;
    bool needs_valid = false;
    if (t[size_t(root)].kind != NQE_EXPR_BINARY || !build_program(in, t, root, &P, &needs_valid)) return false;
    needs_valid |= km.pvalid != nullptr;
    const int64_t m = km.total;
    const int odt = t[size_t(root)].out_dtype;
    const bool bool_out = odt == NQE_BOOLEAN;
    DevColumn out = bool_out ? make_bool_column(ctx, m, needs_valid) : make_word_column(ctx, odt, m, needs_valid);
    BufRef bool_bytes, valid_bytes;
    if (bool_out) bool_bytes = dev_alloc(ctx, size_t(m) + 8);
    if (needs_valid) valid_bytes = dev_alloc(ctx, size_t(m) + 8);
    if (km.ntiles && m > 0) {
        dim3 grid(stream_grid(ctx, km.ntiles, 4));
        const uint64_t *kp = (const uint64_t *)km.keep->ptr, *pv = km.pvalid ? (const uint64_t *)km.pvalid->ptr : nullptr;
        const uint64_t *to = (const uint64_t *)km.tile_offsets->ptr;
        uint64_t *ow = bool_out ? nullptr : (uint64_t *)out.values->ptr;
        uint8_t *ob = bool_out ? (uint8_t *)bool_bytes->ptr : nullptr, *ov = needs_valid ? (uint8_t *)valid_bytes->ptr : nullptr;
#define NQE_TREE(NU, NC) launch(ctx, "expr_tree_compact", expr_tree_compact_kernel<NU, NC>, grid, dim3(256), 0, P, kp, pv, to, km.n, km.ntiles, ow, ob, ov, ctx->d_flags)
        if (needs_valid) { if (P.ncols <= 2) NQE_TREE(true, 2); else NQE_TREE(true, 4); }
        else { if (P.ncols <= 2) NQE_TREE(false, 2); else NQE_TREE(false, 4); }
#undef NQE_TREE
    }
    if (bool_out) pack_bytes_to_bits(ctx, (const uint8_t *)bool_bytes->ptr, m, (uint64_t *)out.values->ptr);
    if (needs_valid) pack_bytes_to_bits(ctx, (const uint8_t *)valid_bytes->ptr, m, (uint64_t *)out.validity->ptr);
    *result = out;
    return true;
}

bool project_specialised(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, const int32_t *expr_offsets, int num_exprs, const KeepMask &km,
                         std::vector<DevColumn> *result) {
    const char *mr = getenv("NQE_JIT_MIN_ROWS");
    const int64_t min_rows = mr ? atoll(mr) : (int64_t(1) << 22);
    if (getenv("NQE_NO_JIT") || num_exprs < 1 || num_exprs > JP_MAX_OUTS || km.n < min_rows || km.total <= 0) return false;
    JitProj J;
    std::memset(J.col_values, 0, sizeof(J.col_values));
    std::memset(J.col_valid, 0, sizeof(J.col_valid));
    std::memset(J.col_dtype, 0, sizeof(J.col_dtype));
    J.nulls = km.pvalid != nullptr;
    auto slot_of = [&](const void *values, const uint8_t *valid, int dtype) {
        for (int k = 0; k < J.ncols; ++k)
            if (J.col_values[k] == values && J.col_valid[k] == valid && J.col_dtype[k] == dtype) return k;
        if (J.ncols == JP_MAX_COLS) return -1;
        J.col_values[J.ncols] = values;
        J.col_valid[J.ncols] = valid;
        J.col_dtype[J.ncols] = dtype;
        return J.ncols++;
    };
    int steps = 0;
    for (int e = 0; e < num_exprs; ++e) {
        int root;
        std::vector<Node> t = parse(in, nodes + expr_offsets[e], expr_offsets[e + 1] - expr_offsets[e], &root);
        const Node &rt = t[size_t(root)];
        JitProjOut o;
        o.out_dtype = rt.out_dtype;
        if (rt.kind == NQE_EXPR_COLUMN) {
            const DevColumn &c = in->cols[size_t(rt.column)];
            if (!is_word_type(c.dtype) || c.length < km.n) return false; // (Boolean / Utf8 columns: their own compaction kernels)
            o.is_column = true;
            o.col = slot_of(c.values->ptr, c.valid(), c.dtype);
            if (o.col < 0) return false;
            o.needs_valid = c.validity != nullptr || km.pvalid != nullptr;
            J.nulls = J.nulls || c.validity != nullptr;
        } else if (rt.kind == NQE_EXPR_BINARY) {
            bool nv = false;
            if (!build_program(in, t, root, &o.P, &nv)) return false;
            for (int i = 0; i < o.P.n; ++i) {
                ExInstr &I = o.P.ins[i];
                for (int32_t *src : {&I.a_src, &I.b_src}) {
                    if (*src < EX_COL) continue;
                    const int k = *src - EX_COL, u = slot_of(o.P.col_values[k], o.P.col_valid[k], o.P.col_dtype[k]);
                    if (u < 0) return false;
                    *src = EX_COL + u;
                }
            }
            o.bool_out = rt.out_dtype == NQE_BOOLEAN;
            o.needs_valid = nv || km.pvalid != nullptr;
            J.nulls = J.nulls || nv;
            steps += o.P.n;
        } else
            return false; // a bare literal
        J.outs.push_back(o);
    }
    // bare columns and one- or two-step chains are at the memory system's rate in their own kernels (measured: two bare columns
    // 0.51 ms in two compact_column passes, 0.55 ms fused): the fused pass pays when it removes interpretation
    if (steps < 3) return false;
    // outputs
    const int64_t m = km.total;
    std::vector<DevColumn> cols;
    std::vector<BufRef> bool_bytes(static_cast<size_t>(num_exprs)), valid_bytes(static_cast<size_t>(num_exprs));
    uint64_t *ow[JP_MAX_OUTS] = {};
    uint8_t *ob[JP_MAX_OUTS] = {}, *ov[JP_MAX_OUTS] = {};
    for (int e = 0; e < num_exprs; ++e) {
        const JitProjOut &o = J.outs[size_t(e)];
        cols.push_back(o.bool_out ? make_bool_column(ctx, m, o.needs_valid) : make_word_column(ctx, o.out_dtype, m, o.needs_valid));
        if (o.bool_out) {
            bool_bytes[size_t(e)] = dev_alloc(ctx, size_t(m) + 8);
            ob[e] = (uint8_t *)bool_bytes[size_t(e)]->ptr;
        } else
            ow[e] = (uint64_t *)cols.back().values->ptr;
        if (o.needs_valid) {
            valid_bytes[size_t(e)] = dev_alloc(ctx, size_t(m) + 8);
            ov[e] = (uint8_t *)valid_bytes[size_t(e)]->ptr;
        }
    }
    if (!jit_project(ctx, J, km, ow, ob, ov)) return false;
    for (int e = 0; e < num_exprs; ++e) {
        if (ob[e]) pack_bytes_to_bits(ctx, ob[e], m, (uint64_t *)cols[size_t(e)].values->ptr);
        if (ov[e]) pack_bytes_to_bits(ctx, ov[e], m, (uint64_t *)cols[size_t(e)].validity->ptr);
    }
    *result = std::move(cols);
    return true;
}

// Fused ProjectionPlan(SelectionPlan(input)) in ONE pass (expr_jit.hpp: jit_select_project): a predicate TREE (three or more
// operators: the shapes whose mask the specialised kernel computes anyway) over columns without validity, a projection list of
// word-typed outputs over columns without validity, a large input.  Returns false when the shape does not qualify or the kernel is
// still being compiled — the caller then takes the mask + compaction path, whose results are identical.
bool select_project_fused(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *nodes, const int32_t *expr_offsets,
                          int num_exprs, std::vector<DevColumn> *result, int64_t *total_out) {
    const char *mr = getenv("NQE_JIT_MIN_ROWS");
    const int64_t min_rows = mr ? atoll(mr) : (int64_t(1) << 22);
    const int64_t n = in->rows;
    if (getenv("NQE_NO_JIT") || getenv("NQE_NO_FUSED_SELECT") || num_exprs < 1 || num_exprs > JP_MAX_OUTS || n < min_rows || !pred || pred_nodes < 1) return false;
    JitSelProj S;
    JitProj &J = S.proj;
    std::memset(J.col_values, 0, sizeof(J.col_values));
    std::memset(J.col_valid, 0, sizeof(J.col_valid));
    std::memset(J.col_dtype, 0, sizeof(J.col_dtype));
    auto slot_of = [&](const void *values, const uint8_t *valid, int dtype) {
        if (valid || !is_word_type(dtype)) return -1; // (nullable or bit-packed inputs: the two-kernel form handles them)
        for (int k = 0; k < J.ncols; ++k)
            if (J.col_values[k] == values && J.col_dtype[k] == dtype) return k;
        if (J.ncols == JP_MAX_COLS) return -1;
        J.col_values[J.ncols] = values;
        J.col_dtype[J.ncols] = dtype;
        return J.ncols++;
    };
    auto renumber = [&](ExProgram &P, uint32_t *mask) {
        for (int i = 0; i < P.n; ++i) {
            ExInstr &I = P.ins[i];
            for (int32_t *src : {&I.a_src, &I.b_src}) {
                if (*src == EX_LIT_NULL) return false;
                if (*src < EX_COL) continue;
                const int k = *src - EX_COL, u = slot_of(P.col_values[k], P.col_valid[k], P.col_dtype[k]);
                if (u < 0) return false;
                *src = EX_COL + u;
                *mask |= 1u << u;
            }
        }
        return true;
    };
    {
        int root;
        std::vector<Node> t = parse(in, pred, pred_nodes, &root);
        bool nv = false;
        if (t[size_t(root)].kind != NQE_EXPR_BINARY || t[size_t(root)].out_dtype != NQE_BOOLEAN || !build_program(in, t, root, &S.pred, &nv) || nv) return false;
        // (A/B, NQE_FUSED_SELECT_MIN_STEPS=1: plain `id < K` / `age + 100` through this kernel is SLOWER than the two static kernels —
        // C2 0.36 -> 0.42 ms, random ids 0.46 -> 0.49: their mask and compaction passes already read each column once)
        static const int min_steps = getenv("NQE_FUSED_SELECT_MIN_STEPS") ? atoi(getenv("NQE_FUSED_SELECT_MIN_STEPS")) : 3;
        if (S.pred.n < min_steps || !renumber(S.pred, &S.pred_cols)) return false;
    }
    for (int e = 0; e < num_exprs; ++e) {
        int root;
        std::vector<Node> t = parse(in, nodes + expr_offsets[e], expr_offsets[e + 1] - expr_offsets[e], &root);
        const Node &rt = t[size_t(root)];
        JitProjOut o;
        o.out_dtype = rt.out_dtype;
        if (rt.kind == NQE_EXPR_COLUMN) {
            const DevColumn &c = in->cols[size_t(rt.column)];
            if (!is_word_type(c.dtype) || c.validity || c.length < n) return false;
            o.is_column = true;
            o.col = slot_of(c.values->ptr, nullptr, c.dtype);
            if (o.col < 0) return false;
            S.proj_cols |= 1u << o.col;
        } else if (rt.kind == NQE_EXPR_BINARY) {
            bool nv = false;
            if (rt.out_dtype == NQE_BOOLEAN || !build_program(in, t, root, &o.P, &nv) || nv) return false;
            if (!renumber(o.P, &S.proj_cols)) return false;
        } else
            return false; // a bare literal
        J.outs.push_back(o);
    }
    // the outputs are allocated for the worst case (every row kept): beyond 16 GB of them (NQE_FUSED_SELECT_MAX_GB) the two-kernel
    // form, which sizes its outputs exactly, is the better citizen
    static const double max_gb = getenv("NQE_FUSED_SELECT_MAX_GB") ? atof(getenv("NQE_FUSED_SELECT_MAX_GB")) : 16.0;
    if (double(n) * 8.0 * num_exprs > max_gb * double(1ull << 30)) return false;
    JitEntry *kernel = jit_select_project_entry(ctx, S);
    if (!kernel) return false; // being compiled (or no hipRTC): nothing allocated yet
    // worst-case outputs (every row kept), the chunk status words, ticket and total
    const int64_t n_chunks = (n + SP_BLOCK * SP_R - 1) / (SP_BLOCK * SP_R); // steps: one status word per workgroup step
    std::vector<DevColumn> cols;
    uint64_t *ow[JP_MAX_OUTS] = {};
    BufRef status;
    try {
        for (int e = 0; e < num_exprs; ++e) {
            cols.push_back(make_word_column(ctx, J.outs[size_t(e)].out_dtype, n, false));
            ow[e] = (uint64_t *)cols.back().values->ptr;
        }
        status = dev_alloc(ctx, size_t(n_chunks) * 8 + 16);
    } catch (const Error &e) {
        // the worst-case outputs do not fit: the two-kernel form sizes its outputs exactly and may still run
        if (e.code != NQE_ERR_OUT_OF_MEMORY) throw;
        return false;
    }
    NQE_HIP_CHECK(hipMemsetAsync(status->ptr, 0, size_t(n_chunks) * 8 + 16, ctx->stream));
    unsigned long long *st = (unsigned long long *)status->ptr;
    jit_select_project(ctx, kernel, S, n, ow, st, reinterpret_cast<uint32_t *>(st + n_chunks), st + n_chunks + 1);
    const int64_t total = int64_t(read_scalar(ctx, (const unsigned long long *)(st + n_chunks + 1)));
    for (auto &c : cols) {
        c.length = total;
        // fewer than half the rows the buffer holds and more than 64 MB of it unused: give the large block back (the copy moves
        // `total` rows — less than the pass wrote — and the result no longer pins several times the memory it needs)
        if (total * 2 < n && (n - total) * 8 > (int64_t(64) << 20)) c = slice_column(ctx, c, 0, total);
    }
    *total_out = total;
    *result = std::move(cols);
    return true;
}

// The streaming aggregate under a predicate tree through the lean specialised kernel (expr_jit.hpp: nqe_jit_agg).  On success the
// workgroups' tables are in *partials ([grid][span] sums | mins | maxs as doubles, then counts as uint32 with the NaN mark in the
// top bit) for aggregate.hip's merge kernel; false: the shape does not qualify or the kernel is still being compiled.
bool aggregate_tree_specialised(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes, const nqe_expr_node *group, int group_nodes,
                                int val_col, int grid, BufRef *partials, uint32_t *span_out, int64_t *bias_out, bool dry_run) {
    const char *mr = getenv("NQE_JIT_MIN_ROWS");
    const int64_t min_rows = mr ? atoll(mr) : (int64_t(1) << 22);
    const int64_t n = in->rows;
    if (getenv("NQE_NO_JIT") || getenv("NQE_NO_AGG_JIT") || n < min_rows || !group || group_nodes < 1 || val_col < 0) return false;
    const DevColumn &vc = in->cols[size_t(val_col)];
    if (vc.validity || !is_word_type(vc.dtype) || vc.length < n) return false;
    JitAgg G;
    std::memset(G.col, 0, sizeof(G.col));
    std::memset(&G.pred, 0, sizeof(G.pred));
    G.ncols = 0;
    auto slot_of = [&](const void *values, const uint8_t *valid, int dtype) {
        if (valid || !is_word_type(dtype)) return -1;
        for (int k = 0; k < G.ncols; ++k)
            if (G.col[k] == values) return k;
        if (G.ncols == JA_MAX_COLS) return -1;
        G.col[G.ncols] = values;
        return G.ncols++;
    };
    auto renumber = [&](ExProgram &P) {
        for (int i = 0; i < P.n; ++i) {
            ExInstr &I = P.ins[i];
            for (int32_t *src : {&I.a_src, &I.b_src}) {
                if (*src == EX_LIT_NULL) return false;
                if (*src < EX_COL) continue;
                const int k = *src - EX_COL, u = slot_of(P.col_values[k], P.col_valid[k], P.col_dtype[k]);
                if (u < 0) return false;
                *src = EX_COL + u;
            }
        }
        return true;
    };
    { // the key: `… % m`, m a literal other than 0, -1, +-1; integer all the way (the static path has vetted it as fault-free)
        int root;
        std::vector<Node> t = parse(in, group, group_nodes, &root);
        bool nv = false;
        if (t[size_t(root)].kind != NQE_EXPR_BINARY || !build_program(in, t, root, &G.key, &nv) || nv || G.key.n < 1) return false;
        const ExInstr &last = G.key.ins[G.key.n - 1];
        if (last.op != NQE_OP_MODULOS || last.b_src != EX_LIT || !(last.dt == NQE_INT64 || last.dt == NQE_UINT64)) return false;
        for (int i = 0; i < G.key.n; ++i) {
            const ExInstr &I = G.key.ins[i];
            if (!(I.dt == NQE_INT64 || I.dt == NQE_UINT64) || I.op < NQE_OP_PLUS || I.op > NQE_OP_MODULOS) return false;
            if ((I.op == NQE_OP_DIVIDE || I.op == NQE_OP_MODULOS) && (I.b_src != EX_LIT || I.lit_b == 0 || I.lit_b == ~0ull)) return false; // could fault
        }
        G.key_signed = last.dt == NQE_INT64;
        G.modulus = G.key_signed ? uint64_t(int64_t(last.lit_b) < 0 ? 0ull - last.lit_b : last.lit_b) : last.lit_b;
        if (G.modulus < 2) return false;
        const uint64_t span64 = G.key_signed ? 2 * G.modulus - 1 : G.modulus;
        if (span64 < 512 || span64 > 4096) return false; // (fewer slots: every lane of a wave updates the same few words — the static kernel replicates such tables)
        G.span = uint32_t(span64);
        if (!renumber(G.key)) return false;
    }
    G.val_slot = slot_of(vc.values->ptr, nullptr, vc.dtype);
    if (G.val_slot < 0) return false;
    G.val_dtype = vc.dtype;
    if (pred && pred_nodes > 0) {
        int root;
        std::vector<Node> t = parse(in, pred, pred_nodes, &root);
        bool nv = false;
        if (t[size_t(root)].kind != NQE_EXPR_BINARY || t[size_t(root)].out_dtype != NQE_BOOLEAN || !build_program(in, t, root, &G.pred, &nv) || nv) return false;
        if (!renumber(G.pred)) return false;
    }
    JitEntry *kernel = jit_aggregate_entry(ctx, G);
    if (!kernel) return false;
    if (dry_run) return true; // the caller only asks whether this query can take the kernel NOW (its compilation has been started otherwise)
    const size_t cells = size_t(grid) * G.span;
    *partials = dev_alloc(ctx, cells * 28 + 64);
    double *ps = (double *)(*partials)->ptr;
    jit_aggregate_launch(ctx, kernel, G, n, unsigned(grid), ps, ps + cells, ps + 2 * cells, reinterpret_cast<uint32_t *>(ps + 3 * cells));
    *span_out = G.span;
    *bias_out = G.key_signed ? int64_t(G.modulus) - 1 : 0;
    return true;
}

} // namespace nqe

using namespace nqe;

extern "C" {

// A bare column evaluates to the input column itself (the reference's Arc clone, column.rs:41-43).  As an operator's OUTPUT it
// may stay an alias only of memory that an alias keeps alive; a column borrowed from the caller is copied.
static nqe::DevColumn own_output(nqe_ctx *ctx, nqe::DevColumn c) {
    if (c.shareable()) return c;
    nqe::DevColumn o = nqe::slice_column(ctx, c, 0, c.length);
    o.null_count = c.null_count;
    return o;
}

nqe_status nqe_expr_evaluate(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes, int32_t num_nodes,
                             nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    // the error flags are reset and read back (a stream synchronisation) only when the expression can raise one
    const bool fault = analyze_expr(in, nodes, num_nodes).may_fault;
    if (fault) flags_reset(ctx);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = in->rows;
    t->cols.push_back(own_output(ctx, evaluate_expr(ctx, in, nodes, num_nodes)));
    if (fault) throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_projection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *nodes,
                                  const int32_t *expr_offsets, int32_t num_exprs, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !out || num_exprs < 0 || (num_exprs > 0 && (!nodes || !expr_offsets)))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    bool fault = false;
    for (int e = 0; e < num_exprs; ++e) fault = fault || analyze_expr(in, nodes + expr_offsets[e], expr_offsets[e + 1] - expr_offsets[e]).may_fault;
    if (fault) flags_reset(ctx);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = in->rows;
    for (int e = 0; e < num_exprs; ++e)
        t->cols.push_back(own_output(ctx, evaluate_expr(ctx, in, nodes + expr_offsets[e], expr_offsets[e + 1] - expr_offsets[e])));
    if (fault) throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::fill_words_kernel);
