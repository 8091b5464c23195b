// Offline check of the run-time specialised kernels' GENERATORS (csrc/expr_jit.hpp) — no GPU needed: builds representative programs
// by hand, prints the generated source of each kernel family between "//==== <name>" markers; tools/jit_offline/run.sh compiles
// every one with hipRTC exactly as the library does (rtc_compile.cpp) and reports registers / spills from the code objects.
// Includes expr.hip wholesale (the generators live in its anonymous namespace) and links libnqe_hip.so for the host symbols.
#include "../../naive_query_engine_amd/csrc/expr.hip"
#include <cstdio>
using namespace nqe;

static ExInstr ins(int op, int dt, int a, int b, uint64_t la, uint64_t lb) {
    ExInstr i;
    std::memset(&i, 0, sizeof(i));
    i.op = op; i.dt = dt; i.a_src = a; i.b_src = b; i.lit_a = la; i.lit_b = lb;
    return i;
}
static uint64_t f64(double d) { uint64_t w; std::memcpy(&w, &d, 8); return w; }

int main() {
    // columns: 0 = id (Int64), 1 = v (Float64)
    ExProgram pred; // (id + 1) % 10 < 5
    std::memset(&pred, 0, sizeof(pred));
    pred.n = 3; pred.ncols = 2; pred.col_dtype[0] = NQE_INT64; pred.col_dtype[1] = NQE_FLOAT64;
    pred.ins[0] = ins(NQE_OP_PLUS, NQE_INT64, EX_COL + 0, EX_LIT, 0, 1);
    pred.ins[1] = ins(NQE_OP_MODULOS, NQE_INT64, EX_STACK, EX_LIT, 0, 10);
    pred.ins[2] = ins(NQE_OP_LT, NQE_INT64, EX_STACK, EX_LIT, 0, 5);
    ExProgram proj; // v * v + v / 4
    std::memset(&proj, 0, sizeof(proj));
    proj.n = 3; proj.ncols = 2; proj.col_dtype[0] = NQE_INT64; proj.col_dtype[1] = NQE_FLOAT64;
    proj.ins[0] = ins(NQE_OP_MULTIPLY, NQE_FLOAT64, EX_COL + 1, EX_COL + 1, 0, 0);
    proj.ins[1] = ins(NQE_OP_DIVIDE, NQE_FLOAT64, EX_COL + 1, EX_LIT, 0, f64(4.0));
    proj.ins[2] = ins(NQE_OP_PLUS, NQE_FLOAT64, EX_STACK, EX_STACK, 0, 0);
    ExProgram tree; // v < 20 or id % 3 = 0
    std::memset(&tree, 0, sizeof(tree));
    tree.n = 4; tree.ncols = 2; tree.col_dtype[0] = NQE_INT64; tree.col_dtype[1] = NQE_FLOAT64;
    tree.ins[0] = ins(NQE_OP_LT, NQE_FLOAT64, EX_COL + 1, EX_LIT, 0, f64(20.0));
    tree.ins[1] = ins(NQE_OP_MODULOS, NQE_INT64, EX_COL + 0, EX_LIT, 0, 3);
    tree.ins[2] = ins(NQE_OP_EQ, NQE_INT64, EX_STACK, EX_LIT, 0, 0);
    tree.ins[3] = ins(NQE_OP_OR, NQE_BOOLEAN, EX_STACK, EX_STACK, 0, 0);

    printf("//==== nqe_jit_expr\n%s", gen_source(pred, false, true).c_str());
    printf("//==== nqe_jit_expr_nulls\n%s", [&] { ExProgram p = pred; static const uint8_t dummy = 0; p.col_valid[0] = &dummy; return gen_source(p, true, true); }().c_str());

    JitProj J;
    std::memset(J.col_values, 0, sizeof(J.col_values)); std::memset(J.col_valid, 0, sizeof(J.col_valid)); std::memset(J.col_dtype, 0, sizeof(J.col_dtype));
    J.ncols = 2; J.col_dtype[0] = NQE_INT64; J.col_dtype[1] = NQE_FLOAT64;
    JitProjOut o; o.P = proj; o.out_dtype = NQE_FLOAT64;
    J.outs.push_back(o);
    JitProjOut c; c.is_column = true; c.col = 0; c.out_dtype = NQE_INT64;
    J.outs.push_back(c);
    printf("//==== nqe_jit_proj\n%s", gen_source_proj(J).c_str());

    JitSelProj S;
    S.proj = J; S.pred = pred; S.pred_cols = 1; S.proj_cols = 3;
    printf("//==== nqe_jit_selproj\n%s", gen_source_selproj(S).c_str());

    ExProgram keyp; // (id + 1) % 1024
    std::memset(&keyp, 0, sizeof(keyp));
    keyp.n = 2; keyp.ncols = 1; keyp.col_dtype[0] = NQE_INT64;
    keyp.ins[0] = ins(NQE_OP_PLUS, NQE_INT64, EX_COL + 0, EX_LIT, 0, 1);
    keyp.ins[1] = ins(NQE_OP_MODULOS, NQE_INT64, EX_STACK, EX_LIT, 0, 1024);
    JitAgg G;
    std::memset(G.col, 0, sizeof(G.col));
    G.pred = tree; G.key = keyp; G.ncols = 2; G.val_slot = 1; G.val_dtype = NQE_FLOAT64; G.key_signed = true; G.modulus = 1024; G.span = 2047;
    printf("//==== nqe_jit_agg -munsafe-fp-atomics\n%s", gen_source_agg(G).c_str());
    // unsigned key `u % 4096`, Int64 value = the key column, no predicate
    std::memset(&G.pred, 0, sizeof(G.pred));
    std::memset(&G.key, 0, sizeof(G.key));
    G.key.n = 1; G.key.ncols = 1; G.key.col_dtype[0] = NQE_UINT64;
    G.key.ins[0] = ins(NQE_OP_MODULOS, NQE_UINT64, EX_COL + 0, EX_LIT, 0, 4096);
    G.ncols = 1;
    G.val_dtype = NQE_INT64; G.val_slot = 0; G.key_signed = false; G.modulus = 4096; G.span = 4096;
    printf("//==== nqe_jit_agg_u64key_i64val -munsafe-fp-atomics\n%s", gen_source_agg(G).c_str());
    return 0;
}
