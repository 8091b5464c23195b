// aggregate_common.hpp — types, constants and device helpers shared by the aggregate translation units
// (aggregate.hip: general / un-grouped kernels, table kernels, host logic; aggregate_fast*.hip: the specialised streaming
// kernel and its instantiations; aggregate_partition.hip: the partitioned path).  Split so that the ~200 kernel
// instantiations compile in parallel.
#pragma once
#include <cfloat>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {
namespace agg {

constexpr uint64_t EMPTY_KEY = 0x8000000000000000ull; // i64::MIN; that key uses the extra slot [cap]
constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
constexpr int NV = 2;      // value columns per kernel pass
constexpr int NVMAX = 3;   // … of the one kind of instance that takes three (fast kernel, no min/max: C1's count / sum / avg over three columns)
#ifndef NQE_AGG_U
#define NQE_AGG_U 4
#endif
constexpr int AGG_U = NQE_AGG_U;   // rows per thread per iteration
constexpr int AGG_BLOCK = 1024;
constexpr uint32_t NAN_BIT = 0x80000000u;
constexpr uint64_t NAN_BIT64 = 0x8000000000000000ull; // the same mark in a 64-bit count (RangeRec)

// order-preserving map f64 -> u64 (non-NaN): integer min/max atomics give the f64 min/max
__host__ __device__ __forceinline__ uint64_t f64_to_ord(double d) {
    uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (uint64_t)__double_as_longlong(d);
#else
    std::memcpy(&b, &d, 8);
#endif
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_to_f64(uint64_t u) {
    uint64_t b = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    return __longlong_as_double((long long)b);
}

struct ColSrc {
    const void *values;
    const uint8_t *valid;
    int32_t dtype;
    int32_t present;
};

// global group table: arrays are [V][cap+1]; slot `cap` belongs to EMPTY_KEY itself
struct GroupTable {
    uint64_t *keys;     // [cap+1]; keys[cap] != EMPTY_KEY ⇔ special slot in use
    uint64_t *cnt;      // non-null values
    double *sum;
    uint64_t *mn;       // f64_to_ord
    uint64_t *mx;
    uint32_t *nan;      // any NaN seen
    uint32_t cap;       // power of two (hashed mode) / number of dense slots
    int32_t shift;      // 64 - log2(cap)
    int32_t V;
    int32_t pad;
    // Dense mode (partitioned aggregation, one pass over the value columns): partitions hold disjoint key sets, so a
    // workgroup writes the groups of its partition straight to slots [base, base + n) reserved with ONE atomic on this
    // counter — no global hash table, no initialisation, no collect pass.  null = hashed mode.
    uint32_t *dense_count;
};

struct AggArgs {
    int64_t n;
    int32_t pred_mode; // 0 none, 1 SimpleExpr over pred_src, 2 Boolean column (bits + validity) in pred_src, 3 `conj` (fast kernel only)
    int32_t has_key;
    int32_t pred_shares_key;
    int32_t nv;
    ColSrc pred_src;
    ColSrc key_src;
    SimpleExpr pred;
    SimpleExpr key;
    ColSrc val[NVMAX];
    int32_t val_shares_key[NVMAX];
    int32_t need_sum[NVMAX];
    int32_t need_minmax[NVMAX];
    int32_t v0; // first value slot of this pass in the global table
    int32_t lds_cap;
    int32_t lds_shift;
    int32_t allow_partition; // an LDS-table overflow asks the host for the partitioned path instead of falling back to global atomics
    // fast kernel, key = `col % m` with a small m: the key's value range (-m, m) fits the LDS table, so slot = key + direct_bias —
    // no hash, no probe, no compare on the per-row path of inputs whose key changes every row
    int32_t direct;
    int32_t direct_rep; // log2 of the replication of a small direct-mapped table (see the fast kernel): span << direct_rep <= lds_cap
    int64_t direct_bias;
    uint64_t direct_span; // direct == 2 (a MEASURED key range, nqe_ctx::agg_key_ranges): keys with key + direct_bias >= direct_span are not in it
    // fast kernel, key subsets: 2^subsets_log2 workgroups share every row range and each keeps only the keys whose hash bits
    // [subset_shift, subset_shift + subsets_log2) name it — 2^subsets_log2 LDS tables' worth of groups without partitioning the rows
    int32_t subsets_log2;
    int32_t subset_shift;
    // … over a direct-mapped table (direct != 0, round 5): TWO subsets, the two halves of the measured key range — direct_sub_width slots
    // each; subset of a key = (key + direct_bias >= direct_sub_width), slot = the offset inside the half: a key range of up to 2 x 5840
    // values (round 6: the halves are equal, whatever the range — equal work keeps the two readers of a tile together — and as wide as a
    // table without key words gets) is aggregated without hashing, probing or partitioning (0: the subsets are hashed)
    int32_t direct_sub_width;
    // hashed workgroup table of the streaming kernel: distinct keys it accepts (lds_find_or_insert_counted; 0: as many as find room)
    uint32_t lds_limit;
    ConjPred conj; // pred_mode 3
    // pred_mode 4: a predicate tree (nqe_internal.hpp, TreePred) run per row by the fast kernel's stack machine.  The program
    // lives in a small device buffer that the kernel reads through the scalar cache (a by-value array indexed at run time would
    // be copied to scratch memory); word slots: 0 key column, 1 first value column, 2 the predicate column (pred_src)
    uint64_t tree_prog; // device address of TreeInstr[tree_n]
    int32_t tree_n;
    int32_t tree_need_pw; // some operand reads word slot 2
    // the streaming kernel's ONE-tile loop (three value columns, the stack-machine predicate) looks at the overflow flags (its own LDS
    // word, two device words) when (iteration & mask) == mask: each look drains the wave's vector-memory queue — the tile it has just
    // requested — before the flag loads can return (0: every iteration).  The two-tile loop looks every iteration (see there).
    int32_t flag_check_mask;
    // fast kernel, direct-mapped table without validity bitmaps: the workgroup's table leaves WHOLE — partials + ((j * grid + workgroup) *
    // partial_span + key index) per value column j as [sums | mins | maxs] doubles then counts (uint32, NaN mark in the top bit), the layout
    // agg_fold_partials_kernel reads — instead of being folded into the group table with device-scope atomics (0: the atomics).
    // 256 workgroups x 4096 groups x 4 atomics were 0.17 ms of a 0.47 ms pass (10^8 rows, 4096 random keys).
    uint32_t partial_span;
    uint64_t partials;
};

// ------------------------------------------------------------------ predicate trees (pred_mode 4)
typedef const __attribute__((address_space(4))) TreeInstr *TreeProg; // constant address space: uniform loads go through the scalar unit

// Runs the program over a register tile of U rows, operator-major: the instruction's fields are scalars, the (wave-uniform)
// dispatch on operator, type and operand sources happens once per instruction per tile, the U rows are straight-line code under
// it.  Two typed stacks: 64-bit VALUES (depth 1: the accumulator `acc`) and BOOLEANS — which stay what a vector compare makes
// them on this machine, lane masks in scalar registers, so that `and` / `or` are scalar instructions and cost the vector unit
// nothing (depth 3).  Instruction forms (the host normalises to them, match_tree_pred): x ∈ {acc, word k}, y ∈ {literal, acc,
// word k}, `rev` swaps the operands of a subtraction or comparison whose literal stood on the left.  No NULLs (the host admits
// neither nullable columns nor NULL literals) and no faults (divisors are vetted literals).
template <int U>
__device__ __forceinline__ void tree_pred_eval(uint64_t prog_addr, int n, const uint64_t (&w0)[U], const uint64_t (&w1)[U], const uint64_t (&w2)[U],
                                               bool (&res)[U]) {
    TreeProg prog = (TreeProg)prog_addr;
    uint64_t acc[U];
    bool b0[U], b1[U], b2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc[u] = 0;
        b0[u] = b1[u] = b2[u] = false;
    }
    // the NEXT instruction's fields are requested (scalar loads) before the current one executes: fetched on demand, the two or
    // three dependent scalar-cache round trips per instruction were what the machine cost (0.15 ms per instruction per 2x10^8 rows)
    struct Fields {
        int op, dt, x_src, y_src;
        uint64_t rev, lit;
        OpAux aux;
    };
    auto fetch = [&](int pc) {
        Fields f;
        f.op = prog[pc].op;
        f.dt = prog[pc].dt;
        f.x_src = prog[pc].a_src;
        f.y_src = prog[pc].b_src;
        f.rev = prog[pc].lit_a;
        f.lit = prog[pc].lit_b;
        f.aux.pow2_shift = prog[pc].aux.pow2_shift;
        f.aux.more = prog[pc].aux.more;
        f.aux.abs_lit = prog[pc].aux.abs_lit;
        f.aux.magic = prog[pc].aux.magic;
        return f;
    };
    Fields nxt = fetch(0);
    for (int pc = 0; pc < n; ++pc) {
        const Fields cur = nxt;
        nxt = fetch(pc + 1 < n ? pc + 1 : pc);
        const int op = cur.op, dt = cur.dt, x_src = cur.x_src, y_src = cur.y_src;
        if (op == NQE_OP_AND) {
#pragma unroll
            for (int u = 0; u < U; ++u) { b0[u] = b1[u] && b0[u]; b1[u] = b2[u]; }
            continue;
        }
        if (op == NQE_OP_OR) {
#pragma unroll
            for (int u = 0; u < U; ++u) { b0[u] = b1[u] || b0[u]; b1[u] = b2[u]; }
            continue;
        }
        const uint64_t lit = cur.lit;
        const bool rev = cur.rev != 0;
        const OpAux aux = cur.aux;
        uint64_t x[U], y[U];
        if (x_src == TS_STACK) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = acc[u]; }
        else if (x_src == TS_W0) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = w0[u]; }
        else if (x_src == TS_W0 + 1) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = w1[u]; }
        else { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = w2[u]; }
        if (y_src == TS_LIT) { _Pragma("unroll") for (int u = 0; u < U; ++u) y[u] = lit; }
        else if (y_src == TS_STACK) { _Pragma("unroll") for (int u = 0; u < U; ++u) y[u] = acc[u]; }
        else if (y_src == TS_W0) { _Pragma("unroll") for (int u = 0; u < U; ++u) y[u] = w0[u]; }
        else if (y_src == TS_W0 + 1) { _Pragma("unroll") for (int u = 0; u < U; ++u) y[u] = w1[u]; }
        else { _Pragma("unroll") for (int u = 0; u < U; ++u) y[u] = w2[u]; }
        if (rev) {
#pragma unroll
            for (int u = 0; u < U; ++u) { const uint64_t t = x[u]; x[u] = y[u]; y[u] = t; }
        }
        if (op <= NQE_OP_GT_EQ) {
            bool r[U];
#define NQE_TREE_CMP(O)                                                                                                      \
    case O:                                                                                                                  \
        if (dt == NQE_INT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(O, NQE_INT64, x[u], y[u], aux, false, nullptr) != 0; }        \
        else if (dt == NQE_FLOAT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(O, NQE_FLOAT64, x[u], y[u], aux, false, nullptr) != 0; } \
        else { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(O, NQE_UINT64, x[u], y[u], aux, false, nullptr) != 0; }                       \
        break;
            switch (op) {
                NQE_TREE_CMP(NQE_OP_EQ) NQE_TREE_CMP(NQE_OP_NOT_EQ) NQE_TREE_CMP(NQE_OP_LT) NQE_TREE_CMP(NQE_OP_LT_EQ) NQE_TREE_CMP(NQE_OP_GT)
            default:
                if (dt == NQE_INT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(NQE_OP_GT_EQ, NQE_INT64, x[u], y[u], aux, false, nullptr) != 0; }
                else if (dt == NQE_FLOAT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(NQE_OP_GT_EQ, NQE_FLOAT64, x[u], y[u], aux, false, nullptr) != 0; }
                else { _Pragma("unroll") for (int u = 0; u < U; ++u) r[u] = apply_binary<false>(NQE_OP_GT_EQ, NQE_UINT64, x[u], y[u], aux, false, nullptr) != 0; }
                break;
            }
#undef NQE_TREE_CMP
#pragma unroll
            for (int u = 0; u < U; ++u) { b2[u] = b1[u]; b1[u] = b0[u]; b0[u] = r[u]; }
        } else {
#define NQE_TREE_OP(O)                                                                                                       \
    case O:                                                                                                                  \
        if (dt == NQE_INT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(O, NQE_INT64, x[u], y[u], aux, false, nullptr); }        \
        else if (dt == NQE_FLOAT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(O, NQE_FLOAT64, x[u], y[u], aux, false, nullptr); } \
        else { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(O, NQE_UINT64, x[u], y[u], aux, false, nullptr); }                       \
        break;
            switch (op) {
                NQE_TREE_OP(NQE_OP_PLUS) NQE_TREE_OP(NQE_OP_MINUS) NQE_TREE_OP(NQE_OP_MULTIPLY) NQE_TREE_OP(NQE_OP_DIVIDE)
            default:
                if (dt == NQE_INT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(NQE_OP_MODULOS, NQE_INT64, x[u], y[u], aux, false, nullptr); }
                else if (dt == NQE_FLOAT64) { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(NQE_OP_MODULOS, NQE_FLOAT64, x[u], y[u], aux, false, nullptr); }
                else { _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u] = apply_binary<false>(NQE_OP_MODULOS, NQE_UINT64, x[u], y[u], aux, false, nullptr); }
                break;
            }
#undef NQE_TREE_OP
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) res[u] = b0[u];
}

template <bool MOD, bool SGN, bool POW2> __device__ __forceinline__ uint64_t divmod_by_literal(uint64_t a, uint64_t lit, const OpAux &aux);
// ConjPred's general form over a register tile of U rows, test-major: the (wave-uniform) dispatch on a test's source word and
// arithmetic step happens once per test per tile — evaluated row by row (conj_pass) it was repeated for every row, and the scalar
// unit, one instruction per clock per CU, became the bound: +0.75 ms per 2x10^8 rows for two tests.
template <int U>
__device__ __forceinline__ void conj_general_tile(const ConjPred &c, const uint64_t (&w0)[U], const uint64_t (&w1)[U], const uint64_t (&w2)[U],
                                                  bool (&res)[U]) {
    uint32_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = 0;
    // (constant test indices: a run-time index would send the ConjPred to scratch memory)
#define NQE_CONJ_TILE_TEST(T)                                                                                                \
    if (c.n > T) {                                                                                                           \
        uint64_t x[U];                                                                                                       \
        /* (neither three branches around copies nor selects: the compiler turns both into ONE load through a selected     */ \
        /* pointer, and the tile's words — indexed through a pointer — then live in scratch memory)                       */ \
        /* hence bit masks, which are arithmetic on VALUES)                                                                */ \
        const uint64_t m0 = c.t[T].src == 0 ? ~0ull : 0ull, m1 = c.t[T].src == 1 ? ~0ull : 0ull;                              \
        _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = (w0[u] & m0) | (~m0 & ((w1[u] & m1) | (~m1 & w2[u])));             \
        if (c.t[T].pre) {                                                                                                         \
            const uint64_t lit = c.t[T].pre_lit;                                                                                  \
            if (c.t[T].pre_dt == NQE_FLOAT64) {                                                                                   \
                const double dl = u2d(lit);                                                                                  \
                if (c.t[T].pre == NQE_OP_PLUS) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(u2d(x[u]) + dl); }      \
                else if (c.t[T].pre == NQE_OP_MULTIPLY) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(u2d(x[u]) * dl); } \
                else if (c.t[T].pre == NQE_OP_MINUS) {                                                                            \
                    if (c.t[T].pre_rev) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(dl - u2d(x[u])); }             \
                    else { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(u2d(x[u]) - dl); }                       \
                } else if (c.t[T].pre_rev) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(dl / u2d(x[u])); }          \
                else { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = d2u(u2d(x[u]) / dl); }                           \
            } else if (c.t[T].pre == NQE_OP_PLUS) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] += lit; }                  \
            else if (c.t[T].pre == NQE_OP_MULTIPLY) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] *= lit; }                \
            else if (c.t[T].pre == NQE_OP_MINUS) {                                                                                \
                if (c.t[T].pre_rev) { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = lit - x[u]; }                          \
                else { _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] -= lit; }                                          \
            } else {                                                                                                         \
                OpAux aux; aux.pow2_shift = c.t[T].pre_aux.pow2_shift; aux.more = c.t[T].pre_aux.more; aux.abs_lit = c.t[T].pre_aux.abs_lit; aux.magic = c.t[T].pre_aux.magic;                                                                                 \
                const bool mod = c.t[T].pre == NQE_OP_MODULOS, sgn = c.t[T].pre_dt == NQE_INT64, pow2 = aux.pow2_shift >= 0;           \
                if (mod) {                                                                                                   \
                    if (sgn) { if (pow2) { NQE_CONJ_DM(true, true, true); } else { NQE_CONJ_DM(true, true, false); } }        \
                    else { if (pow2) { NQE_CONJ_DM(true, false, true); } else { NQE_CONJ_DM(true, false, false); } }          \
                } else {                                                                                                     \
                    if (sgn) { if (pow2) { NQE_CONJ_DM(false, true, true); } else { NQE_CONJ_DM(false, true, false); } }      \
                    else { if (pow2) { NQE_CONJ_DM(false, false, true); } else { NQE_CONJ_DM(false, false, false); } }        \
                }                                                                                                            \
            }                                                                                                                \
        }                                                                                                                    \
        _Pragma("unroll") for (int u = 0; u < U; ++u) idx[u] |= conj_test(c.t[T], x[u]) ? (1u << T) : 0u;                         \
    }
#define NQE_CONJ_DM(M, S, P) _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = divmod_by_literal<M, S, P>(x[u], lit, aux)
    NQE_CONJ_TILE_TEST(0)
    NQE_CONJ_TILE_TEST(1)
    NQE_CONJ_TILE_TEST(2)
    NQE_CONJ_TILE_TEST(3)
#undef NQE_CONJ_DM
#undef NQE_CONJ_TILE_TEST
#pragma unroll
    for (int u = 0; u < U; ++u) res[u] = (c.truth >> idx[u]) & 1u;
}

__device__ __forceinline__ int lds_find_or_insert(uint64_t *keys, uint64_t key, uint32_t cap, int shift) {
    if (key == EMPTY_KEY) {
        keys[cap] = 0; // mark special slot used (idempotent plain store)
        return int(cap);
    }
    uint32_t slot = uint32_t((key * GOLD) >> shift);
    for (int probe = 0; probe < 48; ++probe) {
        uint64_t k = keys[slot];
        if (k == key) return int(slot);
        if (k == EMPTY_KEY) {
            uint64_t old = atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == EMPTY_KEY || old == key) return int(slot);
        }
        slot = (slot + 1) & (cap - 1);
    }
    return -1; // workgroup table full for this key: caller goes to the global table
}

// … of the streaming kernel: the table also counts its keys and rejects new ones beyond `limit` (AggArgs::lds_limit, three quarters of
// the slots) — linear probing at load 0.73 costs 2.4x the time per row of load 0.6, at 0.85 5.5x (10^8 rows: 2500 keys 0.32 ms, 3000
// 0.74, 3500 1.71; the next tier takes 0.86-0.92), and a table that is merely crowded never says so: the rejected key asks the host
// for the next tier
__device__ __forceinline__ int lds_find_or_insert_counted(uint64_t *keys, uint64_t key, uint32_t cap, int shift, uint32_t *used, uint32_t limit) {
    if (key == EMPTY_KEY) {
        keys[cap] = 0;
        return int(cap);
    }
    uint32_t slot = uint32_t((key * GOLD) >> shift);
    for (int probe = 0; probe < 48; ++probe) {
        uint64_t k = keys[slot];
        if (k == key) return int(slot);
        if (k == EMPTY_KEY) {
            uint64_t old = atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == key) return int(slot);
            if (old == EMPTY_KEY) return (limit && atomicAdd(used, 1u) >= limit) ? -1 : int(slot); // (the slot stays taken: the attempt is abandoned anyway)
        }
        slot = (slot + 1) & (cap - 1);
    }
    return -1;
}

__device__ __forceinline__ int64_t global_find_or_insert(const GroupTable &g, uint64_t key, int *flags) {
    if (key == EMPTY_KEY) {
        __hip_atomic_store(&g.keys[g.cap], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return int64_t(g.cap);
    }
    uint32_t slot = uint32_t((key * GOLD) >> g.shift);
    // bounded probe sequence: a table that needs more than this is treated as full (the host retries with a
    // larger one) — an unbounded walk over a nearly full table is O(rows x capacity)
    const uint32_t max_probe = g.cap < 512u ? g.cap : 512u;
    for (uint32_t probe = 0; probe < max_probe; ++probe) {
        uint64_t k = __hip_atomic_load(&g.keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == key) return int64_t(slot);
        if (k == EMPTY_KEY) {
            uint64_t old = atomicCAS((unsigned long long *)&g.keys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == EMPTY_KEY || old == key) return int64_t(slot);
        }
        slot = (slot + 1) & (g.cap - 1);
    }
    atomicOr(&flags[NQE_FLAG_TABLE_FULL], 1);
    return -1;
}

__device__ __forceinline__ void global_update(const GroupTable &g, int64_t slot, int v, uint64_t cnt, double sum, bool has_sum,
                                              uint64_t mn, uint64_t mx, bool has_minmax, bool nan) {
    size_t o = size_t(v) * (size_t(g.cap) + 1) + size_t(slot);
    if (cnt) atomicAdd((unsigned long long *)&g.cnt[o], (unsigned long long)cnt);
    if (has_sum) unsafeAtomicAdd(&g.sum[o], sum);
    if (has_minmax) {
        atomicMin((unsigned long long *)&g.mn[o], (unsigned long long)mn);
        atomicMax((unsigned long long *)&g.mx[o], (unsigned long long)mx);
    }
    if (nan) atomicOr(&g.nan[o], 1u);
}

// group key of a row for the kernels that compute it inline.  KEY: 0 = the column itself, 1 = `col % ±2^k` (mask), 2 = `col % d`
// (magic multiply), 3 = any fault-free chain of up to four integer operations with literals (`(id + 1) % 1000`, `id / 7 * 3`, …)
// through the generic SimpleExpr interpreter — wave-uniform branches on the operator, no flags (the host admits only chains
// whose divisors are literals other than 0 and -1)
template <int KEY>
__device__ __forceinline__ uint64_t inline_key(const SimpleExpr &ke, uint64_t x, uint64_t key_mask, const OpAux &key_aux, bool key_signed) {
    if (KEY == 0) return x;
    if (KEY == 3) return eval_simple<false>(ke, x, false, nullptr);
    // truncated remainder by a literal: |x| mod |d|, sign of the dividend
    uint64_t sgn = key_signed ? uint64_t((long long)x >> 63) : 0ull;
    if (KEY == 1) {
        // |d| = 2^k: x - ((x + (x < 0 ? 2^k - 1 : 0)) & ~(2^k - 1)) — the multiple of 2^k that truncated division rounds to, without
        // taking and re-applying the sign (8 instead of 12 vector instructions per row)
        const uint64_t y = (x + (sgn & key_mask)) & ~key_mask;
        return x - y;
    }
    uint64_t ux = (x ^ sgn) - sgn;
    uint64_t ur = ux - udiv_magic(ux, key_aux) * key_aux.abs_lit;
    return (ur ^ sgn) - sgn;
}

// truncated division / remainder by a vetted literal (not 0, not -1) with everything that does not depend on the row fixed at
// compile time; same results as apply_binary
template <bool MOD, bool SGN, bool POW2>
__device__ __forceinline__ uint64_t divmod_by_literal(uint64_t a, uint64_t lit, const OpAux &aux) {
    const bool xneg = SGN && (long long)a < 0;
    const uint64_t ux = xneg ? 0ull - a : a;
    const uint64_t uq = POW2 ? ux >> aux.pow2_shift : udiv_magic(ux, aux);
    if (MOD) {
        const uint64_t ur = POW2 ? ux & (aux.abs_lit - 1) : ux - uq * aux.abs_lit;
        return xneg ? 0ull - ur : ur;
    }
    const bool neg = SGN && (xneg != ((long long)lit < 0));
    return neg ? 0ull - uq : uq;
}

// keys of a register tile of U rows.  The interpreted variant (KEY = 3) runs operator-major: the (wave-uniform) dispatch on the
// operator, its type and the divisor's shape happens once per operator per tile, the U rows are straight-line code under it —
// per-row interpretation cost `(id + 1) % 1000` 0.90 ms per 2x10^8 rows against 0.56 ms for the built-in `id % 1000`.
// F64: Float64 steps are compiled in (predicate chains of the one-value-column kernels only: with them in every instance the
// interpreted-key and two-value-column variants spilled 10-120 more VGPRs)
template <int KEY, int U, bool F64 = false>
__device__ __forceinline__ void inline_keys(const SimpleExpr &ke, const uint64_t (&kw)[U], uint64_t (&key)[U], uint64_t key_mask,
                                            const OpAux &key_aux, bool key_signed) {
    if (KEY != 3) {
#pragma unroll
        for (int u = 0; u < U; ++u) key[u] = inline_key<KEY>(ke, kw[u], key_mask, key_aux, key_signed);
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = kw[u];
#pragma unroll
    for (int k = 0; k < SIMPLE_MAX_OPS; ++k) {
        if (k >= ke.nops) break;
        const int op = ke.op[k];
        const bool ll = ke.lit_left[k] != 0;
        const uint64_t lit = ke.lit[k];
        if (F64 && ke.op_dtype[k] == NQE_FLOAT64) {
            // Float64 steps of a predicate chain (`v * 2.0 + 1.0 > 50.0`): IEEE arithmetic, ordered compares (a NaN fails all but !=);
            // the host admits a division only by a non-zero literal on the right (arrow-rs 13 raises DivideByZero on a zero divisor)
            const double dl = u2d(lit);
#define NQE_F64(EXPR) _Pragma("unroll") for (int u = 0; u < U; ++u) { const double x = u2d(key[u]); key[u] = (EXPR); }
            switch (op) {
            case NQE_OP_PLUS: NQE_F64(d2u(x + dl)); break;
            case NQE_OP_MULTIPLY: NQE_F64(d2u(x * dl)); break;
            case NQE_OP_MINUS: if (ll) { NQE_F64(d2u(dl - x)); } else { NQE_F64(d2u(x - dl)); } break;
            case NQE_OP_DIVIDE: NQE_F64(d2u(x / dl)); break;
            case NQE_OP_EQ: NQE_F64(x == dl ? 1ull : 0ull); break;
            case NQE_OP_NOT_EQ: NQE_F64(x != dl ? 1ull : 0ull); break;
            case NQE_OP_LT: if (ll) { NQE_F64(dl < x ? 1ull : 0ull); } else { NQE_F64(x < dl ? 1ull : 0ull); } break;
            case NQE_OP_LT_EQ: if (ll) { NQE_F64(dl <= x ? 1ull : 0ull); } else { NQE_F64(x <= dl ? 1ull : 0ull); } break;
            case NQE_OP_GT: if (ll) { NQE_F64(dl > x ? 1ull : 0ull); } else { NQE_F64(x > dl ? 1ull : 0ull); } break;
            default: if (ll) { NQE_F64(dl >= x ? 1ull : 0ull); } else { NQE_F64(x >= dl ? 1ull : 0ull); } break;
            }
#undef NQE_F64
            continue;
        }
        if (op == NQE_OP_PLUS) {
#pragma unroll
            for (int u = 0; u < U; ++u) key[u] += lit;
        } else if (op == NQE_OP_MULTIPLY) {
#pragma unroll
            for (int u = 0; u < U; ++u) key[u] *= lit;
        } else if (op == NQE_OP_MINUS) {
            if (ll) {
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = lit - key[u];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] -= lit;
            }
        } else if (op <= NQE_OP_GT_EQ) { // comparison with the literal (a predicate chain's last step): 1 / 0
            // lit op x  ==  x op' lit
            const int o = !ll ? op : op == NQE_OP_LT ? NQE_OP_GT : op == NQE_OP_LT_EQ ? NQE_OP_GT_EQ : op == NQE_OP_GT ? NQE_OP_LT : op == NQE_OP_GT_EQ ? NQE_OP_LT_EQ : op;
            const bool sgn = ke.op_dtype[k] == NQE_INT64;
#define NQE_CMP(T, OP)                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < U; ++u) key[u] = (T(key[u]) OP T(lit)) ? 1ull : 0ull
            if (sgn) {
                switch (o) {
                case NQE_OP_EQ: NQE_CMP(int64_t, ==); break;
                case NQE_OP_NOT_EQ: NQE_CMP(int64_t, !=); break;
                case NQE_OP_LT: NQE_CMP(int64_t, <); break;
                case NQE_OP_LT_EQ: NQE_CMP(int64_t, <=); break;
                case NQE_OP_GT: NQE_CMP(int64_t, >); break;
                default: NQE_CMP(int64_t, >=); break;
                }
            } else {
                switch (o) {
                case NQE_OP_EQ: NQE_CMP(uint64_t, ==); break;
                case NQE_OP_NOT_EQ: NQE_CMP(uint64_t, !=); break;
                case NQE_OP_LT: NQE_CMP(uint64_t, <); break;
                case NQE_OP_LT_EQ: NQE_CMP(uint64_t, <=); break;
                case NQE_OP_GT: NQE_CMP(uint64_t, >); break;
                default: NQE_CMP(uint64_t, >=); break;
                }
            }
#undef NQE_CMP
        } else { // DIVIDE / MODULOS by a literal on the right (the host admits nothing else here)
            const OpAux aux = ke.aux[k];
            const bool mod = op == NQE_OP_MODULOS, sgn = ke.op_dtype[k] == NQE_INT64, pow2 = aux.pow2_shift >= 0;
#define NQE_DM(M, S, P)                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < U; ++u) key[u] = divmod_by_literal<M, S, P>(key[u], lit, aux)
            if (mod) {
                if (sgn) { if (pow2) { NQE_DM(true, true, true); } else { NQE_DM(true, true, false); } }
                else { if (pow2) { NQE_DM(true, false, true); } else { NQE_DM(true, false, false); } }
            } else {
                if (sgn) { if (pow2) { NQE_DM(false, true, true); } else { NQE_DM(false, true, false); } }
                else { if (pow2) { NQE_DM(false, false, true); } else { NQE_DM(false, false, false); } }
            }
#undef NQE_DM
        }
    }
}

__device__ __forceinline__ bool row_valid(const ColSrc &c, int64_t row) { return c.valid ? get_bit(c.valid, row) : true; }

// partitioned aggregation (aggregate_partition.hip)
constexpr int PARTS_LOG2 = 9;
constexpr int PARTS = 1 << PARTS_LOG2;

struct PartArgs {
    uint32_t *counts;        // [PARTS][nblocks] (count pass out)
    const uint64_t *offsets; // [PARTS][nblocks] exclusive scan of counts (scatter pass in)
    uint64_t *out_key;
    uint64_t *out_val[NV];
    int64_t chunk;           // rows per workgroup (multiple of AGG_BLOCK*AGG_U)
};

// Slab form of the partitioned path (no count pass): scatter workgroup w appends the tuples of partition p to ITS OWN slab
// (p, w) of fixed capacity — hash partitions of a chunk are near-uniform, so mean + 25 % + 64 tuples holds unless the keys are
// heavily skewed (then NQE_FLAG_SLAB_OVERFLOW sends the query to the exact count → scan → scatter form).  A tuple is
// (key, value...) in ONE stream: a 16-tuple run of a tile is 256 contiguous bytes instead of 128 B in each of two arrays.
struct SlabArgs {
    uint64_t *slabs;   // [W][PARTS][cap] tuples of (1 + nv) words: a scatter workgroup's 512 slabs are one contiguous region (with
                       // [PARTS][W] they lay 4 MB apart — 512 write streams over as many pages per workgroup)
    uint32_t *fill;    // [PARTS][W] tuples written
    int64_t chunk;     // rows per scatter workgroup (multiple of the scatter tile)
    int32_t W;         // scatter workgroups
    int32_t cap;       // tuples per slab
    int32_t parts_log2; // partitions of this run: 8 (twice the run length per partition and tile: the scatter's stores are what bound
                        // it — 0.72-0.82 ms per 10^8 rows against 0.88-0.93 with 512, while 128 leave the second kernel half the
                        // chip) until a partition holds more distinct keys than a workgroup table, then PARTS_LOG2
    // KEY-RANGE partitioning (round 4; 12-byte tuples only) for keys known to lie in a compact range [range_min, range_min + range_span)
    // (`col % m`, a sampled or remembered range): with d = key - range_min and hi = d >> parts_log2,
    //     partition = (d ^ ((hi * 0x9E3779B1) >> (32 - parts_log2))) & (parts - 1),   slot in the partition's table = hi
    // — a bijection between the range and parts x ceil(span / parts) slots, so the second kernel addresses its LDS table directly (no hash,
    // no probe, no key words: agg_slab_segments_direct_kernel) and rebuilds the key from (partition, slot).  The LOW bits pick the
    // partition: consecutive keys (sorted ids, `row number % m`) fall into different partitions — contiguous intervals sent a workgroup's
    // whole chunk to a few of them, whose slabs overflowed.  They are scrambled by a multiplicative hash of the high bits: plain XOR folds
    // of the high bits left keys with a common odd factor (5 k + c) on a quarter of the partitions at four times the load.  A key outside the range raises NQE_FLAG_OOB and the host redoes the query with hash partitions.  range_span 0: hash
    // partitioning.
    int64_t range_min;
    uint64_t range_span;
};

constexpr int SUB_LOG2 = 6;
constexpr int SUB = 1 << SUB_LOG2;

// kernel entry points of the other translation units
using FastKernel = void (*)(AggArgs, FastPred, GroupTable, int *);
// sub: the two-key-subset variant (AggArgs::subsets_log2 = 1), built for one value column without validity bitmaps — nullptr otherwise
// nomm: no aggregate of the pass needs min / max (two value columns, PRED 0/1, built-in keys: an instance whose batch loop fits)
// share: the one value column is the key column's own buffer (an instance that loads it once, eight rows per lane)
FastKernel pick_fast_kernel(int pred, int key, int nv, bool vf64, bool vnull, bool sub = false, bool nomm = false, bool share = false);
using PartKernel = void (*)(AggArgs, FastPred, PartArgs);
PartKernel pick_scatter_kernel(int pred, int key, int nv);
PartKernel pick_part_kernel(int pred, int key, int nv, bool scatter);
using SubpartitionKernel = void (*)(const uint64_t *, int64_t, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, uint64_t *, uint64_t *, uint64_t *);
SubpartitionKernel pick_subpartition_kernel(int nv);
using SegmentsKernel = void (*)(AggArgs, const uint64_t *, int64_t, int, int, int, const uint64_t *, const uint64_t *, const uint64_t *, GroupTable, int *);
SegmentsKernel pick_segments_kernel(int nv, bool vf64);
using SlabScatterKernel = void (*)(AggArgs, FastPred, SlabArgs, int *);
// k32 (one value column): 12-byte tuples {int32 key, value} — the scatter raises NQE_FLAG_KEY32_OVERFLOW on a key outside int32
SlabScatterKernel pick_slab_scatter_kernel(int pred, int key, int nv, bool k32 = false, int soa_threads = 1024); // soa_threads: workgroup size of the K32 (two-stream) form
int slab_scatter_rows_per_thread(int pred, int key, int nv);
int slab_scatter_soa_rows_per_thread(); // the K32 (SoA, whole-block) scatter
int slab_scatter_wg_per_cu();
// key-range partitions: partition of d = key - range_min, and the scramble of the slot that rebuilds the key's low bits from (partition, slot)
__device__ __forceinline__ uint32_t range_scramble(uint32_t hi, int parts_log2) { return uint32_t((uint64_t(hi * 0x9E3779B1u) << parts_log2) >> 32); } // (parts_log2 == 0: 0)
__device__ __forceinline__ uint32_t range_partition(uint64_t d, int parts_log2) {
    return (uint32_t(d) ^ range_scramble(uint32_t(d >> parts_log2), parts_log2)) & ((1u << parts_log2) - 1u);
}
// The RANGE TIER of the partitioned path (round 5; keys in a known range of at most PARTS << 12 values, one value column): the
// partition count follows the RANGE — ceil(span / 2^slots_log2) tables, at least 16, at most 256 (512 beyond 2^20 values) — instead of
// always being 256: the scatter's stores per partition and tile get longer (65536 groups: 64 partitions, scatter 0.70 -> 0.61 ms per
// 10^8 rows), and Q = workgroups / partitions workgroups share a partition in the second kernel, each aggregating every Q-th slab into
// its own LDS table and writing the WHOLE table [partition][q][slot] as 32-byte records (coalesced; no dense compaction, no atomics).
// agg_range_emit_kernel then reads the tables TRANSPOSED — a workgroup takes the slots s0 .. s0 + SB of every partition, i.e. a
// contiguous key interval — adds the Q partials of each slot, lays them out in key order in LDS, counts the occupied ones (decoupled
// look-back across workgroups) and writes keys and aggregates at their rank: the tail of the dense form (min / max of the written
// keys, pos[key - min] = slot + 1, one gathered read of five state words per group: 0.21 ms per step at 2^20 groups) is gone.
struct __attribute__((aligned(16))) RangeRec {
    double sum, mn, mx;
    uint64_t cnt; // rows | NAN_BIT64 (64 bits: a group's rows over all workgroups may pass 2^31 on a 288 GB part)
};
// at most four groups (`col % m`, m <= 4): the group state in registers (aggregate_tiny.hip); partials in the layout of AggArgs::partials
using TinyGroupsKernel = void (*)(AggArgs, FastPred, uint32_t, uint32_t, int *); // (…, m, tiles between two unpackings of the packed counters, flags)
TinyGroupsKernel pick_tiny_groups_kernel(int pred, int nv, bool minmax_last, uint32_t m, bool first_value_is_key);
using RangeSegmentsKernel = void (*)(AggArgs, SlabArgs, int, RangeRec *);
RangeSegmentsKernel pick_range_segments_kernel(bool vf64);
using SlabSegmentsKernel = void (*)(AggArgs, SlabArgs, GroupTable, int *);
SlabSegmentsKernel pick_slab_segments_kernel(int nv, bool vf64, bool k32 = false);
SlabSegmentsKernel pick_slab_segments_direct_kernel(bool vf64); // range partitions (SlabArgs::range_span != 0), one value column, 12-byte tuples

} // namespace agg
} // namespace nqe
