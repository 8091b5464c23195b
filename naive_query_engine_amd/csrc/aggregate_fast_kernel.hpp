// aggregate_fast_kernel.hpp — the specialised streaming kernel of the hash aggregate (see aggregate.hip for the operator
// and its semantics).  Instantiated by aggregate_fast_inst.hip, which the Makefile compiles once per predicate variant and
// validity mode, so that the eight slices of the 128 instantiations compile in parallel; aggregate_fast.hip picks among them.
#pragma once
#include "aggregate_common.hpp"

#ifndef NQE_AGG_BATCH
#define NQE_AGG_BATCH 1 // 0: the round-1 row loop (A/B runs)
#endif
#ifndef NQE_WIDE_TILES
#define NQE_WIDE_TILES 0 // that many (6, 8) rows per lane per tile for the TWO-column one-value instances (A/B): 8 spills 60-76 VGPRs, 6 fits (2 spilled) and is 4 % slower (headline kernel 2.45 -> 2.55 ms)
#endif
#ifndef NQE_AGG_BATCH2
#define NQE_AGG_BATCH2 0 // the batch loop for two value columns, rows one by one (NQE_AGG_BG2): built and measured at the register limit — 8-22 VGPRs spill in every such instance, so it stays off
#endif
#ifndef NQE_AGG_BATCH_TREE
#define NQE_AGG_BATCH_TREE 0 // the batch loop under the general range-test form (PRED = 5): 47-220 VGPRs spill, so it stays off
#endif
#ifndef NQE_TREE_PIPE
#define NQE_TREE_PIPE 0 // the prefetched second tile under a tree predicate (A/B)
#endif
#ifndef NQE_AGG_BG2
#define NQE_AGG_BG2 1
#endif
#ifndef NQE_ML_PIPE
#define NQE_ML_PIPE 1 // the prefetched second tile in the three-column instance with min / max on its last column (A/B: 10-12 VGPRs spill with it)
#endif
#ifndef NQE_AGG_RUN_BUDGET
#define NQE_AGG_RUN_BUDGET 32 // tile pairs between two looks at the keys while in the run loop
#endif

namespace nqe {
namespace agg {
namespace {

// ------------------------------------------------------------------ grouped kernel, fast path
// Sources are plain 8-byte columns without validity; predicate is none / integer `col cmp lit`;
// key is a plain column / `col % ±2^k`.  Differences from the general kernel above:
//   * NVT (value columns) and VF64 (all values Float64) are static, every statistic is always
//     maintained → no flag or dtype branches in the row loop;
//   * the integer predicate is a branch-free range test: the host rewrites `x op lit` into
//     lo <= (x ^ flip) <= hi, optionally negated (flip = sign bit for UInt64 → signed compares);
//   * min/max run on native v_min_f64/v_max_f64 (NaN operands are ignored by the instruction and
//     tracked by a flag, which is exactly OrderedFloat's min rule and makes max NaN at the end);
//   * loads are unconditional (row index clamped to n-1): no exec-mask branches around them;
//   * PIPE: the NEXT tile's words are requested before the current tile is processed (two register
//     tiles), so waits are counted `s_waitcnt vmcnt(k)` and a wave keeps a tile in flight while it
//     computes.
// PRED: 0 none, 1 range test on the key column itself (one load serves both), 2 on another column, 3 a fault-free integer chain
// `col op lit … cmp lit` over any column, interpreted operator-major like the KEY = 3 keys, 4 `A and B` / `A or B` of two range
// tests over the key column, the first value column and at most one more column (AggArgs::conj), 5 any other fault-free predicate
// nesting of and / or over up to four such tests, a test possibly with an arithmetic step (`id % 3 = 0`; ConjPred's general form),
// 6 any other fault-free predicate tree over those columns, run by the typed stack machine of aggregate_common.hpp
// (tree_pred_eval; AggArgs::tree_prog).
// MM = false: no aggregate of the pass asks for min / max (C1's count / sum / avg): their registers and LDS reads are left out,
// which is what lets the batch loop run beside TWO value columns (instances: no predicate or the key-range test, built-in keys).
// SHARE = true: the one value column IS the key column (`sum(id) … group by id % 1024`, SURVEY §8d's 8 B/row form): no second load
// of the same words, and EIGHT rows per lane per tile instead of four — with one 8-byte column a wave's two tiles held only 4 KB
// in flight (64 KB per CU, the chip's latency-bandwidth product is ~16 MB): the kernel ran at 4.6 TB/s with the vector and scalar
// units 30-40 % busy (PMC), i.e. latency-bound.
template <int PRED, int KEY, int NVT, bool VF64, bool VNULL, bool SUB = false, bool MM = true, bool SHARE = false>
__global__ void __launch_bounds__(AGG_BLOCK) agg_grouped_fast_kernel(AggArgs a, FastPred fp, GroupTable g, int *flags) {
    // SHARE with ONE value column: that column is the key column (one load, eight rows per lane).  SHARE with THREE: the FIRST value
    // column is the key column (C1's `count(id), sum(age), avg(score) … group by id % 3`) — the tile holds two value words per row, and
    // a second tile in flight fits the registers (the general three-column instance keeps one)
    constexpr bool SH1 = SHARE && NVT == 1, FK = SHARE && NVT == 3;
    // MM with THREE value columns: only the LAST one carries min / max (the reference's own query, src/main.rs:36-40: count(id), sum(age),
    // sum/avg/max/min(score) — the host orders the pass so): one pair of LDS arrays and one pair of run registers instead of three
    constexpr bool ML = MM && NVT == 3;
    constexpr int NMM = !MM ? 0 : (ML ? 1 : NVT); // value columns with min / max arrays
    auto mmcol = [](int j) { return MM && (!ML || j == NVT - 1); };
    constexpr int NVL = FK ? NVT - 1 : NVT; // value columns the tile loads
    // (key subsets: 2 rows per lane 0.59 ms at 6000 groups, 4: 0.49, 8: 0.46-0.47 and 3 MB more code — profiles/r06/ab_sub_plain_loads.txt)
    constexpr int TU = SH1 ? 2 * AGG_U : ((NQE_WIDE_TILES && NVT == 1 && !VNULL && !SUB && PRED <= 1 && KEY != 3) ? NQE_WIDE_TILES : AGG_U); // rows per lane per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t cap = uint32_t(a.lds_cap);
    const uint32_t slots = cap + 1;
    // A direct-mapped table over sources without validity bitmaps never reads a key word (a used slot is one with a count, its key is its
    // index): the key array is not laid out at all — 28 instead of 36 bytes per slot and value column, so ONE workgroup table takes a key
    // range of up to 5840 values (160 KB) where it took 4096 (round 6; the host sizes the table: AggRun::tier_streaming)
    const bool nokeys = !VNULL && a.direct != 0; // wave-uniform
    uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
    double *lsum = reinterpret_cast<double *>(lkeys + (nokeys ? 0u : slots)); // [NVT][slots]
    // min / max live in LDS as plain doubles (ds_min_f64 / ds_max_f64; a NaN never reaches them: every update is guarded by an
    // ordered compare, which a NaN fails) and take the order-preserving integer form only for the global table at the merge
    // (MM = false — no aggregate of the pass asks for min / max: the two arrays do not exist, 20 instead of 36 bytes per slot and
    // value column, and the host sizes the table accordingly: 2048 slots for three columns)
    double *lmn = lsum + NVT * slots;                                    // [NMM][slots]
    double *lmx = lmn + NMM * slots;                                     // [NMM][slots]
    uint32_t *lcnt = reinterpret_cast<uint32_t *>(MM ? lmx + NMM * slots : lmn); // [NVT][slots]
    auto mmo = [&](int j, uint32_t slot) { return uint32_t(ML ? 0 : j) * slots + slot; }; // a column's word in the min / max arrays
    const uint64_t ORD_MAX = f64_to_ord(DBL_MAX), ORD_MIN = f64_to_ord(-DBL_MAX);
    __shared__ int lds_full_flag;
    __shared__ uint32_t lds_used; // distinct keys of the hashed table (lds_find_or_insert_counted)
    volatile int *lds_full = &lds_full_flag;
    if (threadIdx.x == 0) lds_full_flag = 0, lds_used = 0;
    for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
        if (!nokeys) lkeys[s] = EMPTY_KEY;
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            lsum[j * slots + s] = 0.0;
            if (mmcol(j)) {
                lmn[mmo(j, s)] = DBL_MAX;
                lmx[mmo(j, s)] = -DBL_MAX;
            }
            lcnt[j * slots + s] = 0;
        }
    }
    __syncthreads();

    // LDS atomics per row are what bounds inputs whose key changes on every row of a thread (`id % 3` over row numbers: the run
    // cache never holds).  Without validity bitmaps every value column of a row counts the same rows: only column 0's counter is
    // updated and the merge hands its count to the others (their own words carry just the NaN mark); and a column nobody asks the
    // sum of (count(id)) gets no sum update (AggArgs::need_sum, wave-uniform) — count(id), sum(age), avg(score): 3 instead of 6.
    constexpr bool OWN_CNT = VNULL;
    bool nsum[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) nsum[j] = a.need_sum[j] != 0;
    bool run_live = false;
    uint64_t run_key = 0;
    uint32_t rcnt[NVT];
    double rsum[NVT], rmn[NVT], rmx[NVT];
    bool rnan[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        rcnt[j] = 0; rsum[j] = 0.0; rmn[j] = DBL_MAX; rmx[j] = -DBL_MAX; rnan[j] = false;
    }
    // key subsets (AggArgs::subsets_log2): the workgroups of one XCD that share a row range are neighbours in time, so the second
    // to 2^k-th reader of a tile is served by that XCD's L2 / the Infinity Cache
    // SUB: a separate set of instances — the test inside the one-subset kernels cost the 1000-3000-group cases 4-10 %
    const uint32_t sub_log2 = SUB ? uint32_t(a.subsets_log2) : 0u;
    const uint32_t sub_mask = (1u << sub_log2) - 1u;
    const uint32_t my_subset = (blockIdx.x >> 3) & sub_mask;
    const uint32_t lane_id = SUB ? ((blockIdx.x & 7u) | ((blockIdx.x >> (3 + sub_log2)) << 3)) : blockIdx.x;
    const uint32_t lanes = gridDim.x >> sub_log2;
    // subset of a key: two bits of the slot hash's product below the slot bits, XORed (one bit alone — a rotation sequence for keys
    // in arithmetic progression, like the slot bits — left each half of `5r - 77` clustered: probe sequences of 69 slots at load 0.6)
    const uint32_t dsub_width = SUB ? uint32_t(a.direct_sub_width) : 0u; // direct-mapped table in two key-range subsets (wave-uniform)
    auto foreign = [&](uint64_t key) {
        if (SUB && dsub_width) return uint32_t(uint64_t(int64_t(key) + a.direct_bias) >= uint64_t(dsub_width)) != my_subset; // (a key outside the range: the upper subset's, whose direct_slot rejects it)
        const uint64_t h = key * GOLD;
        return ((uint32_t(h >> a.subset_shift) ^ uint32_t(h >> 23)) & sub_mask) != my_subset;
    };
    // direct-mapped table (`… % m`): slot of a key.  With few groups every row of a wave would hit the same handful of LDS words
    // (same-address atomics serialise: `id % 3` over rows whose key changes every row ran at a fifth of the random-key rate), so a
    // small table is REPLICATED 2^direct_rep times, lane l of a wave updating replica l & (2^rep - 1); the replicas are folded into
    // replica 0 before the merge.
    const uint32_t rep_log2 = uint32_t(a.direct_rep), rep_lane = threadIdx.x & ((1u << rep_log2) - 1u);
    auto direct_slot = [&](uint64_t key) {
        const uint64_t d = uint64_t(int64_t(key) + a.direct_bias);
        if (a.direct == 2 && d >= a.direct_span) return -1; // outside the measured range (wave-uniform test first): the cold path
        if (SUB && dsub_width) return int(uint32_t(d) - my_subset * dsub_width); // this subset's half of the range (foreign keys never get here)
        return int((uint32_t(d) << rep_log2) | rep_lane);
    };
    auto flush_run = [&]() {
        if (SUB && foreign(run_key)) { // another workgroup's key: drop the run
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
                rcnt[j] = 0; rsum[j] = 0.0; rmn[j] = DBL_MAX; rmx[j] = -DBL_MAX; rnan[j] = false;
            }
            return;
        }
        // The hit path must stay minimal: random-key inputs flush once per row (guarding the lookup with a "table is full"
        // test cost them 14 %).  A rejected key is the cold path: it raises the workgroup flag (the tile loop then leaves
        // early and the host redoes the query partitioned) or, for small inputs, goes to the global table.
        int slot;
        if (a.direct) { // wave-uniform
            slot = direct_slot(run_key);
            // the final merge recognises a used slot by its count; only NULL-able values (a group of NULLs has count 0) need the
            // key word as the mark (every writer stores the same word)
            if (VNULL && slot >= 0) lkeys[slot] = run_key;
        } else
            slot = lds_find_or_insert_counted(lkeys, run_key, cap, a.lds_shift, &lds_used, a.lds_limit);
        int64_t gslot = 0;
        if (slot < 0) {
            if (!*lds_full) {
                *lds_full = 1;
                if (a.allow_partition) atomicOr(&flags[NQE_FLAG_NEED_PARTITION], 1); // more distinct keys than a workgroup table holds
            }
            gslot = a.allow_partition ? -1 : global_find_or_insert(g, run_key, flags);
        }
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            if (slot >= 0) {
                uint32_t o = uint32_t(j) * slots + uint32_t(slot);
                if (rcnt[j]) {
                    if (OWN_CNT || j == 0) atomicAdd(&lcnt[o], rcnt[j]);
                    if (nsum[j]) unsafeAtomicAdd(&lsum[o], rsum[j]);
                }
                if (rnan[j]) atomicOr(&lcnt[o], NAN_BIT);
                // read-before-atomic (see the general kernel)
                if (mmcol(j)) {
                    const uint32_t mo = mmo(j, uint32_t(slot));
                    if (rmn[j] < lmn[mo]) unsafeAtomicMin(&lmn[mo], rmn[j]);
                    if (rmx[j] > lmx[mo]) unsafeAtomicMax(&lmx[mo], rmx[j]);
                }
            } else if (gslot >= 0) {
                global_update(g, gslot, a.v0 + j, rcnt[j], rsum[j], true, f64_to_ord(rmn[j]), f64_to_ord(rmx[j]), !ML || mmcol(j), rnan[j]);
            }
            rcnt[j] = 0; rsum[j] = 0.0; rmn[j] = DBL_MAX; rmx[j] = -DBL_MAX; rnan[j] = false;
        }
    };

    const uint64_t *__restrict__ keyp = static_cast<const uint64_t *>(a.key_src.values);
    const uint64_t *__restrict__ predp = static_cast<const uint64_t *>(((PRED >= 2 && (PRED != 5 && PRED != 6)) || ((PRED == 5 || PRED == 6) && a.tree_need_pw)) ? a.pred_src.values : a.key_src.values);
    const uint64_t *__restrict__ valp[NVT];
    const uint64_t *__restrict__ kvalid = reinterpret_cast<const uint64_t *>(a.key_src.valid);
    const uint64_t *__restrict__ pvalid = reinterpret_cast<const uint64_t *>(PRED != 0 ? a.pred_src.valid : nullptr);
    const uint64_t *__restrict__ vvalid[NVT]; // VNULL: validity bitmaps of the value columns (word-padded), null = all valid
    int vdt[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) {
        valp[j] = static_cast<const uint64_t *>(a.val[j].values);
        vvalid[j] = reinterpret_cast<const uint64_t *>(a.val[j].valid);
        vdt[j] = a.val[j].dtype;
    }
    const uint64_t key_mask = a.key.aux[0].abs_lit - 1;
    const OpAux key_aux = a.key.aux[0];
    const bool key_signed = a.key.op_dtype[0] == NQE_INT64;
    const int64_t n = a.n, last = a.n - 1;

    // non-temporal loads — except under key subsets, where 2^k workgroups of one XCD read the same tiles: a plain load leaves the line in
    // that XCD's L2 for the other readers (tools/pair_bench.hip: a free-running pair 0.436 ms with non-temporal loads, 0.362 with plain
    // ones, one reader 0.254; the product at 6000 / 8000 groups 0.521 / 0.510 -> 0.483 / 0.488 ms: profiles/r06/ab_sub_plain_loads.txt)
    constexpr bool NT = !SUB; // non-temporal loads: -2..3 % (and the loop below keeps a prefetched second tile in flight: 3.24 -> 2.69 ms with the lean loop)
    struct Tile {
        uint64_t kw[TU], pw[SH1 ? 1 : TU], vw[SH1 ? 1 : NVL][SH1 ? 1 : TU];
        uint64_t vv[VNULL ? NVT : 1][TU]; // validity word of the wave's 64 rows
        uint64_t kpv[VNULL ? TU : 1];     // key validity AND predicate validity
    };
    // `base` is wave-uniform (the tile loop's control flow depends on nothing but uniform values, see `stream`): a tile that lies
    // wholly inside the table is addressed as scalar tile pointer + loop-invariant 32-bit lane offset — no address arithmetic on the
    // vector unit (the clamped 64-bit row index of the general form cost 8 VALU instructions per row: the kernel is issue-bound)
    const int64_t step = int64_t(AGG_BLOCK) * TU;
    uint32_t lane_row[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) lane_row[u] = uint32_t(u) * AGG_BLOCK + threadIdx.x;
    // rows of tile `base` that exist (wave-uniform, 0 … step)
    auto tile_rows = [&](int64_t base) {
        const int64_t r = n - base;
        return uint32_t(r < 0 ? 0 : (r > step ? step : r));
    };
    // (three value columns: when the first one is the key column itself — `count(id) … group by id % 3` — its word is the key word)
    const bool first_is_key = NVT == 3 && !FK && a.val_shares_key[0] && a.val[0].values == a.key_src.values; // wave-uniform
    auto vword = [&](const Tile &t, int j, int u) -> uint64_t {
        if (SH1) return t.kw[u];
        if (FK) return j == 0 ? t.kw[u] : t.vw[FK ? (j > 0 ? j - 1 : 0) : 0][SH1 ? 0 : u];
        if (NVT == 3 && j == 0 && first_is_key) return t.kw[u];
        return t.vw[SH1 ? 0 : j][SH1 ? 0 : u];
    };
    auto load_tile = [&](Tile &t, int64_t base) {
        if (NT && !VNULL && base + step <= n) {
            const uint64_t *__restrict__ kt = keyp + base;
            const uint64_t *__restrict__ pt = predp + (PRED == 2 ? (base >> fp.row_shift) : base);
            const uint64_t *__restrict__ vt[NVT];
#pragma unroll
            for (int j = 0; j < NVT; ++j) vt[j] = valp[j] + base;
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                t.kw[u] = __builtin_nontemporal_load(&kt[lane_row[u]]);
                // (a Boolean bitmap predicate: word (base + lane_row) >> 6 with base a multiple of the tile = of 64)
                if (PRED == 2) t.pw[SH1 ? 0 : u] = __builtin_nontemporal_load(&pt[lane_row[u] >> fp.row_shift]);
                if (PRED == 3) t.pw[SH1 ? 0 : u] = __builtin_nontemporal_load(&pt[lane_row[u]]);
                if (PRED == 4) t.pw[SH1 ? 0 : u] = a.conj.need_pw ? __builtin_nontemporal_load(&pt[lane_row[u]]) : 0ull; // wave-uniform
                if ((PRED == 5 || PRED == 6)) t.pw[SH1 ? 0 : u] = a.tree_need_pw ? __builtin_nontemporal_load(&pt[lane_row[u]]) : 0ull;
#pragma unroll
                for (int j = 0; j < NVT; ++j)
                    if (!SH1 && !(FK && j == 0) && !(NVT == 3 && j == 0 && first_is_key))
                        t.vw[SH1 ? 0 : (FK ? (j > 0 ? j - 1 : 0) : j)][SH1 ? 0 : u] = __builtin_nontemporal_load(&vt[j][lane_row[u]]);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            row = row < last ? row : last; // clamp: unconditional, in-bounds
            // (the compiler merges the load sequences of the two branches, so both must carry the same non-temporal hint — ordinary
            // loads here made it drop the hint from the pointer form as well)
            if (NT) {
                t.kw[u] = __builtin_nontemporal_load(&keyp[row]);
                if (PRED == 2) t.pw[SH1 ? 0 : u] = __builtin_nontemporal_load(&predp[row >> fp.row_shift]);
                if (PRED == 3) t.pw[SH1 ? 0 : u] = __builtin_nontemporal_load(&predp[row]);
                if (PRED == 4) t.pw[SH1 ? 0 : u] = a.conj.need_pw ? __builtin_nontemporal_load(&predp[row]) : 0ull; // wave-uniform
                if ((PRED == 5 || PRED == 6)) t.pw[SH1 ? 0 : u] = a.tree_need_pw ? __builtin_nontemporal_load(&predp[row]) : 0ull;
#pragma unroll
                for (int j = 0; j < NVT; ++j)
                    if (!SH1 && !(FK && j == 0)) t.vw[SH1 ? 0 : (FK ? (j > 0 ? j - 1 : 0) : j)][SH1 ? 0 : u] = __builtin_nontemporal_load(&valp[j][row]);
                if (VNULL) {
#pragma unroll
                    for (int j = 0; j < NVT; ++j) t.vv[j][u] = vvalid[j] ? vvalid[j][row >> 6] : ~0ull;
                    t.kpv[u] = (kvalid ? kvalid[row >> 6] : ~0ull) & (pvalid ? pvalid[row >> 6] : ~0ull);
                }
            } else {
                t.kw[u] = keyp[row];
                if (PRED == 2) t.pw[SH1 ? 0 : u] = predp[row >> fp.row_shift];
                if (PRED == 3) t.pw[SH1 ? 0 : u] = predp[row];
                if (PRED == 4) t.pw[SH1 ? 0 : u] = a.conj.need_pw ? predp[row] : 0ull; // wave-uniform
#pragma unroll
                for (int j = 0; j < NVT; ++j)
                    if (!SH1 && !(FK && j == 0)) t.vw[SH1 ? 0 : (FK ? (j > 0 ? j - 1 : 0) : j)][SH1 ? 0 : u] = valp[j][row];
            }
        }
    };
    // Rows of a thread whose keys differ inside one register tile (random keys): the run cache would flush once per row, each
    // flush a dependent LDS round trip (slot → read min/max → compare → atomics); the kernel was bound by those waits, not by
    // LDS-atomic throughput (tools/micro_bench.hip: the same update stream issued back to back runs 3x faster).  Such a tile
    // goes to the table directly, all its rows at once: TU slots, one batch of min/max reads, one wait, then the atomics.
    // (two value columns: the rows of a tile go in two halves — eight more min/max words in flight would not fit the registers)
    constexpr int BG = NVT == 1 ? (TU % AGG_U == 0 ? AGG_U : TU / 2) : (MM ? NQE_AGG_BG2 : (NVT == 2 ? 2 : 1));
    auto direct_rows = [&](const Tile &t, int64_t base, const bool (&pass)[TU], const uint64_t (&key)[TU]) {
        bool cold = false;
        int slot[TU];
#pragma unroll
        for (int g0 = 0; g0 < TU; g0 += BG) {
            uint64_t k0[BG];
            if (!a.direct) {
                // first probe of every row, issued together.  (Advancing all four probe sequences in lockstep, one slot of every
                // unresolved row per round, was slower: 0.34 -> 0.38 ms per 10^8 rows at 1000 groups, 0.43 -> 0.46 at 3000 — the
                // kernel is bound by instruction issue, not by the probes' latency.)
#pragma unroll
                for (int i = 0; i < BG; ++i) k0[i] = lkeys[uint32_t((key[g0 + i] * GOLD) >> a.lds_shift)];
            }
#pragma unroll
            for (int i = 0; i < BG; ++i) {
                const int u = g0 + i;
                slot[u] = -1;
                if (!pass[u]) continue;
                if (a.direct) {
                    slot[u] = direct_slot(key[u]);
                    if (VNULL && slot[u] >= 0) lkeys[slot[u]] = key[u];
                } else if (k0[i] == key[u] && key[u] != EMPTY_KEY) {
                    slot[u] = int(uint32_t((key[u] * GOLD) >> a.lds_shift));
                } else {
                    slot[u] = lds_find_or_insert_counted(lkeys, key[u], cap, a.lds_shift, &lds_used, a.lds_limit);
                }
                cold = cold || slot[u] < 0;
            }
            double cmn[MM ? NVT : 1][BG], cmx[MM ? NVT : 1][BG];
            if (MM) {
#pragma unroll
                for (int j = 0; j < NVT; ++j) {
                    if (!mmcol(j)) continue;
#pragma unroll
                    for (int i = 0; i < BG; ++i) {
                        const uint32_t o = mmo(j, uint32_t(slot[g0 + i] < 0 ? 0 : slot[g0 + i]));
                        cmn[MM ? j : 0][i] = lmn[o]; // read-before-atomic, all rows of the group in flight together
                        cmx[MM ? j : 0][i] = lmx[o];
                    }
                }
            }
            // the per-row path: one branch (the row takes part), the updates themselves predicated by ordered compares — a NaN fails
            // both and only raises the group's flag
#pragma unroll
            for (int j = 0; j < NVT; ++j) {
#pragma unroll
                for (int i = 0; i < BG; ++i) {
                    const int u = g0 + i;
                    if (slot[u] < 0) continue;
                    const int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
                    const double x = VF64 ? u2d(vword(t, j, u)) : word_as_f64(vword(t, j, u), vdt[j]);
                    const bool vb = VNULL ? bool((t.vv[VNULL ? j : 0][u] >> (row & 63)) & 1ull) : true;
                    if (!vb) continue; // a NULL value contributes nothing; its row has created the group above
                    const uint32_t o = uint32_t(j) * slots + uint32_t(slot[u]);
                    if (OWN_CNT || j == 0) atomicAdd(&lcnt[o], 1u);
                    if (nsum[j]) unsafeAtomicAdd(&lsum[o], x);
                    if (mmcol(j)) {
                        const uint32_t mo = mmo(j, uint32_t(slot[u]));
                        if (x < cmn[MM ? j : 0][i]) unsafeAtomicMin(&lmn[mo], x);
                        if (x > cmx[MM ? j : 0][i]) unsafeAtomicMax(&lmx[mo], x);
                    }
                    if (x != x) atomicOr(&lcnt[o], NAN_BIT);
                }
            }
        }
        if (cold) { // the table rejected a key (see flush_run): off the per-row path
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                if (!pass[u] || slot[u] >= 0) continue;
                if (!*lds_full) {
                    *lds_full = 1;
                    if (a.allow_partition) atomicOr(&flags[NQE_FLAG_NEED_PARTITION], 1);
                }
                const int64_t gslot = a.allow_partition ? -1 : global_find_or_insert(g, key[u], flags);
                if (gslot < 0) continue;
                const int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
#pragma unroll
                for (int j = 0; j < NVT; ++j) {
                    const double x = VF64 ? u2d(vword(t, j, u)) : word_as_f64(vword(t, j, u), vdt[j]);
                    const bool vb = VNULL ? bool((t.vv[VNULL ? j : 0][u] >> (row & 63)) & 1ull) : true;
                    if (!vb) continue;
                    const bool isn = x != x;
                    const uint64_t xo = f64_to_ord(x);
                    global_update(g, gslot, a.v0 + j, 1, x, true, isn ? ORD_MAX : xo, isn ? ORD_MIN : xo, true, isn);
                }
            }
        }
    };

    // Two row loops.  RUN: register accumulation while a thread's key repeats (sorted ids, `id % m` with m | 1024), where nothing
    // but a changed key touches LDS.  BATCH: every key of the tile up front, tiles whose keys differ go through direct_rows.
    // Hoisting the keys costs the run loop's best case 4-5 % (the headline kernel sits at the HBM/VALU knee; A/B on one box:
    // 2.36 -> 2.47 ms with the test on every tile, still +2-3 % with a per-tile mode branch inside one loop), so they are two
    // separate streaming loops and a wave re-picks between them every 64 tiles from the keys of the tile in hand: batch when a
    // quarter of its lanes see mixed keys (random keys: 2.90 -> 2.57 ms).
    // the interpreted predicates.  PRED = 5: tests with an arithmetic step under any and/or nesting (ConjPred's general form:
    // test-major straight-line code + truth table); PRED = 6: the stack machine's program
    auto eval_interpreted = [&](const Tile &t, bool (&res)[TU]) {
#ifdef NQE_TREE_TRIVIAL // diagnostics: the instance's skeleton without the predicate's work
#pragma unroll
        for (int u = 0; u < TU; ++u) res[u] = (t.kw[u] ^ vword(t, 0, u)) != 12345;
        return;
#endif
        if constexpr (!SH1) {
            if (PRED == 5) conj_general_tile<TU>(a.conj, t.kw, t.vw[0], t.pw, res);
            else tree_pred_eval<TU>(a.tree_prog, a.tree_n, t.kw, t.vw[0], t.pw, res);
        }
    };
    auto accumulate_row = [&](const Tile &t, int u, int64_t row, uint64_t key) {
        if (!run_live || key != run_key) {
            if (run_live) flush_run();
            run_key = key;
            run_live = true;
        }
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            double x = VF64 ? u2d(vword(t, j, u)) : word_as_f64(vword(t, j, u), vdt[j]);
            if (VNULL) {
                // a NULL value contributes nothing (count of non-null, Q10) but its row still creates the group:
                // branch-free — count += bit, sum += 0, and a NaN operand that min/max ignore and the flag skips
                const bool vb = (t.vv[VNULL ? j : 0][u] >> (row & 63)) & 1ull;
                rcnt[j] += vb ? 1u : 0u;
                rsum[j] += vb ? x : 0.0;
                rnan[j] = rnan[j] || (vb && x != x);
                const double xm = vb ? x : __builtin_nan("");
                if (mmcol(j)) {
                    rmn[j] = fmin(rmn[j], xm);
                    rmx[j] = fmax(rmx[j], xm);
                }
            } else {
                rcnt[j] += 1;
                rsum[j] += x;
                rnan[j] = rnan[j] || (x != x);
                if (mmcol(j)) {
                    rmn[j] = fmin(rmn[j], x); // NaN operand ignored
                    rmx[j] = fmax(rmx[j], x);
                }
            }
        }
    };
    auto process_run = [&](const Tile &t, int64_t base) {
        // interpreted keys are computed for the whole tile up front (operator-major); the built-in shapes stay inside the row loop,
        // where the compiler sinks them below the predicate (hoisting them cost the headline 1.5 %)
        uint64_t pvals[PRED == 3 ? TU : 1];
        if constexpr (PRED == 3 && !SH1) inline_keys<3, TU, NVT == 1>(a.pred, t.pw, reinterpret_cast<uint64_t (&)[TU]>(pvals), 0, no_aux(), false);
        bool tpass[(PRED == 5 || PRED == 6) ? TU : 1];
        if ((PRED == 5 || PRED == 6)) eval_interpreted(t, reinterpret_cast<bool (&)[TU]>(tpass));
        uint64_t keys[KEY == 3 ? TU : 1];
        if (KEY == 3) inline_keys<3, TU>(a.key, t.kw, reinterpret_cast<uint64_t (&)[TU]>(keys), key_mask, key_aux, key_signed);
        const uint32_t nrows = tile_rows(base);
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            bool pass = lane_row[u] < nrows;
            if (PRED == 3) pass = pass && pvals[PRED == 3 ? u : 0] != 0;
            else if ((PRED == 5 || PRED == 6)) pass = pass && tpass[(PRED == 5 || PRED == 6) ? u : 0];
            else if (PRED == 4) pass = pass && conj_pass<3, false>(a.conj, t.kw[u], vword(t, 0, u), t.pw[SH1 ? 0 : u]);
            else if (PRED != 0) {
                pass = pass && range_pass(fp, PRED == 1 ? t.kw[u] : pred_extract(fp, t.pw[SH1 ? 0 : u], row));
            }
            if (VNULL) pass = pass && ((t.kpv[u] >> (row & 63)) & 1ull);
            const uint64_t key = KEY == 3 ? keys[KEY == 3 ? u : 0] : inline_key<KEY>(a.key, t.kw[u], key_mask, key_aux, key_signed);
            if (!pass) continue;
            accumulate_row(t, u, row, key);
        }
    };
    auto tile_keys = [&](const Tile &t, uint64_t (&key)[TU]) {
        if (KEY == 3) inline_keys<3, TU>(a.key, t.kw, key, key_mask, key_aux, key_signed);
        else {
#pragma unroll
            for (int u = 0; u < TU; ++u) key[u] = inline_key<KEY>(a.key, t.kw[u], key_mask, key_aux, key_signed);
        }
    };
    auto process_batch = [&](const Tile &t, int64_t base) {
        uint64_t pvals[PRED == 3 ? TU : 1];
        if constexpr (PRED == 3 && !SH1) inline_keys<3, TU, NVT == 1>(a.pred, t.pw, reinterpret_cast<uint64_t (&)[TU]>(pvals), 0, no_aux(), false);
        bool tpass[(PRED == 5 || PRED == 6) ? TU : 1];
        if ((PRED == 5 || PRED == 6)) eval_interpreted(t, reinterpret_cast<bool (&)[TU]>(tpass));
        bool pass[TU];
        uint64_t key[TU];
        tile_keys(t, key);
        const uint32_t nrows = tile_rows(base);
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            int64_t row = base + int64_t(u) * AGG_BLOCK + threadIdx.x;
            pass[u] = lane_row[u] < nrows;
            if (PRED == 3) pass[u] = pass[u] && pvals[PRED == 3 ? u : 0] != 0;
            else if ((PRED == 5 || PRED == 6)) pass[u] = pass[u] && tpass[(PRED == 5 || PRED == 6) ? u : 0];
            else if (PRED == 4) pass[u] = pass[u] && conj_pass<3, false>(a.conj, t.kw[u], vword(t, 0, u), t.pw[SH1 ? 0 : u]);
            else if (PRED != 0) pass[u] = pass[u] && range_pass(fp, PRED == 1 ? t.kw[u] : pred_extract(fp, t.pw[SH1 ? 0 : u], row));
            if (VNULL) pass[u] = pass[u] && ((t.kpv[u] >> (row & 63)) & 1ull);
        }
        bool mixed = false; // keys of rows that fail the predicate take part: a false "mixed" costs nothing but the batch path
#pragma unroll
        for (int u = 1; u < TU; ++u) mixed = mixed || key[u] != key[0];
        if (SUB) {
#pragma unroll
            for (int u = 0; u < TU; ++u) pass[u] = pass[u] && !foreign(key[u]);
        }
        if (mixed) {
            if (run_live) {
                flush_run();
                run_live = false;
            }
            direct_rows(t, base, pass, key);
        } else {
#pragma unroll
            for (int u = 0; u < TU; ++u)
                if (pass[u]) accumulate_row(t, u, base + int64_t(u) * AGG_BLOCK + threadIdx.x, key[u]);
        }
    };

    const int64_t stride = int64_t(lanes) * step;
    int64_t base = int64_t(lane_id) * step;
    // up to 2 x `budget` tiles.  In: A holds tile `base` (< n).  Out: base >= n (done or abandoned), or A holds tile `base`.
    // (PRED = 6: ONE tile in flight — the stack machine's registers take the place of the prefetched tile's; the workgroup's 16
    // waves hide the loads' latency among themselves)
    // (three value columns: one tile is 6-8 KB per wave = 96-128 KB per CU in flight already; a second one spills ~30 VGPRs)
    constexpr bool PIPE = (PRED != 6 || NQE_TREE_PIPE) && (NVT != 3 || FK) && (!ML || NQE_ML_PIPE);
    auto stream = [&](auto &&process, Tile &A, int budget) {
        if constexpr (!PIPE) {
            for (int64_t it = 0; it < 2 * int64_t(budget); ++it) {
                process(A, base);
                base += stride;
                if (base >= n) return;
                load_tile(A, base);
                if ((int(it) & a.flag_check_mask) != a.flag_check_mask) continue;
                if (__builtin_amdgcn_readfirstlane(*lds_full) &&
                    (a.allow_partition || __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[NQE_FLAG_TABLE_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
                    base = n;
                    return;
                }
                if (a.allow_partition && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[NQE_FLAG_NEED_PARTITION], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    base = n;
                    return;
                }
            }
            return;
        }
        Tile B;
        for (int it = 0; it < budget; ++it) {
            load_tile(B, base + stride); // prefetch (clamped, always issued)
            process(A, base);
            base += stride;
            if (base >= n) return;
            load_tile(A, base + stride);
            process(B, base);
            base += stride;
            if (base >= n) return;
            // (The flag loads below — a FLAT load of the LDS word and a device-scope load — are waited for with s_waitcnt vmcnt(0), i.e.
            // behind tile A's prefetch issued a moment ago, and the wave then sits with nothing in flight for a round trip to L2.  Looking
            // only every 8th iteration helps the one-tile loop above — three value columns 4.59 -> 4.42 ms, four A/B rounds — but not
            // this one: headline 2.39-2.55 ms looking every iteration, 2.45-2.70 every 8th over four rounds on two boxes.  Fewer
            // requests in flight per CU read HBM faster here — the same effect as one workgroup per CU, aggregate.hip.)
            // (the flags are read through readfirstlane: every lane reads the same word, and the compiler must see that the loop's
            // control flow — hence `base` — is wave-uniform, or the tile pointers above turn into per-lane 64-bit arithmetic)
            if (__builtin_amdgcn_readfirstlane(*lds_full) &&
                (a.allow_partition || __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[NQE_FLAG_TABLE_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
                base = n; // the host redoes the query (partitioned path / larger table)
                return;
            }
            if (a.allow_partition && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[NQE_FLAG_NEED_PARTITION], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                base = n; // another workgroup's table overflowed: this attempt is abandoned anyway
                return;
            }
        }
    };
    if (base < n) {
        Tile A;
        load_tile(A, base);
        // not where registers are short: the VNULL variants (37 VGPRs spilled: 2.2x slower), two value columns, interpreted predicates
        constexpr bool CAN_BATCH = NQE_AGG_BATCH && !VNULL && PRED != 3 && (PRED != 6 && (PRED != 5 || (NQE_AGG_BATCH_TREE && KEY != 3 && NVT == 1))) && !(PRED == 2 && KEY == 3) && !(PRED == 4 && KEY == 3) &&
                                   (NVT == 1 || ((NQE_AGG_BATCH2 || !MM || ML) && KEY != 3 && PRED != 4));
        while (base < n) {
            bool batch = false; // wave-uniform
            if (CAN_BATCH) {
                uint64_t key[TU];
                tile_keys(A, key);
                bool mixed = false;
#pragma unroll
                for (int u = 1; u < TU; ++u) mixed = mixed || key[u] != key[0];
                batch = __popcll(__ballot(mixed)) >= 16;
            }
            if (CAN_BATCH && batch) stream(process_batch, A, 32);
            else stream(process_run, A, CAN_BATCH ? NQE_AGG_RUN_BUDGET : (1 << 30));
        }
    }
    if (run_live) flush_run();
    __syncthreads();
    // an abandoned attempt (the host re-runs the query partitioned) does not merge: 2048 slots x 512 workgroups of
    // device-scope atomics were two thirds of what the abandoned attempt cost
    if (a.allow_partition && __hip_atomic_load(&flags[NQE_FLAG_NEED_PARTITION], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (a.direct && rep_log2) {
        // fold the replicas of every key into replica 0 (one thread per key and value column; plain LDS reads and writes)
        const uint32_t R = 1u << rep_log2, nkeys = cap >> rep_log2;
        for (uint32_t w = threadIdx.x; w < nkeys * NVT; w += blockDim.x) {
            const uint32_t kslot = (w % nkeys) << rep_log2, o0 = (w / nkeys) * slots + kslot;
            const bool mmw = MM && (!ML || (w / nkeys) == uint32_t(NVT - 1)); // this value column has min / max words
            const uint32_t m0 = (ML ? 0u : (w / nkeys)) * slots + kslot;
            uint32_t c = lcnt[o0];
            double sm = lsum[o0], mn = mmw ? lmn[m0] : 0.0, mx = mmw ? lmx[m0] : 0.0;
            bool used = VNULL && lkeys[kslot] != EMPTY_KEY;
            uint64_t kw = VNULL ? lkeys[kslot] : 0;
            for (uint32_t r = 1; r < R; ++r) {
                const uint32_t cr = lcnt[o0 + r];
                c = ((c & ~NAN_BIT) + (cr & ~NAN_BIT)) | ((c | cr) & NAN_BIT);
                sm += lsum[o0 + r];
                if (mmw) {
                    mn = fmin(mn, lmn[m0 + r]);
                    mx = fmax(mx, lmx[m0 + r]);
                }
                if (VNULL && lkeys[kslot + r] != EMPTY_KEY) {
                    used = true;
                    kw = lkeys[kslot + r];
                }
            }
            lcnt[o0] = c;
            lsum[o0] = sm;
            if (mmw) {
                lmn[m0] = mn;
                lmx[m0] = mx;
            }
            if (VNULL && used && w < nkeys) lkeys[kslot] = kw;
        }
        __syncthreads();
    }
    if (!VNULL && a.direct && a.partials) {
        // the table leaves whole (AggArgs::partials): coalesced stores, no atomics; agg_fold_partials_kernel folds the workgroups' tables
        const uint32_t span = a.partial_span;
        const size_t cells = size_t(gridDim.x) * span;
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            double *__restrict__ ps = reinterpret_cast<double *>(a.partials) + size_t(j) * ((cells * 28 + 7) / 8) + size_t(blockIdx.x) * span;
            uint32_t *__restrict__ pc = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(a.partials) + size_t(j) * ((cells * 28 + 7) / 8) + 3 * cells) + size_t(blockIdx.x) * span;
            for (uint32_t ki = threadIdx.x; ki < span; ki += blockDim.x) {
                const uint32_t s = ki << rep_log2, o = uint32_t(j) * slots + s;
                uint32_t c = lcnt[o];
                if (!OWN_CNT && j > 0) c = (lcnt[s] & ~NAN_BIT) | (c & NAN_BIT);
                ps[ki] = lsum[o];
                ps[cells + ki] = mmcol(j) ? lmn[mmo(j, s)] : DBL_MAX;
                ps[2 * cells + ki] = mmcol(j) ? lmx[mmo(j, s)] : -DBL_MAX;
                pc[ki] = c;
            }
        }
        return;
    }
    for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
        uint64_t k;
        if (a.direct && (s & ((1u << rep_log2) - 1u)) && s != cap) continue; // replicas were folded into replica 0
        if (a.direct && !VNULL) {
            if (lcnt[s] == 0) continue;
            k = uint64_t(int64_t((s >> rep_log2) + ((SUB && dsub_width) ? my_subset * dsub_width : 0u)) - a.direct_bias);
        } else {
            k = lkeys[s];
            if (k == EMPTY_KEY) continue;
        }
        uint64_t key = (s == cap) ? EMPTY_KEY : k;
        int64_t gslot = global_find_or_insert(g, key, flags);
        if (gslot < 0) continue;
#pragma unroll
        for (int j = 0; j < NVT; ++j) {
            uint32_t o = uint32_t(j) * slots + s;
            uint32_t c = lcnt[o];
            if (!OWN_CNT && j > 0) c = (lcnt[s] & ~NAN_BIT) | (c & NAN_BIT);
            global_update(g, gslot, a.v0 + j, uint64_t(c & ~NAN_BIT), lsum[o], true, mmcol(j) ? f64_to_ord(lmn[mmo(j, s)]) : 0ull, mmcol(j) ? f64_to_ord(lmx[mmo(j, s)]) : 0ull,
                          mmcol(j), (c & NAN_BIT) != 0);
        }
    }
}

template <int PRED, int KEY, bool VNULL> FastKernel pick_fast_nv(int nv, bool vf64, bool sub, bool nomm, bool share) {
    if constexpr (!VNULL && PRED <= 1 && KEY != 3) {
        if (share && nv == 1 && !vf64 && !sub)
            return nomm ? agg_grouped_fast_kernel<PRED, KEY, 1, false, false, false, false, true> : agg_grouped_fast_kernel<PRED, KEY, 1, false, false, false, true, true>;
    }
    if constexpr (!VNULL && PRED <= 1 && KEY != 3) {
        if (nomm && nv == 3 && !sub && share) return agg_grouped_fast_kernel<PRED, KEY, 3, false, false, false, false, true>; // the first column is the key column
        if (nomm && nv == 3 && !sub)
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 3, true, false, false, false> : agg_grouped_fast_kernel<PRED, KEY, 3, false, false, false, false>;
        if (nomm && nv == 2 && !sub)
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 2, true, false, false, false> : agg_grouped_fast_kernel<PRED, KEY, 2, false, false, false, false>;
    }
    if constexpr (!VNULL && PRED <= 1 && KEY != 3) {
        // three columns, min / max asked of the LAST one only (the host checks): src/main.rs:36-40's shape in one pass
        if (!nomm && nv == 3 && !sub) {
            if (share) return agg_grouped_fast_kernel<PRED, KEY, 3, false, false, false, true, true>;
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 3, true, false, false, true> : agg_grouped_fast_kernel<PRED, KEY, 3, false, false, false, true>;
        }
    }
    if constexpr (!VNULL && PRED <= 1 && KEY != 3) {
        // one value column nobody asks min / max of (count / sum / avg): no min / max arrays — 12 bytes per slot of a direct-mapped table,
        // 13632 keys in one workgroup table (round 6)
        if (nomm && nv == 1 && !sub && !share)
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 1, true, false, false, false> : agg_grouped_fast_kernel<PRED, KEY, 1, false, false, false, false>;
    }
    if constexpr (!VNULL && PRED <= 1 && KEY == 0) {
        // … and the two halves of a measured key range in two such tables (two key subsets): 2 x 13632 keys
        if (nomm && nv == 1 && sub && !share)
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 1, true, false, true, false> : agg_grouped_fast_kernel<PRED, KEY, 1, false, false, true, false>;
    }
    if (nv == 3) return nullptr; // (three columns: the instances above only)
    if (sub) {
        // (PRED 4: a query that outgrows one table continues with a materialised predicate; its slice is built without validity only)
        if constexpr (VNULL || PRED >= 4) return nullptr;
        else {
            if (nv != 1) return nullptr;
            return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 1, true, false, true> : agg_grouped_fast_kernel<PRED, KEY, 1, false, false, true>;
        }
    }
    if constexpr (PRED >= 5 && KEY == 3) return nullptr; // (a tree predicate under an interpreted key: registers)
    else if (nv == 1) return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 1, true, VNULL> : agg_grouped_fast_kernel<PRED, KEY, 1, false, VNULL>;
    if constexpr (VNULL || PRED >= 5) {
        return nullptr; // nullable sources take one value column per pass (aggregate.hip): the two-column variants spilled 60-135 VGPRs
    } else {
        return vf64 ? agg_grouped_fast_kernel<PRED, KEY, 2, true, VNULL> : agg_grouped_fast_kernel<PRED, KEY, 2, false, VNULL>;
    }
}
template <int PRED, bool VNULL> FastKernel pick_fast_key(int key, int nv, bool vf64, bool sub, bool nomm, bool share) {
    switch (key) {
    case 0: return pick_fast_nv<PRED, 0, VNULL>(nv, vf64, sub, nomm, share);
    case 1: return pick_fast_nv<PRED, 1, VNULL>(nv, vf64, sub, nomm, share);
    case 2: return pick_fast_nv<PRED, 2, VNULL>(nv, vf64, sub, nomm, share);
    default: return pick_fast_nv<PRED, 3, VNULL>(nv, vf64, sub, nomm, share);
    }
}

} // namespace
} // namespace agg
} // namespace nqe
