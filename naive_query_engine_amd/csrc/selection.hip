// selection.hip — SelectionPlan::execute (reference: src/physical_plan/selection.rs:58-107 and
// the row-at-a-time `build_array_by_predicate!` :34-51) as stable stream compaction.
//
// Layout: the predicate becomes a KEEP bitmap (1 bit/row: pred true OR pred NULL — a NULL
// predicate emits a NULL row, quirk Q4) and a per-tile popcount (tile = 4096 rows = 64 bitmap
// words).  An exclusive scan of the tile counts gives each tile its output base; inside a tile a
// wave turns the 64 word popcounts into word offsets with one wave scan, and each lane's rank
// within its word is popc(word & lanemask_lt).  Output order is therefore exactly input order
// (the reference's builder appends in row order).
//
// HBM traffic per payload column: 8 B/row read (coalesced, unconditional) + 8 B per kept row
// written + 1/8 B/row of bitmap — i.e. the algorithmic minimum of SURVEY §8d C2.
#include "device_utils.hpp"
#include "nqe_internal.hpp"

namespace nqe {

namespace {

#ifndef NQE_COMPACT_CHUNK_WORDS
#define NQE_COMPACT_CHUNK_WORDS 16
#endif
constexpr int COMPACT_CHUNK_WORDS = NQE_COMPACT_CHUNK_WORDS; // keep words (x 64 rows) per chunk of the striped compaction
constexpr int SEL_B = 8; // 8-byte loads per lane per step of the streaming selection kernels (two steps are unrolled together: 16 in
                         // flight; an explicit 16 without the unroll measured slower for the compaction, faster for the mask kernel)

__device__ __forceinline__ uint64_t load_bitmap_word(const uint8_t *bm, int64_t w, int64_t nbits) {
    // reads word w of an LSB-first bitmap holding nbits bits; never reads past ceil(nbits/8) bytes
    int64_t nbytes = (nbits + 7) >> 3;
    int64_t b0 = w * 8;
    if (b0 + 8 <= nbytes) return *reinterpret_cast<const uint64_t *>(bm + b0);
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k)
        if (b0 + k < nbytes) v |= uint64_t(bm[b0 + k]) << (8 * k);
    return v;
}

// keep = (~pvalid | pbits) & in-range ; one wave per tile, one word per lane
__global__ void __launch_bounds__(256) keep_from_pred_kernel(const uint8_t *pbits, const uint8_t *pvalid, int64_t pred_len,
                                                             int64_t n, int64_t ntiles, uint64_t *keep,
                                                             uint64_t *pvalid_out, uint32_t *tile_counts) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t nwords = (n + 63) / 64;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        int64_t w = tile * TILE_WORDS + lane_id();
        uint64_t k = 0, pv = ~0ull;
        if (w < nwords) {
            uint64_t pb = load_bitmap_word(pbits, w, pred_len);
            if (pvalid) pv = load_bitmap_word(pvalid, w, pred_len);
            int64_t rem = n - w * 64;
            uint64_t range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
            k = (~pv | pb) & range;
            keep[w] = k;
            if (pvalid_out) pvalid_out[w] = pv;
        }
        uint32_t c = __popcll(k), tot;
        (void)wave_exclusive_scan(c, tot);
        if (lane_id() == 0) tile_counts[tile] = tot;
    }
}

// predicate = SimpleExpr over one streamed column; wave per tile, 64 rows per step (any column type, validity).  A range test over a
// plain 8-byte column takes keep_from_range_tile_kernel below.
__global__ void __launch_bounds__(256) keep_from_simple_kernel(const void *values, const uint8_t *valid, SimpleExpr e, int64_t n, int64_t ntiles, uint64_t *keep, uint64_t *pvalid_out,
                                                               uint32_t *tile_counts, int *flags) {
    const int waves_per_block = blockDim.x / 64;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
        {
#pragma unroll 4
            for (int k = 0; k < TILE_WORDS; ++k) {
                int64_t row = row0 + int64_t(k) * 64 + lane_id();
                bool in = row < n;
                uint64_t v = in ? load_word(values, e.src_dtype, row) : 0;
                bool ok = in && (valid ? get_bit(valid, row) : true);
                bool r = eval_simple(e, v, ok, flags) != 0;
                uint64_t kw = __ballot(in && (!ok || r));
                uint64_t vw = __ballot(ok);
                if (row0 + int64_t(k) * 64 < n && lane_id() == 0) {
                    keep[tile * TILE_WORDS + k] = kw;
                    if (pvalid_out) pvalid_out[tile * TILE_WORDS + k] = vw;
                }
                total += __popcll(kw);
            }
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

// The range-test form.  Round 5 (tools/compact_bench.hip, 10^8 rows, keep and compaction alternating so that nothing is cache-warm): one 256-thread workgroup
// per 4096-row tile, sixteen rows per lane all in flight, the tile's count written ONCE by the workgroup — no zeroing pass, no atomic
// per 512-row chunk: 0.148-0.150 ms -> 0.131-0.133 ms per 0.8 GB column (6.05 TB/s; the atomics were the whole difference — the strided
// form without them measured the same, and a 1024-thread two-tile pipelined form like the aggregate's stream was slower, 0.156-0.166 ms).
template <int RANGE>
__global__ void __launch_bounds__(256) keep_from_range_tile_kernel(const uint64_t *__restrict__ words, FastPred fp, int64_t n, int64_t ntiles, uint64_t *keep, uint32_t *tile_counts) {
    static_assert(TILE_ROWS == 4096, "four waves x sixteen keep words");
    __shared__ uint32_t wtot[4];
    const int lane = lane_id(), wave = int(threadIdx.x) >> 6;
    const int64_t last = n - 1;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TILE_ROWS + wave * 1024 + lane;
        uint64_t v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + r * 64;
            v[r] = __builtin_nontemporal_load(&words[row < last ? row : last]);
        }
        uint32_t total = 0;
        uint64_t mine = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + r * 64;
            const uint64_t kw = __ballot(row < n && range_pass(fp, RANGE == 2 ? f64_order_map(fp, v[r]) : v[r]));
            if (lane == r) mine = kw;
            total += __popcll(kw);
        }
        if (lane < 16 && (tile * TILE_WORDS + wave * 16 + lane) * 64 < n) keep[tile * TILE_WORDS + wave * 16 + lane] = mine;
        if (lane == 0) wtot[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

// keep mask of `A and B [and …]` / `A or B [or …]`, up to four range tests over NC <= 4 plain 8-byte columns (ConjTest::src = the
// column's slot): the WHERE clause's usual shape in one streaming pass over its column(s), without the expression machine's
// Boolean column
struct ConjCols {
    const uint64_t *w[CONJ_MAX];
};
// STRIPED: the rows go to the waves in chunks of 2 x B keep words (chunk c to wave c mod #waves; per-tile totals through one atomic
// per chunk into zeroed tile_counts) instead of one 4096-row tile per wave (the only form instantiated since round 6)
template <int NC, bool STRIPED>
__global__ void __launch_bounds__(256) keep_from_conj_kernel(ConjCols cols, ConjPred c, int64_t n, int64_t ntiles, uint64_t *keep, uint32_t *tile_counts) {
    constexpr int B = NC <= 2 ? SEL_B : SEL_B / 2; // words in flight per lane stay at 8-16
    constexpr int CW = STRIPED ? 2 * B : TILE_WORDS, CPT = TILE_WORDS / CW;
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    for (int64_t chunk = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; chunk < ntiles * CPT;
         chunk += int64_t(gridDim.x) * waves_per_block) {
        const int64_t tile = chunk / CPT;
        const int kfirst = int(chunk % CPT) * CW;
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
#pragma unroll 2
        for (int k0 = kfirst; k0 < kfirst + CW; k0 += B) {
            uint64_t w[CONJ_MAX][B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                row = row < last ? row : last;
#pragma unroll
                for (int q = 0; q < CONJ_MAX; ++q) w[q][k] = q < NC ? __builtin_nontemporal_load(&cols.w[q][row]) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                const uint64_t kw = __ballot(row < n && conj_pass<NC>(c, w[0][k], w[1][k], w[2][k], w[3][k]));
                if (row0 + int64_t(k0 + k) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0 + k] = kw;
                total += __popcll(kw);
            }
        }
        if (STRIPED) {
            if (lane_id() == 0 && total) atomicAdd(&tile_counts[tile], total);
        } else if (lane_id() == 0)
            tile_counts[tile] = total;
    }
}

// Stable compaction of one column (or of a SimpleExpr evaluated on it) by the keep bitmap.
// Word k of the tile lives in lane k; it is broadcast through the scalar unit (readlane).
// PLAINW: 8-byte source without validity and a predicate without nulls → values only.
// GATHER: the source row of output row is gidx[row] (join: build row of the matching probe row).
template <bool EXPR, bool PLAINW, bool GATHER>
__global__ void __launch_bounds__(256) compact_kernel(const void *src_values, int src_dtype, const uint8_t *src_valid,
                                                      const uint32_t *gidx, SimpleExpr e, const uint64_t *keep, const uint64_t *pvalid,
                                                      const uint64_t *tile_offsets, int64_t n, int64_t ntiles,
                                                      uint64_t *out_words, uint8_t *out_bool_bytes,
                                                      uint8_t *out_valid_bytes, int *flags) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t nwords = (n + 63) / 64;
    const int64_t last = n - 1;
    const uint64_t *__restrict__ words = static_cast<const uint64_t *>(src_values);
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        int64_t w = tile * TILE_WORDS + lane_id();
        uint64_t my_word = w < nwords ? keep[w] : 0;
        uint64_t my_pv = (!PLAINW && pvalid && w < nwords) ? pvalid[w] : ~0ull;
        uint32_t tot;
        uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        if (tot == 0) continue; // nothing kept in these 4096 rows (wave-uniform): their source words are not read at all — a filter
                                // on clustered data (sorted ids, time ranges) then moves only the kept part of every other column
        const uint64_t base = tile_offsets[tile];
        if (PLAINW) {
#pragma unroll 2
            for (int k0 = 0; k0 < TILE_WORDS; k0 += SEL_B) {
                uint64_t v[SEL_B];
                if (GATHER) {
                    uint32_t gi[SEL_B];
#pragma unroll
                    for (int k = 0; k < SEL_B; ++k) {
                        int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                        gi[k] = gidx[row < last ? row : last];
                    }
#pragma unroll
                    for (int k = 0; k < SEL_B; ++k) {
                        bool kept = (bcast64(my_word, k0 + k) >> lane_id()) & 1;
                        v[k] = words[kept ? gi[k] : 0u]; // gather (build side is cache resident)
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SEL_B; ++k) {
                        int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                        v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]); // unconditional, coalesced, streamed once
                    }
                }
#pragma unroll
                for (int k = 0; k < SEL_B; ++k) {
                    uint64_t word = bcast64(my_word, k0 + k);
                    uint32_t off = bcast32(my_off, k0 + k);
                    if ((word >> lane_id()) & 1) {
                        uint64_t x = EXPR ? eval_simple(e, v[k], true, flags) : v[k];
                        __builtin_nontemporal_store(x, &out_words[base + off + __popcll(word & lanemask_lt())]);
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < TILE_WORDS; ++k) {
                uint64_t word = bcast64(my_word, k);
                uint64_t pvw = bcast64(my_pv, k);
                uint32_t off = bcast32(my_off, k);
                int64_t row = (tile * TILE_WORDS + k) * 64 + lane_id();
                bool in = row < n;
                bool kept = (word >> lane_id()) & 1;
                int64_t srow = row;
                if (GATHER) srow = (in && kept) ? int64_t(gidx[row]) : 0;
                uint64_t v = (in && (!GATHER || kept)) ? load_word(src_values, src_dtype, srow) : 0;
                if (kept) {
                    bool ok = ((pvw >> lane_id()) & 1) && (src_valid ? get_bit(src_valid, srow) : true);
                    if (EXPR) v = eval_simple(e, v, ok, flags);
                    uint64_t pos = base + off + __popcll(word & lanemask_lt());
                    if (out_words) out_words[pos] = ok ? v : 0;
                    if (out_bool_bytes) out_bool_bytes[pos] = (ok && v) ? 1 : 0;
                    if (out_valid_bytes) out_valid_bytes[pos] = ok ? 1 : 0;
                }
            }
        }
    }
}

// emitted-row list: out[pos] = source row (or -1 when the predicate was NULL there → NULL row, quirk Q4)
__global__ void __launch_bounds__(256) kept_rows_kernel(const uint64_t *keep, const uint64_t *pvalid, const uint64_t *tile_offsets, int64_t n,
                                                        int64_t ntiles, int64_t *out) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t nwords = (n + 63) / 64;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        int64_t w = tile * TILE_WORDS + lane_id();
        uint64_t my_word = w < nwords ? keep[w] : 0;
        uint64_t my_pv = (pvalid && w < nwords) ? pvalid[w] : ~0ull;
        uint32_t tot;
        uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        const uint64_t base = tile_offsets[tile];
        for (int k = 0; k < TILE_WORDS; ++k) {
            uint64_t word = bcast64(my_word, k);
            uint64_t pvw = bcast64(my_pv, k);
            uint32_t off = bcast32(my_off, k);
            if ((word >> lane_id()) & 1) {
                int64_t row = (tile * TILE_WORDS + k) * 64 + lane_id();
                out[base + off + __popcll(word & lanemask_lt())] = ((pvw >> lane_id()) & 1) ? row : -1;
            }
        }
    }
}

} // namespace

KeepMask finish_mask(nqe_ctx *ctx, KeepMask km, BufRef tile_counts) {
    km.tile_offsets = dev_alloc(ctx, size_t(km.ntiles + 1) * 8);
    exclusive_scan_u32_to_u64(ctx, (const uint32_t *)tile_counts->ptr, (uint64_t *)km.tile_offsets->ptr, km.ntiles);
    km.total = int64_t(read_scalar(ctx, (const uint64_t *)km.tile_offsets->ptr + km.ntiles));
    return km;
}

KeepMask build_keep_mask(nqe_ctx *ctx, const DevColumn &pred, int64_t n_rows) {
    KeepMask km;
    km.n = std::min<int64_t>(pred.length, n_rows); // iterator zip truncates (quirk Q3)
    km.ntiles = (km.n + TILE_ROWS - 1) / TILE_ROWS;
    int64_t nwords = (km.n + 63) / 64;
    km.keep = dev_alloc(ctx, size_t(nwords) * 8 + 8);
    if (pred.validity) km.pvalid = dev_alloc(ctx, size_t(nwords) * 8 + 8);
    BufRef counts = dev_alloc(ctx, size_t(km.ntiles + 1) * 4);
    if (km.ntiles)
        launch(ctx, "keep_from_pred", keep_from_pred_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, pred.bits(),
               pred.valid(), pred.length, km.n, km.ntiles, (uint64_t *)km.keep->ptr,
               km.pvalid ? (uint64_t *)km.pvalid->ptr : nullptr, (uint32_t *)counts->ptr);
    return finish_mask(ctx, km, counts);
}

KeepMask build_keep_mask_simple(nqe_ctx *ctx, const nqe_table *in, const SimpleExpr &pred) {
    const DevColumn &c = in->cols[size_t(pred.col)];
    KeepMask km;
    km.n = in->rows;
    km.ntiles = (km.n + TILE_ROWS - 1) / TILE_ROWS;
    int64_t nwords = (km.n + 63) / 64;
    km.keep = dev_alloc(ctx, size_t(nwords) * 8 + 8);
    if (c.validity) km.pvalid = dev_alloc(ctx, size_t(nwords) * 8 + 8);
    BufRef counts = dev_alloc(ctx, size_t(km.ntiles + 1) * 4);
    FastPred fp{};
    const bool range = is_word_type(c.dtype) && !c.validity && make_fast_pred(pred, &fp);
    if (km.ntiles) {
        // a range test over a plain 8-byte column: a tile per 256-thread workgroup pass, counts written once, no atomics (round 5; the
        // striped form with an atomic per chunk and the one-tile-per-wave form it replaced: profiles/r05/compact_bench.txt)
        if (range) {
            dim3 tgrid(stream_grid(ctx, km.ntiles, 1, 8)), block(256);
            if (fp.fmask) launch(ctx, "keep_from_simple", keep_from_range_tile_kernel<2>, tgrid, block, 0, c.words(), fp, km.n, km.ntiles, (uint64_t *)km.keep->ptr, (uint32_t *)counts->ptr);
            else launch(ctx, "keep_from_simple", keep_from_range_tile_kernel<1>, tgrid, block, 0, c.words(), fp, km.n, km.ntiles, (uint64_t *)km.keep->ptr, (uint32_t *)counts->ptr);
        } else
            launch(ctx, "keep_from_simple", keep_from_simple_kernel, dim3(stream_grid(ctx, km.ntiles, 4, 8)), dim3(256), 0, (const void *)c.values->ptr, c.valid(), pred,
                   km.n, km.ntiles, (uint64_t *)km.keep->ptr, km.pvalid ? (uint64_t *)km.pvalid->ptr : nullptr,
                   (uint32_t *)counts->ptr, ctx->d_flags);
    }
    return finish_mask(ctx, km, counts);
}

KeepMask build_keep_mask_conj(nqe_ctx *ctx, const nqe_table *in, ConjPred c, const int *cols) {
    KeepMask km;
    km.n = in->rows;
    km.ntiles = (km.n + TILE_ROWS - 1) / TILE_ROWS;
    const int64_t nwords = (km.n + 63) / 64;
    km.keep = dev_alloc(ctx, size_t(nwords) * 8 + 8);
    BufRef counts = dev_alloc(ctx, size_t(km.ntiles + 1) * 4);
    // the distinct tested columns, each loaded once per row
    ConjCols cc{};
    int slot_col[CONJ_MAX], nc = 0;
    for (int t = 0; t < c.n; ++t) {
        int q = 0;
        while (q < nc && slot_col[q] != cols[t]) ++q;
        if (q == nc) {
            slot_col[nc] = cols[t];
            cc.w[nc++] = in->cols[size_t(cols[t])].words();
        }
        c.t[t].src = q;
    }
    if (km.ntiles) {
        NQE_HIP_CHECK(hipMemsetAsync(counts->ptr, 0, size_t(km.ntiles + 1) * 4, ctx->stream));
        auto k = nc == 1 ? keep_from_conj_kernel<1, true> : nc == 2 ? keep_from_conj_kernel<2, true> : nc == 3 ? keep_from_conj_kernel<3, true> : keep_from_conj_kernel<4, true>;
        launch(ctx, "keep_from_conj", k, dim3(stream_grid(ctx, km.ntiles * 4, 4)), dim3(256), 0, cc, c, km.n, km.ntiles, (uint64_t *)km.keep->ptr, (uint32_t *)counts->ptr);
    }
    return finish_mask(ctx, km, counts);
}

const int64_t *kept_rows(nqe_ctx *ctx, const KeepMask &km) {
    if (!km.kept_idx) {
        km.kept_idx = dev_alloc(ctx, size_t(km.total) * 8 + 8);
        if (km.ntiles && km.total > 0)
            launch(ctx, "kept_rows", kept_rows_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, (const uint64_t *)km.keep->ptr,
                   km.pvalid ? (const uint64_t *)km.pvalid->ptr : nullptr, (const uint64_t *)km.tile_offsets->ptr, km.n, km.ntiles,
                   (int64_t *)km.kept_idx->ptr);
    }
    return (const int64_t *)km.kept_idx->ptr;
}


// compact_kernel's PLAINW form with the rows striped over the waves in 512-row chunks (SEL_B keep words; chunk c goes to wave
// c mod #waves) instead of one 4096-row tile per wave — at any moment the chip reads one compact window of the column (the same
// change made keep_from_range 17 % faster).  A wave re-derives the offsets of its chunk from the tile's 64 keep words (one
// coalesced 512-byte read that hits L2, one wave scan).
template <bool EXPR>
__global__ void __launch_bounds__(256) compact_strided_kernel(const uint64_t *__restrict__ words, SimpleExpr e, const uint64_t *keep, const uint64_t *tile_offsets,
                                                              int64_t n, int64_t ntiles, uint64_t *out_words, int *flags) {
    constexpr int CW = COMPACT_CHUNK_WORDS, CPT = TILE_WORDS / CW; // keep words per chunk, chunks per tile
    const int64_t nwords = (n + 63) / 64, last = n - 1, n_chunks = ntiles * CPT;
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t tile = chunk / CPT;
        const int c0 = int(chunk % CPT) * CW;
        const uint64_t base = tile_offsets[tile];
        if (tile_offsets[tile + 1] == base) continue; // (wave-uniform, two scalar loads) nothing kept in the whole tile: clustered data
        const int64_t w = tile * TILE_WORDS + lane_id();
        const uint64_t my_word = w < nwords ? keep[w] : 0;
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
        for (int k0 = c0; k0 < c0 + CW; k0 += SEL_B) {
            uint64_t kw[SEL_B];
            bool any = false;
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                kw[k] = bcast64(my_word, k0 + k);
                any = any || kw[k] != 0;
            }
            if (!any) continue; // (wave-uniform) nothing kept in these 512 rows: their source words are not read
            uint64_t v[SEL_B];
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                const int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]);
            }
#pragma unroll
            for (int k = 0; k < SEL_B; ++k) {
                const uint32_t off = bcast32(my_off, k0 + k);
                if ((kw[k] >> lane_id()) & 1) {
                    const uint64_t x = EXPR ? eval_simple(e, v[k], true, flags) : v[k];
                    __builtin_nontemporal_store(x, &out_words[base + off + __popcll(kw[k] & lanemask_lt())]);
                }
            }
        }
    }
}

// The PLAINW compaction at stream speed (round 5).  compact_strided_kernel stores straight from the registers: a wave's store covers
// popcount(keep word) x 8 bytes — ~256 B at 50 % selectivity, starting anywhere — so every store instruction touches a partial line on
// each side, and a chunk's neighbours are written by other waves, often on another XCD (another L2: the shared line reaches memory
// twice, each time partially): 4.2-4.7 TB/s of mixed traffic over random ids where a plain copy reaches 5.5-6.  Here a 512-thread
// workgroup takes a whole 4096-row tile: wave w the keep words 8 w .. 8 w + 7 (offsets from one scan of the tile's 64 keep words, as
// before), the kept words are STAGED in LDS at their rank within the tile, and the tile's T kept words leave as whole-wave 512-byte
// stores aligned to the 128-byte lines of the output (the first store is shifted by base mod 16 words) — two partial lines per tile
// instead of two per store.  Two workgroups per CU overlap their barrier phases.  Tiles without kept rows are skipped on two scalar
// loads (clustered data).  tools/compact_bench.hip (10^8 rows, 50 % kept): random ids 0.256 -> 0.195 ms (6.1 TB/s), sorted ids 0.140 =
// 0.140; 256 / 512 / 1024 threads, 2-8 workgroups per CU, a prefetched second register tile and plain instead of non-temporal stores
// were all measured there — prefetching bought nothing (±1 %), plain stores cost 7-10 %, 1024 threads 8-30 %.
// EX: 0 the column itself; 1 any SimpleExpr (interpreted per row: +12 % kernel time); 2 `x * mul + add` in wrapping 64-bit integers —
// `col + lit`, `col - lit`, `lit - col`, `col * lit` (the projection of C2, `age + 100`) — as straight-line code.
template <int EX>
__global__ void __launch_bounds__(512) compact_staged_kernel(const uint64_t *__restrict__ words, SimpleExpr e, uint64_t mul, uint64_t add, const uint64_t *__restrict__ keep,
                                                             const uint64_t *__restrict__ tile_offsets, int64_t n, int64_t ntiles, uint64_t *__restrict__ out_words, int *flags) {
    static_assert(TILE_WORDS == 64, "one keep word of the tile per lane");
    constexpr int R = TILE_WORDS / 8; // keep words (64-row groups) per wave: 8 waves per tile
    __shared__ uint64_t stage[TILE_ROWS];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int64_t nwords = (n + 63) / 64, last = n - 1;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile_offsets[tile];
        const uint32_t T = uint32_t(tile_offsets[tile + 1] - base);
        if (T == 0) continue; // (workgroup-uniform)
        const int64_t w = tile * TILE_WORDS + lane;
        const uint64_t my_word = w < nwords ? keep[w] : 0;
        uint64_t v[R];
        const int64_t row0 = (tile * TILE_WORDS + int64_t(wave) * R) * 64 + lane;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t row = row0 + int64_t(k) * 64;
            v[k] = __builtin_nontemporal_load(&words[row < last ? row : last]);
        }
        uint32_t tot;
        const uint32_t my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint64_t kw = bcast64(my_word, wave * R + k);
            const uint32_t off = bcast32(my_off, wave * R + k);
            if ((kw >> lane) & 1) stage[off + __popcll(kw & lanemask_lt())] = EX == 1 ? eval_simple(e, v[k], true, flags) : EX == 2 ? v[k] * mul + add : v[k];
        }
        __syncthreads();
        const int head = int(base & 15); // words between the preceding line boundary and the tile's first output word
        for (int j = int(threadIdx.x) - head; j < int(T); j += 512)
            if (j >= 0) __builtin_nontemporal_store(stage[j], &out_words[base + uint64_t(j)]);
        __syncthreads();
    }
}

// `x * mul + add` (wrapping) for a one-step Int64 / UInt64 SimpleExpr with a literal: + - * only (division and modulus can fault)
static bool affine_form(const SimpleExpr &e, uint64_t *mul, uint64_t *add) {
    if (e.nops != 1 || !(e.op_dtype[0] == NQE_INT64 || e.op_dtype[0] == NQE_UINT64) || !(e.src_dtype == NQE_INT64 || e.src_dtype == NQE_UINT64)) return false;
    const uint64_t lit = e.lit[0];
    switch (e.op[0]) {
    case NQE_OP_PLUS: *mul = 1; *add = lit; return true;
    case NQE_OP_MINUS:
        if (e.lit_left[0]) { *mul = ~uint64_t(0); *add = lit; } // lit - x
        else { *mul = 1; *add = uint64_t(0) - lit; }
        return true;
    case NQE_OP_MULTIPLY: *mul = lit; *add = 0; return true;
    default: return false;
    }
}

static DevColumn run_compact(nqe_ctx *ctx, const DevColumn &src, const SimpleExpr *e, int out_dtype, const KeepMask &km,
                             const uint32_t *gidx = nullptr) {
    if (src.dtype == NQE_UTF8 && !e && !gidx) // StringBuilder path of selection.rs:82-97: gather by the emitted-row list
        return take_utf8(ctx, src, kept_rows(ctx, km), km.total, km.pvalid != nullptr);
    if (!(is_word_type(src.dtype) || src.dtype == NQE_BOOLEAN))
        fail(NQE_ERR_NOT_SUPPORTED, "unimplemented!() column type in selection (selection.rs:98)");
    const int64_t m = km.total;
    const bool need_valid = src.validity != nullptr || km.pvalid != nullptr;
    const bool bool_out = out_dtype == NQE_BOOLEAN;
    DevColumn out = bool_out ? make_bool_column(ctx, m, need_valid) : make_word_column(ctx, out_dtype, m, need_valid);
    BufRef bool_bytes, valid_bytes;
    if (bool_out) bool_bytes = dev_alloc(ctx, size_t(m) + 8);
    if (need_valid) valid_bytes = dev_alloc(ctx, size_t(m) + 8);
    SimpleExpr dummy;
    std::memset(&dummy, 0, sizeof(dummy));
    if (km.ntiles && km.total > 0) { // nothing kept: nothing to read or write (and a gather source may be empty)
        dim3 grid(stream_grid(ctx, km.ntiles, 4, 8)), block(256);
        const void *sv = (const void *)src.values->ptr;
        const uint64_t *kp = (const uint64_t *)km.keep->ptr;
        const uint64_t *pv = km.pvalid ? (const uint64_t *)km.pvalid->ptr : nullptr;
        const uint64_t *to = (const uint64_t *)km.tile_offsets->ptr;
        uint64_t *ow = bool_out ? nullptr : (uint64_t *)out.values->ptr;
        uint8_t *ob = bool_out ? (uint8_t *)bool_bytes->ptr : nullptr;
        uint8_t *ov = need_valid ? (uint8_t *)valid_bytes->ptr : nullptr;
        const bool plainw = is_word_type(src.dtype) && !need_valid && !bool_out;
        const SimpleExpr &ex = e ? *e : dummy;
#define NQE_COMPACT(NAME, E, P, G)                                                                                     \
    launch(ctx, NAME, compact_kernel<E, P, G>, grid, block, 0, sv, src.dtype, src.valid(), gidx, ex, kp, pv, to, km.n,  \
           km.ntiles, ow, ob, ov, ctx->d_flags)
        // plain 8-byte words: kept words staged in LDS and written in whole aligned lines (512 threads, two workgroups per CU); tables of
        // fewer than 64 tiles: from registers, a chunk per wave.  Gathers, Boolean outputs and validity: the general kernel.
        if (!gidx && plainw && km.ntiles >= 64) {
            dim3 ggrid(unsigned(std::min<int64_t>(km.ntiles, int64_t(ctx->num_cus) * 2))), gblock(512);
            uint64_t mul = 1, add = 0;
            if (e && affine_form(*e, &mul, &add))
                launch(ctx, "compact_expr", compact_staged_kernel<2>, ggrid, gblock, 0, (const uint64_t *)sv, ex, mul, add, kp, to, km.n, km.ntiles, ow, ctx->d_flags);
            else if (e) launch(ctx, "compact_expr", compact_staged_kernel<1>, ggrid, gblock, 0, (const uint64_t *)sv, ex, mul, add, kp, to, km.n, km.ntiles, ow, ctx->d_flags);
            else launch(ctx, "compact_column", compact_staged_kernel<0>, ggrid, gblock, 0, (const uint64_t *)sv, ex, mul, add, kp, to, km.n, km.ntiles, ow, ctx->d_flags);
        } else if (!gidx && plainw) {
            dim3 sgrid(stream_grid(ctx, km.ntiles * (TILE_WORDS / COMPACT_CHUNK_WORDS), 4));
            if (e) launch(ctx, "compact_expr", compact_strided_kernel<true>, sgrid, block, 0, (const uint64_t *)sv, ex, kp, to, km.n, km.ntiles, ow, ctx->d_flags);
            else launch(ctx, "compact_column", compact_strided_kernel<false>, sgrid, block, 0, (const uint64_t *)sv, ex, kp, to, km.n, km.ntiles, ow, ctx->d_flags);
        } else if (gidx) {
            if (plainw) NQE_COMPACT("compact_gather", false, true, true);
            else NQE_COMPACT("compact_gather", false, false, true);
        } else if (e) NQE_COMPACT("compact_expr", true, false, false);
        else NQE_COMPACT("compact_column", false, false, false);
#undef NQE_COMPACT
    }
    if (bool_out) pack_bytes_to_bits(ctx, (const uint8_t *)bool_bytes->ptr, m, (uint64_t *)out.values->ptr);
    if (need_valid) pack_bytes_to_bits(ctx, (const uint8_t *)valid_bytes->ptr, m, (uint64_t *)out.validity->ptr);
    return out;
}

DevColumn compact_column(nqe_ctx *ctx, const DevColumn &src, const KeepMask &km) {
    return run_compact(ctx, src, nullptr, src.dtype, km);
}

DevColumn compact_gather_column(nqe_ctx *ctx, const DevColumn &src, const uint32_t *gidx, const KeepMask &km) {
    return run_compact(ctx, src, nullptr, src.dtype, km, gidx);
}

DevColumn compact_simple_expr(nqe_ctx *ctx, const nqe_table *in, const SimpleExpr &e, const KeepMask &km) {
    return run_compact(ctx, in->cols[size_t(e.col)], e.nops ? &e : nullptr, e.out_dtype, km);
}

static KeepMask mask_for_predicate(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int pred_nodes) {
    ExprInfo info = analyze_expr(in, pred, pred_nodes);
    if (info.out_dtype != NQE_BOOLEAN)
        fail(NQE_ERR_NOT_SUPPORTED, "predicate is not a BooleanArray (selection.rs:61 unwrap panics)");
    if (info.simple) {
        // a chain that is not a range test (`(id + 1) % 10 < 5`) over a large plain column: keep_from_simple_kernel interprets it
        // row by row (1.6 TB/s); the expression machine — tile-wise, and specialised at run time from three steps on — writes the
        // Boolean column at 5+ TB/s, and the mask pass over 1 bit per row is nearly free
        const DevColumn &c = in->cols[size_t(info.s.col)];
        FastPred fp{};
        if (info.s.nops >= 2 && in->rows >= (int64_t(1) << 20) && is_word_type(c.dtype) && !c.validity && !make_fast_pred(info.s, &fp)) {
            DevColumn p = evaluate_expr(ctx, in, pred, pred_nodes);
            return build_keep_mask(ctx, p, in->rows);
        }
        return build_keep_mask_simple(ctx, in, info.s);
    }
    ConjPred conj;
    int conj_cols[CONJ_MAX];
    if (match_conj(in, pred, pred_nodes, &conj, conj_cols)) return build_keep_mask_conj(ctx, in, conj, conj_cols);
    DevColumn p = evaluate_expr(ctx, in, pred, pred_nodes);
    return build_keep_mask(ctx, p, in->rows);
}

} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_filter(nqe_ctx *ctx, const nqe_table *in, const nqe_table *pred_table, int32_t pred_column,
                      nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !pred_table || !out || pred_column < 0 || size_t(pred_column) >= pred_table->cols.size())
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    const DevColumn &p = pred_table->cols[size_t(pred_column)];
    if (p.dtype != NQE_BOOLEAN) fail(NQE_ERR_NOT_SUPPORTED, "predicate is not a BooleanArray (selection.rs:61 unwrap panics)");
    flags_reset(ctx);
    KeepMask km = build_keep_mask(ctx, p, in->rows);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = km.total;
    for (auto &c : in->cols) t->cols.push_back(compact_column(ctx, c, km));
    throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_selection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred, int32_t pred_nodes,
                                 nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    // the error flags are reset and read back (a stream synchronisation) only when the predicate can raise one
    const bool fault = analyze_expr(in, pred, pred_nodes).may_fault;
    if (fault) flags_reset(ctx);
    KeepMask km = mask_for_predicate(ctx, in, pred, pred_nodes);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = km.total;
    for (auto &c : in->cols) t->cols.push_back(compact_column(ctx, c, km));
    if (fault) throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

nqe_status nqe_selection_projection_execute(nqe_ctx *ctx, const nqe_table *in, const nqe_expr_node *pred,
                                            int32_t pred_nodes, const nqe_expr_node *nodes,
                                            const int32_t *expr_offsets, int32_t num_exprs, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !in || !out || num_exprs < 0 || (num_exprs > 0 && (!nodes || !expr_offsets)))
        fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    // type errors of the projection surface before any work, as they would at evaluate()
    std::vector<ExprInfo> infos;
    for (int e = 0; e < num_exprs; ++e) infos.push_back(analyze_expr(in, nodes + expr_offsets[e], expr_offsets[e + 1] - expr_offsets[e]));
    // the error flags are reset and read back (a stream synchronisation) only when some expression can raise one
    bool fault = analyze_expr(in, pred, pred_nodes).may_fault;
    for (auto &inf : infos) fault = fault || inf.may_fault;
    if (fault) flags_reset(ctx);
    {
        // a tree predicate over a large input without NULLs: predicate, compaction and projection list in ONE specialised pass
        std::vector<DevColumn> one_pass;
        int64_t kept = 0;
        if (select_project_fused(ctx, in, pred, pred_nodes, nodes, expr_offsets, num_exprs, &one_pass, &kept)) {
            auto t1 = std::make_unique<nqe_table>();
            t1->ctx = ctx;
            t1->rows = kept;
            for (auto &c : one_pass) t1->cols.push_back(std::move(c));
            if (fault) throw_on_flags(ctx);
            *out = t1.release();
            return NQE_OK;
        }
    }
    KeepMask km = mask_for_predicate(ctx, in, pred, pred_nodes);
    auto t = std::make_unique<nqe_table>();
    t->ctx = ctx;
    t->rows = km.total;
    // per expression: fused single-column shape → compact_expr; any tree that fits the stack machine → evaluated and
    // compacted in one pass over the columns it references; otherwise compact the inputs once, then evaluate on the
    // compacted batch.  In every case rows the filter dropped can never raise DivideByZero, as in the reference.
    std::unique_ptr<nqe_table> sel;
    // large inputs: the whole list in one run-time specialised pass over the kept rows, once that kernel has been compiled
    std::vector<DevColumn> fused;
    const bool all_fused = project_specialised(ctx, in, nodes, expr_offsets, num_exprs, km, &fused);
    if (all_fused)
        for (auto &c : fused) t->cols.push_back(std::move(c));
    for (int e = 0; e < num_exprs && !all_fused; ++e) {
        const nqe_expr_node *en = nodes + expr_offsets[e];
        const int nn = expr_offsets[e + 1] - expr_offsets[e];
        DevColumn c;
        if (infos[size_t(e)].simple) c = compact_simple_expr(ctx, in, infos[size_t(e)].s, km);
        else if (!evaluate_expr_compacted(ctx, in, en, nn, km, &c)) {
            if (!sel) {
                sel = std::make_unique<nqe_table>();
                sel->ctx = ctx;
                sel->rows = km.total;
                for (auto &ic : in->cols) sel->cols.push_back(compact_column(ctx, ic, km));
            }
            c = evaluate_expr(ctx, sel.get(), en, nn);
        }
        t->cols.push_back(std::move(c));
    }
    if (fault) throw_on_flags(ctx);
    *out = t.release();
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::keep_from_pred_kernel);
