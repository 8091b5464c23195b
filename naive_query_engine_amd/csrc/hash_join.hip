// hash_join.hip — HashJoin::execute = build() + probe() (reference: src/physical_plan/hash_join.rs:124-254).
//
// Reference: HashMap<XxHash64(key), Vec<build row>> chains, per probe row a chain walk with an
// equality re-check (:86-101), then `take` of every left column by build index and every right
// column by probe index (:237-246).  Output order: probe-row-major, duplicate build keys in
// ascending build row.
//
// Device design
//   build : stable radix sort of (key, build row) → equal keys adjacent, rows ascending;
//           run heads → unique keys with (start, count) into the sorted row list `perm`;
//           open-addressing table of 16-byte slots {key, start<<32|count} (Fibonacci hash, one
//           128-bit load per probe step).  The XxHash64 value itself is unobservable in results
//           (equality is re-checked), so any hash is parity-safe.  When every build key is
//           unique the slot stores the build row directly (no `perm` indirection).
//   probe : pass 1 looks every probe key up once and records its (start,count) word plus
//           per-tile match counts; an exclusive scan gives each 4096-row tile its output base;
//           pass 2 re-reads the recorded words, ranks rows inside the tile with wave scans and
//           writes all output columns (gather of left columns by build row, copy of right
//           columns) in probe order — exactly the order of the reference's outer_pos/inner_pos.
// Key validity is ignored (quirk Q11): raw 8-byte slot values are compared.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "device_utils.hpp"
#include "nqe_internal.hpp"

struct nqe_join_table {
    nqe_ctx *ctx = nullptr;
    std::vector<nqe::DevColumn> left_cols; // shared buffers of the build side ("self.data")
    int64_t left_rows = 0;
    int key_dtype = NQE_INT64;
    nqe::BufRef slots; // ulonglong2[cap]
    nqe::BufRef perm;  // uint32[left_rows], build rows sorted by (key, row)
    // duplicate build keys: plain 8-byte payload columns re-laid out in `perm` order, so that the matches of one probe row — consecutive
    // entries of the sorted row list — are ADJACENT words (one line per probe row instead of two dependent random reads per
    // output row: perm[...] then the column; PMC showed 20 GB of traffic for 4.8 GB algorithmic on the 4-rows-per-key join)
    std::vector<nqe::BufRef> sorted_cols; // per left column, null where none
    uint32_t cap = 0;
    int shift = 0;
    bool direct = false; // all build keys unique: slot.y>>32 is the build row itself
    // a probe that found a key outside a gap-free build key range: later probes of this table take the two-pass form at once
    mutable bool all_match_failed = false;
    // dense build keys (max-min+1 <= 4n): direct-address table instead of hashing.
    //   unique keys:   dense[key-min] = build row + 1
    //   duplicate keys: dense[key-min] = unique-key index + 1 → (ustart[u], ustart[u+1]-ustart[u])
    int left_key = 0;
    nqe::BufRef dense;   // uint32[span]
    nqe::BufRef ustart;  // uint32[U+1]
    uint64_t dense_min = 0, dense_span = 0; // span = number of entries (0: not dense)
    // unique + dense keys + plain 8-byte payload: payload columns re-laid out by (key - min) so that a probe
    // needs ONE random access per gathered value and no build-row lookup at all
    nqe::BufRef presence;                 // uint32 bitmap over [0, span)
    std::vector<nqe::BufRef> dense_cols;  // per left column (null for the key column)
    // Int64/UInt64 payloads whose value range fits 32 bits are stored as uint32 offsets from their minimum (frame of
    // reference): the gather target halves, so more of it stays in the 4 MB per-XCD L2 (the probe is gather-bound)
    std::vector<int> dense_packed;        // dense_cols[ci] holds value - dense_base[ci] as 1: uint32, 2..25: that many bits per entry
    std::vector<uint64_t> dense_base;
    bool dense_payload = false;
    bool dense_full = false; // every key of the dense range occurs
    // unique hashed keys whose build side has exactly one plain payload column: a second table of 16-byte slots {key, payload} — a
    // probe gets key check and payload with ONE random access (beyond the 4 MB per-XCD L2 that access IS the cost: 5.3e10/s
    // whatever the element size — tools/micro_bench.hip).  Empty slots hold `filler`, a value that is not a build key.
    nqe::BufRef slotsp;
    uint64_t filler = 0;
    int pis_col = -1; // the left column carried in the slot
    // … in 8-byte slots when key and payload fit one word together: (key - pp_kmin) << pp_bits | (payload - pp_base), empty = all
    // ones (key bits + payload bits <= 63), 16-slot = 128-byte buckets, pp_nb of them (not a power of two: load ~0.6) — 10^6 keys
    // of a 2^40 domain with a 20-bit payload: 13 MB instead of 32, so more of the probe's line fetches stay in the 4 MB per-XCD L2
    int pp_bits = 0; // 0: the 16-byte form
    uint32_t pp_nb = 0;
    uint64_t pp_kmin = 0, pp_kspan = 0, pp_base = 0;
    // Utf8 join keys: the build strings are encoded to representative-row codes (strings.hip)
    nqe::Utf8Dict dict;
};

namespace nqe {

namespace {

constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
constexpr int JT_ROWS = 4096; // probe tile
constexpr int JT_BLOCK = 256;
constexpr int JT_ITERS = JT_ROWS / JT_BLOCK;
constexpr int MAX_JOIN_COLS = 32;

// Every probe sequence starts at the first slot of the key's 8-slot bucket = one 128-byte line (capacities are multiples of 64):
// inserts fill a bucket from its start, a lookup that reads the whole line has seen every candidate unless the bucket is full.
__device__ __forceinline__ uint32_t home_slot(uint64_t key, int shift) { return uint32_t((key * GOLD) >> shift) & ~7u; }

__global__ void iota_u32_kernel(uint32_t *out, int64_t n) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = uint32_t(i);
}

// flags[j] = 1 iff sorted key j starts a run; flags[n] = 0 (so the exclusive scan leaves the total there)
__global__ void mark_heads_kernel(const uint64_t *skeys, int64_t n, uint32_t *flags) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j <= n; j += stride)
        flags[j] = (j < n && (j == 0 || skeys[j] != skeys[j - 1])) ? 1u : 0u;
}

// ustart[u] = position of the u-th run head in the sorted order; ustart[U] = n
__global__ void fill_ustart_kernel(const uint32_t *flags, const uint64_t *offs, int64_t n, uint32_t U, uint32_t *ustart) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j <= n; j += stride) {
        if (j == n) ustart[U] = uint32_t(n);
        else if (flags[j]) ustart[offs[j]] = uint32_t(j);
    }
}

// inserts every unique key: claims a slot by CAS on the meta word (0 = empty), then stores the key
__global__ void insert_unique_kernel(const uint64_t *skeys, const uint32_t *ustart, const uint32_t *perm, uint32_t U,
                                     ulonglong2 *slots, uint32_t cap, int shift, int direct) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; u < int64_t(U); u += stride) {
        uint32_t j = ustart[u];
        uint64_t key = skeys[j];
        uint64_t count = uint64_t(ustart[u + 1] - j);
        uint64_t start = direct ? uint64_t(perm[j]) : uint64_t(j);
        uint64_t meta = (start << 32) | count;
        uint32_t slot = home_slot(key, shift);
        for (;;) {
            unsigned long long old = atomicCAS((unsigned long long *)&slots[slot].y, 0ull, (unsigned long long)meta);
            if (old == 0ull) {
                slots[slot].x = key;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
}

__device__ __forceinline__ uint64_t probe_one(const ulonglong2 *__restrict__ slots, uint32_t cap, int shift, uint64_t key) {
    uint32_t slot = home_slot(key, shift);
    for (uint32_t p = 0; p < cap; ++p) {
        ulonglong2 s = slots[slot];
        if (s.y == 0ull) return 0ull;
        if (s.x == key) return s.y;
        slot = (slot + 1) & (cap - 1);
    }
    return 0ull;
}

// ---- sort-free build for unique keys (the common case: a dimension table's primary key).  Uniqueness is established by the
// build itself: a second occupant of a dense slot / a second slot with the same key raises *dup and the host falls back to the
// sort-based build below, which handles duplicates (and their ascending-build-row order).
// unsigned min / max of (value ^ flip) over up to MAX_JOIN_COLS columns in one launch: blockIdx.y = column; one atomic pair per
// workgroup (per wave it was 16 K same-address atomics at ~12 ns each = 0.2 ms of a 1e6-row build)
typedef unsigned long long nt_u64x2 __attribute__((ext_vector_type(2))); // (what __builtin_nontemporal_load takes for a 16-byte access)
struct MinMaxCols {
    const uint64_t *src[MAX_JOIN_COLS];
    uint64_t flip[MAX_JOIN_COLS];
};
// `descents` (column 0 = the key only): the number of rows whose key is below its predecessor's — a build side in (nearly) ascending
// key order, the usual shape of a dimension table, writes and gathers coalesced whatever its size
__global__ void __launch_bounds__(256) minmax_cols_kernel(MinMaxCols mc, int64_t n, unsigned long long *mins, unsigned long long *maxs, unsigned long long *descents) {
    __shared__ uint64_t smn[4], smx[4];
    const int c = blockIdx.y;
    const uint64_t *__restrict__ v = mc.src[c];
    const uint64_t flip = mc.flip[c];
    uint64_t mn = ~0ull, mx = 0;
    uint32_t desc = 0;
    // four independent loads in flight per thread (one at a time read 1.6 GB of a 10^8-row build side at 3.2 TB/s: 0.50 ms of the build)
    const int64_t stride = int64_t(gridDim.x) * blockDim.x, last = n - 1;
    const bool want_desc = c == 0 && descents != nullptr;
    // 16-byte-aligned columns (the library's own always are): PAIRS of words in 16-byte non-temporal loads, four in flight — the word before
    // a pair (the descent test across pairs) is the lane below's second word; lane 0 reads it
    const int64_t npairs = (reinterpret_cast<uintptr_t>(v) & 15) == 0 ? n / 2 : 0;
    for (int64_t p0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p0 - threadIdx.x % 64 < npairs; p0 += 4 * stride) { // (whole waves stay in the loop: shuffles)
        nt_u64x2 x[4];
        uint64_t before[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + u * stride, pc = p < npairs ? p : npairs - 1;
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(v) + pc);
            before[u] = (want_desc && lane_id() == 0 && pc > 0) ? v[2 * pc - 1] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + u * stride;
            const uint64_t a = x[u].x ^ flip, b = x[u].y ^ flip;
            uint64_t prev = __shfl_up((unsigned long long)b, 1, 64); // (the lane below holds the pair before this one: consecutive lanes, consecutive pairs)
            if (lane_id() == 0) prev = before[u] ^ flip;
            if (p >= npairs) continue;
            mn = a < mn ? a : mn;
            mn = b < mn ? b : mn;
            mx = a > mx ? a : mx;
            mx = b > mx ? b : mx;
            if (want_desc) desc += (a > b ? 1u : 0u) + ((p > 0 && prev > a) ? 1u : 0u);
        }
    }
    // (the rest: an odd last word, or the whole column when it is not 16-byte aligned)
    for (int64_t i0 = 2 * npairs + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint64_t x[4], p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride, ic = i < last ? i : last;
            x[u] = v[ic] ^ flip;
            p[u] = want_desc ? (v[ic > 0 ? ic - 1 : 0] ^ flip) : 0; // (the neighbour's word is in the line just read)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n) break;
            mn = x[u] < mn ? x[u] : mn;
            mx = x[u] > mx ? x[u] : mx;
            if (want_desc && i > 0 && p[u] > x[u]) ++desc;
        }
    }
    // (one atomic per WORKGROUP: random keys make every wave count descents, and same-address device atomics retire one at a time —
    // a pair per wave was ~0.1 ms of a 10^8-row build's min/max pass)
    __shared__ uint32_t sdesc[4];
    if (want_desc) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) desc += __shfl_down(desc, d, 64);
        if (lane_id() == 0) sdesc[threadIdx.x / 64] = desc;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t a = __shfl_down((unsigned long long)mn, d, 64), b = __shfl_down((unsigned long long)mx, d, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane_id() == 0) smn[threadIdx.x / 64] = mn, smx[threadIdx.x / 64] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = smn[w] < mn ? smn[w] : mn;
            mx = smx[w] > mx ? smx[w] : mx;
        }
        atomicMin(&mins[c], (unsigned long long)mn);
        atomicMax(&maxs[c], (unsigned long long)mx);
        if (want_desc) {
            const uint32_t dsum = sdesc[0] + sdesc[1] + sdesc[2] + sdesc[3];
            if (dsum) atomicAdd(descents, (unsigned long long)dsum);
        }
    }
}
constexpr int UNIQUE_MAX_PROBE = 128;
struct DensePayload {
    int32_t n;
    int32_t pad;
    const uint64_t *src[MAX_JOIN_COLS];
    void *dst[MAX_JOIN_COLS];
    uint64_t base[MAX_JOIN_COLS];
    int32_t packed[MAX_JOIN_COLS]; // 1: dst holds uint32 (value - base); 2..25: that many BITS per entry (value - base < 2^packed)
};
__global__ void __launch_bounds__(256) dense_unique_build_kernel(const uint64_t *keys, int64_t n, uint64_t dmin, uint32_t *dense, uint32_t *presence,
                                                                 DensePayload dp, int *dup) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint64_t d = keys[r] - dmin;
        const uint32_t bit = 1u << (d & 31);
        const uint32_t old = atomicOr(&presence[d >> 5], bit);
        if (old & bit) {
            *dup = 1; // plain store of a constant: every writer agrees
            continue;
        }
        dense[d] = uint32_t(r) + 1u;
        for (int c = 0; c < dp.n; ++c) {
            const uint64_t v = dp.src[c][r];
            if (dp.packed[c] >= 2) {
                // `packed` bits per entry (<= 25), entry d at bit d * packed of a zeroed table: neighbours share words, so the bits are
                // OR-ed in (at most two aligned words per entry)
                const uint64_t bit = d * uint64_t(dp.packed[c]);
                const uint64_t o = uint64_t(uint32_t(v - dp.base[c])) << (bit & 31);
                uint32_t *w = static_cast<uint32_t *>(dp.dst[c]) + (bit >> 5);
                atomicOr(w, uint32_t(o));
                if (o >> 32) atomicOr(w + 1, uint32_t(o >> 32));
            } else if (dp.packed[c]) static_cast<uint32_t *>(dp.dst[c])[d] = uint32_t(v - dp.base[c]);
            else static_cast<uint64_t *>(dp.dst[c])[d] = v;
        }
    }
}
// The same build WITHOUT device-scope atomics (they run at a flat ~2.4x10^10/s on this chip whatever the table size: two to three per
// row made a 10^8-row build 9.8 ms, a 10^7-row one 1.2 ms).  Pass A scatters row numbers with plain stores — of several rows with
// one key any one wins.  Pass B walks the table in KEY order, 64 entries per wave: the presence words are ballots, the number of
// occupied entries (== rows ⇔ the keys are unique) one atomic per workgroup, and the payload columns are GATHERED by the stored
// row (random reads, which the chip serves at 5-10x10^10/s) and written in whole coalesced words — bit-packed entries are
// assembled in LDS, a wave's 64 entries being exactly 2 x bits words.
__global__ void __launch_bounds__(256) dense_scatter_rows_kernel(const uint64_t *keys, int64_t n, uint64_t dmin, uint32_t *dense) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) dense[keys[r] - dmin] = uint32_t(r) + 1u;
}
// `kord` (the partitioned build below): the entries arrive as key-ordered records of `twp` words {row + 1, payload words…} — the row
// table is written from them here, and the payloads are read from the record of entry d, not gathered by row.
// WHOLE (kord with records of two or four words): a record is read in 16-byte loads — a wave's loads then cover its 1 or 2 KB of records
// once; word-by-word non-temporal loads at a 16-byte stride fetched every line once per word
template <bool WHOLE>
__global__ void __launch_bounds__(256) dense_finish_kernel(uint32_t *dense, uint64_t span, uint32_t *presence, DensePayload dp, unsigned long long *occupied,
                                                           const uint64_t *kord, int twp) {
    __shared__ uint32_t pack[4][2 * 25];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const uint64_t ngroups = (span + 63) / 64;
    uint32_t mine = 0;
    for (uint64_t g = uint64_t(blockIdx.x) * 4 + wave; g < ngroups; g += uint64_t(gridDim.x) * 4) {
        const uint64_t d = g * 64 + lane;
        uint32_t e = 0;
        nt_u64x2 r0 = {0ull, 0ull}, r1 = r0;
        constexpr bool whole = WHOLE;
        if (d < span) {
            if (whole) {
                r0 = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(kord + d * uint64_t(twp)));
                if (twp == 4) r1 = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(kord + d * uint64_t(twp) + 2));
                dense[d] = e = uint32_t(r0.x);
            } else if (kord) dense[d] = e = uint32_t(__builtin_nontemporal_load(&kord[d * uint64_t(twp)]));
            else e = dense[d];
        }
        const bool present = e != 0;
        const uint64_t m = __ballot(present);
        mine += __popcll(m);
        // presence: bit d of 32-bit words — this wave's 64 entries are words 2g and 2g + 1 (the bitmap is allocated in whole pairs)
        if (presence && lane < 2 && 2 * g + lane < (span + 31) / 32) presence[2 * g + lane] = uint32_t(m >> (32 * lane));
        for (int c = 0; c < dp.n; ++c) {
            const uint64_t v = !present ? dp.base[c]
                               : whole  ? (c == 0 ? r0.y : (c == 1 ? r1.x : r1.y)) // (twp 2: one payload word; twp 4: up to three)
                               : kord   ? __builtin_nontemporal_load(&kord[d * uint64_t(twp) + 1 + c])
                                        : dp.src[c][e - 1];
            const int nb = dp.packed[c];
            if (nb >= 2) {
                if (lane < 2 * nb) pack[wave][lane] = 0;
                __builtin_amdgcn_wave_barrier();
                const uint32_t bit = uint32_t(lane) * uint32_t(nb);
                const uint64_t o = uint64_t(uint32_t(v - dp.base[c])) << (bit & 31);
                atomicOr(&pack[wave][bit >> 5], uint32_t(o));
                if (o >> 32) atomicOr(&pack[wave][(bit >> 5) + 1], uint32_t(o >> 32));
                __builtin_amdgcn_wave_barrier();
                // entry d at bit d * nb: the wave's first entry starts word 2 * nb * g
                if (lane < 2 * nb) static_cast<uint32_t *>(dp.dst[c])[g * uint64_t(2 * nb) + lane] = pack[wave][lane];
                __builtin_amdgcn_wave_barrier();
            } else if (d < span) {
                if (nb) static_cast<uint32_t *>(dp.dst[c])[d] = uint32_t(v - dp.base[c]);
                else static_cast<uint64_t *>(dp.dst[c])[d] = v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64); // (every lane of a wave holds the same count: lane 0's sum counts it 64 times)
    if (lane == 0 && mine) atomicAdd(occupied, (unsigned long long)(mine / 64));
}
// ---- Partitioned dense build (builds of >= 2^25 rows).  A table of gigabytes takes neither form above: random 4-byte stores and
// gathers over that range run at a fraction of the chip's line rate (22 ms per 10^8 rows), the atomics at their flat
// 2.4x10^10/s (10-11 ms).  Here the rows are first PARTITIONED BY KEY RANGE — tuples {key - min | row, payload words} in one
// stream; count, scan, then per 8192-row tile a counting sort in LDS so that a partition's tuples leave as one run — into
// slices of the table that fit one XCD's L2 (<= 3 MB of table per partition).  The second pass then scatters partition by
// partition: the workgroups of one XCD (blockIdx % 8) walk the same partitions together, their random stores land in an
// L2-resident slice and leave it as whole lines.  Payload words travel with the tuple (no gather by build row afterwards);
// dense_finish_kernel packs them from the key-ordered copies and counts the occupied entries (== rows <=> unique keys).
constexpr int PB_BLOCK = 1024;
constexpr int PB_MAX_PARTS = 1024;
constexpr int PB_XCDS = 8;
struct PartBuild {
    const uint64_t *keys;
    int64_t n;
    uint64_t dmin;
    int32_t shift; // partition = (key - dmin) >> shift
    int32_t parts;
    int64_t chunk; // rows per workgroup of the count / scatter passes (a multiple of the tile)
    int32_t W;     // workgroups of the count / scatter passes
    int32_t nc;    // payload words per tuple
    const uint64_t *src[MAX_JOIN_COLS];
};
__global__ void __launch_bounds__(PB_BLOCK) part_build_count_kernel(PartBuild pb, uint32_t *counts) {
    __shared__ uint32_t hist[PB_MAX_PARTS];
    for (int p = threadIdx.x; p < PB_MAX_PARTS; p += blockDim.x) hist[p] = 0;
    __syncthreads();
    const int64_t lo = int64_t(blockIdx.x) * pb.chunk;
    const int64_t hi = lo + pb.chunk < pb.n ? lo + pb.chunk : pb.n;
    for (int64_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4 * int64_t(blockDim.x)) { // (four loads in flight per thread)
        uint64_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + u * int64_t(blockDim.x);
            k[u] = __builtin_nontemporal_load(&pb.keys[r < hi ? r : hi - 1]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * int64_t(blockDim.x) < hi) atomicAdd(&hist[uint32_t((k[u] - pb.dmin) >> pb.shift)], 1u);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < pb.parts; p += blockDim.x) counts[size_t(p) * size_t(pb.W) + blockIdx.x] = hist[p];
}
// offsets[p * W + w] (exclusive scan of the counts): where workgroup w's tuples of partition p start in the tuple stream
template <int RPT>
__global__ void __launch_bounds__(PB_BLOCK) part_build_scatter_kernel(PartBuild pb, const uint64_t *offsets, uint64_t *tuples) {
    constexpr int ROWS = PB_BLOCK * RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int TW = 1 + pb.nc;
    uint64_t *stup = reinterpret_cast<uint64_t *>(smem);                          // [ROWS][TW]
    uint32_t *gcur = reinterpret_cast<uint32_t *>(stup + size_t(ROWS) * size_t(TW)); // [PB_MAX_PARTS] next tuple of (partition, this workgroup)
    uint32_t *tcnt = gcur + PB_MAX_PARTS;                                         // tuples of this tile per partition
    uint32_t *tstart = tcnt + PB_MAX_PARTS;                                       // tile-local exclusive scan
    __shared__ uint32_t wave_tot[PB_BLOCK / 64];
    const int parts = pb.parts;
    for (int p = threadIdx.x; p < PB_MAX_PARTS; p += blockDim.x) {
        gcur[p] = p < parts ? uint32_t(offsets[size_t(p) * size_t(pb.W) + blockIdx.x]) : 0u; // (rows < 2^32)
        tcnt[p] = 0;
    }
    __syncthreads();
    const int64_t lo = int64_t(blockIdx.x) * pb.chunk;
    const int64_t hi = lo + pb.chunk < pb.n ? lo + pb.chunk : pb.n;
    for (int64_t base = lo; base < hi; base += ROWS) {
        uint32_t d[RPT], rank[RPT];
        bool ok[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int64_t row = base + int64_t(u) * PB_BLOCK + threadIdx.x;
            ok[u] = row < hi;
            d[u] = ok[u] ? uint32_t(__builtin_nontemporal_load(&pb.keys[row]) - pb.dmin) : 0u;
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) rank[u] = ok[u] ? atomicAdd(&tcnt[d[u] >> pb.shift], 1u) : 0u;
        __syncthreads();
        const uint32_t c = tcnt[threadIdx.x]; // PB_MAX_PARTS == PB_BLOCK: one counter per thread
        uint32_t wt;
        const uint32_t ex = wave_exclusive_scan(c, wt);
        if (lane_id() == 63) wave_tot[threadIdx.x / 64] = wt;
        __syncthreads();
        uint32_t pre = 0, tile_total = 0;
        for (int w = 0; w < PB_BLOCK / 64; ++w) {
            if (w < int(threadIdx.x) / 64) pre += wave_tot[w];
            tile_total += wave_tot[w];
        }
        tstart[threadIdx.x] = pre + ex;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (!ok[u]) continue;
            const int64_t row = base + int64_t(u) * PB_BLOCK + threadIdx.x;
            const uint32_t i = tstart[d[u] >> pb.shift] + rank[u];
            stup[size_t(i) * TW] = (uint64_t(d[u]) << 32) | uint64_t(uint32_t(row));
            for (int cc = 0; cc < pb.nc; ++cc) stup[size_t(i) * TW + 1 + cc] = __builtin_nontemporal_load(&pb.src[cc][row]);
        }
        __syncthreads();
        if (TW == 2) {
            for (uint32_t i = threadIdx.x; i < tile_total; i += PB_BLOCK) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(&stup[size_t(i) * 2]);
                const uint32_t p = uint32_t(t.x >> 32) >> pb.shift;
                *reinterpret_cast<ulonglong2 *>(&tuples[(size_t(gcur[p]) + (i - tstart[p])) * 2]) = t;
            }
        } else {
            // word e of the tile's sorted tuples: consecutive lanes write consecutive words, across tuple boundaries
            const uint32_t words = tile_total * uint32_t(TW);
            for (uint32_t e = threadIdx.x; e < words; e += PB_BLOCK) {
                const uint32_t i = e / uint32_t(TW), k = e - i * uint32_t(TW);
                const uint32_t p = uint32_t(stup[size_t(i) * TW] >> 32) >> pb.shift;
                tuples[(size_t(gcur[p]) + (i - tstart[p])) * size_t(TW) + k] = stup[e];
            }
        }
        __syncthreads();
        gcur[threadIdx.x] += tcnt[threadIdx.x];
        tcnt[threadIdx.x] = 0;
        __syncthreads();
    }
}
// The same scatter for key-only builds and one payload word (NC = 0 / 1: the shapes of the two-level form), with the tile's words in
// REGISTERS from the start — the payload word was requested only after two barriers, its whole latency in front of the staging — and
// the NEXT tile's words requested as soon as this tile's are staged: they arrive during the copy-out.  (One 1024-thread workgroup per
// CU holds the LDS: nothing else overlaps its phases.)
template <int NC>
__global__ void __launch_bounds__(PB_BLOCK) part_build_scatter1_kernel(PartBuild pb, const uint64_t *offsets, uint64_t *tuples) {
    constexpr int RPT = 8, ROWS = PB_BLOCK * RPT, TW = 1 + NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *stup = reinterpret_cast<uint64_t *>(smem);                          // [ROWS][TW]
    uint32_t *gcur = reinterpret_cast<uint32_t *>(stup + size_t(ROWS) * size_t(TW)); // [PB_MAX_PARTS] next tuple of (partition, this workgroup)
    uint32_t *tcnt = gcur + PB_MAX_PARTS;
    uint32_t *tstart = tcnt + PB_MAX_PARTS;
    __shared__ uint32_t wave_tot[PB_BLOCK / 64];
    const int parts = pb.parts;
    for (int p = threadIdx.x; p < PB_MAX_PARTS; p += blockDim.x) {
        gcur[p] = p < parts ? uint32_t(offsets[size_t(p) * size_t(pb.W) + blockIdx.x]) : 0u; // (rows < 2^32)
        tcnt[p] = 0;
    }
    __syncthreads();
    const int64_t lo = int64_t(blockIdx.x) * pb.chunk;
    const int64_t hi = lo + pb.chunk < pb.n ? lo + pb.chunk : pb.n;
    const uint64_t *__restrict__ keys = pb.keys;
    const uint64_t *__restrict__ pay = NC ? pb.src[0] : pb.keys;
    uint64_t kw[RPT], pw[NC ? RPT : 1];
    auto load = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            int64_t row = base + int64_t(u) * PB_BLOCK + threadIdx.x;
            row = row < hi ? row : hi - 1; // clamp: unconditional, in-bounds (lo < hi)
            kw[u] = __builtin_nontemporal_load(&keys[row]);
            if (NC) pw[NC ? u : 0] = __builtin_nontemporal_load(&pay[row]);
        }
    };
    if (lo < hi) load(lo);
    for (int64_t base = lo; base < hi; base += ROWS) {
        uint32_t d[RPT], rank[RPT];
        bool ok[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            ok[u] = base + int64_t(u) * PB_BLOCK + threadIdx.x < hi;
            d[u] = uint32_t(kw[u] - pb.dmin);
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) rank[u] = ok[u] ? atomicAdd(&tcnt[d[u] >> pb.shift], 1u) : 0u;
        __syncthreads();
        const uint32_t c = tcnt[threadIdx.x]; // PB_MAX_PARTS == PB_BLOCK: one counter per thread
        uint32_t wt;
        const uint32_t ex = wave_exclusive_scan(c, wt);
        if (lane_id() == 63) wave_tot[threadIdx.x / 64] = wt;
        __syncthreads();
        uint32_t pre = 0, tile_total = 0;
        for (int w = 0; w < PB_BLOCK / 64; ++w) {
            if (w < int(threadIdx.x) / 64) pre += wave_tot[w];
            tile_total += wave_tot[w];
        }
        tstart[threadIdx.x] = pre + ex;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (!ok[u]) continue;
            const uint32_t row = uint32_t(base + int64_t(u) * PB_BLOCK + threadIdx.x);
            const uint32_t i = tstart[d[u] >> pb.shift] + rank[u];
            const uint64_t x = (uint64_t(d[u]) << 32) | uint64_t(row);
            if (NC) *reinterpret_cast<ulonglong2 *>(&stup[size_t(i) * 2]) = make_ulonglong2(x, pw[NC ? u : 0]);
            else stup[i] = x;
        }
        if (base + ROWS < hi) load(base + ROWS); // (workgroup-uniform) the next tile's words fly during the copy-out
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tile_total; i += PB_BLOCK) {
            if (NC) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(&stup[size_t(i) * 2]);
                const uint32_t p = uint32_t(t.x >> 32) >> pb.shift;
                *reinterpret_cast<ulonglong2 *>(&tuples[(size_t(gcur[p]) + (i - tstart[p])) * 2]) = t;
            } else {
                const uint64_t t = stup[i];
                const uint32_t p = uint32_t(t >> 32) >> pb.shift;
                tuples[size_t(gcur[p]) + (i - tstart[p])] = t;
            }
        }
        __syncthreads();
        gcur[threadIdx.x] += tcnt[threadIdx.x];
        tcnt[threadIdx.x] = 0;
        __syncthreads();
    }
}
// pass 2: the workgroups of XCD x (HW_REG_XCC_ID — a performance matter only) take the partitions p = x, x + 8, … one after the other,
// sharing each 2048 tuples at a time through the partition's cursor; afterwards every workgroup sweeps all cursors once and takes
// what is left (nothing, when the hardware numbers its XCDs 0 … 7), so every tuple is placed whatever the mapping.
constexpr int PB_CHUNK = 2048;
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
// Key-only build sides store the row number into `dense`; with payload words the whole entry goes into a key-ordered record of
// twp = 2 * ceil((1 + nc) / 2) words in 16-byte stores (random stores cost per store, not per byte: 6-8x10^10/s whatever the slice).
__global__ void __launch_bounds__(256) part_build_place_kernel(PartBuild pb, const uint64_t *offsets, const uint64_t *tuples, uint32_t *dense, uint64_t *kord,
                                                               int twp, uint32_t *cursor, int by_block, uint32_t chunk) {
    __shared__ uint32_t got;
    __shared__ int left;
    const int x = by_block ? int(blockIdx.x % PB_XCDS) : int(xcc_id() % PB_XCDS);
    const int TW = 1 + pb.nc;
    for (int sweep = 0; sweep < 2; ++sweep) {
        if (sweep) { // anything left?  (all cursors looked at together; normally nothing is)
            if (threadIdx.x == 0) left = 0;
            __syncthreads();
            for (int p = threadIdx.x; p < pb.parts; p += 256)
                if (offsets[size_t(p) * size_t(pb.W)] + __hip_atomic_load(&cursor[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < offsets[size_t(p + 1) * size_t(pb.W)]) left = 1;
            __syncthreads();
            if (!left) break; // (workgroup-uniform)
        }
        for (int p = sweep ? 0 : x; p < pb.parts; p += sweep ? 1 : PB_XCDS) {
            const uint64_t s = offsets[size_t(p) * size_t(pb.W)], e = offsets[size_t(p + 1) * size_t(pb.W)]; // (offsets[parts * W] = rows)
            for (;;) {
                if (threadIdx.x == 0) got = atomicAdd(&cursor[p], chunk);
                __syncthreads();
                const uint64_t c0 = s + got;
                __syncthreads();
                if (c0 >= e) break;
                const uint64_t c1 = c0 + chunk < e ? c0 + chunk : e;
                for (uint64_t i = c0 + threadIdx.x; i < c1; i += 256) {
                    if (kord && TW == 2) { // {key - min | row, payload}: one 16-byte load, one 16-byte store
                        const nt_u64x2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(tuples + i * 2));
                        *reinterpret_cast<ulonglong2 *>(kord + uint64_t(uint32_t(t.x >> 32)) * 2) = make_ulonglong2(uint64_t(uint32_t(t.x) + 1u), t.y);
                        continue;
                    }
                    const uint64_t w0 = __builtin_nontemporal_load(&tuples[i * TW]);
                    const uint32_t d = uint32_t(w0 >> 32);
                    if (!kord) {
                        dense[d] = uint32_t(w0) + 1u;
                        continue;
                    }
                    uint64_t *rec = kord + uint64_t(d) * uint64_t(twp);
                    uint64_t a = uint64_t(uint32_t(w0) + 1u);
                    for (int k = 0; k < twp; k += 2) {
                        const uint64_t b = k + 1 < TW ? __builtin_nontemporal_load(&tuples[i * TW + k + 1]) : 0ull;
                        *reinterpret_cast<ulonglong2 *>(rec + k) = make_ulonglong2(a, b);
                        a = k + 2 < TW ? __builtin_nontemporal_load(&tuples[i * TW + k + 2]) : 0ull;
                    }
                }
            }
        }
    }
}
// ---- Two-level form of the partitioned build (round 6): what the place pass pays for is one scattered 16-byte store per row —
// 1.9 ms per 10^8 rows whatever its slice size, workgroup count or cursor chunk (profiles/r05/sweep_build_place.txt), plus a zeroed
// 16-byte record per key written and read back (memset 0.3 + finish 0.6 ms).  Here every store is coalesced: the count pass takes a
// FINE histogram (up to 64 bins per partition, each PB_FILL_KEYS keys wide: fine_count), so that after the usual scatter into
// partitions a second one — ONE workgroup per partition, a counting sort of 4096-tuple tiles in LDS, runs of a hundred tuples
// (part_build_split_kernel) — leaves the tuples grouped by fine bin; a fine bin's keys then fit a workgroup's LDS, where its entries
// are laid out in key order and leave as whole lines of the FINAL tables (row table, presence words, the packed payload column:
// part_build_fill_kernel) — no key-ordered records, no finish pass.  Key-only builds and builds with one payload word.
constexpr int PB_FILL_LOG2 = 13, PB_FILL_KEYS = 1 << PB_FILL_LOG2; // keys of a fine bin: 4 + 8 bytes of LDS each
constexpr int PB_FINE_LOG2_MAX = 6;                                 // at most 64 fine bins per partition: partition = key >> (PB_FILL_LOG2 + fine_log2), fine_log2 = pb.shift - PB_FILL_LOG2 (the host picks it: see build_unique_fast)
constexpr int PB_MAX_FINE = 32768;                                  // bins in all: 128 KB of LDS in the count pass (2.7 x 10^8 keys)
// count pass: this workgroup's rows per FINE bin in LDS; the partition counts of the scatter's offsets are sums of 2^fine_log2 of them, and the
// workgroup's fine histogram goes to finehist[w][bin] (added up by part_build_fine_offsets_kernel)
__global__ void __launch_bounds__(PB_BLOCK) part_build_count_fine_kernel(PartBuild pb, uint32_t *counts, uint32_t *finehist, int bins) {
    extern __shared__ uint32_t fhist[];
    for (int b = threadIdx.x; b < bins; b += blockDim.x) fhist[b] = 0;
    __syncthreads();
    const int64_t lo = int64_t(blockIdx.x) * pb.chunk;
    const int64_t hi = lo + pb.chunk < pb.n ? lo + pb.chunk : pb.n;
    for (int64_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4 * int64_t(blockDim.x)) { // (four loads in flight per thread)
        uint64_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + u * int64_t(blockDim.x);
            k[u] = __builtin_nontemporal_load(&pb.keys[r < hi ? r : hi - 1]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * int64_t(blockDim.x) < hi) atomicAdd(&fhist[uint32_t((k[u] - pb.dmin) >> PB_FILL_LOG2)], 1u);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < pb.parts; p += blockDim.x) {
        uint32_t c = 0;
        const int fl = pb.shift - PB_FILL_LOG2;
        for (int f = 0; f < (1 << fl); ++f) c += (p << fl) + f < bins ? fhist[(p << fl) + f] : 0u;
        counts[size_t(p) * size_t(pb.W) + blockIdx.x] = c;
    }
    for (int b = threadIdx.x; b < bins; b += blockDim.x) finehist[size_t(blockIdx.x) * size_t(bins) + b] = fhist[b];
}
// fine_start[b]: where fine bin b's tuples start in the twice-partitioned stream = its partition's start (offsets[p * W], the scatter's
// scan) + the bins of the partition before it.  One workgroup per partition (thread = bin x an eighth of the count workgroups);
// fine_start[bins] = rows.
__global__ void __launch_bounds__(256) part_build_fine_offsets_kernel(const uint32_t *finehist, int W, int bins, int parts, int fine_log2, const uint64_t *offsets, uint64_t *fine_start) {
    const int F = 1 << fine_log2;
    __shared__ uint32_t part[256]; // [256 / F][F]
    const int p = blockIdx.x, f = threadIdx.x % F, q = threadIdx.x / F, b = (p << fine_log2) + f;
    uint32_t c = 0;
    if (b < bins)
        for (int w = q; w < W; w += 256 / F) c += finehist[size_t(w) * size_t(bins) + b];
    part[q * F + f] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t at = offsets[size_t(p) * size_t(W)];
        for (int ff = 0; ff < F && (p << fine_log2) + ff < bins; ++ff) {
            fine_start[(p << fine_log2) + ff] = at;
            for (int qq = 0; qq < 256 / F; ++qq) at += part[qq * F + ff];
        }
        if (p == parts - 1) fine_start[bins] = at;
    }
}
// second scatter: workgroup p sorts partition p's tuples (offsets[p * W] .. offsets[(p + 1) * W) of `tuples`) by fine bin into `out`
// (same positions overall: the partition's range, its bins in order).  Per 4096-tuple tile: rank per bin (LDS atomic), the bins'
// starts inside the tile, tuples staged in LDS by bin, copy-out in runs — the only writer of its range, so the cursors are its own.
constexpr int PB_SPLIT_TILE = 4096;
template <int NC>
__global__ void __launch_bounds__(PB_BLOCK) part_build_split_kernel(PartBuild pb, const uint64_t *offsets, const uint64_t *fine_start, int bins, const uint64_t *tuples, uint64_t *out) {
    constexpr int FMAX = 1 << PB_FINE_LOG2_MAX, RPT = PB_SPLIT_TILE / PB_BLOCK;
    const int fine_log2 = pb.shift - PB_FILL_LOG2, F = 1 << fine_log2;
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    uint64_t *stage = reinterpret_cast<uint64_t *>(pb_smem); // [PB_SPLIT_TILE][1 + NC]
    __shared__ uint32_t tcnt[FMAX], tstart[FMAX + 1];
    __shared__ uint64_t cur[FMAX];
    for (int p = blockIdx.x; p < pb.parts; p += gridDim.x) {
        const uint64_t s = offsets[size_t(p) * size_t(pb.W)], e = offsets[size_t(p + 1) * size_t(pb.W)];
        __syncthreads(); // (the previous partition's cursors are done with)
        if (int(threadIdx.x) < F) {
            const int b = (p << fine_log2) + int(threadIdx.x);
            cur[threadIdx.x] = b < bins ? fine_start[b] : e;
            tcnt[threadIdx.x] = 0;
        }
        __syncthreads();
        uint64_t x[RPT], y[NC ? RPT : 1];
        auto load = [&](uint64_t base) { // (the partition is not empty: s < e)
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const uint64_t i = base + uint64_t(u) * PB_BLOCK + threadIdx.x, ic = i < e ? i : e - 1;
                if (NC) {
                    const nt_u64x2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(tuples + ic * 2));
                    x[u] = t.x;
                    y[NC ? u : 0] = t.y;
                } else
                    x[u] = __builtin_nontemporal_load(&tuples[ic]);
            }
        };
        if (s < e) load(s);
        for (uint64_t base = s; base < e; base += PB_SPLIT_TILE) {
            uint32_t f[RPT], rank[RPT];
            bool ok[RPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                ok[u] = base + uint64_t(u) * PB_BLOCK + threadIdx.x < e;
                f[u] = (uint32_t(x[u] >> 32) >> PB_FILL_LOG2) & uint32_t(F - 1);
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) rank[u] = ok[u] ? atomicAdd(&tcnt[f[u]], 1u) : 0u;
            __syncthreads();
            if (threadIdx.x < 64) { // F <= 64 counters: one wave scans them
                const uint32_t c = int(threadIdx.x) < F ? tcnt[threadIdx.x] : 0u;
                uint32_t tot;
                const uint32_t ex = wave_exclusive_scan(c, tot);
                if (int(threadIdx.x) < F) tstart[threadIdx.x] = ex;
                if (threadIdx.x == 0) tstart[F] = tot;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (!ok[u]) continue;
                const uint32_t i = tstart[f[u]] + rank[u];
                if (NC) *reinterpret_cast<ulonglong2 *>(&stage[size_t(i) * 2]) = make_ulonglong2(x[u], y[NC ? u : 0]);
                else stage[i] = x[u];
            }
            if (base + PB_SPLIT_TILE < e) load(base + PB_SPLIT_TILE); // (workgroup-uniform) the next tile flies during the copy-out
            __syncthreads();
            const uint32_t total = tstart[F];
            for (uint32_t i = threadIdx.x; i < total; i += PB_BLOCK) {
                if (NC) {
                    const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(&stage[size_t(i) * 2]);
                    const uint32_t fb = (uint32_t(t.x >> 32) >> PB_FILL_LOG2) & uint32_t(F - 1);
                    *reinterpret_cast<ulonglong2 *>(&out[(cur[fb] + (i - tstart[fb])) * 2]) = t;
                } else {
                    const uint64_t t = stage[i];
                    const uint32_t fb = (uint32_t(t >> 32) >> PB_FILL_LOG2) & uint32_t(F - 1);
                    out[cur[fb] + (i - tstart[fb])] = t;
                }
            }
            __syncthreads();
            if (int(threadIdx.x) < F) {
                cur[threadIdx.x] += tcnt[threadIdx.x];
                tcnt[threadIdx.x] = 0;
            }
            __syncthreads();
        }
    }
}
// one key-ordered group of 64 table entries (a wave): the row table, the presence words, the payload column of entry d in the form the
// join table keeps it (DensePayload::packed) — dense_finish_kernel's stores.  e: row + 1 (0: no such key), v: the payload word
__device__ __forceinline__ void dense_store_group(uint32_t *dense, uint32_t *presence, const DensePayload &dp, uint32_t (*pack)[2 * 25], int wave, int lane, uint64_t g, uint64_t span,
                                                  uint32_t e, uint64_t v) {
    const uint64_t d = g * 64 + uint64_t(lane);
    if (d < span) dense[d] = e;
    const bool present = e != 0;
    const uint64_t m = __ballot(present);
    if (presence && lane < 2 && 2 * g + lane < (span + 31) / 32) presence[2 * g + lane] = uint32_t(m >> (32 * lane));
    if (dp.n == 0) return;
    if (!present) v = dp.base[0];
    const int nb = dp.packed[0];
    if (nb >= 2) {
        if (lane < 2 * nb) pack[wave][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t bit = uint32_t(lane) * uint32_t(nb);
        const uint64_t o = uint64_t(uint32_t(v - dp.base[0])) << (bit & 31);
        atomicOr(&pack[wave][bit >> 5], uint32_t(o));
        if (o >> 32) atomicOr(&pack[wave][(bit >> 5) + 1], uint32_t(o >> 32));
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * nb) static_cast<uint32_t *>(dp.dst[0])[g * uint64_t(2 * nb) + lane] = pack[wave][lane];
        __builtin_amdgcn_wave_barrier();
    } else if (d < span) {
        if (nb) static_cast<uint32_t *>(dp.dst[0])[d] = uint32_t(v - dp.base[0]);
        else static_cast<uint64_t *>(dp.dst[0])[d] = v;
    }
}
// fill: a workgroup takes fine bins (fine_start[b] .. fine_start[b + 1) of the twice-partitioned tuples = the keys [b, b + 1) <<
// PB_FILL_LOG2), lays their entries out in key order in LDS (a key met twice: one of its rows stays — the occupied count then
// falls short of the rows, and the caller takes the sort-based build) and writes the final tables in whole groups of 64 entries.
template <int NC>
__global__ void __launch_bounds__(PB_BLOCK) part_build_fill_kernel(const uint64_t *fine_start, int bins, const uint64_t *tuples, uint64_t span, uint32_t *dense, uint32_t *presence, DensePayload dp,
                                                                   unsigned long long *occupied) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    uint64_t *lv = reinterpret_cast<uint64_t *>(pb_smem);                              // [NC ? PB_FILL_KEYS : 0] payload words
    uint32_t *le = reinterpret_cast<uint32_t *>(lv + (NC ? PB_FILL_KEYS : 0));         // [PB_FILL_KEYS] row + 1
    __shared__ uint32_t pack[PB_BLOCK / 64][2 * 25];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t mine = 0;
    for (int b = blockIdx.x; b < bins; b += gridDim.x) {
        __syncthreads(); // (the previous bin's entries have left)
        for (int i = threadIdx.x; i < PB_FILL_KEYS; i += PB_BLOCK) le[i] = 0;
        __syncthreads();
        const uint64_t s = fine_start[b], e = fine_start[b + 1];
        const uint32_t d0 = uint32_t(b) << PB_FILL_LOG2;
        for (uint64_t i0 = s + threadIdx.x; i0 < e; i0 += 4 * uint64_t(PB_BLOCK)) { // (four loads in flight per thread)
            uint64_t x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + uint64_t(u) * PB_BLOCK, ic = i < e ? i : e - 1;
                if (NC) {
                    const nt_u64x2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u64x2 *>(tuples + ic * 2));
                    x[u] = t.x;
                    y[u] = t.y;
                } else {
                    x[u] = __builtin_nontemporal_load(&tuples[ic]);
                    y[u] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + uint64_t(u) * PB_BLOCK >= e) continue;
                const uint32_t sl = (uint32_t(x[u] >> 32) - d0) & uint32_t(PB_FILL_KEYS - 1);
                le[sl] = uint32_t(x[u]) + 1u;
                if (NC) lv[NC ? sl : 0] = y[u];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < PB_FILL_KEYS; i += PB_BLOCK) { // (whole waves: i - lane is a multiple of 64)
            const uint64_t g = (uint64_t(d0) + uint64_t(i)) >> 6;
            if (g * 64 >= span) break; // (wave-uniform: the table ends inside the last bin)
            const uint32_t ent = le[i];
            mine += ent != 0 ? 1u : 0u;
            dense_store_group(dense, presence, dp, pack, wave, lane, g, span, ent, NC ? lv[NC ? i : 0] : 0ull);
        }
    }
    // occupied entries: one atomic per workgroup
    __shared__ uint32_t wsum[PB_BLOCK / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0) wsum[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < PB_BLOCK / 64; ++w) t += wsum[w];
        if (t) atomicAdd(occupied, t);
    }
}
// claims the first free slot of the probe sequence for every row (no key comparison: equal keys simply occupy several slots),
// then writes the key (and the 32-byte companion slot at the same index)
__global__ void __launch_bounds__(256) hashed_insert_rows_kernel(const uint64_t *keys, int64_t n, ulonglong2 *slots, uint32_t cap, int shift, int *dup) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint64_t key = keys[r];
        const unsigned long long meta = ((unsigned long long)r << 32) | 1ull;
        uint32_t slot = home_slot(key, shift);
        // bounded walk: with unique keys at load <= 1/2 a sequence this long does not occur; many equal keys (which the sort-based
        // build handles) would otherwise turn the insert into O(n^2)
        bool placed = false;
        for (int p = 0; p < UNIQUE_MAX_PROBE; ++p) {
            if (atomicCAS((unsigned long long *)&slots[slot].y, 0ull, meta) == 0ull) {
                placed = true;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
        if (!placed) {
            *dup = 1;
            continue;
        }
        slots[slot].x = key;
    }
}
// the {key, payload} table: empty slots hold `filler` (not a build key), so the key word itself is claimed by CAS — and an equal
// key already in place IS a duplicate
__global__ void __launch_bounds__(256) fill_pairs_kernel(ulonglong2 *t, uint32_t cap, uint64_t filler) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) t[i] = make_ulonglong2(filler, 0ull);
}
__global__ void __launch_bounds__(256) hashed_insert_pairs_kernel(const uint64_t *keys, const uint64_t *payload, int64_t n, ulonglong2 *t, uint32_t cap, int shift,
                                                                  uint64_t filler, int *dup) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint64_t key = keys[r];
        uint32_t slot = home_slot(key, shift);
        bool placed = false;
        for (int p = 0; p < UNIQUE_MAX_PROBE; ++p) {
            const unsigned long long old = atomicCAS((unsigned long long *)&t[slot].x, (unsigned long long)filler, (unsigned long long)key);
            if (old == filler) {
                t[slot].y = payload[r];
                placed = true;
                break;
            }
            if (old == key) break; // the same key twice
            slot = (slot + 1) & (cap - 1);
        }
        if (!placed) *dup = 1;
    }
}
// the packed form of the {key, payload} table (nqe_join_table::pp_bits)
struct PackedPairs {
    uint64_t kmin, kspan, pbase;
    uint32_t nb;   // buckets of 16 slots
    int32_t pbits; // payload bits (the low ones)
};
constexpr int PACKED_BUCKET = 16;
__device__ __forceinline__ uint32_t packed_home(uint64_t key, uint32_t nb) { return uint32_t((uint64_t(uint32_t((key * GOLD) >> 32)) * nb) >> 32); }
__global__ void __launch_bounds__(256) packed_insert_kernel(const uint64_t *keys, const uint64_t *payload, int64_t n, unsigned long long *tab, PackedPairs pp, int *dup) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const uint32_t total = pp.nb * PACKED_BUCKET;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint64_t key = keys[r], kd = key - pp.kmin;
        const unsigned long long w = (kd << pp.pbits) | (payload[r] - pp.pbase);
        uint32_t slot = packed_home(key, pp.nb) * PACKED_BUCKET;
        bool placed = false;
        for (int p = 0; p < UNIQUE_MAX_PROBE; ++p) {
            const unsigned long long old = atomicCAS(&tab[slot], ~0ull, w);
            if (old == ~0ull) {
                placed = true;
                break;
            }
            if ((old >> pp.pbits) == kd) break; // the same key twice
            slot = slot + 1 == total ? 0 : slot + 1;
        }
        if (!placed) *dup = 1;
    }
}
// after the insert kernel has completed: does any row's probe sequence hold its key twice?
__global__ void __launch_bounds__(256) hashed_check_unique_kernel(const uint64_t *keys, int64_t n, const ulonglong2 *slots, uint32_t cap, int shift, int *dup) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint64_t key = keys[r];
        uint32_t slot = home_slot(key, shift);
        for (int p = 0; p < 2 * UNIQUE_MAX_PROBE; ++p) {
            const ulonglong2 s = slots[slot];
            if (s.y == 0ull) break;
            if (s.x == key && uint32_t(s.y >> 32) != uint32_t(r)) {
                *dup = 1;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
}

struct Lookup {
    const ulonglong2 *slots; // hash table (16-byte slots)
    uint32_t cap;
    int32_t shift;
    const uint32_t *dense;   // direct-address table or null
    const uint32_t *ustart;
    uint64_t dense_min, dense_span;
    int32_t direct;
    int32_t pad;
};

// (start<<32 | count) of `key`, 0 when absent. direct ⇒ start is the build row itself.
__device__ __forceinline__ uint64_t lookup_meta(const Lookup &L, uint64_t key) {
    if (L.dense) {
        uint64_t d = key - L.dense_min;
        if (d >= L.dense_span) return 0ull;
        uint32_t e = L.dense[d];
        if (e == 0) return 0ull;
        if (L.direct) return (uint64_t(e - 1) << 32) | 1ull;
        uint32_t st = L.ustart[e - 1];
        return (uint64_t(st) << 32) | uint64_t(L.ustart[e] - st);
    }
    return probe_one(L.slots, L.cap, L.shift, key);
}

__global__ void fill_dense_kernel(const uint64_t *skeys, const uint32_t *ustart, const uint32_t *perm, uint32_t U, uint64_t dmin,
                                  uint32_t *dense, int direct) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; u < int64_t(U); u += stride) {
        uint32_t j = ustart[u];
        dense[skeys[j] - dmin] = direct ? perm[j] + 1u : uint32_t(u) + 1u;
    }
}

// Unique build keys: one lookup per probe row → match bitmap (the KEEP mask of the compaction
// kernels), 4-byte build row per probe row, per-tile match counts.  Wave per 4096-row tile.
__global__ void __launch_bounds__(256) probe_unique_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, Lookup L, uint64_t *keep,
                                                           uint32_t *bidx, uint32_t *tile_counts) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
        // (issuing the first probe of 16 keys before examining any was measured slower, 2.41 -> 2.63 ms: the hashed probe is
        // bound by line fetches — 10^8 x 128 B at ≈5 TB/s — not by latency, and the extra registers cost occupancy)
#pragma unroll 2
        for (int k0 = 0; k0 < TILE_WORDS; k0 += 8) {
            uint64_t key[8], meta[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                key[k] = rkeys[row < last ? row : last];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) meta[k] = lookup_meta(L, key[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                bool hit = row < n && meta[k] != 0ull;
                uint64_t kw = __ballot(hit);
                if (row < n) bidx[row] = uint32_t(meta[k] >> 32);
                if (row0 + int64_t(k0 + k) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0 + k] = kw;
                total += __popcll(kw);
            }
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

// the same for hashed tables, wave-cooperatively (see probe_pairs_kernel): 8 lanes read the 8 slots of a key's bucket — one
// coalesced line per key, the loads of 8 sub-steps in flight together — and a ballot finds the match
__global__ void __launch_bounds__(256) probe_unique_coop_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, const ulonglong2 *tab, uint32_t cap, int shift,
                                                                uint64_t *keep, uint32_t *bidx, uint32_t *tile_counts) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    const int my_t = lane_id() >> 3, my_g = lane_id() & 7;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
        for (int k0 = 0; k0 < TILE_WORDS; ++k0) {
            const int64_t row = row0 + int64_t(k0) * 64 + lane_id();
            const uint64_t key = __builtin_nontemporal_load(&rkeys[row < last ? row : last]);
            uint64_t kg[8];
            ulonglong2 s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                kg[t] = (uint64_t)__shfl((unsigned long long)key, t * 8 + (lane_id() >> 3), 64);
                s[t] = tab[home_slot(kg[t], shift) + uint32_t(lane_id() & 7)];
            }
            uint64_t meta = 0;
            bool settled = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint64_t m = __ballot(s[t].y != 0ull && s[t].x == kg[t]), f = __ballot(s[t].y == 0ull);
                const uint32_t mb = uint32_t(m >> (8 * my_g)) & 0xFFu, fb = uint32_t(f >> (8 * my_g)) & 0xFFu;
                const int src = 8 * my_g + (mb ? __ffs(int(mb)) - 1 : 0);
                const uint64_t mt = (uint64_t)__shfl((unsigned long long)s[t].y, src, 64);
                if (t == my_t) {
                    meta = mb ? mt : 0ull;
                    settled = mb != 0 || fb != 0;
                }
            }
            if (!settled) { // a full bucket without the key: the following buckets, a whole bucket per round trip (see probe_pairs_kernel)
                uint32_t sl = (home_slot(key, shift) + 8u) & (cap - 1);
                bool done = false;
                for (uint32_t p = 8; p < cap && !done; p += 8) {
                    ulonglong2 c[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) c[i] = tab[sl + uint32_t(i)];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (done) continue;
                        if (c[i].y == 0ull) done = true;
                        else if (c[i].x == key) { meta = c[i].y; done = true; }
                    }
                    sl = (sl + 8u) & (cap - 1);
                }
            }
            const bool hit = row < n && meta != 0ull;
            const uint64_t kw = __ballot(hit);
            if (row < n) bidx[row] = uint32_t(meta >> 32);
            if (row0 + int64_t(k0) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0] = kw;
            total += __popcll(kw);
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

// ---- unique + dense + plain payload: fused probe
__global__ void scatter_dense_payload_kernel(const uint64_t *keys, int64_t n, uint64_t dmin, const uint64_t *src, uint64_t *dst,
                                             uint32_t *presence) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        uint64_t d = keys[r] - dmin;
        if (dst) dst[d] = src[r];
        if (presence) atomicOr(&presence[d >> 5], 1u << (d & 31));
    }
}

// unsigned min / max of (value ^ flip) over a column (flip = sign bit for Int64 → order as signed)
__global__ void __launch_bounds__(256) minmax_u64_kernel(const uint64_t *v, int64_t n, uint64_t flip, unsigned long long *out_min, unsigned long long *out_max) {
    uint64_t mn = ~0ull, mx = 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const uint64_t x = v[i] ^ flip;
        mn = x < mn ? x : mn;
        mx = x > mx ? x : mx;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t a = __shfl_down(mn, d, 64), b = __shfl_down(mx, d, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane_id() == 0) {
        atomicMin(out_min, (unsigned long long)mn);
        atomicMax(out_max, (unsigned long long)mx);
    }
}
__global__ void scatter_dense_payload32_kernel(const uint64_t *keys, int64_t n, uint64_t dmin, const uint64_t *src, uint64_t base, uint32_t *dst,
                                               uint32_t *presence) {
    int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n; r += stride) {
        uint64_t d = keys[r] - dmin;
        dst[d] = uint32_t(src[r] - base);
        if (presence) atomicOr(&presence[d >> 5], 1u << (d & 31));
    }
}

// pass 1: match bitmap + per-tile counts (no build-row output).
// MODE 0: every key of [min, min+span) is present → a range check, no memory access at all;
// MODE 1: presence bitmap staged in LDS (span/8 bytes ≤ 128 KB: random LDS reads instead of one L2
//         request per probe row); MODE 2: presence bitmap read from global memory.
template <int MODE>
__global__ void __launch_bounds__(1024) probe_presence_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, const uint32_t *presence,
                                                              uint64_t dmin, uint64_t span, uint64_t *keep, uint32_t *tile_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lp = reinterpret_cast<uint32_t *>(smem);
    if (MODE == 1) {
        const uint32_t words = uint32_t((span + 31) / 32);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lp[i] = presence[i];
        __syncthreads();
    }
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
#pragma unroll 2
        for (int k0 = 0; k0 < TILE_WORDS; k0 += 8) {
            uint64_t key[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                key[k] = __builtin_nontemporal_load(&rkeys[row < last ? row : last]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int64_t row = row0 + int64_t(k0 + k) * 64 + lane_id();
                uint64_t d = key[k] - dmin;
                bool hit = row < n && d < span;
                if (MODE == 1) hit = hit && ((lp[d >> 5] >> (d & 31)) & 1u);
                if (MODE == 2) hit = hit && ((presence[d >> 5] >> (d & 31)) & 1u);
                uint64_t kw = __ballot(hit);
                if (row0 + int64_t(k0 + k) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0 + k] = kw;
                total += __popcll(kw);
            }
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

struct FusedCols {
    int32_t n;
    int32_t pad;
    int32_t kind[MAX_JOIN_COLS];        // 0: probe-side column (coalesced copy), 1: build key (= probe key), 2: build payload (gather),
                                        // 3: build payload packed as uint32 offsets from base[] (gather), 4: as bits[]-bit offsets
    const uint64_t *src[MAX_JOIN_COLS]; // kind 0: probe column; kind 2/3: key-ordered build column
    uint64_t *dst[MAX_JOIN_COLS];
    uint64_t base[MAX_JOIN_COLS];
    int32_t bits[MAX_JOIN_COLS]; // kind 4: bits per entry
};

// pass 2: one read of the probe keys, every output column written in probe order
// `bidx` null: a build payload is addressed by key - dmin (key-ordered dense columns); non-null: by the build row recorded
// per probe row by probe_unique_kernel (hashed unique keys), gathered from the build columns themselves.
// IDENT (the optimistic form of a PK-FK join, see probe): no keep bitmap and no offsets — every probe row is taken to match, output row =
// probe row; a key outside [dmin, dmin + span) raises *miss and the host discards the output.
template <int FW_B, bool IDENT = false> // FW_B: rows per lane in flight
__global__ void __launch_bounds__(256) join_fused_write_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, const uint64_t *keep,
                                                               const uint64_t *tile_offsets, uint64_t dmin, const uint32_t *bidx, FusedCols fc,
                                                               uint64_t span, int *miss) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t nwords = (n + 63) / 64;
    const int64_t last = n - 1;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        int64_t w = tile * TILE_WORDS + lane_id();
        uint64_t my_word = 0;
        uint32_t tot = 0, my_off = 0;
        uint64_t base = 0;
        if (IDENT) {
            // some wave (or the sampling kernel ahead of this one) found a foreign key without its primary key: the host discards the
            // output, so stop writing it (checked once per 4096-row tile; the flag only ever goes from 0 to 1)
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
        }
        if (!IDENT) {
            my_word = w < nwords ? keep[w] : 0;
            my_off = wave_exclusive_scan(uint32_t(__popcll(my_word)), tot);
            if (tot == 0) continue; // no probe row of this tile matched (wave-uniform): nothing of it is read again
            base = tile_offsets[tile];
        }
        for (int k0 = 0; k0 < TILE_WORDS; k0 += FW_B) {
            uint64_t key[FW_B];
            uint32_t pos[FW_B]; // position inside the tile's output range
            uint32_t kept = 0;
#pragma unroll
            for (int k = 0; k < FW_B; ++k) {
                int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                key[k] = __builtin_nontemporal_load(&rkeys[row < last ? row : last]); // streamed once: keep L2 for the gather
            }
            if (IDENT) {
                bool bad = false;
#pragma unroll
                for (int k = 0; k < FW_B; ++k) {
                    const int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                    const bool in = row < n;
                    const bool ok = key[k] - dmin < span;
                    bad = bad || (in && !ok);
                    pos[k] = uint32_t(row - tile * TILE_ROWS);
                    kept |= uint32_t(in && ok) << k;
                }
                if (bad) *miss = 1; // plain store of a constant
                base = uint64_t(tile) * TILE_ROWS;
            } else {
#pragma unroll
                for (int k = 0; k < FW_B; ++k) {
                    uint64_t word = bcast64(my_word, k0 + k);
                    pos[k] = bcast32(my_off, k0 + k) + __popcll(word & lanemask_lt());
                    kept |= uint32_t((word >> lane_id()) & 1) << k;
                }
            }
            uint64_t gix[FW_B]; // gather index of a build payload
#pragma unroll
            for (int k = 0; k < FW_B; ++k) {
                if (bidx) {
                    int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                    gix[k] = bidx[row < last ? row : last];
                } else gix[k] = key[k] - dmin;
            }
            for (int c = 0; c < fc.n; ++c) {
                const uint64_t *__restrict__ src = fc.src[c];
                uint64_t *__restrict__ dst = fc.dst[c] + base;
                const int kind = fc.kind[c];
                uint64_t v[FW_B];
                if (kind == 0) {
#pragma unroll
                    for (int k = 0; k < FW_B; ++k) {
                        int64_t row = (tile * TILE_WORDS + k0 + k) * 64 + lane_id();
                        v[k] = __builtin_nontemporal_load(&src[row < last ? row : last]);
                    }
                } else if (kind == 1) {
#pragma unroll
                    for (int k = 0; k < FW_B; ++k) v[k] = key[k];
                } else if (kind == 2) {
#pragma unroll
                    for (int k = 0; k < FW_B; ++k) v[k] = src[(kept >> k) & 1 ? gix[k] : 0];
                } else if (kind == 3) {
                    const uint32_t *__restrict__ src32 = reinterpret_cast<const uint32_t *>(src);
                    const uint64_t b0 = fc.base[c];
#pragma unroll
                    for (int k = 0; k < FW_B; ++k) v[k] = b0 + src32[(kept >> k) & 1 ? gix[k] : 0];
                } else { // kind 4: `bits` per entry (<= 25): one unaligned 4-byte load holds the entry wherever it starts (the table is padded)
                    const uint8_t *__restrict__ src8 = reinterpret_cast<const uint8_t *>(src);
                    const uint64_t b0 = fc.base[c];
                    const uint32_t nb = uint32_t(fc.bits[c]), mask = (1u << nb) - 1u;
#pragma unroll
                    for (int k = 0; k < FW_B; ++k) {
                        const uint64_t bit = ((kept >> k) & 1 ? gix[k] : 0) * nb;
                        uint32_t x;
                        __builtin_memcpy(&x, src8 + (bit >> 3), 4);
                        v[k] = b0 + ((x >> (uint32_t(bit) & 7u)) & mask);
                    }
                }
#pragma unroll
                for (int k = 0; k < FW_B; ++k)
                    if ((kept >> k) & 1) __builtin_nontemporal_store(v[k], &dst[pos[k]]);
            }
        }
    }
}

// ahead of the optimistic one-pass probe: 2^16 probe keys spread evenly over the column, tested against the primary key's range.  A
// foreign key column that misses on any noticeable fraction of its rows is caught here in ~10 µs — the one-pass kernel behind
// it then leaves at once (it reads the flag before its first tile) instead of writing an output the host would discard
__global__ void __launch_bounds__(256) join_sample_range_kernel(const uint64_t *rkeys, int64_t n, uint64_t dmin, uint64_t span, int *miss) {
    const int64_t samples = int64_t(gridDim.x) * blockDim.x;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t row = n <= samples ? i : int64_t((__int128)(i) * n / samples);
    const bool bad = row < n && !(rkeys[row < n ? row : n - 1] - dmin < span);
    if (__ballot(bad) && lane_id() == 0) *miss = 1;
}

// ---- unique hashed keys with ONE plain payload column: the lookup IS the gather.  Pass 1 of the two-pass probe looks every
// probe key up in the {key, payload} table (one random 16-byte access) and writes the payload word per probe row next to the
// match bitmap; the payload is then just another probe-side column that pass 2 (join_fused_write_kernel) streams and compacts —
// no build-row list, no second random access per row (the {key, row} form gathers every payload column by build row in pass 2).
// (A single-pass probe — decoupled look-back over per-tile counts, flat or hierarchical, publish-early / consume-a-tile-later —
// was built and measured: the fused kernel runs C4 in 1.00-1.08 ms with the placement given, 1.4-1.6 ms with any of the
// look-back variants: with 8 XCDs every publish / poll is a 2-5 µs fabric round trip per 512-1024-row tile and pollers eat the
// bandwidth the gathers need.  Two passes without inter-workgroup traffic are faster here.)
// Wave-cooperative probing: the table is read in 8-slot buckets = one 128-byte line.  A wave looks up its 64 keys in 8 sub-steps
// of 8 keys: lane (g, i) loads slot i of the bucket of sub-step key g — one coalesced line per key, all 8 sub-steps' loads in
// flight together — and a ballot finds the slot that matches.  Per-lane probing fetches the same one line per key but then walks
// collisions with dependent, divergent loads (2.4-3.0 ms per 1e8 keys against 1.6 ms for the bare random reads).
__global__ void __launch_bounds__(256) probe_pairs_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, const ulonglong2 *tab, uint32_t cap, int shift,
                                                          uint64_t filler, uint64_t *keep, uint64_t *payload, uint32_t *tile_counts) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    const int my_t = lane_id() >> 3, my_g = lane_id() & 7; // this lane owns key my_g of sub-step my_t; as a loader it reads slot my_g... of group lane>>3
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
        for (int k0 = 0; k0 < TILE_WORDS; ++k0) {
            const int64_t row = row0 + int64_t(k0) * 64 + lane_id();
            const uint64_t key = __builtin_nontemporal_load(&rkeys[row < last ? row : last]);
            uint64_t kg[8];
            ulonglong2 s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { // sub-step t serves the keys of lanes 8t .. 8t+7; this lane loads for key group lane >> 3
                kg[t] = (uint64_t)__shfl((unsigned long long)key, t * 8 + (lane_id() >> 3), 64);
                const uint32_t bucket = home_slot(kg[t], shift);
                s[t] = tab[bucket + uint32_t(lane_id() & 7)];
            }
            uint64_t pay = 0;
            bool hit = false, settled = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint64_t m = __ballot(s[t].x == kg[t]), f = __ballot(s[t].x == filler);
                const uint32_t mb = uint32_t(m >> (8 * my_g)) & 0xFFu, fb = uint32_t(f >> (8 * my_g)) & 0xFFu; // the bucket of this lane's own key, if t is its sub-step
                const int src = 8 * my_g + (mb ? __ffs(int(mb)) - 1 : 0);
                const uint64_t pl = (uint64_t)__shfl((unsigned long long)s[t].y, src, 64);
                if (t == my_t) {
                    hit = mb != 0;
                    pay = pl;
                    settled = hit || fb != 0; // found, or the bucket has a free slot: the key is not in the table
                }
            }
            if (!settled) { // a full bucket without the key — the following buckets, alone, a whole bucket (eight loads issued together) per round trip
                uint32_t sl = (home_slot(key, shift) + 8u) & (cap - 1);
                bool done = false;
                for (int hop = 0; hop < 2 * UNIQUE_MAX_PROBE / 8 && !done; ++hop) {
                    ulonglong2 c[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) c[i] = tab[sl + uint32_t(i)]; // (buckets are 8-slot aligned, cap is a multiple of 8)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (done) continue;
                        if (c[i].x == key) { hit = true; pay = c[i].y; done = true; }
                        else if (c[i].x == filler) done = true;
                    }
                    sl = (sl + 8u) & (cap - 1);
                }
            }
            hit = hit && row < n && key != filler;
            const uint64_t kw = __ballot(hit);
            if (row < n) __builtin_nontemporal_store(hit ? pay : 0ull, &payload[row]);
            if (row0 + int64_t(k0) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0] = kw;
            total += __popcll(kw);
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

// the same over the packed table: a bucket is 16 eight-byte slots = the same one line, lane (g, i) loads slots 2i and 2i + 1
__global__ void __launch_bounds__(256) probe_packed_kernel(const uint64_t *rkeys, int64_t n, int64_t ntiles, const ulonglong2 *tab, PackedPairs pp, uint64_t *keep,
                                                           uint64_t *payload, uint32_t *tile_counts) {
    const int waves_per_block = blockDim.x / 64;
    const int64_t last = n - 1;
    const int my_t = lane_id() >> 3, my_g = lane_id() & 7;
    const uint64_t pmask = (1ull << pp.pbits) - 1ull;
    for (int64_t tile = int64_t(blockIdx.x) * waves_per_block + threadIdx.x / 64; tile < ntiles;
         tile += int64_t(gridDim.x) * waves_per_block) {
        const int64_t row0 = tile * TILE_ROWS;
        uint32_t total = 0;
        for (int k0 = 0; k0 < TILE_WORDS; ++k0) {
            const int64_t row = row0 + int64_t(k0) * 64 + lane_id();
            const uint64_t key = __builtin_nontemporal_load(&rkeys[row < last ? row : last]);
            const uint64_t kd = key - pp.kmin;
            uint64_t kdg[8];
            ulonglong2 s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint64_t kg = (uint64_t)__shfl((unsigned long long)key, t * 8 + (lane_id() >> 3), 64);
                kdg[t] = kg - pp.kmin;
                s[t] = tab[packed_home(kg, pp.nb) * (PACKED_BUCKET / 2) + uint32_t(lane_id() & 7)];
            }
            uint64_t pay = 0;
            bool hit = false, settled = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool mx = (s[t].x >> pp.pbits) == kdg[t], my = (s[t].y >> pp.pbits) == kdg[t];
                const uint64_t m = __ballot(mx || my), f = __ballot(s[t].x == ~0ull || s[t].y == ~0ull);
                const uint32_t mb = uint32_t(m >> (8 * my_g)) & 0xFFu, fb = uint32_t(f >> (8 * my_g)) & 0xFFu;
                const int src = 8 * my_g + (mb ? __ffs(int(mb)) - 1 : 0);
                const uint64_t pl = (uint64_t)__shfl((unsigned long long)(mx ? s[t].x : s[t].y), src, 64);
                if (t == my_t) {
                    hit = mb != 0;
                    pay = pl;
                    settled = hit || fb != 0;
                }
            }
            if (!settled) { // a full bucket without the key (2-4 % of the buckets at load 0.6): the following buckets, alone — a whole bucket
                // per round trip (its eight 16-byte loads issued together, then examined in slot order): walking slot by slot made a
                // probe side of mostly absent keys twice as slow as one that matches (a wave waits for its slowest lane)
                uint32_t b = packed_home(key, pp.nb) + 1;
                bool done = false;
                for (int hop = 0; hop < 2 * UNIQUE_MAX_PROBE / PACKED_BUCKET && !done; ++hop) {
                    if (b == pp.nb) b = 0;
                    ulonglong2 w[PACKED_BUCKET / 2];
#pragma unroll
                    for (int i = 0; i < PACKED_BUCKET / 2; ++i) w[i] = tab[size_t(b) * (PACKED_BUCKET / 2) + i];
#pragma unroll
                    for (int i = 0; i < PACKED_BUCKET / 2; ++i) {
                        if (done) continue;
                        if ((w[i].x >> pp.pbits) == kd) { hit = true; pay = w[i].x; done = true; }
                        else if (w[i].x == ~0ull) done = true;
                        else if ((w[i].y >> pp.pbits) == kd) { hit = true; pay = w[i].y; done = true; }
                        else if (w[i].y == ~0ull) done = true;
                    }
                    ++b;
                }
            }
            hit = hit && row < n && kd <= pp.kspan; // (a key outside the build range shifts to bits no stored word has — except the empty word's)
            const uint64_t kw = __ballot(hit);
            if (row < n) __builtin_nontemporal_store(hit ? pp.pbase + (pay & pmask) : 0ull, &payload[row]);
            if (row0 + int64_t(k0) * 64 < n && lane_id() == 0) keep[tile * TILE_WORDS + k0] = kw;
            total += __popcll(kw);
        }
        if (lane_id() == 0) tile_counts[tile] = total;
    }
}

// 16 rows per lane in flight: the kernel is bound by the latency of its gathers, and memory-level parallelism per wave
// beats occupancy (A/B on one box, C4: 4 rows/lane (70 VGPRs, 7 waves/SIMD) 1.53 ms, 8 (116, 4) 1.33 ms, 16 (210, 2)
// 1.24 ms, 32 (256, 1) 1.31 ms)
constexpr int FUSED_WRITE_ROWS = 16;

// pass 1: one table lookup per probe row; records meta and per-tile totals
__global__ void __launch_bounds__(JT_BLOCK) probe_count_kernel(const uint64_t *rkeys, int64_t n, Lookup L, uint64_t *pmeta,
                                                               uint32_t *tile_counts, int *flags) {
    __shared__ uint64_t wave_tot[JT_BLOCK / 64];
    for (int64_t tile = blockIdx.x; tile * JT_ROWS < n; tile += gridDim.x) {
        uint64_t keys[JT_ITERS];
#pragma unroll
        for (int it = 0; it < JT_ITERS; ++it) {
            int64_t i = tile * JT_ROWS + int64_t(it) * JT_BLOCK + threadIdx.x;
            keys[it] = i < n ? rkeys[i] : 0;
        }
        uint64_t local = 0;
#pragma unroll
        for (int it = 0; it < JT_ITERS; ++it) {
            int64_t i = tile * JT_ROWS + int64_t(it) * JT_BLOCK + threadIdx.x;
            if (i < n) {
                uint64_t m = lookup_meta(L, keys[it]);
                pmeta[i] = m;
                local += m & 0xFFFFFFFFull;
            }
        }
        for (int d = 32; d > 0; d >>= 1) local += __shfl_down((unsigned long long)local, d, 64);
        if (lane_id() == 0) wave_tot[threadIdx.x / 64] = local;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < JT_BLOCK / 64; ++w) t += wave_tot[w];
            if (t > 0xFFFFFFFFull) {
                atomicOr(&flags[NQE_FLAG_TABLE_FULL], 1);
                t = 0;
            }
            tile_counts[tile] = uint32_t(t);
        }
        __syncthreads();
    }
}

// dst[i] = src[perm[i]]: a payload column in sorted-row order (build side, once)
__global__ void __launch_bounds__(256) permute_words_kernel(const uint64_t *src, const uint32_t *perm, int64_t n, uint64_t *dst) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) dst[i] = src[perm[i]];
}

struct JoinCols {
    int32_t n;
    int32_t n_left;
    int32_t need_perm;               // some left column is addressed by build row (else: all by position in the sorted row list)
    int32_t by_pos[MAX_JOIN_COLS];   // left column k: src is the `perm`-ordered copy, addressed by start + match number
    const void *src[MAX_JOIN_COLS];
    const uint8_t *src_valid[MAX_JOIN_COLS];
    int32_t dtype[MAX_JOIN_COLS];
    uint64_t *dst_words[MAX_JOIN_COLS];
    uint8_t *dst_bool_bytes[MAX_JOIN_COLS];
    uint8_t *dst_valid_bytes[MAX_JOIN_COLS];
};

// pass 2, output-driven ("load-balanced expansion"): a tile of probe rows is scanned in LDS; lane j of the
// workgroup then produces OUTPUT row base+j: its probe row is found by binary search in the tile's offsets,
// its match number m = j - offset[row], its build row = perm[start + m].  Consecutive lanes write consecutive
// output rows of every column (coalesced), probe-row-major with ascending build row inside a probe row —
// exactly the order of the reference's outer_pos/inner_pos (hash_join.rs:86-101).
constexpr int PW_TILE = 1024; // probe rows per tile (= JT_ROWS / 4); tile_offsets are per JT_ROWS, so 4 sub-tiles share one base
// PLAIN: every source column is a plain 8-byte column without validity written as words (C4 with duplicate build keys: the whole
// output): no dtype dispatch, validity test or byte-array branches in the per-column loop
template <bool PLAIN>
__global__ void __launch_bounds__(JT_BLOCK) probe_write_kernel(const uint64_t *pmeta, int64_t n, const uint64_t *tile_offsets,
                                                               const uint32_t *perm, int direct, JoinCols jc) {
    __shared__ uint32_t off[PW_TILE + 1];
    __shared__ uint32_t startv[PW_TILE];
    __shared__ uint32_t wave_tot[JT_BLOCK / 64];
    constexpr int RPT = PW_TILE / JT_BLOCK; // probe rows per thread
    for (int64_t tile = blockIdx.x; tile * JT_ROWS < n; tile += gridDim.x) {
        uint64_t out_base = tile_offsets[tile];
        for (int sub = 0; sub < JT_ROWS / PW_TILE; ++sub) {
            const int64_t row0 = tile * JT_ROWS + int64_t(sub) * PW_TILE;
            if (row0 >= n) break;
            // ---- exclusive scan of the match counts of this sub-tile (thread t owns RPT consecutive probe rows)
            uint32_t cnt[RPT], local = 0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                int64_t i = row0 + int64_t(threadIdx.x) * RPT + r;
                uint64_t m = i < n ? pmeta[i] : 0ull;
                cnt[r] = uint32_t(m & 0xFFFFFFFFull);
                startv[threadIdx.x * RPT + r] = uint32_t(m >> 32);
                local += cnt[r];
            }
            uint32_t wtot;
            uint32_t ex = wave_exclusive_scan(local, wtot);
            if (lane_id() == 63) wave_tot[threadIdx.x / 64] = wtot;
            __syncthreads();
            uint32_t pre = 0, total = 0;
            for (int w = 0; w < JT_BLOCK / 64; ++w) {
                if (w < int(threadIdx.x) / 64) pre += wave_tot[w];
                total += wave_tot[w];
            }
            uint32_t run = pre + ex;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                off[threadIdx.x * RPT + r] = run;
                run += cnt[r];
            }
            if (threadIdx.x == 0) off[PW_TILE] = total;
            __syncthreads();
            // ---- one lane per output row.  The lanes are shifted by the output position's offset inside its 128-byte line, so that every
            // wave's 64 consecutive words are four whole lines (round 5: a non-temporal store of a partial line is the costliest store there is)
            const int32_t head = int32_t(out_base & 15);
            for (int32_t j0 = -head; j0 < int32_t(total); j0 += JT_BLOCK * 4) {
                uint32_t prow[4], brow[4], bpos[4];
                bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int32_t js = j0 + q * JT_BLOCK + int32_t(threadIdx.x);
                    const uint32_t j = uint32_t(js);
                    live[q] = js >= 0 && js < int32_t(total);
                    uint32_t lo = 0, hi = PW_TILE; // largest lo with off[lo] <= j
                    uint32_t jj = live[q] ? j : 0;
#pragma unroll
                    for (int step = 0; step < 10; ++step) {
                        uint32_t mid = (lo + hi) >> 1;
                        bool go = off[mid] <= jj;
                        lo = go ? mid : lo;
                        hi = go ? hi : mid;
                    }
                    prow[q] = lo;
                    uint32_t mth = jj - off[lo];
                    bpos[q] = live[q] ? startv[lo] + mth : 0u;
                    brow[q] = live[q] ? (direct ? startv[lo] : (jc.need_perm ? perm[startv[lo] + mth] : 0u)) : 0u;
                }
                for (int c = 0; c < jc.n; ++c) {
                    const bool left = c < jc.n_left;
                    const void *src = jc.src[c];
                    if (PLAIN) {
                        const uint64_t *__restrict__ sw = static_cast<const uint64_t *>(src);
                        uint64_t *__restrict__ dw = jc.dst_words[c];
                        const bool by_pos = jc.by_pos[c] != 0;
                        uint64_t v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = sw[left ? int64_t(by_pos ? bpos[q] : brow[q]) : (live[q] ? row0 + prow[q] : int64_t(0))]; // (dead lanes read row 0)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (live[q]) __builtin_nontemporal_store(v[q], &dw[out_base + uint64_t(int64_t(j0) + q * JT_BLOCK + int64_t(threadIdx.x))]);
                        continue;
                    }
                    const uint8_t *sv = jc.src_valid[c];
                    const int dt = jc.dtype[c];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (!live[q]) continue;
                        int64_t srow = left ? int64_t(jc.by_pos[c] ? bpos[q] : brow[q]) : row0 + prow[q];
                        bool ok = sv ? get_bit(sv, srow) : true;
                        uint64_t v = load_word(src, dt, srow);
                        uint64_t pos = out_base + uint64_t(int64_t(j0) + q * JT_BLOCK + int64_t(threadIdx.x));
                        if (jc.dst_words[c]) jc.dst_words[c][pos] = ok ? v : 0;
                        if (jc.dst_bool_bytes[c]) jc.dst_bool_bytes[c][pos] = (ok && v) ? 1 : 0;
                        if (jc.dst_valid_bytes[c]) jc.dst_valid_bytes[c][pos] = ok ? 1 : 0;
                    }
                }
            }
            out_base += total;
            __syncthreads();
        }
    }
}

void check_key_types(int ldt, int rdt) {
    auto joinable = [](int d) { return d == NQE_INT64 || d == NQE_UINT64 || d == NQE_UTF8; };
    if (!joinable(ldt)) fail(NQE_ERR_NOT_IMPLEMENTED, "NotImplemented: join key type (hash_join.rs:161)");
    if (rdt < 0) return;
    if (!joinable(rdt)) fail(NQE_ERR_NOT_IMPLEMENTED, "NotImplemented: join key type (hash_join.rs:232)");
    if (rdt != ldt) fail(NQE_ERR_NOT_SUPPORTED, "join key types differ (downcast unwrap panics, hash_join.rs:83)");
}

// Sort-free build (see the kernels above).  Returns false — with `jt` untouched apart from buffers it will overwrite — when the
// keys turn out not to be unique; the caller then runs the sort-based build.  `plain_key`: the key column is a plain 8-byte
// column without validity (Utf8 keys arrive as codes and keep the generic probe).
bool build_unique_fast(nqe_ctx *ctx, nqe_join_table *jt, const nqe_table *left, const DevColumn &kc, bool plain_key) {
    const int64_t n = left->rows;
    const size_t ncols = left->cols.size();
    // ---- one round trip: min / max of the key and of every integer payload column (frame-of-reference packing)
    bool payload_plain = plain_key;
    for (size_t ci = 0; ci < ncols; ++ci)
        if (int(ci) != jt->left_key) payload_plain = payload_plain && is_word_type(left->cols[ci].dtype) && !left->cols[ci].validity;
    std::vector<int> mm_cols; // columns whose range is measured: [0] = the key
    mm_cols.push_back(-1);
    if (payload_plain)
        for (size_t ci = 0; ci < ncols; ++ci)
            if (int(ci) != jt->left_key && (left->cols[ci].dtype == NQE_INT64 || left->cols[ci].dtype == NQE_UINT64)) mm_cols.push_back(int(ci));
    const size_t K = mm_cols.size();
    BufRef mm = dev_alloc(ctx, K * 16 + 8); // [K mins][K maxs][descents of the key]
    NQE_HIP_CHECK(hipMemsetAsync(mm->ptr, 0xFF, K * 8, ctx->stream));
    NQE_HIP_CHECK(hipMemsetAsync(static_cast<char *>(mm->ptr) + K * 8, 0, K * 8 + 8, ctx->stream));
    MinMaxCols mc;
    std::memset(&mc, 0, sizeof(mc));
    for (size_t k = 0; k < K; ++k) {
        const DevColumn &c = mm_cols[k] < 0 ? kc : left->cols[size_t(mm_cols[k])];
        mc.src[k] = c.words();
        mc.flip[k] = (mm_cols[k] >= 0 && c.dtype == NQE_INT64) ? 0x8000000000000000ull : 0ull; // the key range is taken unsigned
    }
    launch(ctx, "join_build_minmax", minmax_cols_kernel, dim3(unsigned(std::min<int64_t>(n >= (int64_t(1) << 22) ? 4 * ctx->num_cus : 256, (n + 255) / 256)), unsigned(K)), dim3(256), 0, mc, n,
           (unsigned long long *)mm->ptr, (unsigned long long *)mm->ptr + K, (unsigned long long *)mm->ptr + 2 * K);
    std::vector<uint64_t> mmraw(K * 2 + 1), mmh(K * 2);
    NQE_HIP_CHECK(hipMemcpyAsync(mmraw.data(), mm->ptr, K * 16 + 8, hipMemcpyDeviceToHost, ctx->stream));
    sync(ctx);
    // (nearly) ascending keys: fewer than one descent per 64 rows — the scatter / finish form below is then coalesced at any size
    const bool ascending = getenv("NQE_JOIN_NO_ASCENDING") == nullptr && mmraw[K * 2] * 64 <= uint64_t(n);
    for (size_t k = 0; k < K; ++k) mmh[2 * k] = mmraw[k], mmh[2 * k + 1] = mmraw[K + k];
    const uint64_t kmin = mmh[0], kmax = mmh[1];
    const uint64_t span = kmax - kmin + 1; // 0 on wrap-around: not dense
    BufRef dupflag = dev_alloc_zero(ctx, 4);
    int dup = 0;
    if (span != 0 && span <= std::max<uint64_t>(4ull * uint64_t(n), 1024ull) && span < (1ull << 31)) {
        // ---- dense keys: direct-address table (+ key-ordered payload columns and the presence bitmap when everything is plain)
        // (zeroed by zero_tables() below on the paths that scatter into them; the two-level partitioned build writes every entry itself)
        BufRef dense = dev_alloc(ctx, size_t(span) * 4);
        BufRef presence = dev_alloc(ctx, size_t((span + 63) / 64) * 8);
        std::vector<std::pair<BufRef, size_t>> to_zero{{dense, size_t(span) * 4}, {presence, size_t((span + 63) / 64) * 8}};
        DensePayload dp;
        std::memset(&dp, 0, sizeof(dp));
        std::vector<BufRef> dense_cols(ncols);
        std::vector<int> dense_packed(ncols, 0);
        std::vector<uint64_t> dense_base(ncols, 0);
        const bool with_payload = payload_plain && span * 8 * ncols <= (size_t(8) << 30) && ncols <= size_t(MAX_JOIN_COLS);
        if (with_payload) {
            for (size_t ci = 0; ci < ncols; ++ci) {
                if (int(ci) == jt->left_key) continue;
                const DevColumn &pc = left->cols[ci];
                int packed = 0;
                for (size_t k = 1; k < mm_cols.size(); ++k)
                    if (mm_cols[k] == int(ci) && mmh[2 * k + 1] - mmh[2 * k] <= 0xffffffffull) { // value range within 32 bits → uint32 offsets
                        // … within 25 bits → exactly as many bits per entry as the range needs: the smaller the key-ordered table, the more
                        // of the probe's gathers hit the 4 MB L2 (10^6 keys of a 20-bit attribute: 2.5 MB instead of 4)
                        const uint64_t range = mmh[2 * k + 1] - mmh[2 * k];
                        int bits = 2;
                        while (bits < 32 && (range >> bits) != 0) ++bits;
                        packed = bits > 25 ? 1 : bits; // <= 25 bits: any entry lies inside one unaligned 4-byte window
                        dense_base[ci] = mmh[2 * k] ^ (pc.dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull);
                    }
                dense_packed[ci] = packed;
                const size_t col_bytes = packed >= 2 ? size_t((span + 63) / 64) * 8 * size_t(packed) + 16 // whole 64-entry groups (dense_finish_kernel)
                                                     : (packed ? size_t(span) * 4 + 8 : size_t(span) * 8);
                dense_cols[ci] = dev_alloc(ctx, col_bytes);
                if (packed) to_zero.push_back({dense_cols[ci], col_bytes});
                dp.src[dp.n] = pc.words();
                dp.dst[dp.n] = dense_cols[ci]->ptr;
                dp.base[dp.n] = dense_base[ci];
                dp.packed[dp.n] = dense_packed[ci];
                dp.n++;
            }
        }
        bool zeroed = false;
        auto zero_tables = [&]() {
            if (zeroed) return;
            for (auto &z : to_zero) NQE_HIP_CHECK(hipMemsetAsync(z.first->ptr, 0, z.second, ctx->stream));
            zeroed = true;
        };
        // (the one-kernel form with device atomics serves builds below 2^16 rows; forced on larger ones it measured 10^7 rows 0.67 -> 1.29 ms, 10^8 rows 11 -> 22 ms)
        const char *part_min_env = getenv("NQE_JOIN_PART_BUILD_MIN"); // read per call: the tests lower it for some builds only
        const int64_t part_min = part_min_env ? atoll(part_min_env) : (int64_t(1) << 25);
        bool part_done = false;
        if (n >= part_min && !ascending && (!with_payload || dp.n <= 15) && span <= 0xffffffffull) {
            // ---- partitioned form (see part_build_* above)
            const int nc = with_payload ? dp.n : 0;
            constexpr int slice_kb = 3072; // table bytes per partition (4-byte entries: 1.35 ms per 10^8 rows at 3 MB, 1.8 at 6, 2.5 at 24; 16-byte records: 2.1 either way)
            PartBuild pb;
            std::memset(&pb, 0, sizeof(pb));
            int shift = 10; // the widest slice of 4 + 8 nc bytes per entry within slice_kb, and no more than PB_MAX_PARTS of them
            while (shift < 31 && (uint64_t(2) << shift) * uint64_t(4 + 8 * nc) <= uint64_t(slice_kb) * 1024) ++shift;
            while (((span - 1) >> shift) + 1 > uint64_t(PB_MAX_PARTS)) ++shift;
            // two-level form (part_build_split / part_build_fill above): at most one payload word, a key range of at most PB_MAX_PARTS x
            // 2^fine_log2 fine bins (2.7 x 10^8 keys).  NQE_JOIN_PART_ONE_LEVEL=1 (read per call): the place pass (A/B)
            const bool two_level = nc <= 1 && span <= (uint64_t(PB_MAX_FINE) << PB_FILL_LOG2) && getenv("NQE_JOIN_PART_ONE_LEVEL") == nullptr &&
                                   size_t(PB_FILL_KEYS) * size_t(4 + 8 * nc) + 4096 <= ctx->lds_per_block && size_t(PB_MAX_FINE) * 4 <= ctx->lds_per_block;
            const int bins = two_level ? int(((span - 1) >> PB_FILL_LOG2) + 1) : 0;
            if (two_level) {
                // fine bins per partition (at most 64): as few partitions as keep about one workgroup of the second scatter per CU —
                // the first scatter slows down with its partition count (10^8 rows + a payload: 0.77 ms into 191 partitions, 0.86 into
                // 382, 1.09 into 763), the second one hardly cares how many bins a partition has (profiles/r06/probe_build_fine_bins.txt:
                // 10^8 keys 64 bins x 191 partitions 2.49 ms, 32 x 382 2.60; 2^25 keys 16 x 256 0.91 / 0.51, 32 x 128 0.91 / 0.57)
                int fl = PB_FINE_LOG2_MAX;
                auto parts_at = [&](int l) { return ((span - 1) >> (PB_FILL_LOG2 + l)) + 1; };
                while (fl > 0 && parts_at(fl) * 10 < uint64_t(ctx->num_cus) * 7 && parts_at(fl - 1) <= uint64_t(PB_MAX_PARTS)) --fl;
                while (parts_at(fl) > uint64_t(PB_MAX_PARTS)) ++fl;
                shift = PB_FILL_LOG2 + fl;
            }
            pb.keys = kc.words();
            pb.n = n;
            pb.dmin = kmin;
            pb.shift = shift;
            pb.parts = int(((span - 1) >> shift) + 1);
            pb.nc = nc;
            for (int c = 0; c < nc; ++c) pb.src[c] = dp.src[c];
            int rpt = nc <= 1 ? 8 : (nc <= 3 ? 4 : (nc <= 7 ? 2 : 1)); // 1024 * rpt tuples of 8 * (1 + nc) bytes in <= 128 KB of LDS
            // (gfx950 has 160 KB per workgroup; a device with less takes fewer rows per thread, and the one-kernel build when even one does not fit)
            while (rpt > 1 && size_t(PB_BLOCK) * size_t(rpt) * size_t(1 + nc) * 8 + size_t(PB_MAX_PARTS) * 12 > ctx->lds_per_block) rpt >>= 1;
            const bool part_lds_ok = size_t(PB_BLOCK) * size_t(rpt) * size_t(1 + nc) * 8 + size_t(PB_MAX_PARTS) * 12 <= ctx->lds_per_block;
            const int64_t tile = int64_t(PB_BLOCK) * rpt;
            pb.W = int(std::min<int64_t>(ctx->num_cus, (n + tile - 1) / tile));
            pb.chunk = ((n + pb.W - 1) / pb.W + tile - 1) / tile * tile;
            const size_t cells = size_t(pb.parts) * size_t(pb.W);
            // key-ordered records {row + 1, payload words} of an even number of words (16-byte aligned), zeroed: an entry nobody wrote is absent
            const int twp = nc ? (1 + nc + 1) / 2 * 2 : 0;
            BufRef counts, offsets, tuples, kord, tuples2, finehist, fine_start;
            bool part_oom = false;
            try { // the tuple stream and the records come on top of the table: when they do not fit, the one-kernel form below still may
                if (getenv("NQE_TEST_PART_BUILD_OOM")) fail(NQE_ERR_OUT_OF_MEMORY, "partitioned build (NQE_TEST_PART_BUILD_OOM)"); // tests: as if the allocation had failed
                if (!part_lds_ok) fail(NQE_ERR_OUT_OF_MEMORY, "partitioned build: the scatter tile does not fit this device's LDS");
                counts = dev_alloc(ctx, cells * 4);
                offsets = dev_alloc(ctx, (cells + 1) * 8);
                tuples = dev_alloc(ctx, size_t(n) * size_t(1 + nc) * 8 + 16);
                if (two_level) {
                    tuples2 = dev_alloc(ctx, size_t(n) * size_t(1 + nc) * 8 + 16);
                    finehist = dev_alloc(ctx, size_t(pb.W) * size_t(bins) * 4);
                    fine_start = dev_alloc(ctx, (size_t(bins) + 1) * 8);
                } else if (nc)
                    kord = dev_alloc(ctx, size_t(span) * size_t(twp) * 8);
            } catch (const Error &e) {
                if (e.code != NQE_ERR_OUT_OF_MEMORY) throw;
                part_oom = true;
            }
            if (!part_oom) {
            if (!two_level) zero_tables();
            if (kord) NQE_HIP_CHECK(hipMemsetAsync(kord->ptr, 0, size_t(span) * size_t(twp) * 8, ctx->stream));
            if (two_level)
                launch(ctx, "join_build_part_count", part_build_count_fine_kernel, dim3(unsigned(pb.W)), dim3(PB_BLOCK), size_t(bins) * 4, pb, (uint32_t *)counts->ptr, (uint32_t *)finehist->ptr, bins);
            else
                launch(ctx, "join_build_part_count", part_build_count_kernel, dim3(unsigned(pb.W)), dim3(PB_BLOCK), 0, pb, (uint32_t *)counts->ptr);
            exclusive_scan_u32_to_u64(ctx, (const uint32_t *)counts->ptr, (uint64_t *)offsets->ptr, int64_t(cells));
            const size_t shmem = size_t(tile) * size_t(1 + nc) * 8 + size_t(PB_MAX_PARTS) * 12;
            auto sk = rpt == 8 ? part_build_scatter_kernel<8> : (rpt == 4 ? part_build_scatter_kernel<4> : (rpt == 2 ? part_build_scatter_kernel<2> : part_build_scatter_kernel<1>));
            if (nc <= 1 && rpt == 8) sk = nc ? part_build_scatter1_kernel<1> : part_build_scatter1_kernel<0>; // (its tile in registers, the next one prefetched)
            launch(ctx, "join_build_part_scatter", sk, dim3(unsigned(pb.W)), dim3(PB_BLOCK), shmem, pb, (const uint64_t *)offsets->ptr, (uint64_t *)tuples->ptr);
            if (two_level) {
                launch(ctx, "join_build_part_fine_offsets", part_build_fine_offsets_kernel, dim3(unsigned(pb.parts)), dim3(256), 0, (const uint32_t *)finehist->ptr, pb.W, bins, pb.parts, pb.shift - PB_FILL_LOG2,
                       (const uint64_t *)offsets->ptr, (uint64_t *)fine_start->ptr);
                launch(ctx, "join_build_part_split", nc ? part_build_split_kernel<1> : part_build_split_kernel<0>, dim3(unsigned(std::min(pb.parts, 2 * ctx->num_cus))), dim3(PB_BLOCK),
                       size_t(PB_SPLIT_TILE) * size_t(1 + nc) * 8, pb, (const uint64_t *)offsets->ptr, (const uint64_t *)fine_start->ptr, bins, (const uint64_t *)tuples->ptr, (uint64_t *)tuples2->ptr);
                BufRef occupied = dev_alloc_zero(ctx, 8);
                launch(ctx, "join_build_part_fill", nc ? part_build_fill_kernel<1> : part_build_fill_kernel<0>, dim3(unsigned(std::min(bins, 2 * ctx->num_cus))), dim3(PB_BLOCK),
                       size_t(PB_FILL_KEYS) * size_t(4 + 8 * nc), (const uint64_t *)fine_start->ptr, bins, (const uint64_t *)tuples2->ptr, span, (uint32_t *)dense->ptr, (uint32_t *)presence->ptr, dp,
                       (unsigned long long *)occupied->ptr);
                dup = read_scalar(ctx, (const unsigned long long *)occupied->ptr) != (unsigned long long)n; // (also keeps the tuple streams alive until the kernels are done)
                part_done = true;
            } else {
            BufRef cursor = dev_alloc_zero(ctx, size_t(pb.parts) * 4);
            constexpr int place_by_block = 0; // (1: XCD = blockIdx % 8 instead of the hardware register — no difference measured)
            constexpr int place_chunk = PB_CHUNK;
            constexpr int place_bpc = 3;            // workgroups per CU (measured per 10^8 records: 8 -> 2.6 ms, 2-4 -> 2.1, 1 -> 3.1)
            launch(ctx, "join_build_part_place", part_build_place_kernel, dim3(unsigned(place_bpc * ctx->num_cus)), dim3(256), 0, pb, (const uint64_t *)offsets->ptr,
                   (const uint64_t *)tuples->ptr, (uint32_t *)dense->ptr, kord ? (uint64_t *)kord->ptr : (uint64_t *)nullptr, twp, (uint32_t *)cursor->ptr, place_by_block, uint32_t(place_chunk));
            BufRef occupied = dev_alloc_zero(ctx, 8);
            launch(ctx, "join_build_finish", (kord && (twp == 2 || twp == 4)) ? dense_finish_kernel<true> : dense_finish_kernel<false>, dim3(stream_grid(ctx, int64_t((span + 63) / 64), 4)),
                   dim3(256), 0, (uint32_t *)dense->ptr, span, (uint32_t *)presence->ptr, dp, (unsigned long long *)occupied->ptr,
                   kord ? (const uint64_t *)kord->ptr : (const uint64_t *)nullptr, twp);
            dup = read_scalar(ctx, (const unsigned long long *)occupied->ptr) != (unsigned long long)n; // (also keeps the tuples and records alive until the kernels are done)
            part_done = true;
            }
            }
        }
        if (!part_done) zero_tables();
        if (part_done) {
        } else if (n >= (int64_t(1) << 16) && (n < (int64_t(1) << 25) || ascending)) {
            // larger builds: scatter row numbers, then finish in key order (see dense_finish_kernel) — no device-scope atomics
            BufRef occupied = dev_alloc_zero(ctx, 8);
            launch(ctx, "join_build_dense", dense_scatter_rows_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), n, kmin, (uint32_t *)dense->ptr);
            launch(ctx, "join_build_finish", dense_finish_kernel<false>, dim3(stream_grid(ctx, int64_t((span + 63) / 64), 4)), dim3(256), 0, (uint32_t *)dense->ptr, span,
                   (uint32_t *)presence->ptr, dp, (unsigned long long *)occupied->ptr, (const uint64_t *)nullptr, 0);
            dup = read_scalar(ctx, (const unsigned long long *)occupied->ptr) != (unsigned long long)n;
        } else {
            launch(ctx, "join_build_dense", dense_unique_build_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), n, kmin, (uint32_t *)dense->ptr,
                   (uint32_t *)presence->ptr, dp, (int *)dupflag->ptr);
            dup = read_scalar(ctx, (const int *)dupflag->ptr);
        }
        if (dup) return false;
        jt->direct = true;
        jt->dense = dense;
        jt->dense_min = kmin;
        jt->dense_span = span;
        jt->cap = 0; // no hash table: every lookup goes through the direct-address table
        if (with_payload) {
            jt->presence = presence;
            jt->dense_cols = dense_cols;
            jt->dense_packed = dense_packed;
            jt->dense_base = dense_base;
            jt->dense_payload = true;
            jt->dense_full = (span == uint64_t(n));
        }
        return true;
    }
    // ---- sparse keys: open addressing, one slot per row
    uint32_t cap = 64;
    while (uint64_t(cap) < 2ull * uint64_t(n)) cap <<= 1;
    int lg = 0;
    while ((1u << lg) < cap) ++lg;
    BufRef slots = dev_alloc_zero(ctx, size_t(cap) * 16);
    launch(ctx, "join_build_insert", hashed_insert_rows_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), n, (ulonglong2 *)slots->ptr, cap, 64 - lg,
           (int *)dupflag->ptr);
    BufRef slotsp;
    uint64_t filler = 0;
    int pis_col = -1;
    const bool have_filler = kmax != ~0ull || kmin != 0ull; // a value outside [kmin, kmax] (unsigned) is not a build key
    auto bit_length = [](uint64_t x) {
        int b = 0;
        while (b < 64 && (x >> b) != 0) ++b;
        return b;
    };
    PackedPairs pp{};
    if (payload_plain && ncols == 2 && mm_cols.size() == 2) { // one integer payload column whose offset fits a word together with the key's
        const int kbits = std::max(1, bit_length(kmax - kmin)), pbits = std::max(1, bit_length(mmh[3] - mmh[2]));
        const uint64_t nb = uint64_t(n) * 5 / 48 + 1; // 16-slot buckets at load 0.6
        if (kbits + pbits <= 63 && nb * PACKED_BUCKET < (1ull << 31)) {
            pis_col = jt->left_key == 0 ? 1 : 0;
            pp.kmin = kmin;
            pp.kspan = kmax - kmin;
            pp.pbase = mmh[2] ^ (left->cols[size_t(pis_col)].dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull);
            pp.nb = uint32_t(nb);
            pp.pbits = pbits;
            slotsp = dev_alloc(ctx, size_t(nb) * PACKED_BUCKET * 8);
            NQE_HIP_CHECK(hipMemsetAsync(slotsp->ptr, 0xFF, size_t(nb) * PACKED_BUCKET * 8, ctx->stream));
            launch(ctx, "join_build_insert", packed_insert_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), left->cols[size_t(pis_col)].words(), n,
                   (unsigned long long *)slotsp->ptr, pp, (int *)dupflag->ptr);
        }
    }
    if (slotsp) { // (the packed insert has checked uniqueness itself)
    } else if (payload_plain && ncols == 2 && have_filler) { // exactly one payload column: it rides in the slot
        filler = kmax != ~0ull ? kmax + 1 : kmin - 1;
        pis_col = jt->left_key == 0 ? 1 : 0;
        slotsp = dev_alloc(ctx, size_t(cap) * 16);
        launch(ctx, "join_build_fill", fill_pairs_kernel, dim3(stream_grid(ctx, cap, 256)), dim3(256), 0, (ulonglong2 *)slotsp->ptr, cap, filler);
        launch(ctx, "join_build_insert", hashed_insert_pairs_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), left->cols[size_t(pis_col)].words(), n,
               (ulonglong2 *)slotsp->ptr, cap, 64 - lg, filler, (int *)dupflag->ptr);
    } else {
        launch(ctx, "join_build_check", hashed_check_unique_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), n, (const ulonglong2 *)slots->ptr,
               cap, 64 - lg, (int *)dupflag->ptr);
    }
    dup = read_scalar(ctx, (const int *)dupflag->ptr);
    if (dup) return false;
    jt->direct = true;
    jt->slots = slots;
    jt->cap = cap;
    jt->shift = 64 - lg;
    jt->slotsp = slotsp;
    jt->filler = filler;
    jt->pis_col = pis_col;
    jt->pp_bits = pp.pbits;
    jt->pp_nb = pp.nb;
    jt->pp_kmin = pp.kmin;
    jt->pp_kspan = pp.kspan;
    jt->pp_base = pp.pbase;
    return true;
}

std::unique_ptr<nqe_join_table> build_table(nqe_ctx *ctx, const nqe_table *left, int left_key) {
    if (left_key < 0 || size_t(left_key) >= left->cols.size()) fail(NQE_ERR_LOGICAL, "ColumnExpr must has name or idx");
    const DevColumn &kc_orig = left->cols[size_t(left_key)];
    check_key_types(kc_orig.dtype, -1);
    const int64_t n = left->rows;
    Utf8Dict dict;
    DevColumn kc_codes;
    if (kc_orig.dtype == NQE_UTF8) kc_codes = utf8_encode_build(ctx, kc_orig, &dict);
    const DevColumn &kc = kc_orig.dtype == NQE_UTF8 ? kc_codes : kc_orig;
    if (n >= (int64_t(1) << 32)) fail(NQE_ERR_NOT_SUPPORTED, "build side with 2^32 or more rows is not supported");
    auto jt = std::make_unique<nqe_join_table>();
    jt->ctx = ctx;
    jt->left_cols = left->cols;
    jt->left_rows = n;
    jt->key_dtype = kc_orig.dtype;
    jt->left_key = left_key;
    jt->dict = dict;

    if (n > 0 && build_unique_fast(ctx, jt.get(), left, kc, kc_orig.dtype != NQE_UTF8 && !kc.validity)) {
        if (getenv("NQE_DEBUG"))
            fprintf(stderr, "[nqe] join build (sort-free): n=%lld dense_span=%llu dense_payload=%d dense_full=%d cap=%u pis=%d\n", (long long)n,
                    (unsigned long long)jt->dense_span, int(jt->dense_payload), int(jt->dense_full), jt->cap, jt->pis_col);
        return jt;
    }
    // ---- sort-based build: duplicate keys (their matches must come out in ascending build row), or an empty build side
    BufRef idx = dev_alloc(ctx, size_t(n) * 4 + 8), skeys = dev_alloc(ctx, size_t(n) * 8 + 8);
    jt->perm = dev_alloc(ctx, size_t(n) * 4 + 8);
    uint32_t U = 0;
    BufRef ustart;
    if (n) {
        launch(ctx, "join_iota", iota_u32_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, (uint32_t *)idx->ptr, n);
        radix_sort_pairs_u64(ctx, kc.words(), (const uint32_t *)idx->ptr, (uint64_t *)skeys->ptr, (uint32_t *)jt->perm->ptr, n,
                             false);
        // run heads of the sorted keys → unique keys
        BufRef flags = dev_alloc(ctx, size_t(n + 1) * 4);
        launch(ctx, "join_mark_heads", mark_heads_kernel, dim3(stream_grid(ctx, n + 1, 256)), dim3(256), 0,
               (const uint64_t *)skeys->ptr, n, (uint32_t *)flags->ptr);
        BufRef offs = dev_alloc(ctx, size_t(n + 2) * 8);
        exclusive_scan_u32_to_u64(ctx, (const uint32_t *)flags->ptr, (uint64_t *)offs->ptr, n + 1);
        U = uint32_t(read_scalar(ctx, (const uint64_t *)offs->ptr + (n + 1)));
        ustart = dev_alloc(ctx, size_t(U + 1) * 4);
        launch(ctx, "join_fill_ustart", fill_ustart_kernel, dim3(stream_grid(ctx, n + 1, 256)), dim3(256), 0,
               (const uint32_t *)flags->ptr, (const uint64_t *)offs->ptr, n, U, (uint32_t *)ustart->ptr);
    }
    jt->direct = (int64_t(U) == n);
    uint32_t cap = 64;
    while (uint64_t(cap) < 2ull * U) cap <<= 1;
    int lg = 0;
    while ((1u << lg) < cap) ++lg;
    jt->cap = cap;
    jt->shift = 64 - lg;
    jt->slots = dev_alloc_zero(ctx, size_t(cap) * 16);
    if (U)
        launch(ctx, "join_insert", insert_unique_kernel, dim3(stream_grid(ctx, U, 256)), dim3(256), 0, (const uint64_t *)skeys->ptr,
               (const uint32_t *)ustart->ptr, (const uint32_t *)jt->perm->ptr, U, (ulonglong2 *)jt->slots->ptr, cap, jt->shift,
               jt->direct ? 1 : 0);
    // dense key range → direct-address table (keys are sorted unsigned: first/last are min/max)
    if (n > 0 && U > 0) {
        uint64_t kmin = read_scalar(ctx, (const uint64_t *)skeys->ptr);
        uint64_t kmax = read_scalar(ctx, (const uint64_t *)skeys->ptr + (n - 1));
        uint64_t span = kmax - kmin + 1; // 0 on wrap-around: not dense
        if (span != 0 && span <= std::max<uint64_t>(4ull * uint64_t(n), 1024ull) && span < (1ull << 31)) {
            jt->dense = dev_alloc_zero(ctx, size_t(span) * 4);
            jt->dense_min = kmin;
            jt->dense_span = span;
            launch(ctx, "join_fill_dense", fill_dense_kernel, dim3(stream_grid(ctx, U, 256)), dim3(256), 0, (const uint64_t *)skeys->ptr,
                   (const uint32_t *)ustart->ptr, (const uint32_t *)jt->perm->ptr, U, kmin, (uint32_t *)jt->dense->ptr, jt->direct ? 1 : 0);
            if (!jt->direct) jt->ustart = ustart;
            // unique keys and plain payload: key-ordered copies of the payload columns + presence bitmap
            bool plain = jt->direct && !kc.validity && kc_orig.dtype != NQE_UTF8;
            for (size_t ci = 0; ci < left->cols.size(); ++ci)
                if (int(ci) != left_key) plain = plain && is_word_type(left->cols[ci].dtype) && !left->cols[ci].validity;
            if (plain && span * 8 * left->cols.size() <= (size_t(8) << 30)) {
                jt->presence = dev_alloc_zero(ctx, size_t((span + 31) / 32) * 4);
                jt->dense_cols.resize(left->cols.size());
                jt->dense_packed.assign(left->cols.size(), 0);
                jt->dense_base.assign(left->cols.size(), 0);
                bool first = true;
                for (size_t ci = 0; ci < left->cols.size(); ++ci) {
                    if (int(ci) == left_key) continue;
                    const DevColumn &pc = left->cols[ci];
                    if (pc.dtype == NQE_INT64 || pc.dtype == NQE_UINT64) { // value range within 32 bits → uint32 offsets
                        const uint64_t flip = pc.dtype == NQE_INT64 ? 0x8000000000000000ull : 0ull;
                        BufRef mm = dev_alloc(ctx, 16);
                        const uint64_t init[2] = {~0ull, 0ull};
                        NQE_HIP_CHECK(hipMemcpyAsync(mm->ptr, init, 16, hipMemcpyHostToDevice, ctx->stream));
                        launch(ctx, "join_payload_minmax", minmax_u64_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, pc.words(), n, flip,
                               (unsigned long long *)mm->ptr, (unsigned long long *)mm->ptr + 1);
                        const uint64_t mn = read_scalar(ctx, (const uint64_t *)mm->ptr), mx = read_scalar(ctx, (const uint64_t *)mm->ptr + 1);
                        if (mx - mn <= 0xffffffffull) {
                            jt->dense_packed[ci] = 1;
                            jt->dense_base[ci] = mn ^ flip;
                            jt->dense_cols[ci] = dev_alloc_zero(ctx, size_t(span) * 4 + 8);
                            launch(ctx, "join_scatter_payload", scatter_dense_payload32_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, kc.words(), n,
                                   kmin, pc.words(), jt->dense_base[ci], (uint32_t *)jt->dense_cols[ci]->ptr,
                                   first ? (uint32_t *)jt->presence->ptr : (uint32_t *)nullptr);
                            first = false;
                            continue;
                        }
                    }
                    jt->dense_cols[ci] = dev_alloc(ctx, size_t(span) * 8);
                    launch(ctx, "join_scatter_payload", scatter_dense_payload_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                           kc.words(), n, kmin, left->cols[ci].words(), (uint64_t *)jt->dense_cols[ci]->ptr,
                           first ? (uint32_t *)jt->presence->ptr : (uint32_t *)nullptr);
                    first = false;
                }
                if (first) // key-only build side: presence bitmap only
                    launch(ctx, "join_scatter_payload", scatter_dense_payload_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                           kc.words(), n, kmin, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)jt->presence->ptr);
                jt->dense_payload = true;
                jt->dense_full = (span == uint64_t(U));
            }
        }
    }
    if (!jt->direct && n > 0) {
        // duplicate keys: plain payload columns in sorted-row order (see nqe_join_table::sorted_cols).  An optimisation on top of
        // everything the table needs (allocated above): a copy that does not fit is left out and the probe gathers that column
        // through the permutation instead (probe_table: need_perm)
        jt->sorted_cols.resize(left->cols.size());
        const bool test_oom = getenv("NQE_TEST_SORTED_COLS_OOM") != nullptr; // tests: as if the allocation had failed
        for (size_t ci = 0; ci < left->cols.size(); ++ci) {
            const DevColumn &pc = left->cols[ci];
            if (int(ci) == left_key || !is_word_type(pc.dtype) || pc.validity) continue;
            try {
                if (test_oom) fail(NQE_ERR_OUT_OF_MEMORY, "sorted payload copy (NQE_TEST_SORTED_COLS_OOM)");
                jt->sorted_cols[ci] = dev_alloc(ctx, size_t(n) * 8 + 8);
            } catch (const Error &e) {
                if (e.code != NQE_ERR_OUT_OF_MEMORY) throw;
                jt->sorted_cols[ci] = nullptr;
                continue;
            }
            launch(ctx, "join_permute_payload", permute_words_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, pc.words(), (const uint32_t *)jt->perm->ptr, n,
                   (uint64_t *)jt->sorted_cols[ci]->ptr);
        }
    }
    sync(ctx); // skeys/flags/ustart are released on return
    if (getenv("NQE_DEBUG"))
        fprintf(stderr, "[nqe] join build: n=%lld U=%u direct=%d dense_span=%llu dense_payload=%d dense_full=%d cap=%u\n", (long long)n, U, int(jt->direct),
                (unsigned long long)jt->dense_span, int(jt->dense_payload), int(jt->dense_full), jt->cap);
    return jt;
}

// the build-side key column of an equi-join's output can alias the probe-side one: plain integer keys of one type, where equal
// means identical bits (Float64 keys compare -0.0 == 0.0 with different bits; Utf8 keys are not 8-byte words)
static bool share_key_column(const DevColumn &left_key, const DevColumn &right_key) {
    return left_key.dtype == right_key.dtype && (left_key.dtype == NQE_INT64 || left_key.dtype == NQE_UINT64) && !left_key.validity &&
           !right_key.validity;
}

std::unique_ptr<nqe_table> probe_table(nqe_ctx *ctx, const nqe_join_table *jt, const nqe_table *right, int right_key) {
    if (right_key < 0 || size_t(right_key) >= right->cols.size()) fail(NQE_ERR_LOGICAL, "ColumnExpr must has name or idx");
    const DevColumn &rk_orig = right->cols[size_t(right_key)];
    check_key_types(jt->key_dtype, rk_orig.dtype);
    DevColumn rk_codes;
    if (rk_orig.dtype == NQE_UTF8) rk_codes = utf8_encode_probe(ctx, rk_orig, jt->dict);
    const DevColumn &rk = rk_orig.dtype == NQE_UTF8 ? rk_codes : rk_orig;
    const int64_t n = right->rows;
    const size_t ncols = jt->left_cols.size() + right->cols.size();
    if (ncols > size_t(MAX_JOIN_COLS)) fail(NQE_ERR_NOT_SUPPORTED, "join output wider than 32 columns");
    bool utf8_left = false, utf8_right = false;
    for (auto &c : jt->left_cols) utf8_left |= c.dtype == NQE_UTF8;
    for (auto &c : right->cols) utf8_right |= c.dtype == NQE_UTF8;

    Lookup L;
    std::memset(&L, 0, sizeof(L));
    L.slots = jt->slots ? (const ulonglong2 *)jt->slots->ptr : nullptr;
    L.cap = jt->cap;
    L.shift = jt->shift;
    L.dense = jt->dense ? (const uint32_t *)jt->dense->ptr : nullptr;
    L.ustart = jt->ustart ? (const uint32_t *)jt->ustart->ptr : nullptr;
    L.dense_min = jt->dense_min;
    L.dense_span = jt->dense_span;
    L.direct = jt->direct ? 1 : 0;

    bool right_plain = true;
    for (auto &c : right->cols) right_plain = right_plain && is_word_type(c.dtype) && !c.validity;
    // the output may reference the probe table's buffers only where an alias keeps them alive (the library's own memory) or the
    // caller has promised to (NQE_TABLE_IMMUTABLE); borrowed columns are copied — the caller may free them once the join returned
    bool probe_shareable = true;
    for (auto &c : right->cols) probe_shareable = probe_shareable && c.shareable();
    bool left_all_plain = ncols <= size_t(MAX_JOIN_COLS);
    for (auto &c : jt->left_cols) left_all_plain = left_all_plain && is_word_type(c.dtype) && !c.validity;
    if (jt->dense_payload && right_plain) {
        // `share_probe` (the optimistic form: output row = probe row): the probe-side columns of the output ARE the probe table's
        // columns — tables are immutable and their columns may share buffers — so only the build payloads are written
        const bool no_share_probe = getenv("NQE_JOIN_NO_SHARED_PROBE_COLUMNS") != nullptr; // (read per call: bench.py times both forms)
        auto build_out = [&](int64_t out_rows, FusedCols &fc, bool share_probe = false) {
        share_probe = share_probe && !no_share_probe && probe_shareable;
        auto out = std::make_unique<nqe_table>();
        out->ctx = ctx;
        out->rows = out_rows;
        std::memset(&fc, 0, sizeof(fc));
        // In an equi-join on an integer key the build key column of the output IS the probe key column of the output, bit for
        // bit: it is written once and the two output columns share the buffer (tables are immutable; 8 of the 32 output bytes
        // per row of C4 are never written)
        const bool share_key = share_key_column(jt->left_cols[size_t(jt->left_key)], rk);
        int share_pos = -1, right_key_pos = -1;
        for (size_t ci = 0; ci < jt->left_cols.size(); ++ci) {
            const DevColumn &c = jt->left_cols[ci];
            if (int(ci) == jt->left_key && share_key) {
                share_pos = int(out->cols.size());
                out->cols.push_back(DevColumn{});
                continue;
            }
            out->cols.push_back(make_word_column(ctx, c.dtype, out_rows, false));
            fc.kind[fc.n] = int(ci) == jt->left_key ? 1 : (jt->dense_packed[ci] >= 2 ? 4 : (jt->dense_packed[ci] ? 3 : 2));
            fc.bits[fc.n] = jt->dense_packed[ci];
            fc.base[fc.n] = int(ci) == jt->left_key ? 0 : jt->dense_base[ci];
            fc.src[fc.n] = int(ci) == jt->left_key ? nullptr : (const uint64_t *)jt->dense_cols[ci]->ptr;
            fc.dst[fc.n] = (uint64_t *)out->cols.back().values->ptr;
            fc.n++;
        }
        for (size_t cj = 0; cj < right->cols.size(); ++cj) {
            const DevColumn &c = right->cols[cj];
            if (share_probe) {
                out->cols.push_back(c);
                out->cols.back().null_count = 0;
                if (int(cj) == right_key) right_key_pos = int(out->cols.size()) - 1;
                continue;
            }
            out->cols.push_back(make_word_column(ctx, c.dtype, out_rows, false));
            if (int(cj) == right_key) right_key_pos = int(out->cols.size()) - 1;
            // the probe key column is in registers already (kind 1): loading it again as a probe-side column cost pass 2 8-15 %
            // (C4 1.13 -> 1.04 ms, a 10 %-match join 0.70 -> 0.60 ms).  Requesting one more probe-side column together with the
            // keys (33 more VGPRs) was neutral on top of that.
            fc.kind[fc.n] = int(cj) == right_key ? 1 : 0;
            fc.src[fc.n] = c.words();
            fc.dst[fc.n] = (uint64_t *)out->cols.back().values->ptr;
            fc.n++;
        }
        if (share_pos >= 0) {
            out->cols[size_t(share_pos)] = out->cols[size_t(right_key_pos)];
            out->cols[size_t(share_pos)].dtype = jt->left_cols[size_t(jt->left_key)].dtype;
        }
        return out;
        };
        // Optimistic form: a build side whose keys fill their range without gaps is a primary key; a foreign key then matches on
        // every row, the output row of a probe row is the probe row, and ONE pass (no presence pass, no scan, the probe keys read
        // once) writes everything — while checking every key against the range.  A key outside it discards the output, and this
        // join table takes the two-pass form from then on.
        // what the context remembers of this join (the table itself may be a fresh one: nqe_hash_join_execute builds per call)
        uint64_t jhint = 1469598103934665603ull;
        {
            const void *lp = jt->left_cols[size_t(jt->left_key)].values ? jt->left_cols[size_t(jt->left_key)].values->ptr : nullptr, *rp = rk.values->ptr;
            const int64_t lrows = jt->left_cols[size_t(jt->left_key)].length;
            auto mix = [&](const void *p, size_t nb) {
                const unsigned char *b = static_cast<const unsigned char *>(p);
                for (size_t i = 0; i < nb; ++i) jhint = (jhint ^ b[i]) * 1099511628211ull;
            };
            mix(&lp, sizeof(lp));
            mix(&lrows, sizeof(lrows));
            mix(&rp, sizeof(rp));
            mix(&n, sizeof(n));
        }
        if (ctx->join_hints.count(jhint)) jt->all_match_failed = true;
        if (jt->dense_full && !jt->all_match_failed && n > 0) {
            FusedCols fc;
            auto out = build_out(n, fc, true);
            BufRef miss = dev_alloc_zero(ctx, 4);
            const int64_t ntiles = (n + TILE_ROWS - 1) / TILE_ROWS;
            if (n >= (int64_t(1) << 20))
                launch(ctx, "join_sample_range", join_sample_range_kernel, dim3(256), dim3(256), 0, rk.words(), n, jt->dense_min, jt->dense_span, (int *)miss->ptr);
            launch(ctx, "join_fused_write", join_fused_write_kernel<FUSED_WRITE_ROWS, true>, dim3(stream_grid(ctx, ntiles, 4)), dim3(256), 0, rk.words(), n, ntiles,
                   (const uint64_t *)nullptr, (const uint64_t *)nullptr, jt->dense_min, (const uint32_t *)nullptr, fc, jt->dense_span, (int *)miss->ptr);
            if (!read_scalar(ctx, (const int *)miss->ptr)) return out;
            jt->all_match_failed = true;
            if (ctx->join_hints.size() >= 256) ctx->join_hints.clear();
            ctx->join_hints[jhint] = 1;
        }
        // PK–FK fast path: presence test + counts, scan, then one fused write of every output column
        KeepMask km;
        km.n = n;
        km.ntiles = (n + TILE_ROWS - 1) / TILE_ROWS;
        int64_t nwords = (n + 63) / 64;
        km.keep = dev_alloc(ctx, size_t(nwords) * 8 + 8);
        BufRef counts = dev_alloc(ctx, size_t(km.ntiles + 1) * 4);
        dim3 grid(stream_grid(ctx, km.ntiles, 4)), block(256);
        if (km.ntiles) {
            const size_t pbytes = size_t((jt->dense_span + 31) / 32) * 4;
            dim3 pgrid(stream_grid(ctx, km.ntiles, 16, 1)), pblock(1024);
            const uint32_t *pp = (const uint32_t *)jt->presence->ptr;
            uint64_t *kp = (uint64_t *)km.keep->ptr;
            uint32_t *cp = (uint32_t *)counts->ptr;
            if (jt->dense_full)
                launch(ctx, "join_probe_presence", probe_presence_kernel<0>, dim3(stream_grid(ctx, km.ntiles, 16, 2)), pblock, 0, rk.words(),
                       n, km.ntiles, pp, jt->dense_min, jt->dense_span, kp, cp);
            else if (pbytes <= 128 * 1024)
                launch(ctx, "join_probe_presence", probe_presence_kernel<1>, pgrid, pblock, pbytes, rk.words(), n, km.ntiles, pp,
                       jt->dense_min, jt->dense_span, kp, cp);
            else
                launch(ctx, "join_probe_presence", probe_presence_kernel<2>, dim3(stream_grid(ctx, km.ntiles, 16, 2)), pblock, 0, rk.words(),
                       n, km.ntiles, pp, jt->dense_min, jt->dense_span, kp, cp);
        }
        km = finish_mask(ctx, km, counts);
        FusedCols fc;
        auto out = build_out(km.total, fc, km.total == n); // (every probe row matched after all — a primary key with gaps, say: share the probe columns)
        if (km.ntiles && km.total > 0)
            launch(ctx, "join_fused_write", join_fused_write_kernel<FUSED_WRITE_ROWS>, grid, block, 0, rk.words(), n, km.ntiles, (const uint64_t *)km.keep->ptr,
                   (const uint64_t *)km.tile_offsets->ptr, jt->dense_min, (const uint32_t *)nullptr, fc, uint64_t(0), (int *)nullptr);
        sync(ctx);
        return out;
    }
    if (jt->direct) {
        // unique build keys: every probe row yields 0/1 rows → stream compaction with a gather
        KeepMask km;
        km.n = n;
        km.ntiles = (n + TILE_ROWS - 1) / TILE_ROWS;
        int64_t nwords = (n + 63) / 64;
        km.keep = dev_alloc(ctx, size_t(nwords) * 8 + 8);
        BufRef counts = dev_alloc(ctx, size_t(km.ntiles + 1) * 4);
        bool left_plain = jt->left_cols.size() + right->cols.size() <= size_t(MAX_JOIN_COLS);
        for (auto &c : jt->left_cols) left_plain = left_plain && is_word_type(c.dtype) && !c.validity;
        // one plain payload column riding in the table's slots: its values arrive with the lookup, per probe row
        const bool pairs = jt->slotsp != nullptr && left_plain && right_plain && !jt->dense;
        BufRef bidx, payload;
        if (pairs) {
            payload = dev_alloc(ctx, size_t(n) * 8 + 8);
            if (km.ntiles && jt->pp_bits) {
                PackedPairs pp{jt->pp_kmin, jt->pp_kspan, jt->pp_base, jt->pp_nb, jt->pp_bits};
                launch(ctx, "join_probe_pairs", probe_packed_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, rk.words(), n, km.ntiles,
                       (const ulonglong2 *)jt->slotsp->ptr, pp, (uint64_t *)km.keep->ptr, (uint64_t *)payload->ptr, (uint32_t *)counts->ptr);
            } else if (km.ntiles)
                launch(ctx, "join_probe_pairs", probe_pairs_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, rk.words(), n, km.ntiles,
                       (const ulonglong2 *)jt->slotsp->ptr, jt->cap, jt->shift, jt->filler, (uint64_t *)km.keep->ptr, (uint64_t *)payload->ptr,
                       (uint32_t *)counts->ptr);
        } else {
            bidx = dev_alloc(ctx, size_t(n) * 4 + 8);
            if (km.ntiles && !jt->dense)
                launch(ctx, "join_probe_unique", probe_unique_coop_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, rk.words(), n, km.ntiles,
                       (const ulonglong2 *)jt->slots->ptr, jt->cap, jt->shift, (uint64_t *)km.keep->ptr, (uint32_t *)bidx->ptr, (uint32_t *)counts->ptr);
            else if (km.ntiles)
                launch(ctx, "join_probe_unique", probe_unique_kernel, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, rk.words(), n,
                       km.ntiles, L, (uint64_t *)km.keep->ptr, (uint32_t *)bidx->ptr, (uint32_t *)counts->ptr);
        }
        km = finish_mask(ctx, km, counts);
        auto out = std::make_unique<nqe_table>();
        out->ctx = ctx;
        out->rows = km.total;
        const bool no_share_probe = getenv("NQE_JOIN_NO_SHARED_PROBE_COLUMNS") != nullptr; // (read per call)
        if (pairs && km.total == n && n > 0 && !no_share_probe && probe_shareable && share_key_column(jt->left_cols[size_t(jt->left_key)], rk)) {
            // every probe row matched: output row = probe row.  The payload words the lookup wrote per probe row ARE the build
            // payload column of the output, and the other three columns are the probe table's own (shared buffers): no second pass
            for (size_t ci = 0; ci < jt->left_cols.size(); ++ci) {
                DevColumn c;
                if (int(ci) == jt->left_key) {
                    c = rk;
                } else {
                    c.length = n;
                    c.values = payload;
                }
                c.dtype = jt->left_cols[ci].dtype;
                c.null_count = 0;
                out->cols.push_back(c);
            }
            for (auto &c : right->cols) out->cols.push_back(c);
            sync(ctx);
            return out;
        }
        if (left_plain && right_plain) {
            // every column is a plain 8-byte column: ONE pass writes all of them (probe columns streamed, the build key taken
            // from the probe key, build payloads gathered by the recorded build row) instead of one compaction per column
            FusedCols fc;
            std::memset(&fc, 0, sizeof(fc));
            const bool share_key = share_key_column(jt->left_cols[size_t(jt->left_key)], rk); // as in the dense path
            int share_pos = -1, right_key_pos = -1;
            for (size_t ci = 0; ci < jt->left_cols.size(); ++ci) {
                const DevColumn &c = jt->left_cols[ci];
                if (int(ci) == jt->left_key && share_key) {
                    share_pos = int(out->cols.size());
                    out->cols.push_back(DevColumn{});
                    continue;
                }
                out->cols.push_back(make_word_column(ctx, c.dtype, km.total, false));
                fc.kind[fc.n] = int(ci) == jt->left_key ? 1 : (pairs ? 0 : 2);
                fc.src[fc.n] = pairs && int(ci) != jt->left_key ? (const uint64_t *)payload->ptr : c.words();
                fc.dst[fc.n] = (uint64_t *)out->cols.back().values->ptr;
                fc.n++;
            }
            const bool share_probe = km.total == n && !no_share_probe && probe_shareable; // every probe row matched: the probe-side columns are the probe table's own
            for (size_t cj = 0; cj < right->cols.size(); ++cj) {
                const DevColumn &c = right->cols[cj];
                if (share_probe) {
                    out->cols.push_back(c);
                    out->cols.back().null_count = 0;
                    if (int(cj) == right_key) right_key_pos = int(out->cols.size()) - 1;
                    continue;
                }
                out->cols.push_back(make_word_column(ctx, c.dtype, km.total, false));
                if (int(cj) == right_key) right_key_pos = int(out->cols.size()) - 1;
                fc.kind[fc.n] = int(cj) == right_key ? 1 : 0;
                fc.src[fc.n] = c.words();
                fc.dst[fc.n] = (uint64_t *)out->cols.back().values->ptr;
                fc.n++;
            }
            if (share_pos >= 0) {
                out->cols[size_t(share_pos)] = out->cols[size_t(right_key_pos)];
                out->cols[size_t(share_pos)].dtype = jt->left_cols[size_t(jt->left_key)].dtype;
            }
            if (km.ntiles && km.total > 0)
                launch(ctx, "join_fused_write", join_fused_write_kernel<FUSED_WRITE_ROWS>, dim3(stream_grid(ctx, km.ntiles, 4)), dim3(256), 0, rk.words(), n, km.ntiles,
                       (const uint64_t *)km.keep->ptr, (const uint64_t *)km.tile_offsets->ptr, uint64_t(0), pairs ? (const uint32_t *)nullptr : (const uint32_t *)bidx->ptr, fc,
                       uint64_t(0), (int *)nullptr);
            sync(ctx);
            return out;
        }
        DevColumn outer_pos;
        for (size_t ci = 0; ci < jt->left_cols.size(); ++ci) {
            const DevColumn &c = jt->left_cols[ci];
            if (int(ci) == jt->left_key && !c.validity && c.dtype != NQE_UTF8) {
                // the build key of a matched row is bit-identical to the probe key: produce the column by the
                // coalesced compaction of the probe keys instead of a random gather (validity comes from the
                // LEFT column, which has none here)
                DevColumn as_left = rk;
                as_left.validity = nullptr;
                as_left.null_count = 0;
                out->cols.push_back(compact_column(ctx, as_left, km));
            } else if (c.dtype == NQE_UTF8) {
                // outer_pos (the reference's Int64 index array, hash_join.rs:230-231) = build rows of the matches,
                // obtained by gathering a row-number column; then the Utf8 `take`
                if (!outer_pos.values) {
                    DevColumn rowid;
                    rowid.dtype = NQE_INT64;
                    rowid.length = jt->left_rows;
                    rowid.values = iota_i64(ctx, 0, jt->left_rows);
                    outer_pos = compact_gather_column(ctx, rowid, (const uint32_t *)bidx->ptr, km);
                }
                out->cols.push_back(take_utf8(ctx, c, (const int64_t *)outer_pos.words(), km.total, false));
            } else {
                out->cols.push_back(compact_gather_column(ctx, c, (const uint32_t *)bidx->ptr, km));
            }
        }
        for (auto &c : right->cols) out->cols.push_back(compact_column(ctx, c, km));
        sync(ctx); // bidx / mask are released on return
        return out;
    }

    const int64_t ntiles = (n + JT_ROWS - 1) / JT_ROWS;
    BufRef pmeta = dev_alloc(ctx, size_t(n) * 8 + 8);
    BufRef counts = dev_alloc(ctx, size_t(ntiles + 1) * 4);
    BufRef offs = dev_alloc(ctx, size_t(ntiles + 1) * 8);
    int grid = int(std::max<int64_t>(1, std::min<int64_t>(ntiles, int64_t(ctx->num_cus) * 8)));
    if (n)
        launch(ctx, "join_probe_count", probe_count_kernel, dim3(grid), dim3(JT_BLOCK), 0, rk.words(), n, L, (uint64_t *)pmeta->ptr,
               (uint32_t *)counts->ptr, ctx->d_flags);
    exclusive_scan_u32_to_u64(ctx, (const uint32_t *)counts->ptr, (uint64_t *)offs->ptr, ntiles);
    const int64_t M = int64_t(read_scalar(ctx, (const uint64_t *)offs->ptr + ntiles));
    {
        int f[NQE_NUM_FLAGS];
        flags_read(ctx, f);
        if (f[NQE_FLAG_TABLE_FULL]) fail(NQE_ERR_OUT_OF_MEMORY, "join output of one probe tile exceeds 2^32 rows");
    }

    auto out = std::make_unique<nqe_table>();
    out->ctx = ctx;
    out->rows = M;
    JoinCols jc;
    std::memset(&jc, 0, sizeof(jc));
    // Utf8 payload: the kernel emits outer_pos / inner_pos (gathers of row-number columns), then Utf8 `take`
    DevColumn left_rowid, right_rowid, outer_pos, inner_pos;
    std::vector<const DevColumn *> srcs;
    std::vector<int> out_slot; // output column index of each kernel column, -1 outer_pos, -2 inner_pos
    // the build key of a match is bit-identical to the probe key: emit that column from the probe side (coalesced)
    DevColumn key_from_probe;
    const bool key_shortcut = jt->key_dtype != NQE_UTF8 && !jt->left_cols[size_t(jt->left_key)].validity;
    for (size_t c = 0; c < jt->left_cols.size(); ++c)
        if (jt->left_cols[c].dtype != NQE_UTF8 && !(key_shortcut && int(c) == jt->left_key)) { srcs.push_back(&jt->left_cols[c]); out_slot.push_back(int(c)); }
    if (utf8_left) {
        left_rowid.dtype = NQE_INT64; left_rowid.length = jt->left_rows; left_rowid.values = iota_i64(ctx, 0, jt->left_rows);
        srcs.push_back(&left_rowid); out_slot.push_back(-1);
    }
    jc.n_left = int(srcs.size());
    // … and when equal keys are identical bits the two key columns of the output are ONE buffer (as in the unique-key forms): a
    // column less to write
    const bool key_shared = key_shortcut && share_key_column(jt->left_cols[size_t(jt->left_key)], rk_orig);
    if (key_shortcut && !key_shared) {
        key_from_probe = rk;
        key_from_probe.validity = nullptr;
        key_from_probe.null_count = 0;
        srcs.push_back(&key_from_probe); out_slot.push_back(jt->left_key);
    }
    for (size_t c = 0; c < right->cols.size(); ++c)
        if (right->cols[c].dtype != NQE_UTF8) { srcs.push_back(&right->cols[c]); out_slot.push_back(int(jt->left_cols.size() + c)); }
    if (utf8_right) {
        right_rowid.dtype = NQE_INT64; right_rowid.length = n; right_rowid.values = iota_i64(ctx, 0, n);
        srcs.push_back(&right_rowid); out_slot.push_back(-2);
    }
    jc.n = int(srcs.size());
    if (jc.n > MAX_JOIN_COLS) fail(NQE_ERR_NOT_SUPPORTED, "join output wider than 32 columns");
    // left columns with a copy in sorted-row order are read at start + match number: the matches of a probe row are adjacent words
    std::vector<DevColumn> by_pos_cols(size_t(jc.n_left));
    for (int k = 0; k < jc.n_left; ++k) {
        const int ci = out_slot[size_t(k)];
        if (!jt->direct && ci >= 0 && size_t(ci) < jt->sorted_cols.size() && jt->sorted_cols[size_t(ci)]) {
            by_pos_cols[size_t(k)] = *srcs[size_t(k)];
            by_pos_cols[size_t(k)].values = jt->sorted_cols[size_t(ci)];
            srcs[size_t(k)] = &by_pos_cols[size_t(k)];
            jc.by_pos[k] = 1;
        } else
            jc.need_perm = 1;
    }
    out->cols.resize(ncols);
    std::vector<BufRef> bool_bytes(srcs.size()), valid_bytes(srcs.size());
    std::vector<DevColumn> dsts(srcs.size());
    for (size_t k = 0; k < srcs.size(); ++k) {
        const DevColumn &src = *srcs[k];
        const bool v = src.validity != nullptr;
        DevColumn dst = src.dtype == NQE_BOOLEAN ? make_bool_column(ctx, M, v) : make_word_column(ctx, src.dtype, M, v);
        jc.src[k] = src.values ? src.values->ptr : nullptr;
        jc.src_valid[k] = src.valid();
        jc.dtype[k] = src.dtype;
        if (src.dtype == NQE_BOOLEAN) {
            bool_bytes[k] = dev_alloc(ctx, size_t(M) + 8);
            jc.dst_bool_bytes[k] = (uint8_t *)bool_bytes[k]->ptr;
        } else {
            jc.dst_words[k] = (uint64_t *)dst.values->ptr;
        }
        if (v) {
            valid_bytes[k] = dev_alloc(ctx, size_t(M) + 8);
            jc.dst_valid_bytes[k] = (uint8_t *)valid_bytes[k]->ptr;
        }
        dsts[k] = std::move(dst);
    }
    bool all_plain = true;
    for (size_t k = 0; k < srcs.size(); ++k)
        all_plain = all_plain && is_word_type(srcs[k]->dtype) && !srcs[k]->validity && jc.dst_words[k] && !jc.dst_bool_bytes[k] && !jc.dst_valid_bytes[k] && srcs[k]->length > 0;
    if (n && M)
        launch(ctx, "join_probe_write", all_plain ? probe_write_kernel<true> : probe_write_kernel<false>, dim3(grid), dim3(JT_BLOCK), 0, (const uint64_t *)pmeta->ptr, n,
               (const uint64_t *)offs->ptr, (const uint32_t *)jt->perm->ptr, jt->direct ? 1 : 0, jc);
    for (size_t k = 0; k < srcs.size(); ++k) {
        if (bool_bytes[k]) pack_bytes_to_bits(ctx, (const uint8_t *)bool_bytes[k]->ptr, M, (uint64_t *)dsts[k].values->ptr);
        if (valid_bytes[k]) pack_bytes_to_bits(ctx, (const uint8_t *)valid_bytes[k]->ptr, M, (uint64_t *)dsts[k].validity->ptr);
        if (out_slot[k] >= 0) out->cols[size_t(out_slot[k])] = dsts[k];
        else if (out_slot[k] == -1) outer_pos = dsts[k];
        else inner_pos = dsts[k];
    }
    if (key_shared) out->cols[size_t(jt->left_key)] = out->cols[jt->left_cols.size() + size_t(right_key)];
    for (size_t c = 0; c < jt->left_cols.size(); ++c)
        if (jt->left_cols[c].dtype == NQE_UTF8)
            out->cols[c] = take_utf8(ctx, jt->left_cols[c], (const int64_t *)outer_pos.words(), M, false);
    for (size_t c = 0; c < right->cols.size(); ++c)
        if (right->cols[c].dtype == NQE_UTF8)
            out->cols[jt->left_cols.size() + c] = take_utf8(ctx, right->cols[c], (const int64_t *)inner_pos.words(), M, false);
    sync(ctx); // temporaries above are released on return; keep the stream drained for simplicity
    return out;
}

} // namespace

} // namespace nqe

using namespace nqe;

extern "C" {

nqe_status nqe_hash_join_build(nqe_ctx *ctx, const nqe_table *left, int32_t left_key, nqe_join_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !left || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    flags_reset(ctx);
    *out = build_table(ctx, left, left_key).release();
    NQE_API_END()
}

nqe_status nqe_hash_join_probe(nqe_ctx *ctx, const nqe_join_table *build, const nqe_table *right, int32_t right_key,
                               nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !build || !right || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    flags_reset(ctx);
    *out = probe_table(ctx, build, right, right_key).release();
    NQE_API_END()
}

nqe_status nqe_join_table_release(nqe_join_table *jt) {
    delete jt;
    return NQE_OK;
}

nqe_status nqe_hash_join_execute(nqe_ctx *ctx, const nqe_table *left, const nqe_table *right, int32_t left_key,
                                 int32_t right_key, nqe_table **out) {
    NQE_API_BEGIN(ctx)
    if (!ctx || !left || !right || !out) fail(NQE_ERR_INVALID_ARGUMENT, "bad arguments");
    if (left_key < 0 || right_key < 0) // empty `on` (hash_join.rs:125-129)
        fail(NQE_ERR_PLAN, "Inner Join on Conditions can't not be empty");
    flags_reset(ctx);
    std::unique_ptr<nqe_join_table> jt = build_table(ctx, left, left_key);
    *out = probe_table(ctx, jt.get(), right, right_key).release();
    NQE_API_END()
}

} // extern "C"

// this translation unit's code object is loaded when a context is created, not by the first query that needs it (context.hip: load_modules)
NQE_MODULE_PROBE(nqe::iota_u32_kernel);
