// aggregate_tail.hpp — what aggregate.hip (the operator's host side and its streaming kernels) needs of aggregate_tail.hip: the kernels
// behind the streaming pass — table set-up, collect / finalize, the dense and range-tier tails, the ranked single-launch tail, the
// merges of exchanged states, the key range / key sample passes, the folds of per-workgroup tables — their launch constants and the
// argument block of the finalizers.
#pragma once
#include "aggregate_common.hpp"

namespace nqe {
namespace agg {

struct FinalizeArgs {
    int32_t naggs;
    int32_t partial; // 1: emit raw state (4 columns per DISTINCT value column = table slot: count, sum, min, max)
    int32_t nslots;  // distinct value columns (partial mode)
    int32_t func[16];
    int32_t vslot[16];
    uint64_t *out[64];
};

constexpr int DR_BLOCK = 256;
constexpr int RE_BLOCK = 1024;
constexpr uint32_t RANK_MAX_CAP = 8192;
constexpr int RANK_SLOTS = 64;
constexpr int RANK_WAVES = 16;
constexpr int RANK_PASSES = (RANK_MAX_CAP + 1 + RANK_WAVES * 64 - 1) / (RANK_WAVES * 64);
constexpr int KEY_SAMPLE = 1 << 16, KEY_SAMPLE_SLOTS_LOG2 = 18;
constexpr int64_t KEY_SAMPLE_MIN_ROWS = int64_t(1) << 18;

__global__ void table_init_kernel(GroupTable g, int mark_slot0);
__global__ void __launch_bounds__(256) collect_kernel(GroupTable g, uint64_t *out_keys, uint32_t *out_slots, uint32_t *counter);
__global__ void finalize_kernel(GroupTable g, const uint32_t *sorted_slots, int64_t G, FinalizeArgs f);
__global__ void __launch_bounds__(256) dense_key_range_kernel(const uint64_t *__restrict__ keys, uint32_t cap, uint64_t flip, uint32_t *dense);
__global__ void dense_rank_mark_kernel(const uint64_t *__restrict__ keys, uint32_t G, uint64_t flip, uint64_t ordmin, uint32_t *pos);
template <int DR_ITEMS>
__global__ void __launch_bounds__(DR_BLOCK) dense_rank_emit_kernel(GroupTable g, const uint32_t *__restrict__ pos, uint64_t span, uint64_t flip, uint64_t ordmin, unsigned long long *status, uint64_t *out_keys, FinalizeArgs f);
template <int ITEMS>
__global__ void __launch_bounds__(RE_BLOCK) agg_range_emit_kernel(const RangeRec *__restrict__ tab, int parts_log2, int Q, uint32_t W, uint64_t span, uint64_t key_min, unsigned long long *status, uint64_t *out_keys, FinalizeArgs f, unsigned long long *total);
__global__ void __launch_bounds__(RANK_WAVES * 64) rank_finalize_kernel(GroupTable g, int signed_order, FinalizeArgs f, uint64_t *out_keys, const int *flags, int *mirror);
__global__ void merge_states_kernel(GroupTable g, const uint64_t *keys, int64_t n, int naggs, const uint64_t *const *state_cols, int *flags);
__global__ void merge_packed_kernel(GroupTable g, const uint64_t *src, int nparts, int64_t stride, int nk, int naggs, int *flags);
__global__ void iota_slots_kernel(uint32_t *out, int64_t n);
__global__ void __launch_bounds__(256) key_range_kernel(const uint64_t *keys, int64_t n, uint64_t flip, unsigned long long *out);
__global__ void __launch_bounds__(256) key_sample_kernel(const uint64_t *keys, int64_t n, SimpleExpr ke, uint64_t flip, unsigned long long *set, unsigned long long *out);
__global__ void __launch_bounds__(256) agg_merge_partials_kernel(const double *psum, const double *pmn, const double *pmx, const uint32_t *pcnt, int grid, uint32_t span, int64_t bias, GroupTable g, int v, int *flags);
__global__ void __launch_bounds__(256) agg_fold_partials_kernel(const double *psum, const double *pmn, const double *pmx, const uint32_t *pcnt, int grid, uint32_t span, int64_t bias, int need_minmax, GroupTable g, int v, int *flags, int sub_log2, RangeRec *tab);
__global__ void store_tree_kernel(TreePred p, TreeInstr *dst);

} // namespace agg
} // namespace nqe
