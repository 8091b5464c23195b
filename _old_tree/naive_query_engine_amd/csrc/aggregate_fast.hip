// aggregate_fast.hip — instantiations of the fast aggregate kernel for inputs without validity bitmaps, and the entry point
// the host logic (aggregate.hip) uses to pick a variant.
#include "aggregate_fast_kernel.hpp"

namespace nqe {
namespace agg {

FastKernel pick_fast_kernel_nullable(int pred, int key, int nv, bool vf64); // aggregate_fast_null.hip

FastKernel pick_fast_kernel(int pred, int key, int nv, bool vf64, bool vnull) {
    return vnull ? pick_fast_kernel_nullable(pred, key, nv, vf64) : pick_fast_pred<false>(pred, key, nv, vf64);
}

} // namespace agg
} // namespace nqe
