// aggregate_fast_null.hip — instantiations of the fast aggregate kernel that read validity bitmaps (VNULL = true).
#include "aggregate_fast_kernel.hpp"

namespace nqe {
namespace agg {

FastKernel pick_fast_kernel_nullable(int pred, int key, int nv, bool vf64) { return pick_fast_pred<true>(pred, key, nv, vf64); }

} // namespace agg
} // namespace nqe
