// aggregate_fast_inst.hip — one slice of the fast aggregate kernel's instantiations: predicate variant NQE_FAST_PRED (0..3) with
// (NQE_FAST_VNULL = 1) or without validity bitmaps, all key / value-count / value-type variants of it.  The Makefile compiles
// this file eight times (aggregate_fast_p<P>_v<V>.o): the 128 instances in two translation units were the critical path of the
// build (2 min 10 s each), eight slices compile in parallel.
#include "aggregate_fast_kernel.hpp"

#ifndef NQE_FAST_PRED
#define NQE_FAST_PRED 0
#endif
#ifndef NQE_FAST_VNULL
#define NQE_FAST_VNULL 0
#endif
#define NQE_FAST_CAT2(p, v) pick_fast_p##p##_v##v
#define NQE_FAST_CAT(p, v) NQE_FAST_CAT2(p, v)

namespace nqe {
namespace agg {

FastKernel NQE_FAST_CAT(NQE_FAST_PRED, NQE_FAST_VNULL)(int key, int nv, bool vf64, bool sub, bool nomm, bool share) {
    return pick_fast_key<NQE_FAST_PRED, NQE_FAST_VNULL != 0>(key, nv, vf64, sub, nomm, share);
}

} // namespace agg
} // namespace nqe

// this slice's code object is loaded when a context is created, not by the first query that picks one of its instances
// (context.hip: load_modules): any instance names the module
static const bool nqe_module_probe_ = (nqe::register_module_probe(reinterpret_cast<const void *>(
                                           nqe::agg::pick_fast_key<NQE_FAST_PRED, NQE_FAST_VNULL != 0>(0, 1, true, false, false, false))), true);
