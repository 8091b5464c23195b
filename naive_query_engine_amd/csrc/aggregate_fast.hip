// aggregate_fast.hip — the entry point the host logic (aggregate.hip) uses to pick a variant of the fast aggregate kernel; the
// instances live in nine slices (aggregate_fast_inst.hip compiled per predicate variant, with and without validity bitmaps).
#include "aggregate_fast_kernel.hpp"

namespace nqe {
namespace agg {

#define NQE_FAST_DECL(p)                                   \
    FastKernel pick_fast_p##p##_v0(int key, int nv, bool vf64, bool sub, bool nomm, bool share); \
    FastKernel pick_fast_p##p##_v1(int key, int nv, bool vf64, bool sub, bool nomm, bool share);
NQE_FAST_DECL(0)
NQE_FAST_DECL(1)
NQE_FAST_DECL(2)
NQE_FAST_DECL(3)
#undef NQE_FAST_DECL
FastKernel pick_fast_p4_v0(int key, int nv, bool vf64, bool sub, bool nomm, bool share);
FastKernel pick_fast_p5_v0(int key, int nv, bool vf64, bool sub, bool nomm, bool share);
FastKernel pick_fast_p6_v0(int key, int nv, bool vf64, bool sub, bool nomm, bool share);

FastKernel pick_fast_kernel(int pred, int key, int nv, bool vf64, bool vnull, bool sub, bool nomm, bool share) {
    switch (pred) {
    case 0: return vnull ? pick_fast_p0_v1(key, nv, vf64, sub, nomm, share) : pick_fast_p0_v0(key, nv, vf64, sub, nomm, share);
    case 1: return vnull ? pick_fast_p1_v1(key, nv, vf64, sub, nomm, share) : pick_fast_p1_v0(key, nv, vf64, sub, nomm, share);
    case 2: return vnull ? pick_fast_p2_v1(key, nv, vf64, sub, nomm, share) : pick_fast_p2_v0(key, nv, vf64, sub, nomm, share);
    case 3: return vnull ? pick_fast_p3_v1(key, nv, vf64, sub, nomm, share) : pick_fast_p3_v0(key, nv, vf64, sub, nomm, share);
    case 4: return vnull ? nullptr : pick_fast_p4_v0(key, nv, vf64, sub, nomm, share);
    case 5: return vnull ? nullptr : pick_fast_p5_v0(key, nv, vf64, sub, nomm, share);
    default: return vnull ? nullptr : pick_fast_p6_v0(key, nv, vf64, sub, nomm, share);
    }
}

} // namespace agg
} // namespace nqe
