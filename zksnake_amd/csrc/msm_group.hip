// msm_group.hip -- instantiates the MSM kernels and plan for ONE curve group (ZK_GROUP = Bn254G1, ...).
// Compiled four times by the Makefile; field multiplications are inlined here (29-bit limb products are
// ~230 instructions), which is why the instantiations are kept in separate translation units.
#include "msm_impl.cuh"

#ifndef ZK_GROUP
#error "compile with -DZK_GROUP=<Bn254G1|Bn254G2|Bls381G1|Bls381G2>"
#endif
#define ZK_CAT2(a, b) a##b
#define ZK_CAT(a, b) ZK_CAT2(a, b)

namespace zkmi {

int ZK_CAT(msm_plan_create_, ZK_GROUP)(uint64_t n, const void* bases, int on_device, int flags, int window_bits,
                                       MsmPlanBase** out) {
    MsmPlan<ZK_GROUP>* p = new MsmPlan<ZK_GROUP>();
    int rc = p->init(n, bases, on_device, flags, window_bits);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return ZK_OK;
}

int ZK_CAT(msm_batch_mul_, ZK_GROUP)(uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast,
                                     uint64_t* out) {
    return batch_mul_impl<ZK_GROUP>(n, scalars, bases, broadcast, out);
}

}  // namespace zkmi
