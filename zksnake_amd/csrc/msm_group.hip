// msm_group.hip -- instantiates the MSM kernels and plan for ONE curve group (ZK_GROUP = Bn254G1, ...).
// Compiled by the Makefile once per group and per part, so that the heavy kernels (field products inlined:
// a BLS12-381 G2 mixed addition is ~14k instructions) build in parallel:
//   ZK_PART 0: the plan (host code) and the small kernels; the heavy kernels are only declared (extern template)
//   ZK_PART 1: accumulate / bases_to_mont kernels
//   ZK_PART 2: combine (3 tiers) / strided_sum / weighted_sum kernels
//   ZK_PART 3: setup-side kernels (batched normalisation, fixed-base table rows, batch scalar multiplication,
//              batched point (de)compression)
#include "msm_impl.hip.h"

#ifndef ZK_GROUP
#error "compile with -DZK_GROUP=<Bn254G1|Bn254G2|Bls381G1|Bls381G2> -DZK_PART=<0|1|2>"
#endif
#ifndef ZK_PART
#define ZK_PART 0
#endif
#define ZK_CAT2(a, b) a##b
#define ZK_CAT(a, b) ZK_CAT2(a, b)

namespace zkmi {

// the pair-split accumulate kernel exists for the Fp2 groups only (msm_accumulate.hip.h: AccumulateSplit)
#define ZK_SPLIT_Bn254G1 0
#define ZK_SPLIT_Bn254G2 1
#define ZK_SPLIT_Bls381G1 0
#define ZK_SPLIT_Bls381G2 1

#if ZK_PART == 1
template __global__ void accumulate_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*);
#if ZK_CAT(ZK_SPLIT_, ZK_GROUP)
template __global__ void accumulate_split_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*);
#endif
template __global__ void bases_to_mont_kernel<ZK_GROUP>(const uint32_t*, uint64_t, uint32_t*, int);
#elif ZK_PART == 2
template __global__ void combine_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*);
template __global__ void strided_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t*, SumJob, SumJob, uint32_t);
template __global__ void weighted_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t*);
#elif ZK_PART == 3
ZK_SETUP_INSTANTIATE(ZK_GROUP)
ZK_CODEC_INSTANTIATE(ZK_GROUP)
#else

int ZK_CAT(msm_plan_create_, ZK_GROUP)(uint64_t n, const void* bases, int on_device, int flags, int window_bits,
                                       int window_first, int window_count, MsmPlanBase** out) {
    MsmPlan<ZK_GROUP>* p = new MsmPlan<ZK_GROUP>();
    int rc = p->init(n, bases, on_device, flags, window_bits, window_first, window_count);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return ZK_OK;
}

void ZK_CAT(msm_fixed_table_free_, ZK_GROUP)() { FixedTable<ZK_GROUP>::get().release(); }

int ZK_CAT(msm_batch_mul_, ZK_GROUP)(uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast,
                                     uint64_t* out) {
    return batch_mul_impl<ZK_GROUP>(n, scalars, bases, broadcast, out);
}

int ZK_CAT(msm_points_codec_, ZK_GROUP)(uint64_t n, const void* in, void* out, int to_bytes, uint64_t* bad_index) {
    return codec_impl<ZK_GROUP>(n, in, out, to_bytes, bad_index);
}

#endif

}  // namespace zkmi
