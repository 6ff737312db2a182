// msm_common.hip.h -- device load / store of points, register-form copies and the endomorphism traits shared by the MSM
// kernel headers (msm_sort.hip.h, msm_accumulate.hip.h, msm_reduce.hip.h) and the plan (msm_impl.hip.h).
#pragma once
#include "common.hip.h"
#include "msm_plan.h"
#include "pair.hip.h"
#include "glv_params.h"

namespace zkmi {

// ---- device load/store of field elements / points (packed u32 words, 16-byte vectors) ------

template <int WORDS>
__device__ __forceinline__ void load_words(uint32_t* dst, const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) {
        uint4 t = q[i];
        dst[4 * i] = t.x; dst[4 * i + 1] = t.y; dst[4 * i + 2] = t.z; dst[4 * i + 3] = t.w;
    }
}
template <int WORDS>
__device__ __forceinline__ void store_words(uint32_t* dst, const uint32_t* src) {
    uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) q[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}

// Points in memory are packed 32-bit words (LIMBS per coordinate); registers hold 29-bit limbs.
template <class F>
__device__ __forceinline__ Affine<F> load_affine(const uint32_t* p) {
    uint32_t w[2 * F::LIMBS];
    load_words<2 * F::LIMBS>(w, p);
    return {F::load(w), F::load(w + F::LIMBS)};
}
template <class F>
__device__ __forceinline__ void store_affine(uint32_t* p, const Affine<F>& a) {
    uint32_t w[2 * F::LIMBS];
    F::store(w, a.x);
    F::store(w + F::LIMBS, a.y);
    store_words<2 * F::LIMBS>(p, w);
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const uint32_t* p) {
    uint32_t w[4 * F::LIMBS];
    load_words<4 * F::LIMBS>(w, p);
    return {F::load(w), F::load(w + F::LIMBS), F::load(w + 2 * F::LIMBS), F::load(w + 3 * F::LIMBS)};
}
template <class F>
__device__ __forceinline__ void store_xyzz(uint32_t* p, const XYZZ<F>& a) {
    uint32_t w[4 * F::LIMBS];
    F::store(w, a.X);
    F::store(w + F::LIMBS, a.Y);
    F::store(w + 2 * F::LIMBS, a.ZZ);
    F::store(w + 3 * F::LIMBS, a.ZZZ);
    store_words<4 * F::LIMBS>(p, w);
}
template <class F>
static XYZZ<F> load_xyzz_host(const uint32_t* w) {
    return {F::load(w), F::load(w + F::LIMBS), F::load(w + 2 * F::LIMBS), F::load(w + 3 * F::LIMBS)};
}

// register-form copies (LDS staging, wave shuffles): XYZZ<F> is a plain struct of u32 registers
template <class F>
struct XyzzRegs { static constexpr int COUNT = sizeof(XYZZ<F>) / 4; };

template <class F>
__device__ __forceinline__ XYZZ<F> shfl_xor_xyzz(const XYZZ<F>& a, int mask) {
    XYZZ<F> r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) d[i] = __shfl_xor(s[i], mask, 64);
    return r;
}
template <class F>
__device__ __forceinline__ void lds_put_xyzz(uint32_t* slot, const XYZZ<F>& a) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) slot[i] = s[i];
}
template <class F>
__device__ __forceinline__ XYZZ<F> lds_get_xyzz(const uint32_t* slot) {
    XYZZ<F> r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) d[i] = slot[i];
    return r;
}

// Buckets with more than COMBINE_SMALL_MAX runs (skewed scalars: e.g. the short top window, or many equal
// scalars) are listed in big_list and reduced by a whole workgroup each instead of one lane.
constexpr uint32_t COMBINE_SMALL_MAX = 16;    // <= 16 runs: one lane adds them up
constexpr uint32_t COMBINE_WAVE_MAX = 2048;   // <= 2048 runs: one wave per bucket; above: one workgroup


// ---- G1 endomorphism (GLV) ---------------------------------------------------------------------------
// General (not fixed-base) G1 plans run the MSM over 2n points (P_i, phi(P_i)) with the two ~127-bit halves of every
// scalar, k = k1 + lambda k2: the same number of bucket additions (2n entries in half the windows), but half the bucket
// sets to reduce and half the doublings in the host tail.  Constants and the decomposition: tools/gen_glv_params.py.
template <class G> struct GlvOf { static constexpr bool OK = false; };
template <> struct GlvOf<Bn254G1> { static constexpr bool OK = true; typedef Bn254Glv P; };
template <> struct GlvOf<Bls381G1> { static constexpr bool OK = true; typedef Bls381Glv P; };
// G2: psi^2 (x, y) = (c x, -y) with c in Fp (tools/gen_glv_params.py); its eigenvalue differs from G1's, so a split-scalar
// G1 plan and a split-scalar G2 plan cannot share digits or a sort (SortExport::endo)
template <> struct GlvOf<Bn254G2> { static constexpr bool OK = true; typedef Bn254G2Glv P; };
template <> struct GlvOf<Bls381G2> { static constexpr bool OK = true; typedef Bls381G2Glv P; };

}  // namespace zkmi
