// fp2_split.hip.h -- the G2 bucket-accumulation step with every Fp2 value held by a LANE PAIR: the even lane keeps c0, the
// odd lane c1 (round 4).
//
// Why: over Fp2 the mixed addition of accumulate_kernel needs 256 registers (BN254, seven of them spilled) or 401 (BLS12-381) per
// lane, i.e. two waves or ONE wave per SIMD, and a lone wave issues a multiply-add every 7.4 cycles where four waves issue one
// every 5.1 (tools/ubench2.hip).  Split by component, a lane holds half of every value.  The arithmetic is the SAME: a
// component of an Fp2 product is one double product,
//        c0 = a0 b0 + (K p - a1) b1          c1 = a0 b1 + a1 b0
// so both lanes run one instruction stream -- fp_mul2 on operands selected by the lane's parity -- and every product, every
// carry pass and every range bound of the single-lane step (curve.hip.h: xyzz_add_affine_relaxed2, replayed on integers by
// tools/model_relaxed_g2.py) carries over literally.  What is new is data movement: the other component of an operand comes
// through a DPP quad permute (lane ^ 1), 11 coordinate-sized exchanges per addition (~1.5 % of its instructions).
// Control flow is uniform within a pair: every flag that steers it is exchanged first.
//
// Replaces nothing in the reference by itself (ark's MSM adds into its buckets on one core, src/bn254/curve.rs:375-392 ->
// ark-ec VariableBaseMSM); it is how the same additions are laid out for 64-wide waves with a 512-register file per SIMD lane.
#pragma once
#include "pair.hip.h"

namespace zkmi {

// one component (this lane's) of each coordinate of an XYZZ point over Fp2
template <class P>
struct SplitXYZZ {
    Fp<P> X, Y, ZZ, ZZZ;
};

template <class P>
__device__ __forceinline__ Fp<P> sp_sel(bool odd, const Fp<P>& even_v, const Fp<P>& odd_v) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = odd ? odd_v.v[i] : even_v.v[i];
    return r;
}

// this lane's component of a * b, a.c1 negated lazily against K p (a.c1 < (K - 1) p): fp2_mul_relaxed<P, K> by component
template <class P, int K>
__device__ __forceinline__ Fp<P> sp_mul(const Fp<P>& a, const Fp<P>& b, bool odd) {
    const Fp<P> xa = pair_xch<Fp<P>>(a), xb = pair_xch<Fp<P>>(b);
    const Fp<P> nxa = fp_neg_lazy_k<P, K>(xa);   // meaningful in the even lane: K p - a1
    // even: a0 b0 + (K p - a1) b1        odd: a0 b1 + a1 b0
    return fp_mul2<P>(sp_sel<P>(odd, a, xa), b, sp_sel<P>(odd, nxa, a), xb);
}

// this lane's component of a^2 as ((a0 + a1)(a0 - a1 + KD p), (2 a0) a1): fp2_sqr_relaxed<P, KD> by component
template <class P, int KD>
__device__ __forceinline__ Fp<P> sp_sqr(const Fp<P>& a, bool odd) {
    const Fp<P> xa = pair_xch<Fp<P>>(a);
    // even lane: a = a0, xa = a1; odd lane: a = a1, xa = a0
    const Fp<P> ts = fp_add_nosel<P>(a, xa);
    const Fp<P> td = fp_sub_k<P, KD>(a, xa);
    return fp_mul<P>(sp_sel<P>(odd, ts, fp_dbl_lazy<P>(xa)), sp_sel<P>(odd, td, a));
}

// plain-range square (fp2_sqr) for the rare doubling: ((a0 + a1)(a0 - a1), 2 (a0 a1)), inputs and result below 2p
template <class P>
__device__ __forceinline__ Fp<P> sp_sqr_plain(const Fp<P>& a, bool odd) {
    const Fp<P> xa = pair_xch<Fp<P>>(a);
    const Fp<P> prod = fp_mul<P>(sp_sel<P>(odd, fp_add<P>(a, xa), xa), sp_sel<P>(odd, fp_sub<P>(a, xa), a));
    return sp_sel<P>(odd, prod, fp_dbl<P>(prod));
}

// both components zero mod p?  (every lane of the pair gets the same answer)
template <class P>
__device__ __forceinline__ bool sp_is_zero(const Fp<P>& a) {
    const bool z = fp_is_zero<P>(a);
    return pair_xch_flag(z) && z;
}
// both components exact zeros (the sentinels: an all-zero base row, an empty accumulator)
template <class P>
__device__ __forceinline__ bool sp_is_zero_limbs(const Fp<P>& a) {
    const bool z = fp_is_zero_limbs<P>(a);
    return pair_xch_flag(z) && z;
}

template <class P>
__device__ __forceinline__ SplitXYZZ<P> sp_inf() {
    return {fp_zero<P>(), fp_zero<P>(), fp_zero<P>(), fp_zero<P>()};
}

// 2 q for an affine q (mdbl-2008-s-1, a = 0), plain ranges: the rare "same point twice in one bucket" case
template <class P>
__device__ __forceinline__ SplitXYZZ<P> sp_dbl_affine(const Fp<P>& x, const Fp<P>& y, bool odd) {
    if (sp_is_zero<P>(y)) return sp_inf<P>();   // a point of order two: not in these groups, kept for symmetry with xyzz_dbl_affine
    const Fp<P> U = fp_dbl<P>(y);
    const Fp<P> V = sp_sqr_plain<P>(U, odd);
    const Fp<P> W = sp_mul<P, 4>(U, V, odd);
    const Fp<P> S = sp_mul<P, 4>(x, V, odd);
    const Fp<P> xx = sp_sqr_plain<P>(x, odd);
    const Fp<P> M = fp_add<P>(fp_dbl<P>(xx), xx);
    const Fp<P> X3 = fp_sub<P>(sp_sqr_plain<P>(M, odd), fp_dbl<P>(S));
    const Fp<P> Y3 = fp_sub<P>(sp_mul<P, 4>(M, fp_sub<P>(S, X3), odd), sp_mul<P, 4>(W, y, odd));
    return {X3, Y3, V, W};
}

// acc + (+-q), q affine: xyzz_add_affine_relaxed2 component by component (same ranges: X in [0, 4p), Y, ZZ, ZZZ and the base
// in [0, 2p]).  Both lanes of a pair call it together.
template <class P>
__device__ __forceinline__ void sp_add_affine(SplitXYZZ<P>& acc, const Fp<P>& qx, const Fp<P>& qy, bool negate, bool odd) {
    // the sentinel row and an empty accumulator are exact zeros (written as such)
    {
        const bool qz = fp_is_zero_limbs<P>(qx) && fp_is_zero_limbs<P>(qy);
        if (pair_xch_flag(qz) && qz) return;
    }
    if (sp_is_zero_limbs<P>(acc.ZZ)) {
        const Fp<P> one = sp_sel<P>(odd, fp_one<P>(), fp_zero<P>());   // (1, 0)
        acc = {qx, negate ? fp_neg<P>(qy) : qy, one, one};
        return;
    }
    const Fp<P> U2 = sp_mul<P, 4>(qx, acc.ZZ, odd);
    // -y as 2p - y (normalised, <= 2p): a product operand like any other
    Fp<P> ny = fp_sub_k<P, 2>(fp_zero<P>(), qy);
#pragma unroll
    for (int i = 0; i < P::N; ++i) ny.v[i] = negate ? ny.v[i] : qy.v[i];
    const Fp<P> S2 = sp_mul<P, 4>(ny, acc.ZZZ, odd);
    const Fp<P> Pd = fp_sub_k<P, 4>(U2, acc.X);     // components below 6p
    const Fp<P> R = fp_sub_k<P, 2>(S2, acc.Y);      // below 4p
    const Fp<P> PP = sp_sqr<P, 8>(Pd, odd);
    const Fp<P> RR = sp_sqr<P, 4>(R, odd);
    if (sp_is_zero<P>(PP)) {   // P^2 = 0 in the field: same x
        if (sp_is_zero<P>(RR)) acc = sp_dbl_affine<P>(qx, negate ? fp_neg<P>(qy) : qy, odd);
        else acc = sp_inf<P>();
        return;
    }
    const Fp<P> PPP = sp_mul<P, 8>(Pd, PP, odd);
    const Fp<P> Q = sp_mul<P, 8>(acc.X, PP, odd);
    const Fp<P> X3 = fp_sub_twice_sel4<P>(fp_sub_k<P, 2>(RR, PPP), Q);
    const Fp<P> D = fp_sub_k<P, 4>(Q, X3);           // below 6p
    if constexpr (P::N <= 9) {
        // Y3 = R D - Y PPP as ONE four-product reduction per component:
        //   c0 = R0 D0 + (8p - R1) D1 + (4p - Y0) P0 + Y1 P1          c1 = R0 D1 + R1 D0 + (4p - Y0) P1 + (4p - Y1) P0
        const Fp<P> xR = pair_xch<Fp<P>>(R), xD = pair_xch<Fp<P>>(D), xY = pair_xch<Fp<P>>(acc.Y), xP = pair_xch<Fp<P>>(PPP);
        const Fp<P> nY = fp_neg_lazy_k<P, 4>(acc.Y), nxY = fp_neg_lazy_k<P, 4>(xY), nxR = fp_neg_lazy_k<P, 8>(xR);
        //                 a                      b   c                       d    e                      f    g                     h
        acc.Y = fp_mul4<P>(sp_sel<P>(odd, R, xR), D, sp_sel<P>(odd, nxR, R), xD, sp_sel<P>(odd, nY, nxY), PPP, sp_sel<P>(odd, xY, nY), xP);
    } else {
        acc.Y = fp_sub<P>(sp_mul<P, 8>(R, D, odd), sp_mul<P, 4>(acc.Y, PPP, odd));
    }
    acc.X = X3;
    acc.ZZ = sp_mul<P, 4>(acc.ZZ, PP, odd);
    acc.ZZZ = sp_mul<P, 4>(acc.ZZZ, PPP, odd);
}

}  // namespace zkmi
