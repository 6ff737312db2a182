// curve.hip.h -- short-Weierstrass (a = 0) group arithmetic for G1 (over Fp) and G2 (over Fp2) of
// BN254 and BLS12-381, generic over the field facade F (FpOps / Fp2Ops from field.hip.h).
//
// Replaces what the reference gets from ark-ec 0.4.2 `short_weierstrass::{Affine, Projective}`
// behind PointG1/PointG2 (src/bn254/curve.rs:19-324, src/bls12_381/curve.rs twins) and inside
// VariableBaseMSM (src/bn254/curve.rs:356-392).  ark stores Jacobian coordinates; the MSM
// buckets here use extended-Jacobian XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) because
// the mixed addition costs 8M + 2S with no field inversion -- every result is converted to
// the unique affine representative before it leaves the library, so outputs are
// representation-independent and bit-exact against any correct implementation.
#pragma once
#include "field.hip.h"

namespace zkmi {

template <class F>
struct Affine {
    typename F::T x, y;  // (0, 0) encodes the point at infinity (never on a curve with b != 0)
};

template <class F>
struct XYZZ {
    typename F::T X, Y, ZZ, ZZZ;  // ZZ == 0 <=> infinity
};

template <class F>
ZK_HD bool aff_is_inf(const Affine<F>& p) {
    return F::is_zero(p.x) && F::is_zero(p.y);
}

template <class F>
ZK_HD Affine<F> aff_neg(const Affine<F>& p) {
    return {p.x, F::neg(p.y)};
}

template <class F>
ZK_HD XYZZ<F> xyzz_inf() {
    return {F::zero(), F::zero(), F::zero(), F::zero()};
}

template <class F>
ZK_HD bool xyzz_is_inf(const XYZZ<F>& p) {
    return F::is_zero(p.ZZ);
}

template <class F>
ZK_HD XYZZ<F> xyzz_from_affine(const Affine<F>& p) {
    if (aff_is_inf<F>(p)) return xyzz_inf<F>();
    return {p.x, p.y, F::one(), F::one()};
}

template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) {
    return {p.X, F::neg(p.Y), p.ZZ, p.ZZZ};
}

// 2*P for an affine P (mdbl-2008-s-1 with a = 0)
template <class F>
ZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& p) {
    typedef typename F::T T;
    if (aff_is_inf<F>(p) || F::is_zero(p.y)) return xyzz_inf<F>();
    T U = F::dbl(p.y);
    T V = F::sqr(U);
    T W = F::mul(U, V);
    T S = F::mul(p.x, V);
    T xx = F::sqr(p.x);
    T M = F::add(F::dbl(xx), xx);
    T X3 = F::sub(F::sqr(M), F::dbl(S));
    T Y3 = F::mul_diff(M, S, X3, W, p.y);
    return {X3, Y3, V, W};
}

// 2*P (dbl-2008-s-1, a = 0)
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    typedef typename F::T T;
    if (xyzz_is_inf<F>(p) || F::is_zero(p.Y)) return xyzz_inf<F>();
    T U = F::dbl(p.Y);
    T V = F::sqr(U);
    T W = F::mul(U, V);
    T S = F::mul(p.X, V);
    T xx = F::sqr(p.X);
    T M = F::add(F::dbl(xx), xx);
    T X3 = F::sub(F::sqr(M), F::dbl(S));
    T Y3 = F::mul_diff(M, S, X3, W, p.Y);
    T ZZ3 = F::mul(V, p.ZZ);
    T ZZZ3 = F::mul(W, p.ZZZ);
    return {X3, Y3, ZZ3, ZZZ3};
}

// acc + q, q affine (madd-2008-s): 8M + 2S.  Handles infinity, doubling and cancellation.
template <class F>
ZK_HD void xyzz_add_affine(XYZZ<F>& acc, const Affine<F>& q) {
    typedef typename F::T T;
    if (aff_is_inf<F>(q)) return;
    if (xyzz_is_inf<F>(acc)) {
        acc = {q.x, q.y, F::one(), F::one()};
        return;
    }
    T U2 = F::mul(q.x, acc.ZZ);
    T S2 = F::mul(q.y, acc.ZZZ);
    T Pd = F::sub(U2, acc.X);
    T R = F::sub(S2, acc.Y);
    if (F::is_zero(Pd)) {
        if (F::is_zero(R)) acc = xyzz_dbl_affine<F>(q);
        else acc = xyzz_inf<F>();
        return;
    }
    T PP = F::sqr(Pd);
    T PPP = F::mul(Pd, PP);
    T Q = F::mul(acc.X, PP);
    T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    T Y3 = F::mul_diff(R, Q, X3, acc.Y, PPP);
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = F::mul(acc.ZZ, PP);
    acc.ZZZ = F::mul(acc.ZZZ, PPP);
}

// acc + (+-q) for the quadratic-extension groups with relaxed ranges, component by component (P = F::Params, every
// component normalised):
//   X in [0, 4p); Y, ZZ, ZZZ and the base in [0, 2p]
//   P = U2 - X + 4p < 6p,  R = S2 - Y + 2p < 4p           carries only, no range selection
//   PP = P^2 as ((P0 + P1)(P0 - P1 + 8p), (2 P0) P1): 12 * 14 = 168 <= R/p (169.3 for BN254 Fq), both < 2p;  RR likewise
//   PPP = P PP, Q = X PP with the c1 of the left factor negated lazily against 8p
//   X3 = RR - PPP + 2p - 2Q brought into [0, 4p) per component
//   Y3 = R (Q - X3 + 4p) - Y PPP: nine limbs -> one four-product reduction per component (no subtraction at all),
//        wider fields -> two products and an ordinary subtraction
// tools/model_relaxed_g2.py replays this on integers with the limb, column and value bounds asserted; the host build of the
// library runs the same code against the plain formulas (tests/test_host_lib.py).
template <class F>
ZK_HD void xyzz_add_affine_relaxed2(XYZZ<F>& acc, const Affine<F>& q, bool negate) {
    typedef typename F::T T;
    typedef typename F::Params P;
    // the sentinel and an empty accumulator are exact zeros (written as such), so the limb test is enough here
    if (F::is_zero_limbs(q.x) && F::is_zero_limbs(q.y)) return;
    if (F::is_zero_limbs(acc.ZZ)) {
        acc = {q.x, negate ? F::neg(q.y) : q.y, F::one(), F::one()};
        return;
    }
    const T U2 = fp2_mul_relaxed<P, 4>(q.x, acc.ZZ);
    // -y as 2p - y (normalised, <= 2p): a product operand like any other
    T ny = {fp_sub_k<P, 2>(fp_zero<P>(), q.y.c0), fp_sub_k<P, 2>(fp_zero<P>(), q.y.c1)};
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        ny.c0.v[i] = negate ? ny.c0.v[i] : q.y.c0.v[i];
        ny.c1.v[i] = negate ? ny.c1.v[i] : q.y.c1.v[i];
    }
    const T S2 = fp2_mul_relaxed<P, 4>(ny, acc.ZZZ);
    const T Pd = {fp_sub_k<P, 4>(U2.c0, acc.X.c0), fp_sub_k<P, 4>(U2.c1, acc.X.c1)};
    const T R = {fp_sub_k<P, 2>(S2.c0, acc.Y.c0), fp_sub_k<P, 2>(S2.c1, acc.Y.c1)};
    const T PP = fp2_sqr_relaxed<P, 8>(Pd);
    const T RR = fp2_sqr_relaxed<P, 4>(R);
    if (fp_is_zero<P>(PP.c0) && fp_is_zero<P>(PP.c1)) {  // P^2 = 0 in the field: same x
        if (fp_is_zero<P>(RR.c0) && fp_is_zero<P>(RR.c1)) {
            Affine<F> d = q;
            if (negate) d.y = F::neg(d.y);
            acc = xyzz_dbl_affine<F>(d);
        } else {
            acc = xyzz_inf<F>();
        }
        return;
    }
    const T PPP = fp2_mul_relaxed<P, 8>(Pd, PP);
    const T Q = fp2_mul_relaxed<P, 8>(acc.X, PP);
    const T X3 = {fp_sub_twice_sel4<P>(fp_sub_k<P, 2>(RR.c0, PPP.c0), Q.c0), fp_sub_twice_sel4<P>(fp_sub_k<P, 2>(RR.c1, PPP.c1), Q.c1)};
    const T D = {fp_sub_k<P, 4>(Q.c0, X3.c0), fp_sub_k<P, 4>(Q.c1, X3.c1)};   // < 6p
    if constexpr (P::N <= 9) {
        const Fp<P> nR1 = fp_neg_lazy_k<P, 8>(R.c1), nY0 = fp_neg_lazy_k<P, 4>(acc.Y.c0), nY1 = fp_neg_lazy_k<P, 4>(acc.Y.c1);
        acc.Y = {fp_mul4<P>(R.c0, D.c0, nR1, D.c1, nY0, PPP.c0, acc.Y.c1, PPP.c1),
                 fp_mul4<P>(R.c0, D.c1, nY0, PPP.c1, nY1, PPP.c0, R.c1, D.c0)};
    } else {
        acc.Y = fp2_sub<P>(fp2_mul_relaxed<P, 8>(R, D), fp2_mul_relaxed<P, 4>(acc.Y, PPP));
    }
    acc.X = X3;
    acc.ZZ = fp2_mul_relaxed<P, 4>(acc.ZZ, PP);
    acc.ZZZ = fp2_mul_relaxed<P, 4>(acc.ZZZ, PPP);
}

// The relaxed steps leave X in [0, 4p).  For the base-field groups every later use of X is a product, which does not mind; the
// Fp2 arithmetic adds and subtracts components (fp2_sqr in the doublings of the reduction stages), so a lane that is done
// accumulating brings X back below 2p before the point leaves its registers.
template <class F>
ZK_HD void xyzz_relaxed_finish(XYZZ<F>& acc) {
    if constexpr (F::RELAXED2) {
        typedef typename F::Params P;
        acc.X = {fp_reduce_2p<P>(acc.X.c0), fp_reduce_2p<P>(acc.X.c1)};
    }
}

#if defined(__HIPCC__)
// Device form used by the MSM inner loop: the affine operand is read from memory (16-byte vector loads of
// the packed Montgomery row) and dies right after U2/S2, so it does not occupy registers across the ten
// field multiplications; the (rare) doubling case simply reads it again.
template <class F>
__device__ __forceinline__ Affine<F> load_affine_row(const uint32_t* p) {
    uint32_t w[2 * F::LIMBS];
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2 * F::LIMBS / 4; ++i) {
        uint4 t = q[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
    return {F::load(w), F::load(w + F::LIMBS)};
}

template <class F>
__device__ __forceinline__ void xyzz_add_affine_mem(XYZZ<F>& acc, const uint32_t* src, bool negate) {
    typedef typename F::T T;
    if constexpr (F::RELAXED) {
        // Base-field groups: the accumulator's X lives in [0, 4p) (Y, ZZ, ZZZ in [0, 2p)) and the differences feed
        // products without the range selection of F::sub -- every consumer of an accumulator coordinate is a product,
        // which tolerates operands up to (a/p)(b/p) <= R/p (>= 168 for the two base fields):
        //   P = U2 - X + 4p < 6p,  R = S2 - Y + 2p < 4p,  PP = P^2 (36),  PPP = P PP (12),  Q = X PP (8),  R^2 (16),
        //   X3 = R^2 - PPP + 2p - 2Q brought into [0, 4p),  Y3 = R (Q - X3) - Y PPP as one double product.
        T U2, S2;
        {
            Affine<F> q = load_affine_row<F>(src);
            // the sentinel and an empty accumulator are exact zeros (written as such), so the limb test is enough here
            if (F::is_zero_limbs(q.x) && F::is_zero_limbs(q.y)) return;
            if (F::is_zero_limbs(acc.ZZ)) {
                acc = {q.x, negate ? F::neg(q.y) : q.y, F::one(), F::one()};
                return;
            }
            U2 = F::mul(q.x, acc.ZZ);
            T ny = F::neg_for_mul(q.y);  // 4p - y, limbs < 2*2^29: one operand of the next product only
#pragma unroll
            for (int i = 0; i < F::REGS; ++i) ny.v[i] = negate ? ny.v[i] : q.y.v[i];
            S2 = F::mul(ny, acc.ZZZ);
        }
        T Pd = F::template sub_k<4>(U2, acc.X);
        T R = F::template sub_k<2>(S2, acc.Y);
        T PP = F::sqr(Pd);
        T RR = F::sqr(R);
        if (F::is_zero(PP)) {  // P = 0 mod p (p is prime): same x
            if (F::is_zero(RR)) {
                Affine<F> q = load_affine_row<F>(src);
                if (negate) q.y = F::neg(q.y);
                acc = xyzz_dbl_affine<F>(q);
            } else {
                acc = xyzz_inf<F>();
            }
            return;
        }
        T PPP = F::mul(Pd, PP);
        T Q = F::mul(acc.X, PP);
        T X3 = F::x3_sel4(F::template sub_k<2>(RR, PPP), Q);
        acc.Y = F::y3_relaxed(R, Q, X3, acc.Y, PPP);
        acc.X = X3;
        acc.ZZ = F::mul(acc.ZZ, PP);
        acc.ZZZ = F::mul(acc.ZZZ, PPP);
    } else if constexpr (F::RELAXED2) {
        xyzz_add_affine_relaxed2<F>(acc, load_affine_row<F>(src), negate);
    } else {
        T U2, S2;
        {
            Affine<F> q = load_affine_row<F>(src);
            // the sentinel and an empty accumulator are exact zeros (written as such), so the limb test is enough here
            if (F::is_zero_limbs(q.x) && F::is_zero_limbs(q.y)) return;
            if (negate) q.y = F::neg(q.y);
            if (F::is_zero_limbs(acc.ZZ)) {
                acc = {q.x, q.y, F::one(), F::one()};
                return;
            }
            U2 = F::mul(q.x, acc.ZZ);
            S2 = F::mul(q.y, acc.ZZZ);
        }
        T Pd = F::sub(U2, acc.X);
        T R = F::sub(S2, acc.Y);
        if (F::is_zero(Pd)) {
            if (F::is_zero(R)) {
                Affine<F> q = load_affine_row<F>(src);
                if (negate) q.y = F::neg(q.y);
                acc = xyzz_dbl_affine<F>(q);
            } else {
                acc = xyzz_inf<F>();
            }
            return;
        }
        T PP = F::sqr(Pd);
        T PPP = F::mul(Pd, PP);
        T Q = F::mul(acc.X, PP);
        T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
        acc.Y = F::mul_diff(R, Q, X3, acc.Y, PPP);
        acc.X = X3;
        acc.ZZ = F::mul(acc.ZZ, PP);
        acc.ZZZ = F::mul(acc.ZZZ, PPP);
    }
}
#endif

// acc + q, both XYZZ (add-2008-s): 12M + 2S
template <class F>
ZK_HD XYZZ<F> xyzz_add(const XYZZ<F>& p, const XYZZ<F>& q) {
    typedef typename F::T T;
    if (xyzz_is_inf<F>(q)) return p;
    if (xyzz_is_inf<F>(p)) return q;
    T U1 = F::mul(p.X, q.ZZ);
    T U2 = F::mul(q.X, p.ZZ);
    T S1 = F::mul(p.Y, q.ZZZ);
    T S2 = F::mul(q.Y, p.ZZZ);
    T Pd = F::sub(U2, U1);
    T R = F::sub(S2, S1);
    if (F::is_zero(Pd)) {
        if (F::is_zero(R)) return xyzz_dbl<F>(p);
        return xyzz_inf<F>();
    }
    T PP = F::sqr(Pd);
    T PPP = F::mul(Pd, PP);
    T Q = F::mul(U1, PP);
    T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    T Y3 = F::mul_diff(R, Q, X3, S1, PPP);
    T ZZ3 = F::mul(F::mul(p.ZZ, q.ZZ), PP);
    T ZZZ3 = F::mul(F::mul(p.ZZZ, q.ZZZ), PPP);
    return {X3, Y3, ZZ3, ZZZ3};
}

// unique affine representative: x = X/ZZ, y = Y/ZZZ  (one inversion: 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2 ...
// simpler: invert ZZ*ZZZ once)
template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    typedef typename F::T T;
    if (xyzz_is_inf<F>(p)) return {F::zero(), F::zero()};
    T inv = F::inv(F::mul(p.ZZ, p.ZZZ));  // 1/(ZZ*ZZZ)
    T izz = F::mul(inv, p.ZZZ);           // 1/ZZ
    T izzz = F::mul(inv, p.ZZ);           // 1/ZZZ
    return {F::mul(p.X, izz), F::mul(p.Y, izzz)};
}

// k*P by left-to-right double-and-add; k canonical little-endian 32-bit limbs (already < r)
template <class F>
ZK_HD XYZZ<F> xyzz_scalar_mul(const Affine<F>& p, const uint32_t* k, int nlimbs) {
    XYZZ<F> acc = xyzz_inf<F>();
    bool started = false;
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        if (started) acc = xyzz_dbl<F>(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) {
            xyzz_add_affine<F>(acc, p);
            started = true;
        }
    }
    return acc;
}

// y^2 == x^3 + b
template <class F>
ZK_HD bool aff_on_curve(const Affine<F>& p, const typename F::T& b) {
    if (aff_is_inf<F>(p)) return true;
    typename F::T lhs = F::sqr(p.y);
    typename F::T rhs = F::add(F::mul(F::sqr(p.x), p.x), b);
    return F::eq(lhs, rhs);
}

}  // namespace zkmi
