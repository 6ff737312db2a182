// pairing.hip -- host-side optimal-ate pairing for BN254 and BLS12-381 (verification only).
//
// Stands in for pairing / multi_pairing of the reference (src/bn254/curve.rs:417-437 -> ark-ec
// `Bn254::multi_pairing`, bls12_381 twin), which also run on the CPU; only Groth16.verify uses them
// (python/zksnake/groth16/protocol.py:167-186).  SURVEY.md 8f row 3.
//
// Tower: Fp2 = Fp[u]/(u^2+1) (field.hip.h), Fp12 = Fp2[w]/(w^6 - xi), xi = 9+u (BN254) / 1+u (BLS12-381);
// an Fp12 element is six Fp2 coefficients of 1, w, .., w^5.  The Miller loop keeps T affine on the twist
// (one Fp2 inversion per step); the final exponentiation is f -> conj(f)/f, then a plain square-and-multiply by (p^6+1)/r --
// ~10 ms on 64-bit limbs (host64.hip.h), which keeps the code free of curve-specific Frobenius tables (only the two BN254
// loop-tail points need the twist Frobenius constants).  Groth16.verify folds its four pairings into ONE call, i.e. one
// final exponentiation.
#include <vector>
#include "common.hip.h"
#include "pairing_params.h"

namespace zkmi {

template <class P>
struct Fp12 {
    Fp2_64<P> c[6];
};

template <class P, class PP>
struct PairingEngine {
    // host arithmetic on 64-bit limbs (host64.hip.h): this engine never runs on the device
    typedef Fp64Ops<P> B;
    typedef Fp2Ops64<P> O2;
    typedef Fp64<P> E1;
    typedef Fp2_64<P> E2;
    typedef Fp12<P> E12;

    static E2 konst(const uint32_t* w) { return O2::from_canonical(w); }

    static E12 one() {
        E12 r;
        for (int i = 0; i < 6; ++i) r.c[i] = O2::zero();
        r.c[0] = O2::one();
        return r;
    }

    // schoolbook product modulo w^6 = xi
    static E12 mul(const E12& a, const E12& b, const E2& xi) {
        E2 t[11];
        for (int i = 0; i < 11; ++i) t[i] = O2::zero();
        for (int i = 0; i < 6; ++i) {
            if (O2::is_zero(a.c[i])) continue;
            for (int j = 0; j < 6; ++j) {
                if (O2::is_zero(b.c[j])) continue;
                t[i + j] = O2::add(t[i + j], O2::mul(a.c[i], b.c[j]));
            }
        }
        E12 r;
        for (int k = 0; k < 6; ++k) {
            r.c[k] = t[k];
            if (k + 6 < 11) r.c[k] = O2::add(r.c[k], O2::mul(t[k + 6], xi));
        }
        return r;
    }

    static E12 conj(const E12& a) {  // Frobenius p^6: w -> -w
        E12 r = a;
        for (int i = 1; i < 6; i += 2) r.c[i] = O2::neg(a.c[i]);
        return r;
    }

    struct Pt2 { E2 x, y; };

    // line through T and Q (tangent when they coincide) evaluated at P = (xp, yp); T <- T + Q
    static E12 line_and_step(Pt2& T, const Pt2& Q, bool doubling, const E1& xp, const E1& yp) {
        E2 lam;
        if (doubling) {
            E2 xx = O2::sqr(T.x);
            lam = O2::mul(O2::add(O2::dbl(xx), xx), O2::inv(O2::dbl(T.y)));
        } else {
            lam = O2::mul(O2::sub(Q.y, T.y), O2::inv(O2::sub(Q.x, T.x)));
        }
        E2 x3 = O2::sub(O2::sub(O2::sqr(lam), T.x), Q.x);
        E2 y3 = O2::sub(O2::mul(lam, O2::sub(T.x, x3)), T.y);
        E2 cterm = O2::sub(O2::mul(lam, T.x), T.y);               // lam * xT - yT
        E2 lxp = {B::neg(B::mul(lam.c0, xp)), B::neg(B::mul(lam.c1, xp))};  // -lam * xp
        E2 ypl = {yp, B::zero()};
        E12 l;
        for (int i = 0; i < 6; ++i) l.c[i] = O2::zero();
        if (PP::IS_BN) {
            // D-type twist: yp - lam xp w + (lam xT - yT) w^3
            l.c[0] = ypl; l.c[1] = lxp; l.c[3] = cterm;
        } else {
            // M-type twist, scaled by w^3 (an element of Fp4, killed by the final exponentiation):
            // (lam xT - yT) - lam xp w^2 + yp w^3
            l.c[0] = cterm; l.c[2] = lxp; l.c[3] = ypl;
        }
        T.x = x3;
        T.y = y3;
        return l;
    }

    static E12 miller(const E1& xp, const E1& yp, const Pt2& Q, const E2& xi) {
        Pt2 T = Q;
        E12 f = one();
        for (int i = PP::LOOP_BITS - 2; i >= 0; --i) {
            E12 l = line_and_step(T, T, true, xp, yp);
            f = mul(mul(f, f, xi), l, xi);
            if ((PP::LOOP[i >> 5] >> (i & 31)) & 1) {
                l = line_and_step(T, Q, false, xp, yp);
                f = mul(f, l, xi);
            }
        }
        if (PP::IS_BN) {
            Pt2 q1 = {O2::mul(E2{Q.x.c0, B::neg(Q.x.c1)}, konst(PP::G_X1)), O2::mul(E2{Q.y.c0, B::neg(Q.y.c1)}, konst(PP::G_Y1))};
            Pt2 q2 = {O2::mul(Q.x, konst(PP::G_X2)), O2::neg(O2::mul(Q.y, konst(PP::G_Y2)))};
            E12 l = line_and_step(T, q1, false, xp, yp);
            f = mul(f, l, xi);
            l = line_and_step(T, q2, false, xp, yp);
            f = mul(f, l, xi);
        }
        if (PP::LOOP_NEGATIVE) f = conj(f);
        return f;
    }

    // ---- inversion: Fp12 = Fp6[w]/(w^2 - v), Fp6 = Fp2[v]/(v^3 - xi); a = a0 + a1 w with a0 = (c0, c2, c4), a1 = (c1, c3, c5)
    struct E6 { E2 x[3]; };
    static E6 mul6(const E6& a, const E6& b, const E2& xi) {
        E2 t[5];
        for (int i = 0; i < 5; ++i) t[i] = O2::zero();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) t[i + j] = O2::add(t[i + j], O2::mul(a.x[i], b.x[j]));
        return {{O2::add(t[0], O2::mul(t[3], xi)), O2::add(t[1], O2::mul(t[4], xi)), t[2]}};
    }
    static E6 sub6(const E6& a, const E6& b) { return {{O2::sub(a.x[0], b.x[0]), O2::sub(a.x[1], b.x[1]), O2::sub(a.x[2], b.x[2])}}; }
    static E6 mul_v(const E6& a, const E2& xi) { return {{O2::mul(a.x[2], xi), a.x[0], a.x[1]}}; }  // times v
    static E6 inv6(const E6& a, const E2& xi) {
        const E2 c0 = O2::sub(O2::sqr(a.x[0]), O2::mul(xi, O2::mul(a.x[1], a.x[2])));
        const E2 c1 = O2::sub(O2::mul(xi, O2::sqr(a.x[2])), O2::mul(a.x[0], a.x[1]));
        const E2 c2 = O2::sub(O2::sqr(a.x[1]), O2::mul(a.x[0], a.x[2]));
        const E2 t = O2::add(O2::mul(a.x[0], c0), O2::mul(xi, O2::add(O2::mul(a.x[2], c1), O2::mul(a.x[1], c2))));
        const E2 ti = O2::inv(t);
        return {{O2::mul(c0, ti), O2::mul(c1, ti), O2::mul(c2, ti)}};
    }
    static E12 inv12(const E12& a, const E2& xi) {
        const E6 a0 = {{a.c[0], a.c[2], a.c[4]}}, a1 = {{a.c[1], a.c[3], a.c[5]}};
        const E6 d = inv6(sub6(mul6(a0, a0, xi), mul_v(mul6(a1, a1, xi), xi)), xi);   // 1 / (a0^2 - v a1^2)
        const E6 r0 = mul6(a0, d, xi), r1 = mul6(a1, d, xi);
        E12 r;
        for (int i = 0; i < 3; ++i) {
            r.c[2 * i] = r0.x[i];
            r.c[2 * i + 1] = O2::neg(r1.x[i]);
        }
        return r;
    }

    // f^((p^12 - 1) / r) = (conj(f) / f)^((p^6 + 1) / r): the factor p^6 - 1 costs a conjugation and one inversion, the rest is a
    // plain square-and-multiply with an exponent of half the length
    static E12 final_exp(const E12& f0, const E2& xi) {
        const E12 f = mul(conj(f0), inv12(f0, xi), xi);
        E12 acc = one();
        bool started = false;
        for (int i = PP::FINAL_EXP_WORDS * 32 - 1; i >= 0; --i) {
            if (started) acc = mul(acc, acc, xi);
            if ((PP::FINAL_EXP[i >> 5] >> (i & 31)) & 1) {
                acc = started ? mul(acc, f, xi) : f;
                started = true;
            }
        }
        return acc;
    }

    // product of pairings; points as canonical 64-bit limb arrays (all-zero = infinity)
    static int run(uint64_t n, const uint64_t* g1, const uint64_t* g2, uint64_t* out) {
        const E2 xi = konst(PP::XI);
        E12 f = one();
        const size_t s1 = 2 * P::W / 2, s2 = 4 * P::W / 2;  // 64-bit limbs per G1 / G2 point
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t* a = reinterpret_cast<const uint32_t*>(g1 + i * s1);
            const uint32_t* b = reinterpret_cast<const uint32_t*>(g2 + i * s2);
            bool inf1 = true, inf2 = true;
            for (int k = 0; k < 2 * P::W; ++k) inf1 = inf1 && a[k] == 0;
            for (int k = 0; k < 4 * P::W; ++k) inf2 = inf2 && b[k] == 0;
            if (inf1 || inf2) continue;
            E1 xp = B::from_canonical(a), yp = B::from_canonical(a + P::W);
            Pt2 Q = {konst(b), konst(b + 2 * P::W)};
            f = mul(f, miller(xp, yp, Q, xi), xi);
        }
        f = final_exp(f, xi);
        uint32_t* o = reinterpret_cast<uint32_t*>(out);
        for (int k = 0; k < 6; ++k) {
            B::to_canonical(o + (2 * k) * P::W, f.c[k].c0);
            B::to_canonical(o + (2 * k + 1) * P::W, f.c[k].c1);
        }
        return ZK_OK;
    }
};

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_gt_limbs(int curve) {
    int f = zk_fq_limbs(curve);
    return f < 0 ? -1 : 12 * f;
}

int zk_multi_pairing(int curve, uint64_t n, const uint64_t* g1_points, const uint64_t* g2_points, uint64_t* out) {
    if (curve == ZK_CURVE_BN254) return PairingEngine<BnFqParams, Bn254Pairing>::run(n, g1_points, g2_points, out);
    if (curve == ZK_CURVE_BLS12_381) return PairingEngine<BlsFqParams, Bls381Pairing>::run(n, g1_points, g2_points, out);
    return fail(ZK_ERR_ARG, "unknown curve");
}

}  // extern "C"
