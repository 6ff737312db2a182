// pairing.hip -- host-side optimal-ate pairing for BN254 and BLS12-381 (verification only).
//
// Stands in for pairing / multi_pairing of the reference (src/bn254/curve.rs:417-437 -> ark-ec
// `Bn254::multi_pairing`, bls12_381 twin), which also run on the CPU; only Groth16.verify uses them
// (python/zksnake/groth16/protocol.py:167-186).  SURVEY.md 8f row 3.
//
// Tower: Fp2 = Fp[u]/(u^2+1) (field.cuh), Fp12 = Fp2[w]/(w^6 - xi), xi = 9+u (BN254) / 1+u (BLS12-381);
// an Fp12 element is six Fp2 coefficients of 1, w, .., w^5.  The Miller loop keeps T affine on the twist
// (one Fp2 inversion per step); the final exponentiation is a plain square-and-multiply by (p^12-1)/r --
// a few tens of milliseconds, which is irrelevant next to a proof and keeps the code free of
// curve-specific Frobenius tables (only the two BN254 loop-tail points need the twist Frobenius constants).
#include <vector>
#include "common.cuh"
#include "pairing_params.h"

namespace zkmi {

template <class P>
struct Fp12 {
    Fp2<P> c[6];
};

template <class P, class PP>
struct PairingEngine {
    typedef Fp2<P> E2;
    typedef Fp12<P> E12;

    static E2 konst(const uint32_t* w) { return {fp_from_canonical<P>(w), fp_from_canonical<P>(w + P::W)}; }

    static E12 one() {
        E12 r;
        for (int i = 0; i < 6; ++i) r.c[i] = fp2_zero<P>();
        r.c[0] = fp2_one<P>();
        return r;
    }

    // schoolbook product modulo w^6 = xi
    static E12 mul(const E12& a, const E12& b, const E2& xi) {
        E2 t[11];
        for (int i = 0; i < 11; ++i) t[i] = fp2_zero<P>();
        for (int i = 0; i < 6; ++i) {
            if (fp2_is_zero<P>(a.c[i])) continue;
            for (int j = 0; j < 6; ++j) {
                if (fp2_is_zero<P>(b.c[j])) continue;
                t[i + j] = fp2_add<P>(t[i + j], fp2_mul<P>(a.c[i], b.c[j]));
            }
        }
        E12 r;
        for (int k = 0; k < 6; ++k) {
            r.c[k] = t[k];
            if (k + 6 < 11) r.c[k] = fp2_add<P>(r.c[k], fp2_mul<P>(t[k + 6], xi));
        }
        return r;
    }

    static E12 conj(const E12& a) {  // Frobenius p^6: w -> -w
        E12 r = a;
        for (int i = 1; i < 6; i += 2) r.c[i] = fp2_neg<P>(a.c[i]);
        return r;
    }

    struct Pt2 { E2 x, y; };

    // line through T and Q (tangent when they coincide) evaluated at P = (xp, yp); T <- T + Q
    static E12 line_and_step(Pt2& T, const Pt2& Q, bool doubling, const Fp<P>& xp, const Fp<P>& yp) {
        E2 lam;
        if (doubling) {
            E2 xx = fp2_sqr<P>(T.x);
            lam = fp2_mul<P>(fp2_add<P>(fp2_dbl<P>(xx), xx), fp2_inv<P>(fp2_dbl<P>(T.y)));
        } else {
            lam = fp2_mul<P>(fp2_sub<P>(Q.y, T.y), fp2_inv<P>(fp2_sub<P>(Q.x, T.x)));
        }
        E2 x3 = fp2_sub<P>(fp2_sub<P>(fp2_sqr<P>(lam), T.x), Q.x);
        E2 y3 = fp2_sub<P>(fp2_mul<P>(lam, fp2_sub<P>(T.x, x3)), T.y);
        E2 cterm = fp2_sub<P>(fp2_mul<P>(lam, T.x), T.y);               // lam * xT - yT
        E2 lxp = {fp_neg<P>(fp_mul<P>(lam.c0, xp)), fp_neg<P>(fp_mul<P>(lam.c1, xp))};  // -lam * xp
        E2 ypl = {yp, fp_zero<P>()};
        E12 l;
        for (int i = 0; i < 6; ++i) l.c[i] = fp2_zero<P>();
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

    static E12 miller(const Fp<P>& xp, const Fp<P>& yp, const Pt2& Q, const E2& xi) {
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
            Pt2 q1 = {fp2_mul<P>({Q.x.c0, fp_neg<P>(Q.x.c1)}, konst(PP::G_X1)), fp2_mul<P>({Q.y.c0, fp_neg<P>(Q.y.c1)}, konst(PP::G_Y1))};
            Pt2 q2 = {fp2_mul<P>(Q.x, konst(PP::G_X2)), fp2_neg<P>(fp2_mul<P>(Q.y, konst(PP::G_Y2)))};
            E12 l = line_and_step(T, q1, false, xp, yp);
            f = mul(f, l, xi);
            l = line_and_step(T, q2, false, xp, yp);
            f = mul(f, l, xi);
        }
        if (PP::LOOP_NEGATIVE) f = conj(f);
        return f;
    }

    static E12 final_exp(const E12& f, const E2& xi) {
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
            Fp<P> xp = fp_from_canonical<P>(a), yp = fp_from_canonical<P>(a + P::W);
            Pt2 Q = {konst(b), konst(b + 2 * P::W)};
            f = mul(f, miller(xp, yp, Q, xi), xi);
        }
        f = final_exp(f, xi);
        uint32_t* o = reinterpret_cast<uint32_t*>(out);
        for (int k = 0; k < 6; ++k) {
            fp_to_canonical<P>(o + (2 * k) * P::W, f.c[k].c0);
            fp_to_canonical<P>(o + (2 * k + 1) * P::W, f.c[k].c1);
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
