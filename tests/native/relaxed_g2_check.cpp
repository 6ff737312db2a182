// Host build of the library's own field / curve code (csrc/field.hip.h, csrc/curve.hip.h are host + device): the relaxed-range
// G2 mixed addition of the accumulate kernel (xyzz_add_affine_relaxed2) against the plain formulas (xyzz_add_affine) on
// chains of additions, both Fp2 fields, both signs, with the special cases (empty accumulator, infinity base, P + P, P - P).
// Built and run by tests/test_host_lib.py with g++; exits non-zero on the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../zksnake_amd/csrc/curve.hip.h"
using namespace zkmi;

template <class P> static Fp<P> rnd(uint64_t& s, bool small) {
    uint32_t w[P::W];
    for (int i = 0; i < P::W; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = small ? 0u : (uint32_t)(s >> 20); }
    if (small) w[0] = (uint32_t)(s >> 40) & 3u;
    w[P::W - 1] &= 0x00FFFFFF;
    return fp_from_canonical<P>(w);
}
template <class P> static bool same(const Fp2<P>& a, const Fp2<P>& b) {
    uint32_t x[2 * P::W], y[2 * P::W];
    fp_to_canonical<P>(x, a.c0); fp_to_canonical<P>(x + P::W, a.c1);
    fp_to_canonical<P>(y, b.c0); fp_to_canonical<P>(y + P::W, b.c1);
    return memcmp(x, y, sizeof(x)) == 0;
}
template <class P> static bool below_2p(const Fp<P>& a) {   // a < 2p  <=>  a - 2p is negative
    int64_t c = 0;
    for (int i = 0; i < P::N; ++i) { c += (int64_t)a.v[i] - (int64_t)P::P2[i]; c >>= (i < P::N - 1 ? LIMB_BITS : 0); }
    return c < 0;
}
template <class P> static int run(const char* name) {
    typedef Fp2Ops<P> F;
    uint64_t s = 20261005;
    for (int t = 0; t < 400; ++t) {
        const bool small = (t % 7) == 3;   // tiny coordinates: zero components and carries at the low end
        XYZZ<F> r1 = xyzz_inf<F>(), r2 = r1;
        Affine<F> last = {F::zero(), F::zero()};
        for (int step = 0; step < 24; ++step) {
            Affine<F> q = {{rnd<P>(s, small), rnd<P>(s, small)}, {rnd<P>(s, small), rnd<P>(s, small)}};
            int neg = (int)(s & 1);
            if (step == 7) q = {F::zero(), F::zero()};                   // the infinity sentinel: no-op
            if (step == 11) { q = last; }                                // same x: doubling or cancellation follows below
            Affine<F> qq = q;
            if (neg) qq.y = F::neg(qq.y);
            xyzz_add_affine<F>(r1, qq);
            xyzz_add_affine_relaxed2<F>(r2, q, neg != 0);
            last = q;
            XYZZ<F> fin = r2;
            xyzz_relaxed_finish<F>(fin);
            const bool ok = same<P>(r1.X, r2.X) && same<P>(r1.Y, r2.Y) && same<P>(r1.ZZ, r2.ZZ) && same<P>(r1.ZZZ, r2.ZZZ) &&
                            same<P>(fin.X, r1.X) && below_2p<P>(fin.X.c0) && below_2p<P>(fin.X.c1);
            if (!ok) { printf("%s: mismatch in chain %d at step %d (negate %d)\n", name, t, step, neg); return 1; }
        }
        // P + P and P - P through the relaxed step: accumulator = one affine point, then the same point again
        for (int neg = 0; neg < 2; ++neg) {
            Affine<F> q = {{rnd<P>(s, false), rnd<P>(s, false)}, {rnd<P>(s, false), rnd<P>(s, false)}};
            XYZZ<F> a1 = xyzz_inf<F>(), a2 = a1;
            xyzz_add_affine<F>(a1, q); xyzz_add_affine_relaxed2<F>(a2, q, false);
            Affine<F> qq = q; if (neg) qq.y = F::neg(qq.y);
            xyzz_add_affine<F>(a1, qq); xyzz_add_affine_relaxed2<F>(a2, q, neg != 0);
            if (xyzz_is_inf<F>(a1) != xyzz_is_inf<F>(a2) || (!xyzz_is_inf<F>(a1) && !(same<P>(a1.X, a2.X) && same<P>(a1.Y, a2.Y) && same<P>(a1.ZZ, a2.ZZ)))) {
                printf("%s: doubling / cancellation differs (negate %d)\n", name, neg); return 1;
            }
        }
    }
    printf("%s: ok\n", name);
    return 0;
}
int main() { return run<BnFqParams>("BN254 Fq2") + run<BlsFqParams>("BLS12-381 Fq2"); }
