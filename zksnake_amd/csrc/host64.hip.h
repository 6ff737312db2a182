// host64.hip.h -- prime-field arithmetic on 64-bit limbs for the HOST side of the library: the sequential tail of an MSM
// (Horner over the windows: c doublings per window, one inversion) and single-point operations.
//
// The device representation (29-bit limbs, field.hip.h) is built around what a GPU lane can issue; on an x86 core the same
// product is a 4 x 4 (BN254) or 6 x 6 (BLS12-381) schoolbook on 64 x 64 -> 128-bit multiplications, about three times
// faster than running the 29-bit code there.  What the reference does at this point is ark-ff's 64-bit Montgomery
// arithmetic as well (ark-ff 0.4.2 `MontBackend`, behind src/bn254/curve.rs:77-132 point operations).
//
// Montgomery form with R = 2^(64 L), values fully reduced in [0, p).  All constants are derived at first use from the
// modulus words of field_params.h.
#pragma once
#include <cstdint>
#include <cstring>
#include "curve.hip.h"

namespace zkmi {

template <class P>
struct Fp64 {
    static constexpr int L = P::W / 2;
    uint64_t v[L];
};

template <class P>
struct Fp64Consts {
    static constexpr int L = P::W / 2;
    uint64_t p[L], one[L], r2[L], from29[L];  // modulus, R mod p, R^2 mod p, R^2 / 2^(29 N) mod p
    uint64_t n0;                              // -p^-1 mod 2^64
};

template <class P>
struct Fp64Ops {
    typedef Fp64<P> T;
    typedef P Params;
    static constexpr int L = P::W / 2;
    static constexpr int LIMBS = P::W;  // 32-bit words per element in device memory (as in FpOps)

    static inline bool geq(const uint64_t* a, const uint64_t* b) {
        for (int i = L - 1; i >= 0; --i)
            if (a[i] != b[i]) return a[i] > b[i];
        return true;
    }
    static inline void sub_raw(uint64_t* r, const uint64_t* a, const uint64_t* b) {
        unsigned __int128 borrow = 0;
        for (int i = 0; i < L; ++i) {
            unsigned __int128 t = (unsigned __int128)a[i] - b[i] - (uint64_t)borrow;
            r[i] = (uint64_t)t;
            borrow = (t >> 64) & 1;
        }
    }
    // CIOS Montgomery product with an explicit modulus (used while the constants are being derived)
    static inline void mont(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* p, uint64_t n0) {
        uint64_t t[L + 2];
        for (int i = 0; i < L + 2; ++i) t[i] = 0;
        for (int i = 0; i < L; ++i) {
            unsigned __int128 c = 0;
            for (int j = 0; j < L; ++j) {
                c += (unsigned __int128)a[j] * b[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L] = (uint64_t)c;
            t[L + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0;
            c = (unsigned __int128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < L; ++j) {
                c += (unsigned __int128)m * p[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L - 1] = (uint64_t)c;
            t[L] = t[L + 1] + (uint64_t)(c >> 64);
        }
        if (t[L] || geq(t, p)) sub_raw(r, t, p);
        else for (int i = 0; i < L; ++i) r[i] = t[i];
    }

    static const Fp64Consts<P>& K() {
        static const Fp64Consts<P> k = [] {
            Fp64Consts<P> c;
            for (int i = 0; i < L; ++i) c.p[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
            uint64_t inv = 1;
            for (int i = 0; i < 6; ++i) inv *= 2 - c.p[0] * inv;
            c.n0 = (uint64_t)0 - inv;
            auto dbl_mod = [&](uint64_t* x) {  // x = 2 x mod p, x < p
                uint64_t carry = 0;
                for (int i = 0; i < L; ++i) {
                    uint64_t nc = x[i] >> 63;
                    x[i] = (x[i] << 1) | carry;
                    carry = nc;
                }
                if (carry || geq(x, c.p)) sub_raw(x, x, c.p);
            };
            uint64_t x[L];
            for (int i = 0; i < L; ++i) x[i] = 0;
            x[0] = 1;
            for (int i = 0; i < 64 * L; ++i) dbl_mod(x);
            memcpy(c.one, x, sizeof(x));
            for (int i = 0; i < 64 * L; ++i) dbl_mod(x);
            memcpy(c.r2, x, sizeof(x));
            // from29 = R^2 / 2^(29 N): mont(v 2^(29 N), from29) = v R
            uint64_t r29[L];
            for (int i = 0; i < L; ++i) r29[i] = 0;
            r29[0] = 1;
            for (int i = 0; i < LIMB_BITS * P::N; ++i) dbl_mod(r29);
            uint64_t a[L], acc[L];
            mont(a, r29, c.r2, c.p, c.n0);          // 2^(29 N) R
            memcpy(acc, c.one, sizeof(acc));        // a^(p - 2) by square and multiply
            for (int i = 32 * P::W - 1; i >= 0; --i) {
                mont(acc, acc, acc, c.p, c.n0);
                if ((P::PM2[i >> 5] >> (i & 31)) & 1) mont(acc, acc, a, c.p, c.n0);
            }
            mont(c.from29, acc, c.r2, c.p, c.n0);   // (2^-(29 N) R) * R^2 / R = R^2 / 2^(29 N)
            return c;
        }();
        return k;
    }

    static inline T zero() { T r; for (int i = 0; i < L; ++i) r.v[i] = 0; return r; }
    static inline T one() { T r; memcpy(r.v, K().one, sizeof(r.v)); return r; }
    static inline bool is_zero(const T& a) { uint64_t z = 0; for (int i = 0; i < L; ++i) z |= a.v[i]; return z == 0; }
    static inline bool eq(const T& a, const T& b) { uint64_t z = 0; for (int i = 0; i < L; ++i) z |= a.v[i] ^ b.v[i]; return z == 0; }
    static inline T add(const T& a, const T& b) {
        T r;
        unsigned __int128 c = 0;
        for (int i = 0; i < L; ++i) {
            c += (unsigned __int128)a.v[i] + b.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || geq(r.v, K().p)) sub_raw(r.v, r.v, K().p);
        return r;
    }
    static inline T sub(const T& a, const T& b) {
        T r;
        if (geq(a.v, b.v)) {
            sub_raw(r.v, a.v, b.v);
        } else {
            uint64_t t[L];
            sub_raw(t, b.v, a.v);
            sub_raw(r.v, K().p, t);
        }
        return r;
    }
    static inline T neg(const T& a) { return is_zero(a) ? a : sub(zero(), a); }
    static inline T dbl(const T& a) { return add(a, a); }
    static inline T mul(const T& a, const T& b) { T r; mont(r.v, a.v, b.v, K().p, K().n0); return r; }
    static inline T sqr(const T& a) { return mul(a, a); }
    static inline T sub_for_mul(const T& a, const T& b) { return sub(a, b); }
    static inline T mul_diff(const T& m, const T& s, const T& x, const T& w, const T& y) { return sub(mul(m, sub(s, x)), mul(w, y)); }
    static inline T inv(const T& a) {
        T acc = one();
        for (int i = 32 * P::W - 1; i >= 0; --i) {
            acc = sqr(acc);
            if ((P::PM2[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
    // device form (W packed words of a 29-bit-limb Montgomery value, semi-reduced < 2p) -> this form
    static inline T from_device(const uint32_t* w) {
        T x, r;
        for (int i = 0; i < L; ++i) x.v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        mont(r.v, x.v, K().from29, K().p, K().n0);
        return r;
    }
    // canonical integer words (the ABI form) <-> this form
    static inline T from_canonical(const uint32_t* w) {
        T x, r;
        for (int i = 0; i < L; ++i) x.v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        while (geq(x.v, K().p)) sub_raw(x.v, x.v, K().p);
        mont(r.v, x.v, K().r2, K().p, K().n0);
        return r;
    }
    static inline void to_canonical(uint32_t* w, const T& a) {
        uint64_t o[L], r[L];
        for (int i = 0; i < L; ++i) o[i] = 0;
        o[0] = 1;
        mont(r, a.v, o, K().p, K().n0);
        memcpy(w, r, sizeof(r));
    }
};

template <class P>
struct Fp2_64 {
    Fp64<P> c0, c1;
};

// Fp[u] / (u^2 + 1)
template <class P>
struct Fp2Ops64 {
    typedef Fp2_64<P> T;
    typedef Fp64Ops<P> B;
    typedef P Params;
    static constexpr int LIMBS = 2 * P::W;
    static inline T zero() { return {B::zero(), B::zero()}; }
    static inline T one() { return {B::one(), B::zero()}; }
    static inline bool is_zero(const T& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    static inline bool eq(const T& a, const T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
    static inline T add(const T& a, const T& b) { return {B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static inline T sub(const T& a, const T& b) { return {B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static inline T neg(const T& a) { return {B::neg(a.c0), B::neg(a.c1)}; }
    static inline T dbl(const T& a) { return {B::dbl(a.c0), B::dbl(a.c1)}; }
    static inline T mul(const T& a, const T& b) {  // Karatsuba: three base products
        auto t0 = B::mul(a.c0, b.c0), t1 = B::mul(a.c1, b.c1);
        auto t2 = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return {B::sub(t0, t1), B::sub(B::sub(t2, t0), t1)};
    }
    static inline T sqr(const T& a) {
        auto t0 = B::mul(B::add(a.c0, a.c1), B::sub(a.c0, a.c1));
        auto t1 = B::mul(a.c0, a.c1);
        return {t0, B::dbl(t1)};
    }
    static inline T sub_for_mul(const T& a, const T& b) { return sub(a, b); }
    static inline T mul_diff(const T& m, const T& s, const T& x, const T& w, const T& y) { return sub(mul(m, sub(s, x)), mul(w, y)); }
    static inline T inv(const T& a) {
        auto d = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return {B::mul(a.c0, d), B::neg(B::mul(a.c1, d))};
    }
    static inline T from_device(const uint32_t* w) { return {B::from_device(w), B::from_device(w + P::W)}; }
    static inline T from_canonical(const uint32_t* w) { return {B::from_canonical(w), B::from_canonical(w + P::W)}; }
    static inline void to_canonical(uint32_t* w, const T& a) {
        B::to_canonical(w, a.c0);
        B::to_canonical(w + P::W, a.c1);
    }
};

// the tail's view of a curve group: field facade + conversions of whole points
template <class F64>
struct HostTail64 : F64 {
    static XYZZ<HostTail64> xyzz_from_device(const uint32_t* w) {
        constexpr int L = F64::LIMBS;
        return {F64::from_device(w), F64::from_device(w + L), F64::from_device(w + 2 * L), F64::from_device(w + 3 * L)};
    }
    static void affine_to_canonical(const Affine<HostTail64>& a, uint64_t* out) {
        uint32_t w[2 * F64::LIMBS];
        F64::to_canonical(w, a.x);
        F64::to_canonical(w + F64::LIMBS, a.y);
        memcpy(out, w, sizeof(w));
    }
};

}  // namespace zkmi
