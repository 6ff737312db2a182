// field.cuh -- prime-field arithmetic for the zksnake hot path on gfx950 (and the host side of
// the same library).  32-bit limbs, Montgomery form, R = 2^(32 N).
//
// Replaces what the reference gets from ark-ff 0.4.2 `Fp<MontBackend<..>, N>` behind
// src/bn254/{curve,polynomial}.rs and src/bls12_381/{curve,polynomial}.rs (the crates are not
// vendored in /root/reference; the algorithm here is textbook CIOS Montgomery multiplication).
//
// CDNA4 notes: the only wide integer multiplier is v_mad_u64_u32 (32x32+64 -> 64); every
// partial product below is written as `(uint64_t)a * b + c` so hipcc selects it.  All loops
// are fully unrolled over compile-time limb counts and the modulus limbs are constexpr, so
// they become literals / SGPRs instead of VGPRs.
#pragma once
#include <cstdint>
#include "field_params.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#if defined(ZK_NOINLINE_MUL)
#define ZK_MUL __host__ __device__ __attribute__((noinline))
#else
#define ZK_MUL ZK_HD
#endif
#else
#define ZK_HD inline
#define ZK_D inline
#define ZK_MUL inline
#endif

namespace zkmi {

template <class P>
struct Fp {
    static constexpr int N = P::N;
    typedef P Params;
    uint32_t v[P::N];
};

// ---- helpers --------------------------------------------------------------------------

template <class P>
ZK_HD Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = 0;
    return r;
}

template <class P>
ZK_HD Fp<P> fp_one() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = P::ONE[i];
    return r;
}

template <class P>
ZK_HD Fp<P> fp_const(const uint32_t (&c)[P::N]) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = c[i];
    return r;
}

template <class P>
ZK_HD bool fp_is_zero(const Fp<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) o |= a.v[i];
    return o == 0;
}

template <class P>
ZK_HD bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = a - p, returns the final borrow (1 when a < p)
template <class P>
ZK_HD uint32_t fp_sub_mod_raw(uint32_t* r, const uint32_t* a) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t t = (uint64_t)a[i] - P::MOD[i] - borrow;
        r[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return (uint32_t)borrow;
}

// conditional final subtraction: a in [0, 2p) -> [0, p)
template <class P>
ZK_HD void fp_reduce_once(Fp<P>& a) {
    uint32_t t[P::N];
    uint32_t borrow = fp_sub_mod_raw<P>(t, a.v);
#pragma unroll
    for (int i = 0; i < P::N; ++i) a.v[i] = borrow ? a.v[i] : t[i];
}

template <class P>
ZK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + carry;
        r.v[i] = (uint32_t)t;
        carry = t >> 32;
    }
    // all four moduli leave at least one spare bit in the top limb, so a + b < 2^(32N)
    fp_reduce_once<P>(r);
    return r;
}

template <class P>
ZK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t t = (uint64_t)r.v[i] + (P::MOD[i] & mask) + carry;
        r.v[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P>& a) {
    Fp<P> r;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) nz |= a.v[i];
    uint32_t mask = nz ? 0xFFFFFFFFu : 0u;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t t = (uint64_t)(P::MOD[i] & mask) - a.v[i] - borrow;
        r.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return r;
}

template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P>& a) {
    return fp_add<P>(a, a);
}

// Montgomery product a*b*R^-1 mod p, inputs and output in [0, p).
// Finely-integrated operand scanning: for each limb of b one pass adds a*b_i and m*p
// together, so the accumulator never needs more than N+1 limbs.
template <class P>
ZK_MUL Fp<P> fp_mul(const Fp<P> a, const Fp<P> b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t bi = b.v[i];
        uint64_t s = (uint64_t)a.v[0] * bi + t[0];
        const uint32_t m = (uint32_t)s * P::INV;
        uint64_t r = (uint64_t)m * P::MOD[0] + (uint32_t)s;
        uint32_t c1 = (uint32_t)(s >> 32);
        uint32_t c2 = (uint32_t)(r >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            s = (uint64_t)a.v[j] * bi + t[j] + c1;
            c1 = (uint32_t)(s >> 32);
            r = (uint64_t)m * P::MOD[j] + (uint32_t)s + c2;
            c2 = (uint32_t)(r >> 32);
            t[j - 1] = (uint32_t)r;
        }
        uint64_t z = (uint64_t)t[N] + c1 + c2;
        t[N - 1] = (uint32_t)z;
        t[N] = (uint32_t)(z >> 32);
    }
    // t < 2p < 2^(32N): t[N] is zero here
    Fp<P> out;
#pragma unroll
    for (int i = 0; i < N; ++i) out.v[i] = t[i];
    fp_reduce_once<P>(out);
    return out;
}

template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P>& a) {
    return fp_mul<P>(a, a);
}

// canonical integer (< 2^(32N), any value) -> Montgomery form
template <class P>
ZK_HD Fp<P> fp_from_canonical(const uint32_t* a) {
    Fp<P> x;
#pragma unroll
    for (int i = 0; i < P::N; ++i) x.v[i] = a[i];
    // inputs may exceed p (the reference reduces on entry: Fr::from(BigUint)); 2^(32N) < 8p for
    // every field here except BLS Fr/Fq (< 3p / < 10p): a short subtract loop is enough.
    for (int k = 0; k < 10; ++k) {
        uint32_t t[P::N];
        uint32_t borrow = fp_sub_mod_raw<P>(t, x.v);
        if (borrow) break;
#pragma unroll
        for (int i = 0; i < P::N; ++i) x.v[i] = t[i];
    }
    return fp_mul<P>(x, fp_const<P>(P::R2));
}

template <class P>
ZK_HD void fp_to_canonical(uint32_t* out, const Fp<P>& a) {
    Fp<P> o = fp_zero<P>();
    o.v[0] = 1;
    Fp<P> r = fp_mul<P>(a, o);
#pragma unroll
    for (int i = 0; i < P::N; ++i) out[i] = r.v[i];
}

// a^e, e as little-endian 32-bit limbs (public exponent; not constant time)
template <class P>
ZK_HD Fp<P> fp_pow(const Fp<P>& a, const uint32_t* e, int nlimbs) {
    Fp<P> acc = fp_one<P>();
    bool started = false;
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        if (started) acc = fp_sqr<P>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fp_mul<P>(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

template <class P>
ZK_HD Fp<P> fp_inv(const Fp<P>& a) {
    return fp_pow<P>(a, P::PM2, P::N);
}

// lexicographic "a > (p-1)/2" on the canonical value (the sign flag of the compressed codecs)
template <class P>
ZK_HD bool fp_canonical_gt_half(const uint32_t* c) {
    for (int i = P::N - 1; i >= 0; --i) {
        if (c[i] != P::HALF[i]) return c[i] > P::HALF[i];
    }
    return false;
}

// ---- quadratic extension Fp[u]/(u^2 + 1) (both curves use this tower for G2) ------------

template <class P>
struct Fp2 {
    typedef P Params;
    Fp<P> c0, c1;
};

template <class P> ZK_HD Fp2<P> fp2_zero() { return {fp_zero<P>(), fp_zero<P>()}; }
template <class P> ZK_HD Fp2<P> fp2_one() { return {fp_one<P>(), fp_zero<P>()}; }
template <class P> ZK_HD bool fp2_is_zero(const Fp2<P>& a) { return fp_is_zero<P>(a.c0) && fp_is_zero<P>(a.c1); }
template <class P> ZK_HD bool fp2_eq(const Fp2<P>& a, const Fp2<P>& b) { return fp_eq<P>(a.c0, b.c0) && fp_eq<P>(a.c1, b.c1); }
template <class P> ZK_HD Fp2<P> fp2_add(const Fp2<P>& a, const Fp2<P>& b) { return {fp_add<P>(a.c0, b.c0), fp_add<P>(a.c1, b.c1)}; }
template <class P> ZK_HD Fp2<P> fp2_sub(const Fp2<P>& a, const Fp2<P>& b) { return {fp_sub<P>(a.c0, b.c0), fp_sub<P>(a.c1, b.c1)}; }
template <class P> ZK_HD Fp2<P> fp2_neg(const Fp2<P>& a) { return {fp_neg<P>(a.c0), fp_neg<P>(a.c1)}; }
template <class P> ZK_HD Fp2<P> fp2_dbl(const Fp2<P>& a) { return {fp_dbl<P>(a.c0), fp_dbl<P>(a.c1)}; }

template <class P>
ZK_HD Fp2<P> fp2_mul(const Fp2<P>& a, const Fp2<P>& b) {
    Fp<P> t0 = fp_mul<P>(a.c0, b.c0);
    Fp<P> t1 = fp_mul<P>(a.c1, b.c1);
    Fp<P> t2 = fp_mul<P>(fp_add<P>(a.c0, a.c1), fp_add<P>(b.c0, b.c1));
    return {fp_sub<P>(t0, t1), fp_sub<P>(fp_sub<P>(t2, t0), t1)};
}

template <class P>
ZK_HD Fp2<P> fp2_sqr(const Fp2<P>& a) {
    Fp<P> t0 = fp_mul<P>(fp_add<P>(a.c0, a.c1), fp_sub<P>(a.c0, a.c1));
    Fp<P> t1 = fp_mul<P>(a.c0, a.c1);
    return {t0, fp_dbl<P>(t1)};
}

template <class P>
ZK_HD Fp2<P> fp2_inv(const Fp2<P>& a) {
    Fp<P> d = fp_inv<P>(fp_add<P>(fp_sqr<P>(a.c0), fp_sqr<P>(a.c1)));
    return {fp_mul<P>(a.c0, d), fp_neg<P>(fp_mul<P>(a.c1, d))};
}

// ---- uniform "field ops" facades so curve code is generic over Fp / Fp2 -------------------

template <class P>
struct FpOps {
    typedef Fp<P> T;
    typedef P Params;
    static constexpr int LIMBS = P::N;  // 32-bit words per element
    static ZK_HD T zero() { return fp_zero<P>(); }
    static ZK_HD T one() { return fp_one<P>(); }
    static ZK_HD T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    static ZK_HD T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    static ZK_HD T mul(const T& a, const T& b) { return fp_mul<P>(a, b); }
    static ZK_HD T sqr(const T& a) { return fp_sqr<P>(a); }
    static ZK_HD T neg(const T& a) { return fp_neg<P>(a); }
    static ZK_HD T dbl(const T& a) { return fp_dbl<P>(a); }
    static ZK_HD T inv(const T& a) { return fp_inv<P>(a); }
    static ZK_HD bool is_zero(const T& a) { return fp_is_zero<P>(a); }
    static ZK_HD bool eq(const T& a, const T& b) { return fp_eq<P>(a, b); }
    static ZK_HD T from_canonical(const uint32_t* w) { return fp_from_canonical<P>(w); }
    static ZK_HD void to_canonical(uint32_t* w, const T& a) { fp_to_canonical<P>(w, a); }
};

template <class P>
struct Fp2Ops {
    typedef Fp2<P> T;
    typedef P Params;
    static constexpr int LIMBS = 2 * P::N;
    static ZK_HD T zero() { return fp2_zero<P>(); }
    static ZK_HD T one() { return fp2_one<P>(); }
    static ZK_HD T add(const T& a, const T& b) { return fp2_add<P>(a, b); }
    static ZK_HD T sub(const T& a, const T& b) { return fp2_sub<P>(a, b); }
    static ZK_HD T mul(const T& a, const T& b) { return fp2_mul<P>(a, b); }
    static ZK_HD T sqr(const T& a) { return fp2_sqr<P>(a); }
    static ZK_HD T neg(const T& a) { return fp2_neg<P>(a); }
    static ZK_HD T dbl(const T& a) { return fp2_dbl<P>(a); }
    static ZK_HD T inv(const T& a) { return fp2_inv<P>(a); }
    static ZK_HD bool is_zero(const T& a) { return fp2_is_zero<P>(a); }
    static ZK_HD bool eq(const T& a, const T& b) { return fp2_eq<P>(a, b); }
    static ZK_HD T from_canonical(const uint32_t* w) {
        return {fp_from_canonical<P>(w), fp_from_canonical<P>(w + P::N)};
    }
    static ZK_HD void to_canonical(uint32_t* w, const T& a) {
        fp_to_canonical<P>(w, a.c0);
        fp_to_canonical<P>(w + P::N, a.c1);
    }
};

}  // namespace zkmi
