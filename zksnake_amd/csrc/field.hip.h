// field.hip.h -- prime-field arithmetic for the zksnake hot path on gfx950 (and the host side of the same
// library): 29-bit limbs, Montgomery form with R = 2^(29 N), values kept semi-reduced in [0, 2p).
//
// Replaces what the reference gets from ark-ff 0.4.2 `Fp<MontBackend<..>, N>` behind
// src/bn254/{curve,polynomial}.rs and src/bls12_381/{curve,polynomial}.rs (the crates are not vendored in
// /root/reference; Montgomery multiplication is textbook).
//
// Why 29-bit limbs on CDNA4 (measured, profiles/r01_ubench_valu.log and DESIGN.md section 4):
//   v_mad_u64_u32 (32x32+64 -> 64) costs ~6 cycles per wave, a plain integer op ~4, and there is no
//   carry-in multiply-add.  With full 32-bit limbs every partial product needs two extra add/mov
//   instructions to thread carries (a 254-bit product took ~1900 cycles/wave, only 40 % of it in the
//   128 multiply-adds).  With 29-bit limbs a 64-bit column accumulator absorbs all 2N partial products
//   of a column (2 * 14 * 2^58 < 2^64) with NO carry handling: the product is N^2 + N^2 + N bare
//   v_mad_u64_u32 plus one shift/mask per column -- ~1100 cycles/wave (1.8x), and 2.8x for a lone wave,
//   which is what the latency-bound reduction stages see.
//   The spare bits (R / p >= 70) also remove every conditional subtraction from the product:
//   inputs < 2p give an output < 2p.
//
// Memory / ABI form of an element: W little-endian 32-bit words (8 or 12); `fp_unpack` / `fp_pack` convert.
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

constexpr int LIMB_BITS = 29;
constexpr uint32_t LIMB_MASK = (1u << LIMB_BITS) - 1;

// N limbs of 29 bits (the top limb may be shorter), normalized, value in [0, 2p)
template <class P>
struct Fp {
    static constexpr int N = P::N;
    typedef P Params;
    uint32_t v[P::N];
};

// ---- constants / predicates ---------------------------------------------------------------------

template <class P>
ZK_HD Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = 0;
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
ZK_HD Fp<P> fp_one() {
    return fp_const<P>(P::ONE);
}

// zero mod p: the representative is 0 or p
template <class P>
ZK_HD bool fp_is_zero(const Fp<P>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        z |= a.v[i];
        e |= a.v[i] ^ P::M[i];
    }
    return z == 0 || e == 0;
}

// all limbs zero: the exact representative 0 (not p).  Enough where zero only ever arises by assignment -- the (0, 0)
// infinity sentinel of an affine row, the ZZ = 0 of an empty accumulator -- and a third of the instructions of fp_is_zero
template <class P>
ZK_HD bool fp_is_zero_limbs(const Fp<P>& a) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) z |= a.v[i];
    return z == 0;
}

// ---- addition / subtraction: both candidates (x and x -/+ 2p) are carried through one pass of signed
// carries and the in-range one is selected, so the result is normalized and again in [0, 2p) ------------

template <class P>
ZK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> s, t;
    int32_t cs = 0, ct = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t x = (int32_t)(a.v[i] + b.v[i]);
        int32_t u = x + cs;
        int32_t w = x - (int32_t)P::P2[i] + ct;
        if (i < N - 1) {
            s.v[i] = (uint32_t)u & LIMB_MASK;
            cs = u >> LIMB_BITS;
            t.v[i] = (uint32_t)w & LIMB_MASK;
            ct = w >> LIMB_BITS;
        } else {
            s.v[i] = (uint32_t)u;
            t.v[i] = (uint32_t)w;
        }
    }
    const bool below = (int32_t)t.v[N - 1] < 0;  // a + b < 2p
#pragma unroll
    for (int i = 0; i < N; ++i) s.v[i] = below ? s.v[i] : t.v[i];
    return s;
}

template <class P>
ZK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> s, t;
    int32_t cs = 0, ct = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i];
        int32_t u = x + cs;
        int32_t w = x + (int32_t)P::P2[i] + ct;
        if (i < N - 1) {
            s.v[i] = (uint32_t)u & LIMB_MASK;
            cs = u >> LIMB_BITS;
            t.v[i] = (uint32_t)w & LIMB_MASK;
            ct = w >> LIMB_BITS;
        } else {
            s.v[i] = (uint32_t)u;
            t.v[i] = (uint32_t)w;
        }
    }
    const bool neg = (int32_t)s.v[N - 1] < 0;  // a < b
#pragma unroll
    for (int i = 0; i < N; ++i) s.v[i] = neg ? t.v[i] : s.v[i];
    return s;
}

// a - b + 4p with NO carry propagation and no range selection: limbs stay below 2^31, the value below 6p.
// Only valid as ONE operand of fp_mul (the other one normalized): 9 partial products of 2^31 * 2^29 plus the
// reduction terms still fit the 64-bit column accumulators (used by the NTT butterflies, N = 9).  The constant is
// 4p written with "borrow-proof" limbs (every limb but the top one borrows 2^29 from the limb above), so no limb
// goes negative for any b < 2p.
template <class P>
ZK_HD Fp<P> fp_sub_lazy(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t c = P::P4[i] + (i < P::N - 1 ? (1u << LIMB_BITS) : 0u) - (i > 0 ? 1u : 0u);
        r.v[i] = a.v[i] + c - b.v[i];
    }
    return r;
}

// ---- "relaxed" forms for the MSM inner loop: no range selection, so half the instructions of fp_add / fp_sub.
// Products tolerate operands well above 2p ((a/p)(b/p) <= R/p), so a difference may stay in (0, (K+2)p). ------

// a - b + K p (K = 2 or 4) with the carries propagated (normalized limbs), for a < 2p and 0 <= b < K p: value in (0, (K+2)p)
template <class P, int K>
ZK_HD Fp<P> fp_sub_k(const Fp<P>& a, const Fp<P>& b) {
    static_assert(K == 2 || K == 4 || K == 8, "K p constants exist for K = 2, 4, 8");
    constexpr int N = P::N;
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)(K == 2 ? P::P2[i] : (K == 4 ? P::P4[i] : P::P8[i])) + c;
        if (i < N - 1) {
            r.v[i] = (uint32_t)x & LIMB_MASK;
            c = x >> LIMB_BITS;
        } else {
            r.v[i] = (uint32_t)x;
        }
    }
    return r;
}

// t - 2q brought into [0, 4p) for t in [0, 4p), q in [0, 2p): both candidates (x, x + 4p) in one pass, like fp_sub
template <class P>
ZK_HD Fp<P> fp_sub_twice_sel4(const Fp<P>& t, const Fp<P>& q) {
    constexpr int N = P::N;
    Fp<P> s, u;
    int32_t cs = 0, cu = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t x = (int32_t)t.v[i] - 2 * (int32_t)q.v[i];
        int32_t a = x + cs;
        int32_t b = x + (int32_t)P::P4[i] + cu;
        if (i < N - 1) {
            s.v[i] = (uint32_t)a & LIMB_MASK;
            cs = a >> LIMB_BITS;
            u.v[i] = (uint32_t)b & LIMB_MASK;
            cu = b >> LIMB_BITS;
        } else {
            s.v[i] = (uint32_t)a;
            u.v[i] = (uint32_t)b;
        }
    }
    const bool neg = (int32_t)s.v[N - 1] < 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s.v[i] = neg ? u.v[i] : s.v[i];
    return s;
}

// a - b + 8p, carry-free with borrow-proof limbs (< 3*2^29), for b < 4p... any normalized b below 8p; one product operand only
template <class P>
ZK_HD Fp<P> fp_sub_lazy8(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t c = P::P8[i] + (i < P::N - 1 ? (1u << LIMB_BITS) : 0u) - (i > 0 ? 1u : 0u);
        r.v[i] = a.v[i] + c - b.v[i];
    }
    return r;
}

// 4p - b with the same borrow-proof limbs: a carry-free negation, valid as one operand of a product
template <class P>
ZK_HD Fp<P> fp_neg_lazy(const Fp<P>& b) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t c = P::P4[i] + (i < P::N - 1 ? (1u << LIMB_BITS) : 0u) - (i > 0 ? 1u : 0u);
        r.v[i] = c - b.v[i];
    }
    return r;
}

// K p - b (K = 4, 8) with the same borrow-proof limbs (< 2^30): a carry-free negation for ONE operand of a product.  The top
// limb of K p does not borrow, so b must stay clear of it: b < (K - 1) p.
template <class P, int K>
ZK_HD Fp<P> fp_neg_lazy_k(const Fp<P>& b) {
    static_assert(K == 4 || K == 8, "K p constants exist for K = 4, 8");
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t c = (K == 4 ? P::P4[i] : P::P8[i]) + (i < P::N - 1 ? (1u << LIMB_BITS) : 0u) - (i > 0 ? 1u : 0u);
        r.v[i] = c - b.v[i];
    }
    return r;
}

// a + b with the carries propagated and NO range selection: normalised limbs, the values simply add up
template <class P>
ZK_HD Fp<P> fp_add_nosel(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t x = a.v[i] + b.v[i] + c;
        if (i < N - 1) {
            r.v[i] = x & LIMB_MASK;
            c = x >> LIMB_BITS;
        } else {
            r.v[i] = x;
        }
    }
    return r;
}

// 2a limb by limb (limbs < 2^30): one operand of a single product only
template <class P>
ZK_HD Fp<P> fp_dbl_lazy(const Fp<P>& a) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = a.v[i] << 1;
    return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P>& a) {
    return fp_sub<P>(fp_zero<P>(), a);
}

template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P>& a) {
    return fp_add<P>(a, a);
}

template <class P>
ZK_HD bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    return fp_is_zero<P>(fp_sub<P>(a, b));
}

// ---- Montgomery product a*b*R^-1 mod p: column-wise (product scanning) with the reduction folded into the
// same columns.  Inputs normalized with (a/p)*(b/p) <= R/p (>= 70 for all four fields), output < 2p. -------

template <class P>
ZK_MUL Fp<P> fp_mul(const Fp<P> a, const Fp<P> b) {
    constexpr int N = P::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        m[k] = ((uint32_t)acc * P::INV) & LIMB_MASK;
        acc += (uint64_t)m[k] * P::M[0];
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        r.v[k - N] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
}

// a*b + c*d in ONE Montgomery reduction (the Fp2 product needs exactly this shape).  The column accumulators
// take 3N products of < 2^58: 3 * 14 * 2^58 < 2^64.  Inputs < 2p each: (4p^2 + 4p^2) / R + p < 2p.
template <class P>
ZK_MUL Fp<P> fp_mul2(const Fp<P> a, const Fp<P> b, const Fp<P> c, const Fp<P> d) {
    constexpr int N = P::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        m[k] = ((uint32_t)acc * P::INV) & LIMB_MASK;
        acc += (uint64_t)m[k] * P::M[0];
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
        }
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        r.v[k - N] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
}

// a*b + c*d + e*f + g*h in ONE Montgomery reduction (the Y3 of an Fp2 mixed addition is exactly this shape).  Column room:
// with b, d, f, h normalised, a and g normalised and c, e lazy negations (limbs < 2^30) a column holds
// N (1 + 2 + 2 + 1 + 1) 2^58 = 63 * 2^58 < 2^64 for N = 9 only.  Value: the four products together must stay below R p.
template <class P>
ZK_MUL Fp<P> fp_mul4(const Fp<P> a, const Fp<P> b, const Fp<P> c, const Fp<P> d, const Fp<P> e, const Fp<P> f, const Fp<P> g, const Fp<P> h) {
    constexpr int N = P::N;
    static_assert(N <= 9, "four products and the reduction overflow the 64-bit column accumulators beyond nine limbs");
    uint64_t acc = 0;
    uint32_t m[N];
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
            acc += (uint64_t)e.v[i] * f.v[k - i];
            acc += (uint64_t)g.v[i] * h.v[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        m[k] = ((uint32_t)acc * P::INV) & LIMB_MASK;
        acc += (uint64_t)m[k] * P::M[0];
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)c.v[i] * d.v[k - i];
            acc += (uint64_t)e.v[i] * f.v[k - i];
            acc += (uint64_t)g.v[i] * h.v[k - i];
        }
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)m[i] * P::M[k - i];
        r.v[k - N] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
}

// squaring: the cross terms are computed once against pre-doubled limbs (N(N+1)/2 instead of N^2 products)
template <class P>
ZK_MUL Fp<P> fp_sqr(const Fp<P> a) {
    constexpr int N = P::N;
    uint64_t acc = 0;
    uint32_t m[N], d[N];
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = a.v[i] << 1;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            const int j = k - i;
            if (i < j) acc += (uint64_t)d[i] * a.v[j];
            else if (i == j) acc += (uint64_t)a.v[i] * a.v[i];
        }
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P::M[k - i];
            m[k] = ((uint32_t)acc * P::INV) & LIMB_MASK;
            acc += (uint64_t)m[k] * P::M[0];
        } else {
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)m[i] * P::M[k - i];
            r.v[k - N] = (uint32_t)acc & LIMB_MASK;
        }
        acc >>= LIMB_BITS;
    }
    r.v[N - 1] = (uint32_t)acc;
    return r;
}

// ---- memory form: W 32-bit words <-> N 29-bit limbs ---------------------------------------------------

// any integer below 2^(32 W) -> normalized limbs (no reduction)
template <class P>
ZK_HD Fp<P> fp_unpack(const uint32_t* w) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const int bit = LIMB_BITS * i;
        const int word = bit >> 5, sh = bit & 31;
        uint32_t x = 0;
        if (word < P::W) x = w[word] >> sh;
        if (sh > 32 - LIMB_BITS && word + 1 < P::W) x |= w[word + 1] << (32 - sh);
        r.v[i] = x & LIMB_MASK;
    }
    return r;
}

// normalized limbs of a value below 2^(32 W) -> words
template <class P>
ZK_HD void fp_pack(uint32_t* w, const Fp<P>& a) {
#pragma unroll
    for (int j = 0; j < P::W; ++j) {
        // word j collects bits [32 j, 32 j + 32): limbs whose range overlaps
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) {
            const int lo = LIMB_BITS * i;            // first bit of limb i
            const int rel = lo - 32 * j;             // position of that bit inside word j
            if (rel > -LIMB_BITS - 3 && rel < 32) {
                if (rel >= 0) x |= a.v[i] << rel;
                else if (-rel < 32) x |= a.v[i] >> (-rel);
            }
        }
        w[j] = x;
    }
}

// conditional subtraction of p: [0, 2p) -> [0, p)
template <class P>
ZK_HD Fp<P> fp_reduce_full(const Fp<P>& a) {
    constexpr int N = P::N;
    Fp<P> t;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t w = (int32_t)a.v[i] - (int32_t)P::M[i] + c;
        if (i < N - 1) {
            t.v[i] = (uint32_t)w & LIMB_MASK;
            c = w >> LIMB_BITS;
        } else {
            t.v[i] = (uint32_t)w;
        }
    }
    const bool below = (int32_t)t.v[N - 1] < 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t.v[i] = below ? a.v[i] : t.v[i];
    return t;
}

// conditional subtraction of 2p: [0, 4p) -> [0, 2p)
template <class P>
ZK_HD Fp<P> fp_reduce_2p(const Fp<P>& a) {
    constexpr int N = P::N;
    Fp<P> t;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int32_t w = (int32_t)a.v[i] - (int32_t)P::P2[i] + c;
        if (i < N - 1) {
            t.v[i] = (uint32_t)w & LIMB_MASK;
            c = w >> LIMB_BITS;
        } else {
            t.v[i] = (uint32_t)w;
        }
    }
    const bool below = (int32_t)t.v[N - 1] < 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t.v[i] = below ? a.v[i] : t.v[i];
    return t;
}

// canonical integer (any value below 2^(32 W): reduced like Fr::from(BigUint)) -> Montgomery form
template <class P>
ZK_HD Fp<P> fp_from_canonical(const uint32_t* w) {
    return fp_mul<P>(fp_unpack<P>(w), fp_const<P>(P::R2));
}

// Montgomery form -> canonical integer in [0, p) as W words
template <class P>
ZK_HD void fp_to_canonical(uint32_t* w, const Fp<P>& a) {
    Fp<P> o = fp_zero<P>();
    o.v[0] = 1;
    fp_pack<P>(w, fp_reduce_full<P>(fp_mul<P>(a, o)));
}

// storage of Montgomery-form values (semi-reduced, < 2p < 2^(32 W)) in HBM / host buffers
template <class P>
ZK_HD Fp<P> fp_load(const uint32_t* w) {
    return fp_unpack<P>(w);
}
template <class P>
ZK_HD void fp_store(uint32_t* w, const Fp<P>& a) {
    fp_pack<P>(w, a);
}

// ---- helpers on canonical 32-bit words ------------------------------------------------------------------

// r = a - p on W-word integers, returns the final borrow (1 when a < p)
template <class P>
ZK_HD uint32_t fp_sub_mod_raw(uint32_t* r, const uint32_t* a) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::W; ++i) {
        uint64_t t = (uint64_t)a[i] - P::MOD[i] - borrow;
        r[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return (uint32_t)borrow;
}

// lexicographic "a > (p-1)/2" on the canonical value (the sign flag of the compressed codecs)
template <class P>
ZK_HD bool fp_canonical_gt_half(const uint32_t* c) {
    for (int i = P::W - 1; i >= 0; --i) {
        if (c[i] != P::HALF[i]) return c[i] > P::HALF[i];
    }
    return false;
}

// a^e, e as little-endian 32-bit words (public exponent; not constant time)
template <class P>
ZK_HD Fp<P> fp_pow(const Fp<P>& a, const uint32_t* e, int nwords) {
    Fp<P> acc = fp_one<P>();
    bool started = false;
    for (int i = nwords * 32 - 1; i >= 0; --i) {
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
    return fp_pow<P>(a, P::PM2, P::W);
}

// ---- quadratic extension Fp[u]/(u^2 + 1) (both curves use this tower for G2) ----------------------------

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

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u.  Two double products with one reduction each
// (3N^2 + N multiply-adds apiece) and a single negation: cheaper here than Karatsuba's three products plus five
// additions, because an addition costs ~1/4 of a product on this machine.
template <class P>
ZK_HD Fp2<P> fp2_mul(const Fp2<P>& a, const Fp2<P>& b) {
    // -a1 as the carry-free 4p - a1 (limbs < 2*2^29): the columns of fp_mul2 then hold N*(1 + 2 + 1)*2^58 < 2^64
    // for N <= 14, and (2p*2p + 4p*2p)/R + p < 2p
    Fp<P> na1 = fp_neg_lazy<P>(a.c1);
    return {fp_mul2<P>(a.c0, b.c0, na1, b.c1), fp_mul2<P>(a.c0, b.c1, a.c1, b.c0)};
}

template <class P>
ZK_HD Fp2<P> fp2_sqr(const Fp2<P>& a) {
    Fp<P> t0 = fp_mul<P>(fp_add<P>(a.c0, a.c1), fp_sub<P>(a.c0, a.c1));
    Fp<P> t1 = fp_mul<P>(a.c0, a.c1);
    return {t0, fp_dbl<P>(t1)};
}

// ---- relaxed-range Fp2 pieces of the G2 bucket accumulation (curve.hip.h; bounds replayed by tools/model_relaxed_g2.py) ----
// product with a.c1 negated lazily against K p (a.c1 < (K - 1) p): the components of a may exceed 2p as far as the value
// bound of fp_mul2 allows, (a0 b0 + K p b1) <= R p and (a0 b1 + a1 b0) <= R p
template <class P, int K>
ZK_HD Fp2<P> fp2_mul_relaxed(const Fp2<P>& a, const Fp2<P>& b) {
    const Fp<P> na1 = fp_neg_lazy_k<P, K>(a.c1);
    return {fp_mul2<P>(a.c0, b.c0, na1, b.c1), fp_mul2<P>(a.c0, b.c1, a.c1, b.c0)};
}
// square with components below KD p / 2 ... precisely: ((a0 + a1)(a0 - a1 + KD p), (2 a0) a1), a1 < KD p; no range selection,
// both results below 2p (the products reduce)
template <class P, int KD>
ZK_HD Fp2<P> fp2_sqr_relaxed(const Fp2<P>& a) {
    const Fp<P> ts = fp_add_nosel<P>(a.c0, a.c1);
    const Fp<P> td = fp_sub_k<P, KD>(a.c0, a.c1);
    return {fp_mul<P>(ts, td), fp_mul<P>(fp_dbl_lazy<P>(a.c0), a.c1)};
}

template <class P>
ZK_HD Fp2<P> fp2_inv(const Fp2<P>& a) {
    Fp<P> d = fp_inv<P>(fp_add<P>(fp_sqr<P>(a.c0), fp_sqr<P>(a.c1)));
    return {fp_mul<P>(a.c0, d), fp_neg<P>(fp_mul<P>(a.c1, d))};
}

// ---- uniform "field ops" facades so curve code is generic over Fp / Fp2 ---------------------------------
// LIMBS = 32-bit words per element in memory; REGS = 32-bit registers per element.

template <class P>
struct FpOps {
    typedef Fp<P> T;
    typedef P Params;
    static constexpr int LIMBS = P::W;
    static constexpr int REGS = P::N;
    static ZK_HD T zero() { return fp_zero<P>(); }
    static ZK_HD T one() { return fp_one<P>(); }
    static ZK_HD T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    static ZK_HD T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    // a - b for immediate use as ONE operand of mul(): carry-free when the column accumulators have the room
    // (N * 2^31 * 2^29 + N * 2^58 < 2^64, i.e. N <= 12), the ordinary subtraction otherwise
    static ZK_HD T sub_for_mul(const T& a, const T& b) {
        if constexpr (P::N <= 12) return fp_sub_lazy<P>(a, b);
        else return fp_sub<P>(a, b);
    }
    // m * (s - x) - w * y with ONE Montgomery reduction (the Y3 of every addition / doubling formula).  With limbs
    // < 3*2^29 for the lazy difference and < 2*2^29 for the lazy negation a column holds N*(3+2+1)*2^58 < 2^64 for
    // N <= 10, so both stay carry-free; wider fields use ordinary operands (3N products fit up to N = 14).
    // Value: (2p*6p + 4p*2p)/R + p < 2p since R/p >= 70.
    static ZK_HD T mul_diff(const T& m, const T& s, const T& x, const T& w, const T& y) {
        if constexpr (P::N <= 10) return fp_mul2<P>(m, fp_sub_lazy<P>(s, x), fp_neg_lazy<P>(w), y);
        else return fp_mul2<P>(m, fp_sub<P>(s, x), fp_neg<P>(w), y);
    }
    static ZK_HD T mul(const T& a, const T& b) { return fp_mul<P>(a, b); }
    static ZK_HD T sqr(const T& a) { return fp_sqr<P>(a); }
    static ZK_HD T neg(const T& a) { return fp_neg<P>(a); }
    static ZK_HD T dbl(const T& a) { return fp_dbl<P>(a); }
    static ZK_HD T inv(const T& a) { return fp_inv<P>(a); }
    static ZK_HD bool is_zero(const T& a) { return fp_is_zero<P>(a); }
    static ZK_HD bool is_zero_limbs(const T& a) { return fp_is_zero_limbs<P>(a); }
    static ZK_HD bool eq(const T& a, const T& b) { return fp_eq<P>(a, b); }
    static ZK_HD T from_canonical(const uint32_t* w) { return fp_from_canonical<P>(w); }
    static ZK_HD void to_canonical(uint32_t* w, const T& a) { fp_to_canonical<P>(w, a); }
    static ZK_HD T load(const uint32_t* w) { return fp_load<P>(w); }
    static ZK_HD void store(uint32_t* w, const T& a) { fp_store<P>(w, a); }

    // ---- relaxed-range pieces of the bucket-accumulation step (curve.hip.h, xyzz_add_affine_mem) ----
    static constexpr bool RELAXED = true;
    static constexpr bool RELAXED2 = false;
    static ZK_HD T neg_for_mul(const T& a) { return fp_neg_lazy<P>(a); }                 // 4p - a, carry-free
    template <int K> static ZK_HD T sub_k(const T& a, const T& b) { return fp_sub_k<P, K>(a, b); }
    static ZK_HD T x3_sel4(const T& t, const T& q) { return fp_sub_twice_sel4<P>(t, q); }
    // r * (q - x3) - y * ppp with r < 4p, x3 < 4p, the rest < 2p; one reduction.
    // N <= 10: q - x3 + 8p and 4p - y carry-free: columns N (3 + 2 + 1) 2^58 < 2^64, value (4p 10p + 4p 2p)/R + p < 2p.
    // wider: q - x3 + 4p normalized, 4p - y carry-free: columns N (1 + 2 + 1) 2^58 < 2^64 for N <= 14.
    static ZK_HD T y3_relaxed(const T& r, const T& q, const T& x3, const T& y, const T& ppp) {
        if constexpr (P::N <= 10) return fp_mul2<P>(r, fp_sub_lazy8<P>(q, x3), fp_neg_lazy<P>(y), ppp);
        else return fp_mul2<P>(r, fp_sub_k<P, 4>(q, x3), fp_neg_lazy<P>(y), ppp);
    }
};

template <class P>
struct Fp2Ops {
    typedef Fp2<P> T;
    typedef P Params;
    static constexpr int LIMBS = 2 * P::W;
    static constexpr int REGS = 2 * P::N;
    static ZK_HD T zero() { return fp2_zero<P>(); }
    static ZK_HD T one() { return fp2_one<P>(); }
    static ZK_HD T add(const T& a, const T& b) { return fp2_add<P>(a, b); }
    static ZK_HD T sub(const T& a, const T& b) { return fp2_sub<P>(a, b); }
    static constexpr bool RELAXED = false;
    static constexpr bool RELAXED2 = true;   // Fp2 form of the relaxed bucket-accumulation step (curve.hip.h) and of pair_add (pair.hip.h)
    template <int K> static ZK_HD T sub_k(const T& a, const T& b) { return {fp_sub_k<P, K>(a.c0, b.c0), fp_sub_k<P, K>(a.c1, b.c1)}; }
    static ZK_HD T x3_sel4(const T& t, const T& q) { return {fp_sub_twice_sel4<P>(t.c0, q.c0), fp_sub_twice_sel4<P>(t.c1, q.c1)}; }
    template <int K> static ZK_HD T mul_rel(const T& a, const T& b) { return fp2_mul_relaxed<P, K>(a, b); }   // a.c1 < (K - 1) p
    template <int KD> static ZK_HD T sqr_rel(const T& a) { return fp2_sqr_relaxed<P, KD>(a); }                // a.c1 < KD p
    static ZK_HD T reduce_2p(const T& a) { return {fp_reduce_2p<P>(a.c0), fp_reduce_2p<P>(a.c1)}; }
    static ZK_HD T sub_for_mul(const T& a, const T& b) { return fp2_sub<P>(a, b); }  // fp_mul2 has no spare room
    static ZK_HD T mul_diff(const T& m, const T& s, const T& x, const T& w, const T& y) {
        return fp2_sub<P>(fp2_mul<P>(m, fp2_sub<P>(s, x)), fp2_mul<P>(w, y));
    }
    static ZK_HD T mul(const T& a, const T& b) { return fp2_mul<P>(a, b); }
    static ZK_HD T sqr(const T& a) { return fp2_sqr<P>(a); }
    static ZK_HD T neg(const T& a) { return fp2_neg<P>(a); }
    static ZK_HD T dbl(const T& a) { return fp2_dbl<P>(a); }
    static ZK_HD T inv(const T& a) { return fp2_inv<P>(a); }
    static ZK_HD bool is_zero(const T& a) { return fp2_is_zero<P>(a); }
    static ZK_HD bool is_zero_limbs(const T& a) { return fp_is_zero_limbs<P>(a.c0) && fp_is_zero_limbs<P>(a.c1); }
    static ZK_HD bool eq(const T& a, const T& b) { return fp2_eq<P>(a, b); }
    static ZK_HD T from_canonical(const uint32_t* w) {
        return {fp_from_canonical<P>(w), fp_from_canonical<P>(w + P::W)};
    }
    static ZK_HD void to_canonical(uint32_t* w, const T& a) {
        fp_to_canonical<P>(w, a.c0);
        fp_to_canonical<P>(w + P::W, a.c1);
    }
    static ZK_HD T load(const uint32_t* w) { return {fp_load<P>(w), fp_load<P>(w + P::W)}; }
    static ZK_HD void store(uint32_t* w, const T& a) {
        fp_store<P>(w, a.c0);
        fp_store<P>(w + P::W, a.c1);
    }
};

}  // namespace zkmi
