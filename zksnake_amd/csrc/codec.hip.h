// codec.hip.h -- compressed point encodings, host and device (one routine per point; the batched kernels below put one
// point on a lane).
//
// Formats as the reference's to_bytes / from_hex produce and accept them (src/bn254/curve.rs:283-324 through
// ark-serialize 0.4.2 `serialize_compressed` / `deserialize_compressed`; src/bls12_381/curve.rs twins use the zcash
// layout of ark-bls12-381), SURVEY Appendix A:
//   BN254      little-endian x (c0 first for Fp2), top bits of the LAST byte: 0x80 = y is the larger root, 0x40 = infinity
//   BLS12-381  big-endian x (c1 first for Fp2), top bits of the FIRST byte: 0x80 compressed, 0x40 infinity, 0x20 larger y
// Decoding validates what ark validates: flags, x < p, x on the curve, and membership of the prime-order subgroup.
// Key files hold millions of such points (serialization.py:60-141), hence the kernels: a square root is ~380 field
// products and the subgroup check ~3500, all independent per point.
#pragma once
#include "common.hip.h"
#include "curve_consts.h"

namespace zkmi {

template <class G> struct CurveConsts;
template <> struct CurveConsts<Bn254G1> { static ZK_HD const uint32_t* gen() { return Bn254Consts::G1_GEN; } static ZK_HD const uint32_t* b() { return Bn254Consts::G1_B; } };
template <> struct CurveConsts<Bn254G2> { static ZK_HD const uint32_t* gen() { return Bn254Consts::G2_GEN; } static ZK_HD const uint32_t* b() { return Bn254Consts::G2_B; } };
template <> struct CurveConsts<Bls381G1> { static ZK_HD const uint32_t* gen() { return Bls381Consts::G1_GEN; } static ZK_HD const uint32_t* b() { return Bls381Consts::G1_B; } };
template <> struct CurveConsts<Bls381G2> { static ZK_HD const uint32_t* gen() { return Bls381Consts::G2_GEN; } static ZK_HD const uint32_t* b() { return Bls381Consts::G2_B; } };

enum CodecStatus {
    CODEC_OK = 0,
    CODEC_NOT_ON_CURVE,      // encode
    CODEC_UNCOMPRESSED,      // decode ...
    CODEC_BAD_FLAGS,
    CODEC_INF_NONZERO_X,
    CODEC_X_RANGE,
    CODEC_X_NOT_ON_CURVE,
    CODEC_NOT_IN_SUBGROUP,
};

inline const char* codec_message(int code) {
    switch (code) {
        case CODEC_NOT_ON_CURVE: return "point is not on the curve";
        case CODEC_UNCOMPRESSED: return "Cannot deserialize point: uncompressed encoding";
        case CODEC_BAD_FLAGS: return "Cannot deserialize point: invalid flags";
        case CODEC_INF_NONZERO_X: return "Cannot deserialize point: non-zero x with the infinity flag";
        case CODEC_X_RANGE: return "Cannot deserialize point: x is not a field element";
        case CODEC_X_NOT_ON_CURVE: return "Cannot deserialize point: x is not on the curve";
        case CODEC_NOT_IN_SUBGROUP: return "Cannot deserialize point: not in the prime-order subgroup";
    }
    return "ok";
}

// ---- square roots (both base fields have p = 3 mod 4) ---------------------------------------------

template <class P>
ZK_HD bool fp_sqrt(const Fp<P>& a, Fp<P>* out) {
    Fp<P> s = fp_pow<P>(a, P::SQRT_E, P::W);
    if (!fp_eq<P>(fp_sqr<P>(s), a)) return false;
    *out = s;
    return true;
}

// complex method: with alpha = sqrt(norm a), x0^2 = (a0 +- alpha) / 2 and x1 = a1 / (2 x0)
template <class P>
ZK_HD bool fp2_sqrt(const Fp2<P>& a, Fp2<P>* out) {
    if (fp2_is_zero<P>(a)) { *out = a; return true; }
    Fp<P> s;
    if (fp_is_zero<P>(a.c1)) {
        if (fp_sqrt<P>(a.c0, &s)) { *out = {s, fp_zero<P>()}; return true; }
        if (fp_sqrt<P>(fp_neg<P>(a.c0), &s)) { *out = {fp_zero<P>(), s}; return true; }
        return false;
    }
    Fp<P> norm = fp_add<P>(fp_sqr<P>(a.c0), fp_sqr<P>(a.c1));
    Fp<P> alpha;
    if (!fp_sqrt<P>(norm, &alpha)) return false;
    Fp<P> two = fp_dbl<P>(fp_one<P>());
    Fp<P> inv2 = fp_inv<P>(two);
    Fp<P> delta = fp_mul<P>(fp_add<P>(a.c0, alpha), inv2);
    Fp<P> x0;
    if (!fp_sqrt<P>(delta, &x0)) {
        delta = fp_mul<P>(fp_sub<P>(a.c0, alpha), inv2);
        if (!fp_sqrt<P>(delta, &x0)) return false;
    }
    Fp<P> x1 = fp_mul<P>(a.c1, fp_inv<P>(fp_dbl<P>(x0)));
    Fp2<P> cand = {x0, x1};
    if (!fp2_eq<P>(fp2_sqr<P>(cand), a)) return false;
    *out = cand;
    return true;
}

template <class P> ZK_HD bool coord_sqrt(const Fp<P>& a, Fp<P>* o) { return fp_sqrt<P>(a, o); }
template <class P> ZK_HD bool coord_sqrt(const Fp2<P>& a, Fp2<P>* o) { return fp2_sqrt<P>(a, o); }

// "y is the lexicographically larger of {y, -y}" (ark: y > -y; Fp2 compares c1 first, then c0)
template <class P>
ZK_HD bool coord_is_larger(const Fp<P>& y) {
    uint32_t c[P::W];
    fp_to_canonical<P>(c, y);
    return fp_canonical_gt_half<P>(c);
}
template <class P>
ZK_HD bool coord_is_larger(const Fp2<P>& y) {
    if (!fp_is_zero<P>(y.c1)) return coord_is_larger<P>(y.c1);
    return coord_is_larger<P>(y.c0);
}

ZK_HD void words_to_bytes(uint8_t* dst, const uint32_t* w, int nbytes, bool big_endian) {
    for (int i = 0; i < nbytes; ++i) {
        uint8_t b = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        dst[big_endian ? nbytes - 1 - i : i] = b;
    }
}
ZK_HD void bytes_to_words(uint32_t* w, int nwords, const uint8_t* src, int nbytes, bool big_endian) {
    for (int i = 0; i < nwords; ++i) w[i] = 0;
    for (int i = 0; i < nbytes; ++i) {
        uint8_t b = src[big_endian ? nbytes - 1 - i : i];
        w[i >> 2] |= (uint32_t)b << (8 * (i & 3));
    }
}
template <class P>
ZK_HD bool canonical_lt_mod(const uint32_t* c) {
    uint32_t t[P::W];
    return fp_sub_mod_raw<P>(t, c) != 0;
}

template <class G>
struct CodecLayout {
    typedef typename G::F F;
    typedef typename F::Params P;
    static constexpr bool BLS = G::CURVE == ZK_CURVE_BLS12_381;
    static constexpr int FB = BLS ? 48 : 32;        // bytes per base-field element
    static constexpr int COMPS = F::LIMBS / P::W;   // 1 (G1) or 2 (G2)
    static constexpr int TOTAL = FB * COMPS;
    // BN254 G1 has cofactor 1: every point of the curve is in the subgroup
    static constexpr bool SUBGROUP_CHECK = !(G::CURVE == ZK_CURVE_BN254 && G::GROUP == ZK_G1);
};

// a = canonical affine words (x | y, all zero = infinity) -> out[TOTAL]
template <class G>
ZK_HD int point_encode(const uint32_t* a, uint8_t* out) {
    typedef typename G::F F;
    typedef typename F::Params P;
    typedef CodecLayout<G> L;
    Affine<F> p = {F::from_canonical(a), F::from_canonical(a + F::LIMBS)};
    for (int i = 0; i < L::TOTAL; ++i) out[i] = 0;
    if (aff_is_inf<F>(p)) {
        if (L::BLS) out[0] = 0xC0; else out[L::TOTAL - 1] = 0x40;
        return CODEC_OK;
    }
    if (!aff_on_curve<F>(p, F::from_canonical(CurveConsts<G>::b()))) return CODEC_NOT_ON_CURVE;
    bool larger = coord_is_larger(p.y);
    if (L::BLS) {
        for (int k = 0; k < L::COMPS; ++k) words_to_bytes(out + k * L::FB, a + (L::COMPS - 1 - k) * P::W, L::FB, true);
        out[0] |= 0x80;
        if (larger) out[0] |= 0x20;
    } else {
        for (int k = 0; k < L::COMPS; ++k) words_to_bytes(out + k * L::FB, a + k * P::W, L::FB, false);
        if (larger) out[L::TOTAL - 1] |= 0x80;
    }
    return CODEC_OK;
}

// in[TOTAL] -> out = canonical affine words
template <class G>
ZK_HD int point_decode(const uint8_t* in, uint32_t* out) {
    typedef typename G::F F;
    typedef typename F::Params P;
    typedef typename G::Fr FrP;
    typedef CodecLayout<G> L;
    uint8_t buf[L::TOTAL];
    for (int i = 0; i < L::TOTAL; ++i) buf[i] = in[i];
    bool inf, larger;
    if (L::BLS) {
        uint8_t flags = buf[0] & 0xE0;
        buf[0] &= 0x1F;
        if (!(flags & 0x80)) return CODEC_UNCOMPRESSED;
        inf = flags & 0x40;
        larger = flags & 0x20;
    } else {
        uint8_t flags = buf[L::TOTAL - 1] & 0xC0;
        buf[L::TOTAL - 1] &= 0x3F;
        if (flags == 0xC0) return CODEC_BAD_FLAGS;
        inf = flags & 0x40;
        larger = flags & 0x80;
    }
    uint32_t xw[F::LIMBS];
    for (int k = 0; k < L::COMPS; ++k) {
        if (L::BLS) bytes_to_words(xw + (L::COMPS - 1 - k) * P::W, P::W, buf + k * L::FB, L::FB, true);
        else bytes_to_words(xw + k * P::W, P::W, buf + k * L::FB, L::FB, false);
    }
    if (inf) {
        for (int i = 0; i < F::LIMBS; ++i)
            if (xw[i]) return CODEC_INF_NONZERO_X;
        if (larger) return CODEC_BAD_FLAGS;
        for (int i = 0; i < 2 * F::LIMBS; ++i) out[i] = 0;
        return CODEC_OK;
    }
    for (int k = 0; k < L::COMPS; ++k)
        if (!canonical_lt_mod<P>(xw + k * P::W)) return CODEC_X_RANGE;
    typename F::T x = F::from_canonical(xw);
    typename F::T rhs = F::add(F::mul(F::sqr(x), x), F::from_canonical(CurveConsts<G>::b()));
    typename F::T y;
    if (!coord_sqrt(rhs, &y)) return CODEC_X_NOT_ON_CURVE;
    if (coord_is_larger(y) != larger) y = F::neg(y);
    if (L::SUBGROUP_CHECK) {
        Affine<F> p = {x, y};
        if (!xyzz_is_inf<F>(xyzz_scalar_mul<F>(p, FrP::MOD, FrP::W))) return CODEC_NOT_IN_SUBGROUP;
    }
    F::to_canonical(out, x);
    F::to_canonical(out + F::LIMBS, y);
    return CODEC_OK;
}

// ---- batched: one point per lane --------------------------------------------------------------------------------
// first_error: min over failing points of (index << 8 | code), ~0 when all passed
template <class G>
__global__ __launch_bounds__(128) void points_decode_kernel(const uint8_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out,
                                                            unsigned long long* __restrict__ first_error) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int code = point_decode<G>(in + i * CodecLayout<G>::TOTAL, out + i * 2 * G::F::LIMBS);
    if (code) atomicMin(first_error, (unsigned long long)((i << 8) | (uint64_t)code));
}

template <class G>
__global__ __launch_bounds__(128) void points_encode_kernel(const uint32_t* __restrict__ in, uint64_t n, uint8_t* __restrict__ out,
                                                            unsigned long long* __restrict__ first_error) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int code = point_encode<G>(in + i * 2 * G::F::LIMBS, out + i * CodecLayout<G>::TOTAL);
    if (code) atomicMin(first_error, (unsigned long long)((i << 8) | (uint64_t)code));
}

#define ZK_CODEC_EXTERN_TEMPLATES(G)                                                                              \
    extern template __global__ void points_decode_kernel<G>(const uint8_t*, uint64_t, uint32_t*, unsigned long long*); \
    extern template __global__ void points_encode_kernel<G>(const uint32_t*, uint64_t, uint8_t*, unsigned long long*);
#define ZK_CODEC_INSTANTIATE(G)                                                                            \
    template __global__ void points_decode_kernel<G>(const uint8_t*, uint64_t, uint32_t*, unsigned long long*); \
    template __global__ void points_encode_kernel<G>(const uint32_t*, uint64_t, uint8_t*, unsigned long long*);

}  // namespace zkmi
