// pair.hip.h -- one XYZZ point held by a LANE PAIR: the even lane keeps (X, ZZ), the odd lane (Y, ZZZ).
//
// The bucket-reduction stages of the MSM (msm_impl.hip.h stages 6-7) are latency-bound: few points, long chains of
// dependent additions, most SIMDs idle.  A single lane needs the 12M + 2S of add-2008-s one after the other (~9 us per
// addition for BN254 G1 at lone-wave speed).  Split over two lanes the same addition is SEVEN multiplications deep,
// both lanes run the same instruction stream (only the operands differ), and no lane is wasted:
//
//        even lane (X, ZZ)                    odd lane (Y, ZZZ)
//   1    U1  = X1 ZZ2                         S1    = Y1 ZZZ2
//   2    U2  = X2 ZZ1                         S2    = Y2 ZZZ1
//        P   = U2 - U1                        R     = S2 - S1
//   3    PP  = P^2                            RR    = R^2
//   4    PPP = P PP                           ZZZ12 = ZZZ1 ZZZ2            exchange: RR -> even, PPP -> odd
//   5    Q   = U1 PP                          ZZZ3  = ZZZ12 PPP
//        X3  = RR - PPP - 2Q                                               exchange: Q - X3 -> odd
//   6    ZZ12 = ZZ1 ZZ2                       t     = R (Q - X3)
//   7    ZZ3 = ZZ12 PP                        Y3    = t - S1 PPP
//
// Three coordinate-sized exchanges per addition travel through DPP quad permutes (no LDS).  Replaces nothing in the
// reference (ark's MSM reduces its buckets with a serial running sum on one core, src/bn254/curve.rs:356-373 ->
// ark-ec VariableBaseMSM); this is how the same sum is laid out for 64-wide waves.
#pragma once
#include "curve.hip.h"

namespace zkmi {

template <class F>
struct HalfPt {
    typename F::T a, b;  // even lane: (X, ZZ); odd lane: (Y, ZZZ)
};

template <class T>
struct RegCount { static constexpr int N = sizeof(T) / 4; };

// value held by the other lane of the pair (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
template <class T>
__device__ __forceinline__ T pair_xch(const T& v) {
    T r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < RegCount<T>::N; ++i) d[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)s[i], 0xB1, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ bool pair_xch_flag(bool f) {
    return __builtin_amdgcn_mov_dpp(f ? 1 : 0, 0xB1, 0xF, 0xF, true) != 0;
}

template <class T>
__device__ __forceinline__ T pair_sel(bool odd, const T& even_v, const T& odd_v) {
    T r;
    const uint32_t* e = reinterpret_cast<const uint32_t*>(&even_v);
    const uint32_t* o = reinterpret_cast<const uint32_t*>(&odd_v);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < RegCount<T>::N; ++i) d[i] = odd ? o[i] : e[i];
    return r;
}

template <class F>
__device__ __forceinline__ HalfPt<F> half_inf() {
    return {F::zero(), F::zero()};
}
// ZZ = 0 <=> ZZZ = 0 (ZZ^3 = ZZZ^2): either lane can tell from its own second component
template <class F>
__device__ __forceinline__ bool half_is_inf(const HalfPt<F>& p) {
    return F::is_zero(p.b);
}

// memory form stays the AoS XYZZ row [X | Y | ZZ | ZZZ] (LIMBS words each): a pair reads / writes one row together
template <class F>
__device__ __forceinline__ HalfPt<F> half_load(const uint32_t* row, bool odd) {
    constexpr int L = F::LIMBS;
    uint32_t w[2 * L];
    const uint4* q0 = reinterpret_cast<const uint4*>(row + (odd ? L : 0));
    const uint4* q1 = reinterpret_cast<const uint4*>(row + (odd ? 3 * L : 2 * L));
#pragma unroll
    for (int i = 0; i < L / 4; ++i) {
        uint4 t = q0[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
        uint4 u = q1[i];
        w[L + 4 * i] = u.x; w[L + 4 * i + 1] = u.y; w[L + 4 * i + 2] = u.z; w[L + 4 * i + 3] = u.w;
    }
    return {F::load(w), F::load(w + L)};
}
template <class F>
__device__ __forceinline__ void half_store(uint32_t* row, bool odd, const HalfPt<F>& p) {
    constexpr int L = F::LIMBS;
    uint32_t w[2 * L];
    F::store(w, p.a);
    F::store(w + L, p.b);
    uint4* q0 = reinterpret_cast<uint4*>(row + (odd ? L : 0));
    uint4* q1 = reinterpret_cast<uint4*>(row + (odd ? 3 * L : 2 * L));
#pragma unroll
    for (int i = 0; i < L / 4; ++i) {
        q0[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
        q1[i] = make_uint4(w[L + 4 * i], w[L + 4 * i + 1], w[L + 4 * i + 2], w[L + 4 * i + 3]);
    }
}

template <class F>
struct HalfRegs { static constexpr int COUNT = sizeof(HalfPt<F>) / 4; };

template <class F>
__device__ __forceinline__ HalfPt<F> half_shfl_xor(const HalfPt<F>& p, int mask) {
    HalfPt<F> r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < HalfRegs<F>::COUNT; ++i) d[i] = __shfl_xor(s[i], mask, 64);
    return r;
}
// LDS staging is word-major (word i of every lane contiguous): conflict-free for any register count
template <class F>
__device__ __forceinline__ void half_lds_put(uint32_t* sh, uint32_t stride, uint32_t slot, const HalfPt<F>& p) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
#pragma unroll
    for (int i = 0; i < HalfRegs<F>::COUNT; ++i) sh[(uint32_t)i * stride + slot] = s[i];
}
template <class F>
__device__ __forceinline__ HalfPt<F> half_lds_get(const uint32_t* sh, uint32_t stride, uint32_t slot) {
    HalfPt<F> r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < HalfRegs<F>::COUNT; ++i) d[i] = sh[(uint32_t)i * stride + slot];
    return r;
}

// p + q, both split over the lane pair.  Both lanes of a pair must call this together (same control flow).
template <class F>
__device__ __forceinline__ HalfPt<F> pair_add(const HalfPt<F>& p, const HalfPt<F>& q, bool odd) {
    typedef typename F::T T;
    if (half_is_inf<F>(q)) return p;
    if (half_is_inf<F>(p)) return q;
    if constexpr (F::RELAXED) {
        // Base-field groups: the differences that only feed products skip the range selection (carries only), as in the
        // accumulate step (curve.hip.h): 480 -> 268 instructions outside the seven products of the chain.  X may sit in [0, 4p)
        // (it only ever meets products); Y, ZZ, ZZZ stay below 2p (Y is doubled with a range-selecting addition in xyzz_dbl).
        //   d = m2 - m1 + 2p < 4p;  dd = d^2 (16 <= R/p);  PPP = d dd (8);  X3 = RR - PPP + 2p - 2Q in [0, 4p);
        //   Q - X3 + 4p < 6p;  R (Q - X3) (4 * 6 = 24)
        const T m1 = F::mul(p.a, q.b);   // U1 | S1
        const T m2 = F::mul(q.a, p.b);   // U2 | S2
        const T d = F::template sub_k<2>(m2, m1);   // P | R, in (0, 4p)
        const T dd = F::sqr(d);                     // PP | RR: zero exactly when d is zero mod p
        const bool dz = F::is_zero(dd);
        const bool dz_other = pair_xch_flag(dz);
        if (dz || dz_other) {
            const bool p_zero = odd ? dz_other : dz, r_zero = odd ? dz : dz_other;
            if (p_zero) {
                if (!r_zero) return half_inf<F>();  // P = -Q
                const T oa = pair_xch<T>(p.a), ob = pair_xch<T>(p.b);
                XYZZ<F> full = odd ? XYZZ<F>{oa, p.a, ob, p.b} : XYZZ<F>{p.a, oa, p.b, ob};
                full = xyzz_dbl<F>(full);
                return odd ? HalfPt<F>{full.Y, full.ZZZ} : HalfPt<F>{full.X, full.ZZ};
            }
        }
        const T t4 = F::mul(pair_sel<T>(odd, d, p.b), pair_sel<T>(odd, dd, q.b));   // PPP | ZZZ12
        const T x_dd = pair_xch<T>(dd);                                    // even: RR, odd: PP
        const T x_t4 = pair_xch<T>(t4);                                    // odd: PPP
        const T t5 = F::mul(pair_sel<T>(odd, m1, t4), pair_sel<T>(odd, dd, x_t4));  // Q | ZZZ3
        const T x3 = F::x3_sel4(F::template sub_k<2>(x_dd, t4), t5);      // even: X3 in [0, 4p) (odd: unused)
        const T qx = F::template sub_k<4>(t5, x3);                         // even: Q - X3 + 4p
        const T x_qx = pair_xch<T>(qx);
        const T t6 = F::mul(pair_sel<T>(odd, p.b, d), pair_sel<T>(odd, q.b, x_qx));  // ZZ12 | R (Q - X3)
        const T t7 = F::mul(pair_sel<T>(odd, t6, m1), pair_sel<T>(odd, dd, x_t4));   // ZZ3  | S1 PPP
        HalfPt<F> r;
        r.a = pair_sel<T>(odd, x3, F::sub(t6, t7));
        r.b = pair_sel<T>(odd, t7, t5);
        return r;
    }
    if constexpr (F::RELAXED2) {
        // Quadratic-extension groups, the same relaxation component by component (bounds as in xyzz_add_affine_relaxed2):
        // X in [0, 4p) -- it meets products with its c1 negated against 8p -- Y, ZZ, ZZZ below 2p;
        //   d = m2 - m1 + 2p < 4p;  dd = d^2 as ((d0 + d1)(d0 - d1 + 4p), (2 d0) d1): 8 * 8 = 64 <= R/p;
        //   PPP = d dd: 4*2 + 8*2 = 24;  Q - X3 + 4p < 6p;  R (Q - X3): 4*6 + 8*6 = 72 <= 169
        const T m1 = F::template mul_rel<8>(p.a, q.b);   // U1 | S1
        const T m2 = F::template mul_rel<8>(q.a, p.b);   // U2 | S2
        const T d = F::template sub_k<2>(m2, m1);        // P | R, components in (0, 4p)
        const T dd = F::template sqr_rel<4>(d);          // PP | RR, below 2p: zero exactly when d is zero in the field
        const bool dz = F::is_zero(dd);
        const bool dz_other = pair_xch_flag(dz);
        if (dz || dz_other) {
            const bool p_zero = odd ? dz_other : dz, r_zero = odd ? dz : dz_other;
            if (p_zero) {
                if (!r_zero) return half_inf<F>();  // P = -Q
                const T oa = pair_xch<T>(p.a), ob = pair_xch<T>(p.b);
                XYZZ<F> full = odd ? XYZZ<F>{oa, p.a, ob, p.b} : XYZZ<F>{p.a, oa, p.b, ob};
                full.X = F::reduce_2p(full.X);      // the plain doubling adds and subtracts components of X
                full = xyzz_dbl<F>(full);
                return odd ? HalfPt<F>{full.Y, full.ZZZ} : HalfPt<F>{full.X, full.ZZ};
            }
        }
        const T t4 = F::template mul_rel<8>(pair_sel<T>(odd, d, p.b), pair_sel<T>(odd, dd, q.b));   // PPP | ZZZ12
        const T x_dd = pair_xch<T>(dd);                                    // even: RR, odd: PP
        const T x_t4 = pair_xch<T>(t4);                                    // odd: PPP
        const T t5 = F::mul(pair_sel<T>(odd, m1, t4), pair_sel<T>(odd, dd, x_t4));  // Q | ZZZ3
        const T x3 = F::x3_sel4(F::template sub_k<2>(x_dd, t4), t5);      // even: X3 in [0, 4p) (odd: unused)
        const T qx = F::template sub_k<4>(t5, x3);                         // even: Q - X3 + 4p
        const T x_qx = pair_xch<T>(qx);
        const T t6 = F::template mul_rel<8>(pair_sel<T>(odd, p.b, d), pair_sel<T>(odd, q.b, x_qx));  // ZZ12 | R (Q - X3)
        const T t7 = F::mul(pair_sel<T>(odd, t6, m1), pair_sel<T>(odd, dd, x_t4));   // ZZ3  | S1 PPP
        HalfPt<F> r;
        r.a = pair_sel<T>(odd, x3, F::sub(t6, t7));
        r.b = pair_sel<T>(odd, t7, t5);
        return r;
    }
    const T m1 = F::mul(p.a, q.b);   // U1 | S1
    const T m2 = F::mul(q.a, p.b);   // U2 | S2
    const T d = F::sub(m2, m1);      // P  | R
    const bool dz = F::is_zero(d);
    const bool dz_other = pair_xch_flag(dz);
    if (dz || dz_other) {
        const bool p_zero = odd ? dz_other : dz, r_zero = odd ? dz : dz_other;
        if (p_zero) {
            if (!r_zero) return half_inf<F>();  // P = -Q
            // P = Q (equal buckets: duplicate bases with equal scalars): rebuild the whole point in both lanes and
            // double it with the one-lane formula; rare, so the redundancy does not matter
            const T oa = pair_xch<T>(p.a), ob = pair_xch<T>(p.b);
            XYZZ<F> full = odd ? XYZZ<F>{oa, p.a, ob, p.b} : XYZZ<F>{p.a, oa, p.b, ob};
            full = xyzz_dbl<F>(full);
            return odd ? HalfPt<F>{full.Y, full.ZZZ} : HalfPt<F>{full.X, full.ZZ};
        }
        // only R = 0 (same y-ratio, different x): an ordinary addition, fall through
    }
    const T dd = F::sqr(d);                                            // PP | RR
    const T t4 = F::mul(pair_sel<T>(odd, d, p.b), pair_sel<T>(odd, dd, q.b));   // PPP | ZZZ12
    const T x_dd = pair_xch<T>(dd);                                    // even: RR, odd: PP
    const T x_t4 = pair_xch<T>(t4);                                    // odd: PPP
    const T t5 = F::mul(pair_sel<T>(odd, m1, t4), pair_sel<T>(odd, dd, x_t4));  // Q | ZZZ3
    const T x3 = F::sub(F::sub(x_dd, t4), F::dbl(t5));                 // even: X3 = RR - PPP - 2Q (odd: unused)
    const T qx = F::sub(t5, x3);                                       // even: Q - X3
    const T x_qx = pair_xch<T>(qx);
    const T t6 = F::mul(pair_sel<T>(odd, p.b, d), pair_sel<T>(odd, q.b, x_qx));  // ZZ12 | R (Q - X3)
    const T t7 = F::mul(pair_sel<T>(odd, t6, m1), pair_sel<T>(odd, dd, x_t4));   // ZZ3  | S1 PPP
    HalfPt<F> r;
    r.a = pair_sel<T>(odd, x3, F::sub(t6, t7));
    r.b = pair_sel<T>(odd, t7, t5);
    return r;
}

}  // namespace zkmi
