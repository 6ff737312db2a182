// setup_impl.hip.h -- setup-side group kernels: batched normalisation (XYZZ -> affine with ONE field inversion per
// 1024 points), the fixed-base table 2^(c w) P_i of ZK_MSM_PRECOMPUTE plans, and fixed-base batch scalar
// multiplication.
//
// Stands in for batch_multi_scalar_g1/_g2 (reference src/bn254/curve.rs:326-354: a rayon map of `g * s`), which
// Groth16.setup drives with ONE base and n scalars (python/zksnake/groth16/protocol.py:81-97, ecc.py:93-94), and for the
// per-call `into_affine` of every base that multiscalar_mul_g1 pays (curve.rs:362-365) -- here paid once per key.
// Results are the unique affine representatives, so they are bit-identical with any correct implementation.
#pragma once
#include <map>
#include <vector>
#include "common.hip.h"

namespace zkmi {

// ---- batched normalisation ----------------------------------------------------------------------------------
// x = X / ZZ, y = Y / ZZZ for n XYZZ points.  Montgomery's trick at two levels: a lane multiplies up the z = ZZ ZZZ of
// its NORM_E consecutive points (prefix products parked in the x slot of the output rows), the 256 lane totals are
// combined by a prefix and a suffix product scan through LDS, ONE lane inverts the workgroup total (Fermat, ~380
// products) and every lane unwinds: ~13 products per point instead of ~390.
constexpr int NORM_THREADS = 256;
constexpr int NORM_E = 4;

template <class F>
__device__ __forceinline__ void lds_put_t(uint32_t* sh, uint32_t slot, const typename F::T& v) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(typename F::T) / 4); ++i) sh[(uint32_t)i * NORM_THREADS + slot] = s[i];
}
template <class F>
__device__ __forceinline__ typename F::T lds_get_t(const uint32_t* sh, uint32_t slot) {
    typename F::T r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(typename F::T) / 4); ++i) d[i] = sh[(uint32_t)i * NORM_THREADS + slot];
    return r;
}

template <class F, int WORDS>
__device__ __forceinline__ void ld_words(uint32_t* dst, const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) {
        uint4 t = q[i];
        dst[4 * i] = t.x; dst[4 * i + 1] = t.y; dst[4 * i + 2] = t.z; dst[4 * i + 3] = t.w;
    }
}
template <class F, int WORDS>
__device__ __forceinline__ void st_words(uint32_t* dst, const uint32_t* src) {
    uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) q[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}
template <class F>
__device__ __forceinline__ typename F::T ld_coord(const uint32_t* p) {
    uint32_t w[F::LIMBS];
    ld_words<F, F::LIMBS>(w, p);
    return F::load(w);
}
template <class F>
__device__ __forceinline__ void st_coord(uint32_t* p, const typename F::T& v) {
    uint32_t w[F::LIMBS];
    F::store(w, v);
    st_words<F, F::LIMBS>(p, w);
}

// out: affine rows of 2 * LIMBS words; canonical != 0 -> canonical integers (the ABI form), else Montgomery (plan tables)
template <class G>
__global__ __launch_bounds__(NORM_THREADS) void normalize_kernel(const uint32_t* __restrict__ xyzz, uint64_t n,
                                                                 uint32_t* __restrict__ out, int canonical) {
    typedef typename G::F F;
    typedef typename F::T T;
    constexpr int L = F::LIMBS, AW = 2 * L, XW = 4 * L;
    __shared__ uint32_t sh[NORM_THREADS * (sizeof(T) / 4)];
    const uint32_t tid = threadIdx.x;
    const uint64_t first = ((uint64_t)blockIdx.x * NORM_THREADS + tid) * NORM_E;
    // forward: running product of z_e = ZZ_e ZZZ_e (1 for a point at infinity); prefix e parked in out[e].x
    T run = F::one();
    for (int e = 0; e < NORM_E; ++e) {
        const uint64_t i = first + e;
        if (i >= n) break;
        const T zz = ld_coord<F>(xyzz + i * XW + 2 * L), zzz = ld_coord<F>(xyzz + i * XW + 3 * L);
        if (!F::is_zero(zz)) run = F::mul(run, F::mul(zz, zzz));
        st_coord<F>(out + i * AW, run);
    }
    // inclusive prefix products P and inclusive suffix products S of the lane totals
    T pre = run, suf = run;
    for (int off = 1; off < NORM_THREADS; off <<= 1) {
        lds_put_t<F>(sh, tid, pre);
        __syncthreads();
        if ((int)tid >= off) pre = F::mul(pre, lds_get_t<F>(sh, tid - off));
        __syncthreads();
        lds_put_t<F>(sh, tid, suf);
        __syncthreads();
        if ((int)tid + off < NORM_THREADS) suf = F::mul(suf, lds_get_t<F>(sh, tid + off));
        __syncthreads();
    }
    // 1 / (product of everything), by the last lane
    if (tid == NORM_THREADS - 1) lds_put_t<F>(sh, 0, F::inv(pre));
    __syncthreads();
    const T inv_all = lds_get_t<F>(sh, 0);
    __syncthreads();
    lds_put_t<F>(sh, tid, pre);
    __syncthreads();
    T before = tid > 0 ? lds_get_t<F>(sh, tid - 1) : F::one();   // P_{l-1}
    __syncthreads();
    lds_put_t<F>(sh, tid, suf);
    __syncthreads();
    T after = tid + 1 < NORM_THREADS ? lds_get_t<F>(sh, tid + 1) : F::one();  // S_{l+1}
    // 1 / (lane total) = P_{l-1} S_{l+1} / (all)
    T inv_run = F::mul(F::mul(before, after), inv_all);
    // backward: unwind the lane's own points
    for (int e = NORM_E - 1; e >= 0; --e) {
        const uint64_t i = first + e;
        if (i >= n) continue;
        const T zz = ld_coord<F>(xyzz + i * XW + 2 * L), zzz = ld_coord<F>(xyzz + i * XW + 3 * L);
        uint32_t w[AW];
        if (F::is_zero(zz)) {
#pragma unroll
            for (int k = 0; k < AW; ++k) w[k] = 0;  // infinity: the (0, 0) sentinel in both forms
        } else {
            const T prev = e > 0 ? ld_coord<F>(out + (i - 1) * AW) : F::one();  // prefix up to e - 1 (same lane wrote it)
            const T inv_z = F::mul(inv_run, prev);             // 1 / (ZZ ZZZ)
            inv_run = F::mul(inv_run, F::mul(zz, zzz));
            const T x = F::mul(ld_coord<F>(xyzz + i * XW), F::mul(inv_z, zzz));          // X / ZZ
            const T y = F::mul(ld_coord<F>(xyzz + i * XW + L), F::mul(inv_z, zz));       // Y / ZZZ
            if (canonical) {
                F::to_canonical(w, x);
                F::to_canonical(w + L, y);
            } else {
                F::store(w, x);
                F::store(w + L, y);
            }
        }
        st_words<F, AW>(out + i * AW, w);
    }
}

// temp[i] = 2^doublings * (src affine row i, or temp[i] itself when src == nullptr), XYZZ, in place
template <class G>
__global__ __launch_bounds__(256) void dbl_rows_kernel(uint32_t* __restrict__ temp, uint64_t n, int doublings, const uint32_t* __restrict__ src) {
    typedef typename G::F F;
    constexpr int L = F::LIMBS, AW = 2 * L, XW = 4 * L;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> acc;
    if (src) {
        Affine<F> p = {ld_coord<F>(src + i * AW), ld_coord<F>(src + i * AW + L)};
        acc = xyzz_from_affine<F>(p);
    } else {
        acc = {ld_coord<F>(temp + i * XW), ld_coord<F>(temp + i * XW + L), ld_coord<F>(temp + i * XW + 2 * L), ld_coord<F>(temp + i * XW + 3 * L)};
    }
    for (int k = 0; k < doublings; ++k) acc = xyzz_dbl<F>(acc);
    st_coord<F>(temp + i * XW, acc.X);
    st_coord<F>(temp + i * XW + L, acc.Y);
    st_coord<F>(temp + i * XW + 2 * L, acc.ZZ);
    st_coord<F>(temp + i * XW + 3 * L, acc.ZZZ);
}

// ---- fixed-base batch multiplication ---------------------------------------------------------------------------
// table[j][d - 1] = d * 2^(16 j) * G for d = 1 .. 2^15 (affine, Montgomery): a scalar is 16 signed 16-bit digits, i.e.
// at most 16 mixed additions and no doubling.  FIXED_WINDOWS * 2^15 rows: 32 MiB for BN254 G1, 96 MiB for BLS12-381 G2.
constexpr int FIXED_C = 16;
constexpr uint32_t FIXED_HALF = 1u << (FIXED_C - 1);

// temp[(j, d-1)] = d * B_j in XYZZ, B_j = window_bases[j] (affine, Montgomery)
template <class G>
__global__ __launch_bounds__(256) void fixed_table_kernel(const uint32_t* __restrict__ window_bases, int nwin, uint32_t* __restrict__ temp) {
    typedef typename G::F F;
    constexpr int L = F::LIMBS, AW = 2 * L, XW = 4 * L;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nwin * FIXED_HALF) return;
    const uint32_t j = (uint32_t)(i / FIXED_HALF), d = (uint32_t)(i % FIXED_HALF) + 1;
    Affine<F> b = {ld_coord<F>(window_bases + (size_t)j * AW), ld_coord<F>(window_bases + (size_t)j * AW + L)};
    uint32_t k[1] = {d};
    XYZZ<F> acc = xyzz_scalar_mul<F>(b, k, 1);
    st_coord<F>(temp + i * XW, acc.X);
    st_coord<F>(temp + i * XW + L, acc.Y);
    st_coord<F>(temp + i * XW + 2 * L, acc.ZZ);
    st_coord<F>(temp + i * XW + 3 * L, acc.ZZZ);
}

struct FixedBias {
    uint32_t v[13];
};

// temp[i] = k_i * G in XYZZ from the table (scalars canonical words, reduced mod r here like Fr::from(BigUint))
template <class G>
__global__ __launch_bounds__(256) void fixed_mul_kernel(const uint32_t* __restrict__ scalars, uint64_t n, const uint32_t* __restrict__ table,
                                                        int nwin, FixedBias bias, uint32_t* __restrict__ temp) {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    constexpr int L = F::LIMBS, AW = 2 * L, XW = 4 * L, N = FrP::W;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[N + 1];
    ld_words<F, N>(s, scalars + i * N);
    for (int k = 0; k < 10; ++k) {
        uint32_t t[N];
        if (fp_sub_mod_raw<FrP>(t, s)) break;
#pragma unroll
        for (int l = 0; l < N; ++l) s[l] = t[l];
    }
    // signed digits through the bias trick of the MSM (s + sum 2^15 2^(16 j), then plain 16-bit fields minus 2^15)
    uint64_t carry = 0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
        uint64_t t = (uint64_t)s[l] + bias.v[l] + carry;
        s[l] = (uint32_t)t;
        carry = t >> 32;
    }
    s[N] = (uint32_t)carry + bias.v[N];
    XYZZ<F> acc = xyzz_inf<F>();
    for (int j = 0; j < nwin; ++j) {
        const int word = j >> 1, off = (j & 1) * 16;
        const int v = (int)((s[word] >> off) & 0xFFFFu) - (int)FIXED_HALF;
        if (v != 0) {
            const uint32_t d = (uint32_t)(v < 0 ? -v : v);
            xyzz_add_affine_mem<F>(acc, table + ((size_t)j * FIXED_HALF + d - 1) * AW, v < 0);
        }
    }
    xyzz_relaxed_finish<F>(acc);
    st_coord<F>(temp + i * XW, acc.X);
    st_coord<F>(temp + i * XW + L, acc.Y);
    st_coord<F>(temp + i * XW + 2 * L, acc.ZZ);
    st_coord<F>(temp + i * XW + 3 * L, acc.ZZZ);
}

// temp[i] = k_i * P_i (XYZZ), per-element bases or one base for all (small batches): double-and-add per lane
template <class G>
__global__ __launch_bounds__(128) void varbase_mul_kernel(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ bases, int broadcast,
                                                          uint64_t n, uint32_t* __restrict__ temp) {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    constexpr int L = F::LIMBS, AW = 2 * L, XW = 4 * L;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[FrP::W];
    ld_words<F, FrP::W>(k, scalars + i * FrP::W);
    for (int r = 0; r < 10; ++r) {
        uint32_t t[FrP::W];
        if (fp_sub_mod_raw<FrP>(t, k)) break;
#pragma unroll
        for (int l = 0; l < FrP::W; ++l) k[l] = t[l];
    }
    uint32_t w[AW];
    ld_words<F, AW>(w, bases + (broadcast ? 0 : i * AW));
    Affine<F> p;
    p.x = F::from_canonical(w);
    p.y = F::from_canonical(w + L);
    XYZZ<F> acc = xyzz_scalar_mul<F>(p, k, FrP::W);
    st_coord<F>(temp + i * XW, acc.X);
    st_coord<F>(temp + i * XW + L, acc.Y);
    st_coord<F>(temp + i * XW + 2 * L, acc.ZZ);
    st_coord<F>(temp + i * XW + 3 * L, acc.ZZZ);
}

#define ZK_SETUP_EXTERN_TEMPLATES(G)                                                                                         \
    extern template __global__ void normalize_kernel<G>(const uint32_t*, uint64_t, uint32_t*, int);                          \
    extern template __global__ void dbl_rows_kernel<G>(uint32_t*, uint64_t, int, const uint32_t*);                           \
    extern template __global__ void fixed_table_kernel<G>(const uint32_t*, int, uint32_t*);                                  \
    extern template __global__ void fixed_mul_kernel<G>(const uint32_t*, uint64_t, const uint32_t*, int, FixedBias, uint32_t*); \
    extern template __global__ void varbase_mul_kernel<G>(const uint32_t*, const uint32_t*, int, uint64_t, uint32_t*);
#define ZK_SETUP_INSTANTIATE(G)                                                                                       \
    template __global__ void normalize_kernel<G>(const uint32_t*, uint64_t, uint32_t*, int);                          \
    template __global__ void dbl_rows_kernel<G>(uint32_t*, uint64_t, int, const uint32_t*);                           \
    template __global__ void fixed_table_kernel<G>(const uint32_t*, int, uint32_t*);                                  \
    template __global__ void fixed_mul_kernel<G>(const uint32_t*, uint64_t, const uint32_t*, int, FixedBias, uint32_t*); \
    template __global__ void varbase_mul_kernel<G>(const uint32_t*, const uint32_t*, int, uint64_t, uint32_t*);

}  // namespace zkmi
