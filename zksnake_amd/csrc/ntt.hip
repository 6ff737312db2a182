// ntt.hip -- scalar-field radix-2 NTT / iNTT, element-wise vector ops and the fused QAP
// quotient pipeline for BN254 Fr and BLS12-381 Fr on gfx950.
//
// Stands in for fft/ifft/coset_fft/coset_ifft, add/mul_over_evaluation_domain and
// Polynomial.divide_by_vanishing_poly of the reference (src/bn254/polynomial.rs:466-489,
// 535-634 and the bls12_381 twin), which call ark-poly 0.4.2's Radix2EvaluationDomain.
//
// Layout in HBM: a vector is 2^k elements x 8 u32 (32 B), canonical integers, natural order.
// The transform is linear, so it runs directly on canonical data with twiddles kept in
// Montgomery form: mont_mul(x, w*R) = x*w.  No conversion pass is needed on either side.
//
// Kernel structure: decimation-in-frequency, several butterfly stages per launch on an LDS
// tile (structure-of-arrays: limb-major, so consecutive lanes hit consecutive banks), then one
// bit-reversal pass that also applies 1/N for the inverse transform.  Twiddles w^k (k < N/2)
// are precomputed once per (field, size, direction) and streamed from HBM.
#include <algorithm>
#include <mutex>
#include <map>
#include <vector>
#include "common.hip.h"
#include "fr_mem.hip.h"

namespace zkmi {

constexpr int NTT_TILE_LOG = 11;   // 2048 elements = 72 KiB of LDS per workgroup (limb-major), two workgroups per CU
#ifndef NTT_THREADS_N
#define NTT_THREADS_N 512
#endif
#ifndef NTT_MIN_BLOCKS
#define NTT_MIN_BLOCKS 2
#endif
constexpr int NTT_THREADS = NTT_THREADS_N;
constexpr int NTT_MAX_DIGIT = 8;   // stages per pass: 2^8 x 2^3 runs of 256 B at least

// ---- twiddle generation -------------------------------------------------------------------
// Stage-major table, independent of the transform size: stage s of a DIF transform pairs (i, i + 2^s) and multiplies the
// difference by zeta_{s+1}^(i mod 2^s), zeta_k = primitive 2^k-th root of unity.  Row s holds zeta_{s+1}^j, j < 2^s, at
// offset 2^s - 1: consecutive butterflies read consecutive entries (the former single table w_n^k made every butterfly of
// a low stage fetch its own 64-byte sector: one L2 request each).
template <class P>
__global__ void twiddle_kernel(uint32_t* out, Fp<P> zeta, uint32_t count) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t e[1] = {k};
    Fp<P> r = fp_pow<P>(zeta, e, 1);
    store_fr<P>(out + (size_t)k * P::W, r);
}

// out[k] = g^k as canonical integers (setup: the powers of tau), square-and-multiply per element
template <class P>
__global__ void powers_kernel(uint32_t* out, Fp<P> g, uint32_t count) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t e[1] = {k};
    Fp<P> r = fp_pow<P>(g, e, 1);
    uint32_t w[P::W];
    fp_to_canonical<P>(w, r);
    uint4* q = reinterpret_cast<uint4*>(out + (size_t)k * P::W);
#pragma unroll
    for (int i = 0; i < P::W / 4; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ---- lazy-range arithmetic of the butterfly passes ---------------------------------------------------------------
// A radix-4 step of the passes below used to spend a fifth of its VALU instructions on four range-selecting additions
// (fp_add: two candidate results carried through one pass, then a select).  Inside a pass the values now live in a wider
// range instead: every element of the LDS tile is NORMALISED (limbs < 2^29) with value < 9p, sums are formed limb by limb
// without carries, and only the one output per step that is a sum of sums is brought back -- by ONE estimated multiple of
// 8p and a signed carry pass.  For a step on (x00, x01, x10, x11), all < 9p:
//     a0 = x00 + x10, b0 = x01 + x11             limb-wise: limbs < 2^30, value < 18p
//     a1 = w (x00 - x10 + 18p*), b1 = w' (..)    operand limbs < 3 2^29, value < 27p; product < 2p, normalised
//     x00' = reduce8(a0 + b0)                    limbs < 2^31, value < 36p  ->  normalised, < 8.7p
//     x01' = w2 (a0 - b0 + 36p*)                 operand limbs < 5 2^29, value < 54p: columns 9 (5 + 1) 2^58 < 2^64
//     x10' = normalise(a1 + b1)                  < 4p
//     x11' = w2 (a1 - b1 + 4p*)                  as before
// (kp* = k p written with borrow-proof limbs).  A Montgomery product needs (a/p)(b/p) <= R/p = 2^261/p (168 for BN254 Fr,
// 70.7 for BLS12-381 Fr): the twiddles of the 9-word table are canonical (< p), so 54 * 1 fits both fields.
// tools/model_lazy_ntt.py replays these steps on integers with the limb and column bounds asserted.

// k p as normalised 29-bit limbs, at compile time
template <class P>
struct LimbConst { uint32_t v[P::N]; };
template <class P>
constexpr LimbConst<P> times_p(uint32_t k) {
    LimbConst<P> r{};
    uint64_t carry = 0;
    for (int i = 0; i < P::N; ++i) {
        const uint64_t t = (uint64_t)P::M[i] * k + carry;
        r.v[i] = i < P::N - 1 ? (uint32_t)(t & LIMB_MASK) : (uint32_t)t;
        carry = t >> LIMB_BITS;
    }
    return r;
}

// a + b, limb by limb (no carries): the caller accounts for the limb width
template <class P>
__device__ __forceinline__ Fp<P> lz_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a - b + K p with borrow-proof limbs: every limb of K p but the top one borrows 2^BITS from the limb above, so no limb goes
// negative for b with limbs < 2^BITS and value <= K p / 2.  One operand of a product only.
template <class P, int K, int BITS>
__device__ __forceinline__ Fp<P> lz_sub(const Fp<P>& a, const Fp<P>& b) {
    constexpr LimbConst<P> kp = times_p<P>(K);
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const uint32_t c = kp.v[i] + (i < P::N - 1 ? (1u << BITS) : 0u) - (i > 0 ? (1u << (BITS - LIMB_BITS)) : 0u);
        r.v[i] = a.v[i] + c - b.v[i];
    }
    return r;
}

// carry pass: limbs < 2^31 in, normalised limbs out, same value
template <class P>
__device__ __forceinline__ Fp<P> lz_norm(const Fp<P>& a) {
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const uint32_t t = a.v[i] + c;
        if (i < P::N - 1) {
            r.v[i] = t & LIMB_MASK;
            c = t >> LIMB_BITS;
        } else {
            r.v[i] = t;
        }
    }
    return r;
}

// a - k U p for the estimate k = floor(top(a) / (top(U p) + 1)) (one multiply-high), then a signed carry pass: limbs < 2^31 and
// value < 4.5 U p in, normalised limbs and value < 1.09 U p out (k <= 4, so k * limb(U p) < 2^31 and every limb difference
// fits a signed 32-bit register).  U = 8 inside a pass, U = 2 on the way to a canonical result.
template <class P, int U>
__device__ __forceinline__ Fp<P> lz_reduce(const Fp<P>& a) {
    constexpr int N = P::N;
    constexpr LimbConst<P> up = times_p<P>(U);
    constexpr uint32_t MAGIC = (uint32_t)((1ull << 32) / ((uint64_t)up.v[N - 1] + 1));
    const uint32_t t = a.v[N - 1] + (a.v[N - 2] >> LIMB_BITS);
    const uint32_t k = __umulhi(t, MAGIC);
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int32_t d = (int32_t)(a.v[i] - k * up.v[i]) + c;
        if (i < N - 1) {
            r.v[i] = (uint32_t)d & LIMB_MASK;
            c = d >> LIMB_BITS;
        } else {
            r.v[i] = (uint32_t)d;
        }
    }
    return r;
}

// normalised, value < 9p  ->  canonical: one estimated multiple of 2p, then two conditional subtractions of p
template <class P>
__device__ __forceinline__ Fp<P> lz_canonical(const Fp<P>& a) {
    return fp_reduce_full<P>(fp_reduce_full<P>(lz_reduce<P, 2>(a)));
}

// an element as NINE raw limb words (36 bytes): the twiddle table of the passes and the scratch vectors between two passes
// keep register form, so that neither side pays the 8-word pack / unpack (78 VALU instructions per radix-4 step for the three
// twiddles alone) and values above 2^256 (< 9p between passes) have room
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <class P>
__device__ __forceinline__ Fp<P> load_limbs9(const uint32_t* p) {
    static_assert(P::N == 9, "scalar fields of nine 29-bit limbs");
    Fp<P> r;
    const u32x4_a4 a = *reinterpret_cast<const u32x4_a4*>(p), b = *reinterpret_cast<const u32x4_a4*>(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = p[8];
    return r;
}
template <class P>
__device__ __forceinline__ void store_limbs9(uint32_t* p, const Fp<P>& a) {
    u32x4_a4 x, y;
    x.x = a.v[0]; x.y = a.v[1]; x.z = a.v[2]; x.w = a.v[3];
    y.x = a.v[4]; y.y = a.v[5]; y.z = a.v[6]; y.w = a.v[7];
    *reinterpret_cast<u32x4_a4*>(p) = x;
    *reinterpret_cast<u32x4_a4*>(p + 4) = y;
    p[8] = a.v[8];
}
constexpr int TW_WORDS = 9;   // words per entry of the stage-major twiddle table and of a scratch element

// stage-major twiddle table in 9-word form: canonical Montgomery representatives (< p), see the range notes above
template <class P>
__global__ void twiddle9_kernel(uint32_t* out, Fp<P> zeta, uint32_t count) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t e[1] = {k};
    store_limbs9<P>(out + (size_t)k * TW_WORDS, fp_reduce_full<P>(fp_pow<P>(zeta, e, 1)));
}

// ---- butterfly passes -----------------------------------------------------------------------
// Decimation in frequency with the digit reversal folded into the passes.  Before a pass, `done` low bits of the
// physical index hold finished (bit-reversed) frequency digits and the upper rem = log_n - done bits the remaining
// logical index in natural order.  A pass takes the top m remaining bits D: a workgroup owns all 2^m values of D for a
// run of 2^q consecutive values of the rest of the index (so every global access is a run of 2^q elements), performs
// stages rem-1 .. rem-m on them in LDS and writes element (D, low, fin) to
//        (low << (done + m)) | (bitrev_m(D) << done) | fin,
// i.e. D's frequency digit lands just above the digits already finished.  After the last pass the vector is in natural
// order: no scattered final store, no separate reordering pass.  `in` and `out` must be different vectors unless the
// pass is the only one (a single workgroup that reads everything before it writes).
// Element formats: the first pass reads canonical 8-word elements, the last pass writes them (multiplied by `scale` when
// use_scale != 0: 1/N of the inverse transform); between two passes the elements travel as nine raw limb words with values
// below 9p (in_raw / out_raw), see the range notes above.
// What a pass can take on from the kernels around a transform (the QAP chain, qap_h_dev_impl):
//   first pass   in_v != nullptr: the element is not read but FORMED, in[p] * in_v[p] / R (the point-wise product of two coset
//                evaluation vectors; canonical inputs);
//   last pass    scale_tab != nullptr: the output is multiplied by scale_tab[index] (canonical Montgomery representatives, 8
//                words: g^i / N, or a multiple of g^-i -- the coset shift rides on the product the inverse transform pays for 1/N
//                anyway); with out2 != nullptr that goes to out2 and `out` still receives x * scale (u and u g^i from one pass);
//                with add_tab != nullptr a second vector is added (h = w / 2 + d (-g^-i / 2N), see qap_h_dev_impl).
struct NttFuse {
    const uint32_t* in_v = nullptr;       // first pass: the element is not read but FORMED, mont(in[p], in_v[p]) = in[p] in_v[p] / R
    const uint32_t* scale_tab = nullptr;
    uint32_t* out2 = nullptr;
    const uint32_t* add_tab = nullptr;    // last pass, with scale_tab and without out2: out = x * scale_tab[index] + add_tab[index]
};

template <class P>
__global__ __launch_bounds__(NTT_THREADS, NTT_MIN_BLOCKS) void ntt_pass_kernel(const uint32_t* in, uint32_t* out,
                                                               const uint32_t* __restrict__ tw, int log_n,
                                                               int rem, int m, int q, int in_raw, int final_pass, Fp<P> scale, int use_scale,
                                                               NttFuse fuse) {
    constexpr int N = P::N;  // register limbs (LDS is limb-major)
    constexpr int W = P::W;  // words per canonical element in HBM
    constexpr int T = TW_WORDS;
    extern __shared__ uint32_t lds[];  // [N][tile]
    const int done = log_n - rem;
    const int tile_log = m + q;
    const uint32_t tile = 1u << tile_log;
    const int rest_bits = log_n - m;
    const uint32_t rest0 = blockIdx.x << q;
    const uint32_t qmask = (1u << q) - 1;
    // LDS element index, skewed by one bank per 2^(tile_log - 5) elements: the first pass reads its output along the
    // bit-reversed digit (stride 2^q elements over the TOP five bits of the lane group), which would otherwise hit one bank
    const int pad_shift = tile_log >= 10 ? tile_log - 5 : 31;
    const uint32_t row = tile + 32;
    // ... and bits 3-4 XORed with bits 5-6: a radix-4 step on element bits (3, 4) -- the last step of a pass with q = 3 -- reads
    // and writes 8 consecutive elements per 8 lanes at a stride of 32 elements, four lane groups on the same 8 banks (4-way
    // conflicts: 37.5 % of the LDS-active cycles of the first and last pass of a 2^22 transform,
    // profiles/r04_ntt_lds_counters_baseline.json); with the swizzle the four groups land on four different bank octets, and
    // every other access pattern of the kernel keeps bits 5-6 constant within a 32-lane group
    const uint32_t swz = tile_log >= 7 ? 3u : 0u;
#define LIDX(e) ((((e) ^ ((((e) >> 5) & swz) << 3))) + ((e) >> pad_shift))

    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        const uint32_t d = e >> q, r = e & qmask;
        const uint32_t p = (d << rest_bits) | rest0 | r;
        Fp<P> x;
        if (in_raw) {
            x = load_limbs9<P>(in + (size_t)p * T);
        } else if (fuse.in_v) {
            // the point-wise product of two canonical vectors, u v / R (the R travels with the linear transform and is undone by the
            // last pass's table)
            x = fp_mul<P>(load_fr<P>(in + (size_t)p * W), load_fr<P>(fuse.in_v + (size_t)p * W));   // < 2p, normalised
        } else {
            x = load_fr<P>(in + (size_t)p * W);
        }
        const uint32_t le = LIDX(e);
#pragma unroll
        for (int l = 0; l < N; ++l) lds[l * row + le] = x.v[l];
    }
    __syncthreads();

    // Stages in pairs: a thread takes the four elements that differ in bits (b, b-1) of D, runs both stages on them
    // in registers (four butterflies, three distinct twiddles) and writes them back: half the LDS traffic and half the
    // barriers of one stage per round trip.  An odd stage count starts with a single radix-2 stage.
    int b = m - 1;
    if (m & 1) {
        const int s = rem - m + b;
        const int pos = b + q;
        const uint32_t* tws = tw + ((size_t)(1u << s) - 1) * T;
        for (uint32_t u = threadIdx.x; u < (tile >> 1); u += NTT_THREADS) {
            const uint32_t e0 = ((u >> pos) << (pos + 1)) | (u & ((1u << pos) - 1));
            const uint32_t e1 = e0 | (1u << pos);
            const uint32_t d0 = e0 >> q, r = e0 & qmask;
            const uint32_t low = (rest0 | r) >> done;                              // remaining logical bits below D
            const uint32_t j = ((d0 & ((1u << b) - 1)) << (rem - m)) | low;        // i mod 2^s
            Fp<P> x, y;
            const uint32_t l0 = LIDX(e0), l1 = LIDX(e1);
#pragma unroll
            for (int l = 0; l < N; ++l) { x.v[l] = lds[l * row + l0]; y.v[l] = lds[l * row + l1]; }
            Fp<P> w = load_limbs9<P>(tws + (size_t)j * T);
            Fp<P> sum = lz_reduce<P, 8>(lz_add<P>(x, y));              // < 18p -> < 8.7p
            Fp<P> dif = fp_mul<P>(w, lz_sub<P, 18, 29>(x, y));         // operand < 27p
#pragma unroll
            for (int l = 0; l < N; ++l) { lds[l * row + l0] = sum.v[l]; lds[l * row + l1] = dif.v[l]; }
        }
        __syncthreads();
        --b;
    }
    for (; b >= 1; b -= 2) {
        const int s = rem - m + b;          // upper stage of the pair; the lower one is s - 1
        const int pos_lo = b - 1 + q;
        const uint32_t* tw_hi = tw + ((size_t)(1u << s) - 1) * T;
        const uint32_t* tw_lo = tw + ((size_t)(1u << (s - 1)) - 1) * T;
        for (uint32_t u = threadIdx.x; u < (tile >> 2); u += NTT_THREADS) {
            const uint32_t e00 = ((u >> pos_lo) << (pos_lo + 2)) | (u & ((1u << pos_lo) - 1));
            const uint32_t e01 = e00 | (1u << pos_lo), e10 = e00 | (2u << pos_lo), e11 = e00 | (3u << pos_lo);
            const uint32_t d0 = e00 >> q, r = e00 & qmask;
            const uint32_t low = (rest0 | r) >> done;
            const uint32_t j0 = ((d0 & ((1u << b) - 1)) << (rem - m)) | low;            // stage s, bit b-1 of D clear
            const uint32_t j1 = j0 + ((1u << (b - 1)) << (rem - m));                    // stage s, bit b-1 set
            const uint32_t j2 = ((d0 & ((1u << (b - 1)) - 1)) << (rem - m)) | low;      // stage s-1 (both pairs)
            const uint32_t l00 = LIDX(e00), l01 = LIDX(e01), l10 = LIDX(e10), l11 = LIDX(e11);
            // the three twiddles first: their loads travel while the tile elements come out of LDS (left where they were used,
            // each load sat directly in front of its product with a full memory latency exposed, three times per step)
            const Fp<P> w0 = load_limbs9<P>(tw_hi + (size_t)j0 * T);
            const Fp<P> w1 = load_limbs9<P>(tw_hi + (size_t)j1 * T);
            const Fp<P> w2 = load_limbs9<P>(tw_lo + (size_t)j2 * T);
            Fp<P> x00, x01, x10, x11;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                x00.v[l] = lds[l * row + l00]; x01.v[l] = lds[l * row + l01];
                x10.v[l] = lds[l * row + l10]; x11.v[l] = lds[l * row + l11];
            }
            // stage s: (x00, x10) and (x01, x11); every input normalised and < 9p (range notes above)
            Fp<P> a0 = lz_add<P>(x00, x10);
            Fp<P> a1 = fp_mul<P>(w0, lz_sub<P, 18, 29>(x00, x10));
            Fp<P> b0 = lz_add<P>(x01, x11);
            Fp<P> b1 = fp_mul<P>(w1, lz_sub<P, 18, 29>(x01, x11));
            // stage s-1: (a0, b0) and (a1, b1), one twiddle
            x00 = lz_reduce<P, 8>(lz_add<P>(a0, b0));
            x01 = fp_mul<P>(w2, lz_sub<P, 36, 30>(a0, b0));
            x10 = lz_norm<P>(lz_add<P>(a1, b1));
            x11 = fp_mul<P>(w2, fp_sub_lazy<P>(a1, b1));
#pragma unroll
            for (int l = 0; l < N; ++l) {
                lds[l * row + l00] = x00.v[l]; lds[l * row + l01] = x01.v[l];
                lds[l * row + l10] = x10.v[l]; lds[l * row + l11] = x11.v[l];
            }
        }
        // A step on element bits (pos_lo, pos_lo + 1) with pos_lo <= 6 stays inside blocks of 256 consecutive elements, and with
        // one element quad per lane (tile / 4 == NTT_THREADS) wave w holds exactly block w: when this step and the next are both of
        // that kind the next one reads only what this wave wrote, and the LDS serves one wave's accesses in order -- no workgroup
        // barrier, the waves drift apart for two steps (the last two steps of a pass with q = 3)
        const bool wave_private = (tile >> 2) == NTT_THREADS && pos_lo <= 6 && b >= 3;
        if (wave_private) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        else __syncthreads();
    }

    const uint32_t fin_mask = done >= 32 ? 0xFFFFFFFFu : ((1u << done) - 1);
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        // walk the OUTPUT in its run order: with done >= q the run (low q bits) stays in place and D' is next
        uint32_t d, r;
        if (done >= q) { r = e & qmask; d = e >> q; }
        else { d = e & ((1u << m) - 1); r = e >> m; }      // first pass: D' is the fastest output digit
        const uint32_t src = (__brev(d) >> (32 - m)) << q | r;  // element whose D equals bitrev(d)
        const uint32_t rest = rest0 | r;
        const uint32_t po = ((rest >> done) << (done + m)) | (d << done) | (rest & fin_mask);
        Fp<P> x;
        const uint32_t ls = LIDX(src);
#pragma unroll
        for (int l = 0; l < N; ++l) x.v[l] = lds[l * row + ls];
        if (final_pass) {
            // 1/N of the inverse transform rides on a product (9 * 1 <= R/p: the result is below 2p); a forward transform
            // comes down from < 9p by an estimated multiple of 2p
            if (fuse.scale_tab) {
                // x < 9p times a canonical constant: 9 * 1 <= R/p, the product is below 2p
                const Fp<P> t = load_fr<P>(fuse.scale_tab + (size_t)po * W);
                if (fuse.out2) {
                    store_fr<P>(fuse.out2 + (size_t)po * W, fp_reduce_full<P>(fp_mul<P>(x, t)));
                    store_fr<P>(out + (size_t)po * W, fp_reduce_full<P>(fp_mul<P>(x, scale)));
                } else if (fuse.add_tab) {
                    const Fp<P> y = fp_add<P>(fp_mul<P>(x, t), load_fr<P>(fuse.add_tab + (size_t)po * W));   // < 2p + canonical -> below 2p
                    store_fr<P>(out + (size_t)po * W, fp_reduce_full<P>(y));
                } else {
                    store_fr<P>(out + (size_t)po * W, fp_reduce_full<P>(fp_mul<P>(x, t)));
                }
            } else {
                x = use_scale ? fp_reduce_full<P>(fp_mul<P>(x, scale)) : lz_canonical<P>(x);
                store_fr<P>(out + (size_t)po * W, x);
            }
        } else {
            store_limbs9<P>(out + (size_t)po * T, x);
        }
    }
#undef LIDX
}

// x[i] *= g^(+-i) with g = w_n: row log_n - 1 of the stage-major table holds w_n^k for k < n/2; w_n^(n/2) = -1
template <class P>
__global__ void coset_scale_kernel(uint32_t* data, const uint32_t* tw, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n == 1) return;
    uint32_t half = n >> 1;
    Fp<P> w = load_limbs9<P>(tw + ((size_t)(half - 1) + (i & (half - 1))) * TW_WORDS);
    if (i >= half) w = fp_neg<P>(w);
    Fp<P> a = load_fr<P>(data + (size_t)i * P::W);
    store_fr<P>(data + (size_t)i * P::W, fp_reduce_full<P>(fp_mul<P>(a, w)));
}

// op: 0 mul, 1 add, 2 sub on canonical data.  mul: mont(mont(a,b), R^2) = a*b
template <class P>
__global__ void vec_op_kernel(int op, uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> x = load_fr<P>(a + i * P::W), y = load_fr<P>(b + i * P::W), z;
    if (op == 0) z = fp_mul<P>(fp_mul<P>(x, y), fp_const<P>(P::R2));
    else if (op == 1) z = fp_add<P>(x, y);
    else z = fp_sub<P>(x, y);
    store_fr<P>(out + i * P::W, fp_reduce_full<P>(z));
}

// reduce every element below the modulus (inputs of the host API may be >= r)
template <class P>
__global__ void canon_kernel(uint64_t n, uint32_t* a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[P::W];
    uint4* q = reinterpret_cast<uint4*>(a + i * P::W);
#pragma unroll
    for (int k = 0; k < P::W / 4; ++k) {
        uint4 t = q[k];
        x[4 * k] = t.x; x[4 * k + 1] = t.y; x[4 * k + 2] = t.z; x[4 * k + 3] = t.w;
    }
    for (int k = 0; k < 10; ++k) {
        uint32_t t[P::W];
        if (fp_sub_mod_raw<P>(t, x)) break;
#pragma unroll
        for (int l = 0; l < P::W; ++l) x[l] = t[l];
    }
#pragma unroll
    for (int k = 0; k < P::W / 4; ++k) q[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
}

// q[j] = sum_{k>=1} c[j + k n],  rem[i] = c[i] + q[i]   (c has len coefficients)
template <class P>
__global__ void div_vanishing_kernel(uint64_t n, uint64_t len, const uint32_t* c, uint32_t* q, uint32_t* rem,
                                     int* nonzero) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t top = len < n ? len : n;
    uint64_t qlen = len > n ? len - n : 0;
    uint64_t lim = qlen > top ? qlen : top;
    if (i >= lim) return;
    // q[i] for i < qlen (may exceed n when len > 2n: then q[i] also folds further blocks)
    Fp<P> acc = fp_zero<P>();
    if (i < qlen) {
        for (uint64_t j = i + n; j < len; j += n) acc = fp_add<P>(acc, load_fr<P>(c + j * P::W));
        store_fr<P>(q + i * P::W, fp_reduce_full<P>(acc));
    }
    if (i < top) {
        Fp<P> r = fp_reduce_full<P>(fp_add<P>(load_fr<P>(c + i * P::W), acc));
        store_fr<P>(rem + i * P::W, r);
        if (!fp_is_zero<P>(r)) atomicOr(nonzero, 1);
    }
}

// CSR sparse matrix-vector product over Fr: out[row] = sum vals[k] * w[cols[k]].
// Canonical in/out: mont(mont(v, x), R^2) = v*x; the sum is accumulated in Montgomery-by-R^-1 form
// and fixed up once per row.
// Short rows (the bulk of an R1CS) take one lane each.  A row longer than `skip_longer_than` is left to the long-row
// kernels below: the constant-one wire and the input wires of a real circuit (and every row of the transposed matrices
// that Groth16.setup multiplies) can hold 2^20 entries, and one lane walking them took 0.8 s.
template <class P>
__global__ void spmv_kernel(uint64_t n_rows, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ cols,
                            const uint32_t* __restrict__ vals, const uint32_t* __restrict__ w, uint32_t* __restrict__ out,
                            uint32_t skip_longer_than) {
    uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const uint32_t k0 = row_ptr[row], k1 = row_ptr[row + 1];
    if (skip_longer_than && k1 - k0 > skip_longer_than) return;
    Fp<P> acc = fp_zero<P>();
    for (uint32_t k = k0; k < k1; ++k) {
        Fp<P> v = load_fr<P>(vals + (size_t)k * P::W);
        Fp<P> x = load_fr<P>(w + (size_t)cols[k] * P::W);
        acc = fp_add<P>(acc, fp_mul<P>(v, x));  // v*x/R
    }
    store_fr<P>(out + row * P::W, fp_reduce_full<P>(fp_mul<P>(acc, fp_const<P>(P::R2))));
}

// Long rows, two launches and no cross-workgroup hand-off: the host cuts every long row into work items [k0, k1) of at
// most a few thousand entries (static per matrix); one workgroup per item leaves its partial sum (still v*x/R form,
// semi-reduced) in `partials`; one workgroup per long row then adds up its items and writes the canonical result.
constexpr int SPMV_LONG_THREADS = 256;

template <class P>
__device__ __forceinline__ Fp<P> block_sum_fr(Fp<P> acc, uint32_t* sh) {
    constexpr int N = P::N;
    const uint32_t tid = threadIdx.x;
    for (uint32_t off = SPMV_LONG_THREADS / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int l = 0; l < N; ++l) sh[l * SPMV_LONG_THREADS + tid] = acc.v[l];
        __syncthreads();
        if (tid < off) {
            Fp<P> o;
#pragma unroll
            for (int l = 0; l < N; ++l) o.v[l] = sh[l * SPMV_LONG_THREADS + tid + off];
            acc = fp_add<P>(acc, o);
        }
        __syncthreads();
    }
    return acc;
}

template <class P>
__global__ __launch_bounds__(SPMV_LONG_THREADS) void spmv_items_kernel(const uint32_t* __restrict__ items /* (k0, k1) pairs */,
                                                                       const uint32_t* __restrict__ cols, const uint32_t* __restrict__ vals,
                                                                       const uint32_t* __restrict__ w, uint32_t* __restrict__ partials) {
    __shared__ uint32_t sh[P::N * SPMV_LONG_THREADS];
    const uint32_t k0 = items[2 * blockIdx.x], k1 = items[2 * blockIdx.x + 1];
    Fp<P> acc = fp_zero<P>();
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += SPMV_LONG_THREADS) {
        Fp<P> v = load_fr<P>(vals + (size_t)k * P::W);
        Fp<P> x = load_fr<P>(w + (size_t)cols[k] * P::W);
        acc = fp_add<P>(acc, fp_mul<P>(v, x));
    }
    acc = block_sum_fr<P>(acc, sh);
    if (threadIdx.x == 0) store_fr<P>(partials + (size_t)blockIdx.x * P::W, acc);
}

template <class P>
__global__ __launch_bounds__(SPMV_LONG_THREADS) void spmv_rows_kernel(const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ item_ptr,
                                                                      const uint32_t* __restrict__ partials, uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[P::N * SPMV_LONG_THREADS];
    const uint32_t i0 = item_ptr[blockIdx.x], i1 = item_ptr[blockIdx.x + 1];
    Fp<P> acc = fp_zero<P>();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += SPMV_LONG_THREADS) acc = fp_add<P>(acc, load_fr<P>(partials + (size_t)i * P::W));
    acc = block_sum_fr<P>(acc, sh);
    if (threadIdx.x == 0) store_fr<P>(out + (size_t)long_rows[blockIdx.x] * P::W, fp_reduce_full<P>(fp_mul<P>(acc, fp_const<P>(P::R2))));
}

// QAP tail: flag |= (lo[i] + hi[i] - w[i] != 0)
template <class P>
__global__ void qap_check_kernel(uint64_t n, const uint32_t* lo, const uint32_t* hi, const uint32_t* w, int* nonzero) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> r = fp_sub<P>(fp_add<P>(load_fr<P>(lo + i * P::W), load_fr<P>(hi + i * P::W)), load_fr<P>(w + i * P::W));
    if (!fp_is_zero<P>(r)) atomicOr(nonzero, 1);
}

// ---- host side ------------------------------------------------------------------------------

struct TwiddleSet {
    uint32_t* fwd = nullptr;
    uint32_t* inv = nullptr;
    int stages = 0;  // rows 0 .. stages-1 are built (enough for transforms up to 2^stages)
};

static std::mutex g_tw_mutex;
static std::map<int, TwiddleSet> g_twiddles;          // per curve
static std::vector<void*> g_tw_retired;               // smaller tables replaced by a bigger one: other streams may still read them

template <class P>
static Fp<P> host_root(int log_order, bool inverse) {
    // primitive 2^log_order-th root of unity (or its inverse)
    Fp<P> w = fp_const<P>(inverse ? P::ROOT_INV : P::ROOT);
    for (int k = 0; k < P::TWO_ADICITY - log_order; ++k) w = fp_sqr<P>(w);
    return w;
}

template <class P>
static int get_twiddles(int curve, int log_n, TwiddleSet* out, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    TwiddleSet& cur = g_twiddles[curve];
    if (cur.stages >= log_n) { *out = cur; return ZK_OK; }
    TwiddleSet ts;
    ts.stages = log_n < 16 ? 16 : log_n;  // small transforms share one 2 MiB table
    if (ts.stages > P::TWO_ADICITY) ts.stages = P::TWO_ADICITY;
    const size_t bytes = (((size_t)1 << ts.stages) - 1) * TW_WORDS * 4;
    ZK_HIP(hipMalloc(&ts.fwd, bytes));
    ZK_HIP(hipMalloc(&ts.inv, bytes));
    for (int s = 0; s < ts.stages; ++s) {
        const uint32_t count = 1u << s;
        const size_t off = ((size_t)count - 1) * TW_WORDS;
        hipLaunchKernelGGL(twiddle9_kernel<P>, dim3((count + 255) / 256), dim3(256), 0, stream, ts.fwd + off, host_root<P>(s + 1, false), count);
        hipLaunchKernelGGL(twiddle9_kernel<P>, dim3((count + 255) / 256), dim3(256), 0, stream, ts.inv + off, host_root<P>(s + 1, true), count);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream));
    if (cur.fwd) { g_tw_retired.push_back(cur.fwd); g_tw_retired.push_back(cur.inv); }
    cur = ts;
    *out = ts;
    return ZK_OK;
}

static void free_scratch();
static void free_coset_tables();
static void free_uv_events();
static void free_twiddles() {
    free_uv_events();
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    for (auto& kv : g_twiddles) {
        (void)hipFree(kv.second.fwd);
        (void)hipFree(kv.second.inv);
    }
    g_twiddles.clear();
    for (void* p : g_tw_retired) (void)hipFree(p);
    g_tw_retired.clear();
    free_coset_tables();
}

// Scratch vectors of the transform, one set per stream (work on a stream is ordered, so reuse is safe); grow-only.
struct NttScratch {
    uint32_t* ptr = nullptr;
    size_t bytes = 0;
};
static std::mutex g_scratch_mutex;
static std::map<hipStream_t, NttScratch> g_scratch;

static int get_scratch(hipStream_t stream, size_t bytes, uint32_t** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    NttScratch& sc = g_scratch[stream];
    if (sc.bytes < bytes) {
        if (sc.ptr) {
            ZK_HIP(hipStreamSynchronize(stream));
            (void)hipFree(sc.ptr);
            sc.ptr = nullptr;
            sc.bytes = 0;
        }
        ZK_HIP(hipMalloc(&sc.ptr, bytes));
        sc.bytes = bytes;
    }
    *out = sc.ptr;
    return ZK_OK;
}

static void free_scratch() {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    for (auto& kv : g_scratch) (void)hipFree(kv.second.ptr);
    g_scratch.clear();
}

// Pass plan: ceil(log_n / 8) passes, the stages split as evenly as possible with the larger digits LAST (the later
// passes share one twiddle among the 2^q elements of a run and read small table rows; the first pass streams one twiddle
// per butterfly).  2^22 = 7 + 7 + 8 stages on tiles of 128 x 16, 128 x 16 and 256 x 8 elements: every global access is a
// run of 256 or 512 bytes.  Data flow: d -> scratch A -> scratch B -> ... -> d (a single pass runs in place: one
// workgroup holds the whole vector).
// what qap_h_dev_impl folds into a transform: NttFuse for the kernel, plus the vectors the first pass reads from (src, instead
// of d) and the last pass writes to (dst, instead of d)
template <class P>
struct NttPlanFuse {
    const uint32_t* src = nullptr;
    uint32_t* dst = nullptr;
    NttFuse k;
    bool own_scale = false;   // an inverse transform multiplies by `scale` (canonical Montgomery form) instead of 1/N
    Fp<P> scale;
};

template <class P>
static int ntt_dev_impl(int curve, int inverse, int log_n, uint32_t* d, hipStream_t stream, const NttPlanFuse<P>* fuse = nullptr) {
    if (log_n < 0 || log_n > P::TWO_ADICITY || log_n > 30) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    if (log_n == 0) return ZK_OK;
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(P::N * ((1u << NTT_TILE_LOG) + 32) * 4)));
        attr_set = true;
    }
    TwiddleSet ts;
    int rc = get_twiddles<P>(curve, log_n, &ts, stream);
    if (rc) return rc;
    const uint32_t* tw = inverse ? ts.inv : ts.fwd;
    const size_t bytes = ((size_t)1 << log_n) * TW_WORDS * 4;   // a scratch vector: nine raw limb words per element
    Fp<P> scale = fp_one<P>();
    if (inverse) {
        uint32_t nn[P::W] = {0};
        nn[0] = 1u << log_n;
        scale = fp_reduce_full<P>(fp_inv<P>(fp_from_canonical<P>(nn)));   // canonical: the last pass multiplies values < 9p by it
        if (fuse && fuse->own_scale) scale = fuse->scale;
    }
    const int passes = log_n <= NTT_TILE_LOG ? 1 : (log_n + NTT_MAX_DIGIT - 1) / NTT_MAX_DIGIT;
    uint32_t* scratch = nullptr;
    if (passes > 1 && (rc = get_scratch(stream, bytes * (passes > 2 ? 2 : 1), &scratch))) return rc;
    uint32_t* bufs[2] = {scratch, scratch ? scratch + (bytes / 4) : nullptr};
    const uint32_t* src = fuse && fuse->src ? fuse->src : d;
    uint32_t* final_dst = fuse && fuse->dst ? fuse->dst : d;
    int rem = log_n, which = 0;
    // Stages per pass.  An odd count costs a radix-2 step (616 instructions for one stage of four elements against 1054 for the two
    // stages of a radix-4 step), so odd digits are paired up into even ones where the sum allows it, and the smallest digit goes in
    // the middle (2^22: 8 + 6 + 8 runs 0.557 ms, 8 + 8 + 6 0.564, 6 + 8 + 8 0.571, the balanced 7 + 7 + 8 0.580-0.592)
    int digits[8] = {0};
    {
        for (int i = 0, r = log_n; i < passes; ++i) { digits[i] = r / (passes - i); r -= digits[i]; }
        for (;;) {
            int a = -1, b = -1;
            for (int i = 0; i < passes; ++i) if (digits[i] & 1) { if (a < 0) a = i; else if (b < 0) b = i; }
            if (b < 0) break;
            if (digits[a] < NTT_MAX_DIGIT) { ++digits[a]; --digits[b]; }
            else if (digits[b] < NTT_MAX_DIGIT) { ++digits[b]; --digits[a]; }
            else break;
        }
        std::sort(digits, digits + passes, [](int x, int y) { return x > y; });
        if (passes >= 3) {   // smallest to the middle: rotate it in from the end
            const int small = digits[passes - 1], mid = passes / 2;
            for (int i = passes - 1; i > mid; --i) digits[i] = digits[i - 1];
            digits[mid] = small;
        }
    }
    for (int i = 0; i < passes; ++i) {
        const int m = digits[i];
        const int rest_bits = log_n - m;
        const int q = passes == 1 ? 0 : std::min(NTT_TILE_LOG - m, rest_bits);
        const bool last = i == passes - 1;
        uint32_t* dst = last ? final_dst : bufs[which];
        const uint32_t tile = 1u << (m + q);
        const size_t lds_bytes = (size_t)P::N * (tile + 32) * 4;
        NttFuse kf;
        if (fuse && i == 0) kf.in_v = fuse->k.in_v;
        if (fuse && last) { kf.scale_tab = fuse->k.scale_tab; kf.out2 = fuse->k.out2; kf.add_tab = fuse->k.add_tab; }
        hipLaunchKernelGGL(ntt_pass_kernel<P>, dim3(1u << (rest_bits - q)), dim3(NTT_THREADS), lds_bytes, stream, src, dst, tw, log_n, rem, m, q,
                           i > 0 ? 1 : 0, last ? 1 : 0, scale, (last && inverse) ? 1 : 0, kf);
        src = dst;
        which ^= 1;
        rem -= m;
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int coset_scale_impl(int curve, int inverse, int log_n, uint32_t* d, hipStream_t stream) {
    if (log_n == 0) return ZK_OK;
    TwiddleSet ts;
    int rc = get_twiddles<P>(curve, log_n, &ts, stream);
    if (rc) return rc;
    uint32_t n = 1u << log_n;
    hipLaunchKernelGGL(coset_scale_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, stream, d, inverse ? ts.inv : ts.fwd, n);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int vec_op_dev_impl(int op, uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t stream) {
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(vec_op_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, n, a, b, out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int ntt_host_impl(int curve, int inverse, int coset, uint64_t n_in, const uint64_t* in, uint64_t size, uint64_t* out) {
    uint64_t n = next_pow2_u64(size == 0 ? 1 : size);
    int log_n = log2_u64(n);
    if (log_n > P::TWO_ADICITY || log_n > 30) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    const size_t eb = P::W * 4;
    uint32_t* d = nullptr;
    // An input longer than the domain is TRUNCATED to the domain size: the reference hands the raw slice to
    // EvaluationDomain::fft / ifft (polynomial.rs:541-542,567-568), whose radix-2 fft_in_place / ifft_in_place start with
    // `coeffs.resize(self.size(), zero)` [ark-poly 0.4.2, not vendored: parity unpinned, see DESIGN.md section 2]; only
    // DensePolynomial::evaluate_over_domain folds modulo X^n - 1, and the reference does not call it on this path.
    if (n_in > n) n_in = n;
    const uint64_t alloc = n;
    ZK_ALLOC(&d, alloc * eb);
    int rc = ZK_OK;
    do {
        if (hipMemset(d, 0, alloc * eb) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemset failed"); break; }
        if (n_in && hipMemcpy(d, in, n_in * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((alloc + 255) / 256)), dim3(256), 0, 0, alloc, d);
        if (coset && !inverse) { rc = coset_scale_impl<P>(curve, 0, log_n, d, 0); if (rc) break; }
        rc = ntt_dev_impl<P>(curve, inverse, log_n, d, 0);
        if (rc) break;
        if (coset && inverse) { rc = coset_scale_impl<P>(curve, 1, log_n, d, 0); if (rc) break; }
        if (hipMemcpy(out, d, n * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    dev_free_cached(d);
    return rc;
}

template <class P>
static int vec_op_host_impl(int op, uint64_t size, uint64_t n_a, const uint64_t* a, uint64_t n_b, const uint64_t* b, uint64_t* out) {
    if (size == 0) return ZK_OK;
    const size_t eb = P::W * 4;
    uint32_t* d = nullptr;
    ZK_ALLOC(&d, 2 * size * eb);
    int rc = ZK_OK;
    do {
        if (hipMemset(d, 0, 2 * size * eb) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemset failed"); break; }
        uint64_t ca = n_a < size ? n_a : size, cb = n_b < size ? n_b : size;
        if (ca && hipMemcpy(d, a, ca * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        if (cb && hipMemcpy(d + size * P::W, b, cb * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((2 * size + 255) / 256)), dim3(256), 0, 0, 2 * size, d);
        rc = vec_op_dev_impl<P>(op, size, d, d + size * P::W, d, 0);
        if (rc) break;
        if (hipMemcpy(out, d, size * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    dev_free_cached(d);
    return rc;
}

template <class P>
static int div_vanishing_host_impl(uint64_t n, uint64_t len, const uint64_t* coeffs, uint64_t* q, uint64_t* rem, int* rem_is_zero) {
    if (n == 0) return fail(ZK_ERR_ARG, "vanishing polynomial of an empty domain");
    *rem_is_zero = 1;
    if (len == 0) return ZK_OK;
    const size_t eb = P::W * 4;
    uint64_t qlen = len > n ? len - n : 0, top = len < n ? len : n;
    uint32_t *dc = nullptr, *dq = nullptr, *dr = nullptr;
    int* dflag = nullptr;
    ZK_ALLOC(&dc, len * eb);
    int rc = ZK_OK;
    do {
        if (dev_alloc_cached((void**)&dq, (qlen ? qlen : 1) * eb) != ZK_OK || dev_alloc_cached((void**)&dr, top * eb) != ZK_OK ||
            dev_alloc_cached((void**)&dflag, sizeof(int)) != ZK_OK) { rc = fail(ZK_ERR_HIP, "hipMalloc failed"); break; }
        if (hipMemcpy(dc, coeffs, len * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        (void)hipMemset(dflag, 0, sizeof(int));
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, len, dc);
        uint64_t lim = qlen > top ? qlen : top;
        hipLaunchKernelGGL(div_vanishing_kernel<P>, dim3((unsigned)((lim + 255) / 256)), dim3(256), 0, 0, n, len, dc, dq, dr, dflag);
        int flag = 0;
        if (hipMemcpy(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        if (qlen && hipMemcpy(q, dq, qlen * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        if (hipMemcpy(rem, dr, top * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        *rem_is_zero = flag ? 0 : 1;
    } while (0);
    dev_free_cached(dc); dev_free_cached(dq); dev_free_cached(dr); dev_free_cached(dflag);
    return rc;
}

// ---- QAP quotient on a coset ------------------------------------------------------------------------------------
// h = (u v - w) / (X^n - 1).  The reference multiplies u and v over the doubled domain and divides the 2n coefficients
// (qap.py:51-67 -> mul_over_fft, divide_by_vanishing_poly).  Write u v = P_lo + X^n P_hi (both parts of degree < n): modulo X^n - 1
// that is w = P_lo + P_hi, and the quotient is h = P_hi.  On the coset g H with g = the primitive 2n-th root of unity (g^n = -1)
// the n values u(g w^i) v(g w^i) interpolate u v mod (X^n + 1) = P_lo - P_hi =: d, so
//                                  h = (w - d) / 2
// -- SIX size-n transforms (round 4; seven until then, on the coset 5 H with a point-wise division by the constant g^n - 1): the
// three inverse transforms of a, b, c, two forward transforms of the g^i-shifted u and v, and one inverse transform of their
// point-wise product.  And nothing between them (NttFuse): the shift g^i rides on the 1/N product of the inverse transforms'
// last pass (u and v leave it twice: plain, for their MSMs, and shifted), w is read from c and written as w / 2, the product
// u v is formed by the FIRST pass of the last inverse transform as it loads, and its LAST pass multiplies by -g^-i / (2N) and adds
// w / 2: it writes h.  The quotient is exact iff a_i b_i = c_i on the domain itself, which is checked directly on the evaluation
// vectors (the reference's "remainder must be zero", qap.py:68-69).
struct QapTabs {
    uint32_t *fwd = nullptr, *inv = nullptr;   // g^i / N and -g^-i R / (2N) as canonical Montgomery representatives, g = w_2n
};
static std::map<std::pair<int, int>, QapTabs> g_qap_tabs;   // (curve, log_n), under g_tw_mutex
static void free_coset_tables() {  // caller holds g_tw_mutex
    for (auto& kv : g_qap_tabs) { (void)hipFree(kv.second.fwd); (void)hipFree(kv.second.inv); }
    g_qap_tabs.clear();
}

// row log_n of the stage-major twiddle tables holds w_2n^(+-i), i < n, as nine-word Montgomery representatives
template <class P>
__global__ void qap_tabs_kernel(uint32_t* __restrict__ fwd, uint32_t* __restrict__ inv, uint64_t n, const uint32_t* __restrict__ row_f,
                                const uint32_t* __restrict__ row_i, Fp<P> inv_n, Fp<P> c_inv) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fr<P>(fwd + i * P::W, fp_reduce_full<P>(fp_mul<P>(load_limbs9<P>(row_f + i * TW_WORDS), inv_n)));   // (g^i R)(R / N) / R
    store_fr<P>(inv + i * P::W, fp_reduce_full<P>(fp_mul<P>(load_limbs9<P>(row_i + i * TW_WORDS), c_inv)));   // (g^-i R)(-R^2 / 2N) / R
}

template <class P>
static int get_qap_tabs(int curve, int log_n, QapTabs* out, hipStream_t stream) {
    TwiddleSet ts;
    int rc = get_twiddles<P>(curve, log_n + 1, &ts, stream);   // takes g_tw_mutex itself
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto it = g_qap_tabs.find({curve, log_n});
    if (it != g_qap_tabs.end()) { *out = it->second; return ZK_OK; }
    const uint64_t n = 1ull << log_n;
    QapTabs t;
    ZK_HIP(hipMalloc(&t.fwd, n * P::W * 4));
    ZK_HIP(hipMalloc(&t.inv, n * P::W * 4));
    uint32_t nn[P::W] = {0};
    nn[0] = (uint32_t)n;
    const Fp<P> inv_n = fp_inv<P>(fp_from_canonical<P>(nn));                            // R / N
    const Fp<P> half = fp_inv<P>(fp_add<P>(fp_one<P>(), fp_one<P>()));                  // R / 2
    const Fp<P> c_inv = fp_mul<P>(fp_neg<P>(fp_mul<P>(inv_n, half)), fp_const<P>(P::R2));   // -(R / 2N) R^2 / R = -R^2 / (2N)
    const size_t row = (((size_t)1 << log_n) - 1) * TW_WORDS;
    hipLaunchKernelGGL(qap_tabs_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t.fwd, t.inv, n, ts.fwd + row, ts.inv + row, inv_n, c_inv);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream));
    g_qap_tabs[{curve, log_n}] = t;
    *out = t;
    return ZK_OK;
}

// flag |= (a[i] * b[i] != c[i]) on the domain: the witness satisfies the constraints iff this never fires
template <class P>
__global__ void qap_eval_check_kernel(uint64_t n, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                      const uint32_t* __restrict__ c, int* nonzero) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // mont(a, b) = a b / R against mont(c, 1) = c / R
    Fp<P> one = fp_zero<P>();
    one.v[0] = 1;
    Fp<P> r = fp_sub<P>(fp_mul<P>(load_fr<P>(a + i * P::W), load_fr<P>(b + i * P::W)), fp_mul<P>(load_fr<P>(c + i * P::W), one));
    if (!fp_is_zero<P>(r)) atomicOr(nonzero, 1);
}

static std::mutex g_uv_mutex;
static std::map<hipStream_t, hipEvent_t> g_uv_events;  // one "u and v are final" event per stream (zk_qap_h_dev_begin)

static int uv_event_for(hipStream_t stream, hipEvent_t* out) {
    std::lock_guard<std::mutex> lock(g_uv_mutex);
    auto it = g_uv_events.find(stream);
    if (it == g_uv_events.end()) {
        hipEvent_t e = nullptr;
        ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        it = g_uv_events.emplace(stream, e).first;
    }
    *out = it->second;
    return ZK_OK;
}
static void free_uv_events() {
    std::lock_guard<std::mutex> lock(g_uv_mutex);
    for (auto& kv : g_uv_events) (void)hipEventDestroy(kv.second);
    g_uv_events.clear();
}

// divisible == nullptr: enqueue only (zk_qap_h_dev_begin); uv_ready (optional) is recorded once u and v are final
template <class P>
static int qap_h_dev_impl(int curve, int log_n, uint32_t* a_u, uint32_t* b_v, const uint32_t* c, uint32_t* h,
                          uint32_t* work, int* divisible, hipStream_t stream, hipEvent_t uv_ready = nullptr) {
    // the coset generator is the primitive 2n-th root of unity: one level of two-adicity above the domain
    if (log_n + 1 > P::TWO_ADICITY || log_n > 30) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    const uint64_t n = 1ull << log_n;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    uint32_t* W0 = work;                     // w / 2 (coefficients)
    uint32_t* U1 = work + n * P::W;          // u g^i, then u on the coset
    uint32_t* V1 = work + 2 * n * P::W;      // v g^i, then v on the coset
    int* dflag = reinterpret_cast<int*>(work + 3 * n * P::W);
    QapTabs qt;
    int rc;
    if ((rc = get_qap_tabs<P>(curve, log_n, &qt, stream))) return rc;
    ZK_HIP(hipMemsetAsync(dflag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(qap_eval_check_kernel<P>, dim3(blocks), dim3(256), 0, stream, n, a_u, b_v, c, dflag);
    NttPlanFuse<P> f;
    f.k.scale_tab = qt.fwd;
    f.k.out2 = U1;
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, a_u, stream, &f))) return rc;  // u, and u g^i into U1
    f.k.out2 = V1;
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, b_v, stream, &f))) return rc;  // v, and v g^i into V1
    if (uv_ready) ZK_HIP(hipEventRecord(uv_ready, stream));
    NttPlanFuse<P> fw;   // w / 2 straight from c: the inverse transform's scale is 1 / (2N)
    fw.src = c;
    fw.own_scale = true;
    {
        uint32_t nn[P::W] = {0};
        nn[0] = (uint32_t)n;
        const Fp<P> two_n = fp_add<P>(fp_from_canonical<P>(nn), fp_from_canonical<P>(nn));
        fw.scale = fp_reduce_full<P>(fp_inv<P>(two_n));
    }
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, W0, stream, &fw))) return rc;
    if ((rc = ntt_dev_impl<P>(curve, 0, log_n, U1, stream))) return rc;       // u on g H
    if ((rc = ntt_dev_impl<P>(curve, 0, log_n, V1, stream))) return rc;       // v on g H
    NttPlanFuse<P> fq;   // h = w / 2 + iNTT_unscaled(u v / R) * (-g^-i R / 2N)
    fq.k.in_v = V1;
    fq.k.scale_tab = qt.inv;
    fq.k.add_tab = W0;
    fq.dst = h;
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, U1, stream, &fq))) return rc;
    ZK_HIP(hipGetLastError());
    if (!divisible) return ZK_OK;
    int flag = 0;
    ZK_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, stream));
    ZK_HIP(hipStreamSynchronize(stream));
    *divisible = flag ? 0 : 1;
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_ntt(int curve, int inverse, int coset, uint64_t n_in, const uint64_t* in, uint64_t size, uint64_t* out) {
#define CALL(P) return ntt_host_impl<P>(curve, inverse, coset, n_in, in, size, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_op(int curve, int op, uint64_t size, uint64_t n_a, const uint64_t* a, uint64_t n_b, const uint64_t* b, uint64_t* out) {
    if (op < 0 || op > 2) return fail(ZK_ERR_ARG, "unknown vector op");
#define CALL(P) return vec_op_host_impl<P>(op, size, n_a, a, n_b, b, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_poly_div_vanishing(int curve, uint64_t n, uint64_t len, const uint64_t* coeffs, uint64_t* q, uint64_t* rem, int* rem_is_zero) {
#define CALL(P) return div_vanishing_host_impl<P>(n, len, coeffs, q, rem, rem_is_zero)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_ntt_dev(int curve, int inverse, int log_n, void* d_data, void* stream) {
#define CALL(P) return ntt_dev_impl<P>(curve, inverse, log_n, (uint32_t*)d_data, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_op_dev(int curve, int op, uint64_t n, const void* d_a, const void* d_b, void* d_out, void* stream) {
    if (op < 0 || op > 2) return fail(ZK_ERR_ARG, "unknown vector op");
#define CALL(P) return vec_op_dev_impl<P>(op, n, (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t*)d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_qap_h_dev(int curve, int log_n, void* d_a_u, void* d_b_v, const void* d_c, void* d_h, void* d_work, int* divisible, void* stream) {
#define CALL(P) return qap_h_dev_impl<P>(curve, log_n, (uint32_t*)d_a_u, (uint32_t*)d_b_v, (const uint32_t*)d_c, (uint32_t*)d_h, (uint32_t*)d_work, divisible, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_qap_h_dev_begin(int curve, int log_n, void* d_a_u, void* d_b_v, const void* d_c, void* d_h, void* d_work, void* stream, void** uv_ready) {
    hipEvent_t ev = nullptr;
    if (uv_ready) {
        int rc = uv_event_for((hipStream_t)stream, &ev);
        if (rc) return rc;
        *uv_ready = (void*)ev;
    }
#define CALL(P) return qap_h_dev_impl<P>(curve, log_n, (uint32_t*)d_a_u, (uint32_t*)d_b_v, (const uint32_t*)d_c, (uint32_t*)d_h, (uint32_t*)d_work, nullptr, (hipStream_t)stream, ev)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_qap_uv_dev(int curve, int log_n, void* d_a_u, void* d_b_v, void* stream, void** uv_ready) {
    hipEvent_t ev = nullptr;
    if (uv_ready) {
        int rc = uv_event_for((hipStream_t)stream, &ev);
        if (rc) return rc;
        *uv_ready = (void*)ev;
    }
    int rc = ZK_OK;
    if (d_a_u && (rc = zk_ntt_dev(curve, 1, log_n, d_a_u, stream))) return rc;
    if (d_b_v && (rc = zk_ntt_dev(curve, 1, log_n, d_b_v, stream))) return rc;
    if (ev) ZK_HIP(hipEventRecord(ev, (hipStream_t)stream));
    return ZK_OK;
}

int zk_qap_h_dev_end(int curve, int log_n, const void* d_work, int* divisible, void* stream) {
    if (curve != ZK_CURVE_BN254 && curve != ZK_CURVE_BLS12_381) return fail(ZK_ERR_ARG, "unknown curve");
    if (log_n < 0 || log_n > 32) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    const int* dflag = reinterpret_cast<const int*>(reinterpret_cast<const uint32_t*>(d_work) + ((size_t)3 << log_n) * 8);  // both scalar fields: 8 words
    int flag = 0;
    ZK_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ZK_HIP(hipStreamSynchronize((hipStream_t)stream));
    *divisible = flag ? 0 : 1;
    return ZK_OK;
}

int zk_spmv_dev(int curve, uint64_t n_rows, const void* d_row_ptr, const void* d_cols, const void* d_vals,
                const void* d_w, void* d_out, uint32_t skip_longer_than, void* stream) {
    if (n_rows == 0) return ZK_OK;
#define CALL(P)                                                                                                  \
    {                                                                                                            \
        hipLaunchKernelGGL(spmv_kernel<P>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, \
                           n_rows, (const uint32_t*)d_row_ptr, (const uint32_t*)d_cols, (const uint32_t*)d_vals,  \
                           (const uint32_t*)d_w, (uint32_t*)d_out, skip_longer_than);                             \
        ZK_HIP(hipGetLastError());                                                                               \
        return ZK_OK;                                                                                            \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_spmv_long_dev(int curve, uint64_t n_long, const void* d_long_rows, const void* d_item_ptr, uint64_t n_items,
                     const void* d_items, const void* d_cols, const void* d_vals, const void* d_w, void* d_partials,
                     void* d_out, void* stream) {
    if (n_long == 0 || n_items == 0) return ZK_OK;
#define CALL(P)                                                                                                          \
    {                                                                                                                    \
        hipLaunchKernelGGL(spmv_items_kernel<P>, dim3((unsigned)n_items), dim3(SPMV_LONG_THREADS), 0, (hipStream_t)stream, \
                           (const uint32_t*)d_items, (const uint32_t*)d_cols, (const uint32_t*)d_vals,                   \
                           (const uint32_t*)d_w, (uint32_t*)d_partials);                                                 \
        hipLaunchKernelGGL(spmv_rows_kernel<P>, dim3((unsigned)n_long), dim3(SPMV_LONG_THREADS), 0, (hipStream_t)stream,   \
                           (const uint32_t*)d_long_rows, (const uint32_t*)d_item_ptr, (const uint32_t*)d_partials,       \
                           (uint32_t*)d_out);                                                                            \
        ZK_HIP(hipGetLastError());                                                                                       \
        return ZK_OK;                                                                                                    \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_powers_dev(int curve, uint64_t n, const uint64_t* g, void* d_out, void* stream) {
    if (n == 0) return ZK_OK;
    if (n > (1ull << 32)) return fail(ZK_ERR_ARG, "too many powers");
#define CALL(P)                                                                                                        \
    {                                                                                                                  \
        Fp<P> gm = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(g));                                         \
        hipLaunchKernelGGL(powers_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,     \
                           (uint32_t*)d_out, gm, (uint32_t)n);                                                         \
        ZK_HIP(hipGetLastError());                                                                                     \
        return ZK_OK;                                                                                                  \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_canon_dev(int curve, uint64_t n, void* d_x, void* stream) {
    if (n == 0) return ZK_OK;
#define CALL(P)                                                                                                       \
    {                                                                                                                 \
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,  \
                           (uint32_t*)d_x);                                                                           \
        ZK_HIP(hipGetLastError());                                                                                    \
        return ZK_OK;                                                                                                 \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

void zk_ntt_free_cache(void) {
    free_twiddles();
    free_scratch();
}

}  // extern "C"
