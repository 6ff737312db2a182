// msm_impl.cuh -- Pippenger bucket multi-scalar multiplication over G1/G2 of BN254 and BLS12-381 on
// gfx950, plus batched scalar multiplication.
//
// Stands in for multiscalar_mul_g1/_g2 (reference src/bn254/curve.rs:356-392,
// src/bls12_381/curve.rs:366-402 -> ark-ec 0.4.2 VariableBaseMSM::msm) and
// batch_multi_scalar_g1/_g2 (src/bn254/curve.rs:326-354).  The result of an MSM is a single
// group element, returned as its unique affine representative, so any correct bucket method
// is bit-identical with the reference's.
//
// Pipeline (one stream, no host round trip until the tail):
//   1. digits      scalar -> signed c-bit window digits (bias trick: s + sum 2^(c-1) 2^(wc),
//                  then plain bit fields), 2 B per (window, scalar), coalesced.
//   2. histogram   one workgroup per (window, chunk): 2^(c-1) counters live in LDS (128 KiB at
//                  c = 16), LDS atomics only; the table is flushed with coalesced stores.
//   3. prefix/scan per-bucket prefix over chunks, bucket offsets, segment offsets.
//   4. scatter     same grid as 2: LDS cursors, point references written bucket-sorted.
//   5. accumulate  THE dominant kernel: one lane per bucket *segment* (<= S entries, so skewed
//                  scalar distributions cannot starve a wave), XYZZ accumulator in registers,
//                  bases gathered as whole 64..192-byte rows.
//   6. combine     per-bucket sum of its segment partials.
//   7. reduce      sum_b (b+1) B_b through the split b = hi*C + lo: wave-per-row / wave-per-column
//                  sums (shuffle trees), then a log-depth suffix-scan on <=256 points per array.
//   8. host tail   3 points per window come back; Horner over windows and the affine
//                  conversion (one inversion) run on the host -- a lone GPU wave needs ~1.3 us per
//                  field multiplication, the host ~50 ns.
#pragma once
#include <algorithm>
#include <chrono>
#include <memory>
#include <vector>
#include "common.cuh"
#include "msm_plan.h"
#include "pair.cuh"
#include "setup_impl.cuh"
#include "codec.cuh"
#include "glv_params.h"

namespace zkmi {


// ---- device load/store of field elements / points (packed u32 words, 16-byte vectors) ------

template <int WORDS>
__device__ __forceinline__ void load_words(uint32_t* dst, const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) {
        uint4 t = q[i];
        dst[4 * i] = t.x; dst[4 * i + 1] = t.y; dst[4 * i + 2] = t.z; dst[4 * i + 3] = t.w;
    }
}
template <int WORDS>
__device__ __forceinline__ void store_words(uint32_t* dst, const uint32_t* src) {
    uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < WORDS / 4; ++i) q[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}

// Points in memory are packed 32-bit words (LIMBS per coordinate); registers hold 29-bit limbs.
template <class F>
__device__ __forceinline__ Affine<F> load_affine(const uint32_t* p) {
    uint32_t w[2 * F::LIMBS];
    load_words<2 * F::LIMBS>(w, p);
    return {F::load(w), F::load(w + F::LIMBS)};
}
template <class F>
__device__ __forceinline__ void store_affine(uint32_t* p, const Affine<F>& a) {
    uint32_t w[2 * F::LIMBS];
    F::store(w, a.x);
    F::store(w + F::LIMBS, a.y);
    store_words<2 * F::LIMBS>(p, w);
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const uint32_t* p) {
    uint32_t w[4 * F::LIMBS];
    load_words<4 * F::LIMBS>(w, p);
    return {F::load(w), F::load(w + F::LIMBS), F::load(w + 2 * F::LIMBS), F::load(w + 3 * F::LIMBS)};
}
template <class F>
__device__ __forceinline__ void store_xyzz(uint32_t* p, const XYZZ<F>& a) {
    uint32_t w[4 * F::LIMBS];
    F::store(w, a.X);
    F::store(w + F::LIMBS, a.Y);
    F::store(w + 2 * F::LIMBS, a.ZZ);
    F::store(w + 3 * F::LIMBS, a.ZZZ);
    store_words<4 * F::LIMBS>(p, w);
}
template <class F>
static XYZZ<F> load_xyzz_host(const uint32_t* w) {
    return {F::load(w), F::load(w + F::LIMBS), F::load(w + 2 * F::LIMBS), F::load(w + 3 * F::LIMBS)};
}

// register-form copies (LDS staging, wave shuffles): XYZZ<F> is a plain struct of u32 registers
template <class F>
struct XyzzRegs { static constexpr int COUNT = sizeof(XYZZ<F>) / 4; };

template <class F>
__device__ __forceinline__ XYZZ<F> shfl_xor_xyzz(const XYZZ<F>& a, int mask) {
    XYZZ<F> r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) d[i] = __shfl_xor(s[i], mask, 64);
    return r;
}
template <class F>
__device__ __forceinline__ void lds_put_xyzz(uint32_t* slot, const XYZZ<F>& a) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) slot[i] = s[i];
}
template <class F>
__device__ __forceinline__ XYZZ<F> lds_get_xyzz(const uint32_t* slot) {
    XYZZ<F> r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < XyzzRegs<F>::COUNT; ++i) d[i] = slot[i];
    return r;
}

// Buckets with more than COMBINE_SMALL_MAX runs (skewed scalars: e.g. the short top window, or many equal
// scalars) are listed in big_list and reduced by a whole workgroup each instead of one lane.
constexpr uint32_t COMBINE_SMALL_MAX = 16;    // <= 16 runs: one lane adds them up
constexpr uint32_t COMBINE_WAVE_MAX = 2048;   // <= 2048 runs: one wave per bucket; above: one workgroup


// ---- G1 endomorphism (GLV) ---------------------------------------------------------------------------
// General (not fixed-base) G1 plans run the MSM over 2n points (P_i, phi(P_i)) with the two ~127-bit halves of every
// scalar, k = k1 + lambda k2: the same number of bucket additions (2n entries in half the windows), but half the bucket
// sets to reduce and half the doublings in the host tail.  Constants and the decomposition: tools/gen_glv_params.py.
template <class G> struct GlvOf { static constexpr bool OK = false; };
template <> struct GlvOf<Bn254G1> { static constexpr bool OK = true; typedef Bn254Glv P; };
template <> struct GlvOf<Bls381G1> { static constexpr bool OK = true; typedef Bls381Glv P; };

#if !defined(ZK_PART) || ZK_PART == 0  // sort-stage kernels and the plan live in part 0 only
// ---- 1. digits -----------------------------------------------------------------------------------

struct DigitBias {
    uint32_t v[13];  // bias limbs (up to 12 + 1)
};

// eight consecutive digits of a row as 32-bit values (rows are padded to 8 digits and 16-byte aligned)
template <class DIG>
__device__ __forceinline__ void load8_digits(const DIG* p, uint32_t* v);
template <>
__device__ __forceinline__ void load8_digits<uint16_t>(const uint16_t* p, uint32_t* v) {
    const uint4 pk = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
}
template <>
__device__ __forceinline__ void load8_digits<uint32_t>(const uint32_t* p, uint32_t* v) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// DIG = uint16_t for windows up to 16 bits, uint32_t for the wider windows of fixed-base plans
template <class FrP, class DIG>
__global__ void digits_kernel(const uint32_t* __restrict__ scalars, uint32_t n, uint32_t dstride, int c, int w_first, int w_count,
                              DigitBias bias, DIG* __restrict__ dig, uint32_t* __restrict__ big_count) {
    constexpr int N = FrP::W;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) big_count[0] = big_count[1] = 0;  // consumed by runs_scan_block_kernel later in the same stream
    if (i >= n) return;
    uint32_t s[N + 1];
    load_words<N>(s, scalars + (size_t)i * N);
    // Fr::from(BigUint) semantics: reduce below r
    for (int k = 0; k < 10; ++k) {
        uint32_t t[N];
        if (fp_sub_mod_raw<FrP>(t, s)) break;
#pragma unroll
        for (int l = 0; l < N; ++l) s[l] = t[l];
    }
    uint64_t carry = 0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
        uint64_t t = (uint64_t)s[l] + bias.v[l] + carry;
        s[l] = (uint32_t)t;
        carry = t >> 32;
    }
    s[N] = (uint32_t)carry + bias.v[N];
    const uint32_t mask = (1u << c) - 1;
    for (int w = w_first; w < w_first + w_count; ++w) {  // only the windows of this run (a rank's share when sharded)
        int bit = w * c;
        int word = bit >> 5, off = bit & 31;
        uint64_t two = (uint64_t)s[word];
        if (word + 1 <= N) two |= (uint64_t)s[word + 1] << 32;
        uint32_t u = (uint32_t)(two >> off) & mask;
        dig[(size_t)w * dstride + i] = (DIG)u;  // rows padded to 8 digits: 16-byte aligned vector reads
    }
}

// NA x NB words -> NA + NB words
template <int NA, int NB>
__device__ __forceinline__ void mul_words(uint32_t* out, const uint32_t* a, const uint32_t* b) {
#pragma unroll
    for (int k = 0; k < NA + NB; ++k) out[k] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        out[i + NB] = (uint32_t)carry;
    }
}
// acc (4 words) -= a * b mod 2^128
__device__ __forceinline__ void submul_lo4(uint32_t* acc, const uint32_t* a, const uint32_t* b) {
    uint32_t p[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j + i < 4; ++j) {
            const uint64_t t = (uint64_t)a[i] * b[j] + p[i + j] + carry;
            p[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
    }
    uint64_t borrow = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const uint64_t t = (uint64_t)acc[l] - p[l] - borrow;
        acc[l] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
}
// (k g + 2^319) >> 320, four words
__device__ __forceinline__ void glv_round_quotient(uint32_t* c, const uint32_t* k8, const uint32_t* g7) {
    uint32_t prod[15];
    mul_words<8, 7>(prod, k8, g7);
    uint64_t carry = 0x80000000ull;  // 2^319 = bit 31 of word 9
#pragma unroll
    for (int l = 9; l < 15; ++l) {
        const uint64_t t = (uint64_t)prod[l] + carry;
        prod[l] = (uint32_t)t;
        carry = t >> 32;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) c[l] = prod[10 + l];
}

// digits of the two halves of every scalar: entry 2i carries k1 (against P_i), entry 2i + 1 carries k2 (against phi(P_i)).
// The halves are signed; a signed value plus the bias is still a plain string of c-bit fields.
template <class FrP>
__global__ void glv_digits_kernel(const uint32_t* __restrict__ scalars, uint32_t m, uint32_t dstride, int c, int w_first, int w_count,
                                  DigitBias bias, GlvConsts K, uint16_t* __restrict__ dig, uint32_t* __restrict__ big_count) {
    constexpr int N = FrP::W;
    static_assert(N == 8, "scalar fields of 8 words");
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) big_count[0] = big_count[1] = 0;
    if (i >= m) return;
    uint32_t s[N];
    load_words<N>(s, scalars + (size_t)i * N);
    for (int k = 0; k < 10; ++k) {
        uint32_t t[N];
        if (fp_sub_mod_raw<FrP>(t, s)) break;
#pragma unroll
        for (int l = 0; l < N; ++l) s[l] = t[l];
    }
    uint32_t c1[4], c2[4];
    glv_round_quotient(c1, s, K.g1);
    glv_round_quotient(c2, s, K.g2);
    uint32_t k1[4] = {s[0], s[1], s[2], s[3]}, k2[4] = {0, 0, 0, 0};
    submul_lo4(k1, c1, K.a1);
    submul_lo4(k1, c2, K.a2);
    submul_lo4(k2, c1, K.b1);
    submul_lo4(k2, c2, K.b2);
    uint32_t t1[6], t2[6];
    {
        const uint32_t e1 = (k1[3] >> 31) ? 0xFFFFFFFFu : 0u, e2 = (k2[3] >> 31) ? 0xFFFFFFFFu : 0u;
        uint64_t ca = 0, cb = 0;
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            const uint64_t a = (uint64_t)(l < 4 ? k1[l] : e1) + bias.v[l] + ca;
            const uint64_t b = (uint64_t)(l < 4 ? k2[l] : e2) + bias.v[l] + cb;
            t1[l] = (uint32_t)a; ca = a >> 32;
            t2[l] = (uint32_t)b; cb = b >> 32;
        }
        t1[5] = t2[5] = 0;
    }
    const uint32_t mask = (1u << c) - 1;
    uint32_t* dig32 = reinterpret_cast<uint32_t*>(dig);
    for (int w = w_first; w < w_first + w_count; ++w) {
        const int bit = w * c, word = bit >> 5, off = bit & 31;
        const uint32_t u1 = (uint32_t)((((uint64_t)t1[word + 1] << 32) | t1[word]) >> off) & mask;
        const uint32_t u2 = (uint32_t)((((uint64_t)t2[word + 1] << 32) | t2[word]) >> off) & mask;
        dig32[((size_t)w * dstride >> 1) + i] = u1 | (u2 << 16);
    }
}

// ---- 2. histogram / 4. scatter ---------------------------------------------------------------------

constexpr int SORT_THREADS = 1024;

static __global__ __launch_bounds__(SORT_THREADS) void hist_kernel(const uint16_t* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                            int w_first, int nchunk, uint32_t chunk_len,
                                                            uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lds[];
    const uint32_t B = 1u << (c - 1);
    const int wl = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const int w = w_first + wl;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) lds[b] = 0;
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len;
    if (hi > n) hi = n;
    const uint16_t* d = dig + (size_t)w * dstride;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        int v = (int)d[i] - (int)B;
        if (v != 0) {
            uint32_t b = (uint32_t)(v < 0 ? -v : v) - 1;
            atomicAdd(&lds[b], 1u);
        }
    }
    __syncthreads();
    uint32_t* out = hist + (size_t)blockIdx.x * B;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) out[b] = lds[b];
}

// per bucket key: exclusive prefix over the sub-histograms of its group, total, segment count
static __global__ void prefix_kernel(uint32_t* __restrict__ hist, int group_size, uint32_t B, uint32_t n_keys,
                              uint32_t* __restrict__ total) {
    uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= n_keys) return;
    uint32_t g = key / B, b = key % B;
    uint32_t run = 0;
    for (int h = 0; h < group_size; ++h) {
        size_t idx = ((size_t)g * group_size + h) * B + b;
        uint32_t t = hist[idx];
        hist[idx] = run;
        run += t;
    }
    total[key] = run;
}

// The sorted entry list is cut into uniform segments of seg_len entries (one lane each), whatever the bucket
// sizes are.  A "run" is the part of one bucket inside one segment; bucket `key` owns
//   nruns = 1 + (last_entry / seg_len) - (first_entry / seg_len)   consecutive partial slots.
// two-level exclusive scan: blocks of 1024
constexpr int SCAN_BLOCK = 1024;
static __global__ __launch_bounds__(SCAN_BLOCK) void scan_block_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                                uint32_t* __restrict__ out, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t v = i < n ? in[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint32_t t = threadIdx.x >= (uint32_t)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n) out[i] = sh[threadIdx.x] - v;  // exclusive
    if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
}
// the run counts are computed on the fly from the bucket offsets and scanned in the same launch (first level of the
// run-offset scan); buckets with many runs go to big_list, filled from the front with wave-tier buckets and from the
// back with workgroup-tier buckets (big_count[0] / big_count[1] are the two lengths, zeroed by digits_kernel)
static __global__ __launch_bounds__(SCAN_BLOCK) void runs_scan_block_kernel(const uint32_t* __restrict__ bucket_start, uint32_t n_keys,
                                                                            uint32_t seg_len, uint32_t* __restrict__ out,
                                                                            uint32_t* __restrict__ block_sums,
                                                                            uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    __shared__ uint32_t sh[SCAN_BLOCK / 64];
    const uint32_t key = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t r = 0;
    if (key < n_keys) {
        const uint32_t s0 = bucket_start[key], s1 = bucket_start[key + 1];
        r = s1 > s0 ? 1 + (s1 - 1) / seg_len - s0 / seg_len : 0;
        if (r > COMBINE_WAVE_MAX) big_list[n_keys - 1 - atomicAdd(big_count + 1, 1u)] = key;
        else if (r > COMBINE_SMALL_MAX) big_list[atomicAdd(big_count, 1u)] = key;
    }
    // wave-level inclusive scan, then the 16 wave totals
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = r;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w2 = 0; w2 < wave; ++w2) before += sh[w2];
    if (key < n_keys) out[key] = before + incl - r;  // exclusive
    if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = before + incl;
}
// single block: exclusive scan of the block sums in place (n_blocks <= 1024 * 64)
static __global__ __launch_bounds__(SCAN_BLOCK) void scan_sums_kernel(uint32_t* sums, uint32_t n_blocks, uint32_t* grand_total) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += SCAN_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
            uint32_t t = threadIdx.x >= (uint32_t)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) sums[i] = carry + sh[threadIdx.x] - v;
        uint32_t blk_total = sh[SCAN_BLOCK - 1];
        __syncthreads();
        carry += blk_total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
static __global__ void scan_add_kernel(uint32_t* out, uint32_t n, const uint32_t* block_sums, const uint32_t* grand_total) {
    uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
    if (i == 0) out[n] = *grand_total;  // out has n + 1 entries
}

static __global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(const uint16_t* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                               int w_first, int w_count, int nchunk, uint32_t chunk_len,
                                                               int shared_buckets, uint32_t table_stride, int table_w0,
                                                               const uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ bucket_start,
                                                               uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t lds[];
    const uint32_t B = 1u << (c - 1);
    // XCD-aware block -> (window, chunk) map: workgroups are dealt round-robin over the 8 XCDs, and all chunks
    // of one window write 4-byte entries into the same cache lines (the window's bucket regions).  Putting them
    // on one XCD lets that XCD's L2 merge the partial lines instead of eight caches writing them back separately.
    int wl, chunk;
    if (shared_buckets) {
        wl = blockIdx.x / nchunk;
        chunk = blockIdx.x % nchunk;
    } else {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        wl = x + 8 * (j / nchunk);
        chunk = j % nchunk;
        if (wl >= w_count) return;
    }
    const int w = w_first + wl;
    const uint32_t* pre = hist + ((size_t)wl * nchunk + chunk) * B;
    const uint32_t* start = bucket_start + (shared_buckets ? 0 : (size_t)wl * B);
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) lds[b] = start[b] + pre[b];
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len;
    if (hi > n) hi = n;
    const uint16_t* d = dig + (size_t)w * dstride;
    // with shared buckets the point reference addresses the precomputed table row (w, i)
    const uint32_t ref_base = shared_buckets ? (uint32_t)(w - table_w0) * table_stride : 0;  // table rows start at the plan's first window
    for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        int v = (int)d[i] - (int)B;
        if (v != 0) {
            uint32_t b = (uint32_t)(v < 0 ? -v : v) - 1;
            uint32_t pos = atomicAdd(&lds[b], 1u);
            sorted[pos] = (ref_base + i) | (v < 0 ? 0x80000000u : 0u);
        }
    }
}

// ---- 2''/4''. two-level counting sort (general mode, large inputs) --------------------------------------
// The chunked scatter above writes 4-byte references to random bucket regions of its window: every 32-byte sector
// of `sorted` is touched by several workgroups at different times and goes to HBM as partial writes (8x write
// amplification, the 0.2 ms of the stage).  Here the sort is split:
//   level A  (window, chunk) workgroups partition their entries by the COARSE bin = bucket >> fine_log (128 bins per
//            window at c = 16, fine_log = 8), tile by tile through LDS, so that each bin receives coalesced runs;
//            an entry travels as one word (sign | low bucket bits | reference);
//   level B  one workgroup per (bucket set, coarse bin) sorts its entries by the low bits with LDS counters, places
//            them in an LDS copy of its contiguous slice of `sorted` and writes the slice -- and the bucket offsets --
//            with coalesced stores.
// Both levels were first written with direct 4-byte scattered stores and were bound by the L2 request rate (one
// request per entry: 0.10 + 0.07 ms at 2^20); staging the output in LDS halved them.
// Skew: when a whole wave hits one counter (many equal scalars, boolean witnesses) the increment is aggregated into
// one atomic per wave.
constexpr int FINE_LOG_MAX = 8;  // fine buckets per coarse bin = 2^fine_log, fine_log = 8 (n <= 2^23) or 7 (n <= 2^24):
                                // a level-A entry is ONE word, sign | low bucket bits | point index

// atomicAdd(&counter[idx], 1) returning the old value, with the wave-uniform case folded into one atomic
__device__ __forceinline__ uint32_t lds_count(uint32_t* counter, uint32_t idx) {
    const uint32_t first = __builtin_amdgcn_readfirstlane(idx);
    const uint64_t act = __ballot(1);
    const uint64_t same = __ballot(idx == first);
    if (same == act) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&counter[first], (uint32_t)__popcll(act));
        base = __builtin_amdgcn_readfirstlane(base);
        return base + rank;
    }
    return atomicAdd(&counter[idx], 1u);
}

template <class DIG>
static __global__ __launch_bounds__(SORT_THREADS) void hist_hi_kernel(const DIG* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                                      int w_first, int nchunk, uint32_t chunk_len, int fine_log,
                                                                      uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lds[];
    const uint32_t B = 1u << (c - 1), NB = B >> fine_log;
    const int wl = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    for (uint32_t b = threadIdx.x; b < NB; b += SORT_THREADS) lds[b] = 0;
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len;
    if (hi > n) hi = n;
    const DIG* d = dig + (size_t)(w_first + wl) * dstride;
    // chunk_len is a multiple of 8 and the digit rows are 16-byte aligned: eight digits per lane
    for (uint32_t i = lo + threadIdx.x * 8; i < hi; i += SORT_THREADS * 8) {
        uint32_t dg[8];
        load8_digits<DIG>(d + i, dg);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int v = (int)dg[k] - (int)B;
            if (i + k < hi && v != 0) (void)lds_count(lds, ((uint32_t)(v < 0 ? -v : v) - 1) >> fine_log);
        }
    }
    __syncthreads();
    uint32_t* out = hist + (size_t)blockIdx.x * NB;
    for (uint32_t b = threadIdx.x; b < NB; b += SORT_THREADS) out[b] = lds[b];
}

// counts[(set, sub, bin)] -> start offsets in (set, bin, sub) order, in place; bin_start[(set, bin)] (+ the grand
// total as last entry, also stored at *total_out = bucket_start[n_keys]).  One workgroup; pairs = sets * NB <= 4096.
// The sub-histograms of a (set, bin) pair are cut into `tpp` contiguous slices, one thread each, with the bin index
// running fastest over the lanes (coalesced rows of the histogram): a window-range run of a sharded MSM has few pairs and
// many sub-histograms per pair, and a thread per pair would walk them one dependent load at a time (44 us for two windows).
static __global__ __launch_bounds__(1024) void bins_scan_kernel(uint32_t* __restrict__ hist, int sets, int subs /* sub-histograms per set */, uint32_t NB,
                                                                uint32_t* __restrict__ bin_start, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t tot[4096];
    __shared__ uint32_t ssum[4096];
    __shared__ uint32_t sums[1024];
    const uint32_t pairs = (uint32_t)sets * NB;
    uint32_t tpp = pairs >= 4096 ? 1u : 4096u / pairs;   // slices per pair (power of two: NB is one, sets need not be)
    while (tpp & (tpp - 1)) tpp &= tpp - 1;
    if (tpp > (uint32_t)subs) tpp = 1u << (31 - __clz(subs));
    const uint32_t slice_len = ((uint32_t)subs + tpp - 1) / tpp;
    const uint32_t items = pairs * tpp;                    // <= 4096
    // item = (set, slice, bin), bin fastest
    for (uint32_t it = threadIdx.x; it < items; it += blockDim.x) {
        const uint32_t bin = it % NB, sl = (it / NB) % tpp, set = it / (NB * tpp);
        const uint32_t ch0 = sl * slice_len, ch1 = min((uint32_t)subs, ch0 + slice_len);
        uint32_t t = 0;
#pragma unroll 8
        for (uint32_t ch = ch0; ch < ch1; ++ch) t += hist[((size_t)set * subs + ch) * NB + bin];
        ssum[it] = t;
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < pairs; p += blockDim.x) {
        const uint32_t set = p / NB, bin = p % NB;
        uint32_t t = 0;
        for (uint32_t sl = 0; sl < tpp; ++sl) t += ssum[(set * tpp + sl) * NB + bin];
        tot[p] = t;
    }
    __syncthreads();
    {
        // exclusive scan of tot[0 .. pairs): four consecutive entries per thread, then a Hillis-Steele scan of the 1024 sums
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = threadIdx.x * 4 + k;
            v[k] = p < pairs ? tot[p] : 0u;
            sum += v[k];
        }
        sums[threadIdx.x] = sum;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint32_t o = (int)threadIdx.x >= d ? sums[threadIdx.x - d] : 0u;
            __syncthreads();
            sums[threadIdx.x] += o;
            __syncthreads();
        }
        uint32_t run = sums[threadIdx.x] - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = threadIdx.x * 4 + k;
            if (p < pairs) tot[p] = run;
            run += v[k];
        }
        if (threadIdx.x == 1023) {
            bin_start[pairs] = sums[1023];
            *total_out = sums[1023];
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < pairs; p += blockDim.x) bin_start[p] = tot[p];
    for (uint32_t it = threadIdx.x; it < items; it += blockDim.x) {
        const uint32_t bin = it % NB, sl = (it / NB) % tpp, set = it / (NB * tpp);
        const uint32_t ch0 = sl * slice_len, ch1 = min((uint32_t)subs, ch0 + slice_len);
        uint32_t run = tot[set * NB + bin];
        for (uint32_t k = 0; k < sl; ++k) run += ssum[(set * tpp + k) * NB + bin];
        for (uint32_t ch = ch0; ch < ch1; ++ch) {
            const size_t idx = ((size_t)set * subs + ch) * NB + bin;
            uint32_t t = hist[idx];
            hist[idx] = run;
            run += t;
        }
    }
}

// The same in three launches for large tables (fixed-base plans: up to 4096 bins x ~250 sub-histograms = 1 M counters, 0.2 ms
// in one workgroup): a grid of workgroups of 64 bins x 16 slices of the sub-histograms.
constexpr int BINS_SLICES = 16;
static __global__ __launch_bounds__(1024) void bins_partial_kernel(const uint32_t* __restrict__ hist, int subs, uint32_t NB, uint32_t pairs,
                                                                   uint32_t* __restrict__ slice_sums, uint32_t* __restrict__ tot) {
    __shared__ uint32_t sh[BINS_SLICES][64];
    const uint32_t lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * 64 + lane;
    const uint32_t slice_len = ((uint32_t)subs + BINS_SLICES - 1) / BINS_SLICES;
    uint32_t t = 0;
    if (p < pairs) {
        const uint32_t set = p / NB, bin = p % NB;
        const uint32_t ch0 = sl * slice_len, ch1 = min((uint32_t)subs, ch0 + slice_len);
#pragma unroll 4
        for (uint32_t ch = ch0; ch < ch1; ++ch) t += hist[((size_t)set * subs + ch) * NB + bin];
        slice_sums[(size_t)p * BINS_SLICES + sl] = t;
    }
    sh[sl][lane] = t;
    __syncthreads();
    if (sl == 0 && p < pairs) {
        uint32_t a = 0;
#pragma unroll
        for (int k = 0; k < BINS_SLICES; ++k) a += sh[k][lane];
        tot[p] = a;
    }
}
// exclusive scan of tot[0 .. pairs), pairs <= 4096, one workgroup
static __global__ __launch_bounds__(1024) void bins_scan_tot_kernel(const uint32_t* __restrict__ tot, uint32_t pairs,
                                                                    uint32_t* __restrict__ bin_start, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t sums[1024];
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p = threadIdx.x * 4 + k;
        v[k] = p < pairs ? tot[p] : 0u;
        sum += v[k];
    }
    sums[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t o = (int)threadIdx.x >= d ? sums[threadIdx.x - d] : 0u;
        __syncthreads();
        sums[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t run = sums[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p = threadIdx.x * 4 + k;
        if (p < pairs) bin_start[p] = run;
        run += v[k];
    }
    if (threadIdx.x == 1023) {
        bin_start[pairs] = sums[1023];
        *total_out = sums[1023];
    }
}
static __global__ __launch_bounds__(1024) void bins_prefix_kernel(uint32_t* __restrict__ hist, int subs, uint32_t NB, uint32_t pairs,
                                                                  const uint32_t* __restrict__ slice_sums, const uint32_t* __restrict__ bin_start) {
    const uint32_t lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * 64 + lane;
    if (p >= pairs) return;
    const uint32_t slice_len = ((uint32_t)subs + BINS_SLICES - 1) / BINS_SLICES;
    const uint32_t set = p / NB, bin = p % NB;
    uint32_t run = bin_start[p];
    for (uint32_t k = 0; k < sl; ++k) run += slice_sums[(size_t)p * BINS_SLICES + k];
    const uint32_t ch0 = sl * slice_len, ch1 = min((uint32_t)subs, ch0 + slice_len);
    for (uint32_t ch = ch0; ch < ch1; ++ch) {
        const size_t idx = ((size_t)set * subs + ch) * NB + bin;
        const uint32_t t = hist[idx];
        hist[idx] = run;
        run += t;
    }
}

// Level A with the output staged through LDS: the chunk is processed in tiles of 8192 entries (one 16-byte digit
// load per lane); a tile is counted and ranked per bin in LDS, the bin counts are scanned (every wave its share of the
// bins, then the 16 wave totals), the entries are placed bin-sorted into an LDS buffer and written out run by run, so
// that a run is one coalesced store instead of one four-byte request per entry -- the direct form is bound by the L2
// request rate.  Dynamic LDS: buf[8192] u32 | tcnt, toff, gcur [NBP] u32 | slot_bin[8192] u16, NBP = bins padded to 128.
constexpr uint32_t SCATTER_TILE = SORT_THREADS * 8;

template <class DIG>
static __global__ __launch_bounds__(SORT_THREADS) void scatter_hi_staged_kernel(const DIG* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                                                int w_first, int nchunk, uint32_t chunk_len, int fine_log,
                                                                                int shared_buckets, uint32_t table_stride, int table_w0,
                                                                                const uint32_t* __restrict__ offsets,
                                                                                uint32_t* __restrict__ tmp, uint8_t* __restrict__ tmp_fine) {
    // tmp_fine != nullptr: the reference alone fills the 31 bits below the sign (fixed-base keys above 2^20 points with
    // 20-bit windows: 13 x 2^22 table rows), and the fine bucket bits travel in a byte array beside the entries
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wave_tot[SORT_THREADS / 64];
    const bool split = tmp_fine != nullptr;
    const uint32_t B = 1u << (c - 1), NB = B >> fine_log;
    const uint32_t NBP = (NB + 127) & ~127u;          // multiple of 2 bins x 64 lanes
    const uint32_t per_wave = NBP / (SORT_THREADS / 64);  // bins scanned by one wave: <= 256 (NBP <= 4096)
    uint32_t* buf = lds;
    uint32_t* tcnt = lds + SCATTER_TILE;
    uint32_t* toff = tcnt + NBP;
    uint32_t* gcur = toff + NBP;
    uint16_t* slot_bin = reinterpret_cast<uint16_t*>(gcur + NBP);
    uint8_t* slot_fine = reinterpret_cast<uint8_t*>(slot_bin + SCATTER_TILE);   // used when split
    const int index_bits = 31 - fine_log;
    const int wl = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const uint32_t* off = offsets + (size_t)blockIdx.x * NB;
    for (uint32_t b = threadIdx.x; b < NBP; b += SORT_THREADS) {
        gcur[b] = b < NB ? off[b] : 0u;
        tcnt[b] = 0;
    }
    __syncthreads();
    uint32_t lo = chunk * chunk_len, hi = lo + chunk_len;
    if (hi > n) hi = n;
    const DIG* d = dig + (size_t)(w_first + wl) * dstride;
    const uint32_t ref_base = shared_buckets ? (uint32_t)(w_first + wl - table_w0) * table_stride : 0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t base = lo; base < hi; base += SCATTER_TILE) {
        const uint32_t i = base + threadIdx.x * 8;
        uint32_t val[8], rank[8];
        uint16_t bin[8];
        uint8_t fine[8];
        uint32_t dg[8] = {B, B, B, B, B, B, B, B};
        if (i < hi) load8_digits<DIG>(d + i, dg);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int v = (int)dg[k] - (int)B;
            bin[k] = 0xFFFF;
            if (i + k < hi && v != 0) {
                const uint32_t b = (uint32_t)(v < 0 ? -v : v) - 1;
                bin[k] = (uint16_t)(b >> fine_log);
                fine[k] = (uint8_t)(b & ((1u << fine_log) - 1));
                val[k] = (v < 0 ? 0x80000000u : 0u) | (split ? 0u : ((uint32_t)fine[k] << index_bits)) | (ref_base + i + k);
                rank[k] = lds_count(tcnt, bin[k]);
            }
        }
        __syncthreads();
        // exclusive scan of the tile's bin counts: wave w scans bins [w per_wave, (w+1) per_wave), per_wave / 64 = up to
        // four consecutive bins per lane (NBP <= 4096 coarse bins)
        const uint32_t bpl = (per_wave + 63) / 64;
        uint32_t cb[4] = {0, 0, 0, 0}, incl = 0;
        {
            const uint32_t b0 = wave * per_wave + lane * bpl;
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
                if (t < bpl && lane * bpl + t < per_wave) cb[t] = tcnt[b0 + t];
            incl = cb[0] + cb[1] + cb[2] + cb[3];
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                uint32_t o = __shfl_up(incl, dd, 64);
                if ((int)lane >= dd) incl += o;
            }
            if (lane == 63) wave_tot[wave] = incl;
        }
        __syncthreads();
        {
            uint32_t run = 0;
            for (uint32_t w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
            run += incl - (cb[0] + cb[1] + cb[2] + cb[3]);
            const uint32_t b0 = wave * per_wave + lane * bpl;
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
                if (t < bpl && lane * bpl + t < per_wave) {
                    toff[b0 + t] = run;
                    run += cb[t];
                }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (bin[k] != 0xFFFF) {
                const uint32_t slot = toff[bin[k]] + rank[k];
                buf[slot] = val[k];
                slot_bin[slot] = bin[k];
                if (split) slot_fine[slot] = fine[k];
            }
        }
        __syncthreads();
        const uint32_t count = toff[NBP - 1] + tcnt[NBP - 1];  // entries of this tile
        for (uint32_t sidx = threadIdx.x; sidx < count; sidx += SORT_THREADS) {
            const uint32_t b = slot_bin[sidx];
            const uint32_t pos = gcur[b] + (sidx - toff[b]);
            tmp[pos] = buf[sidx];
            if (split) tmp_fine[pos] = slot_fine[sidx];
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < NBP; b += SORT_THREADS) {
            gcur[b] += tcnt[b];
            tcnt[b] = 0;
        }
        __syncthreads();
    }
}

constexpr int SORT_LO_THREADS = 1024;

static __global__ __launch_bounds__(SORT_LO_THREADS) void sort_lo_kernel(const uint32_t* __restrict__ bin_start, const uint32_t* __restrict__ tmp, const uint8_t* __restrict__ tmp_fine,
                                                                         uint32_t B, int fine_log, uint32_t stage_cap,
                                                                         uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ sorted) {
    const bool split = tmp_fine != nullptr;  // fine bucket bits beside the entries (see scatter_hi_staged_kernel)
    constexpr uint32_t FINE = 1u << FINE_LOG_MAX;  // counters; the upper ones stay zero when fine_log < FINE_LOG_MAX
    __shared__ uint32_t cnt[FINE];
    extern __shared__ uint32_t stage[];  // stage_cap entries
    const uint32_t NB = B >> fine_log;
    const int index_bits = 31 - fine_log;
    const uint32_t fine_mask = (1u << fine_log) - 1;
    const uint32_t wl = blockIdx.x / NB, bin = blockIdx.x % NB;
    const uint32_t s0 = bin_start[blockIdx.x], s1 = bin_start[blockIdx.x + 1];
    for (uint32_t f = threadIdx.x; f < FINE; f += SORT_LO_THREADS) cnt[f] = 0;
    __syncthreads();
    for (uint32_t e = s0 + threadIdx.x; e < s1; e += SORT_LO_THREADS) (void)lds_count(cnt, split ? (uint32_t)tmp_fine[e] : (tmp[e] >> index_bits) & fine_mask);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the 256 counts by one wave: 4 per lane + a shuffle scan
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = cnt[threadIdx.x * 4 + k];
            sum += v[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += o;
        }
        uint32_t run = incl - sum;  // offsets relative to the bin's slice
        uint32_t* bs = bucket_start + (size_t)wl * B + ((size_t)bin << fine_log);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t f = threadIdx.x * 4 + k;
            cnt[f] = run;
            if (f <= fine_mask) bs[f] = s0 + run;
            run += v[k];
        }
    }
    __syncthreads();
    // The placing pass scatters inside the bin's own slice: done in LDS when the slice fits (the typical n / 128
    // entries), so that HBM/L2 see 16-byte-per-lane coalesced stores instead of one 4-byte request per entry -- the
    // scattered form is bound by the L2 request rate, not by bytes.
    const bool staged = s1 - s0 <= stage_cap;
    for (uint32_t e = s0 + threadIdx.x; e < s1; e += SORT_LO_THREADS) {
        const uint32_t t = tmp[e];
        const uint32_t pos = lds_count(cnt, split ? (uint32_t)tmp_fine[e] : (t >> index_bits) & fine_mask);
        const uint32_t ref = split ? t : (t & 0x80000000u) | (t & ((1u << index_bits) - 1));
        if (staged) stage[pos] = ref;
        else sorted[s0 + pos] = ref;
    }
    if (staged) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < s1 - s0; e += SORT_LO_THREADS) sorted[s0 + e] = stage[e];
    }
}

// ---- 2'/4'. bucket-range partition (general mode) -----------------------------------------------------
// One workgroup per (window, bucket range): it scans ALL digits of its window (2 B each, L2-resident, read
// 16 B per lane) and keeps only the entries whose bucket falls in its range.  Compared with the chunked
// scheme above this reads the digits `n_range` times, but every output line is written by ONE workgroup, so
// the 4-byte scattered stores are merged in its L2 instead of being written back as 8x amplified partial
// lines from 16 different XCD caches, and no per-chunk histogram / prefix pass is needed.
static __global__ __launch_bounds__(SORT_THREADS) void hist_range_kernel(const uint16_t* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                                         int w_first, int range_log,
                                                                         uint32_t* __restrict__ total) {
    extern __shared__ uint32_t lds[];
    const uint32_t B = 1u << (c - 1);
    const uint32_t n_range = B >> range_log;
    const uint32_t wl = blockIdx.x / n_range, r = blockIdx.x % n_range;
    const uint32_t range = 1u << range_log;
    const uint32_t lo_bucket = r << range_log;
    for (uint32_t b = threadIdx.x; b < range; b += SORT_THREADS) lds[b] = 0;
    __syncthreads();
    const uint16_t* d = dig + (size_t)(w_first + wl) * dstride;
    const uint32_t n8 = n & ~7u;
    for (uint32_t i = threadIdx.x * 8; i < n8; i += SORT_THREADS * 8) {
        uint4 pk = *reinterpret_cast<const uint4*>(d + i);
        const uint32_t wds[4] = {pk.x, pk.y, pk.z, pk.w};
        // bucket index relative to this range: rel < range <=> the entry is ours (v == 0 wraps to a huge value)
        uint32_t rel[8];
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int v = (int)((wds[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu) - (int)B;
            rel[k] = (uint32_t)(v < 0 ? -v : v) - 1u - lo_bucket;
            any |= (rel[k] < range) ? 1u : 0u;
        }
        if (any) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (rel[k] < range) atomicAdd(&lds[rel[k]], 1u);
        }
    }
    for (uint32_t i = n8 + threadIdx.x; i < n; i += SORT_THREADS) {
        int v = (int)d[i] - (int)B;
        uint32_t rel = (uint32_t)(v < 0 ? -v : v) - 1u - lo_bucket;
        if (rel < range) atomicAdd(&lds[rel], 1u);
    }
    __syncthreads();
    uint32_t* out = total + (size_t)wl * B + (size_t)r * range;
    for (uint32_t b = threadIdx.x; b < range; b += SORT_THREADS) out[b] = lds[b];
}

static __global__ __launch_bounds__(SORT_THREADS) void scatter_range_kernel(const uint16_t* __restrict__ dig, uint32_t n, uint32_t dstride, int c,
                                                                            int w_first, int range_log,
                                                                            const uint32_t* __restrict__ bucket_start,
                                                                            uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t lds[];
    const uint32_t B = 1u << (c - 1);
    const uint32_t n_range = B >> range_log;
    const uint32_t wl = blockIdx.x / n_range, r = blockIdx.x % n_range;
    const uint32_t range = 1u << range_log;
    const uint32_t lo_bucket = r << range_log;
    const uint32_t* start = bucket_start + (size_t)wl * B + (size_t)r * range;
    for (uint32_t b = threadIdx.x; b < range; b += SORT_THREADS) lds[b] = start[b];
    __syncthreads();
    const uint16_t* d = dig + (size_t)(w_first + wl) * dstride;
    const uint32_t n8 = n & ~7u;
    for (uint32_t i = threadIdx.x * 8; i < n8; i += SORT_THREADS * 8) {
        uint4 pk = *reinterpret_cast<const uint4*>(d + i);
        const uint32_t wds[4] = {pk.x, pk.y, pk.z, pk.w};
        uint32_t rel[8];
        uint32_t any = 0, negs = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int v = (int)((wds[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu) - (int)B;
            rel[k] = (uint32_t)(v < 0 ? -v : v) - 1u - lo_bucket;
            negs |= (v < 0 ? 1u : 0u) << k;
            any |= (rel[k] < range) ? 1u : 0u;
        }
        if (any) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (rel[k] < range) {
                    uint32_t pos = atomicAdd(&lds[rel[k]], 1u);
                    sorted[pos] = (i + k) | (((negs >> k) & 1u) << 31);
                }
        }
    }
    for (uint32_t i = n8 + threadIdx.x; i < n; i += SORT_THREADS) {
        int v = (int)d[i] - (int)B;
        uint32_t rel = (uint32_t)(v < 0 ? -v : v) - 1u - lo_bucket;
        if (rel < range) {
            uint32_t pos = atomicAdd(&lds[rel], 1u);
            sorted[pos] = i | (v < 0 ? 0x80000000u : 0u);
        }
    }
}

#endif  // ZK_PART == 0

// ---- 5. accumulate (dominant kernel) ----------------------------------------------------------------

// where lane t's run of bucket `key` goes: a bucket that lies within ONE segment has one run, which is the bucket sum itself
// and is written straight to the bucket array (combine_kernel skips such buckets); otherwise the run's slot in `partials`
template <class F>
__device__ __forceinline__ uint32_t* run_slot(uint32_t* partials, uint32_t* buckets, const uint32_t* run_start, const uint32_t* bucket_start,
                                              uint32_t key, uint32_t t, uint32_t seg_len) {
    constexpr int XW = 4 * F::LIMBS;
    const uint32_t r0 = run_start[key];
    if (run_start[key + 1] - r0 == 1) return buckets + (size_t)key * XW;
    return partials + (size_t)(r0 + t - bucket_start[key] / seg_len) * XW;
}

template <class G>
__global__ __launch_bounds__(256) void accumulate_kernel(const uint32_t* __restrict__ bases,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ bucket_start,
                                                         const uint32_t* __restrict__ run_start, uint32_t n_keys,
                                                         uint32_t seg_len, uint32_t* __restrict__ partials,
                                                         uint32_t* __restrict__ buckets) {
    typedef typename G::F F;
    constexpr int AW = 2 * F::LIMBS;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = bucket_start[n_keys];
    const uint32_t begin = t * seg_len;
    if (begin >= total) return;
    uint32_t end = begin + seg_len;
    if (end > total) end = total;
    // bucket of the first entry: largest key with bucket_start[key] <= begin (its end is > begin)
    uint32_t lo = 0, hi = n_keys;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (bucket_start[mid] <= begin) lo = mid; else hi = mid;
    }
    uint32_t key = lo;
    uint32_t next = bucket_start[key + 1];
    XYZZ<F> acc = xyzz_inf<F>();
    for (uint32_t e = begin; e < end; ++e) {
        if (e == next) {
            // the bucket ends inside this segment: flush its run, move to the next non-empty bucket
            store_xyzz<F>(run_slot<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), acc);
            acc = xyzz_inf<F>();
            do {
                ++key;
                next = bucket_start[key + 1];
            } while (next <= e);
        }
        uint32_t ref = sorted[e];
        const uint32_t* src = bases + (size_t)(ref & 0x7FFFFFFFu) * AW;
        xyzz_add_affine_mem<F>(acc, src, (ref >> 31) != 0);
    }
    store_xyzz<F>(run_slot<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), acc);
}

// ---- 6. combine ---------------------------------------------------------------------------------
// bucket = sum of its runs, three tiers in ONE launch (every dependent launch costs ~5 us of latency, and the two upper
// tiers are empty unless the scalars are skewed).  A point is held by a lane pair (pair.cuh) in all tiers.
//   blocks [0, small_blocks)                 one pair per bucket with 2 .. COMBINE_SMALL_MAX runs
//   blocks [small_blocks, + COMBINE_WAVE_BLOCKS)  one wave (32 pairs) per listed bucket, <= COMBINE_WAVE_MAX runs
//   the rest                                 one workgroup (128 pairs) per listed bucket
constexpr int COMBINE_THREADS = 256;
constexpr uint32_t COMBINE_WAVE_BLOCKS = 128, COMBINE_BIG_BLOCKS = 64;

template <class G>
__global__ __launch_bounds__(COMBINE_THREADS) void combine_kernel(const uint32_t* __restrict__ partials,
                                                                  const uint32_t* __restrict__ run_start, uint32_t n_keys,
                                                                  uint32_t small_blocks, const uint32_t* __restrict__ big_list,
                                                                  const uint32_t* __restrict__ big_count,
                                                                  uint32_t* __restrict__ buckets) {
    typedef typename G::F F;
    constexpr int XW = 4 * F::LIMBS;
    constexpr int HW = HalfRegs<F>::COUNT;
    __shared__ uint32_t sh[COMBINE_THREADS * HW];
    const bool odd = (threadIdx.x & 1) != 0;
    if (blockIdx.x < small_blocks) {
        const uint32_t key = (blockIdx.x * COMBINE_THREADS + threadIdx.x) >> 1;
        if (key >= n_keys) return;
        const uint32_t s0 = run_start[key], s1 = run_start[key + 1];
        // a single run: accumulate_kernel wrote the bucket itself; no run: it is never read... but the reduction reads
        // every bucket, so an empty one is set to infinity here
        if (s1 - s0 == 1 || s1 - s0 > COMBINE_SMALL_MAX) return;
        HalfPt<F> acc = half_inf<F>();
        if (s1 > s0) {
            acc = half_load<F>(partials + (size_t)s0 * XW, odd);
            // the next run is loaded before the addition of the current one: the chain is additions only, not
            // load-then-add round trips
            HalfPt<F> cur = s0 + 1 < s1 ? half_load<F>(partials + (size_t)(s0 + 1) * XW, odd) : half_inf<F>();
            for (uint32_t r = s0 + 1; r < s1; ++r) {
                HalfPt<F> nxt = r + 1 < s1 ? half_load<F>(partials + (size_t)(r + 1) * XW, odd) : half_inf<F>();
                acc = pair_add<F>(acc, cur, odd);
                cur = nxt;
            }
        }
        half_store<F>(buckets + (size_t)key * XW, odd, acc);
    } else if (blockIdx.x < small_blocks + COMBINE_WAVE_BLOCKS) {
        const uint32_t count = big_count[0];
        const uint32_t lane = threadIdx.x & 63, pair = lane >> 1;
        const uint32_t wave = ((blockIdx.x - small_blocks) * COMBINE_THREADS + threadIdx.x) >> 6;
        const uint32_t n_waves = (COMBINE_WAVE_BLOCKS * COMBINE_THREADS) >> 6;
        for (uint32_t b = wave; b < count; b += n_waves) {
            const uint32_t key = big_list[b];
            const uint32_t s0 = run_start[key], s1 = run_start[key + 1];
            HalfPt<F> v = half_inf<F>();
            for (uint32_t r = s0 + pair; r < s1; r += 32) v = pair_add<F>(v, half_load<F>(partials + (size_t)r * XW, odd), odd);
            for (int m = 32; m >= 2; m >>= 1) v = pair_add<F>(v, half_shfl_xor<F>(v, m), odd);
            if (lane < 2) half_store<F>(buckets + (size_t)key * XW, odd, v);
        }
    } else {
        const uint32_t count = big_count[1];
        const uint32_t j = threadIdx.x, pair = j >> 1;
        for (uint32_t b = blockIdx.x - small_blocks - COMBINE_WAVE_BLOCKS; b < count; b += COMBINE_BIG_BLOCKS) {
            const uint32_t key = big_list[n_keys - 1 - b];
            const uint32_t s0 = run_start[key], s1 = run_start[key + 1];
            HalfPt<F> v = half_inf<F>();
            for (uint32_t r = s0 + pair; r < s1; r += COMBINE_THREADS / 2) v = pair_add<F>(v, half_load<F>(partials + (size_t)r * XW, odd), odd);
            for (uint32_t off = COMBINE_THREADS / 4; off >= 1; off >>= 1) {  // tree over the 128 pairs
                half_lds_put<F>(sh, COMBINE_THREADS, j, v);
                __syncthreads();
                if (pair < off) v = pair_add<F>(v, half_lds_get<F>(sh, COMBINE_THREADS, j + 2 * off), odd);
                __syncthreads();
            }
            if (j < 2) half_store<F>(buckets + (size_t)key * XW, odd, v);
        }
    }
}

// ---- 7. bucket reduction ----------------------------------------------------------------------------
// Both kernels hold a point as a lane pair (pair.cuh): an addition is seven multiplications deep instead of
// fourteen and a half, which is what these latency-bound stages are made of.

// Two strided sums in one launch (rows and columns run side by side):
//   out[o] = sum_{j < count} in[(o / per_group) * group_stride + (o % per_group) * outer + j * inner]
// lpo lanes = lpo / 2 pairs per output element; the pairs stride over j, then a shuffle tree over the pairs.
struct SumJob {
    uint32_t n_out, per_group, group_stride, outer, inner, count;
    uint32_t out_offset;  // in points, into the shared output array
    uint32_t split = 1, outer2 = 0;  // the index x inside a group is taken apart: (x / split) * outer + (x % split) * outer2
    uint32_t in_offset = 0;          // in points, into the input array
};

template <class G>
__global__ __launch_bounds__(256) void strided_sum_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          SumJob j0, SumJob j1, uint32_t lpo) {
    typedef typename G::F F;
    constexpr int XW = 4 * F::LIMBS;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t o = gid / lpo;
    const uint32_t sub = gid % lpo;
    const bool odd = (sub & 1) != 0;
    const uint32_t pair = sub >> 1, n_pairs = lpo >> 1;
    SumJob job = j0;
    if (o >= j0.n_out) {
        o -= j0.n_out;
        job = j1;
    }
    const bool live = o < job.n_out;  // dead groups still take part in the shuffles
    const uint32_t x = live ? o % job.per_group : 0;
    size_t base = live ? (size_t)job.in_offset + (size_t)(o / job.per_group) * job.group_stride + (size_t)(x / job.split) * job.outer + (size_t)(x % job.split) * job.outer2 : 0;
    HalfPt<F> acc = half_inf<F>();
    if (live && pair < job.count) {
        // the next point is requested before the current addition starts: a step of the chain is an addition, not a
        // load followed by an addition
        HalfPt<F> cur = half_load<F>(in + (base + (size_t)pair * job.inner) * XW, odd);
        for (uint32_t j = pair; j < job.count; j += n_pairs) {
            const uint32_t jn = j + n_pairs < job.count ? j + n_pairs : j;
            HalfPt<F> nxt = half_load<F>(in + (base + (size_t)jn * job.inner) * XW, odd);
            acc = pair_add<F>(acc, cur, odd);
            cur = nxt;
        }
    }
    for (uint32_t m = lpo >> 1; m >= 2; m >>= 1) {
        HalfPt<F> other = half_shfl_xor<F>(acc, (int)m);
        acc = pair_add<F>(acc, other, odd);
    }
    if (live && sub < 2) half_store<F>(out + ((size_t)job.out_offset + o) * XW, odd, acc);
}

// S = sum_j j * X_j and T = sum_j X_j over BLOCKS of at most WS_BLOCK points of the input arrays, one workgroup per
// block, two lanes per point (so a workgroup is four waves: one per SIMD of its CU), via an inclusive suffix scan
// (log m steps) followed by a tree sum of the suffixes 1..m-1.  n0 arrays of m0 points from in0, then arrays of m1
// points from in1; block k of an array covers its points [k WS_BLOCK, ..) with LOCAL weights 0, 1, ..: the host tail adds
// k WS_BLOCK T_k along its Horner chain, where those doublings cost nothing extra.  out: (S, T) per block, arrays of
// in0 first.  Dynamic LDS: HalfRegs<F>::COUNT words per lane, word-major.
constexpr int WS_BLOCK = 128;
constexpr int WS_BLOCK_LOG = 7;
constexpr int HS_THREADS = 2 * WS_BLOCK;
template <class G>
__global__ __launch_bounds__(HS_THREADS) void weighted_sum_kernel(const uint32_t* __restrict__ in0, uint32_t m0, uint32_t n0,
                                                                  const uint32_t* __restrict__ in1, uint32_t m1,
                                                                  uint32_t* __restrict__ out) {
    typedef typename G::F F;
    constexpr int XW = 4 * F::LIMBS;
    extern __shared__ uint32_t sh[];
    const uint32_t tid = threadIdx.x, pj = tid >> 1;
    const bool odd = (tid & 1) != 0;
    const uint32_t bpa0 = (m0 + WS_BLOCK - 1) / WS_BLOCK, bpa1 = (m1 + WS_BLOCK - 1) / WS_BLOCK;
    const bool first = blockIdx.x < n0 * bpa0;
    const uint32_t rel = first ? blockIdx.x : blockIdx.x - n0 * bpa0;
    const uint32_t bpa = first ? bpa0 : bpa1, ma = first ? m0 : m1;
    const uint32_t arr_i = rel / bpa, blk = rel % bpa;
    const uint32_t m = min((uint32_t)WS_BLOCK, ma - blk * WS_BLOCK);
    const uint32_t* arr = (first ? in0 : in1) + ((size_t)arr_i * ma + (size_t)blk * WS_BLOCK) * XW;
    HalfPt<F> v = pj < m ? half_load<F>(arr + (size_t)pj * XW, odd) : half_inf<F>();
    for (uint32_t off = 1; off < m; off <<= 1) {
        half_lds_put<F>(sh, HS_THREADS, tid, v);
        __syncthreads();
        if (pj + off < m) {
            HalfPt<F> o = half_lds_get<F>(sh, HS_THREADS, tid + 2 * off);
            v = pair_add<F>(v, o, odd);
        }
        __syncthreads();
    }
    // v = suffix sum s_j
    if (pj == 0) half_store<F>(out + ((size_t)blockIdx.x * 2 + 1) * XW, odd, v);  // T = s_0
    if (pj == 0 || pj >= m) v = half_inf<F>();
    uint32_t top = 1;
    while (top < m) top <<= 1;
    for (uint32_t off = top / 2; off >= 1; off >>= 1) {
        half_lds_put<F>(sh, HS_THREADS, tid, v);
        __syncthreads();
        if (pj < off) {
            HalfPt<F> o = half_lds_get<F>(sh, HS_THREADS, tid + 2 * off);
            v = pair_add<F>(v, o, odd);
        }
        __syncthreads();
    }
    if (pj == 0) half_store<F>(out + (size_t)blockIdx.x * 2 * XW, odd, v);  // S
}

// ---- bases: canonical -> Montgomery; batch scalar multiplication -------------------------------------

// glv != 0: rows 2i = P_i and 2i + 1 = phi(P_i) = (beta x, y)
template <class G>
__global__ void bases_to_mont_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out, int glv) {
    typedef typename G::F F;
    constexpr int AW = 2 * F::LIMBS;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[AW];
    load_words<AW>(w, in + i * AW);
    Affine<F> p;
    p.x = F::from_canonical(w);
    p.y = F::from_canonical(w + F::LIMBS);
    if constexpr (GlvOf<G>::OK) {
        if (glv) {
            store_affine<F>(out + 2 * i * AW, p);
            p.x = F::mul(p.x, F::from_canonical(GlvOf<G>::P::BETA));
            store_affine<F>(out + (2 * i + 1) * AW, p);
            return;
        }
    }
    store_affine<F>(out + i * AW, p);
}

#if defined(ZK_GROUP) && (!defined(ZK_PART) || ZK_PART == 0)
// the plan's translation unit does not instantiate the heavy kernels (see msm_group.hip)
extern template __global__ void accumulate_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*);
extern template __global__ void bases_to_mont_kernel<ZK_GROUP>(const uint32_t*, uint64_t, uint32_t*, int);
extern template __global__ void combine_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*);
extern template __global__ void strided_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t*, SumJob, SumJob, uint32_t);
extern template __global__ void weighted_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t*);
ZK_SETUP_EXTERN_TEMPLATES(ZK_GROUP)
ZK_CODEC_EXTERN_TEMPLATES(ZK_GROUP)
#endif

#if !defined(ZK_PART) || ZK_PART == 0
// ---- host: plan --------------------------------------------------------------------------------------

// ZK_MSM_PRECOMPUTE: table row k = 2^(c (w_first + k)) * P_i (affine, Montgomery) for the w_count windows of the plan; on
// entry row 0 holds the bases themselves.  With these rows every window adds into ONE shared bucket set: the bucket
// reduction and the host tail shrink by the number of windows and the Horner pass disappears.  The chain of doublings
// runs in XYZZ on a scratch vector; each row is normalised with the batched inversion (one Fermat inversion per 1024
// points instead of one per point and row: 46 -> ~15 ms for a 2^20-point BN254 G1 key).
template <class G>
static int precompute_table(uint32_t* table, uint64_t n, int c, int w_first, int w_count) {
    typedef typename G::F F;
    constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
    if (w_first == 0 && w_count == 1) return ZK_OK;
    uint32_t* temp = nullptr;
    ZK_ALLOC(&temp, n * XW * 4);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const unsigned nblocks = (unsigned)((n + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E));
    hipLaunchKernelGGL(dbl_rows_kernel<G>, dim3(blocks), dim3(256), 0, 0, temp, n, c * w_first, (const uint32_t*)table);
    if (w_first > 0) hipLaunchKernelGGL(normalize_kernel<G>, dim3(nblocks), dim3(NORM_THREADS), 0, 0, temp, n, table, 0);
    for (int k = 1; k < w_count; ++k) {
        hipLaunchKernelGGL(dbl_rows_kernel<G>, dim3(blocks), dim3(256), 0, 0, temp, n, c, (const uint32_t*)nullptr);
        hipLaunchKernelGGL(normalize_kernel<G>, dim3(nblocks), dim3(NORM_THREADS), 0, 0, temp, n, table + (size_t)k * n * AW, 0);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    dev_free_cached(temp);
    ZK_HIP(e);
    return ZK_OK;
}

// fixed-base table of the last broadcast base of this group (Groth16.setup multiplies the same generator five times)
template <class G>
struct FixedTable {
    std::vector<uint64_t> base;   // canonical affine limbs the table was built for
    uint32_t* d_table = nullptr;
    int nwin = 0;
    std::mutex mu;
    static FixedTable& get() {
        static FixedTable t;
        return t;
    }
    void release() {
        std::lock_guard<std::mutex> lock(mu);
        if (d_table) dev_free_cached(d_table);
        d_table = nullptr;
        base.clear();
    }
    // caller holds mu
    int ensure(const uint64_t* base_limbs) {
        typedef typename G::F F;
        typedef typename G::Fr FrP;
        constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
        const size_t words64 = AW / 2;
        if (d_table && base.size() == words64 && memcmp(base.data(), base_limbs, words64 * 8) == 0) return ZK_OK;
        if (d_table) dev_free_cached(d_table);
        d_table = nullptr;
        base.clear();
        nwin = (FrP::BITS + 1 + FIXED_C - 1) / FIXED_C;
        // window bases 2^(16 j) G on the host (16 points)
        std::vector<uint32_t> wb((size_t)nwin * AW);
        {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(base_limbs);
            Affine<F> g = {F::from_canonical(w), F::from_canonical(w + F::LIMBS)};
            XYZZ<F> acc = xyzz_from_affine<F>(g);
            for (int j = 0; j < nwin; ++j) {
                Affine<F> a = xyzz_to_affine<F>(acc);
                F::store(wb.data() + (size_t)j * AW, a.x);
                F::store(wb.data() + (size_t)j * AW + F::LIMBS, a.y);
                if (j + 1 < nwin) for (int k = 0; k < FIXED_C; ++k) acc = xyzz_dbl<F>(acc);
            }
        }
        const uint64_t rows = (uint64_t)nwin * FIXED_HALF;
        uint32_t *d_wb = nullptr, *temp = nullptr;
        ZK_ALLOC(&d_table, rows * AW * 4);
        hipError_t e = dev_alloc_cached((void**)&d_wb, wb.size() * 4) == ZK_OK ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = dev_alloc_cached((void**)&temp, rows * XW * 4) == ZK_OK ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpy(d_wb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(fixed_table_kernel<G>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, d_wb, nwin, temp);
            hipLaunchKernelGGL(normalize_kernel<G>, dim3((unsigned)((rows + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E))),
                               dim3(NORM_THREADS), 0, 0, temp, rows, d_table, 0);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (d_wb) dev_free_cached(d_wb);
        if (temp) dev_free_cached(temp);
        if (e != hipSuccess) {
            dev_free_cached(d_table);
            d_table = nullptr;
            return fail(ZK_ERR_HIP, std::string("fixed-base table: ") + hipGetErrorString(e));
        }
        base.assign(base_limbs, base_limbs + words64);
        return ZK_OK;
    }
};

template <class G>
struct MsmPlan : MsmPlanBase {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    static constexpr int AW = 2 * F::LIMBS;
    static constexpr int XW = 4 * F::LIMBS;
    static constexpr uint64_t SEG_TARGET_LANES = 256ull * 1024;  // 4 waves per SIMD on 256 CUs
    static constexpr int MAX_C = 20;           // widest window (fixed-base plans; digits are then 32-bit)
    bool wide = false;                         // c > 16: 32-bit digits, two-level sort only

    // device workspace of one run (stages 2..7)
    struct Work {
        uint32_t *hist = nullptr, *total = nullptr, *bstart = nullptr, *sstart = nullptr;
        uint32_t *bsums = nullptr, *grand = nullptr, *big_list = nullptr, *big_count = nullptr;
        uint32_t *sorted = nullptr, *partials = nullptr, *buckets = nullptr, *rows = nullptr, *fin = nullptr;
        uint32_t* parts = nullptr;  // partial row / column sums of the two-step strided sums
        uint32_t *tmp_ref = nullptr, *bin_start = nullptr, *slice_sums = nullptr, *bin_tot = nullptr;  // two-level sort
        uint8_t* tmp_fine = nullptr;  // fine bucket bits of the level-A entries when the reference needs all 31 bits
        hipEvent_t ev_begin = nullptr, ev_acc0 = nullptr, ev_accs = nullptr, ev_acc1 = nullptr, ev_done = nullptr;
        hipEvent_t ev_release = nullptr;  // recorded by a borrower of this run's sort (enqueue_shared) after its last read
        bool lent = false;
        uint32_t seg_len = 0;
        int w_first = 0, w_count = 0;  // windows of the run in flight
        uint32_t groups = 0;           // bucket sets of the run in flight
    };

    uint64_t n = 0;      // entries per window (2 n_api with the endomorphism)
    uint64_t n_api = 0;  // points of the plan as the caller counts them
    bool glv = false;    // general G1 plan over (P_i, phi(P_i)) with half-length scalars
    bool pre = false;  // ZK_MSM_PRECOMPUTE: shared bucket set over a table of 2^(c w) P_i
    int pw_first = 0, pw_count = 0;  // windows this plan can run (a sharded rank's share; all of them by default)
    uint32_t B = 0, R = 0, C = 0;
    uint32_t bpr = 0, bpc = 0;       // weighted-sum blocks per row array / per column array
    int range_log = 0;  // general mode: log2(buckets per sort workgroup)
    Work ws;
    // shared device buffers
    std::shared_ptr<DeviceBlock> bases_block;  // the (table of) bases: shared by the clones of a plan
    uint32_t* d_bases = nullptr;
    uint32_t* d_scalars = nullptr;
    void* d_dig = nullptr;  // windows x (n + 8) digits, uint16_t (c <= 16) or uint32_t
    uint32_t* h_final = nullptr;  // pinned: (S, T) per weighted-sum block
    hipEvent_t ev_start = nullptr, ev_digits = nullptr, ev_end = nullptr;

    ~MsmPlan() override {
        // blocks go back to the caching allocator, which (unlike hipFree) does not wait for the device: make sure no run of
        // this plan is still in flight
        (void)hipDeviceSynchronize();
        void* bufs[] = {ws.hist, ws.total, ws.bstart, ws.sstart, ws.bsums, ws.grand, ws.big_list, ws.big_count,
                        ws.sorted, ws.partials, ws.buckets, ws.rows, ws.parts, ws.fin, ws.tmp_ref, ws.tmp_fine, ws.bin_start, ws.slice_sums, ws.bin_tot,
                        d_scalars, d_dig};
        for (void* q : bufs) dev_free_cached(q);
        pinned_free_cached(h_final);
        for (hipEvent_t e : {ws.ev_begin, ws.ev_acc0, ws.ev_accs, ws.ev_acc1, ws.ev_done, ws.ev_release, ev_start, ev_digits, ev_end}) if (e) (void)hipEventDestroy(e);
        stream_release((create_flags & ZK_MSM_HIGH_PRIORITY) != 0, own_stream);
    }

    // Every allocation lands in a member that the destructor frees, and the factory deletes the plan when init() fails
    // (msm_group.hip), so a failing hipMalloc half way through leaks nothing.
    // `share` != nullptr: a clone -- same bases (and fixed-base table), own workspace and stream
    int init(uint64_t n_points, const void* bases, int bases_on_device, int flags, int window_bits, int win_first, int win_count,
             const MsmPlan* share = nullptr) {
        static const bool trace = getenv("ZKMI_TRACE_INIT") != nullptr;
        auto t_prev = std::chrono::steady_clock::now();
        auto mark = [&](const char* what) {
            if (!trace) return;
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[plan init] %-24s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
            t_prev = now;
        };
        create_flags = flags;
        pre = (flags & ZK_MSM_PRECOMPUTE) != 0;
        if (n_points == 0 || n_points > (1ull << 26)) return fail(ZK_ERR_ARG, "MSM size must be in [1, 2^26]");
        const MsmLayout lay = msm_layout(FrP::BITS, GlvOf<G>::OK, n_points, flags, window_bits, win_count <= 0);
        glv = share ? share->glv : lay.glv;
        n_api = n_points;
        n = glv ? 2 * n_points : n_points;  // entries per window: the kernels see an MSM over (P_i, phi(P_i)) pairs
        entries_per_window = n;
        c = lay.c;
        if (c < 2 || c > MAX_C) return fail(ZK_ERR_ARG, "window bits must be in [2, 20]");
        if (c > 16 && !pre) return fail(ZK_ERR_ARG, "windows wider than 16 bits need a fixed-base plan (ZK_MSM_PRECOMPUTE)");
        wide = c > 16;
        nwin = glv ? glv_window_count(c) : window_count(FrP::BITS + 1, c);
        if (win_count <= 0) { win_first = 0; win_count = nwin; }
        if (win_first < 0 || win_first + win_count > nwin) return fail(ZK_ERR_ARG, "window range out of bounds");
        pw_first = win_first;
        pw_count = win_count;
        B = 1u << (c - 1);
        int rl = (c - 1 + 1) / 2;
        if (rl > 8 && !wide) rl = 8;
        R = 1u << rl;
        C = B / R;
        if (C > 1024 || R > 1024) return fail(ZK_ERR_ARG, "window too wide for the reduction stage");
        bpr = (R + WS_BLOCK - 1) / WS_BLOCK;
        bpc = (C + WS_BLOCK - 1) / WS_BLOCK;
        // bucket ranges (general mode, small inputs): about 256 sort workgroups in total, at least 64 buckets each
        {
            uint32_t wgs = 256u;
            if (const char* e = getenv("ZKMI_SORT_WGS")) wgs = (uint32_t)atoi(e);  // tuning knob
            uint32_t want = std::max<uint32_t>(1u, wgs / (uint32_t)std::max(1, pw_count));
            uint32_t per = std::max<uint32_t>(64u, B / want);
            if (per > B) per = B;
            range_log = log2_u64(per);
            if ((1u << range_log) > B) range_log = c - 1;
        }
        const uint64_t entries = (uint64_t)pw_count * n;
        if (entries > 0x7FFFFFFFull) return fail(ZK_ERR_ARG, "MSM too large");

        ZK_HIP_RC(stream_acquire((flags & ZK_MSM_HIGH_PRIORITY) != 0, &own_stream));
        if (share) {
            bases_block = share->bases_block;
            d_bases = share->d_bases;
        } else {
            bases_block = std::make_shared<DeviceBlock>();
            ZK_ALLOC(&bases_block->ptr, (pre ? (uint64_t)pw_count : 1ull) * n * AW * 4);
            d_bases = (uint32_t*)bases_block->ptr;
            if (bases_on_device) {
                hipLaunchKernelGGL(bases_to_mont_kernel<G>, dim3((unsigned)((n_api + 127) / 128)), dim3(128), 0, 0,
                                   (const uint32_t*)bases, n_api, d_bases, glv ? 1 : 0);
            } else {
                uint32_t* tmp = nullptr;
                ZK_ALLOC(&tmp, n_api * AW * 4);
                hipError_t e = hipMemcpy(tmp, bases, n_api * AW * 4, hipMemcpyHostToDevice);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(bases_to_mont_kernel<G>, dim3((unsigned)((n_api + 127) / 128)), dim3(128), 0, 0, tmp, n_api, d_bases, glv ? 1 : 0);
                    e = hipDeviceSynchronize();
                }
                dev_free_cached(tmp);
                ZK_HIP(e);
            }
            ZK_HIP(hipGetLastError());
            if (pre) {
                // rows 2^(c w) P_i for the windows of this plan only (a sharded rank never builds the other ranks' rows)
                int rc = precompute_table<G>(d_bases, n, c, pw_first, pw_count);
                if (rc) return rc;
            }
        }
        mark("stream + bases");
        ZK_ALLOC(&d_scalars, n_api * FrP::W * 4);
        ZK_ALLOC(&d_dig, (size_t)pw_count * (n + 8) * (wide ? 4 : 2));
        if (wide && !two_level_ok()) return fail(ZK_ERR_ARG, "this size does not fit the two-level sort that wide windows need");
        const uint64_t max_sets = pre ? 1ull : (uint64_t)pw_count;
        ZK_HIP_RC(pinned_alloc_cached((void**)&h_final, (size_t)max_sets * (bpr + bpc) * 2 * XW * 4));
        mark("scalars/digits/pinned");
        for (hipEvent_t* e : {&ev_start, &ev_digits, &ev_end}) ZK_HIP(hipEventCreate(e));
        {
            const uint64_t keys = max_sets * B;
            // a window-range run picks its own (shorter) segments: at most SEG_TARGET_LANES of them, or entries / 8
            const uint32_t seg_full = pick_seg_len(entries);
            const uint64_t max_segs = std::max<uint64_t>(entries / seg_full, std::min<uint64_t>(entries / 8, SEG_TARGET_LANES)) + keys + 8;
            // windows x chunks <= max(256, windows) sub-histograms: of all B buckets (one-level sort) or of the coarse bins only
            ZK_ALLOC(&ws.hist, (size_t)std::max<uint64_t>(256, pw_count) * (wide ? (B >> fine_log_for(n)) : B) * 4);
            ZK_ALLOC(&ws.total, keys * 4);
            ZK_ALLOC(&ws.bstart, (keys + 1) * 4);
            ZK_ALLOC(&ws.sstart, (keys + 1) * 4);
            ZK_ALLOC(&ws.bsums, ((keys + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * 4);
            ZK_ALLOC(&ws.grand, 4);
            ZK_ALLOC(&ws.big_list, keys * 4);
            ZK_ALLOC(&ws.big_count, 8);
            ZK_ALLOC(&ws.sorted, entries * 4);
            if (two_level_ok()) {
                ZK_ALLOC(&ws.tmp_ref, entries * 4);
                if (split_fine()) ZK_ALLOC(&ws.tmp_fine, entries);
                ZK_ALLOC(&ws.bin_start, (max_sets * (B >> fine_log_for(n)) + 1) * 4);
                ZK_ALLOC(&ws.slice_sums, 4096 * BINS_SLICES * 4);
                ZK_ALLOC(&ws.bin_tot, 4096 * 4);
            }
            ZK_ALLOC(&ws.partials, max_segs * XW * 4);
            ZK_ALLOC(&ws.buckets, keys * XW * 4);
            ZK_ALLOC(&ws.rows, max_sets * (R + C) * XW * 4);
            if (sum_part_len() >= 2) ZK_ALLOC(&ws.parts, max_sets * 2 * (uint64_t)(B / sum_part_len()) * XW * 4);
            ZK_ALLOC(&ws.fin, max_sets * (bpr + bpc) * 2 * XW * 4);
            for (hipEvent_t* e : {&ws.ev_begin, &ws.ev_acc0, &ws.ev_accs, &ws.ev_acc1, &ws.ev_done, &ws.ev_release}) ZK_HIP(hipEventCreate(e));
        }
        mark("workspace + events");
        // LDS above 64 KiB needs the opt-in
        int lds_bytes = (int)((wide ? (1u << 15) : B) * 4);  // the one-level kernels never run for wide windows
        ZK_HIP(hipFuncSetAttribute((const void*)hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)hist_range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_hi_staged_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_hi_staged_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)sort_lo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)weighted_sum_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(HalfRegs<F>::COUNT * HS_THREADS * 4)));
        mark("func attributes");
        ZK_HIP(hipDeviceSynchronize());
        mark("device sync");
        return ZK_OK;
    }

    // chunked sort: windows x chunks workgroups of 1024 threads, ONE per CU (the LDS histogram takes 128 KiB at
    // c = 16), so their number is kept at or just below the 256 CUs: 272 workgroups would run as 256 + 16,
    // i.e. take twice as long
    static int chunks_for(int windows, uint64_t count) {
        int k = 256 / std::max(1, windows);
        if (k < 1) k = 1;
        uint64_t cap = (count + 4095) / 4096;  // at least 4096 entries per chunk
        if ((uint64_t)k > cap) k = (int)std::max<uint64_t>(1, cap);
        return k;
    }

    int exclusive_scan(const uint32_t* in, uint32_t cnt, uint32_t* out, hipStream_t st) {
        uint32_t blocks = (cnt + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(scan_block_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, in, cnt, out, ws.bsums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, ws.bsums, blocks, ws.grand);
        hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, out, cnt, ws.bsums, ws.grand);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }

    // fine bucket bits of the two-level sort for n points: the largest of 8, 7 that leaves room for the index in a 32-bit
    // entry, with at least four coarse bins per window and at most 4096 (window, bin) pairs (one-workgroup scan); 0 = n/a
    int fine_log_for(uint64_t points) const {
        // fixed-base mode: ONE bucket set over references into the (window, point) table, so the references are wider
        // and the coarse bins are shared by all windows -- more, smaller bins keep level B parallel
        const uint64_t refs = pre ? (uint64_t)pw_count * points : points;
        const uint64_t sets = pre ? 1 : (uint64_t)pw_count;
        if (wide) {
            const int f = c - 13;  // 4096 coarse bins; the fine bits move out of the entry when the reference needs the room
            return refs <= 0x7FFFFFFFull ? f : 0;
        }
        static const int f_env = getenv("ZKMI_FINE_LOG") ? atoi(getenv("ZKMI_FINE_LOG")) : 0;  // tuning knob (general mode)
        if (f_env && !pre && refs <= (1ull << (31 - f_env)) && c - 1 >= f_env + 2 && sets * (B >> f_env) <= 4096) return f_env;
        const int f_hi = pre ? 5 : FINE_LOG_MAX, f_lo = pre ? 4 : FINE_LOG_MAX - 1;
        // general mode: the widest bins that still hold about 8192 entries each (one level-B workgroup sorts a bin in LDS;
        // 2^21 split-scalar entries per window: 7 fine bits, 0.175 ms for digits + sort against 0.20 with 8)
        int best = 0;
        for (int f = f_hi; f >= f_lo; --f) {
            if (!(refs <= (1ull << (31 - f)) && c - 1 >= f + 2 && sets * (B >> f) <= 4096)) continue;
            if (pre || (points >> (c - 1 - f)) <= 8192) return f;
            best = f;
        }
        return best;
    }

    // level-A entries carry (sign, fine bucket bits, reference) in 32 bits while that fits; wide windows over a big table
    // (13 x n rows, n > 2^20) keep the fine bits in a byte array beside them
    bool split_fine() const {
        return wide && pre && (uint64_t)pw_count * n > (1ull << (31 - (c - 13)));
    }

    // buckets one lane pair adds up in the first step of the two-step strided sums (0 = one step)
    uint32_t sum_part_len() const {
        const uint32_t k = 16u;
        return (R >= 4 * k && C >= 4 * k) ? k : 0u;
    }

    bool two_level_ok() const {
        static const bool off = getenv("ZKMI_NO_TWO_LEVEL") != nullptr;
        return !off && fine_log_for(n) > 0;
    }

    // Segment length for a run over `entries` sorted entries: aim at >= 4 waves per SIMD worth of lanes (a window-range
    // run of a sharded MSM has far fewer entries than the plan's full set; with the plan-wide length its lanes would
    // be too few and each would walk 64 additions at lone-wave speed).
    uint32_t pick_seg_len(uint64_t entries) const {
        static const uint64_t target = getenv("ZKMI_SEG_LANES") ? (uint64_t)atoll(getenv("ZKMI_SEG_LANES")) : SEG_TARGET_LANES;
        uint64_t sl = (entries + target - 1) / target;
        if (sl < 8) sl = 8;
        if (sl > 64) sl = 64;
        if (pre) {
            // shared bucket set: keep a bucket within ~12 runs so that one lane pair can combine it
            uint64_t per_bucket = entries / B;
            uint64_t want = (per_bucket + 11) / 12;
            if (want > sl) sl = want;
            if (sl > 1024) sl = 1024;
        }
        return (uint32_t)sl;
    }

    // stages 2..7 + D2H for the windows [ws.w_first, ws.w_first + ws.w_count) on stream st
    // `borrowed`: the digits and the sort of another plan's run over the same scalars (enqueue_shared); stages 1-4 are skipped
    // phase: 0 = everything, 1 = up to the sorted entry list only, 2 = from the accumulate kernel on (after phase 1).
    // gate: waited for right before the accumulate kernel (another plan's accumulate has finished), so that the
    // accumulate kernels of several plans run one after the other while their sorts and reductions overlap.
    int run_stages(uint32_t m, uint32_t dstride, hipStream_t st, const SortExport* borrowed = nullptr, int phase = 0, hipEvent_t gate = nullptr) {
        Work& l = ws;
        const int w_first = l.w_first, w_count = l.w_count;
        const uint32_t groups = l.groups;
        const uint32_t n_keys = groups * B;
        const int nchunk = chunks_for(w_count, m);  // this run's windows fill the chip
        const uint32_t seg_len = borrowed ? borrowed->seg_len : pick_seg_len((uint64_t)w_count * m);
        l.seg_len = seg_len;
        const uint32_t ch_len = (m + nchunk - 1) / nchunk;
        if (phase != 2) ZK_HIP(hipEventRecord(l.ev_begin, st));
        const uint32_t *p_sorted = l.sorted, *p_bstart = l.bstart, *p_sstart = l.sstart, *p_big_list = l.big_list, *p_big_count = l.big_count;
        if (phase == 2) {
            // sorted in phase 1
        } else if (borrowed) {
            p_sorted = borrowed->sorted; p_bstart = borrowed->bstart; p_sstart = borrowed->sstart;
            p_big_list = borrowed->big_list; p_big_count = borrowed->big_count;
            ZK_HIP(hipStreamWaitEvent(st, borrowed->sorted_ready, 0));
        } else {
        // digit rows are stored relative to the plan's first window; the kernels index them with absolute windows
        const uintptr_t dig_base = reinterpret_cast<uintptr_t>(this->d_dig) - (uintptr_t)pw_first * dstride * (wide ? 4 : 2);
        const uint16_t* d_dig = reinterpret_cast<const uint16_t*>(dig_base);
        const uint32_t* d_dig32 = reinterpret_cast<const uint32_t*>(dig_base);
        // general mode, small inputs: bucket-range partition (measured faster up to 2^18); otherwise the two-level sort
        const bool ranged = !pre && !wide && m < (1u << 19);
        const bool two_level = !ranged && l.tmp_ref != nullptr;
        if (two_level) {
            const int fl = fine_log_for(n);
            const uint32_t NB = B >> fl;
            const uint32_t ch8 = (ch_len + 7) & ~7u;  // the kernels read eight digits per load
            if (wide) hipLaunchKernelGGL(hist_hi_kernel<uint32_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), NB * 4, st, d_dig32, m, dstride, c, w_first, nchunk, ch8, fl, l.hist);
            else hipLaunchKernelGGL(hist_hi_kernel<uint16_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), NB * 4, st, d_dig, m, dstride, c, w_first, nchunk, ch8, fl, l.hist);
            // fixed-base mode: one bucket set fed by all (window, chunk) sub-histograms; general mode: one set per window
            const int sets = pre ? 1 : w_count, subs = pre ? w_count * nchunk : nchunk;
            const uint32_t pairs = (uint32_t)sets * NB;
            if ((uint64_t)pairs * subs >= (1u << 17)) {
                const unsigned bb = (pairs + 63) / 64;
                hipLaunchKernelGGL(bins_partial_kernel, dim3(bb), dim3(1024), 0, st, l.hist, subs, NB, pairs, l.slice_sums, l.bin_tot);
                hipLaunchKernelGGL(bins_scan_tot_kernel, dim3(1), dim3(1024), 0, st, l.bin_tot, pairs, l.bin_start, l.bstart + n_keys);
                hipLaunchKernelGGL(bins_prefix_kernel, dim3(bb), dim3(1024), 0, st, l.hist, subs, NB, pairs, l.slice_sums, l.bin_start);
            } else {
                hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, st, l.hist, sets, subs, NB, l.bin_start, l.bstart + n_keys);
            }
            {
                const uint32_t NBP = (NB + 127) & ~127u;
                const size_t lds_a = (size_t)SCATTER_TILE * 4 + (size_t)NBP * 12 + (size_t)SCATTER_TILE * 2 + (l.tmp_fine ? SCATTER_TILE : 0);
                if (wide) hipLaunchKernelGGL(scatter_hi_staged_kernel<uint32_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), lds_a, st, d_dig32, m, dstride, c, w_first, nchunk, ch8, fl, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.tmp_ref, l.tmp_fine);
                else hipLaunchKernelGGL(scatter_hi_staged_kernel<uint16_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), lds_a, st, d_dig, m, dstride, c, w_first, nchunk, ch8, fl, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.tmp_ref, (uint8_t*)nullptr);
            }
            // LDS stage of level B: 1.5x the expected entries of a coarse bin, capped at 96 KiB
            uint64_t expect = ((uint64_t)w_count * m) / ((uint64_t)sets * NB);
            uint32_t stage_cap = (uint32_t)std::min<uint64_t>(24576, std::max<uint64_t>(2048, expect + expect / 2));
            hipLaunchKernelGGL(sort_lo_kernel, dim3(sets * NB), dim3(SORT_LO_THREADS), (size_t)stage_cap * 4, st, l.bin_start, l.tmp_ref, (const uint8_t*)l.tmp_fine, B, fl, stage_cap, l.bstart, l.sorted);
        } else if (ranged) {
            hipLaunchKernelGGL(hist_range_kernel, dim3(w_count * (B >> range_log)), dim3(SORT_THREADS), (4u << range_log), st, d_dig, m, dstride, c, w_first, range_log, l.total);
        } else {
            hipLaunchKernelGGL(hist_kernel, dim3(w_count * nchunk), dim3(SORT_THREADS), B * 4, st, d_dig, m, dstride, c, w_first, nchunk, ch_len, l.hist);
            hipLaunchKernelGGL(prefix_kernel, dim3((n_keys + 255) / 256), dim3(256), 0, st, l.hist, pre ? w_count * nchunk : nchunk, B, n_keys, l.total);
        }
        int rc;
        if (!two_level && (rc = exclusive_scan(l.total, n_keys, l.bstart, st))) return rc;
        {
            // run offsets: run counts computed on the fly + three-launch scan
            const uint32_t blocks = (n_keys + SCAN_BLOCK - 1) / SCAN_BLOCK;
            hipLaunchKernelGGL(runs_scan_block_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, l.bstart, n_keys, seg_len, l.sstart, l.bsums, l.big_list, l.big_count);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, l.bsums, blocks, l.grand);
            hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, l.sstart, n_keys, l.bsums, l.grand);
        }
        if (two_level) {
            // already sorted
        } else if (ranged) {
            hipLaunchKernelGGL(scatter_range_kernel, dim3(w_count * (B >> range_log)), dim3(SORT_THREADS), (4u << range_log), st, d_dig, m, dstride, c, w_first, range_log, l.bstart, l.sorted);
        } else {
            const unsigned blocks = pre ? (unsigned)(w_count * nchunk) : (unsigned)(8 * ((w_count + 7) / 8) * nchunk);
            hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(SORT_THREADS), B * 4, st, d_dig, m, dstride, c, w_first, w_count, nchunk, ch_len, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.bstart, l.sorted);
        }
        }  // !borrowed
        if (phase != 2) ZK_HIP(hipEventRecord(l.ev_acc0, st));
        if (phase == 1) return ZK_OK;
        if (gate) ZK_HIP(hipStreamWaitEvent(st, gate, 0));
        ZK_HIP(hipEventRecord(l.ev_accs, st));
        // 5. accumulate
        uint64_t lanes_needed = ((uint64_t)w_count * m + seg_len - 1) / seg_len;
        hipLaunchKernelGGL(accumulate_kernel<G>, dim3((unsigned)((lanes_needed + 255) / 256)), dim3(256), 0, st, d_bases, p_sorted, p_bstart, p_sstart, n_keys, seg_len, l.partials, l.buckets);
        ZK_HIP(hipEventRecord(l.ev_acc1, st));
        // 6. combine (three tiers, one launch)
        const uint32_t small_blocks = (2 * n_keys + COMBINE_THREADS - 1) / COMBINE_THREADS;
        hipLaunchKernelGGL(combine_kernel<G>, dim3(small_blocks + COMBINE_WAVE_BLOCKS + COMBINE_BIG_BLOCKS), dim3(COMBINE_THREADS), 0, st,
                           l.partials, p_sstart, n_keys, small_blocks, p_big_list, p_big_count, l.buckets);
        if (borrowed) ZK_HIP(hipEventRecord(borrowed->release, st));  // the lender's buffers are no longer read
        // 7. reduce: rows (sum over lo), cols (sum over hi), weighted sums
        uint32_t n_rows = groups * R, n_cols = groups * C;
        SumJob rows = {n_rows, R, B, C, 1u, C, 0u};
        SumJob cols = {n_cols, C, B, 1u, C, R, n_rows};
        // Two steps when the sums are long: one lane pair walks a run of K buckets with no idle lanes (the tree of the
        // one-step form leaves half of its lane-steps empty), then a short tree adds the R / K or C / K partial sums.
        static const bool one_step = getenv("ZKMI_SUM_ONE_STEP") != nullptr;  // A/B knob
        // worth it from 2^18 buckets on (8 windows of 2^15, or the 2^19-bucket set of a fixed-base plan): with fewer the sums
        // are a latency chain and the second launch only lengthens it (measured: 2^17 buckets 0.223 vs 0.212 ms)
        const uint32_t K = (one_step || n_keys < (1u << 18)) ? 0u : sum_part_len();
        if (K >= 2 && l.parts && C % K == 0 && R % K == 0 && C / K >= 2 && R / K >= 2) {
            const uint32_t pr = C / K, pc = R / K;  // partial sums per row sum / per column sum
            // step 1: partial (row r, part p) = sum of buckets r C + p K + [0, K); (column j, part p) = sum of (p K + i) C + j
            SumJob prow = {n_rows * pr, R * pr, B, C, 1u, K, 0u, pr, K};
            SumJob pcol = {n_cols * pc, C * pc, B, 1u, C, K, n_rows * pr, pc, K * C};
            hipLaunchKernelGGL(strided_sum_kernel<G>, dim3((unsigned)((((uint64_t)prow.n_out + pcol.n_out) * 2 + 255) / 256)), dim3(256), 0, st, l.buckets, l.parts, prow, pcol, 2u);
            // step 2: contiguous runs of pr (pc) partial sums
            SumJob frow = {n_rows, n_rows, 0u, pr, 1u, pr, 0u};
            SumJob fcol = {n_cols, n_cols, 0u, pc, 1u, pc, n_rows, 1u, 0u, n_rows * pr};
            const uint32_t lpo2 = 2 * std::min<uint32_t>(32u, std::max(pr, pc) / 2);
            hipLaunchKernelGGL(strided_sum_kernel<G>, dim3(((n_rows + n_cols) * lpo2 + 255) / 256), dim3(256), 0, st, l.parts, l.rows, frow, fcol, lpo2);
        } else {
        // lanes per output (two lanes = one point): many outputs (one bucket set per window) -> 16 pairs each walk
        // count/16 buckets and finish with a 4-level tree; few outputs (shared bucket set) -> 32 pairs, shortest chain
        static const uint32_t lpo_env = getenv("ZKMI_LPO") ? (uint32_t)atoi(getenv("ZKMI_LPO")) : 0u;
        const uint32_t lpo = lpo_env ? lpo_env : ((n_rows + n_cols) >= 4096 ? 32u : 64u);
        hipLaunchKernelGGL(strided_sum_kernel<G>, dim3(((n_rows + n_cols) * lpo + 255) / 256), dim3(256), 0, st, l.buckets, l.rows, rows, cols, lpo);
        }
        hipLaunchKernelGGL(weighted_sum_kernel<G>, dim3(groups * (bpr + bpc)), dim3(HS_THREADS), (size_t)HalfRegs<F>::COUNT * HS_THREADS * 4, st,
                           l.rows, R, groups, l.rows + (size_t)n_rows * XW, C, l.fin);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(h_final, l.fin, (size_t)groups * (bpr + bpc) * 2 * XW * 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipEventRecord(l.ev_done, st));
        return ZK_OK;
    }

    int create_flags = 0;
    int clone(MsmPlanBase** out) override {
        MsmPlan* p = new MsmPlan();
        int rc = p->init(n_api, nullptr, 0, create_flags & ~ZK_MSM_HIGH_PRIORITY, c, pw_first, pw_count, this);
        if (rc) {
            delete p;
            return rc;
        }
        *out = p;
        return ZK_OK;
    }

    // state carried from enqueue() to finish()
    int q_first = 0, q_count = 0;
    uint32_t q_m = 0;
    hipStream_t q_stream = nullptr;
    bool q_pending = false;

    bool q_sorted = false;  // enqueue_sort done, enqueue_rest still to come

    int enqueue(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st) override {
        return enqueue_phase(n_scalars, scalars, on_device, w_first, w_count, st, 0);
    }
    int enqueue_sort(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st) override {
        return enqueue_phase(n_scalars, scalars, on_device, w_first, w_count, st, 1);
    }
    hipEvent_t accumulate_done_event() override { return ws.ev_acc1; }
    int enqueue_rest(MsmPlanBase* after) override {
        std::lock_guard<std::mutex> lock(mu);
        if (!q_sorted) return fail(ZK_ERR_ARG, "zk_msm_plan_enqueue_rest without zk_msm_plan_enqueue_sort");
        q_sorted = false;
        if (q_m > 0) {
            int rc = run_stages(q_m, (q_m + 7u) & ~7u, q_stream, nullptr, 2, after ? after->accumulate_done_event() : nullptr);
            if (rc) return rc;
            ZK_HIP(hipEventRecord(ev_end, q_stream));
        }
        q_pending = true;
        return ZK_OK;
    }

    int enqueue_phase(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st, int phase) {
        std::lock_guard<std::mutex> lock(mu);
        if (q_pending || q_sorted) return fail(ZK_ERR_ARG, "MSM plan already has a run in flight: call zk_msm_plan_finish first");
        if (n_scalars > n_api) return fail(ZK_ERR_LENGTH, "Number of points and scalars mismatch");
        if (w_count <= 0) { w_first = pw_first; w_count = pw_count; }
        if (w_first < pw_first || w_first + w_count > pw_first + pw_count)
            return fail(ZK_ERR_ARG, "window range out of bounds (the plan was created for windows [" + std::to_string(pw_first) + ", " +
                                        std::to_string(pw_first + pw_count) + "))");
        const uint32_t m_api = (uint32_t)n_scalars;
        const uint32_t m = glv ? 2 * m_api : m_api;  // entries per window
        q_first = w_first; q_count = w_count; q_m = m; q_stream = st;
        if (ws.lent) {  // a borrower of the previous run's sort may still be reading the buffers this run overwrites
            ZK_HIP(hipStreamWaitEvent(st, ws.ev_release, 0));
            ws.lent = false;
        }
        if (m > 0) {
            const uint32_t* sc = (const uint32_t*)scalars;
            if (!on_device) {
                ZK_HIP(hipMemcpyAsync(d_scalars, scalars, (size_t)m_api * FrP::W * 4, hipMemcpyHostToDevice, st));
                sc = d_scalars;
            }
            const uint32_t dstride = (m + 7u) & ~7u;
            ZK_HIP(hipEventRecord(ev_start, st));
            // 1. digits (the windows of this run); digit rows are stored relative to the plan's first window
            DigitBias bias;
            memset(&bias, 0, sizeof(bias));
            for (int w = 0; w < nwin; ++w) {
                int bit = w * c + (c - 1);
                bias.v[bit >> 5] |= 1u << (bit & 31);
            }
            const uintptr_t dig_base = reinterpret_cast<uintptr_t>(d_dig) - (uintptr_t)pw_first * dstride * (wide ? 4 : 2);
            if (glv) {
                if constexpr (GlvOf<G>::OK)
                    hipLaunchKernelGGL(glv_digits_kernel<FrP>, dim3((m_api + 255) / 256), dim3(256), 0, st, sc, m_api, dstride, c, w_first, w_count, bias,
                                       GlvOf<G>::P::K, reinterpret_cast<uint16_t*>(dig_base), ws.big_count);
            } else if (wide) hipLaunchKernelGGL((digits_kernel<FrP, uint32_t>), dim3((m + 255) / 256), dim3(256), 0, st, sc, m, dstride, c, w_first, w_count, bias,
                                         reinterpret_cast<uint32_t*>(dig_base), ws.big_count);
            else hipLaunchKernelGGL((digits_kernel<FrP, uint16_t>), dim3((m + 255) / 256), dim3(256), 0, st, sc, m, dstride, c, w_first, w_count, bias,
                                    reinterpret_cast<uint16_t*>(dig_base), ws.big_count);
            ZK_HIP(hipEventRecord(ev_digits, st));
            ws.w_first = w_first;
            ws.w_count = w_count;
            ws.groups = pre ? 1u : (uint32_t)w_count;
            int rc = run_stages(m, dstride, st, nullptr, phase);
            if (rc) return rc;
            if (phase == 0) ZK_HIP(hipEventRecord(ev_end, st));
        }
        if (phase == 1) q_sorted = true;
        else q_pending = true;
        return ZK_OK;
    }

    int cancel() override {
        std::lock_guard<std::mutex> lock(mu);
        if ((q_pending || q_sorted) && q_stream) ZK_HIP(hipStreamSynchronize(q_stream));
        q_pending = q_sorted = false;
        return ZK_OK;
    }

    int export_sort(SortExport* out) override {
        std::lock_guard<std::mutex> lock(mu);
        if ((!q_pending && !q_sorted) || q_m == 0) return fail(ZK_ERR_ARG, "the lending plan has no run in flight");
        out->sorted = ws.sorted; out->bstart = ws.bstart; out->sstart = ws.sstart;
        out->big_list = ws.big_list; out->big_count = ws.big_count;
        out->n = n; out->m = q_m; out->seg_len = ws.seg_len; out->groups = ws.groups;
        out->c = c; out->nwin = nwin; out->w_first = q_first; out->w_count = q_count;
        out->pw_first = pw_first; out->pw_count = pw_count; out->scalar_bits = FrP::BITS; out->pre = pre; out->glv = glv;
        out->sorted_ready = ws.ev_acc0;
        out->release = ws.ev_release;
        ws.lent = true;
        return ZK_OK;
    }

    // Run this plan on the digits and the sorted entry list of `lender`'s run in flight: same scalars, other bases
    // (Groth16's <tau_1, v> and <tau_2, v>).  Both plans must have the same size, window layout, mode and window range;
    // the entry list addresses points (or table rows) by index, which is independent of the group.
    int enqueue_shared(MsmPlanBase* lender, hipStream_t st) override {
        SortExport ex;
        int rc = lender->export_sort(&ex);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(mu);
        if (q_pending || q_sorted) return fail(ZK_ERR_ARG, "MSM plan already has a run in flight: call zk_msm_plan_finish first");
        if (ex.n != n || ex.c != c || ex.nwin != nwin || ex.pre != pre || ex.glv != glv || ex.scalar_bits != FrP::BITS ||
            ex.pw_first != pw_first || ex.pw_count != pw_count)
            return fail(ZK_ERR_ARG, "plans differ in size, window layout or mode: the sort cannot be shared");
        if (ws.lent) {
            ZK_HIP(hipStreamWaitEvent(st, ws.ev_release, 0));
            ws.lent = false;
        }
        q_first = ex.w_first; q_count = ex.w_count; q_m = ex.m; q_stream = st;
        ZK_HIP(hipEventRecord(ev_start, st));
        ZK_HIP(hipEventRecord(ev_digits, st));
        ws.w_first = ex.w_first;
        ws.w_count = ex.w_count;
        ws.groups = ex.groups;
        rc = run_stages(ex.m, (ex.m + 7u) & ~7u, st, &ex);
        if (rc) return rc;
        ZK_HIP(hipEventRecord(ev_end, st));
        q_pending = true;
        return ZK_OK;
    }

    int finish(uint64_t* out) override {
        std::lock_guard<std::mutex> lock(mu);
        if (q_sorted) return fail(ZK_ERR_ARG, "zk_msm_plan_finish before zk_msm_plan_enqueue_rest");
        if (!q_pending) return fail(ZK_ERR_ARG, "zk_msm_plan_finish without a pending run");
        q_pending = false;
        typedef typename G::HostF HF;  // 64-bit-limb host arithmetic for the sequential tail (host64.cuh)
        XYZZ<HF> total = xyzz_inf<HF>();
        if (q_m > 0) {
            ZK_HIP(hipEventSynchronize(ev_end));
            // 8. host tail.  Per bucket set, with row blocks (S_k, T_k), column blocks (S'_k, T'_k), C = 2^lc, WS_BLOCK = 2^7:
            //        W = 2^(lc+7) X_R + 2^lc sum S_k + 2^7 X_C + (sum S'_k + sum T_k),   X = sum_k k T_k  (k = 1 at most)
            //    general mode: total = sum_w 2^(c w) W_w by Horner from the top window down; the factors ride along the c
            //    doublings between two windows, so a set costs c doublings whatever its block structure;
            //    fixed-base mode: one set, the same chain.
            int lc = 0;
            while ((1u << lc) < C) ++lc;
            const int groups = (int)ws.groups;
            const uint32_t* rows_fin = h_final;
            const uint32_t* cols_fin = h_final + (size_t)groups * bpr * 2 * XW;
            auto pt = [](const uint32_t* p) { return HF::xyzz_from_device(p); };
            bool first_set = true;
            for (int g = groups - 1; g >= 0; --g) {
                // X = sum_k k T_k as a running sum of suffixes (from the top block down): two additions per block
                XYZZ<HF> sum_s = xyzz_inf<HF>(), sum_t = xyzz_inf<HF>(), x_r = xyzz_inf<HF>();
                for (int k = (int)bpr - 1; k >= 0; --k) {
                    const uint32_t* q = rows_fin + ((size_t)g * bpr + k) * 2 * XW;
                    sum_s = xyzz_add<HF>(sum_s, pt(q));
                    if (k > 0) {
                        sum_t = xyzz_add<HF>(sum_t, pt(q + XW));  // T_k + .. + T_top
                        x_r = xyzz_add<HF>(x_r, sum_t);
                    } else {
                        sum_t = xyzz_add<HF>(sum_t, pt(q + XW));
                    }
                }
                XYZZ<HF> sum_sc = xyzz_inf<HF>(), x_c = xyzz_inf<HF>(), suf_c = xyzz_inf<HF>();
                for (int k = (int)bpc - 1; k >= 0; --k) {
                    const uint32_t* q = cols_fin + ((size_t)g * bpc + k) * 2 * XW;
                    sum_sc = xyzz_add<HF>(sum_sc, pt(q));
                    if (k > 0) {
                        suf_c = xyzz_add<HF>(suf_c, pt(q + XW));
                        x_c = xyzz_add<HF>(x_c, suf_c);
                    }
                }
                // terms in descending order of their exponent; `at` = exponent the accumulator currently sits at
                int at = first_set ? -1 : c;
                auto step = [&](int exp, const XYZZ<HF>& term, bool present) {
                    if (!present) return;
                    if (at >= 0) for (int k = 0; k < at - exp; ++k) total = xyzz_dbl<HF>(total);
                    total = xyzz_add<HF>(total, term);
                    at = exp;
                };
                if (bpr > 1 && lc + WS_BLOCK_LOG > c && !first_set) return fail(ZK_ERR_ARG, "internal: reduction layout does not fit the window");
                step(lc + WS_BLOCK_LOG, x_r, bpr > 1);
                step(lc, sum_s, true);
                step(WS_BLOCK_LOG, x_c, bpc > 1);
                step(0, xyzz_add<HF>(sum_sc, sum_t), true);
                first_set = false;
            }
            if (!pre) for (int k = 0; k < c * q_first; ++k) total = xyzz_dbl<HF>(total);
            float sort_ms = 0, acc_ms = 0, red_ms = 0, t = 0;
            if (hipEventElapsedTime(&t, ws.ev_begin, ws.ev_acc0) == hipSuccess) sort_ms += t;
            if (hipEventElapsedTime(&t, ws.ev_accs, ws.ev_acc1) == hipSuccess) acc_ms += t;
            if (hipEventElapsedTime(&t, ws.ev_acc1, ws.ev_done) == hipSuccess) red_ms += t;
            if (hipEventElapsedTime(&t, ev_start, ev_digits) == hipSuccess) sort_ms += t;
            timings[0] = sort_ms;   // digits + sort
            timings[1] = acc_ms;    // accumulate kernel
            timings[2] = red_ms;    // combine + reduction + D2H
        }
        HF::affine_to_canonical(xyzz_to_affine<HF>(total), out);
        if (q_m > 0) {
            float t = 0;
            ZK_HIP(hipEventRecord(ev_digits, q_stream));  // reuse as "host tail done" marker
            ZK_HIP(hipEventSynchronize(ev_digits));
            (void)hipEventElapsedTime(&timings[3], ev_end, ev_digits);
            (void)hipEventElapsedTime(&timings[4], ev_start, ev_digits);
            (void)t;
        }
        return ZK_OK;
    }
};

// batch_multi_scalar_g1/_g2: out[i] = k_i * P_i (canonical affine).  One base for many scalars (what Groth16.setup
// does) goes through the fixed-base table: <= 16 mixed additions per scalar; everything ends in the batched normalisation.
constexpr uint64_t FIXED_BASE_MIN = 2048;  // below this the table (2^19 rows) costs more than it saves

template <class G>
static int batch_mul_impl(uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast, uint64_t* out) {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
    if (n == 0) return ZK_OK;
    uint32_t *ds = nullptr, *db = nullptr, *dout = nullptr, *temp = nullptr;
    int rc = ZK_OK;
    const bool fixed = broadcast && n >= FIXED_BASE_MIN;
    uint64_t nb = broadcast ? 1 : n;
    ZK_ALLOC(&ds, n * FrP::W * 4);
    do {
        if (dev_alloc_cached((void**)&temp, n * XW * 4) != ZK_OK || dev_alloc_cached((void**)&dout, n * AW * 4) != ZK_OK) { rc = ZK_ERR_HIP; break; }
        if (hipMemcpy(ds, scalars, n * FrP::W * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        if (fixed) {
            FixedTable<G>& ft = FixedTable<G>::get();
            std::lock_guard<std::mutex> lock(ft.mu);
            if ((rc = ft.ensure(bases))) break;
            FixedBias bias;
            memset(&bias, 0, sizeof(bias));
            for (int j = 0; j < ft.nwin; ++j) {
                int bit = j * FIXED_C + (FIXED_C - 1);
                bias.v[bit >> 5] |= 1u << (bit & 31);
            }
            hipLaunchKernelGGL(fixed_mul_kernel<G>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ds, n, ft.d_table, ft.nwin, bias, temp);
            if (hipDeviceSynchronize() != hipSuccess) { rc = fail(ZK_ERR_HIP, "fixed-base multiplication kernel failed"); break; }
        } else {
            if (dev_alloc_cached((void**)&db, nb * AW * 4) != ZK_OK) { rc = ZK_ERR_HIP; break; }
            if (hipMemcpy(db, bases, nb * AW * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
            hipLaunchKernelGGL(varbase_mul_kernel<G>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, ds, db, broadcast, n, temp);
        }
        hipLaunchKernelGGL(normalize_kernel<G>, dim3((unsigned)((n + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E))),
                           dim3(NORM_THREADS), 0, 0, temp, n, dout, 1);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * AW * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "batch_mul kernel / D2H failed"); break; }
    } while (0);
    dev_free_cached(ds); dev_free_cached(db); dev_free_cached(dout); dev_free_cached(temp);
    return rc;
}

// batched (de)compression of n points between host buffers: `to_bytes` != 0 encodes canonical affine rows, 0 decodes.
// On a failing point the first one (lowest index) decides the error, as a sequential loop over the file would.
template <class G>
static int codec_impl(uint64_t n, const void* in, void* out, int to_bytes, uint64_t* bad_index) {
    typedef typename G::F F;
    constexpr size_t ROW = (size_t)2 * F::LIMBS * 4, ENC = CodecLayout<G>::TOTAL;
    if (n == 0) return ZK_OK;
    const size_t in_bytes = n * (to_bytes ? ROW : ENC), out_bytes = n * (to_bytes ? ENC : ROW);
    uint8_t *din = nullptr, *dout = nullptr;
    unsigned long long* derr = nullptr;
    int rc = ZK_OK;
    ZK_ALLOC(&din, in_bytes);
    do {
        if (dev_alloc_cached((void**)&dout, out_bytes) != ZK_OK || dev_alloc_cached((void**)&derr, 8) != ZK_OK) { rc = ZK_ERR_HIP; break; }
        unsigned long long first = ~0ull;
        if (hipMemcpy(din, in, in_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(derr, &first, 8, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed");
            break;
        }
        const dim3 grid((unsigned)((n + 127) / 128)), block(128);
        if (to_bytes) hipLaunchKernelGGL(points_encode_kernel<G>, grid, block, 0, 0, (const uint32_t*)din, n, dout, derr);
        else hipLaunchKernelGGL(points_decode_kernel<G>, grid, block, 0, 0, (const uint8_t*)din, n, (uint32_t*)dout, derr);
        if (hipGetLastError() != hipSuccess || hipMemcpy(&first, derr, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "point codec kernel failed"); break; }
        if (first != ~0ull) {
            if (bad_index) *bad_index = (uint64_t)(first >> 8);
            rc = fail(ZK_ERR_POINT, codec_message((int)(first & 0xFF)));
            break;
        }
        if (hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    dev_free_cached(din); dev_free_cached(dout); dev_free_cached(derr);
    return rc;
}


#endif  // ZK_PART == 0

}  // namespace zkmi
