// msm_sort.hip.h -- stages 1-4 of the MSM pipeline: scalar digits (plain and split by the endomorphism), the one-level,
// bucket-range and two-level counting sorts, and the scans between them.  Group-independent: compiled into part 0 of every
// group translation unit only (msm_group.hip).  Pipeline overview: msm_impl.hip.h.
#pragma once
#include "msm_common.hip.h"

namespace zkmi {

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

}  // namespace zkmi
