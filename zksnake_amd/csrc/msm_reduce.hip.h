// msm_reduce.hip.h -- stages 6-7 of the MSM pipeline: per-bucket combination of the segment partials and the bucket
// reduction sum_b (b + 1) B_b on lane pairs (pair.hip.h).  Pipeline overview: msm_impl.hip.h.
#pragma once
#include "msm_common.hip.h"

namespace zkmi {

// ---- 6. combine ---------------------------------------------------------------------------------
// bucket = sum of its runs, three tiers in ONE launch (every dependent launch costs ~5 us of latency, and the two upper
// tiers are empty unless the scalars are skewed).  A point is held by a lane pair (pair.hip.h) in all tiers.
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
// Both kernels hold a point as a lane pair (pair.hip.h): an addition is seven multiplications deep instead of
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

}  // namespace zkmi
