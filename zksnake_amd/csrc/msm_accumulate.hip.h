// msm_accumulate.hip.h -- stage 5 of the MSM pipeline, the dominant kernel (bucket accumulation over uniform segments of the
// sorted entry list), and the conversion of the bases to Montgomery form.  Pipeline overview: msm_impl.hip.h.
#pragma once
#include "msm_common.hip.h"
#include "fp2_split.hip.h"
// waves per SIMD the pair-split G2 kernels are compiled for (the second launch-bound parameter); tuned by measurement
#ifndef ZK_SPLIT_WAVES_BN
#define ZK_SPLIT_WAVES_BN 3
#endif
#ifndef ZK_SPLIT_WAVES_BLS
#define ZK_SPLIT_WAVES_BLS 1
#endif

namespace zkmi {

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

// Waves per SIMD the kernel is compiled for (a workgroup is four waves, one per SIMD).  The BN254 G2 step needs 265 registers
// left to itself -- one wave per SIMD -- and fits the 256 of two waves at the price of seven spilled registers.
template <class G> struct AccumulateWaves { static constexpr int PER_SIMD = 1; };
template <> struct AccumulateWaves<Bn254G2> { static constexpr int PER_SIMD = 2; };

template <class G>
__global__ __launch_bounds__(256, AccumulateWaves<G>::PER_SIMD) void accumulate_kernel(const uint32_t* __restrict__ bases,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ bucket_start,
                                                         const uint32_t* __restrict__ run_start, uint32_t n_keys,
                                                         uint32_t seg_len, uint32_t prio_steps, uint32_t* __restrict__ partials,
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
    // The SIMD issues from its oldest wave first.  Left alone, the four waves that share a SIMD finish one after the other -- at 28,
    // 50, 75 and 100 % of the kernel's time -- and the last of them runs alone, at the issue rate of a lone wave, for the last quarter
    // (tools/acc_tail_probe.hip: 90 ps per addition against 71 with eight-entry segments).  So every wave lowers its own issue priority
    // as it passes 75, 90 and 97 % of its segment: the waves of a SIMD wait for each other there, all four stay resident to the end
    // and only the last 3 % of the work runs unsynchronised.  Only when the kernel has the chip to itself (prio_steps: a plain run of
    // a plan): inside a proof the slots a finished wave leaves are taken by the kernels of the other plans at once, the steps gain
    // nothing there and starve those kernels (measured).  The step counter is the same in every lane, but only a scalar makes the
    // branch around s_setprio a scalar branch: under a lane mask the compiler lets the (scalar) instruction run whatever the mask
    // is, i.e. in every iteration.
    const uint32_t q1 = seg_len - seg_len / 4, q2 = seg_len - seg_len / 10, q3 = seg_len - (seg_len + 31) / 32;
    if (prio_steps) __builtin_amdgcn_s_setprio(3);
    for (uint32_t e = begin; e < end; ++e) {
        if (prio_steps) {
            const uint32_t step = __builtin_amdgcn_readfirstlane(e - begin);
            if (step == q1) __builtin_amdgcn_s_setprio(2);
            else if (step == q2) __builtin_amdgcn_s_setprio(1);
            else if (step == q3) __builtin_amdgcn_s_setprio(0);
        }
        if (e == next) {
            // the bucket ends inside this segment: flush its run, move to the next non-empty bucket
            xyzz_relaxed_finish<F>(acc);
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
    xyzz_relaxed_finish<F>(acc);
    store_xyzz<F>(run_slot<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), acc);
}

// ---- 5'. the same kernel for the Fp2 groups with every value split over a LANE PAIR (fp2_split.hip.h): the even lane of a pair
// holds the c0 components of the accumulator and of the base, the odd lane the c1 components; a pair owns one segment.  Half the
// state per lane, the same products, the same results, 14 % more instructions in all (operand selects, exchanges, the sums and
// differences both lanes form).  Measured at 2^20 pairs (tools/group_msm_bench.py, one box, accumulate kernel only):
//   BLS12-381 G2   401 registers, ONE wave per SIMD: 7.98 ms   ->  split, 245 registers, two waves: 7.49-7.61 ms   (default ON)
//                  (held to three waves: 8.00)
//   BN254 G2       256 registers, two waves: 3.10 ms            ->  split at two / three / four waves: 3.35 / 3.27 / 3.48   (default OFF:
//                  the one-lane kernel already runs two waves; a third does not buy back the extra instructions)
// Plan option "split_pairs" (-1 = the group's default, 0, 1) or ZKMI_SPLIT_PAIRS=0/1 choose per plan.
template <class G> struct AccumulateSplit { static constexpr bool ON = false; static constexpr bool DEFAULT = false; static constexpr int WAVES = 1; };
template <> struct AccumulateSplit<Bn254G2> { static constexpr bool ON = true; static constexpr bool DEFAULT = false; static constexpr int WAVES = ZK_SPLIT_WAVES_BN; };
template <> struct AccumulateSplit<Bls381G2> { static constexpr bool ON = true; static constexpr bool DEFAULT = true; static constexpr int WAVES = ZK_SPLIT_WAVES_BLS; };

template <class P>
__device__ __forceinline__ Fp<P> load_component(const uint32_t* p) {
    uint32_t w[P::W];
    load_words<P::W>(w, p);
    return fp_load<P>(w);
}
// this lane's component of the four coordinates of an XYZZ row [X.c0 X.c1 | Y.c0 Y.c1 | ZZ.. | ZZZ..]
template <class P>
__device__ __forceinline__ void store_split_xyzz(uint32_t* row, bool odd, const SplitXYZZ<P>& a) {
    constexpr int W = P::W;
    uint32_t w[W];
    uint32_t* base = row + (odd ? W : 0);
    fp_store<P>(w, a.X);   store_words<W>(base, w);
    fp_store<P>(w, a.Y);   store_words<W>(base + 2 * W, w);
    fp_store<P>(w, a.ZZ);  store_words<W>(base + 4 * W, w);
    fp_store<P>(w, a.ZZZ); store_words<W>(base + 6 * W, w);
}

template <class G>
__global__ __launch_bounds__(256, AccumulateSplit<G>::WAVES) void accumulate_split_kernel(const uint32_t* __restrict__ bases,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ bucket_start,
                                                         const uint32_t* __restrict__ run_start, uint32_t n_keys,
                                                         uint32_t seg_len, uint32_t prio_steps, uint32_t* __restrict__ partials,
                                                         uint32_t* __restrict__ buckets) {
    typedef typename G::F F;
    typedef typename F::Params P;
    constexpr int W = P::W;        // words per component
    constexpr int AW = 4 * W;      // affine row: x.c0 x.c1 y.c0 y.c1
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1;    // the pair's segment
    const bool odd = (gt & 1) != 0;
    const uint32_t total = bucket_start[n_keys];
    const uint32_t begin = t * seg_len;
    if (begin >= total) return;    // both lanes of a pair leave together
    uint32_t end = begin + seg_len;
    if (end > total) end = total;
    uint32_t lo = 0, hi = n_keys;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (bucket_start[mid] <= begin) lo = mid; else hi = mid;
    }
    uint32_t key = lo;
    uint32_t next = bucket_start[key + 1];
    SplitXYZZ<P> acc = sp_inf<P>();
    const uint32_t q1 = seg_len - seg_len / 4, q2 = seg_len - seg_len / 10, q3 = seg_len - (seg_len + 31) / 32;
    if (prio_steps) __builtin_amdgcn_s_setprio(3);   // the priority steps of accumulate_kernel, see there
    for (uint32_t e = begin; e < end; ++e) {
        if (prio_steps) {
            const uint32_t step = __builtin_amdgcn_readfirstlane(e - begin);
            if (step == q1) __builtin_amdgcn_s_setprio(2);
            else if (step == q2) __builtin_amdgcn_s_setprio(1);
            else if (step == q3) __builtin_amdgcn_s_setprio(0);
        }
        if (e == next) {
            acc.X = fp_reduce_2p<P>(acc.X);   // xyzz_relaxed_finish, by component
            store_split_xyzz<P>(run_slot<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), odd, acc);
            acc = sp_inf<P>();
            do {
                ++key;
                next = bucket_start[key + 1];
            } while (next <= e);
        }
        const uint32_t ref = sorted[e];
        const uint32_t* src = bases + (size_t)(ref & 0x7FFFFFFFu) * AW + (odd ? W : 0);
        const Fp<P> qx = load_component<P>(src), qy = load_component<P>(src + 2 * W);
        sp_add_affine<P>(acc, qx, qy, (ref >> 31) != 0, odd);
    }
    acc.X = fp_reduce_2p<P>(acc.X);
    store_split_xyzz<P>(run_slot<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), odd, acc);
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
            typedef typename F::Params FP;
            store_affine<F>(out + 2 * i * AW, p);
            const Fp<FP> beta = fp_from_canonical<FP>(GlvOf<G>::P::BETA);
            if constexpr (GlvOf<G>::P::NEG_Y) {   // G2: (c x, -y), c in the base field
                p.x = {fp_mul<FP>(p.x.c0, beta), fp_mul<FP>(p.x.c1, beta)};
                p.y = F::neg(p.y);
            } else {                              // G1: (beta x, y)
                p.x = fp_mul<FP>(p.x, beta);
            }
            store_affine<F>(out + (2 * i + 1) * AW, p);
            return;
        }
    }
    store_affine<F>(out + i * AW, p);
}

}  // namespace zkmi
