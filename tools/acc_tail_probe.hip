// tools/acc_tail_probe.hip -- where the time of a one-round bucket-accumulation launch goes (not part of the library).
// The accumulate kernel is fastest per addition with short segments (many rounds of waves) and 8-10 % slower with the 64-entry
// segments that fill the chip exactly once.  This probe runs the same XYZZ chain (BN254 G1, bases gathered from a 128 MiB table)
// with k additions per lane for k = 64 / 32 / 16 / 8 and records every wave's start, end (wall_clock64, 100 MHz) and hardware
// id, then prints how many waves were resident over time and how the end times spread over the XCDs.
//   hipcc -O3 --offload-arch=gfx950 -o build/probe/acc_tail_probe tools/acc_tail_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zksnake_amd/csrc/msm_common.hip.h"
#include "../zksnake_amd/csrc/curve.hip.h"
#include "../zksnake_amd/csrc/curve_consts.h"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef FpOps<BnFqParams> F;
constexpr int AW = 2 * F::LIMBS;

// MODE 0: plain; 1: the wave lowers its own issue priority as it passes each quarter of its segment (s_setprio), so that the
// waves that share a SIMD stay within a quarter of each other; 2: workgroups of 16 waves (four per SIMD) meet at a barrier every
// eight additions
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void chain_kernel(const uint32_t* table, const uint32_t* idx, int k, uint32_t* out,
                                                    unsigned long long* stamps, uint32_t* hwid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool lead = (threadIdx.x & 63) == 0;
    unsigned long long t0 = 0;
    if (lead) t0 = wall_clock64();
    XYZZ<F> acc = xyzz_inf<F>();
    if (MODE == 1) __builtin_amdgcn_s_setprio(3);
    const int q1 = k - k / 4, q2 = k - k / 10, q3 = k - (k + 31) / 32;
    for (int j = 0; j < k; ++j) {
        if (MODE == 1) {
            if (j == q1) __builtin_amdgcn_s_setprio(2);
            else if (j == q2) __builtin_amdgcn_s_setprio(1);
            else if (j == q3) __builtin_amdgcn_s_setprio(0);
        }
        if (MODE == 2 && (j & 7) == 0 && j) __syncthreads();
        const uint32_t ref = idx[(size_t)t * k + j];
        xyzz_add_affine_mem<F>(acc, table + (size_t)(ref & 0x7FFFFFFFu) * AW, (ref >> 31) != 0);
    }
    xyzz_relaxed_finish<F>(acc);
    uint32_t w[F::LIMBS];
    F::store(w, F::add(F::add(acc.X, acc.Y), F::add(acc.ZZ, acc.ZZZ)));
    for (int i = 0; i < F::LIMBS; ++i) out[(size_t)t * F::LIMBS + i] = w[i];
    if (lead) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[2 * (t >> 6)] = t0;
        stamps[2 * (t >> 6) + 1] = wall_clock64();
        hwid[t >> 6] = (hw & 0xFFFFFFu) | ((xcc & 0xFu) << 24);
    }
}

// the library's accumulate kernel (msm_accumulate.hip.h) with the same time stamps: sorted entry list, bucket offsets, binary
// search for the first bucket, a flush at every bucket end
template <class FF>
__device__ __forceinline__ uint32_t* slot_of(uint32_t* partials, uint32_t* buckets, const uint32_t* run_start, const uint32_t* bucket_start,
                                             uint32_t key, uint32_t t, uint32_t seg_len) {
    constexpr int XW = 4 * FF::LIMBS;
    const uint32_t r0 = run_start[key];
    if (run_start[key + 1] - r0 == 1) return buckets + (size_t)key * XW;
    return partials + (size_t)(r0 + t - bucket_start[key] / seg_len) * XW;
}
template <int PRIO>
__global__ __launch_bounds__(256) void real_kernel(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                   const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ run_start,
                                                   uint32_t n_keys, uint32_t seg_len, uint32_t* __restrict__ partials,
                                                   uint32_t* __restrict__ buckets, unsigned long long* stamps, uint32_t* hwid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool lead = (threadIdx.x & 63) == 0;
    unsigned long long t0 = 0;
    if (lead) t0 = wall_clock64();
    const uint32_t total = bucket_start[n_keys];
    const uint32_t begin = t * seg_len;
    if (begin < total) {
        uint32_t end = begin + seg_len;
        if (end > total) end = total;
        uint32_t lo = 0, hi = n_keys;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (bucket_start[mid] <= begin) lo = mid; else hi = mid;
        }
        uint32_t key = lo;
        uint32_t next = bucket_start[key + 1];
        XYZZ<F> acc = xyzz_inf<F>();
        // the step counter is the same in every lane, but only a scalar makes the branch around s_setprio a scalar branch: under a
        // lane mask the compiler lets the (scalar) instruction run whatever the mask is, i.e. in every iteration
        const uint32_t q1 = seg_len - seg_len / 4, q2 = seg_len - seg_len / 10, q3 = seg_len - (seg_len + 31) / 32;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        for (uint32_t e = begin; e < end; ++e) {
            if (PRIO) {
                const uint32_t j = __builtin_amdgcn_readfirstlane(e - begin);
                if (j == q1) __builtin_amdgcn_s_setprio(2);
                else if (j == q2) __builtin_amdgcn_s_setprio(1);
                else if (j == q3) __builtin_amdgcn_s_setprio(0);
            }
            if (e == next) {
                xyzz_relaxed_finish<F>(acc);
                store_xyzz<F>(slot_of<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), acc);
                acc = xyzz_inf<F>();
                do { ++key; next = bucket_start[key + 1]; } while (next <= e);
            }
            uint32_t ref = sorted[e];
            xyzz_add_affine_mem<F>(acc, bases + (size_t)(ref & 0x7FFFFFFFu) * AW, (ref >> 31) != 0);
        }
        xyzz_relaxed_finish<F>(acc);
        store_xyzz<F>(slot_of<F>(partials, buckets, run_start, bucket_start, key, t, seg_len), acc);
    }
    if (lead) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[2 * (t >> 6)] = t0;
        stamps[2 * (t >> 6) + 1] = wall_clock64();
        hwid[t >> 6] = (xcc & 0xFu) << 24;
    }
}

static void report(const char* what, int k, unsigned lanes, unsigned waves, float ms, unsigned long long* stamps, double adds_done) {
    std::vector<unsigned long long> st(2 * (size_t)waves);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (unsigned w = 0; w < waves; ++w) { tmin = std::min(tmin, st[2 * w]); tmax = std::max(tmax, st[2 * w + 1]); }
    const double span_us = (tmax - tmin) / 100.0;
    printf("%s k=%2d lanes=%7u waves=%6u  %7.3f ms (event)  span %8.1f us  %5.1f ps/add\n", what, k, lanes, waves, ms, span_us, ms * 1e9 / adds_done);
    const int S = 20;
    std::vector<double> resident(S, 0.0);
    std::vector<double> dur(waves);
    for (unsigned w = 0; w < waves; ++w) {
        const double a = (st[2 * w] - tmin) / 100.0, b = (st[2 * w + 1] - tmin) / 100.0;
        dur[w] = b - a;
        for (int i = 0; i < S; ++i) {
            const double lo = span_us * i / S, hi2 = span_us * (i + 1) / S;
            const double ov = std::min(b, hi2) - std::max(a, lo);
            if (ov > 0) resident[i] += ov / (hi2 - lo);
        }
    }
    printf("   resident waves per 5 %% slice:");
    for (int i = 0; i < S; ++i) printf(" %4.0f", resident[i]);
    std::sort(dur.begin(), dur.end());
    printf("\n   wave duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", dur[0], dur[waves / 10], dur[waves / 2],
           dur[waves * 9 / 10], dur[waves - 1]);
}

static void run_real(const uint32_t* table, size_t rows, unsigned long long* stamps, uint32_t* hwid) {
    // eight bucket sets of 2^15 buckets, 2^24 entries: bucket sizes binomial around 64, as a 2^20-point split-scalar MSM has them
    const uint32_t n_keys = 8u << 15, total = 1u << 24;
    std::vector<uint32_t> cnt(n_keys, 0), bstart(n_keys + 1), sorted(total);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint32_t i = 0; i < total; ++i) ++cnt[(uint32_t)(rnd() >> 20) % n_keys];
    bstart[0] = 0;
    for (uint32_t k = 0; k < n_keys; ++k) bstart[k + 1] = bstart[k] + cnt[k];
    for (auto& w : sorted) w = (uint32_t)((rnd() >> 20) % rows) | ((uint32_t)(rnd() >> 63) << 31);
    uint32_t *d_sorted, *d_bstart, *d_rstart, *d_partials, *d_buckets;
    CK(hipMalloc(&d_sorted, (size_t)total * 4));
    CK(hipMalloc(&d_bstart, (size_t)(n_keys + 1) * 4));
    CK(hipMalloc(&d_rstart, (size_t)(n_keys + 1) * 4));
    CK(hipMemcpy(d_sorted, sorted.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bstart, bstart.data(), (size_t)(n_keys + 1) * 4, hipMemcpyHostToDevice));
    constexpr int XW = 4 * F::LIMBS;
    CK(hipMalloc(&d_partials, ((size_t)total / 8 + n_keys + 8) * XW * 4));
    CK(hipMalloc(&d_buckets, (size_t)n_keys * XW * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t cfg : {64u, 32u, 16u, 8u, 1064u, 1032u}) {
        const uint32_t seg_len = cfg % 1000u;
        const bool prio = cfg >= 1000u;
        std::vector<uint32_t> rstart(n_keys + 1);
        rstart[0] = 0;
        for (uint32_t k = 0; k < n_keys; ++k) {
            const uint32_t s0 = bstart[k], s1 = bstart[k + 1];
            rstart[k + 1] = rstart[k] + (s1 > s0 ? 1 + (s1 - 1) / seg_len - s0 / seg_len : 0);
        }
        CK(hipMemcpy(d_rstart, rstart.data(), (size_t)(n_keys + 1) * 4, hipMemcpyHostToDevice));
        const unsigned lanes = (total + seg_len - 1) / seg_len, waves = (lanes + 255) / 256 * 4;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (prio) hipLaunchKernelGGL(real_kernel<1>, dim3((lanes + 255) / 256), dim3(256), 0, 0, table, d_sorted, d_bstart, d_rstart, n_keys, seg_len, d_partials, d_buckets, stamps, hwid);
            else hipLaunchKernelGGL(real_kernel<0>, dim3((lanes + 255) / 256), dim3(256), 0, 0, table, d_sorted, d_bstart, d_rstart, n_keys, seg_len, d_partials, d_buckets, stamps, hwid);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        report(prio ? "library kernel + priority steps" : "library kernel", (int)seg_len, lanes, waves, ms, stamps, (double)total);
    }
}

int main() {
    const size_t rows = (size_t)1 << 21;   // 2^20 points and their endomorphism images, 64 B each
    uint32_t *table, *idx, *out, *hwid;
    unsigned long long* stamps;
    CK(hipMalloc(&table, rows * AW * 4));
    std::vector<uint32_t> h(rows * AW);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16) & 0x0FFFFFFFu; }
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const size_t adds = 16ull << 20;
    std::vector<uint32_t> hi(adds);
    for (auto& w : hi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)((s >> 20) % rows); }
    CK(hipMalloc(&idx, hi.size() * 4));
    CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (adds / 8) * F::LIMBS * 4));
    const size_t max_waves = adds / 8 / 64;
    CK(hipMalloc(&stamps, max_waves * 16));
    CK(hipMalloc(&hwid, max_waves * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode_k : {64, 8, 1064, 2064}) {
        const int mode = mode_k / 1000, k = mode_k % 1000;
        const unsigned lanes = (unsigned)(adds / k) / 1024 * 1024;
        const unsigned waves = lanes / 64;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((chain_kernel<0, 256>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, k, out, stamps, hwid);
            else if (mode == 1) hipLaunchKernelGGL((chain_kernel<1, 256>), dim3(lanes / 256), dim3(256), 0, 0, table, idx, k, out, stamps, hwid);
            else hipLaunchKernelGGL((chain_kernel<2, 1024>), dim3(lanes / 1024), dim3(1024), 0, 0, table, idx, k, out, stamps, hwid);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> st(2 * (size_t)waves);
        std::vector<uint32_t> hw(waves);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hw.data(), hwid, hw.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        for (unsigned w = 0; w < waves; ++w) { tmin = std::min(tmin, st[2 * w]); tmax = std::max(tmax, st[2 * w + 1]); }
        const double span_us = (tmax - tmin) / 100.0;
        printf("mode=%d k=%2d lanes=%7u waves=%6u  %7.3f ms (event)  span of the stamps %8.1f us  %5.1f ps/add\n", mode, k, lanes, waves, ms, span_us,
               ms * 1e9 / ((double)lanes * k));
        // resident waves over time, in 20 slices of the span
        const int S = 20;
        std::vector<double> resident(S, 0.0);
        for (unsigned w = 0; w < waves; ++w) {
            const double a = (st[2 * w] - tmin) / 100.0, b = (st[2 * w + 1] - tmin) / 100.0;
            for (int i = 0; i < S; ++i) {
                const double lo = span_us * i / S, hi2 = span_us * (i + 1) / S;
                const double ov = std::min(b, hi2) - std::max(a, lo);
                if (ov > 0) resident[i] += ov / (hi2 - lo);
            }
        }
        printf("   resident waves per 5 %% slice:");
        for (int i = 0; i < S; ++i) printf(" %4.0f", resident[i]);
        printf("\n");
        // wave durations and, per XCC, the time its last wave ended
        std::vector<double> dur(waves);
        double xcc_end[16] = {0}; unsigned xcc_waves[16] = {0};
        for (unsigned w = 0; w < waves; ++w) {
            dur[w] = (st[2 * w + 1] - st[2 * w]) / 100.0;
            const unsigned x = (hw[w] >> 24) & 15;
            xcc_end[x] = std::max(xcc_end[x], (st[2 * w + 1] - tmin) / 100.0);
            ++xcc_waves[x];
        }
        std::sort(dur.begin(), dur.end());
        printf("   wave duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", dur[0], dur[waves / 10], dur[waves / 2],
               dur[waves * 9 / 10], dur[waves - 1]);
        printf("   per XCC (waves, last end us):");
        for (int x = 0; x < 8; ++x) printf(" %u/%.0f", xcc_waves[x], xcc_end[x]);
        printf("\n");
    }
    run_real(table, rows, stamps, hwid);
    CK(hipGetLastError());
    return 0;
}
