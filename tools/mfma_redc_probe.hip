// mfma_redc_probe.hip -- can the idle matrix unit take the Montgomery REDUCTION of the bucket-accumulation kernel?
//
// Round-3 verdict, item 3: half of every field product in accumulate_kernel is `m * p` with a CONSTANT p (81 of 162
// v_mad_u64_u32 for the nine-limb fields, plus nine quotient products); as a digit Toeplitz matrix that is a dense contraction
// an i8 MFMA can take while the VALU does `a * b`.  This probe builds that path completely -- register layout conversion
// and carry recombination included -- checks it bit for bit against the VALU reduction on random inputs, and times both at
// the same occupancy.  BN254 Fq, R = 2^261 (nine 29-bit limbs), as csrc/field.hip.h.
//
//   VALU form   (what fp_mul does after a*b): for k = 0..8: m_k = (t_k * INV) mod 2^29; T += m_k p 2^(29k)  ->  T / 2^261
//   MFMA form   m = T_lo * N' mod 2^261 on the VALU (a 9x9 LOW product: 45 multiply-adds -- the quotient has to exist before a
//               matrix product can use it), m as 38 seven-bit digits (non-negative in i8), digits to the A-operand layout
//               through LDS, 4 x 3 v_mfma_i32_16x16x64_i8 against the Toeplitz matrix of p's 37 digits (columns 26..73 of the
//               74-column product: the low 261 bits of m p are known to be -T_lo, so only the top columns and a rounding constant
//               are needed), C tiles back to one lane per element through LDS, 48 column sums recombined into 29-bit limbs,
//               + T_hi.
//
// build:  hipcc --offload-arch=gfx950 -O3 -o build/probe/mfma_redc_probe tools/mfma_redc_probe.hip
// run:    build/probe/mfma_redc_probe            (prints one JSON object)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int N = 9, LB = 29;
constexpr uint32_t MASK = (1u << LB) - 1;
__device__ __constant__ uint32_t P[N] = {0x187cfd47, 0x10460b6, 0x1c72a34f, 0x2d522d0, 0x1585d978, 0x2db40c0, 0xa6e141, 0xe5c2634, 0x30644e};
__device__ __constant__ uint32_t NPRIME[N] = {0x4866389, 0x1e903c17, 0x129ab261, 0x1cfaca3d, 0x1da809ed, 0x5e80c19, 0x11af62bf, 0x16f23111, 0xff57a22};
constexpr uint32_t INV29 = 0x4866389;
__device__ __constant__ uint8_t P7[37] = {71, 122, 115, 67, 109, 2, 35, 16, 60, 26, 42, 14, 7, 45, 36, 53, 1, 47, 118, 66, 21, 48, 32, 91, 69, 32, 97, 77, 2, 52, 76, 112, 114, 28, 17, 3, 3};

constexpr int K0 = 26;            // first product column kept (7-bit columns): bit 182; columns 26 .. 73
constexpr int NCOL = 48;          // three N-tiles of 16
constexpr int OUT_STRIDE = 52;    // dwords per element row of the C image in LDS (16-byte aligned rows, 52 mod 32 = 20: few conflicts)

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- the VALU reduction: T (18 normalised limbs, T < p 2^261) -> T / 2^261 mod p, below 2p ------------------------------
__device__ __forceinline__ void redc_valu(const uint32_t* t, uint32_t* r) {
    uint64_t acc = 0;
    uint32_t m[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        acc += t[k];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P[k - i];
        m[k] = ((uint32_t)acc * INV29) & MASK;
        acc += (uint64_t)m[k] * P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        acc += t[k];
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) acc += (uint64_t)m[i] * P[k - i];
        r[k - N] = (uint32_t)acc & MASK;
        acc >>= LB;
    }
    r[N - 1] = (uint32_t)acc + t[2 * N - 1];
}

// ---- the MFMA reduction --------------------------------------------------------------------------------------------------
// LDS per wave: digit image 64 x 12 dwords, then (reused) C image 64 x OUT_STRIDE dwords
constexpr int WAVE_LDS_DWORDS = 64 * OUT_STRIDE;

__device__ __forceinline__ void redc_mfma(const uint32_t* t, uint32_t* r, uint32_t* lds /* this wave's region */, const v4i (&btile)[3]) {
    const int lane = threadIdx.x & 63;
    // (1) m = T_lo * N' mod 2^261: low product, 45 multiply-adds
    uint32_t m[N];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)t[i] * NPRIME[k - i];
            m[k] = (uint32_t)acc & MASK;
            acc >>= LB;
        }
    }
    // (2) 38 seven-bit digits, four to a dword (digit d = bits 7d .. 7d+6 of m)
    uint32_t dig[12];
#pragma unroll
    for (int w = 0; w < 12; ++w) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 4 * w + j, bit = 7 * d;
            if (d < 38) {
                const int li = bit / LB, sh = bit % LB;
                uint32_t v = m[li] >> sh;
                if (sh > LB - 7 && li + 1 < N) v |= m[li + 1] << (LB - sh);
                x |= (v & 127u) << (8 * j);
            }
        }
        dig[w] = x;
    }
    // (3) A-operand layout through LDS: lane l of row tile t wants digits 16 (l >> 4) .. +15 of element 16 t + (l & 15)
    uint4* img = reinterpret_cast<uint4*>(lds);
    img[lane * 3 + 0] = make_uint4(dig[0], dig[1], dig[2], dig[3]);
    img[lane * 3 + 1] = make_uint4(dig[4], dig[5], dig[6], dig[7]);
    img[lane * 3 + 2] = make_uint4(dig[8], dig[9], dig[10], dig[11]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    v4i acc[4][3];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int kb = lane >> 4;
        uint4 a4 = kb < 3 ? img[(16 * tt + (lane & 15)) * 3 + kb] : make_uint4(0, 0, 0, 0);
        v4i a = {(int)a4.x, (int)a4.y, (int)a4.z, (int)a4.w};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            v4i z = {0, 0, 0, 0};
            acc[tt][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, btile[j], z, 0, 0, 0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the digit image is dead: the C image reuses the region
    // (4) C tiles (col = lane & 15, row = 4 (lane >> 4) + reg) -> one row of 48 column sums per element
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) lds[(16 * tt + 4 * (lane >> 4) + v) * OUT_STRIDE + 16 * j + (lane & 15)] = (uint32_t)acc[tt][j][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t col[NCOL];
    const uint4* row = reinterpret_cast<const uint4*>(lds + lane * OUT_STRIDE);
#pragma unroll
    for (int i = 0; i < NCOL / 4; ++i) {
        uint4 q = row[i];
        col[4 * i] = q.x; col[4 * i + 1] = q.y; col[4 * i + 2] = q.z; col[4 * i + 3] = q.w;
    }
    // (5) recombination.  X = sum col[i] 2^(7 (K0 + i)) = m p - D with the dropped columns D < 2^203; T + m p is a multiple of
    // 2^261, so (T + m p) / 2^261 = floor((X + T + 2^203) / 2^261).  Limbs 6 .. 17 (bits 174 ..) are all that matter.
    uint64_t a64[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a64[i] = 0;
#pragma unroll
    for (int i = 0; i < NCOL; ++i) {
        const int bit = 7 * (K0 + i) - 6 * LB, li = bit / LB, sh = bit % LB;
        a64[li] += (uint64_t)col[i] << sh;
    }
    // + T (its limbs 6 .. 17; the limbs below add to zeros and carry nothing) + 2^203 (bit 0 of limb 7), one carry pass
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        int64_t v = (int64_t)a64[i] + carry + t[6 + i] + (i == 1 ? 1 : 0);
        if (i < 11) {
            a64[i] = (uint64_t)v & MASK;
            carry = v >> LB;
        } else {
            a64[i] = (uint64_t)v;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = (uint32_t)a64[3 + i];
}

// B operand tiles: B[k][c] = P7[(K0 + 16 j + c) - k] for lane l: column c = l & 15, k = 16 (l >> 4) + byte
__device__ __forceinline__ void make_btiles(v4i (&b)[3]) {
    const int lane = threadIdx.x & 63, c = lane & 15, kb = lane >> 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int byte = 0; byte < 16; ++byte) {
            const int k = 16 * kb + byte, idx = K0 + 16 * j + c - k;
            const uint32_t d = (k < 38 && idx >= 0 && idx < 37) ? P7[idx] : 0u;
            w[byte >> 2] |= d << (8 * (byte & 3));
        }
        b[j] = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
}

template <int MODE>   // 0 = VALU, 1 = MFMA
__global__ __launch_bounds__(256) void probe_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
    extern __shared__ uint32_t lds[];
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t[2 * N], r[N];
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) t[i] = in[gid * 2 * N + i];
    v4i bt[3];
    if (MODE == 1) make_btiles(bt);
    uint32_t* wl = lds + (threadIdx.x >> 6) * WAVE_LDS_DWORDS;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) redc_valu(t, r);
        else redc_mfma(t, r, wl, bt);
        // feed the result back as the next low half (keeps the loop dependent, the values stay below p 2^261)
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = r[i] & MASK;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[gid * N + i] = r[i];
}

// independent oracle: bit-serial Montgomery reduction (261 times: make T even by adding p, halve), one element per lane
__global__ void oracle_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    uint32_t t[2 * N + 1];
    for (int i = 0; i < 2 * N; ++i) t[i] = in[gid * 2 * N + i];
    t[2 * N] = 0;
    for (int step = 0; step < LB * N; ++step) {
        if (t[0] & 1) {
            uint32_t c = 0;
            for (int i = 0; i < 2 * N + 1; ++i) {
                uint32_t v = t[i] + (i < N ? P[i] : 0u) + c;
                t[i] = v & MASK;
                c = v >> LB;
            }
        }
        for (int i = 0; i < 2 * N; ++i) t[i] = (t[i] >> 1) | ((t[i + 1] & 1u) << (LB - 1));
        t[2 * N] >>= 1;
    }
    for (int i = 0; i < N; ++i) out[gid * N + i] = t[i];   // below 2^(252 + 1): the top limb stays within 29 bits
}

int main() {
    const int blocks = 1024, threads = 256;           // 4096 waves = 4 per SIMD on 256 CUs, once
    const size_t n = (size_t)blocks * threads;
    std::vector<uint32_t> h_in(n * 2 * N);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t e = 0; e < n; ++e) {
        for (int i = 0; i < 2 * N; ++i) h_in[e * 2 * N + i] = (uint32_t)next() & MASK;
        h_in[e * 2 * N + 2 * N - 1] &= (1u << 20) - 1;   // T < 2^(261 + 252) < p 2^261
        if (e % 97 == 0) for (int i = 0; i < N; ++i) h_in[e * 2 * N + i] = 0;          // T_lo = 0: no carry out of the low half
        if (e % 101 == 0) for (int i = 0; i < 6; ++i) h_in[e * 2 * N + i] = 0;         // only the upper low limbs set
    }
    uint32_t *d_in, *d_o0, *d_o1;
    HIP_OK(hipMalloc(&d_in, h_in.size() * 4));
    HIP_OK(hipMalloc(&d_o0, n * N * 4));
    HIP_OK(hipMalloc(&d_o1, n * N * 4));
    HIP_OK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
    const size_t lds_bytes = (size_t)(threads / 64) * WAVE_LDS_DWORDS * 4;
    HIP_OK(hipFuncSetAttribute((const void*)probe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    // correctness: one reduction each way, all elements equal; a sample against host big-integer arithmetic
    probe_kernel<0><<<blocks, threads, 0>>>(d_in, d_o0, 1);
    probe_kernel<1><<<blocks, threads, lds_bytes>>>(d_in, d_o1, 1);
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint32_t> r0(n * N), r1(n * N);
    HIP_OK(hipMemcpy(r0.data(), d_o0, r0.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(r1.data(), d_o1, r1.size() * 4, hipMemcpyDeviceToHost));
    size_t mismatches = 0, first_bad = n;
    for (size_t e = 0; e < n; ++e)
        for (int i = 0; i < N; ++i)
            if (r0[e * N + i] != r1[e * N + i]) { ++mismatches; if (first_bad == n) first_bad = e; break; }
    // the VALU form against the bit-serial oracle, every element
    oracle_kernel<<<blocks, threads>>>(d_in, d_o1, n);
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint32_t> ro(n * N);
    HIP_OK(hipMemcpy(ro.data(), d_o1, ro.size() * 4, hipMemcpyDeviceToHost));
    int host_ok = 1;
    for (size_t k = 0; k < n * N; ++k) if (ro[k] != r0[k]) { host_ok = 0; break; }
    // timing
    const int iters = 2000;
    float ms[2] = {0, 0};
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipEventRecord(e0));
            if (mode == 0) probe_kernel<0><<<blocks, threads, 0>>>(d_in, d_o0, iters);
            else probe_kernel<1><<<blocks, threads, lds_bytes>>>(d_in, d_o1, iters);
            HIP_OK(hipEventRecord(e1));
            HIP_OK(hipEventSynchronize(e1));
            HIP_OK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
    }
    const double reds = (double)n * iters;
    printf("{\"elements\": %zu, \"iterations\": %d, \"mfma_equals_valu\": %s, \"mismatching_elements\": %zu, \"first_mismatch\": %zd, "
           "\"valu_equals_bit_serial_oracle\": %s, \"valu_ms\": %.3f, \"mfma_ms\": %.3f, \"valu_ps_per_reduction\": %.2f, "
           "\"mfma_ps_per_reduction\": %.2f, \"mfma_over_valu\": %.3f, \"lds_bytes_per_workgroup_mfma\": %zu}\n",
           n, iters, mismatches == 0 ? "true" : "false", mismatches, first_bad == n ? (ssize_t)-1 : (ssize_t)first_bad,
           host_ok ? "true" : "false", ms[0], ms[1], ms[0] * 1e9 / reds, ms[1] * 1e9 / reds, ms[1] / ms[0], lds_bytes);
    return mismatches == 0 && host_ok ? 0 : 1;
}
