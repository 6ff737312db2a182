// tools/ubench.hip -- VALU issue-rate microbenchmark for gfx950 (design input for field.cuh).
// Measures sustained per-CU throughput of the instructions a big-integer Montgomery product
// can be built from, and of fp_mul itself, at 1/2/4/8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zksnake_amd/csrc/field.cuh"

using namespace zkmi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8];
    uint32_t x = a + threadIdx.x, y = b + threadIdx.x * 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i];
        x ^= (uint32_t)acc[0];
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mullo(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] * (b + i);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mulhi(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __umulhi(acc[i], b + i) + 0x9e3779b9u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad24(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __umul24(acc[i], b + i) + acc[i];
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add32(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (acc[i] + b) ^ (uint32_t)it;
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = ((uint64_t)a << 32) + i + threadIdx.x;
    uint64_t inc = ((uint64_t)b << 31) | 0xFFFFFFF1u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] + inc + (uint64_t)i;
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_dfma(uint64_t* out, uint32_t a, uint32_t b) {
    double acc[8];
    double x = 1.0 + 1e-9 * a, y = 1e-12 * b;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], x, y);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

template <class P>
__global__ void k_fpmul(uint64_t* out, uint32_t a, uint32_t b) {
    Fp<P> x = fp_one<P>(), y = fp_one<P>();
    x.v[0] += a + threadIdx.x;
    y.v[0] += b + 7 * threadIdx.x;
    for (int it = 0; it < ITERS / 16; ++it) {
        x = fp_mul<P>(x, y);
        y = fp_mul<P>(y, x);
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) s += x.v[i] ^ y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// V2: products as separate lo/hi words, accumulated with explicit 32-bit carry chains
template <class P>
__device__ __forceinline__ Fp<P> mul_v2(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t lo[N], hi[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { uint64_t p = (uint64_t)a.v[j] * b.v[i]; lo[j] = (uint32_t)p; hi[j] = (uint32_t)(p >> 32); }
        uint32_t c = 0, co;
#pragma unroll
        for (int j = 0; j < N; ++j) { t[j] = __builtin_addc(t[j], lo[j], c, &co); c = co; }
        t[N] = __builtin_addc(t[N], 0u, c, &co); c = co;
        t[N + 1] += c;
        c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) { t[j + 1] = __builtin_addc(t[j + 1], hi[j], c, &co); c = co; }
        t[N + 1] += c;
        uint32_t m = t[0] * P::INV;
#pragma unroll
        for (int j = 0; j < N; ++j) { uint64_t p = (uint64_t)m * P::MOD[j]; lo[j] = (uint32_t)p; hi[j] = (uint32_t)(p >> 32); }
        c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) { t[j] = __builtin_addc(t[j], lo[j], c, &co); c = co; }
        t[N] = __builtin_addc(t[N], 0u, c, &co); c = co;
        t[N + 1] += c;
        c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) { t[j + 1] = __builtin_addc(t[j + 1], hi[j], c, &co); c = co; }
        t[N + 1] += c;
#pragma unroll
        for (int j = 0; j <= N; ++j) t[j] = t[j + 1];
        t[N + 1] = 0;
    }
    Fp<P> out;
#pragma unroll
    for (int i = 0; i < N; ++i) out.v[i] = t[i];
    fp_reduce_once<P>(out);
    return out;
}


template <class P>
__global__ void k_fpmul2(uint64_t* out, uint32_t a, uint32_t b) {
    Fp<P> x = fp_one<P>(), y = fp_one<P>();
    x.v[0] += a + threadIdx.x;
    y.v[0] += b + 7 * threadIdx.x;
    for (int it = 0; it < ITERS / 16; ++it) {
        x = mul_v2<P>(x, y);
        y = mul_v2<P>(y, x);
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) s += x.v[i] ^ y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class P>
__global__ void k_fpadd(uint64_t* out, uint32_t a, uint32_t b) {
    Fp<P> x = fp_one<P>(), y = fp_one<P>();
    x.v[0] += a + threadIdx.x;
    y.v[0] += b + 7 * threadIdx.x;
    for (int it = 0; it < ITERS / 16; ++it) {
        x = fp_add<P>(x, y);
        y = fp_sub<P>(y, x);
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) s += x.v[i] ^ y.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kernel, double ops_per_thread, uint64_t* d_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256-thread blocks: 4 waves -> one per SIMD; wps blocks per CU
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 6789u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r, 6789u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double t = ms * 1e-3 / reps;
        double total_ops = ops_per_thread * blocks * 256.0;
        double waves = blocks * 4.0;
        // cycles per wave-instruction per SIMD at 2.4 GHz, if all 1024 SIMDs are busy
        double cyc = t * 2.4e9 / (ops_per_thread * waves / 1024.0);
        printf("%-12s waves/SIMD=%d  %8.3f ms  %9.2f Gop/s  ~%6.2f cyc/wave-op/SIMD\n", name, wps, t * 1e3, total_ops / t * 1e-9, cyc);
    }
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs=%d, clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run("mad_u64_u32", k_mad64, 8.0 * ITERS, d_out);
    run("mul_lo_u32", k_mullo, 8.0 * ITERS, d_out);
    run("mul_hi_u32", k_mulhi, 8.0 * ITERS, d_out);
    run("mad_u32_u24", k_mad24, 8.0 * ITERS, d_out);
    run("add_u32(+xor)", k_add32, 8.0 * ITERS, d_out);
    run("add_u64", k_add64, 8.0 * ITERS, d_out);
    run("fma_f64", k_dfma, 8.0 * ITERS, d_out);
    run("fpmul BnFq", k_fpmul<BnFqParams>, 2.0 * (ITERS / 16), d_out);
    run("fpmul2 BnFq", k_fpmul2<BnFqParams>, 2.0 * (ITERS / 16), d_out);
    run("fpmul2 BlsFq", k_fpmul2<BlsFqParams>, 2.0 * (ITERS / 16), d_out);
    run("fpmul BlsFq", k_fpmul<BlsFqParams>, 2.0 * (ITERS / 16), d_out);
    run("fpadd BnFq", k_fpadd<BnFqParams>, 2.0 * (ITERS / 16), d_out);
    return 0;
}
