// tools/ubench.hip -- VALU issue-rate microbenchmark for gfx950 (design input for csrc/field.hip.h).
// Sustained per-SIMD cost of the instructions a big-integer Montgomery product can be built from, and of the
// library's own field operations, at 1/2/4/8 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/ubench tools/ubench.hip && ./build/ubench
// The first-round run that motivated the 29-bit limb design (32-bit-limb CIOS: ~1900 cycles per BN254 product)
// is kept in profiles/r01_ubench_valu.log.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../zksnake_amd/csrc/field.hip.h"

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

__global__ void k_addxor(uint64_t* out, uint32_t a, uint32_t b) {
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

// OP: 0 mul, 1 sqr, 2 add+sub, 3 mul2
template <class P, int OP>
__global__ void k_field(uint64_t* out, uint32_t a, uint32_t b) {
    Fp<P> x = fp_one<P>(), y = fp_const<P>(P::R2);
    x.v[0] = (x.v[0] + a + threadIdx.x) & LIMB_MASK;
    y.v[0] = (y.v[0] + b + 7 * threadIdx.x) & LIMB_MASK;
    for (int it = 0; it < ITERS / 16; ++it) {
        if (OP == 0) { x = fp_mul<P>(x, y); y = fp_mul<P>(y, x); }
        if (OP == 1) { x = fp_sqr<P>(y); y = fp_sqr<P>(x); }
        if (OP == 2) { x = fp_add<P>(x, y); y = fp_sub<P>(y, x); }
        if (OP == 3) { x = fp_mul2<P>(x, y, y, x); y = fp_mul2<P>(y, x, x, x); }
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
        double cyc = t * 2.4e9 / (ops_per_thread * waves / 1024.0);  // nominal 2.4 GHz, all 1024 SIMDs busy
        printf("%-16s waves/SIMD=%d  %8.3f ms  %9.2f Gop/s  ~%7.1f cyc/wave-op/SIMD\n", name, wps, t * 1e3, total_ops / t * 1e-9, cyc);
    }
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    run("mad_u64_u32", k_mad64, 8.0 * ITERS, d_out);
    run("mul_hi_u32+add", k_mulhi, 8.0 * ITERS, d_out);
    run("add+xor (2 ops)", k_addxor, 8.0 * ITERS, d_out);
    run("fma_f64", k_dfma, 8.0 * ITERS, d_out);
    const double f = 2.0 * (ITERS / 16);
    run("fp_mul BnFq", k_field<BnFqParams, 0>, f, d_out);
    run("fp_sqr BnFq", k_field<BnFqParams, 1>, f, d_out);
    run("fp_add|sub BnFq", k_field<BnFqParams, 2>, f, d_out);
    run("fp_mul2 BnFq", k_field<BnFqParams, 3>, f, d_out);
    run("fp_mul BlsFq", k_field<BlsFqParams, 0>, f, d_out);
    run("fp_sqr BlsFq", k_field<BlsFqParams, 1>, f, d_out);
    run("fp_add|sub BlsFq", k_field<BlsFqParams, 2>, f, d_out);
    run("fp_mul2 BlsFq", k_field<BlsFqParams, 3>, f, d_out);
    return 0;
}
