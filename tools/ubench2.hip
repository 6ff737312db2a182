// tools/ubench2.hip -- issue cost of the NON-multiply instructions of the 29-bit-limb field product on gfx950
// (v_lshrrev_b64, v_lshl_add_u64, v_mul_lo_u32, v_alignbit_b32, v_and_b32, v_add3_u32), eight independent chains per lane,
// at 1 / 2 / 4 / 8 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 -o build/ubench2 tools/ubench2.hip && ./build/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096;

#define KERNEL64(name, ASM)                                                              \
    __global__ void name(uint64_t* out, uint32_t a, uint32_t b) {                        \
        uint64_t acc[8];                                                                 \
        for (int i = 0; i < 8; ++i) acc[i] = ((uint64_t)(a + i) << 32) | (threadIdx.x + b); \
        uint64_t k = ((uint64_t)b << 32) | a;                                            \
        for (int it = 0; it < ITERS; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(k), "v"(a)); \
        }                                                                                \
        uint64_t s = 0;                                                                  \
        for (int i = 0; i < 8; ++i) s ^= acc[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
    }
#define KERNEL32(name, ASM)                                                              \
    __global__ void name(uint64_t* out, uint32_t a, uint32_t b) {                        \
        uint32_t acc[8];                                                                 \
        for (int i = 0; i < 8; ++i) acc[i] = a + i + threadIdx.x;                        \
        for (int it = 0; it < ITERS; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(b), "v"(a)); \
        }                                                                                \
        uint32_t s = 0;                                                                  \
        for (int i = 0; i < 8; ++i) s ^= acc[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
    }
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %2, %2, %0")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshr32, "v_lshrrev_b32 %0, 3, %0")

template <class K>
static void run(const char* name, K kernel, uint64_t* d_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-16s", name);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 6789u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r, 6789u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double t = ms * 1e-3 / 5;
        double cyc = t * 2.4e9 / (8.0 * ITERS * wps);   // per wave-instruction per SIMD at nominal 2.4 GHz (one wave per SIMD per block)
        printf("  %dw: %6.2f cyc", wps, cyc);
    }
    printf("\n");
}
int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    run("v_mad_u64_u32", k_mad64, d_out);
    run("v_lshrrev_b64", k_lshr64, d_out);
    run("v_lshl_add_u64", k_lshladd64, d_out);
    run("v_mul_lo_u32", k_mullo, d_out);
    run("v_mul_hi_u32", k_mulhi, d_out);
    run("v_alignbit_b32", k_alignbit, d_out);
    run("v_and_b32", k_and, d_out);
    run("v_add3_u32", k_add3, d_out);
    run("v_lshrrev_b32", k_lshr32, d_out);
    return 0;
}
