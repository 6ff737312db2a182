// tools/affine_probe.hip -- feasibility probe (not part of the library): cost per point addition of
//   (A) the XYZZ mixed addition of the accumulate kernel (one serial chain per lane, bases gathered from a table), against
//   (B) affine additions with one shared inversion per workgroup-batch (Montgomery's trick: per-lane prefix products
//       spilled to a scratch buffer, prefix / suffix scans across the 256 lanes through LDS, one Fermat inversion),
// on G2 of BN254 (Fp2 coordinates).  The operands are random field elements, not curve points: the arithmetic does not
// care, and no special case (doubling, infinity) can occur.  hipcc -O3 --offload-arch=gfx950 -o build/affine_probe tools/affine_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zksnake_amd/csrc/curve.hip.h"
#include "../zksnake_amd/csrc/curve_consts.h"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef Fp2Ops<BnFqParams> F;
typedef F::T T;
constexpr int AW = 2 * F::LIMBS;   // words per affine row (32)
constexpr int FW = F::LIMBS;       // words per Fp2 (16)

__device__ __forceinline__ T load_f(const uint32_t* p) {
    uint32_t w[FW];
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < FW / 4; ++i) { uint4 t = q[i]; w[4*i] = t.x; w[4*i+1] = t.y; w[4*i+2] = t.z; w[4*i+3] = t.w; }
    return F::load(w);
}
__device__ __forceinline__ void store_f(uint32_t* p, const T& a) {
    uint32_t w[FW];
    F::store(w, a);
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < FW / 4; ++i) q[i] = make_uint4(w[4*i], w[4*i+1], w[4*i+2], w[4*i+3]);
}

// (A) serial XYZZ accumulation, k entries per lane
__global__ __launch_bounds__(256) void xyzz_kernel(const uint32_t* table, const uint32_t* idx, int k, uint32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> acc = xyzz_inf<F>();
    for (int j = 0; j < k; ++j) {
        const uint32_t ref = idx[(size_t)t * k + j];
        xyzz_add_affine_mem<F>(acc, table + (size_t)(ref & 0x7FFFFFFFu) * AW, (ref >> 31) != 0);
    }
    store_f(out + (size_t)t * FW, F::add(F::add(acc.X, acc.Y), F::add(acc.ZZ, acc.ZZZ)));
}

// (B) k independent affine additions per lane, one inversion per workgroup
constexpr int WG = 256;
__device__ __forceinline__ void lds_put(uint32_t* sh, uint32_t lane, const T& a) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) sh[i * WG + lane] = s[i];
}
__device__ __forceinline__ T lds_get(const uint32_t* sh, uint32_t lane) {
    T r;
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = sh[i * WG + lane];
    return r;
}
__global__ __launch_bounds__(WG) void affine_kernel(const uint32_t* table, const uint32_t* idx, int k, uint32_t* scratch, uint32_t* out) {
    __shared__ uint32_t shp[sizeof(T) / 4 * WG], shs[sizeof(T) / 4 * WG];
    const uint32_t lane = threadIdx.x;
    const uint32_t t = blockIdx.x * WG + lane;
    const uint32_t* my = idx + (size_t)t * 2 * k;
    uint32_t* spill = scratch + (size_t)blockIdx.x * k * WG * FW;
    // pass 1: running product of the denominators, spilled before each step
    T run = F::one();
    for (int j = 0; j < k; ++j) {
        const T x1 = load_f(table + (size_t)(my[2 * j] & 0x7FFFFFFFu) * AW);
        const T x2 = load_f(table + (size_t)(my[2 * j + 1] & 0x7FFFFFFFu) * AW);
        store_f(spill + ((size_t)j * WG + lane) * FW, run);      // product of the denominators before pair j
        run = F::mul(run, F::sub(x2, x1));
    }
    // inverse of every lane's total: inclusive prefix and suffix products over the 256 lanes, one inversion
    T pre = run, suf = run;
    for (int off = 1; off < WG; off <<= 1) {
        lds_put(shp, lane, pre);
        lds_put(shs, lane, suf);
        __syncthreads();
        if (lane >= (uint32_t)off) pre = F::mul(pre, lds_get(shp, lane - off));
        if (lane + off < WG) suf = F::mul(suf, lds_get(shs, lane + off));
        __syncthreads();
    }
    lds_put(shp, lane, pre);
    lds_put(shs, lane, suf);
    __syncthreads();
    if (lane == 0) {
        T inv = F::inv(lds_get(shp, WG - 1));
        lds_put(shp, WG - 1, inv);   // slot WG-1 now holds the inverse of the grand total (its prefix is no longer needed: only lanes' neighbours are read below, before this write? no -- see barrier)
    }
    // note: lane 0 overwrote shp[WG-1]; lanes read shp[lane-1] (lane-1 <= WG-2) and shs[lane+1]: disjoint from that slot
    __syncthreads();
    T inv = lds_get(shp, WG - 1);
    if (lane > 0) inv = F::mul(inv, lds_get(shp, lane - 1));
    if (lane + 1 < WG) inv = F::mul(inv, lds_get(shs, lane + 1));
    // pass 2: unwind
    T accx = F::zero(), accy = F::zero();
    for (int j = k - 1; j >= 0; --j) {
        const uint32_t r1 = my[2 * j], r2 = my[2 * j + 1];
        const uint32_t* p1 = table + (size_t)(r1 & 0x7FFFFFFFu) * AW;
        const uint32_t* p2 = table + (size_t)(r2 & 0x7FFFFFFFu) * AW;
        const T x1 = load_f(p1), y1 = load_f(p1 + FW), x2 = load_f(p2), y2 = load_f(p2 + FW);
        const T before = load_f(spill + ((size_t)j * WG + lane) * FW);
        const T d = F::sub(x2, x1);
        const T dinv = F::mul(inv, before);
        inv = F::mul(inv, d);
        const T lam = F::mul(F::sub(y2, y1), dinv);
        const T x3 = F::sub(F::sub(F::sqr(lam), x1), x2);
        const T y3 = F::sub(F::mul(lam, F::sub(x1, x3)), y1);
        accx = F::add(accx, x3);   // stands in for the 128-byte store of the result
        accy = F::add(accy, y3);
    }
    store_f(out + (size_t)t * FW, F::add(accx, accy));
}

int main(int argc, char** argv) {
    const int log_rows = argc > 1 ? atoi(argv[1]) : 23;
    const size_t rows = (size_t)1 << log_rows;
    uint32_t *table, *idx, *scratch, *out;
    CK(hipMalloc(&table, rows * AW * 4));
    std::vector<uint32_t> h(rows * AW);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16) & 0x0FFFFFFFu; }   // < p limb-wise-ish: any value below 2^252
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const size_t adds = 13ull << 20;   // 13.6 M additions, as one fixed-base G2 accumulate of a 2^20 proof
    std::vector<uint32_t> hi(2 * adds);
    for (auto& w : hi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)((s >> 20) % rows); }
    CK(hipMalloc(&idx, hi.size() * 4));
    CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, adds * FW * 4));
    CK(hipMalloc(&scratch, adds * FW * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int k : {26, 52, 64}) {
        const unsigned lanes = (unsigned)(adds / k);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(xyzz_kernel, dim3(lanes / 256), dim3(256), 0, 0, table, idx, k, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("XYZZ    k=%3d lanes=%7u  %7.3f ms  %6.1f ps/add\n", k, lanes, ms, ms * 1e9 / ((double)lanes / 256 * 256 * k));
    }
    for (int k : {8, 16, 32, 64, 128}) {
        const unsigned lanes = (unsigned)(adds / k) / WG * WG;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(affine_kernel, dim3(lanes / WG), dim3(WG), 0, 0, table, idx, k, scratch, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("affine  k=%3d lanes=%7u  %7.3f ms  %6.1f ps/add\n", k, lanes, ms, ms * 1e9 / ((double)lanes * k));
    }
    CK(hipGetLastError());
    return 0;
}
