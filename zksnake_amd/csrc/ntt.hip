// ntt.hip -- scalar-field radix-2 NTT / iNTT, element-wise vector ops and the fused QAP
// quotient pipeline for BN254 Fr and BLS12-381 Fr on gfx950.
//
// Stands in for fft/ifft/coset_fft/coset_ifft, add/mul_over_evaluation_domain and
// Polynomial.divide_by_vanishing_poly of the reference (src/bn254/polynomial.rs:466-489,
// 535-634 and the bls12_381 twin), which call ark-poly 0.4.2's Radix2EvaluationDomain.
//
// Layout in HBM: a vector is 2^k elements x 8 u32 (32 B), canonical integers, natural order.
// The transform is linear, so it runs directly on canonical data with twiddles kept in
// Montgomery form: mont_mul(x, w*R) = x*w.  No conversion pass is needed on either side.
//
// Kernel structure: decimation-in-frequency, several butterfly stages per launch on an LDS
// tile (structure-of-arrays: limb-major, so consecutive lanes hit consecutive banks), then one
// bit-reversal pass that also applies 1/N for the inverse transform.  Twiddles w^k (k < N/2)
// are precomputed once per (field, size, direction) and streamed from HBM.
#include <mutex>
#include <map>
#include <vector>
#include "common.cuh"
#include "fr_mem.cuh"

namespace zkmi {

constexpr int NTT_TILE_LOG = 10;  // 1024 elements = 32 KiB of LDS per workgroup
constexpr int NTT_THREADS = 256;
constexpr int NTT_Q = 3;          // 2^3 contiguous elements (256 B) per row of a strided tile

// ---- twiddle generation -------------------------------------------------------------------

template <class P>
__global__ void twiddle_kernel(uint32_t* out, Fp<P> w, uint32_t count) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t e[1] = {k};
    Fp<P> r = fp_pow<P>(w, e, 1);
    store_fr<P>(out + (size_t)k * P::W, r);
}

// ---- butterfly passes -----------------------------------------------------------------------
// One launch covers stages s_hi .. s_hi-m+1 of a DIF transform of size 2^log_n.  A workgroup owns
// the 2^(m+q) elements  i = hi << (s_hi+1) | mid << s_lo | lo_blk << q | lo_in  (mid: m bits,
// lo_in: q bits) and keeps them in LDS for all m stages.

// `in` and `out` may be the same vector (every workgroup reads and writes the same index set) unless `final_pass` is
// set: the last pass stores element i at the bit-reversed index (natural-order result), multiplied by `scale` when
// use_scale != 0 (1/N of the inverse transform) and fully reduced -- so it must write to a different vector.
template <class P>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const uint32_t* in, uint32_t* out,
                                                               const uint32_t* __restrict__ tw, int log_n,
                                                               int s_hi, int m, int q, int final_pass, Fp<P> scale, int use_scale) {
    constexpr int N = P::N;  // register limbs (LDS is limb-major)
    constexpr int W = P::W;  // words per element in HBM
    __shared__ uint32_t lds[N][1 << NTT_TILE_LOG];
    const int tile_log = m + q;
    const int tile = 1 << tile_log;
    const int s_lo = s_hi - m + 1;
    const uint32_t lo_blocks_log = s_lo - q;
    const uint32_t t = blockIdx.x;
    const uint32_t lo_blk = t & ((1u << lo_blocks_log) - 1);
    const uint32_t hi = t >> lo_blocks_log;
    const uint32_t base = (hi << (s_hi + 1)) | (lo_blk << q);

    for (int e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t mid = e >> q, lo_in = e & ((1u << q) - 1);
        uint32_t idx = base | (mid << s_lo) | lo_in;
        Fp<P> x = load_fr<P>(in + (size_t)idx * W);
#pragma unroll
        for (int l = 0; l < N; ++l) lds[l][e] = x.v[l];
    }
    __syncthreads();

    for (int b = m - 1; b >= 0; --b) {
        const int s = s_lo + b;
        const int pos = b + q;
        for (int u = threadIdx.x; u < (tile >> 1); u += NTT_THREADS) {
            uint32_t e0 = ((u >> pos) << (pos + 1)) | (u & ((1u << pos) - 1));
            uint32_t e1 = e0 | (1u << pos);
            uint32_t mid0 = e0 >> q, lo_in = e0 & ((1u << q) - 1);
            uint32_t i0 = base | (mid0 << s_lo) | lo_in;
            uint32_t k = (i0 & ((1u << s) - 1)) << (log_n - 1 - s);
            Fp<P> x, y;
#pragma unroll
            for (int l = 0; l < N; ++l) { x.v[l] = lds[l][e0]; y.v[l] = lds[l][e1]; }
            Fp<P> w = load_fr<P>(tw + (size_t)k * W);
            Fp<P> sum = fp_add<P>(x, y);
            Fp<P> dif = fp_mul<P>(w, fp_sub_lazy<P>(x, y));  // (x - y + 2p) un-normalized: fine as a product operand
#pragma unroll
            for (int l = 0; l < N; ++l) { lds[l][e0] = sum.v[l]; lds[l][e1] = dif.v[l]; }
        }
        __syncthreads();
    }

    for (int e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t mid = e >> q, lo_in = e & ((1u << q) - 1);
        uint32_t idx = base | (mid << s_lo) | lo_in;
        Fp<P> x;
#pragma unroll
        for (int l = 0; l < N; ++l) x.v[l] = lds[l][e];
        if (final_pass) {
            // the passes keep values semi-reduced (< 2p): the result is stored canonical, in natural order
            if (use_scale) x = fp_mul<P>(x, scale);
            idx = __brev(idx) >> (32 - log_n);
            x = fp_reduce_full<P>(x);
        }
        store_fr<P>(out + (size_t)idx * W, x);
    }
}

// x[i] *= g^(+-i) with g = w (table holds w^k for k < n/2; w^(n/2) = -1)
template <class P>
__global__ void coset_scale_kernel(uint32_t* data, const uint32_t* tw, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n == 1) return;
    uint32_t half = n >> 1;
    Fp<P> w = load_fr<P>(tw + (size_t)(i & (half - 1)) * P::W);
    if (i >= half) w = fp_neg<P>(w);
    Fp<P> a = load_fr<P>(data + (size_t)i * P::W);
    store_fr<P>(data + (size_t)i * P::W, fp_reduce_full<P>(fp_mul<P>(a, w)));
}

// op: 0 mul, 1 add, 2 sub on canonical data.  mul: mont(mont(a,b), R^2) = a*b
template <class P>
__global__ void vec_op_kernel(int op, uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> x = load_fr<P>(a + i * P::W), y = load_fr<P>(b + i * P::W), z;
    if (op == 0) z = fp_mul<P>(fp_mul<P>(x, y), fp_const<P>(P::R2));
    else if (op == 1) z = fp_add<P>(x, y);
    else z = fp_sub<P>(x, y);
    store_fr<P>(out + i * P::W, fp_reduce_full<P>(z));
}

// reduce every element below the modulus (inputs of the host API may be >= r)
template <class P>
__global__ void canon_kernel(uint64_t n, uint32_t* a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[P::W];
    uint4* q = reinterpret_cast<uint4*>(a + i * P::W);
#pragma unroll
    for (int k = 0; k < P::W / 4; ++k) {
        uint4 t = q[k];
        x[4 * k] = t.x; x[4 * k + 1] = t.y; x[4 * k + 2] = t.z; x[4 * k + 3] = t.w;
    }
    for (int k = 0; k < 10; ++k) {
        uint32_t t[P::W];
        if (fp_sub_mod_raw<P>(t, x)) break;
#pragma unroll
        for (int l = 0; l < P::W; ++l) x[l] = t[l];
    }
#pragma unroll
    for (int k = 0; k < P::W / 4; ++k) q[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
}

// q[j] = sum_{k>=1} c[j + k n],  rem[i] = c[i] + q[i]   (c has len coefficients)
template <class P>
__global__ void div_vanishing_kernel(uint64_t n, uint64_t len, const uint32_t* c, uint32_t* q, uint32_t* rem,
                                     int* nonzero) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t top = len < n ? len : n;
    uint64_t qlen = len > n ? len - n : 0;
    uint64_t lim = qlen > top ? qlen : top;
    if (i >= lim) return;
    // q[i] for i < qlen (may exceed n when len > 2n: then q[i] also folds further blocks)
    Fp<P> acc = fp_zero<P>();
    if (i < qlen) {
        for (uint64_t j = i + n; j < len; j += n) acc = fp_add<P>(acc, load_fr<P>(c + j * P::W));
        store_fr<P>(q + i * P::W, fp_reduce_full<P>(acc));
    }
    if (i < top) {
        Fp<P> r = fp_reduce_full<P>(fp_add<P>(load_fr<P>(c + i * P::W), acc));
        store_fr<P>(rem + i * P::W, r);
        if (!fp_is_zero<P>(r)) atomicOr(nonzero, 1);
    }
}

// CSR sparse matrix-vector product over Fr: out[row] = sum vals[k] * w[cols[k]].
// Canonical in/out: mont(mont(v, x), R^2) = v*x; the sum is accumulated in Montgomery-by-R^-1 form
// and fixed up once per row.
// Short rows (the bulk of an R1CS) take one lane each.  A row longer than `skip_longer_than` is left to the long-row
// kernels below: the constant-one wire and the input wires of a real circuit (and every row of the transposed matrices
// that Groth16.setup multiplies) can hold 2^20 entries, and one lane walking them took 0.8 s.
template <class P>
__global__ void spmv_kernel(uint64_t n_rows, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ cols,
                            const uint32_t* __restrict__ vals, const uint32_t* __restrict__ w, uint32_t* __restrict__ out,
                            uint32_t skip_longer_than) {
    uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const uint32_t k0 = row_ptr[row], k1 = row_ptr[row + 1];
    if (skip_longer_than && k1 - k0 > skip_longer_than) return;
    Fp<P> acc = fp_zero<P>();
    for (uint32_t k = k0; k < k1; ++k) {
        Fp<P> v = load_fr<P>(vals + (size_t)k * P::W);
        Fp<P> x = load_fr<P>(w + (size_t)cols[k] * P::W);
        acc = fp_add<P>(acc, fp_mul<P>(v, x));  // v*x/R
    }
    store_fr<P>(out + row * P::W, fp_reduce_full<P>(fp_mul<P>(acc, fp_const<P>(P::R2))));
}

// Long rows, two launches and no cross-workgroup hand-off: the host cuts every long row into work items [k0, k1) of at
// most a few thousand entries (static per matrix); one workgroup per item leaves its partial sum (still v*x/R form,
// semi-reduced) in `partials`; one workgroup per long row then adds up its items and writes the canonical result.
constexpr int SPMV_LONG_THREADS = 256;

template <class P>
__device__ __forceinline__ Fp<P> block_sum_fr(Fp<P> acc, uint32_t* sh) {
    constexpr int N = P::N;
    const uint32_t tid = threadIdx.x;
    for (uint32_t off = SPMV_LONG_THREADS / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int l = 0; l < N; ++l) sh[l * SPMV_LONG_THREADS + tid] = acc.v[l];
        __syncthreads();
        if (tid < off) {
            Fp<P> o;
#pragma unroll
            for (int l = 0; l < N; ++l) o.v[l] = sh[l * SPMV_LONG_THREADS + tid + off];
            acc = fp_add<P>(acc, o);
        }
        __syncthreads();
    }
    return acc;
}

template <class P>
__global__ __launch_bounds__(SPMV_LONG_THREADS) void spmv_items_kernel(const uint32_t* __restrict__ items /* (k0, k1) pairs */,
                                                                       const uint32_t* __restrict__ cols, const uint32_t* __restrict__ vals,
                                                                       const uint32_t* __restrict__ w, uint32_t* __restrict__ partials) {
    __shared__ uint32_t sh[P::N * SPMV_LONG_THREADS];
    const uint32_t k0 = items[2 * blockIdx.x], k1 = items[2 * blockIdx.x + 1];
    Fp<P> acc = fp_zero<P>();
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += SPMV_LONG_THREADS) {
        Fp<P> v = load_fr<P>(vals + (size_t)k * P::W);
        Fp<P> x = load_fr<P>(w + (size_t)cols[k] * P::W);
        acc = fp_add<P>(acc, fp_mul<P>(v, x));
    }
    acc = block_sum_fr<P>(acc, sh);
    if (threadIdx.x == 0) store_fr<P>(partials + (size_t)blockIdx.x * P::W, acc);
}

template <class P>
__global__ __launch_bounds__(SPMV_LONG_THREADS) void spmv_rows_kernel(const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ item_ptr,
                                                                      const uint32_t* __restrict__ partials, uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[P::N * SPMV_LONG_THREADS];
    const uint32_t i0 = item_ptr[blockIdx.x], i1 = item_ptr[blockIdx.x + 1];
    Fp<P> acc = fp_zero<P>();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += SPMV_LONG_THREADS) acc = fp_add<P>(acc, load_fr<P>(partials + (size_t)i * P::W));
    acc = block_sum_fr<P>(acc, sh);
    if (threadIdx.x == 0) store_fr<P>(out + (size_t)long_rows[blockIdx.x] * P::W, fp_reduce_full<P>(fp_mul<P>(acc, fp_const<P>(P::R2))));
}

// QAP tail: flag |= (lo[i] + hi[i] - w[i] != 0)
template <class P>
__global__ void qap_check_kernel(uint64_t n, const uint32_t* lo, const uint32_t* hi, const uint32_t* w, int* nonzero) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> r = fp_sub<P>(fp_add<P>(load_fr<P>(lo + i * P::W), load_fr<P>(hi + i * P::W)), load_fr<P>(w + i * P::W));
    if (!fp_is_zero<P>(r)) atomicOr(nonzero, 1);
}

// ---- host side ------------------------------------------------------------------------------

struct TwiddleSet {
    uint32_t* fwd = nullptr;
    uint32_t* inv = nullptr;
};

static std::mutex g_tw_mutex;
static std::map<std::pair<int, int>, TwiddleSet> g_twiddles;  // (curve, log_n)

template <class P>
static Fp<P> host_root(int log_n, bool inverse) {
    Fp<P> w = fp_const<P>(inverse ? P::ROOT_INV : P::ROOT);
    for (int k = 0; k < P::TWO_ADICITY - log_n; ++k) w = fp_sqr<P>(w);
    return w;
}

template <class P>
static int get_twiddles(int curve, int log_n, TwiddleSet* out, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto key = std::make_pair(curve, log_n);
    auto it = g_twiddles.find(key);
    if (it != g_twiddles.end()) { *out = it->second; return ZK_OK; }
    TwiddleSet ts;
    uint32_t count = log_n == 0 ? 1 : (1u << (log_n - 1));
    size_t bytes = (size_t)count * P::W * 4;
    ZK_HIP(hipMalloc(&ts.fwd, bytes));
    ZK_HIP(hipMalloc(&ts.inv, bytes));
    hipLaunchKernelGGL(twiddle_kernel<P>, dim3((count + 255) / 256), dim3(256), 0, stream, ts.fwd, host_root<P>(log_n, false), count);
    hipLaunchKernelGGL(twiddle_kernel<P>, dim3((count + 255) / 256), dim3(256), 0, stream, ts.inv, host_root<P>(log_n, true), count);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(stream));
    g_twiddles[key] = ts;
    *out = ts;
    return ZK_OK;
}

static void free_scratch();
static void free_twiddles() {
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    for (auto& kv : g_twiddles) {
        (void)hipFree(kv.second.fwd);
        (void)hipFree(kv.second.inv);
    }
    g_twiddles.clear();
}

// Scratch vector of the transform, one per stream (work on a stream is ordered, so reuse is safe); grow-only.
struct NttScratch {
    uint32_t* ptr = nullptr;
    size_t bytes = 0;
};
static std::mutex g_scratch_mutex;
static std::map<hipStream_t, NttScratch> g_scratch;

static int get_scratch(hipStream_t stream, size_t bytes, uint32_t** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    NttScratch& sc = g_scratch[stream];
    if (sc.bytes < bytes) {
        if (sc.ptr) {
            ZK_HIP(hipStreamSynchronize(stream));
            (void)hipFree(sc.ptr);
            sc.ptr = nullptr;
            sc.bytes = 0;
        }
        ZK_HIP(hipMalloc(&sc.ptr, bytes));
        sc.bytes = bytes;
    }
    *out = sc.ptr;
    return ZK_OK;
}

static void free_scratch() {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    for (auto& kv : g_scratch) (void)hipFree(kv.second.ptr);
    g_scratch.clear();
}

// Pass plan: the low min(log_n, 10) stages form the last, contiguous pass (1024 consecutive elements per workgroup);
// the stages above are split evenly over ceil(R / 7) strided passes of m stages on 2^m rows x 2^(10-m) contiguous
// elements -- an even split keeps the contiguous runs as long as possible (2^22: 6 + 6 + 10 with 512-byte runs).
// Data flow: first pass d -> scratch, middle passes in place on scratch, last pass scratch -> d at bit-reversed
// indices (+ 1/N, + canonical reduction), so no separate reordering pass.  A transform without strided stages
// (log_n <= 10) first copies d to the scratch vector, because the permuting pass cannot run in place.
template <class P>
static int ntt_dev_impl(int curve, int inverse, int log_n, uint32_t* d, hipStream_t stream) {
    if (log_n < 0 || log_n > P::TWO_ADICITY || log_n > 30) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    if (log_n == 0) return ZK_OK;
    TwiddleSet ts;
    int rc = get_twiddles<P>(curve, log_n, &ts, stream);
    if (rc) return rc;
    const uint32_t* tw = inverse ? ts.inv : ts.fwd;
    const size_t bytes = ((size_t)1 << log_n) * P::W * 4;
    uint32_t* scratch = nullptr;
    if ((rc = get_scratch(stream, bytes, &scratch))) return rc;
    Fp<P> scale = fp_one<P>();
    if (inverse) {
        uint32_t nn[P::W] = {0};
        nn[0] = 1u << log_n;
        scale = fp_inv<P>(fp_from_canonical<P>(nn));
    }
    const int last = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;  // stages of the contiguous pass
    const int upper = log_n - last;                                 // stages of the strided passes
    const int max_m = NTT_TILE_LOG - NTT_Q;
    const int n_strided = (upper + max_m - 1) / max_m;
    const uint32_t* src = d;
    int s = log_n - 1;
    for (int i = 0; i < n_strided; ++i) {
        const int m = upper / n_strided + (i < upper % n_strided ? 1 : 0);
        const int s_lo = s - m + 1;
        const int q = s_lo < NTT_TILE_LOG - m ? s_lo : NTT_TILE_LOG - m;
        const uint32_t tiles = 1u << (log_n - m - q);
        hipLaunchKernelGGL(ntt_pass_kernel<P>, dim3(tiles), dim3(NTT_THREADS), 0, stream, src, scratch, tw, log_n, s, m, q, 0, scale, 0);
        src = scratch;
        s -= m;
    }
    if (n_strided == 0) {
        // no strided pass put the data into the scratch vector: the final pass may not run in place, so stage a copy
        ZK_HIP(hipMemcpyAsync(scratch, d, bytes, hipMemcpyDeviceToDevice, stream));
        src = scratch;
    }
    hipLaunchKernelGGL(ntt_pass_kernel<P>, dim3(1u << (log_n - last)), dim3(NTT_THREADS), 0, stream, src, d, tw, log_n, last - 1, last, 0, 1,
                       scale, inverse ? 1 : 0);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int coset_scale_impl(int curve, int inverse, int log_n, uint32_t* d, hipStream_t stream) {
    if (log_n == 0) return ZK_OK;
    TwiddleSet ts;
    int rc = get_twiddles<P>(curve, log_n, &ts, stream);
    if (rc) return rc;
    uint32_t n = 1u << log_n;
    hipLaunchKernelGGL(coset_scale_kernel<P>, dim3((n + 255) / 256), dim3(256), 0, stream, d, inverse ? ts.inv : ts.fwd, n);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int vec_op_dev_impl(int op, uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t stream) {
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(vec_op_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, n, a, b, out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int ntt_host_impl(int curve, int inverse, int coset, uint64_t n_in, const uint64_t* in, uint64_t size, uint64_t* out) {
    uint64_t n = next_pow2_u64(size == 0 ? 1 : size);
    int log_n = log2_u64(n);
    if (log_n > P::TWO_ADICITY || log_n > 30) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    const size_t eb = P::W * 4;
    uint32_t* d = nullptr;
    // An input longer than the domain is TRUNCATED to the domain size: the reference hands the raw slice to
    // EvaluationDomain::fft / ifft (polynomial.rs:541-542,567-568), whose radix-2 fft_in_place / ifft_in_place start with
    // `coeffs.resize(self.size(), zero)` [ark-poly 0.4.2, not vendored: parity unpinned, see DESIGN.md section 2]; only
    // DensePolynomial::evaluate_over_domain folds modulo X^n - 1, and the reference does not call it on this path.
    if (n_in > n) n_in = n;
    const uint64_t alloc = n;
    ZK_HIP(hipMalloc(&d, alloc * eb));
    int rc = ZK_OK;
    do {
        if (hipMemset(d, 0, alloc * eb) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemset failed"); break; }
        if (n_in && hipMemcpy(d, in, n_in * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((alloc + 255) / 256)), dim3(256), 0, 0, alloc, d);
        if (coset && !inverse) { rc = coset_scale_impl<P>(curve, 0, log_n, d, 0); if (rc) break; }
        rc = ntt_dev_impl<P>(curve, inverse, log_n, d, 0);
        if (rc) break;
        if (coset && inverse) { rc = coset_scale_impl<P>(curve, 1, log_n, d, 0); if (rc) break; }
        if (hipMemcpy(out, d, n * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    (void)hipFree(d);
    return rc;
}

template <class P>
static int vec_op_host_impl(int op, uint64_t size, uint64_t n_a, const uint64_t* a, uint64_t n_b, const uint64_t* b, uint64_t* out) {
    if (size == 0) return ZK_OK;
    const size_t eb = P::W * 4;
    uint32_t* d = nullptr;
    ZK_HIP(hipMalloc(&d, 2 * size * eb));
    int rc = ZK_OK;
    do {
        if (hipMemset(d, 0, 2 * size * eb) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemset failed"); break; }
        uint64_t ca = n_a < size ? n_a : size, cb = n_b < size ? n_b : size;
        if (ca && hipMemcpy(d, a, ca * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        if (cb && hipMemcpy(d + size * P::W, b, cb * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((2 * size + 255) / 256)), dim3(256), 0, 0, 2 * size, d);
        rc = vec_op_dev_impl<P>(op, size, d, d + size * P::W, d, 0);
        if (rc) break;
        if (hipMemcpy(out, d, size * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    (void)hipFree(d);
    return rc;
}

template <class P>
static int div_vanishing_host_impl(uint64_t n, uint64_t len, const uint64_t* coeffs, uint64_t* q, uint64_t* rem, int* rem_is_zero) {
    if (n == 0) return fail(ZK_ERR_ARG, "vanishing polynomial of an empty domain");
    *rem_is_zero = 1;
    if (len == 0) return ZK_OK;
    const size_t eb = P::W * 4;
    uint64_t qlen = len > n ? len - n : 0, top = len < n ? len : n;
    uint32_t *dc = nullptr, *dq = nullptr, *dr = nullptr;
    int* dflag = nullptr;
    ZK_HIP(hipMalloc(&dc, len * eb));
    int rc = ZK_OK;
    do {
        if (hipMalloc(&dq, (qlen ? qlen : 1) * eb) != hipSuccess || hipMalloc(&dr, top * eb) != hipSuccess ||
            hipMalloc(&dflag, sizeof(int)) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMalloc failed"); break; }
        if (hipMemcpy(dc, coeffs, len * eb, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        (void)hipMemset(dflag, 0, sizeof(int));
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, len, dc);
        uint64_t lim = qlen > top ? qlen : top;
        hipLaunchKernelGGL(div_vanishing_kernel<P>, dim3((unsigned)((lim + 255) / 256)), dim3(256), 0, 0, n, len, dc, dq, dr, dflag);
        int flag = 0;
        if (hipMemcpy(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        if (qlen && hipMemcpy(q, dq, qlen * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        if (hipMemcpy(rem, dr, top * eb, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
        *rem_is_zero = flag ? 0 : 1;
    } while (0);
    (void)hipFree(dc); (void)hipFree(dq); (void)hipFree(dr); (void)hipFree(dflag);
    return rc;
}

template <class P>
static int qap_h_dev_impl(int curve, int log_n, uint32_t* a_u, uint32_t* b_v, const uint32_t* c, uint32_t* h,
                          uint32_t* work, int* divisible, hipStream_t stream) {
    if (log_n + 1 > P::TWO_ADICITY) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    const uint64_t n = 1ull << log_n;
    const size_t eb = P::W * 4;
    uint32_t* U2 = work;
    uint32_t* V2 = work + 2 * n * P::W;
    int rc;
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, a_u, stream))) return rc;  // u
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, b_v, stream))) return rc;  // v
    ZK_HIP(hipMemsetAsync(work, 0, 4 * n * eb, stream));
    ZK_HIP(hipMemcpyAsync(U2, a_u, n * eb, hipMemcpyDeviceToDevice, stream));
    ZK_HIP(hipMemcpyAsync(V2, b_v, n * eb, hipMemcpyDeviceToDevice, stream));
    if ((rc = ntt_dev_impl<P>(curve, 0, log_n + 1, U2, stream))) return rc;
    if ((rc = ntt_dev_impl<P>(curve, 0, log_n + 1, V2, stream))) return rc;
    if ((rc = vec_op_dev_impl<P>(0, 2 * n, U2, V2, U2, stream))) return rc;
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n + 1, U2, stream))) return rc;  // uv coefficients
    ZK_HIP(hipMemcpyAsync(V2, c, n * eb, hipMemcpyDeviceToDevice, stream));
    if ((rc = ntt_dev_impl<P>(curve, 1, log_n, V2, stream))) return rc;      // w
    int* dflag = reinterpret_cast<int*>(V2 + n * P::W);  // second half of V2 is free now
    ZK_HIP(hipMemsetAsync(dflag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(qap_check_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, U2, U2 + n * P::W, V2, dflag);
    ZK_HIP(hipMemcpyAsync(h, U2 + n * P::W, n * eb, hipMemcpyDeviceToDevice, stream));
    int flag = 0;
    ZK_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, stream));
    ZK_HIP(hipStreamSynchronize(stream));
    *divisible = flag ? 0 : 1;
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_ntt(int curve, int inverse, int coset, uint64_t n_in, const uint64_t* in, uint64_t size, uint64_t* out) {
#define CALL(P) return ntt_host_impl<P>(curve, inverse, coset, n_in, in, size, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_op(int curve, int op, uint64_t size, uint64_t n_a, const uint64_t* a, uint64_t n_b, const uint64_t* b, uint64_t* out) {
    if (op < 0 || op > 2) return fail(ZK_ERR_ARG, "unknown vector op");
#define CALL(P) return vec_op_host_impl<P>(op, size, n_a, a, n_b, b, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_poly_div_vanishing(int curve, uint64_t n, uint64_t len, const uint64_t* coeffs, uint64_t* q, uint64_t* rem, int* rem_is_zero) {
#define CALL(P) return div_vanishing_host_impl<P>(n, len, coeffs, q, rem, rem_is_zero)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_ntt_dev(int curve, int inverse, int log_n, void* d_data, void* stream) {
#define CALL(P) return ntt_dev_impl<P>(curve, inverse, log_n, (uint32_t*)d_data, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_op_dev(int curve, int op, uint64_t n, const void* d_a, const void* d_b, void* d_out, void* stream) {
    if (op < 0 || op > 2) return fail(ZK_ERR_ARG, "unknown vector op");
#define CALL(P) return vec_op_dev_impl<P>(op, n, (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t*)d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_qap_h_dev(int curve, int log_n, void* d_a_u, void* d_b_v, const void* d_c, void* d_h, void* d_work, int* divisible, void* stream) {
#define CALL(P) return qap_h_dev_impl<P>(curve, log_n, (uint32_t*)d_a_u, (uint32_t*)d_b_v, (const uint32_t*)d_c, (uint32_t*)d_h, (uint32_t*)d_work, divisible, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_spmv_dev(int curve, uint64_t n_rows, const void* d_row_ptr, const void* d_cols, const void* d_vals,
                const void* d_w, void* d_out, uint32_t skip_longer_than, void* stream) {
    if (n_rows == 0) return ZK_OK;
#define CALL(P)                                                                                                  \
    {                                                                                                            \
        hipLaunchKernelGGL(spmv_kernel<P>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, \
                           n_rows, (const uint32_t*)d_row_ptr, (const uint32_t*)d_cols, (const uint32_t*)d_vals,  \
                           (const uint32_t*)d_w, (uint32_t*)d_out, skip_longer_than);                             \
        ZK_HIP(hipGetLastError());                                                                               \
        return ZK_OK;                                                                                            \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_spmv_long_dev(int curve, uint64_t n_long, const void* d_long_rows, const void* d_item_ptr, uint64_t n_items,
                     const void* d_items, const void* d_cols, const void* d_vals, const void* d_w, void* d_partials,
                     void* d_out, void* stream) {
    if (n_long == 0 || n_items == 0) return ZK_OK;
#define CALL(P)                                                                                                          \
    {                                                                                                                    \
        hipLaunchKernelGGL(spmv_items_kernel<P>, dim3((unsigned)n_items), dim3(SPMV_LONG_THREADS), 0, (hipStream_t)stream, \
                           (const uint32_t*)d_items, (const uint32_t*)d_cols, (const uint32_t*)d_vals,                   \
                           (const uint32_t*)d_w, (uint32_t*)d_partials);                                                 \
        hipLaunchKernelGGL(spmv_rows_kernel<P>, dim3((unsigned)n_long), dim3(SPMV_LONG_THREADS), 0, (hipStream_t)stream,   \
                           (const uint32_t*)d_long_rows, (const uint32_t*)d_item_ptr, (const uint32_t*)d_partials,       \
                           (uint32_t*)d_out);                                                                            \
        ZK_HIP(hipGetLastError());                                                                                       \
        return ZK_OK;                                                                                                    \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_canon_dev(int curve, uint64_t n, void* d_x, void* stream) {
    if (n == 0) return ZK_OK;
#define CALL(P)                                                                                                       \
    {                                                                                                                 \
        hipLaunchKernelGGL(canon_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,  \
                           (uint32_t*)d_x);                                                                           \
        ZK_HIP(hipGetLastError());                                                                                    \
        return ZK_OK;                                                                                                 \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

void zk_ntt_free_cache(void) {
    free_twiddles();
    free_scratch();
}

}  // extern "C"
