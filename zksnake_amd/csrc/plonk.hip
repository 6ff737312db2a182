// plonk.hip -- device-resident vector kernels of the PlonK prover over BN254 Fr / BLS12-381 Fr (gfx950).
//
// Stands in for the element-wise polynomial arithmetic of Plonk.prove in the reference
// (python/zksnake/plonk/protocol.py:213-460: fft -> mul_over_evaluation_domain / add_over_evaluation_domain
// chains, Polynomial scalar multiply-adds, Polynomial.__call__), fused so that one proof makes one pass over
// each coset-evaluation vector.  SURVEY.md 8f row 2.
//
// All vectors are canonical Fr elements (32 B) in HBM; scalars cross the ABI as canonical 4-limb integers.
// Products use the Montgomery multiplier directly on canonical data: mont(x, s*R) = x*s, and chains of k
// products of canonical values are corrected once by a scalar carrying R^k (folded into alpha below), so no
// vector is ever converted to Montgomery form.
//
// Roofline: the quotient kernel reads 15 vectors and writes one (512 B per point) against 22 field products;
// at ~1000 cycles per product per wave it is integer-multiply bound like everything else on this path.
#include <vector>
#include "common.hip.h"
#include "fr_mem.hip.h"

namespace zkmi {

template <class P>
__host__ Fp<P> scalar_in(const uint64_t* s) {
    return fp_unpack<P>(reinterpret_cast<const uint32_t*>(s));  // canonical integer as limbs
}

// s * R^k as an integer (k >= 0): what mont() must be fed to undo k divisions by R
template <class P>
__host__ Fp<P> scalar_times_r(const uint64_t* s, int k) {
    Fp<P> v = scalar_in<P>(s);
    for (int i = 0; i < k; ++i) v = fp_mul<P>(v, fp_const<P>(P::R2));
    return fp_reduce_full<P>(v);
}

// out = a*x + b*y + c      (y may be null)
template <class P>
__global__ __launch_bounds__(256) void axpby_kernel(uint64_t n, Fp<P> a_m, const uint32_t* __restrict__ x, Fp<P> b_m,
                                                    const uint32_t* __restrict__ y, Fp<P> c, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> v = fp_add<P>(fp_mul<P>(load_fr<P>(x + i * P::W), a_m), c);
    if (y) v = fp_add<P>(v, fp_mul<P>(load_fr<P>(y + i * P::W), b_m));
    store_fr<P>(out + i * P::W, fp_reduce_full<P>(v));
}

// acc[i] += sum_k s_k x_k[i] (i < count_k), then acc[at_j] += c_j: a whole chain of scalar multiply-adds and single-coefficient
// updates in ONE launch (the linearisation polynomial of a PlonK proof is ten such terms, a blinding two to three updates; as
// separate launches of a few microseconds each they left the GPU idle between them: launch-bound)
constexpr int LINCOMB_MAX_TERMS = 16, LINCOMB_MAX_AT = 8;
template <class P>
struct LincombArgs {
    const uint32_t* x[LINCOMB_MAX_TERMS];
    uint64_t count[LINCOMB_MAX_TERMS];
    Fp<P> s_m[LINCOMB_MAX_TERMS];      // s R: mont(x, s R) = s x
    uint64_t at[LINCOMB_MAX_AT];
    Fp<P> at_val[LINCOMB_MAX_AT];      // canonical
    int k, n_at;
};
template <class P>
__global__ __launch_bounds__(256) void lincomb_kernel(uint64_t n, uint32_t* __restrict__ acc, LincombArgs<P> a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<P> v = load_fr<P>(acc + i * P::W);
    for (int t = 0; t < a.k; ++t)
        if (i < a.count[t]) v = fp_add<P>(v, fp_mul<P>(load_fr<P>(a.x[t] + i * P::W), a.s_m[t]));
    for (int j = 0; j < a.n_at; ++j)
        if (i == a.at[j]) v = fp_add<P>(v, a.at_val[j]);
    store_fr<P>(acc + i * P::W, fp_reduce_full<P>(fp_reduce_full<P>(v)));
}

// out[i] = in[offset + i * stride]: de-interleaves the flat PlonK witness [a0, b0, c0, a1, ..] into its three columns
template <class P>
__global__ __launch_bounds__(256) void gather_kernel(uint64_t n, const uint4* __restrict__ in, uint64_t stride, uint64_t offset, uint4* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int V = P::W / 4;  // 16-byte vectors per element
    if (i >= n * V) return;
    const uint64_t e = i / V, part = i % V;
    out[i] = in[(offset + e * stride) * V + part];
}

template <class P>
__global__ __launch_bounds__(256) void is_zero_kernel(uint64_t n_vec4, const uint4* __restrict__ x, int* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vec4) return;
    uint4 t = x[i];
    if (t.x | t.y | t.z | t.w) atomicOr(flag, 1);
}

// out[i] = prod_j (w_j[i] + beta * label_j[i] + gamma),  j = 0..2      (protocol.py:270-292 on the n-domain)
template <class P>
__global__ __launch_bounds__(256) void perm_terms_kernel(uint64_t n, const uint32_t* __restrict__ w0, const uint32_t* __restrict__ w1,
                                                         const uint32_t* __restrict__ w2, const uint32_t* __restrict__ l0,
                                                         const uint32_t* __restrict__ l1, const uint32_t* __restrict__ l2,
                                                         Fp<P> beta_m, Fp<P> gamma, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = i * P::W;
    Fp<P> u0 = fp_add<P>(fp_add<P>(load_fr<P>(w0 + o), fp_mul<P>(load_fr<P>(l0 + o), beta_m)), gamma);
    Fp<P> u1 = fp_add<P>(fp_add<P>(load_fr<P>(w1 + o), fp_mul<P>(load_fr<P>(l1 + o), beta_m)), gamma);
    Fp<P> u2 = fp_add<P>(fp_add<P>(load_fr<P>(w2 + o), fp_mul<P>(load_fr<P>(l2 + o), beta_m)), gamma);
    const Fp<P> r2 = fp_const<P>(P::R2);
    Fp<P> v = fp_mul<P>(fp_mul<P>(fp_mul<P>(u0, r2), u1), fp_mul<P>(u2, r2));  // (u0 R)(u1)/R = u0 u1; (u0 u1)(u2 R)/R
    store_fr<P>(out + o, fp_reduce_full<P>(v));
}

constexpr int QUOT_MAX_PERIOD = 16;

template <class P>
struct QuotientArgs {
    const uint32_t *a, *b, *c, *z, *pi, *ql, *qr, *qo, *qm, *qc, *s1, *s2, *s3, *x, *l1;
    Fp<P> beta_m;    // beta R
    Fp<P> gamma;     // gamma
    Fp<P> alpha_r4;  // alpha R^4: closes a chain of four products of canonical values
    Fp<P> alpha2_r2; // alpha^2 R^2
    Fp<P> zh_inv_m[QUOT_MAX_PERIOD];  // 1/(x^n - 1) R for the m/n distinct values on the coset
    uint32_t period;  // m / n
};

// t(x) = [ gate + alpha (prod(w_j + beta k_j x + gamma) z(x) - prod(w_j + beta sigma_j + gamma) z(omega x))
//          + alpha^2 (z(x) - 1) L1(x) ] / (x^n - 1)      on the coset g*H_m  (protocol.py:309-347, evaluated pointwise)
template <class P>
__global__ __launch_bounds__(256) void quotient_kernel(uint64_t m, QuotientArgs<P> q, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const size_t o = i * P::W;
    const Fp<P> r2 = fp_const<P>(P::R2);
    Fp<P> a = load_fr<P>(q.a + o), b = load_fr<P>(q.b + o), c = load_fr<P>(q.c + o);
    // gate: (a qL + b qR + c qO + (a b) qM) / R, then back to scale 1, then the constant column and PI
    Fp<P> ab = fp_mul<P>(fp_mul<P>(a, b), r2);
    Fp<P> g = fp_add<P>(fp_add<P>(fp_mul<P>(a, load_fr<P>(q.ql + o)), fp_mul<P>(b, load_fr<P>(q.qr + o))),
                        fp_add<P>(fp_mul<P>(c, load_fr<P>(q.qo + o)), fp_mul<P>(ab, load_fr<P>(q.qm + o))));
    g = fp_add<P>(fp_mul<P>(g, r2), fp_add<P>(load_fr<P>(q.qc + o), load_fr<P>(q.pi + o)));
    // permutation argument
    Fp<P> ag = fp_add<P>(a, q.gamma), bg = fp_add<P>(b, q.gamma), cg = fp_add<P>(c, q.gamma);
    Fp<P> bx = fp_mul<P>(load_fr<P>(q.x + o), q.beta_m);
    Fp<P> bx2 = fp_add<P>(bx, bx);
    Fp<P> z = load_fr<P>(q.z + o);
    Fp<P> left = fp_mul<P>(fp_mul<P>(fp_mul<P>(fp_mul<P>(fp_add<P>(ag, bx), fp_add<P>(bg, bx2)), fp_add<P>(cg, fp_add<P>(bx2, bx))), z), q.alpha_r4);
    uint64_t iw = i + q.period;
    if (iw >= m) iw -= m;
    Fp<P> zw = load_fr<P>(q.z + iw * P::W);
    Fp<P> t1 = fp_add<P>(ag, fp_mul<P>(load_fr<P>(q.s1 + o), q.beta_m));
    Fp<P> t2 = fp_add<P>(bg, fp_mul<P>(load_fr<P>(q.s2 + o), q.beta_m));
    Fp<P> t3 = fp_add<P>(cg, fp_mul<P>(load_fr<P>(q.s3 + o), q.beta_m));
    Fp<P> right = fp_mul<P>(fp_mul<P>(fp_mul<P>(fp_mul<P>(t1, t2), t3), zw), q.alpha_r4);
    Fp<P> one_raw = fp_zero<P>();
    one_raw.v[0] = 1;  // the integer 1 (vectors are canonical, not Montgomery)
    Fp<P> bound = fp_mul<P>(fp_mul<P>(fp_sub<P>(z, one_raw), load_fr<P>(q.l1 + o)), q.alpha2_r2);
    Fp<P> total = fp_add<P>(fp_add<P>(g, fp_sub<P>(left, right)), bound);
    store_fr<P>(out + o, fp_reduce_full<P>(fp_mul<P>(total, q.zh_inv_m[i % q.period])));
}

// Horner evaluation split over threads: thread t evaluates its chunk of `chunk` coefficients at x and weighs it
// with x^(t chunk); a workgroup adds its 256 values in LDS and stores one partial sum (canonical).
template <class P>
__global__ __launch_bounds__(256) void poly_eval_kernel(uint64_t n, uint32_t chunk, const uint32_t* __restrict__ coeffs, Fp<P> x_m,
                                                        uint32_t* __restrict__ partial) {
    __shared__ uint32_t lds[P::N][256];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * chunk;
    Fp<P> acc = fp_zero<P>();
    if (lo < n) {
        uint64_t hi = lo + chunk < n ? lo + chunk : n;
        for (uint64_t k = hi; k-- > lo;) acc = fp_add<P>(fp_mul<P>(acc, x_m), load_fr<P>(coeffs + k * P::W));
        uint32_t e[2] = {(uint32_t)lo, (uint32_t)(lo >> 32)};
        acc = fp_mul<P>(acc, fp_pow<P>(x_m, e, 2));
    }
#pragma unroll
    for (int l = 0; l < P::N; ++l) lds[l][threadIdx.x] = acc.v[l];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            Fp<P> u, v;
#pragma unroll
            for (int l = 0; l < P::N; ++l) { u.v[l] = lds[l][threadIdx.x]; v.v[l] = lds[l][threadIdx.x + s]; }
            u = fp_add<P>(u, v);
#pragma unroll
            for (int l = 0; l < P::N; ++l) lds[l][threadIdx.x] = u.v[l];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Fp<P> u;
#pragma unroll
        for (int l = 0; l < P::N; ++l) u.v[l] = lds[l][0];
        store_fr<P>(partial + (size_t)blockIdx.x * P::W, fp_reduce_full<P>(u));
    }
}

// ---- scans over Fr (grand product, division by a linear factor) ---------------------------------------------
// Three launches: every thread folds its chunk of `chunk` consecutive elements (scan_reduce), ONE workgroup scans the
// <= 16384 chunk totals (scan_block), every thread replays its chunk from its exclusive prefix (scan_apply).
// `reverse` runs the same over the mirrored index, i.e. a suffix scan.  Product scans work in Montgomery form:
// elements are converted on load and the results stay Montgomery (consumed by grand_product_finish_kernel).

constexpr uint32_t SCAN_MAX_PARTIALS = 16384;  // 256 threads x 64 totals in scan_block

struct ScanAdd {
    template <class P> static __device__ __forceinline__ Fp<P> identity() { return fp_zero<P>(); }
    template <class P> static __device__ __forceinline__ Fp<P> load(const Fp<P>& x) { return x; }
    template <class P> static __device__ __forceinline__ Fp<P> op(const Fp<P>& a, const Fp<P>& b) { return fp_add<P>(a, b); }
};
struct ScanMul {
    template <class P> static __device__ __forceinline__ Fp<P> identity() { return fp_one<P>(); }
    template <class P> static __device__ __forceinline__ Fp<P> load(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::R2)); }
    template <class P> static __device__ __forceinline__ Fp<P> op(const Fp<P>& a, const Fp<P>& b) { return fp_mul<P>(a, b); }
};

template <class P, class Op>
__global__ __launch_bounds__(256) void fr_scan_reduce_kernel(uint64_t n, uint32_t chunk, int reverse, const uint32_t* __restrict__ in,
                                                          uint32_t* __restrict__ partial) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * chunk;
    if (lo >= n) return;
    const uint64_t hi = lo + chunk < n ? lo + chunk : n;
    Fp<P> acc = Op::template identity<P>();
    for (uint64_t k = lo; k < hi; ++k) {
        const uint64_t idx = reverse ? n - 1 - k : k;
        acc = Op::template op<P>(acc, Op::template load<P>(load_fr<P>(in + idx * P::W)));
    }
    store_fr<P>(partial + t * P::W, fp_reduce_full<P>(acc));
}

// in place: partial[t] <- partial[0] o .. o partial[t-1]   (exclusive), count <= SCAN_MAX_PARTIALS, one workgroup
template <class P, class Op>
__global__ __launch_bounds__(256) void fr_scan_block_kernel(uint32_t count, uint32_t* __restrict__ partial) {
    __shared__ uint32_t lds[P::N][256];
    const uint32_t per = (count + 255) / 256;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < count ? lo + per : count;
    Fp<P> acc = Op::template identity<P>();
    for (uint32_t k = lo; k < hi; ++k) acc = Op::template op<P>(acc, load_fr<P>(partial + (size_t)k * P::W));
#pragma unroll
    for (int l = 0; l < P::N; ++l) lds[l][threadIdx.x] = acc.v[l];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread totals
        Fp<P> other;
        const bool take = (int)threadIdx.x >= d;
        if (take) {
#pragma unroll
            for (int l = 0; l < P::N; ++l) other.v[l] = lds[l][threadIdx.x - d];
        }
        __syncthreads();
        if (take) {
            acc = Op::template op<P>(other, acc);
#pragma unroll
            for (int l = 0; l < P::N; ++l) lds[l][threadIdx.x] = acc.v[l];
        }
        __syncthreads();
    }
    Fp<P> run = Op::template identity<P>();
    if (threadIdx.x > 0) {
#pragma unroll
        for (int l = 0; l < P::N; ++l) run.v[l] = lds[l][threadIdx.x - 1];
    }
    for (uint32_t k = lo; k < hi; ++k) {
        Fp<P> e = load_fr<P>(partial + (size_t)k * P::W);
        store_fr<P>(partial + (size_t)k * P::W, fp_reduce_full<P>(run));
        run = Op::template op<P>(run, e);
    }
}

// inclusive scan: out[idx(k)] = e_0 o .. o e_k over the (possibly mirrored) order
template <class P, class Op>
__global__ __launch_bounds__(256) void fr_scan_apply_kernel(uint64_t n, uint32_t chunk, int reverse, const uint32_t* __restrict__ in,
                                                         const uint32_t* __restrict__ partial, uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * chunk;
    if (lo >= n) return;
    const uint64_t hi = lo + chunk < n ? lo + chunk : n;
    Fp<P> acc = load_fr<P>(partial + t * P::W);
    for (uint64_t k = lo; k < hi; ++k) {
        const uint64_t idx = reverse ? n - 1 - k : k;
        acc = Op::template op<P>(acc, Op::template load<P>(load_fr<P>(in + idx * P::W)));
        store_fr<P>(out + idx * P::W, fp_reduce_full<P>(acc));
    }
}

// out[i] = x[i] * c0 * base^i   (c0_m, base_m in Montgomery form; x and out canonical)
template <class P>
__global__ __launch_bounds__(256) void geom_scale_kernel(uint64_t n, uint32_t chunk, const uint32_t* __restrict__ x, Fp<P> c0_m, Fp<P> base_m,
                                                         uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * chunk;
    if (lo >= n) return;
    const uint64_t hi = lo + chunk < n ? lo + chunk : n;
    uint32_t e[2] = {(uint32_t)lo, (uint32_t)(lo >> 32)};
    Fp<P> w = fp_mul<P>(c0_m, fp_pow<P>(base_m, e, 2));  // c0 base^lo, Montgomery
    for (uint64_t k = lo; k < hi; ++k) {
        store_fr<P>(out + k * P::W, fp_reduce_full<P>(fp_mul<P>(load_fr<P>(x + k * P::W), w)));
        w = fp_mul<P>(w, base_m);
    }
}

// out[j] = PN[j] * SD[j] / Dtot  for j <= n  (PN, SD Montgomery; dinv the canonical integer 1/Dtot)
template <class P>
__global__ __launch_bounds__(256) void grand_product_finish_kernel(uint64_t count, const uint32_t* __restrict__ pn, const uint32_t* __restrict__ sd,
                                                                   Fp<P> dinv, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fp<P> v = fp_mul<P>(fp_mul<P>(load_fr<P>(pn + i * P::W), load_fr<P>(sd + i * P::W)), dinv);
    store_fr<P>(out + i * P::W, fp_reduce_full<P>(v));
}

template <class P, class Op>
static int inclusive_scan(uint64_t n, int reverse, const uint32_t* in, uint32_t* out, uint32_t* partial, hipStream_t st) {
    if (n == 0) return ZK_OK;
    uint64_t chunk = (n + SCAN_MAX_PARTIALS - 1) / SCAN_MAX_PARTIALS;
    if (chunk < 16) chunk = 16;
    const uint64_t threads = (n + chunk - 1) / chunk;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL((fr_scan_reduce_kernel<P, Op>), dim3(blocks), dim3(256), 0, st, n, (uint32_t)chunk, reverse, in, partial);
    hipLaunchKernelGGL((fr_scan_block_kernel<P, Op>), dim3(1), dim3(256), 0, st, (uint32_t)threads, partial);
    hipLaunchKernelGGL((fr_scan_apply_kernel<P, Op>), dim3(blocks), dim3(256), 0, st, n, (uint32_t)chunk, reverse, in, partial, out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int grand_product_dev_impl(uint64_t n, const void* num, const void* den, void* out, hipStream_t st) {
    const size_t eb = P::W * 4;
    uint32_t* work = nullptr;  // PN (n+1) | SD (n+1) | partials
    ZK_ALLOC(&work, (2 * (n + 1) + SCAN_MAX_PARTIALS) * eb);
    uint32_t *pn = work, *sd = work + (n + 1) * P::W, *partial = work + 2 * (n + 1) * P::W;
    int rc = ZK_OK;
    do {
        uint32_t one_m[P::W], dtot[P::W];
        fp_pack<P>(one_m, fp_reduce_full<P>(fp_one<P>()));
        if (hipMemcpyAsync(pn, one_m, eb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(sd + n * P::W, one_m, eb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(ZK_ERR_HIP, "grand product: upload failed"); break; }
        if ((rc = inclusive_scan<P, ScanMul>(n, 0, (const uint32_t*)num, pn + P::W, partial, st))) break;
        if ((rc = inclusive_scan<P, ScanMul>(n, 1, (const uint32_t*)den, sd, partial, st))) break;
        if (hipMemcpyAsync(dtot, sd, eb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = fail(ZK_ERR_HIP, "grand product: copy back failed");
            break;
        }
        Fp<P> d = fp_unpack<P>(dtot);  // product of all denominators, Montgomery
        if (fp_is_zero<P>(d)) { rc = fail(ZK_ERR_ARG, "grand product: zero denominator"); break; }
        Fp<P> one_raw = fp_zero<P>();
        one_raw.v[0] = 1;
        Fp<P> dinv = fp_reduce_full<P>(fp_mul<P>(fp_inv<P>(d), one_raw));  // canonical integer 1 / Dtot
        hipLaunchKernelGGL(grand_product_finish_kernel<P>, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, n + 1, pn, sd, dinv, (uint32_t*)out);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(ZK_ERR_HIP, "grand product: finish failed");
    } while (0);
    dev_free_cached(work);
    return rc;
}

// coeffs (n) = q (n - 1) * (X - root) + rem:  b_k = c_k root^k, SS_j = sum_{k >= j} b_k, q_j = SS_{j+1} root^-(j+1), rem = SS_0
template <class P>
static int div_linear_dev_impl(uint64_t n, const void* coeffs, const uint64_t* root_c, void* q, uint64_t* rem, hipStream_t st) {
    const size_t eb = P::W * 4;
    uint32_t* r = reinterpret_cast<uint32_t*>(rem);
    for (int k = 0; k < P::W; ++k) r[k] = 0;
    if (n == 0) return ZK_OK;
    Fp<P> root_m = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(root_c));
    if (fp_is_zero<P>(root_m)) {  // division by X: a shift
        ZK_HIP(hipMemcpyAsync(r, coeffs, eb, hipMemcpyDeviceToHost, st));
        if (n > 1) ZK_HIP(hipMemcpyAsync(q, (const uint32_t*)coeffs + P::W, (n - 1) * eb, hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipStreamSynchronize(st));
        return ZK_OK;
    }
    uint32_t* work = nullptr;  // b / SS (n) | partials
    ZK_ALLOC(&work, (n + SCAN_MAX_PARTIALS) * eb);
    uint32_t* partial = work + n * P::W;
    int rc = ZK_OK;
    do {
        const uint32_t chunk = 16;
        const unsigned blocks = (unsigned)(((n + chunk - 1) / chunk + 255) / 256);
        hipLaunchKernelGGL(geom_scale_kernel<P>, dim3(blocks), dim3(256), 0, st, n, chunk, (const uint32_t*)coeffs, fp_one<P>(), root_m, work);
        if ((rc = inclusive_scan<P, ScanAdd>(n, 1, work, work, partial, st))) break;
        if (hipMemcpyAsync(r, work, eb, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = fail(ZK_ERR_HIP, "div_linear: copy back failed"); break; }
        if (n > 1) {
            Fp<P> rinv_m = fp_inv<P>(root_m);
            const unsigned qb = (unsigned)(((n - 1 + chunk - 1) / chunk + 255) / 256);
            hipLaunchKernelGGL(geom_scale_kernel<P>, dim3(qb), dim3(256), 0, st, n - 1, chunk, work + P::W, rinv_m, rinv_m, (uint32_t*)q);
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(ZK_ERR_HIP, "div_linear: kernel failed");
    } while (0);
    dev_free_cached(work);
    return rc;
}

// ---- host wrappers ------------------------------------------------------------------------------------

template <class P>
static int axpby_impl(uint64_t n, const uint64_t* a, const void* x, const uint64_t* b, const void* y, const uint64_t* c, void* out, hipStream_t st) {
    if (n == 0) return ZK_OK;
    if (!a || !x || !out) return fail(ZK_ERR_ARG, "axpby: null argument");
    if ((b == nullptr) != (y == nullptr)) return fail(ZK_ERR_ARG, "axpby: b and y go together");
    Fp<P> am = scalar_times_r<P>(a, 1), bm = b ? scalar_times_r<P>(b, 1) : fp_zero<P>(), cc = c ? scalar_in<P>(c) : fp_zero<P>();
    hipLaunchKernelGGL(axpby_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, am, (const uint32_t*)x, bm,
                       (const uint32_t*)y, cc, (uint32_t*)out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int lincomb_impl(uint64_t n_acc, void* acc, int k, const uint64_t* counts, const void* const* xs, const uint64_t* scalars, int n_at,
                        const uint64_t* at_index, const uint64_t* at_vals, hipStream_t st) {
    if (k < 0 || k > LINCOMB_MAX_TERMS || n_at < 0 || n_at > LINCOMB_MAX_AT) return fail(ZK_ERR_ARG, "lincomb: at most 16 terms and 8 single updates");
    if (n_acc == 0 || (k == 0 && n_at == 0)) return ZK_OK;
    if (!acc || (k && (!counts || !xs || !scalars)) || (n_at && (!at_index || !at_vals))) return fail(ZK_ERR_ARG, "lincomb: null argument");
    LincombArgs<P> a;
    a.k = k;
    a.n_at = n_at;
    uint64_t reach = 0;   // the launch only covers what some term or update touches
    for (int t = 0; t < k; ++t) {
        if (counts[t] > n_acc) return fail(ZK_ERR_ARG, "lincomb: a term is longer than the accumulator");
        if (counts[t] && !xs[t]) return fail(ZK_ERR_ARG, "lincomb: null term");
        a.x[t] = (const uint32_t*)xs[t];
        a.count[t] = counts[t];
        a.s_m[t] = scalar_times_r<P>(scalars + 4 * t, 1);
        reach = std::max<uint64_t>(reach, counts[t]);
    }
    for (int j = 0; j < n_at; ++j) {
        if (at_index[j] >= n_acc) return fail(ZK_ERR_ARG, "lincomb: update index beyond the accumulator");
        a.at[j] = at_index[j];
        a.at_val[j] = scalar_in<P>(at_vals + 4 * j);
        reach = std::max<uint64_t>(reach, at_index[j] + 1);
    }
    hipLaunchKernelGGL(lincomb_kernel<P>, dim3((unsigned)((reach + 255) / 256)), dim3(256), 0, st, reach, (uint32_t*)acc, a);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int is_zero_impl(uint64_t n, const void* x, int* is_zero, hipStream_t st) {
    *is_zero = 1;
    if (n == 0) return ZK_OK;
    int* dflag = nullptr;
    ZK_ALLOC(&dflag, sizeof(int));
    int flag = 0, rc = ZK_OK;
    do {
        if (hipMemsetAsync(dflag, 0, sizeof(int), st) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemsetAsync failed"); break; }
        uint64_t nv = n * (P::W / 4);
        hipLaunchKernelGGL(is_zero_kernel<P>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, nv, (const uint4*)x, dflag);
        if (hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = fail(ZK_ERR_HIP, "is_zero: copy back failed"); break; }
    } while (0);
    dev_free_cached(dflag);
    *is_zero = flag ? 0 : 1;
    return rc;
}

template <class P>
static int perm_terms_impl(uint64_t n, const void* const* w, const void* const* l, const uint64_t* beta, const uint64_t* gamma, void* out, hipStream_t st) {
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(perm_terms_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const uint32_t*)w[0], (const uint32_t*)w[1],
                       (const uint32_t*)w[2], (const uint32_t*)l[0], (const uint32_t*)l[1], (const uint32_t*)l[2], scalar_times_r<P>(beta, 1),
                       scalar_in<P>(gamma), (uint32_t*)out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int quotient_impl(uint64_t m, uint64_t n, const void* const* cols, const uint64_t* zh_inv, const uint64_t* beta, const uint64_t* gamma,
                         const uint64_t* alpha, void* out, hipStream_t st) {
    if (n == 0 || m % n != 0 || m / n > QUOT_MAX_PERIOD || m / n < 2) return fail(ZK_ERR_ARG, "quotient: coset size must be 2..16 times n");
    QuotientArgs<P> q;
    const uint32_t** dst[15] = {&q.a, &q.b, &q.c, &q.z, &q.pi, &q.ql, &q.qr, &q.qo, &q.qm, &q.qc, &q.s1, &q.s2, &q.s3, &q.x, &q.l1};
    for (int k = 0; k < 15; ++k) {
        if (!cols[k]) return fail(ZK_ERR_ARG, "quotient: null column");
        *dst[k] = (const uint32_t*)cols[k];
    }
    q.period = (uint32_t)(m / n);
    q.beta_m = scalar_times_r<P>(beta, 1);
    q.gamma = scalar_in<P>(gamma);
    q.alpha_r4 = scalar_times_r<P>(alpha, 4);
    Fp<P> am = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(alpha));
    Fp<P> a2 = fp_mul<P>(am, am);  // alpha^2 R
    q.alpha2_r2 = fp_reduce_full<P>(fp_mul<P>(a2, fp_const<P>(P::R2)));
    for (uint32_t k = 0; k < QUOT_MAX_PERIOD; ++k)
        q.zh_inv_m[k] = k < q.period ? scalar_times_r<P>(zh_inv + 4 * k, 1) : fp_zero<P>();
    hipLaunchKernelGGL(quotient_kernel<P>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, q, (uint32_t*)out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class P>
static int poly_eval_impl_dev(uint64_t n, const void* coeffs, const uint64_t* x, uint64_t* out, hipStream_t st) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (int k = 0; k < P::W; ++k) o[k] = 0;
    if (n == 0) return ZK_OK;
    const uint32_t chunk = 32;
    uint64_t threads = (n + chunk - 1) / chunk;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    uint32_t* dpart = nullptr;
    ZK_ALLOC(&dpart, (size_t)blocks * P::W * 4);
    std::vector<uint32_t> part((size_t)blocks * P::W);
    int rc = ZK_OK;
    hipLaunchKernelGGL(poly_eval_kernel<P>, dim3(blocks), dim3(256), 0, st, n, chunk, (const uint32_t*)coeffs,
                       fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(x)), dpart);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        rc = fail(ZK_ERR_HIP, "poly_eval: kernel or copy back failed");
    dev_free_cached(dpart);
    if (rc) return rc;
    Fp<P> acc = fp_zero<P>();
    for (unsigned b = 0; b < blocks; ++b) acc = fp_add<P>(acc, fp_unpack<P>(part.data() + (size_t)b * P::W));
    fp_pack<P>(o, fp_reduce_full<P>(acc));
    return ZK_OK;
}

// k evaluations with ONE synchronisation: all kernels go out first (each into its own slice of the partial-sum buffer)
template <class P>
static int poly_eval_many_impl(int k, const uint64_t* counts, const void* const* coeffs, const uint64_t* xs, uint64_t* outs, hipStream_t st) {
    const uint32_t chunk = 32;
    std::vector<unsigned> blocks(k), first(k);
    unsigned total = 0;
    for (int i = 0; i < k; ++i) {
        blocks[i] = (unsigned)(((counts[i] + chunk - 1) / chunk + 255) / 256);
        first[i] = total;
        total += blocks[i];
    }
    for (int i = 0; i < k * P::W / 2; ++i) outs[i] = 0;
    if (total == 0) return ZK_OK;
    uint32_t* dpart = nullptr;
    ZK_ALLOC(&dpart, (size_t)total * P::W * 4);
    std::vector<uint32_t> part((size_t)total * P::W);
    int rc = ZK_OK;
    for (int i = 0; i < k; ++i)
        if (counts[i])
            hipLaunchKernelGGL(poly_eval_kernel<P>, dim3(blocks[i]), dim3(256), 0, st, counts[i], chunk, (const uint32_t*)coeffs[i],
                               fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(xs + (size_t)i * (P::W / 2))), dpart + (size_t)first[i] * P::W);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        rc = fail(ZK_ERR_HIP, "poly_eval: kernel or copy back failed");
    dev_free_cached(dpart);
    if (rc) return rc;
    for (int i = 0; i < k; ++i) {
        Fp<P> acc = fp_zero<P>();
        for (unsigned b = 0; b < blocks[i]; ++b) acc = fp_add<P>(acc, fp_unpack<P>(part.data() + (size_t)(first[i] + b) * P::W));
        fp_pack<P>(reinterpret_cast<uint32_t*>(outs + (size_t)i * (P::W / 2)), fp_reduce_full<P>(acc));
    }
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_poly_eval_many_dev(int curve, int k, const uint64_t* counts, const void* const* d_coeffs, const uint64_t* xs, uint64_t* outs, void* stream) {
    if (k < 0 || k > 64) return fail(ZK_ERR_ARG, "poly_eval_many: 0 .. 64 polynomials");
#define CALL(P) return poly_eval_many_impl<P>(k, counts, d_coeffs, xs, outs, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_axpby_dev(int curve, uint64_t n, const uint64_t* a, const void* d_x, const uint64_t* b, const void* d_y, const uint64_t* c,
                     void* d_out, void* stream) {
#define CALL(P) return axpby_impl<P>(n, a, d_x, b, d_y, c, d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_lincomb_dev(int curve, uint64_t n_acc, void* d_acc, int k, const uint64_t* counts, const void* const* d_x, const uint64_t* scalars,
                       int n_at, const uint64_t* at_index, const uint64_t* at_vals, void* stream) {
#define CALL(P) return lincomb_impl<P>(n_acc, d_acc, k, counts, d_x, scalars, n_at, at_index, at_vals, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_gather_dev(int curve, uint64_t n, const void* d_src, uint64_t stride, uint64_t offset, void* d_dst, void* stream) {
#define CALL(P)                                                                                                            \
    {                                                                                                                      \
        if (n == 0) return ZK_OK;                                                                                          \
        const uint64_t nv = n * (P::W / 4);                                                                                \
        hipLaunchKernelGGL(gather_kernel<P>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,     \
                           (const uint4*)d_src, stride, offset, (uint4*)d_dst);                                            \
        ZK_HIP(hipGetLastError());                                                                                         \
        return ZK_OK;                                                                                                      \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_vec_is_zero_dev(int curve, uint64_t n, const void* d_x, int* is_zero, void* stream) {
#define CALL(P) return is_zero_impl<P>(n, d_x, is_zero, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_poly_eval_dev(int curve, uint64_t n, const void* d_coeffs, const uint64_t* x, uint64_t* out, void* stream) {
#define CALL(P) return poly_eval_impl_dev<P>(n, d_coeffs, x, out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_plonk_grand_product_dev(int curve, uint64_t n, const void* d_num, const void* d_den, void* d_out, void* stream) {
#define CALL(P) return grand_product_dev_impl<P>(n, d_num, d_den, d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_poly_div_linear_dev(int curve, uint64_t n, const void* d_coeffs, const uint64_t* root, void* d_q, uint64_t* rem, void* stream) {
#define CALL(P) return div_linear_dev_impl<P>(n, d_coeffs, root, d_q, rem, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_plonk_perm_terms_dev(int curve, uint64_t n, const void* const* d_wires, const void* const* d_labels, const uint64_t* beta,
                            const uint64_t* gamma, void* d_out, void* stream) {
#define CALL(P) return perm_terms_impl<P>(n, d_wires, d_labels, beta, gamma, d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_plonk_quotient_dev(int curve, uint64_t m, uint64_t n, const void* const* d_cols, const uint64_t* zh_inv, const uint64_t* beta,
                          const uint64_t* gamma, const uint64_t* alpha, void* d_out, void* stream) {
#define CALL(P) return quotient_impl<P>(m, n, d_cols, zh_inv, beta, gamma, alpha, d_out, (hipStream_t)stream)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

}  // extern "C"
