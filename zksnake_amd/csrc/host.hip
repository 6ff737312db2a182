// host.hip -- the host-side part of libzkmi: lifecycle, single-point arithmetic, the
// compressed point codecs and a few scalar-field helpers used by setup/verify.
//
// These stand in for the PointG1/PointG2 pyclass methods of the reference
// (src/bn254/curve.rs:25-186,194-324 and the bls12_381 twin): __add__/__neg__/__mul__,
// to_bytes/from_bytes (ark-serialize 0.4.2 compressed encodings),
// g1()/g2(), and for get_evaluation_point / evaluate_lagrange_coefficients
// (src/bn254/polynomial.rs:518-533,645-652).  The reference runs all of these on the CPU as
// well: they touch one point (or O(n) scalars at setup time), not the proving hot path.
#include <dirent.h>
#include <unistd.h>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "common.hip.h"
#include "codec.hip.h"

extern "C" void zk_ntt_free_cache(void);
extern "C" void zk_msm_free_all(void);

namespace zkmi {

// ---- caching device allocator (common.hip.h) ----------------------------------------------------------------
static std::mutex g_cache_mutex;
static std::multimap<size_t, void*> g_cache_free;          // size -> block
static std::map<void*, size_t> g_cache_live;               // block -> size, for blocks handed out through the cache
static size_t g_cache_bytes = 0;
static const size_t CACHE_BUDGET = 6ull << 30;             // bytes kept for reuse
static const size_t CACHE_MAX_BLOCK = 2ull << 30;          // larger blocks (fixed-base tables) are never kept

int dev_alloc_cached(void** p, size_t bytes) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_cache_free.find(bytes);
        if (it != g_cache_free.end()) {
            *p = it->second;
            g_cache_free.erase(it);
            g_cache_bytes -= bytes;
            g_cache_live[*p] = bytes;
            return ZK_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        dev_cache_release();  // give everything back and try once more
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) return fail(ZK_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    g_cache_live[*p] = bytes;
    return ZK_OK;
}

void dev_free_cached(void* p) {
    if (!p) return;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_cache_live.find(p);
        if (it != g_cache_live.end()) {
            bytes = it->second;
            g_cache_live.erase(it);
            if (bytes <= CACHE_MAX_BLOCK && g_cache_bytes + bytes <= CACHE_BUDGET) {
                g_cache_free.emplace(bytes, p);
                g_cache_bytes += bytes;
                return;
            }
        }
    }
    (void)hipFree(p);
}

static std::vector<hipStream_t> g_stream_pool[2];
static std::multimap<size_t, void*> g_pinned_free;
static std::map<void*, size_t> g_pinned_live;

void dev_cache_release() {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    for (auto& kv : g_cache_free) (void)hipFree(kv.second);
    g_cache_free.clear();
    g_cache_bytes = 0;
    for (auto& pool : g_stream_pool) {
        for (hipStream_t st : pool) (void)hipStreamDestroy(st);
        pool.clear();
    }
    for (auto& kv : g_pinned_free) (void)hipHostFree(kv.second);
    g_pinned_free.clear();
}

int stream_acquire(bool high_priority, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto& pool = g_stream_pool[high_priority ? 1 : 0];
        if (!pool.empty()) {
            *out = pool.back();
            pool.pop_back();
            return ZK_OK;
        }
    }
    int lo = 0, hi = 0;
    ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // numerically lower = higher priority
    ZK_HIP(hipStreamCreateWithPriority(out, hipStreamDefault, high_priority ? hi : lo));
    return ZK_OK;
}

void stream_release(bool high_priority, hipStream_t st) {
    if (!st) return;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto& pool = g_stream_pool[high_priority ? 1 : 0];
    if (pool.size() < 32) pool.push_back(st);
    else (void)hipStreamDestroy(st);
}

int pinned_alloc_cached(void** p, size_t bytes) {
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_pinned_free.find(bytes);
        if (it != g_pinned_free.end()) {
            *p = it->second;
            g_pinned_free.erase(it);
            g_pinned_live[*p] = bytes;
            return ZK_OK;
        }
    }
    ZK_HIP(hipHostMalloc(p, bytes));
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    g_pinned_live[*p] = bytes;
    return ZK_OK;
}

void pinned_free_cached(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_pinned_live.find(p);
    if (it != g_pinned_live.end() && g_pinned_free.size() < 64 && it->second <= (1u << 20)) {
        g_pinned_free.emplace(it->second, p);
        g_pinned_live.erase(it);
        return;
    }
    if (it != g_pinned_live.end()) g_pinned_live.erase(it);
    (void)hipHostFree(p);
}


template <class G>
static Affine<typename G::F> load_point(const uint64_t* src) {
    typedef typename G::F F;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
    Affine<F> p;
    p.x = F::from_canonical(w);
    p.y = F::from_canonical(w + F::LIMBS);
    return p;
}

template <class G>
static void store_point(uint64_t* dst, const Affine<typename G::F>& p) {
    typedef typename G::F F;
    uint32_t* w = reinterpret_cast<uint32_t*>(dst);
    F::to_canonical(w, p.x);
    F::to_canonical(w + F::LIMBS, p.y);
}

// the same on the 64-bit-limb host arithmetic (host64.hip.h): add / sum / scalar multiplication of single points, what a
// proof's assembly spends its host time on (~20 operations per proof, python/zksnake/groth16/protocol.py:133-163)
template <class G>
static Affine<typename G::HostF> load_point64(const uint64_t* src) {
    typedef typename G::HostF HF;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(src);
    return {HF::from_canonical(w), HF::from_canonical(w + HF::LIMBS)};
}
template <class G>
static void store_point64(uint64_t* dst, const Affine<typename G::HostF>& p) {
    G::HostF::affine_to_canonical(p, dst);
}

template <class G>
static bool point_on_curve(const Affine<typename G::F>& p) {
    typedef typename G::F F;
    typename F::T b = F::from_canonical(CurveConsts<G>::b());
    return aff_on_curve<F>(p, b);
}

template <class FrP>
static void reduce_scalar(uint32_t* k, const uint64_t* scalar) {
    memcpy(k, scalar, FrP::W * 4);
    for (int r = 0; r < 10; ++r) {
        uint32_t t[FrP::W];
        if (fp_sub_mod_raw<FrP>(t, k)) break;
        memcpy(k, t, sizeof(t));
    }
}

// ---- compressed encodings (codec.hip.h holds the arithmetic, shared with the batched GPU kernels) ----------

template <class G>
static int compress_impl(const uint64_t* a, uint8_t* out) {
    int code = point_encode<G>(reinterpret_cast<const uint32_t*>(a), out);
    return code ? fail(ZK_ERR_POINT, codec_message(code)) : ZK_OK;
}

template <class G>
static int decompress_impl(const uint8_t* in, uint64_t* out) {
    int code = point_decode<G>(in, reinterpret_cast<uint32_t*>(out));
    return code ? fail(ZK_ERR_POINT, codec_message(code)) : ZK_OK;
}

// ---- scalar-field helpers ---------------------------------------------------------------------------

template <class P>
static Fp<P> fr_root(uint64_t n) {
    int log_n = log2_u64(n);
    Fp<P> w = fp_const<P>(P::ROOT);
    for (int k = 0; k < P::TWO_ADICITY - log_n; ++k) w = fp_sqr<P>(w);
    return w;
}

template <class P>
static int lagrange_impl(uint64_t n_in, const uint64_t* tau_c, uint64_t* out) {
    uint64_t n = next_pow2_u64(n_in == 0 ? 1 : n_in);
    int log_n = log2_u64(n);
    if (log_n > P::TWO_ADICITY) return fail(ZK_ERR_DOMAIN, "Domain size is too large");
    Fp<P> tau = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(tau_c));
    Fp<P> w = fr_root<P>(n);
    Fp<P> tn = tau;
    for (int k = 0; k < log_n; ++k) tn = fp_sqr<P>(tn);
    Fp<P> z = fp_sub<P>(tn, fp_one<P>());
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    if (fp_is_zero<P>(z)) {
        Fp<P> cur = fp_one<P>();
        for (uint64_t i = 0; i < n; ++i) {
            Fp<P> v = fp_eq<P>(cur, tau) ? fp_one<P>() : fp_zero<P>();
            fp_to_canonical<P>(o + i * P::W, v);
            cur = fp_mul<P>(cur, w);
        }
        return ZK_OK;
    }
    uint32_t nn[P::W] = {0};
    nn[0] = (uint32_t)n;
    nn[1] = (uint32_t)(n >> 32);
    Fp<P> zn = fp_mul<P>(z, fp_inv<P>(fp_from_canonical<P>(nn)));
    // L_i = zn * w^i / (tau - w^i): batch-invert the denominators
    std::vector<Fp<P>> den(n), pre(n), wi(n);
    Fp<P> cur = fp_one<P>(), run = fp_one<P>();
    for (uint64_t i = 0; i < n; ++i) {
        wi[i] = cur;
        den[i] = fp_sub<P>(tau, cur);
        pre[i] = run;
        run = fp_mul<P>(run, den[i]);
        cur = fp_mul<P>(cur, w);
    }
    Fp<P> inv = fp_inv<P>(run);
    for (uint64_t i = n; i-- > 0;) {
        Fp<P> di = fp_mul<P>(inv, pre[i]);
        inv = fp_mul<P>(inv, den[i]);
        Fp<P> l = fp_mul<P>(fp_mul<P>(zn, wi[i]), di);
        fp_to_canonical<P>(o + i * P::W, l);
    }
    return ZK_OK;
}

// ---- dense-polynomial helpers over Fr for the PlonK prover (host, O(n) sequential recurrences) ----
template <class P>
static int poly_eval_impl(uint64_t n, const uint64_t* coeffs, const uint64_t* x_c, uint64_t* out) {
    const uint32_t* c = reinterpret_cast<const uint32_t*>(coeffs);
    Fp<P> x = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(x_c));
    Fp<P> acc = fp_zero<P>();
    for (uint64_t i = n; i-- > 0;) acc = fp_add<P>(fp_mul<P>(acc, x), fp_from_canonical<P>(c + i * P::W));
    fp_to_canonical<P>(reinterpret_cast<uint32_t*>(out), acc);
    return ZK_OK;
}

// coeffs (n) = q (n - 1) * (X - root) + rem
template <class P>
static int poly_div_linear_impl(uint64_t n, const uint64_t* coeffs, const uint64_t* root_c, uint64_t* q, uint64_t* rem) {
    const uint32_t* c = reinterpret_cast<const uint32_t*>(coeffs);
    uint32_t* qo = reinterpret_cast<uint32_t*>(q);
    Fp<P> root = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(root_c));
    Fp<P> carry = fp_zero<P>();
    for (uint64_t i = n; i-- > 1;) {
        carry = fp_add<P>(fp_from_canonical<P>(c + i * P::W), fp_mul<P>(carry, root));
        fp_to_canonical<P>(qo + (i - 1) * P::W, carry);
    }
    Fp<P> r0 = n ? fp_add<P>(fp_from_canonical<P>(c), fp_mul<P>(carry, root)) : fp_zero<P>();
    fp_to_canonical<P>(reinterpret_cast<uint32_t*>(rem), r0);
    return ZK_OK;
}

// out[0] = 1, out[i + 1] = out[i] * num[i] / den[i]   (one field inversion in total)
template <class P>
static int grand_product_impl(uint64_t n, const uint64_t* num, const uint64_t* den, uint64_t* out) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(num);
    const uint32_t* b = reinterpret_cast<const uint32_t*>(den);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    std::vector<Fp<P>> d(n), pre(n);
    Fp<P> run = fp_one<P>();
    for (uint64_t i = 0; i < n; ++i) {
        d[i] = fp_from_canonical<P>(b + i * P::W);
        if (fp_is_zero<P>(d[i])) return fail(ZK_ERR_ARG, "grand product: zero denominator");
        pre[i] = run;
        run = fp_mul<P>(run, d[i]);
    }
    Fp<P> inv = fp_inv<P>(run);
    for (uint64_t i = n; i-- > 0;) {  // d[i] <- 1 / den[i]
        Fp<P> di = fp_mul<P>(inv, pre[i]);
        inv = fp_mul<P>(inv, d[i]);
        d[i] = di;
    }
    Fp<P> acc = fp_one<P>();
    fp_to_canonical<P>(o, acc);
    for (uint64_t i = 0; i < n; ++i) {
        acc = fp_mul<P>(fp_mul<P>(acc, fp_from_canonical<P>(a + i * P::W)), d[i]);
        fp_to_canonical<P>(o + (i + 1) * P::W, acc);
    }
    return ZK_OK;
}

// acc[i] += s * x[i] for i < n
template <class P>
static int scale_add_impl(uint64_t n, uint64_t* acc, const uint64_t* x, const uint64_t* s_c) {
    uint32_t* a = reinterpret_cast<uint32_t*>(acc);
    const uint32_t* xs = reinterpret_cast<const uint32_t*>(x);
    Fp<P> s = fp_from_canonical<P>(reinterpret_cast<const uint32_t*>(s_c));
    for (uint64_t i = 0; i < n; ++i) {
        Fp<P> v = fp_add<P>(fp_from_canonical<P>(a + i * P::W), fp_mul<P>(s, fp_from_canonical<P>(xs + i * P::W)));
        fp_to_canonical<P>(a + i * P::W, v);
    }
    return ZK_OK;
}

static int g_device = -1;

// ---- hardware queues ---------------------------------------------------------------------------------------------
// A proof keeps seven HIP streams busy (five MSM plans, the QAP chain, the default stream).  The runtime multiplexes streams
// onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and work of two streams that share a queue runs one after the
// other: with the default the witness-only MSM and the QAP chain of Groth16.prove ended up serialised (kernel trace, round
// 2: ~ +1 ms per proof).  The variable is read when the HIP runtime STARTS (first HIP call of the process), so the library
// puts it in place before its own first HIP call -- unless the caller chose a value, or somebody else (torch, another
// library) started the runtime earlier, which this reports instead of hiding.
constexpr int ZK_WANT_HW_QUEUES = 12;
static std::mutex g_queue_mu;
static int g_queue_status = -1, g_queue_count = 0;

// The ROCm runtime opens /dev/kfd when it starts: an open descriptor on it means the environment was already read.
static bool hip_runtime_started() {
    if (const char* t = getenv("ZKMI_TEST_RUNTIME_STARTED")) return atoi(t) != 0;   // CPU tests: there is no /dev/kfd to open
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool found = false;
    char path[64], target[256];
    while (struct dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(path, sizeof(path), "/proc/self/fd/%s", e->d_name);
        const ssize_t k = readlink(path, target, sizeof(target) - 1);
        if (k <= 0) continue;
        target[k] = 0;
        if (strcmp(target, "/dev/kfd") == 0) { found = true; break; }
    }
    closedir(d);
    return found;
}

static int hw_queues_prepare(int* queues) {
    std::lock_guard<std::mutex> lock(g_queue_mu);
    if (g_queue_status < 0) {
        const bool started = hip_runtime_started();
        const char* env = getenv("GPU_MAX_HW_QUEUES");
        if (env && *env) {
            g_queue_count = atoi(env);
            // set before the runtime started, or by whoever started it: either way it is what the runtime uses.  The
            // Python host of this library applies the same rule when it is imported (it loads the library lazily, see
            // zksnake_amd/_native.py) and says so through ZKMI_HW_QUEUES_SET_BY_LIBRARY.
            const char* ours = getenv("ZKMI_HW_QUEUES_SET_BY_LIBRARY");
            g_queue_status = (ours && *ours == '1' && g_queue_count == ZK_WANT_HW_QUEUES) ? ZK_QUEUES_SET_BY_LIBRARY : ZK_QUEUES_CALLER;
        } else if (!started) {
            char v[16];
            snprintf(v, sizeof(v), "%d", ZK_WANT_HW_QUEUES);
            setenv("GPU_MAX_HW_QUEUES", v, 0);
            setenv("ZKMI_HW_QUEUES_SET_BY_LIBRARY", "1", 1);   // child processes inherit both: they report the same status
            g_queue_count = ZK_WANT_HW_QUEUES;
            g_queue_status = ZK_QUEUES_SET_BY_LIBRARY;
        } else {
            g_queue_count = 0;   // the runtime's default (4)
            g_queue_status = ZK_QUEUES_TOO_LATE;
        }
    }
    if (queues) *queues = g_queue_count;
    return g_queue_status;
}

// busy-wait kernel of the concurrency probe (tests): holds one wave for `ticks` of the constant 100 MHz counter
__global__ void spin_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zk_hw_queues_prepare(int* queues) { return hw_queues_prepare(queues); }

int zk_init(int device) {
    (void)hw_queues_prepare(nullptr);   // before the first HIP call below
    int n = zk_device_count();
    if (n <= 0) return fail(ZK_ERR_HIP, "no HIP device visible: libzkmi has no CPU fallback");
    if (device < 0 || device >= n) return fail(ZK_ERR_ARG, "device index out of range");
    ZK_HIP(hipSetDevice(device));
    g_device = device;
    return ZK_OK;
}

int zk_init_ex(int device, int* queue_status, int* queues) {
    const int st = hw_queues_prepare(queues);
    if (queue_status) *queue_status = st;
    return zk_init(device);
}

int zk_debug_spin_dev(void* stream, uint64_t microseconds) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds * 100ull);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

int zk_shutdown(void) {
    zk_msm_free_all();
    zk_ntt_free_cache();
    dev_cache_release();
    return ZK_OK;
}

int zk_dev_alloc(uint64_t bytes, void** d_ptr) {
    return dev_alloc_cached(d_ptr, bytes ? bytes : 1);
}
int zk_dev_free(void* d_ptr) {
    // hipFree waits for the device before the block can be reused; the caching allocator does not, so wait here
    // (microseconds on an idle device, against ~1 ms for a real hipFree)
    ZK_HIP(hipDeviceSynchronize());
    dev_free_cached(d_ptr);
    return ZK_OK;
}
int zk_dev_upload(void* d_dst, const void* h_src, uint64_t bytes) {
    if (bytes) ZK_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ZK_OK;
}
int zk_dev_upload_async(void* d_dst, const void* h_src, uint64_t bytes, void* stream) {
    if (bytes) ZK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return ZK_OK;
}
int zk_dev_download(void* h_dst, const void* d_src, uint64_t bytes) {
    if (bytes) ZK_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}
int zk_dev_memset(void* d_dst, int value, uint64_t bytes) {
    if (bytes) ZK_HIP(hipMemset(d_dst, value, bytes));
    return ZK_OK;
}
int zk_dev_memset_async(void* d_dst, int value, uint64_t bytes, void* stream) {
    if (bytes) ZK_HIP(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
    return ZK_OK;
}
int zk_host_alloc(uint64_t bytes, void** h_ptr) {
    ZK_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return ZK_OK;
}
int zk_host_free(void* h_ptr) {
    if (h_ptr) ZK_HIP(hipHostFree(h_ptr));
    return ZK_OK;
}
int zk_dev_synchronize(void) {
    ZK_HIP(hipDeviceSynchronize());
    return ZK_OK;
}
int zk_stream_create(int high_priority, void** stream) {
    // a stream that does not synchronise with the legacy default stream, so that work given to it overlaps with the
    // MSM plans' own streams and with default-stream work
    int lo = 0, hi = 0;
    ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // numerically lower = higher priority
    hipStream_t st = nullptr;
    ZK_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? hi : lo));
    *stream = (void*)st;
    return ZK_OK;
}
int zk_stream_destroy(void* stream) {
    if (stream) ZK_HIP(hipStreamDestroy((hipStream_t)stream));
    return ZK_OK;
}
int zk_stream_synchronize(void* stream) {
    ZK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZK_OK;
}

const char* zk_last_error(void) { return last_error_ref().c_str(); }
const char* zk_version(void) { return "zkmi 0.1 (gfx950)"; }

int zk_fq_limbs(int curve) { return curve == ZK_CURVE_BN254 ? 4 : (curve == ZK_CURVE_BLS12_381 ? 6 : -1); }
int zk_point_limbs(int curve, int group) {
    int f = zk_fq_limbs(curve);
    if (f < 0 || (group != ZK_G1 && group != ZK_G2)) return -1;
    return 2 * f * group;
}
int zk_point_bytes(int curve, int group) {
    if (group != ZK_G1 && group != ZK_G2) return -1;
    if (curve == ZK_CURVE_BN254) return 32 * group;
    if (curve == ZK_CURVE_BLS12_381) return 48 * group;
    return -1;
}

int zk_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* out) {
#define CALL(G)                                                              \
    {                                                                        \
        typedef G::HostF F;                                                  \
        XYZZ<F> acc = xyzz_from_affine<F>(load_point64<G>(a));               \
        xyzz_add_affine<F>(acc, load_point64<G>(b));                         \
        store_point64<G>(out, xyzz_to_affine<F>(acc));                       \
        return ZK_OK;                                                        \
    }
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_sum(int curve, int group, uint64_t n, const uint64_t* points, uint64_t* out) {
#define CALL(G)                                                              \
    {                                                                        \
        typedef G::HostF F;                                                  \
        const size_t stride = G::F::LIMBS;  /* 64-bit limbs per affine point */ \
        XYZZ<F> acc = xyzz_inf<F>();                                         \
        for (uint64_t i = 0; i < n; ++i) xyzz_add_affine<F>(acc, load_point64<G>(points + i * stride)); \
        store_point64<G>(out, xyzz_to_affine<F>(acc));                       \
        return ZK_OK;                                                        \
    }
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_neg(int curve, int group, const uint64_t* a, uint64_t* out) {
#define CALL(G)                                                              \
    {                                                                        \
        typedef G::F F;                                                      \
        store_point<G>(out, aff_neg<F>(load_point<G>(a)));                   \
        return ZK_OK;                                                        \
    }
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_mul(int curve, int group, const uint64_t* a, const uint64_t* scalar, uint64_t* out) {
#define CALL(G)                                                              \
    {                                                                        \
        typedef G::HostF F;                                                  \
        uint32_t k[G::Fr::W];                                                \
        reduce_scalar<G::Fr>(k, scalar);                                     \
        XYZZ<F> r = xyzz_scalar_mul<F>(load_point64<G>(a), k, G::Fr::W);     \
        store_point64<G>(out, xyzz_to_affine<F>(r));                         \
        return ZK_OK;                                                        \
    }
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_on_curve(int curve, int group, const uint64_t* a) {
#define CALL(G) return point_on_curve<G>(load_point<G>(a)) ? 1 : 0
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_generator(int curve, int group, uint64_t* out) {
#define CALL(G)                                                                          \
    {                                                                                    \
        memcpy(out, CurveConsts<G>::gen(), (size_t)2 * G::F::LIMBS * 4);                 \
        return ZK_OK;                                                                    \
    }
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_compress(int curve, int group, const uint64_t* a, uint8_t* out) {
#define CALL(G) return compress_impl<G>(a, out)
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_point_decompress(int curve, int group, const uint8_t* in, uint64_t* out) {
#define CALL(G) return decompress_impl<G>(in, out)
    ZK_DISPATCH_GROUP(curve, group, CALL);
#undef CALL
}

int zk_fr_root_of_unity(int curve, uint64_t n, uint64_t* out) {
#define CALL(P)                                                                          \
    {                                                                                    \
        uint64_t m = next_pow2_u64(n == 0 ? 1 : n);                                      \
        if (log2_u64(m) > P::TWO_ADICITY) return fail(ZK_ERR_DOMAIN, "Domain size is too large"); \
        fp_to_canonical<P>(reinterpret_cast<uint32_t*>(out), fr_root<P>(m));             \
        return ZK_OK;                                                                    \
    }
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_fr_lagrange_coeffs(int curve, uint64_t n, const uint64_t* tau, uint64_t* out) {
#define CALL(P) return lagrange_impl<P>(n, tau, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_fr_poly_eval(int curve, uint64_t n, const uint64_t* coeffs, const uint64_t* x, uint64_t* out) {
#define CALL(P) return poly_eval_impl<P>(n, coeffs, x, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_fr_poly_div_linear(int curve, uint64_t n, const uint64_t* coeffs, const uint64_t* root, uint64_t* q, uint64_t* rem) {
#define CALL(P) return poly_div_linear_impl<P>(n, coeffs, root, q, rem)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_fr_grand_product(int curve, uint64_t n, const uint64_t* num, const uint64_t* den, uint64_t* out) {
#define CALL(P) return grand_product_impl<P>(n, num, den, out)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

int zk_fr_scale_add(int curve, uint64_t n, uint64_t* acc, const uint64_t* x, const uint64_t* s) {
#define CALL(P) return scale_add_impl<P>(n, acc, x, s)
    ZK_DISPATCH_FR(curve, CALL);
#undef CALL
}

}  // extern "C"
