// common.hip.h -- shared plumbing of libzkmi: status/error reporting, HIP checks, curve tags.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/zkmi.h"
#include "curve.hip.h"
#include "host64.hip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace zkmi {

// ---- thread-local last error -----------------------------------------------------------
inline std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const std::string& msg) {
    last_error_ref() = msg;
    return code;
}

#if defined(__HIPCC__)
#define ZK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            return ::zkmi::fail(ZK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) +   \
                                                " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
        }                                                                                         \
    } while (0)
#endif

#if defined(__HIPCC__)
// ---- caching device allocator ------------------------------------------------------------------------
// hipMalloc / hipFree cost 50-200 us each and synchronise the device; a one-shot `zk_msm` (plan create + run + destroy)
// makes ~25 of them, 3.5 ms for a two-point MSM.  Freed blocks are kept by exact size (plans of the same shape ask for
// the same sizes again) up to a byte budget; blocks above the budget go straight back to HIP.  zk_shutdown() empties it.
int dev_alloc_cached(void** p, size_t bytes);
void dev_free_cached(void* p);
void dev_cache_release();
// streams and pinned host blocks are pooled the same way (hipStreamCreate costs ~1.5 ms, hipHostMalloc ~0.5 ms)
int stream_acquire(bool high_priority, hipStream_t* out);
void stream_release(bool high_priority, hipStream_t st);
int pinned_alloc_cached(void** p, size_t bytes);
void pinned_free_cached(void* p);
#define ZK_ALLOC(ptr, bytes) ZK_HIP_RC(::zkmi::dev_alloc_cached((void**)(ptr), (bytes)))
#define ZK_HIP_RC(expr)                    \
    do {                                   \
        int _rc = (expr);                  \
        if (_rc != ZK_OK) return _rc;      \
    } while (0)
#endif

// ---- curve / group tags -------------------------------------------------------------------
// Each tag bundles: the coordinate field facade F (Fp or Fp2), the scalar field parameters,
// and the curve constants in canonical form.

struct Bn254G1 {
    typedef FpOps<BnFqParams> F;
    typedef HostTail64<Fp64Ops<BnFqParams>> HostF;  // 64-bit-limb host arithmetic (host64.hip.h)
    typedef BnFrParams Fr;
    static constexpr int CURVE = ZK_CURVE_BN254, GROUP = ZK_G1;
    static constexpr int ENDO_ID = 2 * CURVE + GROUP;   // identity of the group's scalar split (SortExport::endo)
    static constexpr int FQ64 = 4;  // 64-bit limbs per base field element
};
struct Bn254G2 {
    typedef Fp2Ops<BnFqParams> F;
    typedef HostTail64<Fp2Ops64<BnFqParams>> HostF;
    typedef BnFrParams Fr;
    static constexpr int CURVE = ZK_CURVE_BN254, GROUP = ZK_G2;
    static constexpr int ENDO_ID = 2 * CURVE + GROUP;
    static constexpr int FQ64 = 4;
};
struct Bls381G1 {
    typedef FpOps<BlsFqParams> F;
    typedef HostTail64<Fp64Ops<BlsFqParams>> HostF;
    typedef BlsFrParams Fr;
    static constexpr int CURVE = ZK_CURVE_BLS12_381, GROUP = ZK_G1;
    static constexpr int ENDO_ID = 2 * CURVE + GROUP;
    static constexpr int FQ64 = 6;
};
struct Bls381G2 {
    typedef Fp2Ops<BlsFqParams> F;
    typedef HostTail64<Fp2Ops64<BlsFqParams>> HostF;
    typedef BlsFrParams Fr;
    static constexpr int CURVE = ZK_CURVE_BLS12_381, GROUP = ZK_G2;
    static constexpr int ENDO_ID = 2 * CURVE + GROUP;
    static constexpr int FQ64 = 6;
};

// words (u32) per affine point in the packed ABI layout
template <class G>
struct PointLayout {
    static constexpr int COORD_WORDS = G::F::LIMBS;        // u32 per coordinate
    static constexpr int AFFINE_WORDS = 2 * G::F::LIMBS;   // u32 per affine point
    static constexpr int XYZZ_WORDS = 4 * G::F::LIMBS;
};

// call FN<G>(args...) for the (curve, group) pair; FN is a template-template functor
#define ZK_DISPATCH_GROUP(curve, group, CALL)                                           \
    do {                                                                                \
        if ((curve) == ZK_CURVE_BN254 && (group) == ZK_G1) { CALL(::zkmi::Bn254G1); }   \
        else if ((curve) == ZK_CURVE_BN254 && (group) == ZK_G2) { CALL(::zkmi::Bn254G2); } \
        else if ((curve) == ZK_CURVE_BLS12_381 && (group) == ZK_G1) { CALL(::zkmi::Bls381G1); } \
        else if ((curve) == ZK_CURVE_BLS12_381 && (group) == ZK_G2) { CALL(::zkmi::Bls381G2); } \
        else return ::zkmi::fail(ZK_ERR_ARG, "unknown curve/group");                    \
    } while (0)

#define ZK_DISPATCH_FR(curve, CALL)                                                     \
    do {                                                                                \
        if ((curve) == ZK_CURVE_BN254) { CALL(::zkmi::BnFrParams); }                    \
        else if ((curve) == ZK_CURVE_BLS12_381) { CALL(::zkmi::BlsFrParams); }          \
        else return ::zkmi::fail(ZK_ERR_ARG, "unknown curve");                          \
    } while (0)

inline uint64_t next_pow2_u64(uint64_t n) {
    uint64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
inline int log2_u64(uint64_t n) {
    int l = 0;
    while ((1ull << l) < n) ++l;
    return l;
}

}  // namespace zkmi
