// fr_mem.hip.h -- load / store of one scalar-field element between HBM (W packed 32-bit words, 16-byte vector
// accesses) and registers (N 29-bit limbs).  Shared by ntt.hip and plonk.hip.
#pragma once
#include "common.hip.h"

namespace zkmi {

// an element in memory is W packed 32-bit words; in registers N 29-bit limbs
template <class P>
__device__ __forceinline__ Fp<P> load_fr(const uint32_t* p) {
    uint32_t w[P::W];
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::W / 4; ++i) {
        uint4 t = q[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
    return fp_unpack<P>(w);
}

template <class P>
__device__ __forceinline__ void store_fr(uint32_t* p, const Fp<P>& a) {
    uint32_t w[P::W];
    fp_pack<P>(w, a);
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::W / 4; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

}  // namespace zkmi
