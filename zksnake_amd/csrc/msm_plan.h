// msm_plan.h -- interface between the C API (msm.hip) and the per-group kernel translation units
// (msm_group.hip compiled once per curve group, so the four instantiations build in parallel).
#pragma once
#include <cstdint>
#include <mutex>
#include <hip/hip_runtime.h>

namespace zkmi {

// a device allocation with shared ownership (the bases / fixed-base table of a plan and its clones)
void dev_free_cached(void* p);  // common.cuh / host.hip
struct DeviceBlock {
    void* ptr = nullptr;
    ~DeviceBlock() { dev_free_cached(ptr); }
};

// what a plan with a run in flight lends to a second plan that multiplies the SAME scalars against other bases of the
// same length (Groth16: <tau_1, v> in G1 and <tau_2, v> in G2): the digits and the sorted entry list
struct SortExport {
    const uint32_t *sorted = nullptr, *bstart = nullptr, *sstart = nullptr, *big_list = nullptr, *big_count = nullptr;
    uint64_t n = 0;
    uint32_t m = 0, seg_len = 0, groups = 0;
    int c = 0, nwin = 0, w_first = 0, w_count = 0, pw_first = 0, pw_count = 0, scalar_bits = 0;
    bool pre = false;
    hipEvent_t sorted_ready = nullptr;  // recorded on the lender's stream after its sort stage
    hipEvent_t release = nullptr;       // the borrower records it after its last read; the lender's next run waits for it
};

struct MsmPlanBase {
    virtual ~MsmPlanBase() {}
    virtual int export_sort(SortExport* out) = 0;                        // of the run in flight
    virtual int enqueue_shared(MsmPlanBase* lender, hipStream_t stream) = 0;
    // a second plan over the same bases (shared, read-only) with its own workspace and stream: two MSMs against one key
    // in flight together (tau_1 with u and with v in Groth16.prove) without building the fixed-base table twice
    virtual int clone(MsmPlanBase** out) = 0;
    // enqueue() puts every GPU stage plus the D2H copy of the per-window results on `stream` and returns;
    // finish() waits for them and runs the host tail.  run() = enqueue() + finish().
    virtual int enqueue(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count,
                        hipStream_t stream) = 0;
    virtual int finish(uint64_t* out) = 0;
    int run(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, uint64_t* out,
            hipStream_t stream) {
        int rc = enqueue(n_scalars, scalars, on_device, w_first, w_count, stream);
        return rc ? rc : finish(out);
    }
    hipStream_t own_stream = nullptr;  // used when the caller passes ZK_STREAM_PLAN
    int c = 0, nwin = 0;
    float timings[5] = {0, 0, 0, 0, 0};
    std::mutex mu;
};


// factories implemented in msm_group.hip (one object file per group)
#define ZK_DECLARE_GROUP(G)                                                                                              \
    int msm_plan_create_##G(uint64_t n, const void* bases, int on_device, int flags, int window_bits, int window_first,    \
                            int window_count, MsmPlanBase** out);                                                        \
    int msm_batch_mul_##G(uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast, uint64_t* out);     \
    int msm_points_codec_##G(uint64_t n, const void* in, void* out, int to_bytes, uint64_t* bad_index);                  \
    void msm_fixed_table_free_##G();
ZK_DECLARE_GROUP(Bn254G1)
ZK_DECLARE_GROUP(Bn254G2)
ZK_DECLARE_GROUP(Bls381G1)
ZK_DECLARE_GROUP(Bls381G2)
#undef ZK_DECLARE_GROUP

}  // namespace zkmi
