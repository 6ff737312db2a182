// msm.hip -- C API of the MSM / batch multiplication entry points (include/zkmi.h); the kernels live in
// msm_impl.hip.h and are instantiated per curve group in msm_group.hip.
#include <map>
#include <mutex>
#include "common.hip.h"
#include "msm_plan.h"

namespace zkmi {

static std::mutex g_plan_mutex;
static std::map<uint64_t, MsmPlanBase*> g_plans;
static uint64_t g_next_plan = 1;


static int plan_create_dispatch(int curve, int group, uint64_t n, const void* bases, int on_device, int flags,
                                int window_bits, int window_first, int window_count, MsmPlanBase** out) {
#define CALL(G) return msm_plan_create_##G(n, bases, on_device, flags, window_bits, window_first, window_count, out)
    if (curve == ZK_CURVE_BN254 && group == ZK_G1) { CALL(Bn254G1); }
    if (curve == ZK_CURVE_BN254 && group == ZK_G2) { CALL(Bn254G2); }
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G1) { CALL(Bls381G1); }
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G2) { CALL(Bls381G2); }
#undef CALL
    return fail(ZK_ERR_ARG, "unknown curve/group");
}

}  // namespace zkmi


using namespace zkmi;

extern "C" {

int zk_msm_plan_create(int curve, int group, uint64_t n, const void* bases, int bases_on_device, int flags,
                       int window_bits, uint64_t* handle) {
    return zk_msm_plan_create_range(curve, group, n, bases, bases_on_device, flags, window_bits, 0, 0, handle);
}

int zk_msm_plan_create_range(int curve, int group, uint64_t n, const void* bases, int bases_on_device, int flags,
                             int window_bits, int window_first, int window_count, uint64_t* handle) {
    MsmPlanBase* p = nullptr;
    int rc = plan_create_dispatch(curve, group, n, bases, bases_on_device, flags, window_bits, window_first, window_count, &p);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    *handle = g_next_plan++;
    g_plans[*handle] = p;
    return ZK_OK;
}

int zk_msm_plan_clone(uint64_t handle, uint64_t* clone_handle) {
    MsmPlanBase* src = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        auto it = g_plans.find(handle);
        if (it == g_plans.end()) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
        src = it->second;
    }
    MsmPlanBase* p = nullptr;
    int rc = src->clone(&p);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    *clone_handle = g_next_plan++;
    g_plans[*clone_handle] = p;
    return ZK_OK;
}

int zk_msm_plan_destroy(uint64_t handle) {
    MsmPlanBase* p = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        auto it = g_plans.find(handle);
        if (it == g_plans.end()) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
        p = it->second;
        g_plans.erase(it);
    }
    delete p;
    return ZK_OK;
}

static MsmPlanBase* find_plan(uint64_t handle) {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    auto it = g_plans.find(handle);
    return it == g_plans.end() ? nullptr : it->second;
}

static hipStream_t pick_stream(MsmPlanBase* p, void* stream) {
    return stream == ZK_STREAM_PLAN ? p->own_stream : (hipStream_t)stream;
}

int zk_msm_plan_run(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device, int window_first,
                    int window_count, uint64_t* out, void* stream) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->run(n_scalars, scalars, scalars_on_device, window_first, window_count, out, pick_stream(p, stream));
}

int zk_msm_plan_enqueue(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device, int window_first,
                        int window_count, void* stream) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->enqueue(n_scalars, scalars, scalars_on_device, window_first, window_count, pick_stream(p, stream));
}

int zk_msm_plan_enqueue_sort(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device, int window_first,
                             int window_count, void* stream) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->enqueue_sort(n_scalars, scalars, scalars_on_device, window_first, window_count, pick_stream(p, stream));
}

int zk_msm_plan_enqueue_rest(uint64_t handle, uint64_t after_handle) {
    MsmPlanBase* p = find_plan(handle);
    MsmPlanBase* after = after_handle ? find_plan(after_handle) : nullptr;
    if (!p || (after_handle && !after) || p == after) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->enqueue_rest(after);
}

int zk_msm_plan_wait_event(uint64_t handle, void* event) {
    MsmPlanBase* p = find_plan(handle);
    if (!p || !event) return fail(ZK_ERR_ARG, "unknown MSM plan handle or null event");
    ZK_HIP(hipStreamWaitEvent(p->own_stream, (hipEvent_t)event, 0));
    return ZK_OK;
}

int zk_msm_plan_cancel(uint64_t handle) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->cancel();
}

int zk_msm_plan_enqueue_shared(uint64_t handle, uint64_t lender_handle, void* stream) {
    MsmPlanBase* p = find_plan(handle);
    MsmPlanBase* lender = find_plan(lender_handle);
    if (!p || !lender || p == lender) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->enqueue_shared(lender, pick_stream(p, stream));
}

int zk_msm_plan_set_option(uint64_t handle, const char* name, int64_t value) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    if (!name) return fail(ZK_ERR_ARG, "option name is NULL");
    return p->set_option(name, value);
}

int zk_msm_plan_finish(uint64_t handle, uint64_t* out) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    return p->finish(out);
}

int zk_msm_window_layout(int curve, int group, uint64_t n, int flags, int window_bits, int* window_bits_out, int* n_windows) {
    return zk_msm_window_layout_ex(curve, group, n, flags, window_bits, 0, window_bits_out, n_windows);
}

int zk_msm_window_layout_ex(int curve, int group, uint64_t n, int flags, int window_bits, int all_windows, int* window_bits_out,
                            int* n_windows) {
    // the same rule as MsmPlan::init (msm_plan.h: msm_layout): for a rank's share of a window-sharded MSM (all_windows = 0) or
    // for a plan over every window (zk_msm_plan_create)
    if (curve != ZK_CURVE_BN254 && curve != ZK_CURVE_BLS12_381) return fail(ZK_ERR_ARG, "unknown curve");
    if (group != ZK_G1 && group != ZK_G2) return fail(ZK_ERR_ARG, "unknown group");
    const int bits = curve == ZK_CURVE_BN254 ? BnFrParams::BITS : BlsFrParams::BITS;
    const MsmLayout lay = msm_layout(bits, true /* all four groups split their scalars */, n, flags, window_bits, all_windows != 0);
    if (lay.c < 2 || lay.c > 20) return fail(ZK_ERR_ARG, "window bits must be in [2, 20]");
    *window_bits_out = lay.c;
    *n_windows = lay.nwin;
    return ZK_OK;
}

int zk_msm_plan_windows(uint64_t handle, int* window_bits, int* n_windows) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    *window_bits = p->c;
    *n_windows = p->nwin;
    return ZK_OK;
}

int zk_msm_plan_entries(uint64_t handle, uint64_t* entries_per_window) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return fail(ZK_ERR_ARG, "unknown MSM plan handle");
    *entries_per_window = p->entries_per_window;
    return ZK_OK;
}

int zk_msm_plan_timings(uint64_t handle, float* ms, int cap) {
    MsmPlanBase* p = find_plan(handle);
    if (!p) return 0;
    int k = cap < 5 ? cap : 5;
    for (int i = 0; i < k; ++i) ms[i] = p->timings[i];
    return k;
}

int zk_msm(int curve, int group, uint64_t n_points, uint64_t n_scalars, const uint64_t* scalars, const uint64_t* bases,
           uint64_t* out) {
    if (n_points != n_scalars) return fail(ZK_ERR_LENGTH, "Number of points and scalars mismatch");
    int words = zk_point_limbs(curve, group);
    if (words <= 0) return fail(ZK_ERR_ARG, "unknown curve/group");
    if (n_points == 0) { memset(out, 0, (size_t)words * 8); return ZK_OK; }
    uint64_t h = 0;
    int rc = zk_msm_plan_create(curve, group, n_points, bases, 0, 0, 0, &h);
    if (rc) return rc;
    rc = zk_msm_plan_run(h, n_scalars, scalars, 0, 0, 0, out, nullptr);
    zk_msm_plan_destroy(h);
    return rc;
}

int zk_batch_mul(int curve, int group, uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast,
                 uint64_t* out) {
    if (curve == ZK_CURVE_BN254 && group == ZK_G1) return msm_batch_mul_Bn254G1(n, scalars, bases, broadcast, out);
    if (curve == ZK_CURVE_BN254 && group == ZK_G2) return msm_batch_mul_Bn254G2(n, scalars, bases, broadcast, out);
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G1) return msm_batch_mul_Bls381G1(n, scalars, bases, broadcast, out);
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G2) return msm_batch_mul_Bls381G2(n, scalars, bases, broadcast, out);
    return fail(ZK_ERR_ARG, "unknown curve/group");
}

static int points_codec(int curve, int group, uint64_t n, const void* in, void* out, int to_bytes, uint64_t* bad_index) {
    if (bad_index) *bad_index = ~0ull;
    if (curve == ZK_CURVE_BN254 && group == ZK_G1) return msm_points_codec_Bn254G1(n, in, out, to_bytes, bad_index);
    if (curve == ZK_CURVE_BN254 && group == ZK_G2) return msm_points_codec_Bn254G2(n, in, out, to_bytes, bad_index);
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G1) return msm_points_codec_Bls381G1(n, in, out, to_bytes, bad_index);
    if (curve == ZK_CURVE_BLS12_381 && group == ZK_G2) return msm_points_codec_Bls381G2(n, in, out, to_bytes, bad_index);
    return fail(ZK_ERR_ARG, "unknown curve/group");
}

int zk_points_compress(int curve, int group, uint64_t n, const uint64_t* points, uint8_t* out, uint64_t* bad_index) {
    return points_codec(curve, group, n, points, out, 1, bad_index);
}

int zk_points_decompress(int curve, int group, uint64_t n, const uint8_t* in, uint64_t* out, uint64_t* bad_index) {
    return points_codec(curve, group, n, in, out, 0, bad_index);
}

void zk_msm_free_all(void) {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    for (auto& kv : g_plans) delete kv.second;
    g_plans.clear();
    msm_fixed_table_free_Bn254G1();
    msm_fixed_table_free_Bn254G2();
    msm_fixed_table_free_Bls381G1();
    msm_fixed_table_free_Bls381G2();
}

}  // extern "C"
