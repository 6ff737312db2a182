// msm_plan.h -- interface between the C API (msm.hip) and the per-group kernel translation units
// (msm_group.hip compiled once per curve group, so the four instantiations build in parallel).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <hip/hip_runtime.h>
#include "../../include/zkmi.h"

namespace zkmi {

// a device allocation with shared ownership (the bases / fixed-base table of a plan and its clones)
void dev_free_cached(void* p);  // common.hip.h / host.hip
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
    bool pre = false, glv = false;
    int endo = 0;                       // which endomorphism split the scalars (curve * 2 + group; 0 = none): digits of different splits do not mix
    hipEvent_t sorted_ready = nullptr;  // recorded on the lender's stream after its sort stage
    hipEvent_t release = nullptr;       // the borrower records it after its last read; the lender's next run waits for it
};

// ---- window layout of a plan (shared by MsmPlan::init and zk_msm_window_layout) -----------------------------------
constexpr int GLV_BITS = 128;                    // the halves are signed 128-bit values, |k1|, |k2| <= 0.68 * 2^127 (BLS12-381; BN254: < 2^126)
constexpr uint64_t GLV_MAX_POINTS = 1ull << 22;  // 2n entries per window must leave the two-level sort its 8 fine bits
constexpr int MSM_WIDE_C_DEFAULT = 20;           // fixed-base plans over all windows: measured at 2^20 BN254 G1, see DESIGN.md

inline int msm_log2(uint64_t v) {
    int l = 0;
    while (v >>= 1) ++l;
    return l;
}

// Windows of a plan.  The digits kernels add the bias b = sum_w 2^(c-1) 2^(cw) to the scalar (or to a signed half k of it)
// and cut the sum into nwin plain c-bit fields, so they need 0 <= k + b < 2^(nwin c): b >= |k| always holds, and the room
// above b is 2^(nwin c) (2^(c-1) - 1) / (2^c - 1), at least (3/7) 2^(nwin c) for c >= 3: with nwin = ceil(bits / c) that
// covers r - 1 over bits = BITS + 1 for both scalar fields and 0.68 * 2^127 over bits = 128 for the halves (checked for
// every c in 3 .. 20).  For c = 2 the factor is only 1/3 -- 0.667 * 2^127 against BLS12-381 halves of up to 0.673 * 2^127,
// 0.333 * 2^256 against r = 0.453 * 2^256 -- the top digit overflowed and was masked off (round-2 advisor finding), so
// two-bit windows take one more window.
inline int window_count(int bits, int c) {
    return (bits + c - 1) / c + (c == 2 ? 1 : 0);
}
inline int glv_window_count(int c) { return window_count(GLV_BITS, c); }

inline int pick_window_bits(uint64_t entries) {
    // bucket sets must fit the LDS histogram (c <= 16) and stay well filled
    // measured on MI355X (BN254 G1): 2^14 -> 12, 2^16..2^20 -> 16; the tail is latency-bound, so fewer
    // windows win as soon as the buckets are reasonably filled
    int lg = msm_log2(entries < 2 ? 2 : entries);
    int c = lg >= 16 ? 16 : lg - 2;
    if (c < 4) c = 4;
    return c;
}

struct MsmLayout {
    int c = 0, nwin = 0;
    bool glv = false;
    uint64_t entries = 0;  // per window
};

// scalar_bits = bit length of r; has_glv = the group has the (beta x, y) endomorphism wired up (G1);
// all_windows = the plan covers every window (not a rank's share of a window-sharded MSM)
inline MsmLayout msm_layout(int scalar_bits, bool has_glv, uint64_t n_points, int flags, int window_bits, bool all_windows) {
    MsmLayout L;
    const bool pre = (flags & ZK_MSM_PRECOMPUTE) != 0;
    const bool no_glv = getenv("ZKMI_NO_GLV") != nullptr;   // read per call: the layout of a plan is fixed when it is created
    L.glv = has_glv && !pre && !(flags & ZK_MSM_NO_GLV) && !no_glv && n_points <= GLV_MAX_POINTS;
    L.entries = L.glv ? 2 * n_points : n_points;
    int c = window_bits > 0 ? window_bits : pick_window_bits(L.entries);
    if (window_bits <= 0 && pre && all_windows) {
        // fixed-base plans over all windows: one shared bucket set makes wider windows affordable (13 windows of
        // 20 bits instead of 16 of 16: 19 % fewer additions) as long as a table reference fits the 31 bits below the sign of
        // a sort entry.  Window-sharded plans keep 16 windows, which split evenly over 2, 4 or 8 ranks.
        const int pre_c = getenv("ZKMI_PRE_C") ? atoi(getenv("ZKMI_PRE_C")) : 0;
        const bool no_two_level = getenv("ZKMI_NO_TWO_LEVEL") != nullptr;  // wide windows exist in the two-level sort only
        // Candidates whose TOP window is at least half full: all windows feed one bucket set, and a short top window
        // (18 bits: 3 scalar bits left for it) would pile its n entries into a handful of buckets of one coarse bin.
        const int cands[3] = {pre_c ? pre_c : MSM_WIDE_C_DEFAULT, 17, 0};
        for (int k = 0; !no_two_level && cands[k] > 16; ++k) {
            const int cand = cands[k];
            const uint64_t w = (uint64_t)(scalar_bits + 1 + cand - 1) / cand;
            const int top_bits = scalar_bits + 1 - (int)(w - 1) * cand;
            // below 2^20 points the wider bucket set costs more in the (latency-bound) reduction than the windows save
            if (n_points >= (1ull << 20) && w < 16 && (2 * top_bits >= cand || pre_c) && w * n_points <= 0x7FFFFFFFull) { c = cand; break; }
            if (pre_c) break;
        }
    }
    L.c = c;
    L.nwin = c > 0 ? (L.glv ? glv_window_count(c) : window_count(scalar_bits + 1, c)) : 0;
    return L;
}

// ---- per-plan tuning options ----------------------------------------------------------------------------------------
// Defaults come from the ZKMI_* environment variables READ WHEN A PLAN IS CREATED (no process-wide latches: two plans of one
// process may differ, and a test can flip a knob in-process); the run-time ones can also be changed on a live plan through
// zk_msm_plan_set_option.  Layout options decide buffer sizes and the window layout and are fixed once the plan exists.
struct MsmOptions {
    // run-time (settable on a live plan)
    uint64_t segment_lanes = 256ull * 1024;  // ZKMI_SEG_LANES: lanes the accumulate kernel aims at (4 waves per SIMD on 256 CUs)
    bool sum_one_step = false;               // ZKMI_SUM_ONE_STEP: row / column sums in one launch whatever the bucket count
    uint32_t lanes_per_output = 0;           // ZKMI_LPO: lanes per row / column sum of the one-step form (0 = by output count)
    bool two_level_sort = true;              // ZKMI_NO_TWO_LEVEL clears it: one-level / bucket-range sorts only (c <= 16)
    bool priority_steps = true;              // ZKMI_NO_PRIO_STEPS clears it: accumulate waves step their issue priority down near the
                                             // end of their segment (plain runs only, msm_accumulate.hip.h)
    int split_pairs = -1;                    // ZKMI_SPLIT_PAIRS=0/1: the G2 accumulate kernel with a lane PAIR per segment and every Fp2 value
                                             // split by component (fp2_split.hip.h); -1 = the group's default (BLS12-381 G2 on, BN254 G2 off)
    // layout (creation time only)
    uint32_t sort_workgroups = 256;          // ZKMI_SORT_WGS: bucket-range sort workgroups over all windows
    int fine_log = 0;                        // ZKMI_FINE_LOG: fine bucket bits of the two-level sort (0 = automatic)
    bool trace_init = false;                 // ZKMI_TRACE_INIT: time the steps of plan creation on stderr

    static MsmOptions from_env() {
        MsmOptions o;
        if (const char* e = getenv("ZKMI_SEG_LANES")) o.segment_lanes = (uint64_t)atoll(e);
        o.sum_one_step = getenv("ZKMI_SUM_ONE_STEP") != nullptr;
        if (const char* e = getenv("ZKMI_LPO")) o.lanes_per_output = (uint32_t)atoi(e);
        o.two_level_sort = getenv("ZKMI_NO_TWO_LEVEL") == nullptr;
        o.priority_steps = getenv("ZKMI_NO_PRIO_STEPS") == nullptr;
        if (const char* e = getenv("ZKMI_SPLIT_PAIRS")) o.split_pairs = atoi(e) != 0 ? 1 : 0;
        if (const char* e = getenv("ZKMI_SORT_WGS")) o.sort_workgroups = (uint32_t)atoi(e);
        if (const char* e = getenv("ZKMI_FINE_LOG")) o.fine_log = atoi(e);
        o.trace_init = getenv("ZKMI_TRACE_INIT") != nullptr;
        return o;
    }
};

struct MsmPlanBase {
    virtual ~MsmPlanBase() {}
    MsmOptions opt = MsmOptions::from_env();
    // name: "segment_lanes", "sum_one_step", "lanes_per_output", "two_level_sort", "priority_steps", "split_pairs"; ZK_ERR_ARG for anything else, for a value
    // the plan cannot honour (two-level sort without its buffers, one-level sort for windows wider than 16 bits) and while a
    // run is in flight
    virtual int set_option(const char* name, int64_t value) = 0;
    virtual int export_sort(SortExport* out) = 0;                        // of the run in flight
    virtual int enqueue_shared(MsmPlanBase* lender, hipStream_t stream) = 0;
    // a second plan over the same bases (shared, read-only) with its own workspace and stream: two MSMs against one key
    // in flight together (tau_1 with u and with v in Groth16.prove) without building the fixed-base table twice
    virtual int clone(MsmPlanBase** out) = 0;
    // enqueue() puts every GPU stage plus the D2H copy of the per-window results on `stream` and returns;
    // finish() waits for them and runs the host tail.  run() = enqueue() + finish().
    virtual int enqueue(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count,
                        hipStream_t stream) = 0;
    // the same in two steps, for a caller that orders the accumulate kernels of several plans: enqueue_sort puts the
    // digits and the sort on the stream; enqueue_rest the accumulate kernel (after `after`'s, if given), the reduction and
    // the D2H copy.  Left to themselves the accumulate kernels of concurrent plans share the machine and the sorts of
    // later plans starve behind them (a resident accumulate grid holds every wave slot until it ends).
    virtual int enqueue_sort(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t stream) = 0;
    virtual int enqueue_rest(MsmPlanBase* after) = 0;
    virtual hipEvent_t accumulate_done_event() = 0;
    virtual int finish(uint64_t* out) = 0;
    // forget a run whose sort (or all of it) is in flight without collecting a result: waits for the stream
    virtual int cancel() = 0;
    int run(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, uint64_t* out,
            hipStream_t stream) {
        int rc = enqueue(n_scalars, scalars, on_device, w_first, w_count, stream);
        return rc ? rc : finish(out);
    }
    hipStream_t own_stream = nullptr;  // used when the caller passes ZK_STREAM_PLAN
    int c = 0, nwin = 0;
    uint64_t entries_per_window = 0;  // points of the plan, doubled when it runs on (P, phi(P)) pairs
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
