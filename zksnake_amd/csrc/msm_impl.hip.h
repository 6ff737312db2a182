// msm_impl.hip.h -- Pippenger bucket multi-scalar multiplication over G1/G2 of BN254 and BLS12-381 on
// gfx950, plus batched scalar multiplication.
//
// Stands in for multiscalar_mul_g1/_g2 (reference src/bn254/curve.rs:356-392,
// src/bls12_381/curve.rs:366-402 -> ark-ec 0.4.2 VariableBaseMSM::msm) and
// batch_multi_scalar_g1/_g2 (src/bn254/curve.rs:326-354).  The result of an MSM is a single
// group element, returned as its unique affine representative, so any correct bucket method
// is bit-identical with the reference's.
//
// Pipeline (one stream, no host round trip until the tail):
//   1. digits      scalar -> signed c-bit window digits (bias trick: s + sum 2^(c-1) 2^(wc),
//                  then plain bit fields), 2 B per (window, scalar), coalesced.
//   2. histogram   one workgroup per (window, chunk): 2^(c-1) counters live in LDS (128 KiB at
//                  c = 16), LDS atomics only; the table is flushed with coalesced stores.
//   3. prefix/scan per-bucket prefix over chunks, bucket offsets, segment offsets.
//   4. scatter     same grid as 2: LDS cursors, point references written bucket-sorted.
//   5. accumulate  THE dominant kernel: one lane per bucket *segment* (<= S entries, so skewed
//                  scalar distributions cannot starve a wave), XYZZ accumulator in registers,
//                  bases gathered as whole 64..192-byte rows.
//   6. combine     per-bucket sum of its segment partials.
//   7. reduce      sum_b (b+1) B_b through the split b = hi*C + lo: wave-per-row / wave-per-column
//                  sums (shuffle trees), then a log-depth suffix-scan on <=256 points per array.
//   8. host tail   3 points per window come back; Horner over windows and the affine
//                  conversion (one inversion) run on the host -- a lone GPU wave needs ~1.3 us per
//                  field multiplication, the host ~50 ns.
#pragma once
#include <algorithm>
#include <chrono>
#include <memory>
#include <vector>
#include "common.hip.h"
#include "msm_plan.h"
#include "pair.hip.h"
#include "setup_impl.hip.h"
#include "codec.hip.h"
#include "glv_params.h"

#include "msm_common.hip.h"
#if !defined(ZK_PART) || ZK_PART == 0  // sort-stage kernels and the plan live in part 0 only
#include "msm_sort.hip.h"
#endif
#include "msm_accumulate.hip.h"
#include "msm_reduce.hip.h"

namespace zkmi {

#if defined(ZK_GROUP) && (!defined(ZK_PART) || ZK_PART == 0)
// the plan's translation unit does not instantiate the heavy kernels (see msm_group.hip)
extern template __global__ void accumulate_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*);
extern template __global__ void accumulate_split_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*);   // instantiated for the Fp2 groups (msm_group.hip)
extern template __global__ void bases_to_mont_kernel<ZK_GROUP>(const uint32_t*, uint64_t, uint32_t*, int);
extern template __global__ void combine_kernel<ZK_GROUP>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*);
extern template __global__ void strided_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t*, SumJob, SumJob, uint32_t);
extern template __global__ void weighted_sum_kernel<ZK_GROUP>(const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t*);
ZK_SETUP_EXTERN_TEMPLATES(ZK_GROUP)
ZK_CODEC_EXTERN_TEMPLATES(ZK_GROUP)
#endif

#if !defined(ZK_PART) || ZK_PART == 0
// ---- host: plan --------------------------------------------------------------------------------------

// ZK_MSM_PRECOMPUTE: table row k = 2^(c (w_first + k)) * P_i (affine, Montgomery) for the w_count windows of the plan; on
// entry row 0 holds the bases themselves.  With these rows every window adds into ONE shared bucket set: the bucket
// reduction and the host tail shrink by the number of windows and the Horner pass disappears.  The chain of doublings
// runs in XYZZ on a scratch vector; each row is normalised with the batched inversion (one Fermat inversion per 1024
// points instead of one per point and row: 46 -> ~15 ms for a 2^20-point BN254 G1 key).
template <class G>
static int precompute_table(uint32_t* table, uint64_t n, int c, int w_first, int w_count) {
    typedef typename G::F F;
    constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
    if (w_first == 0 && w_count == 1) return ZK_OK;
    uint32_t* temp = nullptr;
    ZK_ALLOC(&temp, n * XW * 4);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const unsigned nblocks = (unsigned)((n + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E));
    hipLaunchKernelGGL(dbl_rows_kernel<G>, dim3(blocks), dim3(256), 0, 0, temp, n, c * w_first, (const uint32_t*)table);
    if (w_first > 0) hipLaunchKernelGGL(normalize_kernel<G>, dim3(nblocks), dim3(NORM_THREADS), 0, 0, temp, n, table, 0);
    for (int k = 1; k < w_count; ++k) {
        hipLaunchKernelGGL(dbl_rows_kernel<G>, dim3(blocks), dim3(256), 0, 0, temp, n, c, (const uint32_t*)nullptr);
        hipLaunchKernelGGL(normalize_kernel<G>, dim3(nblocks), dim3(NORM_THREADS), 0, 0, temp, n, table + (size_t)k * n * AW, 0);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    dev_free_cached(temp);
    ZK_HIP(e);
    return ZK_OK;
}

// fixed-base table of the last broadcast base of this group (Groth16.setup multiplies the same generator five times)
template <class G>
struct FixedTable {
    std::vector<uint64_t> base;   // canonical affine limbs the table was built for
    uint32_t* d_table = nullptr;
    int nwin = 0;
    std::mutex mu;
    static FixedTable& get() {
        static FixedTable t;
        return t;
    }
    void release() {
        std::lock_guard<std::mutex> lock(mu);
        if (d_table) dev_free_cached(d_table);
        d_table = nullptr;
        base.clear();
    }
    // caller holds mu
    int ensure(const uint64_t* base_limbs) {
        typedef typename G::F F;
        typedef typename G::Fr FrP;
        constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
        const size_t words64 = AW / 2;
        if (d_table && base.size() == words64 && memcmp(base.data(), base_limbs, words64 * 8) == 0) return ZK_OK;
        if (d_table) dev_free_cached(d_table);
        d_table = nullptr;
        base.clear();
        nwin = (FrP::BITS + 1 + FIXED_C - 1) / FIXED_C;
        // window bases 2^(16 j) G on the host (16 points)
        std::vector<uint32_t> wb((size_t)nwin * AW);
        {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(base_limbs);
            Affine<F> g = {F::from_canonical(w), F::from_canonical(w + F::LIMBS)};
            XYZZ<F> acc = xyzz_from_affine<F>(g);
            for (int j = 0; j < nwin; ++j) {
                Affine<F> a = xyzz_to_affine<F>(acc);
                F::store(wb.data() + (size_t)j * AW, a.x);
                F::store(wb.data() + (size_t)j * AW + F::LIMBS, a.y);
                if (j + 1 < nwin) for (int k = 0; k < FIXED_C; ++k) acc = xyzz_dbl<F>(acc);
            }
        }
        const uint64_t rows = (uint64_t)nwin * FIXED_HALF;
        uint32_t *d_wb = nullptr, *temp = nullptr;
        ZK_ALLOC(&d_table, rows * AW * 4);
        hipError_t e = dev_alloc_cached((void**)&d_wb, wb.size() * 4) == ZK_OK ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = dev_alloc_cached((void**)&temp, rows * XW * 4) == ZK_OK ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpy(d_wb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(fixed_table_kernel<G>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, d_wb, nwin, temp);
            hipLaunchKernelGGL(normalize_kernel<G>, dim3((unsigned)((rows + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E))),
                               dim3(NORM_THREADS), 0, 0, temp, rows, d_table, 0);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (d_wb) dev_free_cached(d_wb);
        if (temp) dev_free_cached(temp);
        if (e != hipSuccess) {
            dev_free_cached(d_table);
            d_table = nullptr;
            return fail(ZK_ERR_HIP, std::string("fixed-base table: ") + hipGetErrorString(e));
        }
        base.assign(base_limbs, base_limbs + words64);
        return ZK_OK;
    }
};

template <class G>
struct MsmPlan : MsmPlanBase {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    static constexpr int AW = 2 * F::LIMBS;
    static constexpr int XW = 4 * F::LIMBS;
    static constexpr uint64_t SEG_TARGET_LANES = 256ull * 1024;  // 4 waves per SIMD on 256 CUs
    static constexpr int MAX_C = 20;           // widest window (fixed-base plans; digits are then 32-bit)
    bool wide = false;                         // c > 16: 32-bit digits, two-level sort only

    // device workspace of one run (stages 2..7)
    struct Work {
        uint32_t *hist = nullptr, *total = nullptr, *bstart = nullptr, *sstart = nullptr;
        uint32_t *bsums = nullptr, *grand = nullptr, *big_list = nullptr, *big_count = nullptr;
        uint32_t *sorted = nullptr, *partials = nullptr, *buckets = nullptr, *rows = nullptr, *fin = nullptr;
        uint32_t* parts = nullptr;  // partial row / column sums of the two-step strided sums
        uint32_t *tmp_ref = nullptr, *bin_start = nullptr, *slice_sums = nullptr, *bin_tot = nullptr;  // two-level sort
        uint8_t* tmp_fine = nullptr;  // fine bucket bits of the level-A entries when the reference needs all 31 bits
        // A recorded event costs ~3 us of idle GPU between two kernels (tools/event_gap_probe.hip), so a run records only the four
        // that order work or bound a stage: ev_start (plan), ev_acc0 = sorted (also the lender's "sorted_ready"), ev_acc1 =
        // accumulated (the gate of the next plan's accumulate kernel), ev_end (plan); ev_accs only when the accumulate kernel
        // waits for something after the sort (a gate, or the second half of a two-step enqueue)
        hipEvent_t ev_acc0 = nullptr, ev_accs = nullptr, ev_acc1 = nullptr;
        bool acc_from_accs = false;
        hipEvent_t ev_release = nullptr;  // recorded by a borrower of this run's sort (enqueue_shared) after its last read
        bool lent = false;
        uint32_t seg_len = 0;
        int w_first = 0, w_count = 0;  // windows of the run in flight
        uint32_t groups = 0;           // bucket sets of the run in flight
    };

    uint64_t n = 0;      // entries per window (2 n_api with the endomorphism)
    uint64_t n_api = 0;  // points of the plan as the caller counts them
    bool glv = false;    // general G1 plan over (P_i, phi(P_i)) with half-length scalars
    bool pre = false;  // ZK_MSM_PRECOMPUTE: shared bucket set over a table of 2^(c w) P_i
    int pw_first = 0, pw_count = 0;  // windows this plan can run (a sharded rank's share; all of them by default)
    uint32_t B = 0, R = 0, C = 0;
    uint32_t bpr = 0, bpc = 0;       // weighted-sum blocks per row array / per column array
    int range_log = 0;  // general mode: log2(buckets per sort workgroup)
    Work ws;
    // shared device buffers
    std::shared_ptr<DeviceBlock> bases_block;  // the (table of) bases: shared by the clones of a plan
    uint32_t* d_bases = nullptr;
    uint32_t* d_scalars = nullptr;
    void* d_dig = nullptr;  // windows x (n + 8) digits, uint16_t (c <= 16) or uint32_t
    uint32_t* h_final = nullptr;  // pinned: (S, T) per weighted-sum block
    hipEvent_t ev_start = nullptr, ev_end = nullptr;

    ~MsmPlan() override {
        // blocks go back to the caching allocator, which (unlike hipFree) does not wait for the device: make sure no run of
        // this plan is still in flight
        (void)hipDeviceSynchronize();
        void* bufs[] = {ws.hist, ws.total, ws.bstart, ws.sstart, ws.bsums, ws.grand, ws.big_list, ws.big_count,
                        ws.sorted, ws.partials, ws.buckets, ws.rows, ws.parts, ws.fin, ws.tmp_ref, ws.tmp_fine, ws.bin_start, ws.slice_sums, ws.bin_tot,
                        d_scalars, d_dig};
        for (void* q : bufs) dev_free_cached(q);
        pinned_free_cached(h_final);
        for (hipEvent_t e : {ws.ev_acc0, ws.ev_accs, ws.ev_acc1, ws.ev_release, ev_start, ev_end}) if (e) (void)hipEventDestroy(e);
        stream_release((create_flags & ZK_MSM_HIGH_PRIORITY) != 0, own_stream);
    }

    // Every allocation lands in a member that the destructor frees, and the factory deletes the plan when init() fails
    // (msm_group.hip), so a failing hipMalloc half way through leaks nothing.
    // `share` != nullptr: a clone -- same bases (and fixed-base table), own workspace and stream
    int init(uint64_t n_points, const void* bases, int bases_on_device, int flags, int window_bits, int win_first, int win_count,
             const MsmPlan* share = nullptr) {
        const bool trace = opt.trace_init;
        auto t_prev = std::chrono::steady_clock::now();
        auto mark = [&](const char* what) {
            if (!trace) return;
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[plan init] %-24s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
            t_prev = now;
        };
        create_flags = flags;
        seg_lanes_at_init = opt.segment_lanes ? opt.segment_lanes : SEG_TARGET_LANES;
        pre = (flags & ZK_MSM_PRECOMPUTE) != 0;
        if (n_points == 0 || n_points > (1ull << 26)) return fail(ZK_ERR_ARG, "MSM size must be in [1, 2^26]");
        const MsmLayout lay = msm_layout(FrP::BITS, GlvOf<G>::OK, n_points, flags, window_bits, win_count <= 0);
        glv = share ? share->glv : lay.glv;
        n_api = n_points;
        n = glv ? 2 * n_points : n_points;  // entries per window: the kernels see an MSM over (P_i, phi(P_i)) pairs
        entries_per_window = n;
        c = lay.c;
        if (c < 2 || c > MAX_C) return fail(ZK_ERR_ARG, "window bits must be in [2, 20]");
        if (c > 16 && !pre) return fail(ZK_ERR_ARG, "windows wider than 16 bits need a fixed-base plan (ZK_MSM_PRECOMPUTE)");
        wide = c > 16;
        nwin = glv ? glv_window_count(c) : window_count(FrP::BITS + 1, c);
        if (win_count <= 0) { win_first = 0; win_count = nwin; }
        if (win_first < 0 || win_first + win_count > nwin) return fail(ZK_ERR_ARG, "window range out of bounds");
        pw_first = win_first;
        pw_count = win_count;
        B = 1u << (c - 1);
        int rl = (c - 1 + 1) / 2;
        if (rl > 8 && !wide) rl = 8;
        R = 1u << rl;
        C = B / R;
        if (C > 1024 || R > 1024) return fail(ZK_ERR_ARG, "window too wide for the reduction stage");
        bpr = (R + WS_BLOCK - 1) / WS_BLOCK;
        bpc = (C + WS_BLOCK - 1) / WS_BLOCK;
        // bucket ranges (general mode, small inputs): about 256 sort workgroups in total, at least 64 buckets each
        {
            const uint32_t wgs = opt.sort_workgroups;
            uint32_t want = std::max<uint32_t>(1u, wgs / (uint32_t)std::max(1, pw_count));
            uint32_t per = std::max<uint32_t>(64u, B / want);
            if (per > B) per = B;
            range_log = log2_u64(per);
            if ((1u << range_log) > B) range_log = c - 1;
        }
        const uint64_t entries = (uint64_t)pw_count * n;
        if (entries > 0x7FFFFFFFull) return fail(ZK_ERR_ARG, "MSM too large");

        ZK_HIP_RC(stream_acquire((flags & ZK_MSM_HIGH_PRIORITY) != 0, &own_stream));
        if (share) {
            bases_block = share->bases_block;
            d_bases = share->d_bases;
        } else {
            bases_block = std::make_shared<DeviceBlock>();
            ZK_ALLOC(&bases_block->ptr, (pre ? (uint64_t)pw_count : 1ull) * n * AW * 4);
            d_bases = (uint32_t*)bases_block->ptr;
            if (bases_on_device) {
                hipLaunchKernelGGL(bases_to_mont_kernel<G>, dim3((unsigned)((n_api + 127) / 128)), dim3(128), 0, 0,
                                   (const uint32_t*)bases, n_api, d_bases, glv ? 1 : 0);
            } else {
                uint32_t* tmp = nullptr;
                ZK_ALLOC(&tmp, n_api * AW * 4);
                hipError_t e = hipMemcpy(tmp, bases, n_api * AW * 4, hipMemcpyHostToDevice);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(bases_to_mont_kernel<G>, dim3((unsigned)((n_api + 127) / 128)), dim3(128), 0, 0, tmp, n_api, d_bases, glv ? 1 : 0);
                    e = hipDeviceSynchronize();
                }
                dev_free_cached(tmp);
                ZK_HIP(e);
            }
            ZK_HIP(hipGetLastError());
            if (pre) {
                // rows 2^(c w) P_i for the windows of this plan only (a sharded rank never builds the other ranks' rows)
                int rc = precompute_table<G>(d_bases, n, c, pw_first, pw_count);
                if (rc) return rc;
            }
        }
        mark("stream + bases");
        ZK_ALLOC(&d_scalars, n_api * FrP::W * 4);
        ZK_ALLOC(&d_dig, (size_t)pw_count * (n + 8) * (wide ? 4 : 2));
        if (wide && !two_level_ok()) return fail(ZK_ERR_ARG, "this size does not fit the two-level sort that wide windows need");
        const uint64_t max_sets = pre ? 1ull : (uint64_t)pw_count;
        ZK_HIP_RC(pinned_alloc_cached((void**)&h_final, (size_t)max_sets * (bpr + bpc) * 2 * XW * 4));
        mark("scalars/digits/pinned");
        ZK_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableSystemFence));   // timing only: nothing synchronises with it
        ZK_HIP(hipEventCreate(&ev_end));
        {
            const uint64_t keys = max_sets * B;
            // a window-range run picks its own (shorter) segments: at most SEG_TARGET_LANES of them, or entries / 8
            const uint32_t seg_full = pick_seg_len(entries);
            const uint64_t max_segs = std::max<uint64_t>(entries / seg_full, std::min<uint64_t>(entries / 8, seg_lanes_at_init)) + keys + 8;
            // windows x chunks <= max(256, windows) sub-histograms: of all B buckets (one-level sort) or of the coarse bins only
            ZK_ALLOC(&ws.hist, (size_t)std::max<uint64_t>(256, pw_count) * (wide ? (B >> fine_log_for(n)) : B) * 4);
            ZK_ALLOC(&ws.total, keys * 4);
            ZK_ALLOC(&ws.bstart, (keys + 1) * 4);
            ZK_ALLOC(&ws.sstart, (keys + 1) * 4);
            ZK_ALLOC(&ws.bsums, ((keys + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * 4);
            ZK_ALLOC(&ws.grand, 4);
            ZK_ALLOC(&ws.big_list, keys * 4);
            ZK_ALLOC(&ws.big_count, 8);
            ZK_ALLOC(&ws.sorted, entries * 4);
            if (two_level_ok()) {
                ZK_ALLOC(&ws.tmp_ref, entries * 4);
                if (split_fine()) ZK_ALLOC(&ws.tmp_fine, entries);
                ZK_ALLOC(&ws.bin_start, (max_sets * (B >> fine_log_for(n)) + 1) * 4);
                ZK_ALLOC(&ws.slice_sums, 4096 * BINS_SLICES * 4);
                ZK_ALLOC(&ws.bin_tot, 4096 * 4);
            }
            ZK_ALLOC(&ws.partials, max_segs * XW * 4);
            ZK_ALLOC(&ws.buckets, keys * XW * 4);
            ZK_ALLOC(&ws.rows, max_sets * (R + C) * XW * 4);
            if (sum_part_len() >= 2) ZK_ALLOC(&ws.parts, max_sets * 2 * (uint64_t)(B / sum_part_len()) * XW * 4);
            ZK_ALLOC(&ws.fin, max_sets * (bpr + bpc) * 2 * XW * 4);
            for (hipEvent_t* e : {&ws.ev_acc0, &ws.ev_accs, &ws.ev_acc1, &ws.ev_release}) ZK_HIP(hipEventCreate(e));
        }
        mark("workspace + events");
        // LDS above 64 KiB needs the opt-in
        int lds_bytes = (int)((wide ? (1u << 15) : B) * 4);  // the one-level kernels never run for wide windows
        ZK_HIP(hipFuncSetAttribute((const void*)hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)hist_range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_hi_staged_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)scatter_hi_staged_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)sort_lo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)weighted_sum_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(HalfRegs<F>::COUNT * HS_THREADS * 4)));
        mark("func attributes");
        ZK_HIP(hipDeviceSynchronize());
        mark("device sync");
        return ZK_OK;
    }

    // chunked sort: windows x chunks workgroups of 1024 threads, ONE per CU (the LDS histogram takes 128 KiB at
    // c = 16), so their number is kept at or just below the 256 CUs: 272 workgroups would run as 256 + 16,
    // i.e. take twice as long
    static int chunks_for(int windows, uint64_t count) {
        int k = 256 / std::max(1, windows);
        if (k < 1) k = 1;
        uint64_t cap = (count + 4095) / 4096;  // at least 4096 entries per chunk
        if ((uint64_t)k > cap) k = (int)std::max<uint64_t>(1, cap);
        return k;
    }

    int exclusive_scan(const uint32_t* in, uint32_t cnt, uint32_t* out, hipStream_t st) {
        uint32_t blocks = (cnt + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(scan_block_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, in, cnt, out, ws.bsums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, ws.bsums, blocks, ws.grand);
        hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, out, cnt, ws.bsums, ws.grand);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }

    // fine bucket bits of the two-level sort for n points: the largest of 8, 7 that leaves room for the index in a 32-bit
    // entry, with at least four coarse bins per window and at most 4096 (window, bin) pairs (one-workgroup scan); 0 = n/a
    int fine_log_for(uint64_t points) const {
        // fixed-base mode: ONE bucket set over references into the (window, point) table, so the references are wider
        // and the coarse bins are shared by all windows -- more, smaller bins keep level B parallel
        const uint64_t refs = pre ? (uint64_t)pw_count * points : points;
        const uint64_t sets = pre ? 1 : (uint64_t)pw_count;
        if (wide) {
            const int f = c - 13;  // 4096 coarse bins; the fine bits move out of the entry when the reference needs the room
            return refs <= 0x7FFFFFFFull ? f : 0;
        }
        const int f_env = opt.fine_log;  // tuning knob (general mode)
        if (f_env && !pre && refs <= (1ull << (31 - f_env)) && c - 1 >= f_env + 2 && sets * (B >> f_env) <= 4096) return f_env;
        const int f_hi = pre ? 5 : FINE_LOG_MAX, f_lo = pre ? 4 : FINE_LOG_MAX - 1;
        // general mode: the widest bins that still hold about 8192 entries each (one level-B workgroup sorts a bin in LDS;
        // 2^21 split-scalar entries per window: 7 fine bits, 0.175 ms for digits + sort against 0.20 with 8)
        int best = 0;
        for (int f = f_hi; f >= f_lo; --f) {
            if (!(refs <= (1ull << (31 - f)) && c - 1 >= f + 2 && sets * (B >> f) <= 4096)) continue;
            if (pre || (points >> (c - 1 - f)) <= 8192) return f;
            best = f;
        }
        return best;
    }

    // level-A entries carry (sign, fine bucket bits, reference) in 32 bits while that fits; wide windows over a big table
    // (13 x n rows, n > 2^20) keep the fine bits in a byte array beside them
    bool split_fine() const {
        return wide && pre && (uint64_t)pw_count * n > (1ull << (31 - (c - 13)));
    }

    // buckets one lane pair adds up in the first step of the two-step strided sums (0 = one step)
    uint32_t sum_part_len() const {
        const uint32_t k = 16u;
        return (R >= 4 * k && C >= 4 * k) ? k : 0u;
    }

    bool two_level_ok() const {
        return opt.two_level_sort && fine_log_for(n) > 0;
    }

    // Segment length for a run over `entries` sorted entries: aim at >= 4 waves per SIMD worth of lanes (a window-range
    // run of a sharded MSM has far fewer entries than the plan's full set; with the plan-wide length its lanes would
    // be too few and each would walk 64 additions at lone-wave speed).
    uint32_t pick_seg_len(uint64_t entries) const {
        const uint64_t target = opt.segment_lanes ? opt.segment_lanes : SEG_TARGET_LANES;
        uint64_t sl = (entries + target - 1) / target;
        if (sl < 8) sl = 8;
        if (sl > 64) sl = 64;
        if (pre) {
            // shared bucket set: keep a bucket within ~12 runs so that one lane pair can combine it
            uint64_t per_bucket = entries / B;
            uint64_t want = (per_bucket + 11) / 12;
            if (want > sl) sl = want;
            if (sl > 1024) sl = 1024;
        }
        return (uint32_t)sl;
    }

    // ---- the stages of one run, for the windows [ws.w_first, ws.w_first + ws.w_count) on stream st -------------------------

    // stages 2-4: histogram, scans, scatter -> ws.sorted / ws.bstart / ws.sstart (+ the lists of buckets with many runs)
    int stage_sort(uint32_t m, uint32_t dstride, uint32_t seg_len, hipStream_t st) {
        Work& l = ws;
        const int w_first = l.w_first, w_count = l.w_count;
        const uint32_t n_keys = l.groups * B;
        const int nchunk = chunks_for(w_count, m);  // this run's windows fill the chip
        const uint32_t ch_len = (m + nchunk - 1) / nchunk;
        // digit rows are stored relative to the plan's first window; the kernels index them with absolute windows
        const uintptr_t dig_base = reinterpret_cast<uintptr_t>(this->d_dig) - (uintptr_t)pw_first * dstride * (wide ? 4 : 2);
        const uint16_t* d_dig = reinterpret_cast<const uint16_t*>(dig_base);
        const uint32_t* d_dig32 = reinterpret_cast<const uint32_t*>(dig_base);
        // general mode, small inputs: bucket-range partition (measured faster up to 2^18); otherwise the two-level sort
        const bool ranged = !pre && !wide && m < (1u << 19);
        const bool two_level = !ranged && l.tmp_ref != nullptr && opt.two_level_sort;
        if (two_level) {
            const int fl = fine_log_for(n);
            const uint32_t NB = B >> fl;
            const uint32_t ch8 = (ch_len + 7) & ~7u;  // the kernels read eight digits per load
            if (wide) hipLaunchKernelGGL(hist_hi_kernel<uint32_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), NB * 4, st, d_dig32, m, dstride, c, w_first, nchunk, ch8, fl, l.hist);
            else hipLaunchKernelGGL(hist_hi_kernel<uint16_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), NB * 4, st, d_dig, m, dstride, c, w_first, nchunk, ch8, fl, l.hist);
            // fixed-base mode: one bucket set fed by all (window, chunk) sub-histograms; general mode: one set per window
            const int sets = pre ? 1 : w_count, subs = pre ? w_count * nchunk : nchunk;
            const uint32_t pairs = (uint32_t)sets * NB;
            if ((uint64_t)pairs * subs >= (1u << 17)) {
                const unsigned bb = (pairs + 63) / 64;
                hipLaunchKernelGGL(bins_partial_kernel, dim3(bb), dim3(1024), 0, st, l.hist, subs, NB, pairs, l.slice_sums, l.bin_tot);
                hipLaunchKernelGGL(bins_scan_tot_kernel, dim3(1), dim3(1024), 0, st, l.bin_tot, pairs, l.bin_start, l.bstart + n_keys);
                hipLaunchKernelGGL(bins_prefix_kernel, dim3(bb), dim3(1024), 0, st, l.hist, subs, NB, pairs, l.slice_sums, l.bin_start);
            } else {
                hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, st, l.hist, sets, subs, NB, l.bin_start, l.bstart + n_keys);
            }
            {
                const uint32_t NBP = (NB + 127) & ~127u;
                const size_t lds_a = (size_t)SCATTER_TILE * 4 + (size_t)NBP * 12 + (size_t)SCATTER_TILE * 2 + (l.tmp_fine ? SCATTER_TILE : 0);
                if (wide) hipLaunchKernelGGL(scatter_hi_staged_kernel<uint32_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), lds_a, st, d_dig32, m, dstride, c, w_first, nchunk, ch8, fl, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.tmp_ref, l.tmp_fine);
                else hipLaunchKernelGGL(scatter_hi_staged_kernel<uint16_t>, dim3(w_count * nchunk), dim3(SORT_THREADS), lds_a, st, d_dig, m, dstride, c, w_first, nchunk, ch8, fl, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.tmp_ref, (uint8_t*)nullptr);
            }
            // LDS stage of level B: 1.5x the expected entries of a coarse bin, capped at 96 KiB
            uint64_t expect = ((uint64_t)w_count * m) / ((uint64_t)sets * NB);
            uint32_t stage_cap = (uint32_t)std::min<uint64_t>(24576, std::max<uint64_t>(2048, expect + expect / 2));
            hipLaunchKernelGGL(sort_lo_kernel, dim3(sets * NB), dim3(SORT_LO_THREADS), (size_t)stage_cap * 4, st, l.bin_start, l.tmp_ref, (const uint8_t*)l.tmp_fine, B, fl, stage_cap, l.bstart, l.sorted);
        } else if (ranged) {
            hipLaunchKernelGGL(hist_range_kernel, dim3(w_count * (B >> range_log)), dim3(SORT_THREADS), (4u << range_log), st, d_dig, m, dstride, c, w_first, range_log, l.total);
        } else {
            hipLaunchKernelGGL(hist_kernel, dim3(w_count * nchunk), dim3(SORT_THREADS), B * 4, st, d_dig, m, dstride, c, w_first, nchunk, ch_len, l.hist);
            hipLaunchKernelGGL(prefix_kernel, dim3((n_keys + 255) / 256), dim3(256), 0, st, l.hist, pre ? w_count * nchunk : nchunk, B, n_keys, l.total);
        }
        int rc;
        if (!two_level && (rc = exclusive_scan(l.total, n_keys, l.bstart, st))) return rc;
        {
            // run offsets: run counts computed on the fly + three-launch scan
            const uint32_t blocks = (n_keys + SCAN_BLOCK - 1) / SCAN_BLOCK;
            hipLaunchKernelGGL(runs_scan_block_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, l.bstart, n_keys, seg_len, l.sstart, l.bsums, l.big_list, l.big_count);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, l.bsums, blocks, l.grand);
            hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, st, l.sstart, n_keys, l.bsums, l.grand);
        }
        if (two_level) {
            // already sorted
        } else if (ranged) {
            hipLaunchKernelGGL(scatter_range_kernel, dim3(w_count * (B >> range_log)), dim3(SORT_THREADS), (4u << range_log), st, d_dig, m, dstride, c, w_first, range_log, l.bstart, l.sorted);
        } else {
            const unsigned blocks = pre ? (unsigned)(w_count * nchunk) : (unsigned)(8 * ((w_count + 7) / 8) * nchunk);
            hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(SORT_THREADS), B * 4, st, d_dig, m, dstride, c, w_first, w_count, nchunk, ch_len, pre ? 1 : 0, (uint32_t)n, pw_first, l.hist, l.bstart, l.sorted);
        }
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }

    // stage 5: the dominant kernel; stage 6: buckets whose entries span several segments (three tiers, one launch)
    int stage_accumulate(uint32_t m, uint32_t seg_len, hipStream_t st, const uint32_t* p_sorted, const uint32_t* p_bstart,
                         const uint32_t* p_sstart, const uint32_t* p_big_list, const uint32_t* p_big_count, bool prio_steps) {
        Work& l = ws;
        const uint32_t n_keys = l.groups * B;
        const uint64_t lanes_needed = ((uint64_t)l.w_count * m + seg_len - 1) / seg_len;
        bool launched = false;
        if constexpr (AccumulateSplit<G>::ON) {
            if (opt.split_pairs < 0 ? AccumulateSplit<G>::DEFAULT : opt.split_pairs != 0) {
                // Fp2 groups: a lane PAIR per segment, every value split by component (fp2_split.hip.h)
                hipLaunchKernelGGL(accumulate_split_kernel<G>, dim3((unsigned)((2 * lanes_needed + 255) / 256)), dim3(256), 0, st, d_bases, p_sorted, p_bstart, p_sstart, n_keys, seg_len, prio_steps ? 1u : 0u, l.partials, l.buckets);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL(accumulate_kernel<G>, dim3((unsigned)((lanes_needed + 255) / 256)), dim3(256), 0, st, d_bases, p_sorted, p_bstart, p_sstart, n_keys, seg_len, prio_steps ? 1u : 0u, l.partials, l.buckets);
        ZK_HIP(hipEventRecord(l.ev_acc1, st));
        const uint32_t small_blocks = (2 * n_keys + COMBINE_THREADS - 1) / COMBINE_THREADS;
        hipLaunchKernelGGL(combine_kernel<G>, dim3(small_blocks + COMBINE_WAVE_BLOCKS + COMBINE_BIG_BLOCKS), dim3(COMBINE_THREADS), 0, st,
                           l.partials, p_sstart, n_keys, small_blocks, p_big_list, p_big_count, l.buckets);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }

    // stage 7: sum_b (b + 1) B_b per bucket set down to (S, T) per block of row / column sums, copied to h_final
    int stage_reduce(hipStream_t st) {
        Work& l = ws;
        const uint32_t groups = l.groups;
        const uint32_t n_keys = groups * B;
        // 7. reduce: rows (sum over lo), cols (sum over hi), weighted sums
        uint32_t n_rows = groups * R, n_cols = groups * C;
        SumJob rows = {n_rows, R, B, C, 1u, C, 0u};
        SumJob cols = {n_cols, C, B, 1u, C, R, n_rows};
        // Two steps when the sums are long: one lane pair walks a run of K buckets with no idle lanes (the tree of the
        // one-step form leaves half of its lane-steps empty), then a short tree adds the R / K or C / K partial sums.
        const bool one_step = opt.sum_one_step;  // A/B knob
        // worth it from 2^18 buckets on (8 windows of 2^15, or the 2^19-bucket set of a fixed-base plan): with fewer the sums
        // are a latency chain and the second launch only lengthens it (measured: 2^17 buckets 0.223 vs 0.212 ms)
        const uint32_t K = (one_step || n_keys < (1u << 18)) ? 0u : sum_part_len();
        if (K >= 2 && l.parts && C % K == 0 && R % K == 0 && C / K >= 2 && R / K >= 2) {
            const uint32_t pr = C / K, pc = R / K;  // partial sums per row sum / per column sum
            // step 1: partial (row r, part p) = sum of buckets r C + p K + [0, K); (column j, part p) = sum of (p K + i) C + j
            SumJob prow = {n_rows * pr, R * pr, B, C, 1u, K, 0u, pr, K};
            SumJob pcol = {n_cols * pc, C * pc, B, 1u, C, K, n_rows * pr, pc, K * C};
            hipLaunchKernelGGL(strided_sum_kernel<G>, dim3((unsigned)((((uint64_t)prow.n_out + pcol.n_out) * 2 + 255) / 256)), dim3(256), 0, st, l.buckets, l.parts, prow, pcol, 2u);
            // step 2: contiguous runs of pr (pc) partial sums
            SumJob frow = {n_rows, n_rows, 0u, pr, 1u, pr, 0u};
            SumJob fcol = {n_cols, n_cols, 0u, pc, 1u, pc, n_rows, 1u, 0u, n_rows * pr};
            const uint32_t lpo2 = 2 * std::min<uint32_t>(32u, std::max(pr, pc) / 2);
            hipLaunchKernelGGL(strided_sum_kernel<G>, dim3(((n_rows + n_cols) * lpo2 + 255) / 256), dim3(256), 0, st, l.parts, l.rows, frow, fcol, lpo2);
        } else {
        // lanes per output (two lanes = one point): many outputs (one bucket set per window) -> 16 pairs each walk
        // count/16 buckets and finish with a 4-level tree; few outputs (shared bucket set) -> 32 pairs, shortest chain
        const uint32_t lpo_env = opt.lanes_per_output;
        const uint32_t lpo = lpo_env ? lpo_env : ((n_rows + n_cols) >= 4096 ? 32u : 64u);
        hipLaunchKernelGGL(strided_sum_kernel<G>, dim3(((n_rows + n_cols) * lpo + 255) / 256), dim3(256), 0, st, l.buckets, l.rows, rows, cols, lpo);
        }
        hipLaunchKernelGGL(weighted_sum_kernel<G>, dim3(groups * (bpr + bpc)), dim3(HS_THREADS), (size_t)HalfRegs<F>::COUNT * HS_THREADS * 4, st,
                           l.rows, R, groups, l.rows + (size_t)n_rows * XW, C, l.fin);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(h_final, l.fin, (size_t)groups * (bpr + bpc) * 2 * XW * 4, hipMemcpyDeviceToHost, st));
        return ZK_OK;
    }

    // stages 2..7 + D2H
    // `borrowed`: the digits and the sort of another plan's run over the same scalars (enqueue_shared); stages 1-4 are skipped
    // phase: 0 = everything, 1 = up to the sorted entry list only, 2 = from the accumulate kernel on (after phase 1).
    // gate: waited for right before the accumulate kernel (another plan's accumulate has finished), so that the
    // accumulate kernels of several plans run one after the other while their sorts and reductions overlap.
    int run_stages(uint32_t m, uint32_t dstride, hipStream_t st, const SortExport* borrowed = nullptr, int phase = 0, hipEvent_t gate = nullptr) {
        Work& l = ws;
        int rc;
        const uint32_t seg_len = borrowed ? borrowed->seg_len : (phase == 2 ? l.seg_len : pick_seg_len((uint64_t)l.w_count * m));
        l.seg_len = seg_len;
        const uint32_t *p_sorted = l.sorted, *p_bstart = l.bstart, *p_sstart = l.sstart, *p_big_list = l.big_list, *p_big_count = l.big_count;
        if (phase == 2) {
            // sorted in phase 1
        } else if (borrowed) {
            p_sorted = borrowed->sorted; p_bstart = borrowed->bstart; p_sstart = borrowed->sstart;
            p_big_list = borrowed->big_list; p_big_count = borrowed->big_count;
            ZK_HIP(hipStreamWaitEvent(st, borrowed->sorted_ready, 0));
        } else if ((rc = stage_sort(m, dstride, seg_len, st))) {
            return rc;
        }
        if (phase != 2) ZK_HIP(hipEventRecord(l.ev_acc0, st));
        if (phase == 1) return ZK_OK;
        if (gate) ZK_HIP(hipStreamWaitEvent(st, gate, 0));
        l.acc_from_accs = gate != nullptr || phase == 2;
        if (l.acc_from_accs) ZK_HIP(hipEventRecord(l.ev_accs, st));
        // priority steps (msm_accumulate.hip.h) for a run in one piece that waits for nothing: the two-step, gated and shared forms
        // are what a prover uses to overlap several plans
        const bool prio_steps = opt.priority_steps && phase == 0 && !gate && !borrowed;
        if ((rc = stage_accumulate(m, seg_len, st, p_sorted, p_bstart, p_sstart, p_big_list, p_big_count, prio_steps))) return rc;
        if (borrowed) ZK_HIP(hipEventRecord(borrowed->release, st));  // the lender's buffers are no longer read
        return stage_reduce(st);   // the caller records ev_end
    }

    int set_option(const char* name, int64_t value) override {
        std::lock_guard<std::mutex> lock(mu);
        if (q_pending || q_sorted) return fail(ZK_ERR_ARG, "MSM plan has a run in flight: options change between runs");
        if (!strcmp(name, "segment_lanes")) {
            if (value < 64) return fail(ZK_ERR_ARG, "segment_lanes must be at least 64");
            // the partials buffer was sized for the creation-time target: more lanes than that would overrun it
            if ((uint64_t)value > seg_lanes_at_init) return fail(ZK_ERR_ARG, "segment_lanes cannot exceed the value the plan was created with");
            opt.segment_lanes = (uint64_t)value;
        } else if (!strcmp(name, "sum_one_step")) {
            opt.sum_one_step = value != 0;
        } else if (!strcmp(name, "lanes_per_output")) {
            if (value != 0 && (value < 2 || value > 64 || (value & (value - 1)))) return fail(ZK_ERR_ARG, "lanes_per_output: 0 or a power of two in [2, 64]");
            opt.lanes_per_output = (uint32_t)value;
        } else if (!strcmp(name, "priority_steps")) {
            opt.priority_steps = value != 0;
        } else if (!strcmp(name, "split_pairs")) {
            if (value < -1 || value > 1) return fail(ZK_ERR_ARG, "split_pairs: -1 (the group's default), 0 or 1");
            if (value == 1 && !AccumulateSplit<G>::ON) return fail(ZK_ERR_ARG, "split_pairs: this group has no pair-split accumulate kernel (base-field groups)");
            opt.split_pairs = (int)value;
        } else if (!strcmp(name, "two_level_sort")) {
            if (value && !ws.tmp_ref) return fail(ZK_ERR_ARG, "the plan was created without the buffers of the two-level sort");
            if (!value && wide) return fail(ZK_ERR_ARG, "windows wider than 16 bits exist in the two-level sort only");
            opt.two_level_sort = value != 0;
        } else {
            return fail(ZK_ERR_ARG, std::string("unknown or creation-time MSM option: ") + name);
        }
        return ZK_OK;
    }

    uint64_t seg_lanes_at_init = 0;
    int create_flags = 0;
    int clone(MsmPlanBase** out) override {
        MsmPlan* p = new MsmPlan();
        p->opt = opt;
        int rc = p->init(n_api, nullptr, 0, create_flags & ~ZK_MSM_HIGH_PRIORITY, c, pw_first, pw_count, this);
        if (rc) {
            delete p;
            return rc;
        }
        *out = p;
        return ZK_OK;
    }

    // state carried from enqueue() to finish()
    int q_first = 0, q_count = 0;
    uint32_t q_m = 0;
    hipStream_t q_stream = nullptr;
    bool q_pending = false;

    bool q_sorted = false;  // enqueue_sort done, enqueue_rest still to come

    int enqueue(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st) override {
        return enqueue_phase(n_scalars, scalars, on_device, w_first, w_count, st, 0);
    }
    int enqueue_sort(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st) override {
        return enqueue_phase(n_scalars, scalars, on_device, w_first, w_count, st, 1);
    }
    hipEvent_t accumulate_done_event() override { return ws.ev_acc1; }
    int enqueue_rest(MsmPlanBase* after) override {
        std::lock_guard<std::mutex> lock(mu);
        if (!q_sorted) return fail(ZK_ERR_ARG, "zk_msm_plan_enqueue_rest without zk_msm_plan_enqueue_sort");
        q_sorted = false;
        if (q_m > 0) {
            int rc = run_stages(q_m, (q_m + 7u) & ~7u, q_stream, nullptr, 2, after ? after->accumulate_done_event() : nullptr);
            if (rc) return rc;
            ZK_HIP(hipEventRecord(ev_end, q_stream));
        }
        q_pending = true;
        return ZK_OK;
    }

    int enqueue_phase(uint64_t n_scalars, const void* scalars, int on_device, int w_first, int w_count, hipStream_t st, int phase) {
        std::lock_guard<std::mutex> lock(mu);
        if (q_pending || q_sorted) return fail(ZK_ERR_ARG, "MSM plan already has a run in flight: call zk_msm_plan_finish first");
        if (n_scalars > n_api) return fail(ZK_ERR_LENGTH, "Number of points and scalars mismatch");
        if (w_count <= 0) { w_first = pw_first; w_count = pw_count; }
        if (w_first < pw_first || w_first + w_count > pw_first + pw_count)
            return fail(ZK_ERR_ARG, "window range out of bounds (the plan was created for windows [" + std::to_string(pw_first) + ", " +
                                        std::to_string(pw_first + pw_count) + "))");
        const uint32_t m_api = (uint32_t)n_scalars;
        const uint32_t m = glv ? 2 * m_api : m_api;  // entries per window
        q_first = w_first; q_count = w_count; q_m = m; q_stream = st;
        if (ws.lent) {  // a borrower of the previous run's sort may still be reading the buffers this run overwrites
            ZK_HIP(hipStreamWaitEvent(st, ws.ev_release, 0));
            ws.lent = false;
        }
        if (m > 0) {
            const uint32_t* sc = (const uint32_t*)scalars;
            if (!on_device) {
                ZK_HIP(hipMemcpyAsync(d_scalars, scalars, (size_t)m_api * FrP::W * 4, hipMemcpyHostToDevice, st));
                sc = d_scalars;
            }
            const uint32_t dstride = (m + 7u) & ~7u;
            ZK_HIP(hipEventRecord(ev_start, st));
            // 1. digits (the windows of this run); digit rows are stored relative to the plan's first window
            DigitBias bias;
            memset(&bias, 0, sizeof(bias));
            for (int w = 0; w < nwin; ++w) {
                int bit = w * c + (c - 1);
                bias.v[bit >> 5] |= 1u << (bit & 31);
            }
            const uintptr_t dig_base = reinterpret_cast<uintptr_t>(d_dig) - (uintptr_t)pw_first * dstride * (wide ? 4 : 2);
            if (glv) {
                if constexpr (GlvOf<G>::OK)
                    hipLaunchKernelGGL(glv_digits_kernel<FrP>, dim3((m_api + 255) / 256), dim3(256), 0, st, sc, m_api, dstride, c, w_first, w_count, bias,
                                       GlvOf<G>::P::K, reinterpret_cast<uint16_t*>(dig_base), ws.big_count);
            } else if (wide) hipLaunchKernelGGL((digits_kernel<FrP, uint32_t>), dim3((m + 255) / 256), dim3(256), 0, st, sc, m, dstride, c, w_first, w_count, bias,
                                         reinterpret_cast<uint32_t*>(dig_base), ws.big_count);
            else hipLaunchKernelGGL((digits_kernel<FrP, uint16_t>), dim3((m + 255) / 256), dim3(256), 0, st, sc, m, dstride, c, w_first, w_count, bias,
                                    reinterpret_cast<uint16_t*>(dig_base), ws.big_count);
            ws.w_first = w_first;
            ws.w_count = w_count;
            ws.groups = pre ? 1u : (uint32_t)w_count;
            int rc = run_stages(m, dstride, st, nullptr, phase);
            if (rc) return rc;
            if (phase == 0) ZK_HIP(hipEventRecord(ev_end, st));
        }
        if (phase == 1) q_sorted = true;
        else q_pending = true;
        return ZK_OK;
    }

    int cancel() override {
        std::lock_guard<std::mutex> lock(mu);
        if ((q_pending || q_sorted) && q_stream) ZK_HIP(hipStreamSynchronize(q_stream));
        q_pending = q_sorted = false;
        return ZK_OK;
    }

    int export_sort(SortExport* out) override {
        std::lock_guard<std::mutex> lock(mu);
        if ((!q_pending && !q_sorted) || q_m == 0) return fail(ZK_ERR_ARG, "the lending plan has no run in flight");
        out->sorted = ws.sorted; out->bstart = ws.bstart; out->sstart = ws.sstart;
        out->big_list = ws.big_list; out->big_count = ws.big_count;
        out->n = n; out->m = q_m; out->seg_len = ws.seg_len; out->groups = ws.groups;
        out->c = c; out->nwin = nwin; out->w_first = q_first; out->w_count = q_count;
        out->pw_first = pw_first; out->pw_count = pw_count; out->scalar_bits = FrP::BITS; out->pre = pre; out->glv = glv;
        out->endo = glv ? G::ENDO_ID : 0;
        out->sorted_ready = ws.ev_acc0;
        out->release = ws.ev_release;
        ws.lent = true;
        return ZK_OK;
    }

    // Run this plan on the digits and the sorted entry list of `lender`'s run in flight: same scalars, other bases
    // (Groth16's <tau_1, v> and <tau_2, v>).  Both plans must have the same size, window layout, mode and window range;
    // the entry list addresses points (or table rows) by index, which is independent of the group.
    int enqueue_shared(MsmPlanBase* lender, hipStream_t st) override {
        SortExport ex;
        int rc = lender->export_sort(&ex);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(mu);
        if (q_pending || q_sorted) return fail(ZK_ERR_ARG, "MSM plan already has a run in flight: call zk_msm_plan_finish first");
        if (ex.n != n || ex.c != c || ex.nwin != nwin || ex.pre != pre || ex.glv != glv || ex.endo != (glv ? G::ENDO_ID : 0) || ex.scalar_bits != FrP::BITS ||
            ex.pw_first != pw_first || ex.pw_count != pw_count)
            return fail(ZK_ERR_ARG, "plans differ in size, window layout or mode: the sort cannot be shared");
        if (ws.lent) {
            ZK_HIP(hipStreamWaitEvent(st, ws.ev_release, 0));
            ws.lent = false;
        }
        q_first = ex.w_first; q_count = ex.w_count; q_m = ex.m; q_stream = st;
        ZK_HIP(hipEventRecord(ev_start, st));
        ws.w_first = ex.w_first;
        ws.w_count = ex.w_count;
        ws.groups = ex.groups;
        rc = run_stages(ex.m, (ex.m + 7u) & ~7u, st, &ex);
        if (rc) return rc;
        ZK_HIP(hipEventRecord(ev_end, st));
        q_pending = true;
        return ZK_OK;
    }

    int finish(uint64_t* out) override {
        std::lock_guard<std::mutex> lock(mu);
        if (q_sorted) return fail(ZK_ERR_ARG, "zk_msm_plan_finish before zk_msm_plan_enqueue_rest");
        if (!q_pending) return fail(ZK_ERR_ARG, "zk_msm_plan_finish without a pending run");
        q_pending = false;
        typedef typename G::HostF HF;  // 64-bit-limb host arithmetic for the sequential tail (host64.hip.h)
        XYZZ<HF> total = xyzz_inf<HF>();
        std::chrono::steady_clock::time_point tail_t0;
        if (q_m > 0) {
            ZK_HIP(hipEventSynchronize(ev_end));
            tail_t0 = std::chrono::steady_clock::now();
            // 8. host tail.  Per bucket set, with row blocks (S_k, T_k), column blocks (S'_k, T'_k), C = 2^lc, WS_BLOCK = 2^7:
            //        W = 2^(lc+7) X_R + 2^lc sum S_k + 2^7 X_C + (sum S'_k + sum T_k),   X = sum_k k T_k  (k = 1 at most)
            //    general mode: total = sum_w 2^(c w) W_w by Horner from the top window down; the factors ride along the c
            //    doublings between two windows, so a set costs c doublings whatever its block structure;
            //    fixed-base mode: one set, the same chain.
            int lc = 0;
            while ((1u << lc) < C) ++lc;
            const int groups = (int)ws.groups;
            const uint32_t* rows_fin = h_final;
            const uint32_t* cols_fin = h_final + (size_t)groups * bpr * 2 * XW;
            auto pt = [](const uint32_t* p) { return HF::xyzz_from_device(p); };
            bool first_set = true;
            for (int g = groups - 1; g >= 0; --g) {
                // X = sum_k k T_k as a running sum of suffixes (from the top block down): two additions per block
                XYZZ<HF> sum_s = xyzz_inf<HF>(), sum_t = xyzz_inf<HF>(), x_r = xyzz_inf<HF>();
                for (int k = (int)bpr - 1; k >= 0; --k) {
                    const uint32_t* q = rows_fin + ((size_t)g * bpr + k) * 2 * XW;
                    sum_s = xyzz_add<HF>(sum_s, pt(q));
                    if (k > 0) {
                        sum_t = xyzz_add<HF>(sum_t, pt(q + XW));  // T_k + .. + T_top
                        x_r = xyzz_add<HF>(x_r, sum_t);
                    } else {
                        sum_t = xyzz_add<HF>(sum_t, pt(q + XW));
                    }
                }
                XYZZ<HF> sum_sc = xyzz_inf<HF>(), x_c = xyzz_inf<HF>(), suf_c = xyzz_inf<HF>();
                for (int k = (int)bpc - 1; k >= 0; --k) {
                    const uint32_t* q = cols_fin + ((size_t)g * bpc + k) * 2 * XW;
                    sum_sc = xyzz_add<HF>(sum_sc, pt(q));
                    if (k > 0) {
                        suf_c = xyzz_add<HF>(suf_c, pt(q + XW));
                        x_c = xyzz_add<HF>(x_c, suf_c);
                    }
                }
                // terms in descending order of their exponent; `at` = exponent the accumulator currently sits at
                int at = first_set ? -1 : c;
                auto step = [&](int exp, const XYZZ<HF>& term, bool present) {
                    if (!present) return;
                    if (at >= 0) for (int k = 0; k < at - exp; ++k) total = xyzz_dbl<HF>(total);
                    total = xyzz_add<HF>(total, term);
                    at = exp;
                };
                if (bpr > 1 && lc + WS_BLOCK_LOG > c && !first_set) return fail(ZK_ERR_ARG, "internal: reduction layout does not fit the window");
                step(lc + WS_BLOCK_LOG, x_r, bpr > 1);
                step(lc, sum_s, true);
                step(WS_BLOCK_LOG, x_c, bpc > 1);
                step(0, xyzz_add<HF>(sum_sc, sum_t), true);
                first_set = false;
            }
            if (!pre) for (int k = 0; k < c * q_first; ++k) total = xyzz_dbl<HF>(total);
            float sort_ms = 0, acc_ms = 0, red_ms = 0, t = 0;
            if (hipEventElapsedTime(&t, ev_start, ws.ev_acc0) == hipSuccess) sort_ms += t;
            if (hipEventElapsedTime(&t, ws.acc_from_accs ? ws.ev_accs : ws.ev_acc0, ws.ev_acc1) == hipSuccess) acc_ms += t;
            if (hipEventElapsedTime(&t, ws.ev_acc1, ev_end) == hipSuccess) red_ms += t;
            timings[0] = sort_ms;   // digits + sort
            timings[1] = acc_ms;    // accumulate kernel
            timings[2] = red_ms;    // combine + reduction + D2H
        }
        HF::affine_to_canonical(xyzz_to_affine<HF>(total), out);
        if (q_m > 0) {
            // the tail runs on the host: its time comes from the host's clock (an event recorded and awaited here would put a
            // GPU round trip of 10-20 us on the critical path of every MSM just to time it)
            timings[3] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tail_t0).count();
            float t = 0;
            timings[4] = hipEventElapsedTime(&t, ev_start, ev_end) == hipSuccess ? t + timings[3] : 0.0f;
        }
        return ZK_OK;
    }
};

// batch_multi_scalar_g1/_g2: out[i] = k_i * P_i (canonical affine).  One base for many scalars (what Groth16.setup
// does) goes through the fixed-base table: <= 16 mixed additions per scalar; everything ends in the batched normalisation.
constexpr uint64_t FIXED_BASE_MIN = 2048;  // below this the table (2^19 rows) costs more than it saves

template <class G>
static int batch_mul_impl(uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast, uint64_t* out) {
    typedef typename G::F F;
    typedef typename G::Fr FrP;
    constexpr int AW = 2 * F::LIMBS, XW = 4 * F::LIMBS;
    if (n == 0) return ZK_OK;
    uint32_t *ds = nullptr, *db = nullptr, *dout = nullptr, *temp = nullptr;
    int rc = ZK_OK;
    const bool fixed = broadcast && n >= FIXED_BASE_MIN;
    uint64_t nb = broadcast ? 1 : n;
    ZK_ALLOC(&ds, n * FrP::W * 4);
    do {
        if (dev_alloc_cached((void**)&temp, n * XW * 4) != ZK_OK || dev_alloc_cached((void**)&dout, n * AW * 4) != ZK_OK) { rc = ZK_ERR_HIP; break; }
        if (hipMemcpy(ds, scalars, n * FrP::W * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
        if (fixed) {
            FixedTable<G>& ft = FixedTable<G>::get();
            std::lock_guard<std::mutex> lock(ft.mu);
            if ((rc = ft.ensure(bases))) break;
            FixedBias bias;
            memset(&bias, 0, sizeof(bias));
            for (int j = 0; j < ft.nwin; ++j) {
                int bit = j * FIXED_C + (FIXED_C - 1);
                bias.v[bit >> 5] |= 1u << (bit & 31);
            }
            hipLaunchKernelGGL(fixed_mul_kernel<G>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ds, n, ft.d_table, ft.nwin, bias, temp);
            if (hipDeviceSynchronize() != hipSuccess) { rc = fail(ZK_ERR_HIP, "fixed-base multiplication kernel failed"); break; }
        } else {
            if (dev_alloc_cached((void**)&db, nb * AW * 4) != ZK_OK) { rc = ZK_ERR_HIP; break; }
            if (hipMemcpy(db, bases, nb * AW * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed"); break; }
            hipLaunchKernelGGL(varbase_mul_kernel<G>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, ds, db, broadcast, n, temp);
        }
        hipLaunchKernelGGL(normalize_kernel<G>, dim3((unsigned)((n + (uint64_t)NORM_THREADS * NORM_E - 1) / ((uint64_t)NORM_THREADS * NORM_E))),
                           dim3(NORM_THREADS), 0, 0, temp, n, dout, 1);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * AW * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "batch_mul kernel / D2H failed"); break; }
    } while (0);
    dev_free_cached(ds); dev_free_cached(db); dev_free_cached(dout); dev_free_cached(temp);
    return rc;
}

// batched (de)compression of n points between host buffers: `to_bytes` != 0 encodes canonical affine rows, 0 decodes.
// On a failing point the first one (lowest index) decides the error, as a sequential loop over the file would.
template <class G>
static int codec_impl(uint64_t n, const void* in, void* out, int to_bytes, uint64_t* bad_index) {
    typedef typename G::F F;
    constexpr size_t ROW = (size_t)2 * F::LIMBS * 4, ENC = CodecLayout<G>::TOTAL;
    if (n == 0) return ZK_OK;
    const size_t in_bytes = n * (to_bytes ? ROW : ENC), out_bytes = n * (to_bytes ? ENC : ROW);
    uint8_t *din = nullptr, *dout = nullptr;
    unsigned long long* derr = nullptr;
    int rc = ZK_OK;
    ZK_ALLOC(&din, in_bytes);
    do {
        if (dev_alloc_cached((void**)&dout, out_bytes) != ZK_OK || dev_alloc_cached((void**)&derr, 8) != ZK_OK) { rc = ZK_ERR_HIP; break; }
        unsigned long long first = ~0ull;
        if (hipMemcpy(din, in, in_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(derr, &first, 8, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(ZK_ERR_HIP, "hipMemcpy H2D failed");
            break;
        }
        const dim3 grid((unsigned)((n + 127) / 128)), block(128);
        if (to_bytes) hipLaunchKernelGGL(points_encode_kernel<G>, grid, block, 0, 0, (const uint32_t*)din, n, dout, derr);
        else hipLaunchKernelGGL(points_decode_kernel<G>, grid, block, 0, 0, (const uint8_t*)din, n, (uint32_t*)dout, derr);
        if (hipGetLastError() != hipSuccess || hipMemcpy(&first, derr, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "point codec kernel failed"); break; }
        if (first != ~0ull) {
            if (bad_index) *bad_index = (uint64_t)(first >> 8);
            rc = fail(ZK_ERR_POINT, codec_message((int)(first & 0xFF)));
            break;
        }
        if (hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipMemcpy D2H failed"); break; }
    } while (0);
    dev_free_cached(din); dev_free_cached(dout); dev_free_cached(derr);
    return rc;
}


#endif  // ZK_PART == 0

}  // namespace zkmi
