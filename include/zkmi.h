/*
 * zkmi.h -- C ABI of libzkmi.so, the MI355X (gfx950) proving backend that replaces the
 * field/curve hot path of Merricx/zksnake behind Groth16.prove()/Plonk.prove().
 *
 * Drop-in boundary: the reference crosses from Python into Rust through the pyo3 module
 * `zksnake._algebra` (reference src/lib.rs:6-185).  Each entry point below names the pyo3
 * function(s) it stands in for (file:line under the reference tree).  INTEGRATION.md shows the
 * ctypes stub a maintainer would add to python/zksnake/{ecc,polynomial}.py.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; no C++/torch types; never throws; returns an int status;
 *     zk_last_error() gives the message of the last failure on the calling thread.
 *   - field elements: canonical (non-Montgomery) integers as little-endian 64-bit limbs.
 *       Fr (both curves): 4 limbs.  Fq: 4 limbs (BN254) / 6 limbs (BLS12-381).
 *     Values >= modulus are reduced on entry, like `Fr::from(BigUint)` in the reference
 *     (src/bn254/polynomial.rs:538, src/bn254/curve.rs:359).
 *   - points: affine (x, y); G1 = 2 Fq elements, G2 = 4 (x.c0, x.c1, y.c0, y.c1); the point at
 *     infinity is the all-zero encoding ((0,0) is on none of the four curves since b != 0).
 *   - "host" functions take host memory and copy; "_dev" functions take HIP device pointers
 *     (e.g. torch tensor data_ptr()) plus a hipStream_t passed as void* (NULL = default stream).
 *   - the caller owns every buffer it passes; handles are freed explicitly.
 *   - calls may come from several threads (ctypes releases the GIL): the library serialises GPU
 *     work per handle internally.
 */
#ifndef ZKMI_H
#define ZKMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_CURVE_BN254 0
#define ZK_CURVE_BLS12_381 1
#define ZK_G1 1
#define ZK_G2 2

#define ZK_OK 0
#define ZK_ERR_LENGTH 1        /* "Number of points and scalars mismatch" (src/bn254/curve.rs:369-371) */
#define ZK_ERR_DOMAIN 2        /* domain larger than 2^two-adicity (polynomial.rs:48 unwrap / :638 error) */
#define ZK_ERR_HIP 3           /* HIP / RCCL runtime failure, or no GPU present */
#define ZK_ERR_POINT 4         /* bad point encoding / not on curve (curve.rs:137-140) */
#define ZK_ERR_ARG 5           /* invalid curve/group/size argument */
#define ZK_ERR_NOT_DIVISIBLE 6 /* (U*V - W) mod Z != 0 (groth16/qap.py:67-69) */

/* ---- lifecycle ------------------------------------------------------------------------- */

int zk_init(int device);            /* select the HIP device for this process; ZK_ERR_HIP when no GPU */

/* Hardware queues.  A proof keeps seven HIP streams busy; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
 * queues (4 by default) and streams that share a queue run one after the other (~ +1 ms per Groth16 proof at 2^20).  The
 * variable is read once, when the HIP runtime starts (the process's first HIP call).  zk_init / zk_init_ex /
 * zk_hw_queues_prepare therefore set GPU_MAX_HW_QUEUES=12 BEFORE the library's first HIP call when the caller has not set
 * it and the runtime is not running yet (the reference has nothing like it: its prover is single-threaded CPU code behind
 * the GIL, src/lib.rs:6-185).  The status says what happened:
 *   ZK_QUEUES_SET_BY_LIBRARY  the library put the setting in place in time;
 *   ZK_QUEUES_CALLER          GPU_MAX_HW_QUEUES was already in the environment and was left alone (*queues = its value);
 *   ZK_QUEUES_TOO_LATE        the runtime was already running without the setting (e.g. torch touched the GPU before the
 *                             first zk_ call): results are unaffected, concurrent streams may serialise.  Call
 *                             zk_hw_queues_prepare() (no HIP call inside) right after loading the library, or export the
 *                             variable yourself, to avoid it.
 * The decision is taken once per process; later calls return the same status. */
#define ZK_QUEUES_SET_BY_LIBRARY 0
#define ZK_QUEUES_CALLER 1
#define ZK_QUEUES_TOO_LATE 2
int zk_hw_queues_prepare(int* queues /* may be NULL; 0 = runtime default */);
int zk_init_ex(int device, int* queue_status, int* queues);   /* zk_init + the status above (either pointer may be NULL) */
/* test aid: one wave that spins for `microseconds` on `stream` (used to observe whether streams overlap) */
int zk_debug_spin_dev(void* stream, uint64_t microseconds);
int zk_shutdown(void);              /* free cached twiddle tables and workspaces */
int zk_device_count(void);          /* number of visible HIP devices (0 without a GPU; never fails) */
const char* zk_last_error(void);
const char* zk_version(void);

/* raw device memory for bindings that do not bring their own allocator (hipMalloc/hipFree/hipMemcpy);
 * a torch tensor's data_ptr() is equally valid wherever a device pointer is expected. */
int zk_dev_alloc(uint64_t bytes, void** d_ptr);
int zk_dev_free(void* d_ptr);
int zk_dev_upload(void* d_dst, const void* h_src, uint64_t bytes);
int zk_dev_download(void* h_dst, const void* d_src, uint64_t bytes);
/* upload ordered on `stream` and asynchronous to the host when h_src is page-locked (zk_host_alloc): lets a host that
 * marshals a long witness chunk by chunk ship chunk k while it converts chunk k + 1 */
int zk_dev_upload_async(void* d_dst, const void* h_src, uint64_t bytes, void* stream);
int zk_dev_memset(void* d_dst, int value, uint64_t bytes);
/* the same, ordered on `stream` (NULL = default stream): zk_dev_memset runs on the legacy default stream, which does not
 * synchronise with the non-blocking streams of zk_stream_create */
int zk_dev_memset_async(void* d_dst, int value, uint64_t bytes, void* stream);
int zk_dev_synchronize(void);
/* page-locked host memory: a witness marshalled into it (and results copied out of it) moves at the link rate instead of
 * through the runtime's staging copies (32 MB of witness: ~0.6 ms against 2-3 ms from pageable memory) */
int zk_host_alloc(uint64_t bytes, void** h_ptr);
int zk_host_free(void* h_ptr);
/* HIP streams for callers that keep several device-resident pipelines in flight (the Python host has no HIP binding
 * of its own): non-blocking with respect to the default stream; high_priority != 0 asks for the top priority level. */
int zk_stream_create(int high_priority, void** stream);
int zk_stream_destroy(void* stream);
int zk_stream_synchronize(void* stream);

/* limb counts, so bindings do not hard-code them */
int zk_fq_limbs(int curve);         /* 64-bit limbs per base-field element: 4 / 6 */
int zk_point_limbs(int curve, int group); /* 64-bit limbs per affine point */

/* ---- scalar-field vectors (polynomial_bn254 / polynomial_bls12_381 submodules) ---------- */

/* fft / ifft / coset_fft / coset_ifft (src/bn254/polynomial.rs:535-585).
 * `in` holds n_in elements; the domain is next_pow2(size); the input is zero-padded, or, when
 * longer than the domain, truncated to it (ark-poly 0.4.2 fft_in_place/ifft_in_place resize the
 * vector to the domain size; the crate is not vendored: parity unpinned); `out` receives the whole
 * domain in natural order. coset != 0 uses the reference's offset (= the domain generator). */
int zk_ntt(int curve, int inverse, int coset, uint64_t n_in, const uint64_t* in, uint64_t size, uint64_t* out);

/* mul_over_evaluation_domain / add_over_evaluation_domain (polynomial.rs:587-634): element-wise
 * over `size` entries; entries beyond n_a / n_b count as zero.  op: 0 = mul, 1 = add, 2 = sub.
 * Deliberate divergence at THIS level: the reference's add_over_evaluation_domain indexes a[i], b[i] for every i < size
 * and panics on a shorter input (polynomial.rs:594-597), while mul zero-pads (polynomial.rs:617-625); the C entry point zero-pads for
 * every op (one kernel, and the PlonK prover adds vectors of unequal length).  The drop-in Python layer restores the
 * reference's behaviour: `add_over_evaluation_domain` raises IndexError for a short input before it gets here
 * (zksnake_amd/_algebra.py). */
int zk_vec_op(int curve, int op, uint64_t size, uint64_t n_a, const uint64_t* a, uint64_t n_b, const uint64_t* b,
              uint64_t* out);

/* Polynomial.divide_by_vanishing_poly (polynomial.rs:466-489): coeffs (len) = q*(X^n - 1) + rem.
 * q must hold max(len - n, 0) elements, rem min(len, n). *rem_is_zero tells whether rem == 0. */
int zk_poly_div_vanishing(int curve, uint64_t n, uint64_t len, const uint64_t* coeffs, uint64_t* q, uint64_t* rem,
                          int* rem_is_zero);

/* Device-resident forms: in place on a vector of 2^log_n canonical Fr elements. */
int zk_ntt_dev(int curve, int inverse, int log_n, void* d_data, void* stream);
int zk_vec_op_dev(int curve, int op, uint64_t n, const void* d_a, const void* d_b, void* d_out, void* stream);
/* reduce every element below the modulus in place (Fr::from(BigUint), src/bn254/curve.rs:358-361): limb-array
 * witnesses handed to the device-resident provers are only ASSUMED canonical by the kernels behind them. */
int zk_vec_canon_dev(int curve, uint64_t n, void* d_x, void* stream);
/* d_out[k] = g^k for k < n, canonical (the powers of tau of a setup, python/zksnake/groth16/protocol.py:48-55) */
int zk_vec_powers_dev(int curve, uint64_t n, const uint64_t* g, void* d_out, void* stream);

/* SparseArray.dot (python/zksnake/array.py:36-44) as a CSR product over Fr on device-resident arrays:
 * row_ptr u32[n_rows + 1], cols u32[nnz], vals canonical Fr[nnz], w canonical Fr[n_cols] (cols must be < n_cols: the
 * host validates, SparseArray.to_csr), out canonical Fr[n_rows].  One lane per row; rows with more than
 * skip_longer_than entries (0 = never) are left untouched for zk_spmv_long_dev. */
int zk_spmv_dev(int curve, uint64_t n_rows, const void* d_row_ptr, const void* d_cols, const void* d_vals,
                const void* d_w, void* d_out, uint32_t skip_longer_than, void* stream);
/* The long rows of the same product: long_rows u32[n_long] (row indices), item_ptr u32[n_long + 1] into
 * items, items u32[n_items][2] = entry ranges [k0, k1) of at most a few thousand entries each (the host cuts the rows
 * once per matrix), partials = scratch of n_items Fr elements.  One workgroup per item, then one per long row. */
int zk_spmv_long_dev(int curve, uint64_t n_long, const void* d_long_rows, const void* d_item_ptr, uint64_t n_items,
                     const void* d_items, const void* d_cols, const void* d_vals, const void* d_w, void* d_partials,
                     void* d_out, void* stream);

/* Fused QAP.evaluate_witness tail (python/zksnake/groth16/qap.py:57-69): from the evaluation vectors
 * a = A.w, b = B.w, c = C.w (2^log_n canonical Fr elements each, device memory) compute in place the
 * coefficient vectors u = iNTT(a), v = iNTT(b) and write h (2^log_n elements, h[2^log_n - 1] = 0) with
 * u*v - w = h*(X^n - 1).  Everything stays in HBM.  With u v = P_lo + X^n P_hi the quotient is h = P_hi, and on the coset g*H of the
 * 2n-th root of unity g (g^n = -1) the values of u v interpolate P_lo - P_hi, so h = (w - d) / 2 with d = that interpolant: 3 iNTT(n)
 * + 2 NTT(n) of g^i-scaled coefficients + 1 iNTT(n), the scalings and the point-wise product folded into the transforms' passes
 * (the reference goes through the doubled domain: 2 NTT(2n) + iNTT(2n) + fold; same h).  log_n + 1 must not exceed the field's
 * two-adicity (ZK_ERR_DOMAIN).
 * d_work must hold 4 * 2^log_n elements.  *divisible (host int) is set to 0 when a_i b_i != c_i for some i, i.e. when
 * the division would leave a remainder (the reference raises ValueError there); h is meaningless in that case. */
int zk_qap_h_dev(int curve, int log_n, void* d_a_u, void* d_b_v, const void* d_c, void* d_h, void* d_work,
                 int* divisible, void* stream);
/* The same in two steps.  _begin puts the whole chain on the stream and returns; *uv_ready (may be NULL) receives an event,
 * owned by the library and reused by the next _begin on that stream, that fires once u and v are final (a third of the
 * way into the chain): MSMs over u and v can sort beside the rest of it (zk_msm_plan_wait_event).  _end copies the
 * divisibility flag back and synchronises the stream. */
int zk_qap_h_dev_begin(int curve, int log_n, void* d_a_u, void* d_b_v, const void* d_c, void* d_h, void* d_work, void* stream,
                       void** uv_ready);
int zk_qap_h_dev_end(int curve, int log_n, const void* d_work, int* divisible, void* stream);
/* Only the first third of the chain: u = iNTT(a) and / or v = iNTT(b) in place (either pointer may be NULL), enqueued on the
 * stream, with the same "u and v are final" event.  For a rank of a task-partitioned prover whose MSM reads u or v only
 * (python/zksnake/groth16/protocol.py:133-147: <tau_1, u>, <tau_1, v> and <tau_2, v> do not need h); no divisibility check --
 * the rank(s) holding <target_1, h> run the whole chain and report it. */
int zk_qap_uv_dev(int curve, int log_n, void* d_a_u, void* d_b_v, void* stream, void** uv_ready);

/* ---- curve groups (ec_bn254 / ec_bls12_381 submodules) ---------------------------------- */

/* multiscalar_mul_g1 / multiscalar_mul_g2 (src/bn254/curve.rs:356-392, bls12_381 :366-402):
 * out = sum_i scalars[i] * bases[i].  n_scalars != n_points -> ZK_ERR_LENGTH. */
int zk_msm(int curve, int group, uint64_t n_points, uint64_t n_scalars, const uint64_t* scalars,
           const uint64_t* bases, uint64_t* out);
/* zk_msm is the reference's call shape and pays for it on every call: the bases go to the device and into Montgomery form, a
 * workspace is built and torn down (2^20 BN254 G1 pairs: ~3.6 ms, of which the MSM itself is 1.5).  The library does NOT cache
 * bases behind this entry point: recognising "the same array" by pointer and a hash of sampled rows would return a wrong point
 * after an in-place change the samples miss, and hashing all 64 MB costs more than the MSM.  A caller that multiplies against the
 * same points more than once (a proving key) creates a plan once -- zk_msm_plan_create below -- and calls zk_msm_plan_run; the
 * Python layer does exactly that behind EllipticCurve.multiexp when it is handed a PointArray (zksnake_amd/_algebra.py). */

/* batch_multi_scalar_g1 / _g2 (src/bn254/curve.rs:326-354): out[i] = scalars[i] * bases[i];
 * broadcast != 0 means `bases` holds one point used for every scalar (ecc.py:93-94). */
int zk_batch_mul(int curve, int group, uint64_t n, const uint64_t* scalars, const uint64_t* bases, int broadcast,
                 uint64_t* out);

/* Device-resident bases ("proving key stays in HBM").  zk_msm_plan_create uploads/normalises the
 * bases once (host or device pointer per `bases_on_device`) and sizes the workspace for up to n
 * points.  flags: bit0 = precompute the per-window multiples 2^(c*w) * P_i so that every window
 * shares one bucket set (uses n * windows * point bytes of HBM); window_bits 0 = automatic. */
#define ZK_MSM_PRECOMPUTE 1
#define ZK_MSM_HIGH_PRIORITY 2 /* the plan's own stream (ZK_STREAM_PLAN) gets the top stream priority */
/* General (not ZK_MSM_PRECOMPUTE) plans of up to 2^22 points split every scalar with the group's endomorphism,
 * k = k1 + lambda k2 with |k1|, |k2| < 2^127, and run over the 2n points (P_i, phi(P_i)) -- phi = (beta x, y) on G1, the squared
 * untwist-Frobenius-twist map (c x, -y) on G2: the same number of bucket additions in half the windows, so half the bucket
 * sets to reduce and half the doublings in the tail.  The result is the same point.  This flag (or ZKMI_NO_GLV=1 in the
 * environment) keeps the plain 254/255-bit windows.  Two split-scalar plans share a sort (zk_msm_plan_enqueue_shared) only
 * within one group: G1 and G2 split against different eigenvalues. */
#define ZK_MSM_NO_GLV 4
int zk_msm_plan_create(int curve, int group, uint64_t n, const void* bases, int bases_on_device, int flags,
                       int window_bits, uint64_t* handle);
/* The same for a rank of a window-sharded MSM that will only ever run the windows
 * [window_first, window_first + window_count) (window_count 0 = all): the fixed-base table and the
 * workspace are sized for that range only -- 1/8 of the memory and of the table build on 8 ranks.
 * Runs outside the range fail with ZK_ERR_ARG.  Call zk_msm_plan_windows on a throw-away
 * n = 1 plan (or compute ceil((bits + 1) / c)) to learn the window count first. */
int zk_msm_plan_create_range(int curve, int group, uint64_t n, const void* bases, int bases_on_device, int flags,
                             int window_bits, int window_first, int window_count, uint64_t* handle);
/* A second plan over the SAME device-resident bases (and fixed-base table; shared, freed with the last user) with its
 * own workspace and stream, so that two MSMs against one key can be in flight together. */
int zk_msm_plan_clone(uint64_t handle, uint64_t* clone_handle);
int zk_msm_plan_destroy(uint64_t handle);
/* scalars: n_scalars <= plan n canonical Fr elements (host or device per `scalars_on_device`); the
 * first n_scalars bases are used (ecc.py:118-119 truncation rule).  Result: one affine point in host
 * memory.  window_first/window_count select a window range for multi-GPU sharding (count 0 = all):
 * the partial result then is sum over those windows of 2^(c*w) * W_w, so partials of disjoint ranges
 * add up to the full MSM. */
int zk_msm_plan_run(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device,
                    int window_first, int window_count, uint64_t* out, void* stream);
/* Asynchronous form: enqueue puts all GPU stages and the D2H copy of the per-window results on the
 * stream and returns; finish waits and runs the host tail.  One run may be in flight per plan.
 * ZK_STREAM_PLAN selects a stream owned by the plan, so several plans (the five MSMs of a Groth16 proof)
 * overlap: their latency-bound reduction stages hide behind other plans' accumulation kernels.  Plan streams
 * are ordinary blocking streams: they are ordered after earlier work on the NULL stream. */
#define ZK_STREAM_PLAN ((void*)(intptr_t)-1)
int zk_msm_plan_enqueue(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device,
                        int window_first, int window_count, void* stream);
/* zk_msm_plan_enqueue in two steps, for a caller that keeps several plans in flight and wants their accumulate kernels one
 * after the other: _sort puts the digits and the sort on the stream, _rest the accumulate kernel -- not before the
 * accumulate kernel of `after_handle`'s run has finished (0 = no such condition; that run's _rest / enqueue_shared must
 * have been issued already) -- then the reduction and the D2H copy.  A resident accumulate grid holds every wave slot of
 * the chip until it ends, so accumulate kernels that run side by side only delay each other's reductions, and the sorts
 * of plans enqueued later starve behind them; ordered, the sorts run first and every reduction but the last overlaps the
 * next plan's accumulate kernel.  zk_msm_plan_finish as usual. */
int zk_msm_plan_enqueue_sort(uint64_t handle, uint64_t n_scalars, const void* scalars, int scalars_on_device,
                             int window_first, int window_count, void* stream);
int zk_msm_plan_enqueue_rest(uint64_t handle, uint64_t after_handle);
/* the plan's own stream (ZK_STREAM_PLAN) waits for an event of another stream (e.g. zk_qap_h_dev_begin's uv_ready) before
 * whatever is enqueued on it next */
int zk_msm_plan_wait_event(uint64_t handle, void* event);
/* forget a run in flight (sorted only, or complete) without collecting its result: waits for the plan's stream */
int zk_msm_plan_cancel(uint64_t handle);
/* The same scalars against a second set of bases (Groth16: <tau_1, v> in G1 and <tau_2, v> in G2): run `handle` on the
 * digits and the sorted entry list of `lender_handle`'s run in flight instead of sorting again.  Both plans must have the
 * same size, window layout, mode and window range (else ZK_ERR_ARG: enqueue normally); finish both as usual.  A general-mode
 * G1 lender must have been created with ZK_MSM_NO_GLV (a split-scalar sort is of no use to a G2 plan). */
int zk_msm_plan_enqueue_shared(uint64_t handle, uint64_t lender_handle, void* stream);
int zk_msm_plan_finish(uint64_t handle, uint64_t* out);
int zk_msm_plan_windows(uint64_t handle, int* window_bits, int* n_windows);
/* the window width and count a plan over n points will use (window_bits 0 = automatic), without creating one: what a
 * rank needs to pick its share before zk_msm_plan_create_range */
int zk_msm_window_layout(int curve, int group, uint64_t n, int flags, int window_bits, int* window_bits_out, int* n_windows);
/* the same with the choice between the two layouts: all_windows != 0 is what zk_msm_plan_create takes (fixed-base plans of 2^20
 * points or more: 13 windows of 20 bits), 0 what zk_msm_plan_create_range takes (16 windows of 16 bits, which split evenly).  A rank
 * of a task-partitioned prover that holds an MSM whole uses the former. */
int zk_msm_window_layout_ex(int curve, int group, uint64_t n, int flags, int window_bits, int all_windows, int* window_bits_out,
                            int* n_windows);
/* entries one window of the plan sorts and accumulates: n, or 2n when the plan runs on endomorphism pairs */
int zk_msm_plan_entries(uint64_t handle, uint64_t* entries_per_window);
/* milliseconds of the stages of the last zk_msm_plan_run on this plan, measured with HIP events on
 * the launch stream(s): [0] digits+sort, [1] bucket accumulation (dominant kernel; summed over its launches),
 * [2] bucket combination + reduction + D2H, [3] host tail, [4] total.  Returns the number of floats written. */
int zk_msm_plan_timings(uint64_t handle, float* ms, int cap);
/* Tuning options of one plan (not in the reference: ark's MSM has no knobs).  Defaults come from the ZKMI_* environment
 * variables read when the plan is created; the run-time ones can be changed between runs of a live plan:
 *   "segment_lanes"     lanes the accumulate kernel aims at (<= the value at creation, >= 64)        ZKMI_SEG_LANES
 *   "sum_one_step"      1 = row / column sums of the bucket reduction in one launch                 ZKMI_SUM_ONE_STEP
 *   "lanes_per_output"  lanes per row / column sum of the one-step form (0 = automatic, 2 .. 64)    ZKMI_LPO
 *   "two_level_sort"    0 = one-level / bucket-range sorts only (windows <= 16 bits)                ZKMI_NO_TWO_LEVEL
 *   "priority_steps"    0 = the waves of the accumulate kernel keep one issue priority throughout   ZKMI_NO_PRIO_STEPS
 *                       (default 1: they step it down near the end of their segment, which keeps the waves of a SIMD together
 *                       when the kernel has the chip to itself; applies to zk_msm_plan_run / zk_msm_plan_enqueue only)
 *   "split_pairs"       Fp2 groups: 1 = the accumulate kernel gives a segment to a lane PAIR and splits every   ZKMI_SPLIT_PAIRS
 *                       Fp2 value by component over it (half the registers per lane), 0 = one lane per segment,
 *                       -1 = the group's default (BLS12-381 G2: on, BN254 G2: off; measured, msm_accumulate.hip.h)
 * Results never depend on them.  ZK_ERR_ARG for unknown names (including the creation-time options ZKMI_SORT_WGS,
 * ZKMI_FINE_LOG, ZKMI_NO_GLV, ZKMI_PRE_C), for values the plan cannot honour and while a run is in flight. */
int zk_msm_plan_set_option(uint64_t handle, const char* name, int64_t value);

/* ---- single-point host arithmetic (PointG1 / PointG2 methods, src/bn254/curve.rs:25-324) -- */

int zk_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* out);      /* __add__ */
int zk_point_neg(int curve, int group, const uint64_t* a, uint64_t* out);                          /* __neg__ */
/* sum of n affine points with one final inversion (the cross-rank combination of MSM partials) */
int zk_point_sum(int curve, int group, uint64_t n, const uint64_t* points, uint64_t* out);
int zk_point_mul(int curve, int group, const uint64_t* a, const uint64_t* scalar, uint64_t* out);  /* __mul__ */
int zk_point_on_curve(int curve, int group, const uint64_t* a);                                    /* 1 / 0 */
int zk_point_generator(int curve, int group, uint64_t* out);                                       /* g1() / g2() */
/* to_bytes / from_bytes = ark-serialize compressed (curve.rs:127-146): 32/64 B (BN254), 48/96 B (BLS12-381) */
int zk_point_compress(int curve, int group, const uint64_t* a, uint8_t* out);
int zk_point_decompress(int curve, int group, const uint8_t* in, uint64_t* out);

/* The same for the point vectors of a key file (python/zksnake/groth16/serialization.py:60-141 loops to_bytes /
 * from_hex over tau_1, tau_2, target_1, kdelta_1; plonk/serialization.py likewise): n points, one per GPU lane,
 * between host buffers of n * zk_point_limbs words and n * zk_point_bytes bytes.  Decoding validates like
 * zk_point_decompress (flags, x < p, on the curve, prime-order subgroup).  On ZK_ERR_POINT the message is that of the
 * FIRST offending point and *bad_index (may be NULL) its index. */
int zk_points_compress(int curve, int group, uint64_t n, const uint64_t* points, uint8_t* out, uint64_t* bad_index);
int zk_points_decompress(int curve, int group, uint64_t n, const uint8_t* in, uint64_t* out, uint64_t* bad_index);
int zk_point_bytes(int curve, int group);

/* ---- PlonK prover, device-resident vector kernels (csrc/plonk.hip) ----
 * They replace the element-wise polynomial arithmetic of Plonk.prove (python/zksnake/plonk/protocol.py:213-460: chains of
 * fft / mul_over_evaluation_domain / add_over_evaluation_domain, Polynomial scalar multiply-adds, Polynomial.__call__).
 * Vectors are canonical Fr elements in device memory; scalars are canonical 4-limb integers in host memory. */
/* out = a*x + b*y + c (element-wise; b and d_y may both be NULL, c may be NULL); out may alias x or y */
int zk_vec_axpby_dev(int curve, uint64_t n, const uint64_t* a, const void* d_x, const uint64_t* b, const void* d_y, const uint64_t* c,
                     void* d_out, void* stream);
/* acc[i] += sum_{t<k} scalars[t] * x_t[i] for i < counts[t] (counts[t] <= n_acc), then acc[at_index[j]] += at_vals[j] for j < n_at: a chain
 * of Polynomial scalar-multiply-adds and single-coefficient updates (plonk/protocol.py:319-402, :213-236 blinding) in ONE launch -- as
 * separate launches of a few microseconds each they leave the GPU idle between them.  k <= 16, n_at <= 8; scalars / at_vals are 4-limb
 * canonical integers in host memory, k * 4 and n_at * 4 limbs. */
int zk_vec_lincomb_dev(int curve, uint64_t n_acc, void* d_acc, int k, const uint64_t* counts, const void* const* d_x, const uint64_t* scalars,
                       int n_at, const uint64_t* at_index, const uint64_t* at_vals, void* stream);
/* dst[i] = src[offset + i*stride], i < n: one column of the flat witness [a0, b0, c0, a1, ...] (protocol.py:167-169) */
int zk_vec_gather_dev(int curve, uint64_t n, const void* d_src, uint64_t stride, uint64_t offset, void* d_dst, void* stream);
/* *is_zero = 1 when all n elements are zero (synchronises the stream) */
int zk_vec_is_zero_dev(int curve, uint64_t n, const void* d_x, int* is_zero, void* stream);
/* out = sum_i coeffs[i] x^i (Polynomial.__call__, src/bn254/polynomial.rs:491-516); synchronises the stream */
int zk_poly_eval_dev(int curve, uint64_t n, const void* d_coeffs, const uint64_t* x, uint64_t* out, void* stream);
/* k evaluations, polynomial i at xs[i] (4 limbs each) into outs[i], with a single synchronisation (the five opening values
 * and PI(zeta) of a PlonK proof, protocol.py:400-420) */
int zk_poly_eval_many_dev(int curve, int k, const uint64_t* counts, const void* const* d_coeffs, const uint64_t* xs, uint64_t* outs,
                          void* stream);
/* out[i] = prod_{j<3} (wires[j][i] + beta*labels[j][i] + gamma): numerator / denominator terms of the grand product
 * (protocol.py:270-292), labels = the identity or the sigma columns on the n-domain */
int zk_plonk_perm_terms_dev(int curve, uint64_t n, const void* const* d_wires, const void* const* d_labels, const uint64_t* beta,
                            const uint64_t* gamma, void* d_out, void* stream);
/* d_out (n + 1 elements): out[0] = 1, out[i+1] = out[i] * num[i] / den[i] -- the batch_modinv + accumulator loop of
 * protocol.py:296-307 as two product scans, one inversion and one element-wise pass.  ZK_ERR_ARG on a zero denominator. */
int zk_plonk_grand_product_dev(int curve, uint64_t n, const void* d_num, const void* d_den, void* d_out, void* stream);
/* coeffs (n) = q (n - 1 elements, device) * (X - root) + rem (host): Polynomial.__truediv__ by a linear factor
 * (src/bn254/polynomial.rs:404-438) as a geometric rescale, a suffix-sum scan and a rescale back; d_q may not alias d_coeffs */
int zk_poly_div_linear_dev(int curve, uint64_t n, const void* d_coeffs, const uint64_t* root, void* d_q, uint64_t* rem, void* stream);
/* Quotient evaluations on a coset of size m = k*n (2 <= k <= 16) (protocol.py:309-347 evaluated pointwise):
 * out = [gate + alpha*(prod(w_j + beta*k_j*x + gamma)*z - prod(w_j + beta*sigma_j + gamma)*z(omega x)) + alpha^2*(z - 1)*L1] / (x^n - 1)
 * d_cols = 15 vectors of m elements: a, b, c, z, PI, qL, qR, qO, qM, qC, sigma1, sigma2, sigma3, x (the coset points), L1;
 * z(omega x) is read from z at index (i + k) mod m; zh_inv = the k distinct values of 1/(x^n - 1) (host, canonical). */
int zk_plonk_quotient_dev(int curve, uint64_t m, uint64_t n, const void* const* d_cols, const uint64_t* zh_inv, const uint64_t* beta,
                          const uint64_t* gamma, const uint64_t* alpha, void* d_out, void* stream);

/* Dense-polynomial helpers over Fr used by the PlonK prover (python/zksnake/plonk/protocol.py:157-484); canonical
 * coefficients, lowest degree first, host memory.  They replace Polynomial.__call__ (src/bn254/polynomial.rs:491-516),
 * Polynomial.__truediv__ by a linear factor (:404-438), the batch_modinv + accumulator loop of protocol.py:296-307 and
 * Polynomial scalar-multiply-add chains (:319-402). */
int zk_fr_poly_eval(int curve, uint64_t n, const uint64_t* coeffs, const uint64_t* x, uint64_t* out);
/* coeffs (n) = q (n - 1 entries) * (X - root) + rem (1 entry) */
int zk_fr_poly_div_linear(int curve, uint64_t n, const uint64_t* coeffs, const uint64_t* root, uint64_t* q, uint64_t* rem);
/* out (n + 1 entries): out[0] = 1, out[i+1] = out[i] * num[i] / den[i]; ZK_ERR_ARG on a zero denominator */
int zk_fr_grand_product(int curve, uint64_t n, const uint64_t* num, const uint64_t* den, uint64_t* out);
/* acc[i] += s * x[i], i < n */
int zk_fr_scale_add(int curve, uint64_t n, uint64_t* acc, const uint64_t* x, const uint64_t* s);

/* pairing / multi_pairing (src/bn254/curve.rs:417-437): out = prod_i e(g1[i], g2[i]) after the final exponentiation,
 * as 12 base-field elements (coefficients of 1, w, .., w^5 over Fp2; zk_gt_limbs 64-bit limbs).  Host only. */
int zk_gt_limbs(int curve);
int zk_multi_pairing(int curve, uint64_t n, const uint64_t* g1_points, const uint64_t* g2_points, uint64_t* out);

/* ---- scalar-field host helpers (setup/verify side; polynomial.rs:518-533,636-652) -------- */

int zk_fr_root_of_unity(int curve, uint64_t n, uint64_t* out);                 /* get_evaluation_point(n, 1) */
int zk_fr_lagrange_coeffs(int curve, uint64_t n, const uint64_t* tau, uint64_t* out); /* evaluate_lagrange_coefficients */

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
