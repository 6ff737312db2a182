#!/usr/bin/env python3
"""Stage timings and the integer multiply-add roofline of the MSM for every group:
  python tools/group_msm_bench.py [LOG_N] [groups e.g. bn1,bn2,bls1,bls2] [pre]
Per group: one plan at 2^LOG_N pairs (general mode, or fixed-base with `pre`), scalars resident in HBM, result checked
against the closed form (sum s_i k_i) G, stage times from the library's HIP events (zk_msm_plan_timings)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from zksnake_amd import _native as N, workloads as W  # noqa: E402
from zksnake_amd.device import DeviceBuffer  # noqa: E402

# v_mad_u64_u32 per field operation for N 29-bit limbs: product 2N^2, squaring N(N+1)/2+N^2, double product 3N^2
# (the N reduction quotients of each are v_mul_lo_u32 and are not counted)
def _mads(nl):
    return 2 * nl * nl, nl * (nl + 1) // 2 + nl * nl, 3 * nl * nl

def mads_per_mixed_add(cid, grp):
    mul, sqr, mul2 = _mads(9 if cid == 0 else 14)
    if grp == 1:
        return 6 * mul + 2 * sqr + mul2            # six products, two squarings, Y3 as one double product
    nl = 9 if cid == 0 else 14
    y3 = 2 * 5 * nl * nl if nl <= 9 else 2 * (2 * mul2)   # nine limbs: one four-product reduction per component
    return 6 * (2 * mul2) + 2 * (2 * mul) + y3     # Fp2: product = two double products, squaring = two products

MAD_PEAK_T = 33.2   # measured v_mad_u64_u32 rate, see bench.py
BYTES = {(0, 1): 96, (0, 2): 160, (1, 1): 128, (1, 2): 224}   # scalar + affine base per pair
NAMES = {"bn1": ("BN254", 1), "bn2": ("BN254", 2), "bls1": ("BLS12_381", 1), "bls2": ("BLS12_381", 2)}

def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else list(NAMES)
    flags = N.MSM_PRECOMPUTE if len(sys.argv) > 3 and sys.argv[3] == "pre" else 0
    lib = N.ensure_gpu()
    n = 1 << log_n
    out_all = {}
    for key in which:
        curve, grp = NAMES[key]
        cid = N.curve_id(curve)
        r = W.scalar_field(curve)
        PW = N.point_limbs(cid, grp)
        sc = W.splitmix64(W.SEED_MSM_SCALARS + cid, 4 * n).reshape(n, 4)
        ks = W.splitmix64(W.SEED_MSM_BASES + cid, 4 * n).reshape(n, 4)
        sc[:, 3] &= np.uint64((1 << 60) - 1)
        ks[:, 3] &= np.uint64((1 << 60) - 1)
        gen = np.zeros(PW, dtype=np.uint64)
        N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
        bases = np.zeros((n, PW), dtype=np.uint64)
        N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
        from zksnake_amd.frvec import DevVec, FrOps
        V = FrOps(r)
        d_s, d_k, prod = V.d_from(sc), V.d_from(ks), DevVec(n, zero=False)
        V.d_mul(n, d_s.ptr(), d_k.ptr(), prod.ptr())
        dot = V.d_eval(n, prod.ptr(), 1)
        exp = np.zeros(PW, dtype=np.uint64)
        N.check(lib.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
        h = N._u64(0)
        t0 = time.perf_counter()
        N.check(lib.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
        lib.zk_dev_synchronize()
        plan_s = time.perf_counter() - t0
        cb, nw = N._i(0), N._i(0)
        N.check(lib.zk_msm_plan_windows(h, cb, nw))
        ent = N._u64(0)
        N.check(lib.zk_msm_plan_entries(h, ent))
        res = np.zeros(PW, dtype=np.uint64)
        tm = (N.ctypes.c_float * 5)()
        for _ in range(2):
            N.check(lib.zk_msm_plan_run(h, n, d_s.ptr(), 1, 0, 0, N.u64p(res), None))
        ok = bool((res == exp).all())
        reps, stages = 8, []
        t0 = time.perf_counter()
        for _ in range(reps):
            N.check(lib.zk_msm_plan_run(h, n, d_s.ptr(), 1, 0, 0, N.u64p(res), None))
            lib.zk_msm_plan_timings(h, tm, 5)
            stages.append(list(tm))
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = np.array(stages).mean(axis=0)
        acc_s = st[1] * 1e-3
        mads = nw.value * ent.value * mads_per_mixed_add(cid, grp)
        out_all[key] = {"curve": curve, "group": grp, "log_n": log_n, "precompute": bool(flags), "match": ok, "ms": round(ms, 4),
                        "Mscalar/s": round(n / ms / 1e3, 2), "window_bits": cb.value, "windows": nw.value, "plan_create_s": round(plan_s, 3),
                        "stage_ms": {"digits_sort": round(float(st[0]), 4), "accumulate": round(float(st[1]), 4),
                                     "reduce": round(float(st[2]), 4), "host_tail": round(float(st[3]), 4)},
                        "roofline": {"bound": "hbm", "achieved": round(BYTES[(cid, grp)] * n / acc_s / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(BYTES[(cid, grp)] * n / acc_s / 1e9 / 8000.0, 5)},
                        "roofline_valu": {"achieved": round(mads / acc_s / 1e12, 3), "peak": MAD_PEAK_T, "unit": "Tmad/s",
                                          "frac": round(mads / acc_s / 1e12 / MAD_PEAK_T, 4)}}
        print(key, json.dumps(out_all[key]), flush=True)
        N.check(lib.zk_msm_plan_destroy(h))
        if not ok:
            sys.exit(1)

if __name__ == "__main__":
    main()
