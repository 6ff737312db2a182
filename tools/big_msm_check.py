#!/usr/bin/env python3
"""Large-size MSM check against the closed form (sum s_i k_i) G:  python tools/big_msm_check.py CURVE GROUP LOG_N [precompute]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from zksnake_amd import _native as N, workloads as W

curve, grp, log_n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
flags = N.MSM_PRECOMPUTE if len(sys.argv) > 4 else 0
cid = N.curve_id(curve)
r = W.scalar_field(curve)
n = 1 << log_n
lib = N.ensure_gpu()
PW = N.point_limbs(cid, grp)
sc, sc_i = W.field_stream(W.SEED_MSM_SCALARS, n, r)
k, k_i = W.field_stream(W.SEED_MSM_BASES, n, r)
gen = np.zeros(PW, dtype=np.uint64); N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
t = time.time()
bases = np.zeros((n, PW), dtype=np.uint64)
N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(k), N.u64p(gen), 1, N.u64p(bases)))
t_b = time.time() - t
dot = sum(a * b for a, b in zip(sc_i, k_i)) % r
exp = np.zeros(PW, dtype=np.uint64)
N.check(lib.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
h = N._u64(0)
t = time.time()
N.check(lib.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
t_p = time.time() - t
out = np.zeros(PW, dtype=np.uint64)
times = []
for _ in range(4):
    t = time.perf_counter()
    N.check(lib.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
    times.append((time.perf_counter() - t) * 1e3)
tm = (N.ctypes.c_float * 5)(); lib.zk_msm_plan_timings(h, tm, 5)
print(f"{curve} G{grp} 2^{log_n} precompute={bool(flags)}: match={bool((out == exp).all())} batch_mul {t_b:.2f}s plan {t_p:.2f}s "
      f"run(ms, incl. scalar upload) {[round(x, 2) for x in times]} stages(ms) {[round(x, 3) for x in tm]}")
sys.exit(0 if (out == exp).all() else 1)
