#!/usr/bin/env python3
"""Per-rank cost of the window-sharded MSM on ONE GPU: the 2^20 BN254 G1 plan run over all / a half / a quarter / an
eighth of its windows (what a rank does at N = 1 / 2 / 4 / 8), stage breakdown from the library's HIP events.
usage: window_range_bench.py [comma-separated window counts] [plan flags, e.g. 4 = ZK_MSM_NO_GLV]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np  # noqa: E402

from zksnake_amd import _native as N  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.device import DeviceBuffer  # noqa: E402

lib = N.ensure_gpu()
cid, grp, n = 0, 1, 1 << 20
sc = W.splitmix64(1, 4 * n).reshape(n, 4)
ks = W.splitmix64(2, 4 * n).reshape(n, 4)
sc[:, 3] &= np.uint64((1 << 60) - 1)
ks[:, 3] &= np.uint64((1 << 60) - 1)
gen = np.zeros(8, dtype=np.uint64)
N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
bases = np.zeros((n, 8), dtype=np.uint64)
N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
h = N._u64(0)
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N.check(lib.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
cb, nw = N._i(0), N._i(0)
N.check(lib.zk_msm_plan_windows(h, cb, nw))
print(f"plan: {nw.value} windows of {cb.value} bits")
d = DeviceBuffer.from_numpy(sc)
out = np.zeros(8, dtype=np.uint64)
tm = (N.ctypes.c_float * 5)()
for wc in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] else [max(1, nw.value >> k) for k in range(4)]):
    for _ in range(3):
        N.check(lib.zk_msm_plan_run(h, n, d.ptr, 1, 0, wc, N.u64p(out), None))
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        N.check(lib.zk_msm_plan_run(h, n, d.ptr, 1, 0, wc, N.u64p(out), None))
    ms = (time.perf_counter() - t0) / reps * 1e3
    lib.zk_msm_plan_timings(h, tm, 5)
    print(f"windows={wc:2d}  {ms:.3f} ms  stages(ms) sort={tm[0]:.3f} acc={tm[1]:.3f} reduce={tm[2]:.3f} tail={tm[3]:.3f}")
