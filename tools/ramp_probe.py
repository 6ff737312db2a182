#!/usr/bin/env python3
"""How long the headline MSM takes as a function of how many have run back to back since the GPU was idle (clock / power state ramp):
prints the accumulate kernel's time and the step's wall time for steps 1, 2, 3, 5, 8, 13, ... of one burst.  usage: ramp_probe.py [idle seconds]"""
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
N.check(lib.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, 0, 0, h))
d = DeviceBuffer.from_numpy(sc)
out = np.zeros(8, dtype=np.uint64)
tm = (N.ctypes.c_float * 5)()
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for burst in range(2):
    time.sleep(idle)
    rows = []
    for step in range(1, 301):
        t0 = time.perf_counter()
        N.check(lib.zk_msm_plan_run(h, n, d.ptr, 1, 0, 0, N.u64p(out), None))
        wall = (time.perf_counter() - t0) * 1e3
        lib.zk_msm_plan_timings(h, tm, 5)
        rows.append((step, wall, tm[1], tm[0], tm[2], tm[3]))
    marks = [1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 300]
    print(f"burst {burst} after {idle} s idle:", "  ".join(f"#{s}: {rows[s - 1][1]:.3f}/{rows[s - 1][2]:.3f}" for s in marks))
    if burst == 1:
        print("steady state, steps 200-223 (wall / sort / accumulate / reduce / tail):")
        for s in range(200, 224):
            r = rows[s - 1]
            print(f"  #{s}: {r[1]:.3f}  {r[3]:.3f} {r[2]:.3f} {r[4]:.3f} {r[5]:.3f}")
