"""experiment: one 2^20 general-path MSM as k concurrent window-range runs on clones of the plan (own workspace and stream each)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from zksnake_amd import _native as N, workloads as W
from zksnake_amd.parallel import sum_points, window_ranges
lib = N.ensure_gpu()
cid, grp, n = 0, 1, 1 << 20
r = W.scalar_field("BN254")
sc, sc_i = W.field_stream(W.SEED_MSM_SCALARS, n, r)
k, k_i = W.field_stream(W.SEED_MSM_BASES, n, r)
gen = np.zeros(8, dtype=np.uint64); N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
bases = np.zeros((n, 8), dtype=np.uint64)
N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(k), N.u64p(gen), 1, N.u64p(bases)))
dot = sum(a * b for a, b in zip(sc_i, k_i)) % r
exp = np.zeros(8, dtype=np.uint64)
N.check(lib.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
from zksnake_amd.device import DeviceBuffer
d_sc = DeviceBuffer.from_numpy(sc)
h = N._u64(0)
N.check(lib.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, 0, 0, h))
c, nw = N._i(0), N._i(0); N.check(lib.zk_msm_plan_windows(h, c, nw))
for parts in (1, 2, 4):
    hs = [h.value]
    for _ in range(parts - 1):
        hc = N._u64(0); N.check(lib.zk_msm_plan_clone(h, hc)); hs.append(hc.value)
    rng = window_ranges(nw.value, parts)
    outs = [np.zeros(8, dtype=np.uint64) for _ in range(parts)]
    def run():
        if os.environ.get("CHAIN"):
            for hh, (f, cnt) in zip(hs, rng):
                N.check(lib.zk_msm_plan_enqueue_sort(hh, n, d_sc.ptr, 1, f, cnt, N.STREAM_PLAN))
            prev = 0
            for hh in hs:
                N.check(lib.zk_msm_plan_enqueue_rest(hh, prev))
                prev = hh
        else:
          for hh, (f, cnt) in zip(hs, rng):
            N.check(lib.zk_msm_plan_enqueue(hh, n, d_sc.ptr, 1, f, cnt, N.STREAM_PLAN))
        for hh, o in zip(hs, outs):
            N.check(lib.zk_msm_plan_finish(hh, N.u64p(o)))
        return outs[0] if parts == 1 else sum_points(cid, grp, outs)
    for _ in range(3): res = run()
    assert (res == exp).all()
    lib.zk_dev_synchronize(); t = time.perf_counter()
    for _ in range(20): res = run()
    ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"{parts} concurrent window-range runs: {ms:.3f} ms per MSM")
    for hh in hs[1:]: N.check(lib.zk_msm_plan_destroy(hh))
