import sys, time
sys.path.insert(0, '.')
import numpy as np
from zksnake_amd import _native as N, workloads as W
lib = N.ensure_gpu()
cid, grp = 0, 1
for log_n in (1, 10, 16, 20):
    n = 1 << log_n
    r = W.scalar_field("BN254")
    sc = W.splitmix64(5, 4 * n).reshape(n, 4); sc[:, 3] &= np.uint64((1 << 60) - 1)
    ks = W.splitmix64(6, 4 * n).reshape(n, 4); ks[:, 3] &= np.uint64((1 << 60) - 1)
    gen = np.zeros(8, dtype=np.uint64); N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
    bases = np.zeros((n, 8), dtype=np.uint64)
    N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
    out = np.zeros(8, dtype=np.uint64)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        N.check(lib.zk_msm(cid, grp, n, n, N.u64p(sc), N.u64p(bases), N.u64p(out)))
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"zk_msm one-shot 2^{log_n}: {[round(x, 2) for x in ts]} ms")
from zksnake_amd.ecc import EllipticCurve
E = EllipticCurve("BN254")
pts = [E.G1() * (i + 1) for i in range(4)]
for _ in range(3):
    t = time.perf_counter(); E.multiexp(pts, [5, 6, 7, 8]); print("multiexp(4 points lists)", round((time.perf_counter() - t) * 1e3, 2), "ms")
