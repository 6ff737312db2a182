import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from zksnake_amd import _native as N
from zksnake_amd.device import DeviceBuffer, PinnedArray
lib = N.ensure_gpu()
n = 1 << 20
a = np.random.default_rng(1).integers(0, 2**62, size=(n, 4), dtype=np.uint64)
d = DeviceBuffer(n * 32)
pin = PinnedArray((n, 4)); pin.array[:] = a
for name, src in (("pageable", a), ("pinned", pin.array)):
    ts = []
    for _ in range(8):
        lib.zk_dev_synchronize()
        t0 = time.perf_counter(); d.upload(src); lib.zk_dev_synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, [round(t, 3) for t in ts])
