"""GPU: the stream / hardware-queue requirement is part of the C ABI (include/zkmi.h "Hardware queues"): the library sets
GPU_MAX_HW_QUEUES before its first HIP call, reports when it could not, and eight streams really run side by side."""

import os
import subprocess
import sys

import pytest

from zksnake_amd import _native as N

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = """
import ctypes, time, warnings
warnings.simplefilter("ignore")
{prelude}
from zksnake_amd import _native as N
lib = N.ensure_gpu(0)
streams = []
for _ in range(8):
    s = ctypes.c_void_p()
    N.check(lib.zk_stream_create(0, ctypes.byref(s)))
    streams.append(s)
N.check(lib.zk_debug_spin_dev(streams[0], 1000))          # warm-up: module load
N.check(lib.zk_dev_synchronize())
t0 = time.perf_counter()
for s in streams:
    N.check(lib.zk_debug_spin_dev(s, 20000))               # 20 ms each
N.check(lib.zk_dev_synchronize())
print(N.queue_status, round((time.perf_counter() - t0) * 1e3, 1))
"""


def _run(prelude, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "ZKMI_HW_QUEUES_SET_BY_LIBRARY")}
    env.update(env_extra)
    env["PYTHONPATH"] = ROOT
    res = subprocess.run([sys.executable, "-c", _PROBE.format(prelude=prelude)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    st, ms = res.stdout.split()
    return int(st), float(ms)


def test_eight_streams_overlap_with_the_library_setting_and_serialise_without():
    st, ms = _run("", {})
    assert st == N.QUEUES_SET_BY_LIBRARY
    assert ms < 45.0, f"8 x 20 ms spin kernels on 8 streams took {ms} ms: the streams share hardware queues"
    st2, ms2 = _run("", {"GPU_MAX_HW_QUEUES": "2"})
    assert st2 == N.QUEUES_CALLER
    assert ms2 > 70.0, f"two hardware queues should serialise 8 spin kernels into four rounds, took {ms2} ms"


def test_runtime_started_by_torch_first_is_reported():
    st, _ = _run("import torch; torch.cuda.init(); torch.zeros(1, device='cuda')", {})
    assert st == N.QUEUES_TOO_LATE
    # importing torch alone does not start the runtime: the library is still in time
    st, ms = _run("import torch", {})
    assert st == N.QUEUES_SET_BY_LIBRARY and ms < 45.0
