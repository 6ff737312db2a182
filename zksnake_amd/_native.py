"""
ctypes binding of libzkmi.so (the C ABI declared in include/zkmi.h).

This is the only place the Python host touches native code.  There is no CPU fallback:
when the library is missing `load()` raises, and when no GPU is visible every compute entry
point fails with ZkError (status ZK_ERR_HIP).
"""

import ctypes
import operator
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKMI_LIB") or os.path.join(_HERE, "libzkmi.so")  # ZKMI_LIB: kernel-variant experiments

ZK_OK = 0
ZK_ERR_LENGTH = 1
ZK_ERR_DOMAIN = 2
ZK_ERR_HIP = 3
ZK_ERR_POINT = 4
ZK_ERR_ARG = 5
ZK_ERR_NOT_DIVISIBLE = 6

CURVE_BN254 = 0
CURVE_BLS12_381 = 1
G1 = 1
G2 = 2

MSM_PRECOMPUTE = 1
MSM_HIGH_PRIORITY = 2
MSM_NO_GLV = 4
STREAM_PLAN = ctypes.c_void_p(-1)  # ZK_STREAM_PLAN: the plan's own stream

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_u64 = ctypes.c_uint64

# every symbol include/zkmi.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "zk_init": (_i, [_i]),
    "zk_init_ex": (_i, [_i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "zk_hw_queues_prepare": (_i, [ctypes.POINTER(_i)]),
    "zk_debug_spin_dev": (_i, [_vp, _u64]),
    "zk_shutdown": (_i, []),
    "zk_device_count": (_i, []),
    "zk_last_error": (ctypes.c_char_p, []),
    "zk_version": (ctypes.c_char_p, []),
    "zk_dev_alloc": (_i, [_u64, ctypes.POINTER(_vp)]),
    "zk_dev_free": (_i, [_vp]),
    "zk_dev_upload": (_i, [_vp, _vp, _u64]),
    "zk_dev_upload_async": (_i, [_vp, _vp, _u64, _vp]),
    "zk_host_alloc": (_i, [_u64, ctypes.POINTER(_vp)]),
    "zk_host_free": (_i, [_vp]),
    "zk_dev_download": (_i, [_vp, _vp, _u64]),
    "zk_dev_memset": (_i, [_vp, _i, _u64]),
    "zk_dev_memset_async": (_i, [_vp, _i, _u64, _vp]),
    "zk_dev_synchronize": (_i, []),
    "zk_stream_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "zk_stream_destroy": (_i, [_vp]),
    "zk_stream_synchronize": (_i, [_vp]),
    "zk_spmv_dev": (_i, [_i, _u64, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint32, _vp]),
    "zk_spmv_long_dev": (_i, [_i, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zk_fq_limbs": (_i, [_i]),
    "zk_point_limbs": (_i, [_i, _i]),
    "zk_ntt": (_i, [_i, _i, _i, _u64, _u64p, _u64, _u64p]),
    "zk_vec_op": (_i, [_i, _i, _u64, _u64, _u64p, _u64, _u64p, _u64p]),
    "zk_poly_div_vanishing": (_i, [_i, _u64, _u64, _u64p, _u64p, _u64p, ctypes.POINTER(_i)]),
    "zk_ntt_dev": (_i, [_i, _i, _i, _vp, _vp]),
    "zk_vec_op_dev": (_i, [_i, _i, _u64, _vp, _vp, _vp, _vp]),
    "zk_vec_canon_dev": (_i, [_i, _u64, _vp, _vp]),
    "zk_vec_powers_dev": (_i, [_i, _u64, _u64p, _vp, _vp]),
    "zk_qap_h_dev": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_i), _vp]),
    "zk_qap_h_dev_begin": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "zk_qap_h_dev_end": (_i, [_i, _i, _vp, ctypes.POINTER(_i), _vp]),
    "zk_qap_uv_dev": (_i, [_i, _i, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "zk_msm": (_i, [_i, _i, _u64, _u64, _u64p, _u64p, _u64p]),
    "zk_batch_mul": (_i, [_i, _i, _u64, _u64p, _u64p, _i, _u64p]),
    "zk_msm_plan_create": (_i, [_i, _i, _u64, _vp, _i, _i, _i, _u64p]),
    "zk_msm_plan_create_range": (_i, [_i, _i, _u64, _vp, _i, _i, _i, _i, _i, _u64p]),
    "zk_msm_plan_clone": (_i, [_u64, _u64p]),
    "zk_msm_plan_destroy": (_i, [_u64]),
    "zk_msm_plan_run": (_i, [_u64, _u64, _vp, _i, _i, _i, _u64p, _vp]),
    "zk_msm_plan_enqueue": (_i, [_u64, _u64, _vp, _i, _i, _i, _vp]),
    "zk_msm_plan_enqueue_sort": (_i, [_u64, _u64, _vp, _i, _i, _i, _vp]),
    "zk_msm_plan_enqueue_rest": (_i, [_u64, _u64]),
    "zk_msm_plan_wait_event": (_i, [_u64, _vp]),
    "zk_msm_plan_cancel": (_i, [_u64]),
    "zk_msm_plan_enqueue_shared": (_i, [_u64, _u64, _vp]),
    "zk_msm_plan_finish": (_i, [_u64, _u64p]),
    "zk_msm_plan_windows": (_i, [_u64, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "zk_msm_window_layout": (_i, [_i, _i, _u64, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "zk_msm_window_layout_ex": (_i, [_i, _i, _u64, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "zk_msm_plan_entries": (_i, [_u64, ctypes.POINTER(_u64)]),
    "zk_msm_plan_timings": (_i, [_u64, ctypes.POINTER(ctypes.c_float), _i]),
    "zk_msm_plan_set_option": (_i, [_u64, ctypes.c_char_p, ctypes.c_int64]),
    "zk_point_add": (_i, [_i, _i, _u64p, _u64p, _u64p]),
    "zk_point_neg": (_i, [_i, _i, _u64p, _u64p]),
    "zk_point_sum": (_i, [_i, _i, _u64, _u64p, _u64p]),
    "zk_point_mul": (_i, [_i, _i, _u64p, _u64p, _u64p]),
    "zk_point_on_curve": (_i, [_i, _i, _u64p]),
    "zk_point_generator": (_i, [_i, _i, _u64p]),
    "zk_point_compress": (_i, [_i, _i, _u64p, _u8p]),
    "zk_point_decompress": (_i, [_i, _i, _u8p, _u64p]),
    "zk_points_compress": (_i, [_i, _i, _u64, _u64p, _u8p, _u64p]),
    "zk_points_decompress": (_i, [_i, _i, _u64, _u8p, _u64p, _u64p]),
    "zk_point_bytes": (_i, [_i, _i]),
    "zk_vec_axpby_dev": (_i, [_i, _u64, _u64p, _vp, _u64p, _vp, _u64p, _vp, _vp]),
    "zk_vec_lincomb_dev": (_i, [_i, _u64, _vp, _i, _u64p, ctypes.POINTER(_vp), _u64p, _i, _u64p, _u64p, _vp]),
    "zk_vec_gather_dev": (_i, [_i, _u64, _vp, _u64, _u64, _vp, _vp]),
    "zk_vec_is_zero_dev": (_i, [_i, _u64, _vp, ctypes.POINTER(_i), _vp]),
    "zk_poly_eval_dev": (_i, [_i, _u64, _vp, _u64p, _u64p, _vp]),
    "zk_poly_eval_many_dev": (_i, [_i, _i, _u64p, ctypes.POINTER(_vp), _u64p, _u64p, _vp]),
    "zk_plonk_grand_product_dev": (_i, [_i, _u64, _vp, _vp, _vp, _vp]),
    "zk_poly_div_linear_dev": (_i, [_i, _u64, _vp, _u64p, _vp, _u64p, _vp]),
    "zk_plonk_perm_terms_dev": (_i, [_i, _u64, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _u64p, _u64p, _vp, _vp]),
    "zk_plonk_quotient_dev": (_i, [_i, _u64, _u64, ctypes.POINTER(_vp), _u64p, _u64p, _u64p, _u64p, _vp, _vp]),
    "zk_fr_poly_eval": (_i, [_i, _u64, _u64p, _u64p, _u64p]),
    "zk_fr_poly_div_linear": (_i, [_i, _u64, _u64p, _u64p, _u64p, _u64p]),
    "zk_fr_grand_product": (_i, [_i, _u64, _u64p, _u64p, _u64p]),
    "zk_fr_scale_add": (_i, [_i, _u64, _u64p, _u64p, _u64p]),
    "zk_gt_limbs": (_i, [_i]),
    "zk_multi_pairing": (_i, [_i, _u64, _u64p, _u64p, _u64p]),
    "zk_fr_root_of_unity": (_i, [_i, _u64, _u64p]),
    "zk_fr_lagrange_coeffs": (_i, [_i, _u64, _u64p, _u64p]),
}


class ZkError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libzkmi status {status}: {message}")
        self.status = status
        self.message = message


_lib = None
_lock = threading.Lock()
_initialised = False

QUEUES_SET_BY_LIBRARY, QUEUES_CALLER, QUEUES_TOO_LATE = 0, 1, 2
queue_status = None   # set by load(): what zk_hw_queues_prepare decided for this process (include/zkmi.h)


def _prepare_hw_queues_early():
    """The hardware-queue setting belongs to the C library (zk_hw_queues_prepare, include/zkmi.h), but the library itself is
    loaded lazily, on first use: loading it at import time would pull in /opt/rocm's libamdhip64 before a later
    `import torch` loads the copy bundled with torch by path -- two HIP runtimes in one process, and the second one finds no
    device.  So the import of this module only does what zk_hw_queues_prepare would do at that moment (same rule: the
    caller's setting wins, nothing is set once the runtime runs) and leaves a marker for the library to report it as its own."""
    if os.environ.get("GPU_MAX_HW_QUEUES"):
        return
    started = os.environ.get("ZKMI_TEST_RUNTIME_STARTED")
    if started is not None:
        running = started not in ("", "0")
    else:
        running = False
        try:
            for fd in os.listdir("/proc/self/fd"):
                try:
                    if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                        running = True
                        break
                except OSError:
                    continue
        except OSError:
            pass
    if not running:
        os.environ["GPU_MAX_HW_QUEUES"] = "12"
        os.environ["ZKMI_HW_QUEUES_SET_BY_LIBRARY"] = "1"


_prepare_hw_queues_early()


def load():
    """dlopen libzkmi.so and attach prototypes; raises if the library has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C zksnake_amd/csrc` (there is no CPU fallback)"
                )
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
            # hardware queues: the library sets GPU_MAX_HW_QUEUES itself, before its first HIP call, unless the caller chose
            # a value or the HIP runtime is already running (include/zkmi.h "Hardware queues").  No HIP call happens here.
            global queue_status
            queue_status = lib.zk_hw_queues_prepare(None)
            if queue_status == QUEUES_TOO_LATE:
                import warnings
                warnings.warn("the HIP runtime was started before libzkmi could set GPU_MAX_HW_QUEUES: the streams of a proof "
                              "may share hardware queues (slower, not wrong); import zksnake_amd before touching the GPU "
                              "or export GPU_MAX_HW_QUEUES=12", RuntimeWarning, stacklevel=2)
    return _lib


def check(status):
    if status != ZK_OK:
        msg = load().zk_last_error().decode("utf-8", "replace")
        raise ZkError(status, msg)


def ensure_gpu(device=None):
    """select the HIP device once per process; raises ZkError(ZK_ERR_HIP) when there is no GPU."""
    global _initialised
    lib = load()
    if not _initialised or device is not None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, lib.zk_device_count())
        check(lib.zk_init(device))
        _initialised = True
        global _device
        _device = device
    return lib


_device = 0


def bind_thread():
    """HIP's current device is per host thread: a worker thread calls this before its first library call"""
    check(load().zk_init(_device))


def u64p(arr):
    return arr.ctypes.data_as(_u64p)


def u8p(arr):
    return arr.ctypes.data_as(_u8p)


def curve_id(name):
    try:
        return {"BN254": 0, "BN128": 0, "ALT_BN128": 0, "BLS12_381": 1}[name]
    except KeyError:
        raise KeyError(name) from None


def fq_limbs(cid):
    return 4 if cid == 0 else 6


def point_limbs(cid, group):
    return 2 * fq_limbs(cid) * group


# ---- int <-> limb marshalling (the reference marshals Python ints through pyo3 BigUint) ----

try:  # CPython helper built beside libzkmi.so (csrc/pyints.c); marshalling only
    if os.environ.get("ZKMI_PYINTS"):   # another build of the helper (tools/sanitize_cpu.sh)
        import importlib.machinery
        import importlib.util
        _ldr = importlib.machinery.ExtensionFileLoader("zksnake_amd._pyints", os.environ["ZKMI_PYINTS"])
        _pyints = importlib.util.module_from_spec(importlib.util.spec_from_loader("zksnake_amd._pyints", _ldr))
        _ldr.exec_module(_pyints)
    else:
        from . import _pyints
except ImportError:  # pragma: no cover - e.g. a different interpreter than the one it was built for
    _pyints = None


PIPELINE_CHUNK = 1 << 18   # elements per chunk of a pipelined conversion (8 MiB of limbs: 0.16 ms on the link)


def ints_to_limbs(vals, words=4, modulus=None, out=None, chunk_done=None):
    """list of non-negative ints -> (n, words) uint64, little-endian limbs.  Negative ints raise
    OverflowError, as pyo3's BigUint extraction does in the reference; with `modulus` values are
    reduced first, which is what `Fr::from(BigUint)` does (src/bn254/curve.rs:359).
    `out`: a C-contiguous (n, words) uint64 array to fill (a reused / page-locked staging buffer).
    `chunk_done(begin, end)`: called after rows [begin, end) of `out` are final (long lists are converted in chunks of
    PIPELINE_CHUNK elements), so that the caller can start moving them while the rest is converted."""
    if out is not None:
        assert out.dtype == np.uint64 and out.flags.c_contiguous and out.shape == (len(vals), words)
    if modulus is not None:
        modulus = operator.index(modulus)   # an exact int for both paths (numpy integers welcome, floats are a TypeError)
        if modulus <= 0:
            raise ValueError("modulus must be a positive integer")
    if _pyints is not None:
        if not isinstance(vals, (list, tuple)):
            vals = list(vals)
        if out is None:
            out = np.empty((len(vals), words), dtype=np.uint64)
        if chunk_done is None or len(vals) < 2 * PIPELINE_CHUNK:
            _pyints.ints_to_limbs(vals, words, modulus, out)
            if chunk_done is not None:
                chunk_done(0, len(vals))
        else:
            # chunk by chunk WITHOUT slicing the list: the caller ships rows [b, e) while the next chunk is converted
            for b in range(0, len(vals), PIPELINE_CHUNK):
                e = min(len(vals), b + PIPELINE_CHUNK)
                _pyints.ints_to_limbs(vals, words, modulus, out, b, e)
                chunk_done(b, e)
        return out
    nbytes = 8 * words
    chunks = []
    for v in vals:
        v = operator.index(v)  # ints and numpy integers; floats raise TypeError as pyo3's extraction does
        if v < 0:
            raise OverflowError("can't convert negative int to unsigned")
        if modulus is not None and v >= modulus:
            v %= modulus
        chunks.append(v.to_bytes(nbytes, "little"))
    arr = np.frombuffer(b"".join(chunks), dtype=np.uint64).reshape(len(vals), words)
    if out is None:
        out = arr.copy()
    else:
        out[:] = arr
    if chunk_done is not None:
        chunk_done(0, len(vals))
    return out


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    words = arr.shape[-1]
    if _pyints is not None:
        return _pyints.limbs_to_ints(arr, words)
    raw = arr.tobytes()
    step = 8 * words
    return [int.from_bytes(raw[i:i + step], "little") for i in range(0, len(raw), step)]

