"""
oracle/corc.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libzk_oracle.so
(the C++ CPU restatement, oracle/zk_oracle.cpp).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CURVE_ID = {"BN254": 0, "BN128": 0, "ALT_BN128": 0, "BLS12_381": 1}
FQ_WORDS = {0: 4, 1: 6}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ZK_ORACLE_LIB") or os.path.join(_HERE, "libzk_oracle.so")   # ZK_ORACLE_LIB: sanitizer build
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        _LIB.orc_msm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p, u64p, u64p, ctypes.c_int, ctypes.c_int]
        _LIB.orc_batch_mul.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p, u64p, ctypes.c_int, u64p, ctypes.c_int]
        _LIB.orc_point_add.argtypes = [ctypes.c_int, ctypes.c_int, u64p, u64p, u64p]
        _LIB.orc_ntt.argtypes = [ctypes.c_int, u64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _LIB.orc_ntt_spot.argtypes = [ctypes.c_int, u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p, ctypes.c_int]
        _LIB.orc_dot.argtypes = [ctypes.c_int, ctypes.c_uint64, u64p, u64p, u64p, ctypes.c_int]
        _LIB.orc_vec_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p, u64p, u64p]
        _LIB.orc_qap_h.argtypes = [ctypes.c_int, ctypes.c_int, u64p, u64p, u64p, u64p, u64p, u64p, ctypes.c_int]
        _LIB.orc_ark_window.argtypes = [ctypes.c_uint64]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def point_words(curve_id, group):
    return 2 * FQ_WORDS[curve_id] * group


# ---- int <-> limb helpers (canonical little-endian 64-bit limbs) ----

def ints_to_limbs(vals, words=4):
    buf = b"".join(int(v).to_bytes(8 * words, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), words).copy()


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    words = arr.shape[-1]
    raw = arr.tobytes()
    step = 8 * words
    return [int.from_bytes(raw[i:i + step], "little") for i in range(0, len(raw), step)]


def points_to_limbs(points, curve_id, group):
    """oracle.pyref affine points (None = infinity) -> (n, point_words) uint64."""
    w = FQ_WORDS[curve_id]
    rows = []
    for P in points:
        if P is None:
            rows.append(b"\0" * (8 * w * 2 * group))
        elif group == 1:
            rows.append(P[0].to_bytes(8 * w, "little") + P[1].to_bytes(8 * w, "little"))
        else:
            rows.append(b"".join(c.to_bytes(8 * w, "little") for c in (P[0][0], P[0][1], P[1][0], P[1][1])))
    return np.frombuffer(b"".join(rows), dtype=np.uint64).reshape(len(points), 2 * w * group).copy()


def limbs_to_points(arr, curve_id, group):
    w = FQ_WORDS[curve_id]
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 2 * w * group)
    out = []
    for row in arr:
        c = limbs_to_ints(row.reshape(2 * group, w))
        if not any(c):
            out.append(None)
        elif group == 1:
            out.append((c[0], c[1]))
        else:
            out.append(((c[0], c[1]), (c[2], c[3])))
    return out


# ---- operations ----

def msm(curve_id, group, scalars, bases, threads=1, c=0):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    n = scalars.shape[0]
    assert bases.shape[0] == n
    out = np.zeros(point_words(curve_id, group), dtype=np.uint64)
    rc = lib().orc_msm(curve_id, group, n, _p(scalars), _p(bases), _p(out), threads, c)
    assert rc == 0
    return out


def batch_mul(curve_id, group, scalars, bases, threads=8):
    """out[i] = scalars[i]*bases[i]; a single base row is broadcast."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    n = scalars.shape[0]
    fixed = 1 if bases.ndim == 1 or bases.shape[0] == 1 and n != 1 else 0
    out = np.zeros((n, point_words(curve_id, group)), dtype=np.uint64)
    rc = lib().orc_batch_mul(curve_id, group, n, _p(scalars), _p(bases), fixed, _p(out), threads)
    assert rc == 0
    return out


def point_add(curve_id, group, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(point_words(curve_id, group), dtype=np.uint64)
    assert lib().orc_point_add(curve_id, group, _p(a), _p(b), _p(out)) == 0
    return out


def ntt(curve_id, data, inverse=False, threads=1):
    data = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 4)
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    rc = lib().orc_ntt(curve_id, _p(data), log_n, 1 if inverse else 0, threads)
    if rc == 2:
        raise ValueError("Domain size is too large")
    assert rc == 0
    return data


def ntt_spot(curve_id, data, k, inverse=False, threads=1):
    """output k of the transform of `data` (canonical (n, 4) limbs, NOT copied), straight from the definition: O(n)"""
    data = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4)
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().orc_ntt_spot(curve_id, _p(data), log_n, 1 if inverse else 0, k, _p(out), threads)
    if rc == 2:
        raise ValueError("Domain size is too large")
    assert rc == 0
    return out


def dot(curve_id, a, b, threads=1):
    """sum a_i b_i mod r of two canonical (n, 4) limb arrays, as a Python int"""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    assert a.shape == b.shape
    out = np.zeros(4, dtype=np.uint64)
    assert lib().orc_dot(curve_id, a.shape[0], _p(a), _p(b), _p(out), threads) == 0
    return limbs_to_ints(out.reshape(1, 4))[0]


def vec_op(curve_id, op, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(a)
    assert lib().orc_vec_op(curve_id, {"mul": 0, "add": 1, "sub": 2}[op], a.shape[0], _p(a), _p(b), _p(out)) == 0
    return out


def qap_h(curve_id, a, b, c, threads=1):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    c = np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4)
    n = a.shape[0]
    log_n = n.bit_length() - 1
    u = np.zeros_like(a); v = np.zeros_like(a); h = np.zeros_like(a)
    rc = lib().orc_qap_h(curve_id, log_n, _p(a), _p(b), _p(c), _p(u), _p(v), _p(h), threads)
    if rc == 1:
        raise ValueError("(U * V - W) did not divided by Z to zero")
    assert rc == 0
    return u, v, h


def ark_window(n):
    return lib().orc_ark_window(n)
