"""
Vectors of scalar-field elements as (n, 4) uint64 limb arrays (canonical, little-endian) and the
library calls the PlonK prover needs on them: GPU transforms and element-wise ops (zk_ntt, zk_vec_op),
host O(n) recurrences (zk_fr_poly_eval / _div_linear / _grand_product / _scale_add).  Python integers
appear only for single field elements (challenges, evaluations), never per coefficient.
"""

import ctypes

import numpy as np

from . import _native as N
from .constant import BLS12_381_SCALAR_FIELD, BN254_SCALAR_FIELD
from .device import DeviceBuffer

_CID = {BN254_SCALAR_FIELD: N.CURVE_BN254, BLS12_381_SCALAR_FIELD: N.CURVE_BLS12_381}


# Released vectors go back to a size-keyed pool instead of hipFree (which synchronises the device, ~0.2 ms each, and a
# proof allocates a few dozen vectors of the same handful of sizes).  All vector kernels are issued on the NULL stream,
# so reuse is stream-ordered and safe.
_POOL = {}
_POOL_BYTES = [0]
_POOL_LIMIT = 32 << 30


def release_pool():
    """free every pooled device buffer"""
    for bufs in _POOL.values():
        for buf in bufs:
            buf.free()
    _POOL.clear()
    _POOL_BYTES[0] = 0


class DevVec:
    """n canonical Fr elements in HBM"""

    def __init__(self, n, zero=True):
        self.n = n
        nbytes = 32 * max(n, 1)
        bufs = _POOL.get(nbytes)
        if bufs:
            self.buf = bufs.pop()
            _POOL_BYTES[0] -= nbytes
        else:
            self.buf = DeviceBuffer(nbytes)
        if zero:
            self.buf.zero()

    def __del__(self):
        try:
            buf, self.buf = self.buf, None
            if buf is None or buf.ptr is None:
                return
            if _POOL_BYTES[0] + buf.nbytes <= _POOL_LIMIT:
                _POOL.setdefault(buf.nbytes, []).append(buf)
                _POOL_BYTES[0] += buf.nbytes
            else:
                buf.free()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def ptr(self, offset=0):
        assert 0 <= offset <= self.n
        return self.buf.ptr + 32 * offset

    def upload(self, limbs, offset=0):
        assert offset + limbs.shape[0] <= self.n
        self.buf.upload(limbs, 32 * offset)

    def download(self, count=None, offset=0):
        count = self.n - offset if count is None else count
        return self.buf.download((count, 4), np.uint64, 32 * offset)


class FrOps:
    def __init__(self, modulus):
        self.r = modulus
        self.cid = _CID[modulus]

    # -- conversions --
    def limbs(self, values):
        """list[int] | (n, 4) uint64 -> (n, 4) uint64 (ints reduced mod r)"""
        if isinstance(values, np.ndarray):
            return np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        return N.ints_to_limbs(values, 4, self.r)

    def one(self, value):
        return N.ints_to_limbs([value % self.r], 4)

    def ints(self, arr):
        return N.limbs_to_ints(arr)

    def int_at(self, arr, i):
        return int.from_bytes(arr[i].tobytes(), "little")

    def add_at(self, arr, i, value):
        """arr[i] += value (mod r), in place"""
        arr[i] = np.frombuffer(((self.int_at(arr, i) + value) % self.r).to_bytes(32, "little"), dtype=np.uint64)

    def const(self, value, n):
        return np.tile(self.one(value), (n, 1))

    def zeros(self, n):
        return np.zeros((n, 4), dtype=np.uint64)

    @staticmethod
    def strip(arr):
        """drop trailing zero coefficients"""
        nz = np.flatnonzero(arr.any(axis=1))
        return arr[: (int(nz[-1]) + 1 if nz.size else 0)]

    # -- GPU --
    def ntt(self, vals, size, inverse=False, coset=False):
        lib = N.ensure_gpu()
        a = self.limbs(vals)
        n = 1 if size <= 1 else 1 << (size - 1).bit_length()
        out = np.zeros((n, 4), dtype=np.uint64)
        st = lib.zk_ntt(self.cid, int(inverse), int(coset), a.shape[0], N.u64p(a), size, N.u64p(out))
        if st == N.ZK_ERR_DOMAIN:
            raise ValueError("Domain size is too large")
        N.check(st)
        return out

    def _vec(self, op, a, b):
        lib = N.ensure_gpu()
        assert a.shape == b.shape, "element-wise operands differ in length"
        out = np.zeros_like(a)
        if a.shape[0]:
            N.check(lib.zk_vec_op(self.cid, op, a.shape[0], a.shape[0], N.u64p(a), b.shape[0], N.u64p(b), N.u64p(out)))
        return out

    def mul(self, a, b):
        return self._vec(0, a, b)

    def add(self, a, b):
        return self._vec(1, a, b)

    def sub(self, a, b):
        return self._vec(2, a, b)

    def powers(self, g, count):
        """(1, g, g^2, ..) by doubling: log2(count) element-wise products"""
        pw = self.limbs([1])
        while pw.shape[0] < count:
            k = pw.shape[0]
            pw = np.concatenate([pw, self.mul(pw, self.const(pow(g, k, self.r), k))])
        return np.ascontiguousarray(pw[:count])

    # -- host recurrences --
    def eval(self, coeffs, x):
        lib = N.load()
        out = np.zeros(4, dtype=np.uint64)
        c = np.ascontiguousarray(coeffs)
        N.check(lib.zk_fr_poly_eval(self.cid, c.shape[0], N.u64p(c), N.u64p(self.one(x)), N.u64p(out)))
        return int.from_bytes(out.tobytes(), "little")

    def div_linear(self, coeffs, root):
        """(quotient limbs, remainder int) of coeffs / (X - root)"""
        lib = N.load()
        c = np.ascontiguousarray(coeffs)
        q = np.zeros((max(c.shape[0] - 1, 1), 4), dtype=np.uint64)
        rem = np.zeros(4, dtype=np.uint64)
        N.check(lib.zk_fr_poly_div_linear(self.cid, c.shape[0], N.u64p(c), N.u64p(self.one(root)), N.u64p(q), N.u64p(rem)))
        return q[: max(c.shape[0] - 1, 0)], int.from_bytes(rem.tobytes(), "little")

    def grand_product(self, num, den):
        lib = N.load()
        out = np.zeros((num.shape[0] + 1, 4), dtype=np.uint64)
        N.check(lib.zk_fr_grand_product(self.cid, num.shape[0], N.u64p(np.ascontiguousarray(num)),
                                        N.u64p(np.ascontiguousarray(den)), N.u64p(out)))
        return out

    def scale_add(self, acc, x, s):
        """acc[:len(x)] += s * x, in place (acc must be at least as long as x)"""
        lib = N.load()
        assert acc.shape[0] >= x.shape[0] and acc.flags["C_CONTIGUOUS"]
        if x.shape[0]:
            N.check(lib.zk_fr_scale_add(self.cid, x.shape[0], N.u64p(acc), N.u64p(np.ascontiguousarray(x)), N.u64p(self.one(s))))
        return acc

    # -- device-resident forms (vectors stay in HBM between calls; only scalars and a few coefficients move) --
    def d_from(self, limbs, size=None):
        """host limbs -> DevVec of `size` elements (zero padded)"""
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4)
        vec = DevVec(size or limbs.shape[0], zero=bool(size and size > limbs.shape[0]))
        if limbs.shape[0]:
            vec.upload(limbs)
        return vec

    def d_powers(self, g, count):
        """DevVec of (1, g, g^2, ..), computed on the device"""
        vec = DevVec(count, zero=False)
        N.check(N.load().zk_vec_powers_dev(self.cid, count, N.u64p(self.one(g)), vec.ptr(), None))
        return vec

    def d_ntt(self, vec, size, inverse=False):
        """in place on the first `size` (a power of two) elements"""
        assert size & (size - 1) == 0 and size <= vec.n
        N.check(N.load().zk_ntt_dev(self.cid, int(inverse), size.bit_length() - 1, vec.ptr(), None))

    def d_op(self, op, n, a_ptr, b_ptr, out_ptr):
        N.check(N.load().zk_vec_op_dev(self.cid, op, n, a_ptr, b_ptr, out_ptr, None))

    def d_mul(self, n, a_ptr, b_ptr, out_ptr):
        self.d_op(0, n, a_ptr, b_ptr, out_ptr)

    def d_add(self, n, a_ptr, b_ptr, out_ptr):
        self.d_op(1, n, a_ptr, b_ptr, out_ptr)

    def d_axpy(self, n, acc_ptr, s, x_ptr):
        """acc[:n] += s * x[:n]"""
        N.check(N.load().zk_vec_axpby_dev(self.cid, n, N.u64p(self.one(1)), acc_ptr, N.u64p(self.one(s)), x_ptr, None, acc_ptr, None))

    def d_lincomb(self, acc, terms=(), at=()):
        """acc (a DevVec) += sum of s * x[:count] over terms [(count, s, device pointer)], then acc[i] += v over at [(i, v)]: one
        launch for the whole chain (zk_vec_lincomb_dev: at most 16 terms and 8 single updates per call; longer lists are cut)"""
        lib = N.load()
        terms, at = list(terms), list(at)
        while terms or at:
            tk, terms = terms[:16], terms[16:]
            ak, at = at[:8], at[8:]
            k, n_at = len(tk), len(ak)
            counts = np.array([t[0] for t in tk], dtype=np.uint64)
            ptrs = (N._vp * max(k, 1))(*[t[2] for t in tk])
            scalars = np.concatenate([self.one(t[1] % self.r) for t in tk]).reshape(k, 4) if k else np.zeros((1, 4), dtype=np.uint64)
            idx = np.array([a[0] for a in ak], dtype=np.uint64)
            vals = np.concatenate([self.one(a[1] % self.r) for a in ak]).reshape(n_at, 4) if n_at else np.zeros((1, 4), dtype=np.uint64)
            N.check(lib.zk_vec_lincomb_dev(self.cid, acc.n, acc.ptr(), k, N.u64p(counts) if k else None, ptrs if k else None,
                                           N.u64p(scalars) if k else None, n_at, N.u64p(idx) if n_at else None,
                                           N.u64p(vals) if n_at else None, None))

    def d_is_zero(self, n, ptr):
        flag = ctypes.c_int(0)
        N.check(N.load().zk_vec_is_zero_dev(self.cid, n, ptr, ctypes.byref(flag), None))
        return bool(flag.value)

    def d_eval(self, n, ptr, x):
        out = np.zeros(4, dtype=np.uint64)
        N.check(N.load().zk_poly_eval_dev(self.cid, n, ptr, N.u64p(self.one(x)), N.u64p(out), None))
        return int.from_bytes(out.tobytes(), "little")

    def d_eval_many(self, jobs):
        """[(n, device pointer, x)] -> values; one synchronisation for all of them"""
        k = len(jobs)
        counts = np.array([j[0] for j in jobs], dtype=np.uint64)
        ptrs = (N._vp * k)(*[j[1] for j in jobs])
        xs = np.concatenate([self.one(j[2]) for j in jobs]).reshape(k, 4)
        outs = np.zeros((k, 4), dtype=np.uint64)
        N.check(N.load().zk_poly_eval_many_dev(self.cid, k, N.u64p(counts), ptrs, N.u64p(xs), N.u64p(outs), None))
        return self.ints(outs)

    def d_add_at(self, vec, i, value):
        """vec[i] += value (mod r) on the device: a one-element launch on the stream, no round trip through the host"""
        p = vec.ptr(i)
        N.check(N.load().zk_vec_axpby_dev(self.cid, 1, N.u64p(self.one(1)), p, None, None, N.u64p(self.one(value % self.r)), p, None))

    def d_gather(self, n, src_ptr, stride, offset, dst_ptr):
        """dst[i] = src[offset + i * stride]"""
        N.check(N.load().zk_vec_gather_dev(self.cid, n, src_ptr, stride, offset, dst_ptr, None))

    def d_copy(self, n, src_ptr, dst_ptr):
        N.check(N.load().zk_vec_axpby_dev(self.cid, n, N.u64p(self.one(1)), src_ptr, None, None, None, dst_ptr, None))

    def d_grand_product(self, n, num_ptr, den_ptr, out_ptr):
        """out (n + 1): out[0] = 1, out[i+1] = out[i] * num[i] / den[i]"""
        N.check(N.load().zk_plonk_grand_product_dev(self.cid, n, num_ptr, den_ptr, out_ptr, None))

    def d_div_linear(self, n, coeffs_ptr, root, q_ptr):
        """q (n - 1) = coeffs (n) / (X - root); returns the remainder"""
        rem = np.zeros(4, dtype=np.uint64)
        N.check(N.load().zk_poly_div_linear_dev(self.cid, n, coeffs_ptr, N.u64p(self.one(root)), q_ptr, N.u64p(rem), None))
        return int.from_bytes(rem.tobytes(), "little")

    def d_perm_terms(self, n, wires, labels, beta, gamma, out_ptr):
        arr = ctypes.c_void_p * 3
        N.check(N.load().zk_plonk_perm_terms_dev(self.cid, n, arr(*wires), arr(*labels), N.u64p(self.one(beta)), N.u64p(self.one(gamma)),
                                                  out_ptr, None))

    def d_quotient(self, m, n, cols, zh_inv, beta, gamma, alpha, out_ptr):
        arr = ctypes.c_void_p * 15
        N.check(N.load().zk_plonk_quotient_dev(self.cid, m, n, arr(*cols), N.u64p(np.ascontiguousarray(zh_inv)), N.u64p(self.one(beta)),
                                                N.u64p(self.one(gamma)), N.u64p(self.one(alpha)), out_ptr, None))
