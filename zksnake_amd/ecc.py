"""
Curve front end with the reference's surface (python/zksnake/ecc.py:55-149): curve-name dispatch,
generators, pairing, batch_mul and multiexp.  Points and MSMs come from `zksnake_amd._algebra`,
i.e. from libzkmi.so.
"""

from enum import Enum

from . import _algebra
from ._algebra import PointArray
from .constant import BLS12_381_MODULUS, BLS12_381_SCALAR_FIELD, BN254_MODULUS, BN254_SCALAR_FIELD

_BN = ("BN128", "BN254", "ALT_BN128")


class CurveType(Enum):
    BN128 = "ec_bn254"
    BN254 = "ec_bn254"
    ALT_BN128 = "ec_bn254"
    BLS12_381 = "ec_bls12_381"


class CurvePointSize(Enum):
    BN128 = 32
    BN254 = 32
    ALT_BN128 = 32
    BLS12_381 = 48


class CurveScalarSize(Enum):
    BN128 = 32
    BN254 = 32
    ALT_BN128 = 32
    BLS12_381 = 32


def _module(name):
    return getattr(_algebra, CurveType[name].value)


def ispointG1(x):
    return isinstance(x, (_algebra.ec_bn254.PointG1, _algebra.ec_bls12_381.PointG1))


def ispointG2(x):
    return isinstance(x, (_algebra.ec_bn254.PointG2, _algebra.ec_bls12_381.PointG2))


class EllipticCurve:
    def __init__(self, curve: str):
        self.name = curve
        self.curve = _module(curve)  # KeyError for an unknown name, like the reference's Enum lookup
        bn = curve in _BN
        self.curve_id = 0 if bn else 1  # ZK_CURVE_BN254 / ZK_CURVE_BLS12_381
        self.order = BN254_SCALAR_FIELD if bn else BLS12_381_SCALAR_FIELD
        self.field_modulus = BN254_MODULUS if bn else BLS12_381_MODULUS

    def G1(self):
        return self.curve.g1()

    def G2(self):
        return self.curve.g2()

    def pairing(self, a, b):
        return self.curve.pairing(a, b)

    def multi_pairing(self, a, b):
        assert len(a) == len(b), "Length of a and b must be equal"
        return self.curve.multi_pairing(a, b)

    def _group_of(self, g):
        first = g[0] if isinstance(g, (list, tuple)) else g
        if isinstance(g, PointArray):
            return g.group
        if isinstance(first, self.curve.PointG1):
            return 1
        if isinstance(first, self.curve.PointG2):
            return 2
        raise TypeError(f"Invalid curve type: {type(first)}")

    def batch_mul(self, g, s, as_array=False):
        """[s_i * g_i]; a single point g is broadcast over all scalars (ecc.py:88-105)."""
        if isinstance(g, (list, tuple, PointArray)) and len(g) == 0:
            return []
        fn = self.curve.batch_multi_scalar_g1 if self._group_of(g) == 1 else self.curve.batch_multi_scalar_g2
        return fn(g, s, as_array=as_array)

    def multiexp(self, g, s):
        """sum_i s_i * g_i with the reference's length rules (ecc.py:107-126):
        no scalars -> identity; fewer scalars than points -> points truncated; more -> ValueError."""
        assert len(g) > 0
        group = self._group_of(g)
        if len(s) == 0:
            return g[0] * 0
        if len(s) < len(g) and not isinstance(g, PointArray):
            g = g[: len(s)]
        if isinstance(g, PointArray) and len(s) < len(g):
            return self._msm_prefix(g, s, group)
        fn = self.curve.multiscalar_mul_g1 if group == 1 else self.curve.multiscalar_mul_g2
        return fn(g, s)

    def _msm_prefix(self, g, s, group):
        """first len(s) bases of a device-resident array (the plan ignores the tail)"""
        import numpy as np
        from . import _native as N
        lib = N.ensure_gpu()
        sc = _algebra._scalar_limbs(s, g.curve_id)
        out = np.zeros(N.point_limbs(g.curve_id, group), dtype=np.uint64)
        N.check(lib.zk_msm_plan_run(g.plan(), sc.shape[0], sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        return _algebra._point_class(g.curve_id, group)._from_limbs(out)

    def from_hex(self, hexstring: str):
        data = bytes.fromhex(hexstring)
        n = CurvePointSize[self.name].value * 2
        if len(hexstring) == n:
            return self.curve.PointG1.from_bytes(data)
        if len(hexstring) == 2 * n:
            return self.curve.PointG2.from_bytes(data)
        raise ValueError(f"Hexstring size of {n} or {n*2} expected, got {len(hexstring)}")

    def __call__(self, x, y):
        if isinstance(x, (tuple, list)) and isinstance(y, (tuple, list)):
            return self.curve.PointG2(x[0], x[1], y[0], y[1])
        return self.curve.PointG1(x, y)
