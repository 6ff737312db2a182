"""
Byte layouts of Proof / ProvingKey / VerifyingKey -- identical to the reference's
python/zksnake/groth16/serialization.py:5-220:

  Proof          A (G1) | B (G2) | C (G1), compressed points              (4 * n bytes, n = 32 / 48)
  ProvingKey     alpha_1 | beta_2 | delta_2 | beta_1 | delta_1, then four blocks
                 u64-LE count + points: tau_1 (G1), tau_2 (G2), target_1 (G1), kdelta_1 (G1)
  VerifyingKey   alpha_1 | beta_2 | gamma_2 | delta_2 | u64-LE count | ic (G1)

Key vectors may be Python lists of points or device-backed `PointArray`s.
"""

import numpy as np

from .. import _native as N
from .._algebra import PointArray, _point_class
from ..ecc import CurvePointSize, EllipticCurve


def _compress_many(points):
    if isinstance(points, PointArray):
        return points.to_bytes()
    return b"".join(bytes(p.to_bytes()) for p in points)


def _block(points):
    return len(points).to_bytes(8, "little") + _compress_many(points)


def _read_points(E, data, count, width):
    """`count` points of `width` bytes (G1: the curve's point size, G2: twice that) as a PointArray"""
    n = CurvePointSize[E.name].value
    assert len(data) >= count * width, "Invalid key length"
    return PointArray.from_compressed(E.curve_id, 1 if width == n else 2, data, count)


class Proof:
    def __init__(self, A, B, C):
        self.A = A
        self.B = B
        self.C = C

    def __str__(self):
        return f"A = {self.A}\nB = {self.B}\nC = {self.C}"

    __repr__ = __str__

    @classmethod
    def from_bytes(cls, s: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        assert len(s) == 4 * n, f"Length of the Proof must equal {4 * n} bytes"
        return cls(E.from_hex(s[:n].hex()), E.from_hex(s[n:3 * n].hex()), E.from_hex(s[3 * n:].hex()))

    def to_bytes(self) -> bytes:
        return bytes(self.A.to_bytes() + self.B.to_bytes() + self.C.to_bytes())


class ProvingKey:
    def __init__(self, alpha_G1, beta_G1, beta_G2, delta_G1, delta_G2, tau_G1, tau_G2, target_G1, k_delta_G1):
        self.alpha_1 = alpha_G1
        self.beta_1 = beta_G1
        self.beta_2 = beta_G2
        self.delta_1 = delta_G1
        self.delta_2 = delta_G2
        self.tau_1 = tau_G1
        self.tau_2 = tau_G2
        self.target_1 = target_G1
        self.kdelta_1 = k_delta_G1

    def to_bytes(self) -> bytes:
        head = b"".join(bytes(p.to_bytes()) for p in (self.alpha_1, self.beta_2, self.delta_2, self.beta_1, self.delta_1))
        return head + _block(self.tau_1) + _block(self.tau_2) + _block(self.target_1) + _block(self.kdelta_1)

    @classmethod
    def from_bytes(cls, b: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        assert len(b) >= 7 * n, "Invalid proving key length"
        alpha_1 = E.from_hex(b[0:n].hex())
        beta_2 = E.from_hex(b[n:3 * n].hex())
        delta_2 = E.from_hex(b[3 * n:5 * n].hex())
        beta_1 = E.from_hex(b[5 * n:6 * n].hex())
        delta_1 = E.from_hex(b[6 * n:7 * n].hex())
        pos = 7 * n
        vectors = []
        for width in (n, 2 * n, n, n):
            count = int.from_bytes(b[pos:pos + 8], "little")
            pos += 8
            vectors.append(_read_points(E, b[pos:pos + count * width], count, width))
            pos += count * width
        tau_1, tau_2, target_1, kdelta_1 = vectors
        return cls(alpha_1, beta_1, beta_2, delta_1, delta_2, tau_1, tau_2, target_1, kdelta_1)


class VerifyingKey:
    def __init__(self, alpha_G1, beta_G2, gamma_G2, delta_G2, IC):
        self.alpha_1 = alpha_G1
        self.beta_2 = beta_G2
        self.gamma_2 = gamma_G2
        self.delta_2 = delta_G2
        self.ic = IC

    def to_bytes(self) -> bytes:
        head = b"".join(bytes(p.to_bytes()) for p in (self.alpha_1, self.beta_2, self.gamma_2, self.delta_2))
        return head + _block(self.ic)

    @classmethod
    def from_bytes(cls, s: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        assert len(s) >= 7 * n, "Invalid verifying key length"
        alpha_1 = E.from_hex(s[0:n].hex())
        beta_2 = E.from_hex(s[n:3 * n].hex())
        gamma_2 = E.from_hex(s[3 * n:5 * n].hex())
        delta_2 = E.from_hex(s[5 * n:7 * n].hex())
        rest = s[7 * n + 8:]  # the count header is skipped, every remaining block is one ic point
        ic = _read_points(E, rest, len(rest) // n, n)
        return cls(alpha_1, beta_2, gamma_2, delta_2, ic)
