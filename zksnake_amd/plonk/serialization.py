"""
Proof / ProvingKey / VerifyingKey of PlonK with the byte layouts of the reference
(python/zksnake/plonk/serialization.py): Proof = 9 compressed G1 points + 6 scalars (32-byte LE) `:101-126`;
ProvingKey = u64 count + tau_g1 points, 5 selector + 3 permutation commitments, then 17 length-prefixed
scalar vectors (5 selector, 3 permutation, 3 identity polynomials; 5 selector evaluations over 4n; L1 over 4n)
`:245-285`; VerifyingKey = u64 n, tau_g2, the 8 commitments `:340-353`.

Scalar vectors live as (k, 4) uint64 limb arrays -- their bytes ARE the serialized form -- and the
reference's `Polynomial` / list views are built on demand.
"""

import numpy as np

from .. import _native as N
from ..ecc import CurvePointSize, EllipticCurve, PointArray
from ..frvec import FrOps
from ..polynomial import Polynomial

SELECTORS = ("L", "R", "O", "M", "C")
PROOF_POINTS = ("tau_a", "tau_b", "tau_c", "tau_z", "tau_t_lo", "tau_t_mid", "tau_t_hi", "tau_W_zeta", "tau_W_zeta_omega")
PROOF_SCALARS = ("zeta_a", "zeta_b", "zeta_c", "zeta_sigma1", "zeta_sigma2", "zeta_omega")


class Proof:
    def __init__(self, tau_a, tau_b, tau_c, tau_z, tau_t_lo, tau_t_mid, tau_t_hi, tau_W_zeta, tau_W_zeta_omega,
                 zeta_a, zeta_b, zeta_c, zeta_sigma1, zeta_sigma2, zeta_omega):
        args = (tau_a, tau_b, tau_c, tau_z, tau_t_lo, tau_t_mid, tau_t_hi, tau_W_zeta, tau_W_zeta_omega,
                zeta_a, zeta_b, zeta_c, zeta_sigma1, zeta_sigma2, zeta_omega)
        for name, value in zip(PROOF_POINTS + PROOF_SCALARS, args):
            setattr(self, name, value)

    @classmethod
    def from_bytes(cls, s: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        expected = n * len(PROOF_POINTS) + 32 * len(PROOF_SCALARS)
        assert len(s) == expected, f"Length of the Proof must equal {expected} bytes"
        points = [E.from_hex(s[i * n:(i + 1) * n].hex()) for i in range(len(PROOF_POINTS))]
        tail = s[n * len(PROOF_POINTS):]
        scalars = [int.from_bytes(tail[32 * i:32 * (i + 1)], "little") for i in range(len(PROOF_SCALARS))]
        return cls(*points, *scalars)

    def to_bytes(self) -> bytes:
        points = b"".join(bytes(getattr(self, name).to_bytes()) for name in PROOF_POINTS)
        return points + b"".join(getattr(self, name).to_bytes(32, "little") for name in PROOF_SCALARS)


def _limbs_of(x, ops):
    """Polynomial | list[int] | limb array -> limb array"""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    if hasattr(x, "coeffs"):
        x = x.coeffs()
    return ops.limbs(list(x))


def _vector_bytes(arr):
    return len(arr).to_bytes(8, "little") + arr.tobytes()


class ProvingKey:
    def __init__(self, n, tau_G1, selector_poly, selector_eval, permutation_poly, identity_poly, tau_selector,
                 tau_permutation, lagrange_evals, curve: str = "BN254"):
        self.E = EllipticCurve(curve)
        self.order = self.E.order
        self.n = n
        self.tau_g1 = tau_G1
        ops = self._ops = FrOps(self.order)
        self._selector = {k: _limbs_of(selector_poly[k], ops) for k in SELECTORS}
        self._selector_eval = {k: _limbs_of(selector_eval[k], ops) for k in SELECTORS}
        self._permutation = [_limbs_of(p, ops) for p in permutation_poly]
        self._identity = [_limbs_of(p, ops) for p in identity_poly]
        self._lagrange = _limbs_of(lagrange_evals, ops)
        self.tau_selector_poly = tau_selector
        self.tau_permutation_poly = tau_permutation
        self._cache = {}  # prover-side derived vectors (coset evaluations), filled by Plonk.prove

    # the reference's views
    def _poly(self, arr):
        return Polynomial(self._ops.ints(FrOps.strip(arr)) or [0], self.order)

    @property
    def selector_poly(self):
        return {k: self._poly(v) for k, v in self._selector.items()}

    @property
    def permutation_poly(self):
        return [self._poly(v) for v in self._permutation]

    @property
    def identity_poly(self):
        return [self._poly(v) for v in self._identity]

    @property
    def selector_eval(self):
        return {k: self._ops.ints(v) for k, v in self._selector_eval.items()}

    @property
    def lagrange_evals(self):
        return self._ops.ints(self._lagrange)

    @classmethod
    def from_bytes(cls, s: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        view = memoryview(s)
        count = int.from_bytes(view[:8], "little")
        off = 8
        assert len(s) >= off + (count + 8) * n, "Invalid proving key length"
        tau_g1 = PointArray.from_compressed(E.curve_id, 1, view[off:off + count * n], count)
        off += count * n
        commits = [E.from_hex(bytes(view[off + i * n: off + (i + 1) * n]).hex()) for i in range(8)]
        off += 8 * n
        vectors = []
        while off < len(s):
            length = int.from_bytes(view[off:off + 8], "little")
            vectors.append(np.frombuffer(view[off + 8: off + 8 + 32 * length], dtype=np.uint64).reshape(length, 4).copy())
            off += 8 + 32 * length
        assert len(vectors) == 17, "Malformed ProvingKey structure"
        domain = len(vectors[11]) // 4  # selector evaluations always hold 4n entries
        return cls(domain, tau_g1, dict(zip(SELECTORS, vectors[0:5])), dict(zip(SELECTORS, vectors[11:16])),
                   vectors[5:8], vectors[8:11], dict(zip(SELECTORS, commits[:5])), commits[5:8], vectors[16], crv)

    def to_bytes(self) -> bytes:
        out = [len(self.tau_g1).to_bytes(8, "little")]
        out.append(self.tau_g1.to_bytes() if isinstance(self.tau_g1, PointArray) else b"".join(bytes(t.to_bytes()) for t in self.tau_g1))
        out += [bytes(self.tau_selector_poly[k].to_bytes()) for k in SELECTORS]
        out += [bytes(p.to_bytes()) for p in self.tau_permutation_poly]
        for arr in [self._selector[k] for k in SELECTORS] + self._permutation + self._identity:
            out.append(_vector_bytes(FrOps.strip(arr)))  # Polynomial.coeffs() drops trailing zeros
        out += [_vector_bytes(self._selector_eval[k]) for k in SELECTORS]
        out.append(_vector_bytes(self._lagrange))
        return b"".join(out)


class VerifyingKey:
    def __init__(self, n, tau_G2, tau_selector_poly, tau_permutation_poly, curve: str = "BN254"):
        self.E = EllipticCurve(curve)
        self.order = self.E.order
        self.n = n
        self.tau_g2 = tau_G2
        self.tau_selector_poly = tau_selector_poly
        self.tau_permutation_poly = tau_permutation_poly

    @classmethod
    def from_bytes(cls, s: bytes, crv="BN254"):
        E = EllipticCurve(crv)
        n = CurvePointSize[crv].value
        domain = int.from_bytes(s[:8], "little")
        tau_g2 = E.from_hex(s[8:8 + 2 * n].hex())
        off = 8 + 2 * n
        commits = [E.from_hex(s[off + i * n: off + (i + 1) * n].hex()) for i in range(8)]
        return cls(domain, tau_g2, dict(zip(SELECTORS, commits[:5])), commits[5:], crv)

    def to_bytes(self) -> bytes:
        out = [self.n.to_bytes(8, "little"), bytes(self.tau_g2.to_bytes())]
        out += [bytes(self.tau_selector_poly[k].to_bytes()) for k in SELECTORS]
        out += [bytes(p.to_bytes()) for p in self.tau_permutation_poly]
        return b"".join(out)
