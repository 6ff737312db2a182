"""
Pairing front end: `pairing` / `multi_pairing` of the reference (src/bn254/curve.rs:417-437) through the
library's host-side optimal-ate implementation (zk_multi_pairing in csrc/pairing.hip).  Only
`Groth16.verify` needs it; like in the reference it runs on the CPU.
"""

import numpy as np

from . import _native as N


class GT:
    """target-group element (the reference's PointG12: only == and str are exposed).
    `coeffs` are the 12 base-field coefficients (c0, c1 of the Fp2 coefficient of 1, w, .., w^5)."""

    def __init__(self, curve_id, coeffs):
        self.curve_id = curve_id
        self.coeffs = tuple(int(c) for c in coeffs)

    def __eq__(self, other):
        return isinstance(other, GT) and self.curve_id == other.curve_id and self.coeffs == other.coeffs

    def __hash__(self):
        return hash((self.curve_id, self.coeffs))

    def is_one(self):
        return self.coeffs == (1,) + (0,) * 11

    def __repr__(self):
        return "GT(" + ", ".join(f"{c:#x}" for c in self.coeffs) + ")"


def _limbs(points, cid, group):
    """affine integer tuples (None = infinity) -> (n, limbs) uint64"""
    fq = N.fq_limbs(cid)
    rows = []
    for pt in points:
        if pt is None:
            rows.append(b"\0" * (8 * fq * 2 * group))
        elif group == 1:
            rows.append(pt[0].to_bytes(8 * fq, "little") + pt[1].to_bytes(8 * fq, "little"))
        else:
            rows.append(b"".join(c.to_bytes(8 * fq, "little") for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1])))
    return np.frombuffer(b"".join(rows), dtype=np.uint64).reshape(len(points), 2 * fq * group).copy()


def multi_pairing(curve_id, Ps, Qs):
    """prod e(P_i, Q_i); P_i = (x, y) ints in G1, Q_i = ((x0, x1), (y0, y1)) in G2, None = infinity"""
    lib = N.load()
    if len(Ps) != len(Qs):
        raise ValueError("Length of a and b must be equal")
    n = len(Ps)
    fq = N.fq_limbs(curve_id)
    out = np.zeros(12 * fq, dtype=np.uint64)
    g1 = _limbs(Ps, curve_id, 1) if n else np.zeros((1, 2 * fq), dtype=np.uint64)
    g2 = _limbs(Qs, curve_id, 2) if n else np.zeros((1, 4 * fq), dtype=np.uint64)
    N.check(lib.zk_multi_pairing(curve_id, n, N.u64p(g1), N.u64p(g2), N.u64p(out)))
    return GT(curve_id, N.limbs_to_ints(out.reshape(12, fq)))


def pairing(curve_id, P, Q):
    return multi_pairing(curve_id, [P], [Q])
