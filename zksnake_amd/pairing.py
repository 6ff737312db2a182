"""
Host-side optimal-ate pairing for BN254 and BLS12-381 (plain Python integers).

Only `Groth16.verify` needs it (reference: pairing/multi_pairing, src/bn254/curve.rs:417-437 ->
ark-ec `Bn254::multi_pairing`); it is off the proving hot path, runs on the CPU in the reference too,
and is listed as a "next" row for a native C++ version.

Tower used here: Fp2 = Fp[u]/(u^2+1), Fp12 = Fp2[w]/(w^6 - xi), xi = 9+u (BN254) / 1+u (BLS12-381).
An Fp12 element is a list of six Fp2 coefficients (pairs of ints) in the basis 1, w, ..., w^5.
"""

from . import constant


class _Tower:
    def __init__(self, p, r, xi, loop, loop_negative, is_bn, twist_m):
        self.p = p
        self.r = r
        self.xi = xi
        self.loop = loop
        self.loop_negative = loop_negative
        self.is_bn = is_bn
        self.twist_m = twist_m
        self.final_exponent = (p ** 12 - 1) // r
        # Frobenius constants for the twist (only the BN254 loop tail needs them)
        self.g_x1 = self.f2_pow(xi, (p - 1) // 3)
        self.g_y1 = self.f2_pow(xi, (p - 1) // 2)
        self.g_x2 = self.f2_pow(xi, (p * p - 1) // 3)
        self.g_y2 = self.f2_pow(xi, (p * p - 1) // 2)

    # ---- Fp2 ----
    def f2_add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def f2_sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def f2_mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def f2_scale(self, a, k):
        return (a[0] * k % self.p, a[1] * k % self.p)

    def f2_neg(self, a):
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def f2_conj(self, a):
        return (a[0], (-a[1]) % self.p)

    def f2_inv(self, a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, self.p)
        return (a[0] * d % self.p, (-a[1]) * d % self.p)

    def f2_pow(self, a, e):
        out = (1, 0)
        while e:
            if e & 1:
                out = self.f2_mul(out, a)
            a = self.f2_mul(a, a)
            e >>= 1
        return out

    # ---- Fp12 ----
    def one(self):
        return [(1, 0)] + [(0, 0)] * 5

    def mul(self, a, b):
        p = self.p
        t0 = [0] * 11
        t1 = [0] * 11
        for i in range(6):
            a0, a1 = a[i]
            if a0 == 0 and a1 == 0:
                continue
            for j in range(6):
                b0, b1 = b[j]
                t0[i + j] += a0 * b0 - a1 * b1
                t1[i + j] += a0 * b1 + a1 * b0
        x0, x1 = self.xi
        out = []
        for k in range(6):
            c0, c1 = t0[k], t1[k]
            if k + 6 < 11:
                h0, h1 = t0[k + 6], t1[k + 6]
                c0 += h0 * x0 - h1 * x1
                c1 += h0 * x1 + h1 * x0
            out.append((c0 % p, c1 % p))
        return out

    def mul_sparse(self, a, terms):
        """a * sum(c * w^k for k, c in terms) with few non-zero Fp2 coefficients (a line function)."""
        b = [(0, 0)] * 6
        for k, c in terms:
            b[k] = c
        return self.mul(a, b)

    def conj(self, a):
        """Frobenius p^6: w -> -w"""
        return [a[i] if i % 2 == 0 else self.f2_neg(a[i]) for i in range(6)]

    def pow(self, a, e):
        out = self.one()
        while e:
            if e & 1:
                out = self.mul(out, a)
            a = self.mul(a, a)
            e >>= 1
        return out

    # ---- curve arithmetic on the twist (affine, Fp2 coordinates) ----
    def _line_and_step(self, T, Q, P):
        """returns (line terms evaluated at P, T + Q) for points T, Q on the twist (T == Q doubles)."""
        xt, yt = T
        xq, yq = Q
        if T == Q:
            lam = self.f2_mul(self.f2_scale(self.f2_mul(xt, xt), 3), self.f2_inv(self.f2_scale(yt, 2)))
        else:
            lam = self.f2_mul(self.f2_sub(yq, yt), self.f2_inv(self.f2_sub(xq, xt)))
        x3 = self.f2_sub(self.f2_sub(self.f2_mul(lam, lam), xt), xq)
        y3 = self.f2_sub(self.f2_mul(lam, self.f2_sub(xt, x3)), yt)
        xp, yp = P
        c = self.f2_sub(self.f2_mul(lam, xt), yt)
        if self.twist_m:
            # (yp - lam xp w^-1 + c w^-3) * w^3 ; the factor w^3 lies in Fp4 and dies in the final exponentiation
            terms = [(0, c), (2, self.f2_neg(self.f2_scale(lam, xp))), (3, (yp % self.p, 0))]
        else:
            terms = [(0, (yp % self.p, 0)), (1, self.f2_neg(self.f2_scale(lam, xp))), (3, c)]
        return terms, (x3, y3)

    def miller(self, P, Q):
        """P = (x, y) ints in G1, Q = ((x0,x1),(y0,y1)) in G2; None = infinity."""
        f = self.one()
        if P is None or Q is None:
            return f
        T = Q
        for bit in bin(self.loop)[3:]:
            terms, T = self._line_and_step(T, T, P)
            f = self.mul_sparse(self.mul(f, f), terms)
            if bit == "1":
                terms, T = self._line_and_step(T, Q, P)
                f = self.mul_sparse(f, terms)
        if self.is_bn:
            q1 = (self.f2_mul(self.f2_conj(Q[0]), self.g_x1), self.f2_mul(self.f2_conj(Q[1]), self.g_y1))
            q2 = (self.f2_mul(Q[0], self.g_x2), self.f2_neg(self.f2_mul(Q[1], self.g_y2)))
            terms, T = self._line_and_step(T, q1, P)
            f = self.mul_sparse(f, terms)
            terms, T = self._line_and_step(T, q2, P)
            f = self.mul_sparse(f, terms)
        if self.loop_negative:
            f = self.conj(f)
        return f

    def final_exp(self, f):
        return self.pow(f, self.final_exponent)


_TOWERS = {}


def _tower(curve_id):
    if curve_id not in _TOWERS:
        if curve_id == 0:
            _TOWERS[0] = _Tower(constant.BN254_MODULUS, constant.BN254_SCALAR_FIELD, (9, 1),
                                29793968203157093288, False, True, False)
        else:
            _TOWERS[1] = _Tower(constant.BLS12_381_MODULUS, constant.BLS12_381_SCALAR_FIELD, (1, 1),
                                0xD201000000010000, True, False, True)
    return _TOWERS[curve_id]


class GT:
    """target-group element (the reference's PointG12: only == and str are exposed)."""

    def __init__(self, curve_id, coeffs):
        self.curve_id = curve_id
        self.coeffs = [tuple(c) for c in coeffs]

    def __eq__(self, other):
        return isinstance(other, GT) and self.curve_id == other.curve_id and self.coeffs == other.coeffs

    def __hash__(self):
        return hash((self.curve_id, tuple(self.coeffs)))

    def __repr__(self):
        return "GT(" + ", ".join(f"[{c[0]:#x}, {c[1]:#x}]" for c in self.coeffs) + ")"


def pairing(curve_id, P, Q):
    """e(P, Q); P, Q given as affine integer tuples (None = infinity)."""
    t = _tower(curve_id)
    return GT(curve_id, t.final_exp(t.miller(P, Q)))


def multi_pairing(curve_id, Ps, Qs):
    t = _tower(curve_id)
    f = t.one()
    for P, Q in zip(Ps, Qs):
        f = t.mul(f, t.miller(P, Q))
    return GT(curve_id, t.final_exp(f))
