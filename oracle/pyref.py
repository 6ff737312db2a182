"""
oracle/pyref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Pure-Python big-integer restatement of the *definitions* behind the zksnake
hot path (Groth16.prove -> QAP.evaluate_witness -> ifft/fft/multiexp).  The
arithmetic of the reference lives in arkworks crates that are not vendored in
/root/reference and cannot be built here (no Rust toolchain), so this file
restates the published algorithms and anchors them on

  * the mathematical definitions (DFT at powers of omega; sum s_i*P_i),
  * the public known-answer vectors listed in SURVEY.md Appendix A / 8c(4)
    (NTT_4([1,2,3,4]) for both scalar fields, the BLS12-381 compressed G1
    generator, the BN254 G1 generator encoding), and
  * the reference's own algebraic tests (prove -> verify round trip,
    /root/reference/tests/test_groth16.py:68-144; zero remainder,
    tests/test_r1cs_qap.py:9-109; polynomial identities tests/test_algebra.py:6-46).

PARITY UNPINNED: the reference ships no golden MSM/NTT/proof/byte vectors and
its Rust extension cannot run in this container, so bit-parity with the
reference binary itself is not provable here; see DESIGN.md "Oracle".

Reference call sites restated (file:line under /root/reference):
  fft/ifft ............. src/bn254/polynomial.rs:535-571 (ark-poly 0.4.2 Radix2EvaluationDomain)
  mul_over_evaluation .. src/bn254/polynomial.rs:609-634
  divide_by_vanishing .. src/bn254/polynomial.rs:466-489 (ark-poly DensePolynomial)
  multiscalar_mul_g1/g2  src/bn254/curve.rs:356-392     (ark-ec 0.4.2 VariableBaseMSM)
  PointG1/G2 ops ....... src/bn254/curve.rs:25-324
  to_bytes/from_bytes .. src/bn254/curve.rs:127-146 (ark-serialize 0.4.2 compressed)
  pairing .............. src/bn254/curve.rs:417-437
  Groth16.setup/prove .. python/zksnake/groth16/protocol.py:32-165
  QAP.evaluate_witness . python/zksnake/groth16/qap.py:42-71
  SparseArray.dot ...... python/zksnake/array.py:36-44
"""

from __future__ import annotations

# ----------------------------------------------------------------------------
# constants (python/zksnake/constant.py:5-15)
# ----------------------------------------------------------------------------

BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS_P = 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787
BLS_R = 52435875175126190479447740508185965837690552500527637822603658699938581184513


class CurveParams:
    """Static description of one pairing-friendly curve (G1 over Fp, G2 over Fp2 = Fp[u]/(u^2+1))."""

    def __init__(self, name, p, r, b1, b2, g1, g2, fr_gen, two_adicity, fp_bytes,
                 ate_loop, ate_neg, twist_type, fp12_mod, is_bn):
        self.name = name
        self.p = p
        self.r = r
        self.b1 = b1              # G1: y^2 = x^3 + b1
        self.b2 = b2              # G2: y^2 = x^3 + b2, b2 in Fp2 as (c0, c1)
        self.g1 = g1              # affine generator (x, y)
        self.g2 = g2              # affine generator ((x0,x1),(y0,y1))
        self.fr_gen = fr_gen      # multiplicative generator of Fr
        self.two_adicity = two_adicity
        self.fp_bytes = fp_bytes
        self.ate_loop = ate_loop
        self.ate_neg = ate_neg
        self.twist_type = twist_type  # 'D' or 'M'
        self.fp12_mod = fp12_mod  # (c6, c0): w^12 = c6*w^6 + c0  (Fp12 = Fp[w]/(w^12 - c6 w^6 - c0))
        self.is_bn = is_bn

    def root_of_unity(self, n):
        """generator of the size-n subgroup used by ark-poly's Radix2EvaluationDomain
        (SURVEY Appendix A 'Domain rule'): (g^((r-1)/2^s))^(2^(s-log n))."""
        assert n & (n - 1) == 0 and n >= 1
        log_n = n.bit_length() - 1
        if log_n > self.two_adicity:
            raise ValueError("Domain size is too large")
        root = pow(self.fr_gen, (self.r - 1) >> self.two_adicity, self.r)
        return pow(root, 1 << (self.two_adicity - log_n), self.r)


BN254 = CurveParams(
    "BN254", BN254_P, BN254_R, 3,
    (19485874751759354771024239261021720505790618469301721065564631296452457478373,
     266929791119991161246907387137283842545076965332900288569378510910307636690),
    (1, 2),
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    5, 28, 32,
    29793968203157093288,  # 6x+2, x = 4965661367192848881
    False, 'D', (18, -82), True,
)

BLS12_381 = CurveParams(
    "BLS12_381", BLS_P, BLS_R, 4, (4, 4),
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569),
    ((352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
      3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
     (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
      927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582)),
    7, 32, 48,
    0xd201000000010000,  # |x|, x negative
    True, 'M', (2, -2), False,
)

CURVES = {"BN254": BN254, "BN128": BN254, "ALT_BN128": BN254, "BLS12_381": BLS12_381}


def curve_by_name(name) -> CurveParams:
    return CURVES[name]


# ----------------------------------------------------------------------------
# Fp2 helpers (elements are (c0, c1) with u^2 = -1)
# ----------------------------------------------------------------------------

def f2_add(a, b, p):
    return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)


def f2_sub(a, b, p):
    return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)


def f2_neg(a, p):
    return ((-a[0]) % p, (-a[1]) % p)


def f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(a, p):
    d = pow((a[0] * a[0] + a[1] * a[1]) % p, -1, p)
    return (a[0] * d % p, (-a[1]) * d % p)


def f2_pow(a, e, p):
    out = (1, 0)
    base = a
    while e:
        if e & 1:
            out = f2_mul(out, base, p)
        base = f2_mul(base, base, p)
        e >>= 1
    return out


def f2_sqrt(a, p):
    """square root in Fp2 for p = 3 mod 4 (complex method); returns None if a is a non-residue."""
    if a == (0, 0):
        return (0, 0)
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0, p)
        if s is not None:
            return (s, 0)
        s = fp_sqrt((-a0) % p, p)
        return (0, s)
    norm = (a0 * a0 + a1 * a1) % p
    alpha = fp_sqrt(norm, p)
    if alpha is None:
        return None
    inv2 = pow(2, -1, p)
    delta = (a0 + alpha) * inv2 % p
    x0 = fp_sqrt(delta, p)
    if x0 is None:
        delta = (a0 - alpha) * inv2 % p
        x0 = fp_sqrt(delta, p)
        if x0 is None:
            return None
    x1 = a1 * pow(2 * x0, -1, p) % p
    cand = (x0, x1)
    if f2_mul(cand, cand, p) != (a0 % p, a1 % p):
        return None
    return cand


def fp_sqrt(a, p):
    a %= p
    if a == 0:
        return 0
    s = pow(a, (p + 1) // 4, p)  # p = 3 mod 4 for both curves
    return s if s * s % p == a else None


# ----------------------------------------------------------------------------
# Generic short-Weierstrass affine arithmetic; None = point at infinity.
# `F` bundles the field operations so the same code serves G1 (Fp) and G2 (Fp2).
# ----------------------------------------------------------------------------

class _Fp:
    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def small(self, k, a): return k * a % self.p


class _Fp2:
    def __init__(self, p):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): return f2_add(a, b, self.p)
    def sub(self, a, b): return f2_sub(a, b, self.p)
    def mul(self, a, b): return f2_mul(a, b, self.p)
    def neg(self, a): return f2_neg(a, self.p)
    def inv(self, a): return f2_inv(a, self.p)
    def small(self, k, a): return (k * a[0] % self.p, k * a[1] % self.p)


def ec_add(F, P, Q):
    """affine chord-and-tangent addition (a = 0 curves)."""
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if y1 == y2 and y1 != F.zero:
            lam = F.mul(F.small(3, F.mul(x1, x1)), F.inv(F.small(2, y1)))
        else:
            return None
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, P):
    if P is None:
        return None
    return (P[0], F.neg(P[1]))


def ec_mul(F, P, k):
    """double-and-add scalar multiplication with Jacobian intermediates (fast enough for tests)."""
    if P is None or k == 0:
        return None
    if k < 0:
        return ec_mul(F, ec_neg(F, P), -k)
    # Jacobian (X, Y, Z), a = 0
    def dbl(T):
        X, Y, Z = T
        if Y == F.zero:
            return None
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.small(2, F.sub(F.sub(F.mul(t, t), A), C))
        E = F.small(3, A)
        Fq = F.mul(E, E)
        X3 = F.sub(Fq, F.small(2, D))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.small(8, C))
        Z3 = F.small(2, F.mul(Y, Z))
        return (X3, Y3, Z3)

    def madd(T, Qa):
        # T Jacobian + Qa affine
        X1, Y1, Z1 = T
        x2, y2 = Qa
        Z1Z1 = F.mul(Z1, Z1)
        U2 = F.mul(x2, Z1Z1)
        S2 = F.mul(F.mul(y2, Z1), Z1Z1)
        if U2 == X1:
            if S2 == Y1:
                return dbl(T)
            return None
        H = F.sub(U2, X1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        rr = F.sub(S2, Y1)
        V = F.mul(X1, HH)
        X3 = F.sub(F.sub(F.mul(rr, rr), HHH), F.small(2, V))
        Y3 = F.sub(F.mul(rr, F.sub(V, X3)), F.mul(Y1, HHH))
        Z3 = F.mul(Z1, H)
        return (X3, Y3, Z3)

    acc = None
    for bit in bin(k)[2:]:
        if acc is not None:
            acc = dbl(acc)
        if bit == "1":
            acc = (P[0], P[1], F.one) if acc is None else madd(acc, P)
    if acc is None:
        return None
    X, Y, Z = acc
    zi = F.inv(Z)
    zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def on_curve(F, P, b):
    if P is None:
        return True
    x, y = P
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), b)


class Group:
    """G1 or G2 of a curve, with the reference's scalar semantics (scalars reduced mod r,
    src/bn254/curve.rs:101-106 `Fr::from(BigUint)`)."""

    def __init__(self, curve: CurveParams, which: int):
        self.curve = curve
        self.which = which
        if which == 1:
            self.F = _Fp(curve.p)
            self.b = curve.b1
            self.gen = curve.g1
        else:
            self.F = _Fp2(curve.p)
            self.b = curve.b2
            self.gen = curve.g2

    def add(self, P, Q): return ec_add(self.F, P, Q)
    def neg(self, P): return ec_neg(self.F, P)
    def mul(self, P, k): return ec_mul(self.F, P, k % self.curve.r)
    def is_on_curve(self, P): return on_curve(self.F, P, self.b)

    def msm(self, points, scalars):
        """definition of multiscalar_mul_g1/g2: sum_i s_i * P_i (src/bn254/curve.rs:356-392)."""
        if len(points) != len(scalars):
            raise ValueError("Number of points and scalars mismatch")
        acc = None
        for P, s in zip(points, scalars):
            acc = self.add(acc, self.mul(P, s))
        return acc


def G1(curve): return Group(curve, 1)
def G2(curve): return Group(curve, 2)


# ----------------------------------------------------------------------------
# compressed point codec
# ----------------------------------------------------------------------------

def _f2_gt(a, b):
    """ark-ff QuadExt ordering: compare c1 first, then c0."""
    return (a[1], a[0]) > (b[1], b[0])


def compress(curve: CurveParams, which: int, P) -> bytes:
    p = curve.p
    n = curve.fp_bytes
    if curve.name == "BN254":
        # ark-serialize 0.4 SWFlags: little-endian x; bit7 of last byte = y > -y, bit6 = infinity
        if which == 1:
            if P is None:
                out = bytearray(n); out[-1] |= 0x40; return bytes(out)
            x, y = P
            out = bytearray(x.to_bytes(n, "little"))
            if y > (p - y) % p:
                out[-1] |= 0x80
            return bytes(out)
        if P is None:
            out = bytearray(2 * n); out[-1] |= 0x40; return bytes(out)
        x, y = P
        out = bytearray(x[0].to_bytes(n, "little") + x[1].to_bytes(n, "little"))
        if _f2_gt(y, f2_neg(y, p)):
            out[-1] |= 0x80
        return bytes(out)
    # BLS12-381: zcash format, big-endian, flags in the first byte
    if which == 1:
        if P is None:
            out = bytearray(n); out[0] = 0xC0; return bytes(out)
        x, y = P
        out = bytearray(x.to_bytes(n, "big"))
        out[0] |= 0x80
        if y > (p - y) % p:
            out[0] |= 0x20
        return bytes(out)
    if P is None:
        out = bytearray(2 * n); out[0] = 0xC0; return bytes(out)
    x, y = P
    out = bytearray(x[1].to_bytes(n, "big") + x[0].to_bytes(n, "big"))
    out[0] |= 0x80
    if _f2_gt(y, f2_neg(y, p)):
        out[0] |= 0x20
    return bytes(out)


def decompress(curve: CurveParams, which: int, data: bytes):
    p = curve.p
    n = curve.fp_bytes
    data = bytearray(data)
    if len(data) != n * which:
        raise ValueError("Cannot deserialize point: bad length")
    if curve.name == "BN254":
        flags = data[-1] & 0xC0
        data[-1] &= 0x3F
        if flags == 0xC0:
            raise ValueError("Cannot deserialize point: invalid flags")
        inf = bool(flags & 0x40)
        neg = bool(flags & 0x80)
        if which == 1:
            x = int.from_bytes(data, "little")
            if inf:
                return None
            if x >= p:
                raise ValueError("Cannot deserialize point: x not in field")
            y = fp_sqrt((x * x * x + curve.b1) % p, p)
            if y is None:
                raise ValueError("Cannot deserialize point: not on curve")
            if (y > (p - y) % p) != neg:
                y = (p - y) % p
            return (x, y)
        x = (int.from_bytes(data[:n], "little"), int.from_bytes(data[n:], "little"))
        if inf:
            return None
        if x[0] >= p or x[1] >= p:
            raise ValueError("Cannot deserialize point: x not in field")
        rhs = f2_add(f2_mul(f2_mul(x, x, p), x, p), curve.b2, p)
        y = f2_sqrt(rhs, p)
        if y is None:
            raise ValueError("Cannot deserialize point: not on curve")
        if _f2_gt(y, f2_neg(y, p)) != neg:
            y = f2_neg(y, p)
        return (x, y)
    flags = data[0] & 0xE0
    data[0] &= 0x1F
    if not flags & 0x80:
        raise ValueError("Cannot deserialize point: uncompressed encoding")
    inf = bool(flags & 0x40)
    big = bool(flags & 0x20)
    if which == 1:
        x = int.from_bytes(data, "big")
        if inf:
            return None
        if x >= p:
            raise ValueError("Cannot deserialize point: x not in field")
        y = fp_sqrt((x * x * x + curve.b1) % p, p)
        if y is None:
            raise ValueError("Cannot deserialize point: not on curve")
        if (y > (p - y) % p) != big:
            y = (p - y) % p
        return (x, y)
    x = (int.from_bytes(data[n:], "big"), int.from_bytes(data[:n], "big"))
    if inf:
        return None
    rhs = f2_add(f2_mul(f2_mul(x, x, p), x, p), curve.b2, p)
    y = f2_sqrt(rhs, p)
    if y is None:
        raise ValueError("Cannot deserialize point: not on curve")
    if _f2_gt(y, f2_neg(y, p)) != big:
        y = f2_neg(y, p)
    return (x, y)


# ----------------------------------------------------------------------------
# NTT definitions (src/bn254/polynomial.rs:535-571)
# ----------------------------------------------------------------------------

def next_pow2(n):
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


def dft_naive(vals, size, curve: CurveParams, inverse=False):
    """O(N^2) definition: out[i] = sum_j in[j] w^(ij); inverse includes 1/N."""
    r = curve.r
    n = next_pow2(size)
    w = curve.root_of_unity(n)
    if inverse:
        w = pow(w, -1, r)
    v = [x % r for x in vals] + [0] * (n - len(vals))
    out = []
    for i in range(n):
        wi = pow(w, i, r)
        acc = 0
        cur = 1
        for j in range(n):
            acc += v[j] * cur
            cur = cur * wi % r
        out.append(acc % r)
    if inverse:
        ninv = pow(n, -1, r)
        out = [x * ninv % r for x in out]
    return out


def ntt(vals, size, curve: CurveParams, inverse=False):
    """O(N log N) iterative radix-2 with natural-order in and out."""
    r = curve.r
    n = next_pow2(size)
    if len(vals) > n:
        # the reference passes the raw slice to EvaluationDomain::fft/ifft (src/bn254/polynomial.rs:541-542, 567-568);
        # ark-poly 0.4.2's radix-2 fft_in_place / ifft_in_place begin with `coeffs.resize(self.size(), zero)`, i.e. an
        # over-long input is truncated (only DensePolynomial::evaluate_over_domain folds modulo X^n - 1).  ark-poly is
        # not vendored in the reference tree: this row is parity-unpinned.
        vals = vals[:n]
    a = [x % r for x in vals] + [0] * (n - len(vals))
    if n == 1:
        return a
    w = curve.root_of_unity(n)
    if inverse:
        w = pow(w, -1, r)
    log_n = n.bit_length() - 1
    # bit reversal
    for i in range(n):
        j = int(bin(i)[2:].zfill(log_n)[::-1], 2)
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, r)
        half = length // 2
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * wl % r
        for start in range(0, n, length):
            for k in range(half):
                u = a[start + k]
                t = a[start + k + half] * tw[k] % r
                a[start + k] = (u + t) % r
                a[start + k + half] = (u - t) % r
        length <<= 1
    if inverse:
        ninv = pow(n, -1, r)
        a = [x * ninv % r for x in a]
    return a


def coset_ntt(vals, size, curve, inverse=False):
    """coset_fft/coset_ifft with offset = group generator (src/bn254/polynomial.rs:547-585)."""
    r = curve.r
    n = next_pow2(size)
    g = curve.root_of_unity(n)
    if not inverse:
        scaled = [(x % r) * pow(g, i, r) % r for i, x in enumerate(vals)]
        return ntt(scaled, n, curve)
    out = ntt(vals, n, curve, inverse=True)
    gi = pow(g, -1, r)
    return [x * pow(gi, i, r) % r for i, x in enumerate(out)]


def poly_eval(coeffs, x, r):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % r
    return acc


def strip_zeros(coeffs):
    """DensePolynomial::from_coefficients_vec strips trailing zeros."""
    c = list(coeffs)
    while c and c[-1] == 0:
        c.pop()
    return c


def divide_by_vanishing(coeffs, n, r):
    """(q, rem) with coeffs = q*(X^n - 1) + rem, deg rem < n (polynomial.rs:466-489)."""
    c = [x % r for x in coeffs]
    if len(c) < n:
        return [], strip_zeros(c)
    q = c[n:]
    for i in range(len(q) - 1 - n, -1, -1):
        q[i] = (q[i] + q[i + n]) % r
    rem = c[:n]
    for i in range(min(n, len(q))):
        rem[i] = (rem[i] + q[i]) % r
    return strip_zeros(q), strip_zeros(rem)


def pad_pair(a, b):
    """_pad_coeffs (python/zksnake/polynomial.py:126-148): lengths after zero padding."""
    da, db = len(a) - 1, len(b) - 1
    if da != db:
        length = next_pow2(max(da, db))
        if da > db:
            return a + [0] * length, b + [0] * (da + length - db)
        return a + [0] * (db + length - da), b + [0] * length
    return a + [0] * next_pow2(da), b + [0] * next_pow2(da)


def sparse_dot(triplets, n_row, vector, r):
    """SparseArray.dot (python/zksnake/array.py:36-44)."""
    out = [0] * n_row
    for row, col, val in triplets:
        out[row] += vector[col] * val
    return [x % r for x in out]


def qap_evaluate_witness(A, B, C, n_row, witness, curve):
    """QAP.evaluate_witness (python/zksnake/groth16/qap.py:42-71) on triplet lists.
    Returns coefficient lists (u, v, w, h) with trailing zeros stripped."""
    r = curve.r
    a = sparse_dot(A, n_row, witness, r)
    b = sparse_dot(B, n_row, witness, r)
    c = sparse_dot(C, n_row, witness, r)
    u = strip_zeros(ntt(a, n_row, curve, inverse=True))
    v = strip_zeros(ntt(b, n_row, curve, inverse=True))
    w = strip_zeros(ntt(c, n_row, curve, inverse=True))
    if not u or not v:
        uv = []
    else:
        pa, pb = pad_pair(u, v)
        fa = ntt(pa, len(pa), curve)
        fb = ntt(pb, len(pb), curve)
        m = max(len(fa), len(fb))
        fa += [0] * (m - len(fa))
        fb += [0] * (m - len(fb))
        uv = strip_zeros(ntt([x * y % r for x, y in zip(fa, fb)], m, curve, inverse=True))
    m = max(len(uv), len(w))
    hz = strip_zeros([((uv[i] if i < len(uv) else 0) - (w[i] if i < len(w) else 0)) % r for i in range(m)])
    h, rem = divide_by_vanishing(hz, n_row, r)
    if rem:
        raise ValueError("(U * V - W) did not divided by Z to zero")
    return u, v, w, h


# ----------------------------------------------------------------------------
# Pairing (only used to restate Groth16.verify; protocol.py:167-186).  Fp12 is
# represented as Fp[w]/(w^12 - c6 w^6 - c0); G2 is untwisted into E(Fp12).
# ----------------------------------------------------------------------------

class Fp12:
    __slots__ = ("c", "cv")

    def __init__(self, coeffs, cv):
        self.c = coeffs
        self.cv = cv

    @staticmethod
    def one(cv):
        return Fp12([1] + [0] * 11, cv)

    def __eq__(self, o):
        return self.c == o.c

    def mul(self, o):
        p = self.cv.p
        c6, c0 = self.cv.fp12_mod
        t = [0] * 23
        a, b = self.c, o.c
        for i in range(12):
            ai = a[i]
            if ai:
                for j in range(12):
                    t[i + j] += ai * b[j]
        for k in range(22, 11, -1):
            v = t[k]
            if v:
                t[k - 6] += v * c6
                t[k - 12] += v * c0
        return Fp12([x % p for x in t[:12]], self.cv)

    def pow(self, e):
        out = Fp12.one(self.cv)
        base = self
        while e:
            if e & 1:
                out = out.mul(base)
            base = base.mul(base)
            e >>= 1
        return out

    def add(self, o):
        p = self.cv.p
        return Fp12([(x + y) % p for x, y in zip(self.c, o.c)], self.cv)

    def sub(self, o):
        p = self.cv.p
        return Fp12([(x - y) % p for x, y in zip(self.c, o.c)], self.cv)

    def scal(self, k):
        p = self.cv.p
        return Fp12([x * k % p for x in self.c], self.cv)

    def inv(self):
        # a^(p^12 - 2) is too slow; solve via extended Euclid on polynomials over Fp
        p = self.cv.p
        c6, c0 = self.cv.fp12_mod
        mod = [(-c0) % p] + [0] * 5 + [(-c6) % p] + [0] * 5 + [1]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], mod

        def deg(x):
            d = len(x) - 1
            while d and x[d] == 0:
                d -= 1
            return d

        def pdiv(a, b):
            da, db = deg(a), deg(b)
            t = list(a)
            o = [0] * len(a)
            ib = pow(b[db], -1, p)
            for i in range(da - db, -1, -1):
                q = t[db + i] * ib % p
                o[i] = q
                for c in range(db + 1):
                    t[c + i] = (t[c + i] - q * b[c]) % p
            return o[: deg(o) + 1]

        while deg(low):
            r = pdiv(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % p
                    new[i + j] = (new[i + j] - low[i] * r[j]) % p
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, p)
        return Fp12([x * inv0 % p for x in lm[:12]], self.cv)


def _embed_fp(x, cv):
    return Fp12([x % cv.p] + [0] * 11, cv)


def _embed_fp2(a, cv):
    """Fp2 element c0 + c1*u as an Fp12 polynomial in w, where u is expressed through w^6:
    BN254: w^6 = 9 + u  -> u = w^6 - 9;  BLS12-381: w^6 = 1 + u -> u = w^6 - 1."""
    shift = 9 if cv.is_bn else 1
    c = [0] * 12
    c[0] = (a[0] - shift * a[1]) % cv.p
    c[6] = a[1] % cv.p
    return Fp12(c, cv)


def _untwist(Q, cv):
    """map a G2 point on the twist into E(Fp12)."""
    x = _embed_fp2(Q[0], cv)
    y = _embed_fp2(Q[1], cv)
    w = Fp12([0, 1] + [0] * 10, cv)
    w2 = w.mul(w)
    w3 = w2.mul(w)
    if cv.twist_type == 'D':
        return (x.mul(w2), y.mul(w3))
    return (x.mul(w2.inv()), y.mul(w3.inv()))


def _line(P1, P2, T):
    """evaluate at T the line through P1, P2 (points in E(Fp12), affine)."""
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = y2.sub(y1).mul(x2.sub(x1).inv())
        return m.mul(xt.sub(x1)).sub(yt.sub(y1))
    if y1 == y2:
        m = x1.mul(x1).scal(3).mul(y1.scal(2).inv())
        return m.mul(xt.sub(x1)).sub(yt.sub(y1))
    return xt.sub(x1)


def _e12_add(P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if y1 == y2:
            m = x1.mul(x1).scal(3).mul(y1.scal(2).inv())
        else:
            return None
    else:
        m = y2.sub(y1).mul(x2.sub(x1).inv())
    x3 = m.mul(m).sub(x1).sub(x2)
    y3 = m.mul(x1.sub(x3)).sub(y1)
    return (x3, y3)


def _frob_point(P, cv, k=1):
    return (P[0].pow(cv.p ** k), P[1].pow(cv.p ** k))


def miller_loop(cv: CurveParams, P, Q):
    """ate Miller loop f_{T,Q}(P) (not yet final-exponentiated)."""
    if P is None or Q is None:
        return Fp12.one(cv)
    Pt = (_embed_fp(P[0], cv), _embed_fp(P[1], cv))
    Qt = _untwist(Q, cv)
    R = Qt
    f = Fp12.one(cv)
    for bit in bin(cv.ate_loop)[3:]:
        f = f.mul(f).mul(_line(R, R, Pt))
        R = _e12_add(R, R)
        if bit == "1":
            f = f.mul(_line(R, Qt, Pt))
            R = _e12_add(R, Qt)
    if cv.is_bn:
        Q1 = _frob_point(Qt, cv, 1)
        nQ2 = _frob_point(Qt, cv, 2)
        nQ2 = (nQ2[0], Fp12.one(cv).scal(0).sub(nQ2[1]))
        f = f.mul(_line(R, Q1, Pt))
        R = _e12_add(R, Q1)
        f = f.mul(_line(R, nQ2, Pt))
    elif cv.ate_neg:
        # f_{-|x|} = 1/f_{|x|} up to factors killed by the final exponentiation
        f = f.inv()
    return f


def final_exp(cv, f):
    return f.pow((cv.p ** 12 - 1) // cv.r)


def pairing(cv, P, Q):
    return final_exp(cv, miller_loop(cv, P, Q))


def multi_pairing(cv, Ps, Qs):
    f = Fp12.one(cv)
    for P, Q in zip(Ps, Qs):
        f = f.mul(miller_loop(cv, P, Q))
    return final_exp(cv, f)


# ----------------------------------------------------------------------------
# Groth16 restated on plain integers / oracle points (protocol.py:32-186)
# ----------------------------------------------------------------------------

def lagrange_at(n, tau, curve):
    """evaluate_lagrange_coefficients (polynomial.rs:645-652): L_i(tau) for the size-n domain."""
    r = curve.r
    w = curve.root_of_unity(n)
    z = (pow(tau, n, r) - 1) % r
    if z == 0:
        out = [0] * n
        cur = 1
        for i in range(n):
            if cur == tau % r:
                out[i] = 1
            cur = cur * w % r
        return out
    ninv = pow(n, -1, r)
    out = []
    wi = 1
    for _ in range(n):
        out.append(z * ninv % r * wi % r * pow((tau - wi) % r, -1, r) % r)
        wi = wi * w % r
    return out


def groth16_setup(A, B, C, n_row, n_col, n_public, curve, toxic):
    """returns (pk, vk) dicts of oracle affine points given toxic = (tau, alpha, beta, gamma, delta)."""
    tau, alpha, beta, gamma, delta = toxic
    r = curve.r
    g1, g2 = G1(curve), G2(curve)
    lag = lagrange_at(n_row, tau, curve)
    L = [0] * n_col
    R = [0] * n_col
    O = [0] * n_col
    for row, col, v in A:
        L[col] += lag[row] * v
    for row, col, v in B:
        R[col] += lag[row] * v
    for row, col, v in C:
        O[col] += lag[row] * v
    K = [(L[i] * beta + R[i] * alpha + O[i]) % r for i in range(n_col)]
    t = (pow(tau, n_row, r) - 1) % r
    inv_gamma = pow(gamma, -1, r)
    inv_delta = pow(delta, -1, r)
    pw = [pow(tau, i, r) for i in range(n_row)]
    pk = dict(
        alpha_1=g1.mul(g1.gen, alpha), beta_1=g1.mul(g1.gen, beta), beta_2=g2.mul(g2.gen, beta),
        delta_1=g1.mul(g1.gen, delta), delta_2=g2.mul(g2.gen, delta),
        tau_1=[g1.mul(g1.gen, x) for x in pw],
        tau_2=[g2.mul(g2.gen, x) for x in pw],
        target_1=[g1.mul(g1.gen, x * t % r * inv_delta % r) for x in pw],
        kdelta_1=[g1.mul(g1.gen, k * inv_delta % r) for k in K[n_public:]],
    )
    vk = dict(
        alpha_1=pk["alpha_1"], beta_2=pk["beta_2"], gamma_2=g2.mul(g2.gen, gamma),
        delta_2=pk["delta_2"], ic=[g1.mul(g1.gen, k * inv_gamma % r) for k in K[:n_public]],
    )
    return pk, vk


def groth16_prove(pk, A, B, C, n_row, public_w, private_w, curve, rs):
    """Groth16.prove through the MSM/NTT definitions (protocol.py:115-165); rs = (r, s)."""
    rr, ss = rs
    q = curve.r
    g1, g2 = G1(curve), G2(curve)
    u, v, _, h = qap_evaluate_witness(A, B, C, n_row, public_w + private_w, curve)

    def mexp(g, pts, sc):
        if len(sc) == 0:
            return None
        return g.msm(pts[: len(sc)], sc)

    Ap = g1.add(g1.add(mexp(g1, pk["tau_1"], u), pk["alpha_1"]), g1.mul(pk["delta_1"], rr))
    B1 = g1.add(g1.add(mexp(g1, pk["tau_1"], v), pk["beta_1"]), g1.mul(pk["delta_1"], ss))
    B2 = g2.add(g2.add(mexp(g2, pk["tau_2"], v), pk["beta_2"]), g2.mul(pk["delta_2"], ss))
    HZ = mexp(g1, pk["target_1"], h)
    sdw = mexp(g1, pk["kdelta_1"], private_w) if private_w else None
    Cp = g1.add(HZ, sdw)
    Cp = g1.add(Cp, g1.mul(Ap, ss))
    Cp = g1.add(Cp, g1.mul(B1, rr))
    Cp = g1.add(Cp, g1.mul(g1.neg(pk["delta_1"]), rr * ss % q))
    return Ap, B2, Cp


def groth16_closed_form(A, B, C, n_row, n_col, n_public, witness, curve, toxic, rs):
    """the proof as discrete logs, no FFT/MSM involved. Returns (a, b, c) scalars."""
    tau, alpha, beta, gamma, delta = toxic
    rr, ss = rs
    q = curve.r
    lag = lagrange_at(n_row, tau, curve)
    Aw = sparse_dot(A, n_row, witness, q)
    Bw = sparse_dot(B, n_row, witness, q)
    Cw = sparse_dot(C, n_row, witness, q)
    U = sum(l * x for l, x in zip(lag, Aw)) % q
    V = sum(l * x for l, x in zip(lag, Bw)) % q
    W = sum(l * x for l, x in zip(lag, Cw)) % q
    L = [0] * n_col
    R = [0] * n_col
    O = [0] * n_col
    for row, col, val in A:
        L[col] += lag[row] * val
    for row, col, val in B:
        R[col] += lag[row] * val
    for row, col, val in C:
        O[col] += lag[row] * val
    K = [(L[i] * beta + R[i] * alpha + O[i]) % q for i in range(n_col)]
    inv_delta = pow(delta, -1, q)
    a = (alpha + U + rr * delta) % q
    b = (beta + V + ss * delta) % q
    priv = sum(witness[j] * K[j] for j in range(n_public, n_col)) % q
    c = ((U * V - W) + priv) % q * inv_delta % q
    c = (c + ss * a + rr * b - rr * ss % q * delta) % q
    return a, b, c


def groth16_verify(vk, proof, public_w, curve):
    """Groth16.verify (protocol.py:167-186)."""
    g1 = G1(curve)
    Ap, Bp, Cp = proof
    sgw = g1.msm(vk["ic"], public_w)
    lhs = pairing(curve, Ap, Bp)
    rhs = multi_pairing(curve, [vk["alpha_1"], sgw, Cp], [vk["beta_2"], vk["gamma_2"], vk["delta_2"]])
    return lhs == rhs


def proof_bytes(curve, proof):
    """Proof.to_bytes (groth16/serialization.py:40-42)."""
    Ap, Bp, Cp = proof
    return compress(curve, 1, Ap) + compress(curve, 2, Bp) + compress(curve, 1, Cp)


# ----------------------------------------------------------------------------
# deterministic inputs: SplitMix64
# ----------------------------------------------------------------------------

class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def field(self, r):
        """4 consecutive outputs as LE 64-bit limbs, reduced mod r."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % r


def readme_circuit(r):
    """config 1: y == x^3 + x + 5 with x = 3: wires [1, y, x, v1] = [1, 35, 3, 9]."""
    A = [(0, 2, 1), (1, 3, 1)]
    B = [(0, 2, 1), (1, 2, 1)]
    C = [(0, 3, 1), (1, 1, 1), (1, 0, r - 5), (1, 2, r - 1)]
    return A, B, C, 2, 4, 2, [1, 35, 3, 9]


def chain_circuit(n, r, inp=2):
    """benchmarks/benchmark_groth16.py:7-27 shape with n constraints (n power of two, n >= 2).
    wires [1, out, inp, v0..v_{n-2}];  v0 = inp*inp, v_i = v_{i-1}*inp, out = v_{n-2} * 1."""
    A, B, C = [], [], []
    nv = n - 1
    w = [1, 0, inp % r]
    cur = inp % r
    for i in range(nv):
        prev_col = 2 if i == 0 else 3 + i - 1
        A.append((i, prev_col, 1))
        B.append((i, 2, 1))
        C.append((i, 3 + i, 1))
        cur = cur * inp % r
        w.append(cur)
    A.append((n - 1, 3 + nv - 1, 1))
    B.append((n - 1, 0, 1))
    C.append((n - 1, 1, 1))
    w[1] = cur
    return A, B, C, n, 3 + nv, 2, w
