"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's vanilla PlonK on plain Python integers.
Nothing in zksnake_amd/ may import this module; only tests/ use it, as the checker.

Follows python/zksnake/plonk/protocol.py of the reference step by step (setup :39-155, prove :157-484,
challenge replay :486-538, verify :540-647) and python/zksnake/transcript.py:28-71 (blake2b Fiat-Shamir
transcript, including its integer encoding `int.to_bytes(d, d.bit_length(), "big")`).  Polynomials are
dense coefficient lists over Fr with schoolbook products -- deliberately unlike the product, which works
on coset evaluations on the GPU -- and, because the tests know tau, a commitment is P(tau) * G1 (one
scalar multiplication instead of an MSM), so no code is shared with the MSM under test either.

PARITY UNPINNED: the reference holds no golden PlonK proof (tests/test_plonk.py only checks that its own
proofs verify), and its Rust extension cannot be built here.  What pins this file: the verifier equation
(`verify` below, with the oracle's pairing) accepts the proofs and rejects tampered ones.
"""

import hashlib

from . import pyref as R

K1, K2 = 2, 3


# ---- dense polynomials over Fr (lists, lowest degree first) ------------------------------------
def _strip(a):
    a = list(a)
    while a and a[-1] == 0:
        a.pop()
    return a


def padd(a, b, r):
    n = max(len(a), len(b))
    return _strip([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % r for i in range(n)])


def pscale(a, s, r):
    return _strip([x * s % r for x in a])


def psub(a, b, r):
    return padd(a, pscale(b, r - 1, r), r)


def pmul(a, b, r):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % r
    return _strip(out)


def peval(a, x, r):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % r
    return acc


def pdiv_linear(a, root, r):
    """a = q * (X - root) + rem"""
    q, carry = [0] * max(len(a) - 1, 0), 0
    for i in range(len(a) - 1, 0, -1):
        carry = (a[i] + carry * root) % r
        q[i - 1] = carry
    rem = ((a[0] if a else 0) + carry * root) % r
    return _strip(q), rem


def times_vanishing(a, n, r):
    """a * (X^n - 1)"""
    return psub([0] * n + list(a), a, r)


def interpolate(evals, cv):
    return _strip(R.ntt(list(evals), len(evals), cv, inverse=True))


# ---- transcript (transcript.py:28-71) -----------------------------------------------------------
class Transcript:
    def __init__(self, cv):
        self.cv = cv
        self.h = hashlib.new("blake2b", b"")

    def point(self, P):
        self.h.update(R.compress(self.cv, 1, P))

    def scalar(self, d):
        self.h.update(int.to_bytes(d, d.bit_length(), "big"))

    def challenge(self):
        digest = self.h.digest()
        self.h = hashlib.new("blake2b", digest)
        return int.from_bytes(digest, "big") % self.cv.r


# ---- protocol -----------------------------------------------------------------------------------
def roots_of(n, cv):
    w = R.ntt([0, 1] + [0] * (n - 2), n, cv) if n > 1 else [1]
    return w  # NTT of X = [omega^i]


def setup(gates, permutation, n, cv, tau):
    """gates = dict L,R,O,M,C -> n values.  Returns (pk, vk) as dicts."""
    r = cv.r
    g1, g2 = R.G1(cv), R.G2(cv)
    roots = roots_of(n, cv)
    ids = roots + [K1 * w % r for w in roots] + [K2 * w % r for w in roots]
    sigma = [[ids[permutation[i + j * n]] for i in range(n)] for j in range(3)]
    Q = {k: interpolate(gates[k], cv) for k in "LROMC"}
    S = [interpolate(s, cv) for s in sigma]
    commit = lambda p: g1.mul(g1.gen, peval(p, tau, r))  # noqa: E731
    pk = {
        "n": n, "Q": Q, "S": S, "tau": tau, "roots": roots,
        "tau_Q": {k: commit(Q[k]) for k in "LROMC"}, "tau_S": [commit(s) for s in S],
    }
    vk = {"n": n, "tau_g2": g2.mul(g2.gen, tau), "tau_Q": pk["tau_Q"], "tau_S": pk["tau_S"], "roots": roots}
    return pk, vk


def _seed_transcript(cv, tau_Q, tau_S, public):
    t = Transcript(cv)
    for k in "LROMC":
        t.point(tau_Q[k])
    for s in tau_S:
        t.point(s)
    for _, v in public.items():
        t.scalar(v)
    return t


def prove(pk, public, private, cv, blind):
    """public: dict row -> value; private: flat [a0, b0, c0, a1, ...]; blind: the 11 scalars the reference draws,
    in its order (2 + 2 + 2 for the wires, 3 for z, 2 for the quotient split)."""
    r, n, tau = cv.r, pk["n"], pk["tau"]
    g1 = R.G1(cv)
    commit = lambda p: g1.mul(g1.gen, peval(p, tau, r))  # noqa: E731
    roots = pk["roots"]
    pad = lambda v: list(v) + [0] * (n - len(v))  # noqa: E731
    a, b, c = pad(private[0::3]), pad(private[1::3]), pad(private[2::3])
    pi = [0] * n
    for k, v in public.items():
        pi[k] = v
    t = _seed_transcript(cv, pk["tau_Q"], pk["tau_S"], public)
    Q, S = pk["Q"], pk["S"]
    ID = [[0, 1], [0, K1], [0, K2]]

    # round 1
    A = padd(interpolate(a, cv), times_vanishing(_strip(blind[0:2]), n, r), r)
    B = padd(interpolate(b, cv), times_vanishing(_strip(blind[2:4]), n, r), r)
    C = padd(interpolate(c, cv), times_vanishing(_strip(blind[4:6]), n, r), r)
    PI = interpolate(pi, cv)
    G = padd(padd(padd(pmul(A, Q["L"], r), pmul(B, Q["R"], r), r), padd(pmul(C, Q["O"], r), pmul(pmul(A, B, r), Q["M"], r), r), r),
             padd(Q["C"], PI, r), r)
    tau_a, tau_b, tau_c = commit(A), commit(B), commit(C)
    for P in (tau_a, tau_b, tau_c):
        t.point(P)

    # round 2
    beta, gamma = t.challenge(), t.challenge()
    shift = lambda W, X: padd(padd(W, pscale(X, beta, r), r), [gamma], r)  # noqa: E731
    nom = pmul(pmul(shift(A, ID[0]), shift(B, ID[1]), r), shift(C, ID[2]), r)
    den = pmul(pmul(shift(A, S[0]), shift(B, S[1]), r), shift(C, S[2]), r)
    acc = [1]
    for i in range(n):
        acc.append(acc[-1] * peval(nom, roots[i], r) * pow(peval(den, roots[i], r), -1, r) % r)
    assert acc.pop() == 1, "Copy constraints are not satisfied"
    Z = padd(times_vanishing(_strip(blind[6:9]), n, r), interpolate(acc, cv), r)
    tau_z = commit(Z)
    t.point(tau_z)

    # round 3
    alpha = t.challenge()
    Zw = [cf * roots[i % n] % r for i, cf in enumerate(Z)]
    L1 = interpolate([1] + [0] * (n - 1), cv)
    numer = padd(padd(G, pscale(psub(pmul(nom, Z, r), pmul(den, Zw, r), r), alpha, r), r),
                 pscale(pmul(psub(Z, [1], r), L1, r), alpha * alpha % r, r), r)
    T, rem = R.divide_by_vanishing(numer, n, r)
    assert not _strip(rem), "quotient has a remainder"
    T = _strip(T)
    T_lo, T_mid, T_hi = T[:n], T[n:2 * n], T[2 * n:]
    Xn = [0] * n + [1]
    T_lo = padd(T_lo, pscale(Xn, blind[9], r), r)
    T_mid = padd(psub(T_mid, [blind[9]], r), pscale(Xn, blind[10], r), r)
    T_hi = psub(T_hi, [blind[10]], r)
    tau_t = [commit(T_lo), commit(T_mid), commit(T_hi)]
    for P in tau_t:
        t.point(P)

    # round 4
    zeta = t.challenge()
    za, zb, zc = peval(A, zeta, r), peval(B, zeta, r), peval(C, zeta, r)
    zs1, zs2, zzw = peval(S[0], zeta, r), peval(S[1], zeta, r), peval(Zw, zeta, r)
    L1z = peval(L1, zeta, r)
    zh = (pow(zeta, n, r) - 1) % r
    lin = padd(padd(padd(pscale(Q["L"], za, r), pscale(Q["R"], zb, r), r), padd(pscale(Q["O"], zc, r), pscale(Q["M"], za * zb % r, r), r), r),
               padd(Q["C"], [peval(PI, zeta, r)], r), r)
    f1 = (za + beta * zeta + gamma) * (zb + beta * K1 * zeta + gamma) * (zc + beta * K2 * zeta + gamma) % r
    f2 = (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) * zzw % r
    perm = psub(pscale(Z, f1, r), pscale(padd(pscale(S[2], beta, r), [(zc + gamma) % r], r), f2, r), r)
    tsum = padd(padd(T_lo, pscale(T_mid, pow(zeta, n, r), r), r), pscale(T_hi, pow(zeta, 2 * n, r), r), r)
    Rpoly = psub(padd(padd(lin, pscale(perm, alpha, r), r), pscale(psub(Z, [1], r), alpha * alpha * L1z % r, r), r), pscale(tsum, zh, r), r)
    for s in (za, zb, zc, zs1, zs2, zzw):
        t.scalar(s)

    # round 5
    v = t.challenge()
    W = Rpoly
    for k, (poly, val) in enumerate(((A, za), (B, zb), (C, zc), (S[0], zs1), (S[1], zs2)), start=1):
        W = padd(W, pscale(psub(poly, [val], r), pow(v, k, r), r), r)
    Wz, rem = pdiv_linear(W, zeta, r)
    assert rem == 0
    Wzw, rem = pdiv_linear(psub(Z, [zzw], r), zeta * roots[1] % r, r)
    assert rem == 0
    return {"points": [tau_a, tau_b, tau_c, tau_z] + tau_t + [commit(Wz), commit(Wzw)], "scalars": [za, zb, zc, zs1, zs2, zzw]}


def proof_bytes(proof, cv):
    """Proof.to_bytes of plonk/serialization.py:101-126: nine compressed G1 points, six 32-byte LE scalars"""
    return b"".join(R.compress(cv, 1, P) for P in proof["points"]) + b"".join(s.to_bytes(32, "little") for s in proof["scalars"])


def verify(vk, proof, public, cv):
    r, n = cv.r, vk["n"]
    g1, g2 = R.G1(cv), R.G2(cv)
    tau_a, tau_b, tau_c, tau_z, t_lo, t_mid, t_hi, Wz, Wzw = proof["points"]
    za, zb, zc, zs1, zs2, zzw = proof["scalars"]
    t = _seed_transcript(cv, vk["tau_Q"], vk["tau_S"], public)
    for P in (tau_a, tau_b, tau_c):
        t.point(P)
    beta, gamma = t.challenge(), t.challenge()
    t.point(tau_z)
    alpha = t.challenge()
    for P in (t_lo, t_mid, t_hi):
        t.point(P)
    zeta = t.challenge()
    for s in proof["scalars"]:
        t.scalar(s)
    v = t.challenge()
    t.point(Wz)
    t.point(Wzw)
    u = t.challenge()

    omega = vk["roots"][1]
    zh = (pow(zeta, n, r) - 1) % r

    def bary(sparse):
        tot = 0
        for i, val in sparse.items():
            wi = pow(omega, i, r)
            tot += val * wi * pow(zeta - wi, -1, r)
        return zh * pow(n, -1, r) * tot % r

    L1z, PIz = bary({0: 1}), bary(public)
    r0 = (PIz - L1z * alpha * alpha - (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) * (zc + gamma) * zzw * alpha) % r
    mul, add = g1.mul, g1.add
    Qc, Sc = vk["tau_Q"], vk["tau_S"]
    D = add(add(add(mul(Qc["M"], za * zb % r), mul(Qc["L"], za)), add(mul(Qc["R"], zb), mul(Qc["O"], zc))), Qc["C"])
    zcoef = ((za + beta * zeta + gamma) * (zb + beta * K1 * zeta + gamma) * (zc + beta * K2 * zeta + gamma) * alpha + L1z * alpha * alpha + u) % r
    D = add(D, mul(tau_z, zcoef))
    D = add(D, g1.neg(mul(Sc[2], (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) * alpha * beta * zzw % r)))
    tsum = add(add(t_lo, mul(t_mid, pow(zeta, n, r))), mul(t_hi, pow(zeta, 2 * n, r)))
    D = add(D, g1.neg(mul(tsum, zh)))
    F = D
    for k, P in enumerate((tau_a, tau_b, tau_c, Sc[0], Sc[1]), start=1):
        F = add(F, mul(P, pow(v, k, r)))
    e = (-r0 + v * za + pow(v, 2, r) * zb + pow(v, 3, r) * zc + pow(v, 4, r) * zs1 + pow(v, 5, r) * zs2 + u * zzw) % r
    E = mul(g1.gen, e)
    lhs = R.pairing(cv, add(Wz, mul(Wzw, u)), vk["tau_g2"])
    rhs = R.pairing(cv, add(add(add(mul(Wz, zeta), mul(Wzw, u * zeta * omega % r)), F), g1.neg(E)), g2.gen)
    return lhs == rhs


# ---- a circuit that needs no symbolic front end -------------------------------------------------
def chain_gates(n, r, inp=2):
    """PlonK twin of the Groth16 benchmark chain (benchmarks/benchmark_groth16.py:7-27): v_0 = inp * inp,
    v_i = v_{i-1} * inp, the last value public.  Row i < n-1: a * b - c = 0 (qM = 1, qO = -1);
    row n-1: a - out = 0 with `out` public (qL = 1, PI = -out).  Copy constraints tie every b (and a_0) to
    `inp` and c_i to a_{i+1}.  Returns (gates, permutation, public dict, flat private witness)."""
    assert n >= 2 and n & (n - 1) == 0
    vals = [inp * inp % r]
    for _ in range(n - 2):
        vals.append(vals[-1] * inp % r)
    a = [inp] + vals[:-1] + [vals[-1]]
    b = [inp] * (n - 1) + [0]
    c = vals + [0]
    gates = {"L": [0] * (n - 1) + [1], "R": [0] * n, "O": [r - 1] * (n - 1) + [0], "M": [1] * (n - 1) + [0], "C": [0] * n}
    perm = list(range(3 * n))
    inp_cycle = [0] + [n + i for i in range(n - 1)]          # a_0 and all live b_i
    for k, pos in enumerate(inp_cycle):
        perm[pos] = inp_cycle[(k + 1) % len(inp_cycle)]
    for i in range(n - 1):                                   # c_i <-> a_{i+1}
        perm[2 * n + i], perm[i + 1] = i + 1, 2 * n + i
    private = [x for row in zip(a, b, c) for x in row]
    public = {n - 1: (-vals[-1]) % r}
    return gates, perm, public, private
