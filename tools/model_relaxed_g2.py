"""Integer model of the relaxed-range G2 mixed addition (csrc/curve.hip.h, Fp2 branch of xyzz_add_affine_mem): replays the
limb operations on Python integers with every limb, column sum and value bound asserted, on inputs pushed to the ends of
their ranges, and checks the results against plain modular arithmetic.  N = 9 (BN254 Fq) and N = 14 (BLS12-381 Fq)."""
import random
B = 29
MASK = (1 << B) - 1


class Field:
    def __init__(self, p):
        self.p = p
        self.N = -(-p.bit_length() // B)
        if self.N * B - p.bit_length() < 6:
            self.N += 1
        self.R = 1 << (B * self.N)
        self.M = self.limbs(p)
        self.INV = (-pow(p, -1, 1 << B)) % (1 << B)
        self.maxcol = 0

    def limbs(self, v):
        out = [(v >> (B * i)) & MASK for i in range(self.N - 1)]
        out.append(v >> (B * (self.N - 1)))
        assert out[-1] < (1 << 32)
        return out

    @staticmethod
    def val(l):
        return sum(x << (B * i) for i, x in enumerate(l))

    def mulk(self, pairs):
        """(sum of a*b over the pairs) / R with one Montgomery reduction, product scanning as in field.hip.h"""
        N = self.N
        acc, m, r = 0, [0] * N, [0] * N
        for k in range(2 * N - 1):
            lo, hi = (0, k) if k < N else (k - N + 1, N - 1)
            for a, b in pairs:
                for i in range(lo, hi + 1):
                    acc += a[i] * b[k - i]
            if k < N:
                for i in range(k):
                    acc += m[i] * self.M[k - i]
                m[k] = ((acc & 0xFFFFFFFF) * self.INV) & MASK
                acc += m[k] * self.M[0]
                assert acc & MASK == 0
            else:
                for i in range(k - N + 1, N):
                    acc += m[i] * self.M[k - i]
                r[k - N] = acc & MASK
            assert acc < (1 << 64), (k, acc.bit_length())
            self.maxcol = max(self.maxcol, acc)
            acc >>= B
        r[N - 1] = acc
        assert acc < (1 << 32)
        return r

    def kp(self, k):
        return self.limbs(k * self.p)

    def sub_k(self, a, b, K):
        """a - b + K p, signed carries, normalised result (value must be >= 0)"""
        c, r, kp = 0, [], self.kp(K)
        for i in range(self.N):
            x = a[i] - b[i] + kp[i] + c
            assert -(1 << 31) <= x < (1 << 31)
            if i < self.N - 1:
                r.append(x & MASK)
                c = x >> B
            else:
                assert x >= 0, "negative value"
                r.append(x)
        return r

    def add_nosel(self, a, b):
        c, r = 0, []
        for i in range(self.N):
            x = a[i] + b[i] + c
            assert x < (1 << 32)
            if i < self.N - 1:
                r.append(x & MASK)
                c = x >> B
            else:
                r.append(x)
        return r

    def neg_lazy(self, b, K):
        kp = self.kp(K)
        r = [kp[i] + ((1 << B) if i < self.N - 1 else 0) - (1 if i > 0 else 0) - b[i] for i in range(self.N)]
        assert all(0 <= x < (1 << 30) for x in r[:-1]) and r[-1] >= 0, r
        return r

    def dbl_lazy(self, a):
        return [2 * x for x in a]

    def sel4(self, t, q):
        """t - 2q brought into [0, 4p) for t in [0, 4p), q in [0, 2p)"""
        v = self.val(t) - 2 * self.val(q)
        if v < 0:
            v += 4 * self.p
        assert 0 <= v < 4 * self.p
        return self.limbs(v)


def relaxed_add(f, X, Y, ZZ, ZZZ, qx, qy, use_mul4):
    """all arguments: pairs of limb lists (c0, c1).  X comps < 4p, the rest < 2p.  Returns (X3, Y3, ZZ3, ZZZ3)."""
    p = f.p
    v = f.val

    def mul2c(a, b, Ka1):
        """Fp2 product: a.c1 < (Ka1 - 1) p is negated lazily against Ka1 p"""
        na1 = f.neg_lazy(a[1], Ka1)
        return (f.mulk([(a[0], b[0]), (na1, b[1])]), f.mulk([(a[0], b[1]), (a[1], b[0])]))

    def sqr(a, Kd):
        """a.c0, a.c1 < Kd p / ... : ((a0 + a1)(a0 - a1 + Kd p), (2 a0) a1)"""
        ts = f.add_nosel(a[0], a[1])
        td = f.sub_k(a[0], a[1], Kd)
        return (f.mulk([(ts, td)]), f.mulk([(f.dbl_lazy(a[0]), a[1])]))

    U2 = mul2c(qx, ZZ, 4)
    S2 = mul2c(qy, ZZZ, 4)
    Pd = tuple(f.sub_k(U2[i], X[i], 4) for i in range(2))      # < 6p
    R = tuple(f.sub_k(S2[i], Y[i], 2) for i in range(2))       # < 4p
    assert all(v(c) < 6 * p for c in Pd) and all(v(c) < 4 * p for c in R)
    PP = sqr(Pd, 8)
    RR = sqr(R, 4)
    PPP = mul2c(Pd, PP, 8)
    Q = mul2c(X, PP, 8)
    for t in (PP, RR, PPP, Q):
        assert all(v(c) < 2 * p for c in t)
    X3 = tuple(f.sel4(f.sub_k(RR[i], PPP[i], 2), Q[i]) for i in range(2))
    D = tuple(f.sub_k(Q[i], X3[i], 4) for i in range(2))       # < 6p
    if use_mul4:
        Y3 = (f.mulk([(R[0], D[0]), (f.neg_lazy(R[1], 8), D[1]), (f.neg_lazy(Y[0], 4), PPP[0]), (Y[1], PPP[1])]),
              f.mulk([(R[0], D[1]), (R[1], D[0]), (f.neg_lazy(Y[0], 4), PPP[1]), (f.neg_lazy(Y[1], 4), PPP[0])]))
    else:
        t1 = mul2c(R, D, 8)
        t2 = mul2c(Y, PPP, 4)
        Y3 = tuple(f.limbs((v(t1[i]) - v(t2[i])) % (2 * p)) for i in range(2))   # fp_sub: exact range selection
    ZZ3 = mul2c(ZZ, PP, 4)
    ZZZ3 = mul2c(ZZZ, PPP, 4)
    for t in (Y3, ZZ3, ZZZ3):
        assert all(v(c) < 2 * p for c in t)
    return X3, Y3, ZZ3, ZZZ3


def check(p, use_mul4, trials=3000, seed=3):
    f = Field(p)
    rnd = random.Random(seed)
    Rinv = pow(f.R, -1, p)

    def rv(bound):
        m = rnd.random()
        if m < 0.25:
            return bound * p - 1 - rnd.randrange(1 << 16)
        if m < 0.35:
            return rnd.randrange(1 << 16)
        if m < 0.45:
            return (bound * p >> (B * (f.N - 1)) << (B * (f.N - 1))) - 1 - rnd.randrange(4)   # just below a top-limb step
        return rnd.randrange(bound * p)

    def f2mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) * Rinv % p, (a[0] * b[1] + a[1] * b[0]) * Rinv % p)

    for _ in range(trials):
        ints = dict(X=(rv(4), rv(4)), Y=(rv(2), rv(2)), ZZ=(rv(2), rv(2)), ZZZ=(rv(2), rv(2)), qx=(rv(2), rv(2)), qy=(rv(2), rv(2)))
        L = {k: tuple(f.limbs(c) for c in t) for k, t in ints.items()}
        X3, Y3, ZZ3, ZZZ3 = relaxed_add(f, L["X"], L["Y"], L["ZZ"], L["ZZZ"], L["qx"], L["qy"], use_mul4)
        # reference: madd-2008-s on Montgomery representatives (every product carries 1/R)
        U2, S2 = f2mul(ints["qx"], ints["ZZ"]), f2mul(ints["qy"], ints["ZZZ"])
        Pd = tuple((U2[i] - ints["X"][i]) % p for i in range(2))
        Rr = tuple((S2[i] - ints["Y"][i]) % p for i in range(2))
        PP, RRr = f2mul(Pd, Pd), f2mul(Rr, Rr)
        PPP, Q = f2mul(Pd, PP), f2mul(ints["X"], PP)
        eX = tuple((RRr[i] - PPP[i] - 2 * Q[i]) % p for i in range(2))
        t1, t2 = f2mul(Rr, tuple((Q[i] - eX[i]) % p for i in range(2))), f2mul(ints["Y"], PPP)
        eY = tuple((t1[i] - t2[i]) % p for i in range(2))
        eZZ, eZZZ = f2mul(ints["ZZ"], PP), f2mul(ints["ZZZ"], PPP)
        for got, exp in ((X3, eX), (Y3, eY), (ZZ3, eZZ), (ZZZ3, eZZZ)):
            assert tuple(f.val(c) % p for c in got) == exp
        assert all(f.val(c) < 4 * p for c in X3)
    import math
    print(f"ok: N={f.N} R/p={f.R / p:.1f} mul4={use_mul4} max column 2^{math.log2(f.maxcol):.3f}")


if __name__ == "__main__":
    check(21888242871839275222246405745257275088696311157297823662689037894645226208583, True)
    check(21888242871839275222246405745257275088696311157297823662689037894645226208583, False)
    check(0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, False)
