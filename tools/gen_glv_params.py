#!/usr/bin/env python3
"""Generate zksnake_amd/csrc/glv_params.h: the endomorphisms of the four groups and the constants of the scalar decomposition
k = k1 + lambda k2 (mod r) with |k1|, |k2| < 2^127, used by the general (not fixed-base) MSM plans: 2n half-length scalars
need half the windows, hence half the bucket sets to reduce and half the doublings in the tail, for the same number of bucket
additions.
    G1:  phi(x, y) = (beta x, y) = lambda (x, y),  beta a cube root of unity in Fp, lambda one in Fr
    G2:  psi^2(x, y) = (c x, -y) = mu^2 (x, y) on the twist, where psi = twist^-1 . Frobenius . twist acts as p mod r = mu:
         psi(x, y) = (conj(x) gx, conj(y) gy) with gx, gy sixth-root powers of the twist constant, so psi^2 multiplies x by the
         norm c = gx conj(gx) (an element of Fp) and y by gy conj(gy) = -1.  mu^2 gives a balanced two-dimensional split for
         both curves (for BLS12-381 mu itself is the 64-bit curve parameter: unbalanced).

    lattice  v1 = (a1, b1), v2 = (a2, b2),  a_i + b_i lambda = 0 (mod r),  b1 < 0 < b2
    c1 = round(b2 k / r),  c2 = round(-b1 k / r)        (computed as (k g_i + 2^319) >> 320, g_i = round(2^320 |b| / r))
    k1 = k - c1 a1 - c2 a2,  k2 = -c1 b1 - c2 b2        (any integers c1, c2 give a valid pair; rounding makes it short)

`decompose` below is the integer formula of the kernel (glv_digits_kernel, msm_impl.hip.h), word for word; the CPU tests
run it over random and extreme scalars to check the identity and the bound.  Run:  python tools/gen_glv_params.py"""
import os

SHIFT = 320
MASK128 = (1 << 128) - 1

CURVES = {
    "Bn254": dict(
        p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        b=3, gen=(1, 2)),
    "Bls381": dict(
        p=4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787,
        r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
        b=4, gen=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                  0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)),
}


G2_CURVES = {
    "Bn254G2": dict(
        base="Bn254", xi=(9, 1), inverse=False,
        b2=(19485874751759354771024239261021720505790618469301721065564631296452457478373,
            266929791119991161246907387137283842545076965332900288569378510910307636690),
        gen=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
              11559732032986387107991004021392285783925812861821192530917403151452391805634),
             (8495653923123431417604973247489272438418190587263600148770280649306958101930,
              4082367875863433681332203403145435568316851327593401208105741076214120093531))),
    "Bls381G2": dict(
        base="Bls381", xi=(1, 1), inverse=True, b2=(4, 4),
        gen=((352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
              3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
             (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
              927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582))),
}


# ---- Fp2 = Fp[u]/(u^2 + 1) and the twist curve over it (only what the checks below need)
def f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(a, p):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * d % p, -a[1] * d % p)


def f2_pow(a, e, p):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a, p)
        a = f2_mul(a, a, p)
        e >>= 1
    return r


def ec2_add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    sub = lambda a, b: ((a[0] - b[0]) % p, (a[1] - b[1]) % p)  # noqa: E731
    if P[0] == Q[0]:
        if ((P[1][0] + Q[1][0]) % p, (P[1][1] + Q[1][1]) % p) == (0, 0):
            return None
        xx = f2_mul(P[0], P[0], p)
        lam = f2_mul((3 * xx[0] % p, 3 * xx[1] % p), f2_inv((2 * P[1][0] % p, 2 * P[1][1] % p), p), p)
    else:
        lam = f2_mul(sub(Q[1], P[1]), f2_inv(sub(Q[0], P[0]), p), p)
    x = sub(sub(f2_mul(lam, lam, p), P[0]), Q[0])
    return (x, sub(f2_mul(lam, sub(P[0], x), p), P[1]))


def ec2_mul(P, k, p):
    acc = None
    while k:
        if k & 1:
            acc = ec2_add(acc, P, p)
        P = ec2_add(P, P, p)
        k >>= 1
    return acc


def ec_add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return (x, (lam * (P[0] - x) - P[1]) % p)


def ec_mul(P, k, p):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, P, p)
        P = ec_add(P, P, p)
        k >>= 1
    return acc


def cube_roots_of_unity(m):
    """the two primitive cube roots of unity mod the prime m (m = 1 mod 3)"""
    g = 2
    while True:
        w = pow(g, (m - 1) // 3, m)
        if w != 1:
            return w, w * w % m
        g += 1


def short_basis(r, lam):
    """two short vectors (a, b) with a + b lam = 0 mod r (extended Euclid stopped around sqrt(r), as in GLV)"""
    rows = [(r, 0), (lam, 1)]  # (remainder, coefficient t): remainder = t * lam (mod r)
    while rows[-1][0] * rows[-1][0] >= r:
        q = rows[-2][0] // rows[-1][0]
        rows.append((rows[-2][0] - q * rows[-1][0], rows[-2][1] - q * rows[-1][1]))
    (r0, t0), (r1, t1) = rows[-2], rows[-1]
    q = r0 // r1
    r2, t2 = r0 - q * r1, t0 - q * t1
    v1 = (r1, -t1)
    v2 = (r0, -t0) if r0 * r0 + t0 * t0 <= r2 * r2 + t2 * t2 else (r2, -t2)
    return v1, v2


def constants(name):
    cv = CURVES[name]
    p, r = cv["p"], cv["r"]
    G = cv["gen"]
    assert (G[1] * G[1] - G[0] ** 3 - cv["b"]) % p == 0 and ec_mul(G, r, p) is None
    betas, lams = cube_roots_of_unity(p), cube_roots_of_unity(r)
    lam = min(lams)  # either works; fix one
    lamG = ec_mul(G, lam, p)
    beta = next(b for b in betas if (b * G[0] % p, G[1]) == lamG)
    out = dict(name=name, p=p, r=r, beta=beta, lam=lam, neg_y=False)
    out.update(lattice(r, lam))
    return out


def lattice(r, lam):
    v1, v2 = short_basis(r, lam)
    if v1[1] > 0:
        v1 = (-v1[0], -v1[1])
    if v2[1] < 0:
        v2 = (-v2[0], -v2[1])
    if v1[1] > 0 or v2[1] < 0 or v1[1] == 0 or v2[1] == 0:
        raise SystemExit("basis orientation")
    for a, b in (v1, v2):
        assert (a + b * lam) % r == 0
    (a1, b1), (a2, b2) = v1, v2
    det = a1 * b2 - a2 * b1
    assert det in (r, -r)
    g1 = ((b2 << SHIFT) + r // 2) // r
    g2 = ((-b1 << SHIFT) + r // 2) // r
    assert g1 < 1 << 224 and g2 < 1 << 224
    if det == -r:
        # (k, 0) = c1 v1 + c2 v2 has c1 = k b2 / det, c2 = -k b1 / det: with det = -r both change sign.  The kernel computes the
        # non-negative roundings c1' = round(k b2 / r), c2' = round(-k b1 / r); k1 = k + c1' a1 + c2' a2 and k2 = c1' b1 + c2' b2 are
        # its formulas with the vectors negated.
        a1, b1, a2, b2 = -a1, -b1, -a2, -b2
    return dict(a1=a1, b1=b1, a2=a2, b2=b2, g1=g1, g2=g2)


def constants_g2(name):
    """psi^2 on the twist: (x, y) -> (c x, -y) with c in Fp, acting as lam = (p mod r)^2; checked on the generator"""
    g = G2_CURVES[name]
    cv = CURVES[g["base"]]
    p, r = cv["p"], cv["r"]
    G = g["gen"]
    lhs = f2_mul(G[1], G[1], p)
    x3 = f2_mul(f2_mul(G[0], G[0], p), G[0], p)
    assert lhs == ((x3[0] + g["b2"][0]) % p, (x3[1] + g["b2"][1]) % p) and ec2_mul(G, r, p) is None
    gx, gy = f2_pow(g["xi"], (p - 1) // 3, p), f2_pow(g["xi"], (p - 1) // 2, p)
    if g["inverse"]:   # M-type twist (BLS12-381): the inverse constants
        gx, gy = f2_inv(gx, p), f2_inv(gy, p)
    conj = lambda a: (a[0], -a[1] % p)  # noqa: E731
    mu = p % r
    psiG = (f2_mul(conj(G[0]), gx, p), f2_mul(conj(G[1]), gy, p))
    assert psiG == ec2_mul(G, mu, p), "psi does not act as p mod r"
    cx, cy = f2_mul(gx, conj(gx), p), f2_mul(gy, conj(gy), p)
    assert cx[1] == 0 and cy == (p - 1, 0), "psi^2 is not (c x, -y)"
    lam = mu * mu % r
    assert ((cx[0] * G[0][0] % p, cx[0] * G[0][1] % p), (-G[1][0] % p, -G[1][1] % p)) == ec2_mul(G, lam, p)
    out = dict(name=name, p=p, r=r, beta=cx[0], lam=lam, neg_y=True)
    out.update(lattice(r, lam))
    return out


def decompose(k, cs):
    """(k1, k2) as signed ints from the kernel's formula: everything after c1, c2 is arithmetic mod 2^128"""
    assert 0 <= k < cs["r"]
    c1 = ((k * cs["g1"] + (1 << (SHIFT - 1))) >> SHIFT) & MASK128
    c2 = ((k * cs["g2"] + (1 << (SHIFT - 1))) >> SHIFT) & MASK128
    k1 = (k - c1 * (cs["a1"] & MASK128) - c2 * (cs["a2"] & MASK128)) & MASK128
    k2 = (-c1 * (cs["b1"] & MASK128) - c2 * (cs["b2"] & MASK128)) & MASK128
    sign = lambda v: v - (1 << 128) if v >> 127 else v  # noqa: E731
    return sign(k1), sign(k2)


def words(x, n):
    x &= (1 << (32 * n)) - 1
    return "{" + ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(n)) + "}"


def main():
    out = ["// GENERATED by tools/gen_glv_params.py -- do not edit.",
           "// Endomorphisms of the four groups ((beta x, y) on G1, psi^2 = (c x, -y) on G2) and the short-lattice decomposition constants; see the generator.",
           "#pragma once", "#include <cstdint>", "namespace zkmi {",
           "struct GlvConsts {",
           "    uint32_t g1[7], g2[7];              // round(2^320 b2 / r), round(2^320 (-b1) / r)",
           "    uint32_t a1[4], b1[4], a2[4], b2[4];  // two's complement mod 2^128",
           "};"]
    for name in list(CURVES) + list(G2_CURVES):
        cs = constants(name) if name in CURVES else constants_g2(name)
        fq_words = (cs["p"].bit_length() + 31) // 32
        out.append("struct %sGlv {" % name)
        out.append("  // lambda = 0x%x" % cs["lam"])
        out.append("  // the endomorphism: (x, y) -> (BETA x, y) on G1, (BETA x, -y) on G2 (BETA in Fp multiplies both components of x)")
        out.append("  static constexpr bool NEG_Y = %s;" % ("true" if cs["neg_y"] else "false"))
        out.append("  static constexpr uint32_t BETA[%d] = %s;  // canonical" % (fq_words, words(cs["beta"], fq_words)))
        out.append("  static constexpr GlvConsts K = {%s, %s,\n                                  %s, %s,\n                                  %s, %s};" % (
            words(cs["g1"], 7), words(cs["g2"], 7), words(cs["a1"], 4), words(cs["b1"], 4), words(cs["a2"], 4), words(cs["b2"], 4)))
        out.append("};")
    out.append("}  // namespace zkmi")
    path = os.path.join(os.path.dirname(__file__), "..", "zksnake_amd", "csrc", "glv_params.h")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
