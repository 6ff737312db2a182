"""CPU: pins the oracle (the checker) -- pyref against public known-answer vectors and the committed golden
fixtures; the C++ restatement against pyref.  No product code is involved here."""

import json
import os
import random

import numpy as np
import pytest

from oracle import corc, pyref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CURVES = (("BN254", 0), ("BLS12_381", 1))


def test_public_known_answer_vectors():
    r = R.BN254.r
    # SURVEY Appendix A
    assert R.ntt([1, 2, 3, 4], 4, R.BN254) == [
        10, 8815841940592487685082627943775890807874194266836837569428, r - 2,
        21888242871839275213430563804664787403465736456640143535824009919738970926185]
    assert R.ntt([1, 2, 3, 4], 4, R.BLS12_381) == [
        10, 52435875175126190472517450856038661200138013439152152266063153762407857258495, R.BLS12_381.r - 2,
        6930289652147304637552539061375485556540504937530723926014]
    assert R.BN254.root_of_unity(1 << 20) == 17220337697351015657950521176323262483320249231368149235373741788599650842711
    assert R.BN254.root_of_unity(1 << 22) == 12143866164239048021030917283424216263377309185099704096317235600302831912062
    assert R.BLS12_381.root_of_unity(1 << 22) == 4859563557044021881916617240989566298388494151979623102977292742331120628579
    # public encodings: zcash BLS12-381 generators, ark BN254 generator
    assert R.compress(R.BLS12_381, 1, R.BLS12_381.g1).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    assert R.compress(R.BLS12_381, 2, R.BLS12_381.g2).hex() == (
        "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
    assert R.compress(R.BN254, 1, R.BN254.g1).hex() == "01" + "00" * 31
    assert R.compress(R.BLS12_381, 1, None).hex() == "c0" + "00" * 47
    # public keys of the BLS secret keys 2 and 3 (compressed 2 G1, 3 G1): they exercise the y-sign flag on both sides
    g1 = R.Group(R.BLS12_381, 1)
    assert R.compress(R.BLS12_381, 1, g1.mul(R.BLS12_381.g1, 2)).hex() == (
        "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e")
    assert R.compress(R.BLS12_381, 1, g1.mul(R.BLS12_381.g1, 3)).hex() == (
        "89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224")
    # 2 * (1, 2) on alt_bn128: the doubling vector of the EIP-196 precompile tests
    assert R.Group(R.BN254, 1).mul(R.BN254.g1, 2) == (
        0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3,
        0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4)


@pytest.mark.parametrize("name,cid", CURVES)
def test_group_laws_and_codec(name, cid):
    cv = R.curve_by_name(name)
    for grp in (1, 2):
        g = R.Group(cv, grp)
        assert g.is_on_curve(g.gen)
        assert R.ec_mul(g.F, g.gen, cv.r) is None
        assert g.mul(g.gen, cv.r - 1) == g.neg(g.gen)
        P, Q = g.mul(g.gen, 1234567), g.mul(g.gen, 7654321)
        assert g.add(P, Q) == g.mul(g.gen, 1234567 + 7654321)
        assert g.add(P, g.neg(P)) is None and g.add(P, P) == g.mul(P, 2)
        for X in (P, Q, None, g.neg(P)):
            assert R.decompress(cv, grp, R.compress(cv, grp, X)) == X


@pytest.mark.parametrize("name,cid", CURVES)
def test_pairing_is_bilinear(name, cid):
    cv = R.curve_by_name(name)
    g1, g2 = R.G1(cv), R.G2(cv)
    e = R.pairing(cv, g1.mul(g1.gen, 5), g2.mul(g2.gen, 7))
    assert e == R.pairing(cv, g1.gen, g2.gen).pow(35)
    assert e != R.Fp12.one(cv) and e.pow(cv.r) == R.Fp12.one(cv)


@pytest.mark.parametrize("name,cid", CURVES)
def test_groth16_definitions_agree(name, cid):
    """prove through NTT/MSM definitions == closed form; proof verifies; forged input fails
    (the reference's own test style, tests/test_groth16.py:68-144, test_r1cs_qap.py:9-109)"""
    cv = R.curve_by_name(name)
    toxic, rs = (11, 22, 33, 44, 55), (66, 77)
    for circ in (R.readme_circuit(cv.r), R.chain_circuit(8, cv.r)):
        A, B, C, n_row, n_col, n_pub, w = circ
        pk, vk = R.groth16_setup(A, B, C, n_row, n_col, n_pub, cv, toxic)
        pr = R.groth16_prove(pk, A, B, C, n_row, w[:n_pub], w[n_pub:], cv, rs)
        a, b, c = R.groth16_closed_form(A, B, C, n_row, n_col, n_pub, w, cv, toxic, rs)
        g1, g2 = R.G1(cv), R.G2(cv)
        assert pr == (g1.mul(g1.gen, a), g2.mul(g2.gen, b), g1.mul(g1.gen, c))
    assert R.groth16_verify(vk, pr, w[:n_pub], cv)
    assert not R.groth16_verify(vk, pr, [w[0], (w[1] + 1) % cv.r], cv)
    bad = list(w)
    bad[3] = (bad[3] + 1) % cv.r
    with pytest.raises(ValueError):
        R.qap_evaluate_witness(A, B, C, n_row, bad, cv)


def test_golden_oracle_vectors():
    with open(os.path.join(GOLD, "oracle_vectors.json")) as f:
        gold = json.load(f)
    for name, cid in CURVES:
        cv, o = R.curve_by_name(name), gold[name]
        vals = [int(v) for v in o["ntt16"]["in"]]
        assert [str(v) for v in R.ntt(vals, 16, cv)] == o["ntt16"]["fwd"]
        assert [str(v) for v in R.ntt(vals, 16, cv, inverse=True)] == o["ntt16"]["inv"]
        assert [str(v) for v in R.coset_ntt(vals, 16, cv)] == o["ntt16"]["coset_fwd"]
        assert corc.limbs_to_ints(corc.ntt(cid, corc.ints_to_limbs(vals))) == [int(v) for v in o["ntt16"]["fwd"]]
        for n, w in o["root_of_unity"].items():
            assert cv.root_of_unity(int(n)) == int(w)
        for grp in (1, 2):
            g, og = R.Group(cv, grp), o[f"g{grp}"]
            pts = [R.decompress(cv, grp, bytes.fromhex(h)) for h in og["bases_compressed"]]
            sc = [int(s) for s in og["scalars"]]
            assert R.compress(cv, grp, g.msm(pts, sc)).hex() == og["msm_compressed"]
            assert g.mul(g.gen, int(og["msm_dlog"])) == g.msm(pts, sc)
            got = corc.msm(cid, grp, corc.ints_to_limbs(sc), corc.points_to_limbs(pts, cid, grp))
            assert R.compress(cv, grp, corc.limbs_to_points(got, cid, grp)[0]).hex() == og["msm_compressed"]


@pytest.mark.parametrize("name,cid", CURVES)
def test_cpp_restatement_matches_definitions(name, cid):
    cv = R.curve_by_name(name)
    rnd = random.Random(17)
    for log_n in (0, 1, 4, 9):
        n = 1 << log_n
        v = [rnd.randrange(cv.r) for _ in range(n)]
        assert corc.limbs_to_ints(corc.ntt(cid, corc.ints_to_limbs(v))) == R.ntt(v, n, cv)
        assert corc.limbs_to_ints(corc.ntt(cid, corc.ints_to_limbs(v), inverse=True)) == R.ntt(v, n, cv, inverse=True)
    assert R.dft_naive(v, n, cv) == R.ntt(v, n, cv)
    # single outputs from the definition (the checker of the 2^28 GPU test): small sizes against the naive DFT, a size with
    # several 65536-element chunks against the full transform of the restatement, forward and inverse, two thread counts
    vl = corc.ints_to_limbs(v)
    for k in (0, 1, n // 2, n - 1):
        assert corc.limbs_to_ints(corc.ntt_spot(cid, vl, k).reshape(1, 4))[0] == R.dft_naive(v, n, cv)[k]
        assert corc.limbs_to_ints(corc.ntt_spot(cid, vl, k, inverse=True).reshape(1, 4))[0] == R.ntt(v, n, cv, inverse=True)[k]
    big = np.random.default_rng(5).integers(0, 1 << 62, size=(1 << 18, 4), dtype=np.uint64)
    full, full_inv = corc.ntt(cid, big, threads=4), corc.ntt(cid, big, inverse=True, threads=4)
    for k in (0, 3, 65536, (1 << 17) + 12345, (1 << 18) - 1):
        assert (corc.ntt_spot(cid, big, k, threads=1 + (k & 3)) == full[k]).all()
        assert (corc.ntt_spot(cid, big, k, inverse=True, threads=3) == full_inv[k]).all()
    a = [rnd.randrange(cv.r) for _ in range(10)]
    b = [rnd.randrange(cv.r) for _ in range(10)]
    for op, fn in (("mul", lambda x, y: x * y), ("add", lambda x, y: x + y), ("sub", lambda x, y: x - y)):
        assert corc.limbs_to_ints(corc.vec_op(cid, op, corc.ints_to_limbs(a), corc.ints_to_limbs(b))) == [fn(x, y) % cv.r for x, y in zip(a, b)]
    for threads in (1, 3):   # 10 pairs on 3 threads: ragged shares; values above r are reduced first
        assert corc.dot(cid, corc.ints_to_limbs(a), corc.ints_to_limbs(b), threads) == sum(x * y for x, y in zip(a, b)) % cv.r
    top = [(1 << 256) - 1 - i for i in range(5)]
    assert corc.dot(cid, corc.ints_to_limbs(top), corc.ints_to_limbs(top[::-1]), 2) == sum(x * y for x, y in zip(top, top[::-1])) % cv.r
    for grp in (1, 2):
        g = R.Group(cv, grp)
        ks = [rnd.randrange(cv.r) for _ in range(20)] + [0, 1, cv.r - 1]
        pts_l = corc.batch_mul(cid, grp, corc.ints_to_limbs(ks), corc.points_to_limbs([g.gen], cid, grp)[0])
        pts = corc.limbs_to_points(pts_l, cid, grp)
        assert pts[0] == g.mul(g.gen, ks[0]) and pts[-3] is None and pts[-2] == g.gen
        sc = [rnd.randrange(cv.r) for _ in ks]
        sc[0], sc[1], sc[2] = 0, 1, cv.r - 1       # the top digit of r - 1 once overflowed the bucket array
        pts[5] = pts[4]
        pts[7] = g.neg(pts[6]); sc[7] = sc[6]
        for c in (0, 3, 5, 13, 15, 17):
            got = corc.limbs_to_points(corc.msm(cid, grp, corc.ints_to_limbs(sc), corc.points_to_limbs(pts, cid, grp), c=c), cid, grp)[0]
            assert got == g.msm(pts, sc), (grp, c)
    A, B, C, n_row, n_col, n_pub, w = R.chain_circuit(16, cv.r)
    av, bv, cvv = (corc.ints_to_limbs(R.sparse_dot(M, n_row, w, cv.r)) for M in (A, B, C))
    u, v, h = corc.qap_h(cid, av, bv, cvv)
    ru, rv, _, rh = R.qap_evaluate_witness(A, B, C, n_row, w, cv)
    assert R.strip_zeros(corc.limbs_to_ints(u)) == ru and R.strip_zeros(corc.limbs_to_ints(v)) == rv
    assert R.strip_zeros(corc.limbs_to_ints(h)) == rh
    assert corc.ark_window(1 << 20) == 15 and corc.ark_window(1 << 23) == 17 and corc.ark_window(31) == 3
