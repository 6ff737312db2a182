"""GPU parity for the PlonK row: with tau and the eleven blinding scalars pinned, the proof bytes of
zksnake_amd.plonk equal those of the oracle's coefficient-form restatement of the reference prover
(oracle/plonk_ref.py), proofs verify under both verifiers, forged ones do not, byte layouts round-trip
(the reference's own test style, tests/test_plonk.py:90-150)."""

import pytest

from oracle import plonk_ref as PR
from oracle import pyref
from zksnake_amd.arithmetization import Plonkish
from zksnake_amd.plonk import Plonk, Proof, ProvingKey, VerifyingKey

pytestmark = pytest.mark.gpu

TAU = 0x1234567
BLIND = [0x1000 + 17 * i for i in range(11)]


def _circuit(n, curve, inp=3):
    cv = pyref.curve_by_name(curve)
    gates, perm, pub, priv = PR.chain_gates(n, cv.r, inp=inp)
    pl = Plonkish.from_gates(gates["L"], gates["R"], gates["O"], gates["M"], gates["C"], perm, curve)
    return cv, gates, perm, pub, priv, pl


def _as_oracle_proof(proof, cv):
    from zksnake_amd.plonk.serialization import PROOF_POINTS, PROOF_SCALARS
    pts = [getattr(proof, k) for k in PROOF_POINTS]
    return {"points": [None if p.is_zero() else (p.x, p.y) for p in pts], "scalars": [getattr(proof, k) for k in PROOF_SCALARS]}


@pytest.mark.parametrize("curve", ["BN254", "BLS12_381"])
@pytest.mark.parametrize("n", [2, 4, 8, 64])
def test_proof_bytes_equal_oracle(gpu, curve, n):
    cv, gates, perm, pub, priv, pl = _circuit(n, curve)
    assert pl.is_sat(pub, priv)
    plonk = Plonk(pl, curve)
    plonk._tau, plonk._blinding = TAU, BLIND
    plonk.setup()
    proof = plonk.prove(pub, priv)

    opk, ovk = PR.setup(gates, perm, n, cv, TAU)
    oproof = PR.prove(opk, pub, priv, cv, BLIND)
    assert proof.to_bytes() == PR.proof_bytes(oproof, cv)
    # the key commitments agree with P(tau) * G1 as well
    for k in "LROMC":
        assert bytes(plonk.proving_key.tau_selector_poly[k].to_bytes()) == pyref.compress(cv, 1, opk["tau_Q"][k])

    assert plonk.verify(proof, pub)
    if n <= 8:
        assert PR.verify(ovk, _as_oracle_proof(proof, cv), pub, cv)  # independent verifier and pairing
    again = Proof.from_bytes(proof.to_bytes(), curve)
    assert again.to_bytes() == proof.to_bytes() and plonk.verify(again, pub)
    forged_pub = {k: (v + 1) % cv.r for k, v in pub.items()}
    assert not plonk.verify(proof, forged_pub)
    again.zeta_b = (again.zeta_b + 1) % cv.r
    assert not plonk.verify(again, pub)


@pytest.mark.parametrize("curve", ["BN254", "BLS12_381"])
def test_random_setup_and_key_round_trips(gpu, curve):
    cv, gates, perm, pub, priv, pl = _circuit(16, curve, inp=5)
    plonk = Plonk(pl, curve)
    plonk.setup()
    proof = plonk.prove(pub, priv)
    assert plonk.verify(proof, pub)
    assert proof.to_bytes() != plonk.prove(pub, priv).to_bytes()  # fresh blinding every time

    pk_bytes, vk_bytes = plonk.proving_key.to_bytes(), plonk.verifying_key.to_bytes()
    pk2, vk2 = ProvingKey.from_bytes(pk_bytes, curve), VerifyingKey.from_bytes(vk_bytes, curve)
    assert pk2.to_bytes() == pk_bytes and vk2.to_bytes() == vk_bytes and pk2.n == 16 and vk2.n == 16
    # the reference's views of the key
    assert pk2.selector_poly["M"](1) == sum(pk2.selector_poly["M"].coeffs()) % cv.r
    assert len(pk2.selector_eval["L"]) == 64 and len(pk2.lagrange_evals) == 64

    # a prover / verifier that only holds deserialized keys
    other = Plonk(pl, curve)
    other.proving_key, other.verifying_key = pk2, vk2
    other._blinding = BLIND
    plonk._blinding = BLIND
    p1, p2 = plonk.prove(pub, priv), other.prove(pub, priv)
    assert p1.to_bytes() == p2.to_bytes() and other.verify(p2, pub)


def test_short_witness_and_limb_array_input(gpu):
    """a witness shorter than the padded circuit (the reference pads a, b, c with zeros, protocol.py:167-169), given as
    ints or as a limb array: same proof; the device gather against numpy slicing"""
    import numpy as np
    from zksnake_amd import _native as N
    from zksnake_amd.frvec import DevVec, FrOps
    cv, gates, perm, pub, priv, _ = _circuit(16, "BN254")
    # drop the last gate row of the witness: its slots are a = out, b = c = 0, so only `a` changes the circuit
    g = {k: list(v) for k, v in gates.items()}
    g["L"][15] = 0                                       # row 15 becomes the empty gate ...
    pub = {}                                             # ... and nothing is public any more
    perm = list(perm)
    perm[15], perm[2 * 16 + 14] = 15, 2 * 16 + 14       # a_15 and c_14 are free slots now
    pl = Plonkish.from_gates(g["L"], g["R"], g["O"], g["M"], g["C"], perm, "BN254")
    short = priv[:3 * 15]
    assert pl.is_sat(pub, short + [0, 0, 0])
    plonk = Plonk(pl, "BN254")
    plonk._tau, plonk._blinding = TAU, BLIND
    plonk.setup()
    p1 = plonk.prove(pub, short)
    p2 = plonk.prove(pub, N.ints_to_limbs(short))
    p3 = plonk.prove(pub, short + [0, 0, 0])
    assert p1.to_bytes() == p2.to_bytes() == p3.to_bytes() and plonk.verify(p1, pub)
    opk, _ = PR.setup(g, perm, 16, cv, TAU)
    assert p1.to_bytes() == PR.proof_bytes(PR.prove(opk, pub, short, cv, BLIND), cv)

    V = FrOps(cv.r)
    flat = V.limbs(list(range(1, 3 * 7 + 2)))           # 22 entries: columns of 8, 7, 7
    d = V.d_from(flat)
    for j, cnt in ((0, 8), (1, 7), (2, 7)):
        col = DevVec(8)
        V.d_gather(cnt, d.ptr(), 3, j, col.ptr())
        assert (col.download(cnt) == flat[j::3]).all() and not col.download(8 - cnt, cnt).any()


def test_unsatisfied_witnesses_are_rejected(gpu):
    cv, gates, perm, pub, priv, pl = _circuit(8, "BN254")
    plonk = Plonk(pl, "BN254")
    plonk.setup()
    bad = list(priv)
    bad[3 * 2 + 2] = (bad[3 * 2 + 2] + 1) % cv.r      # c_2: breaks gate 2 and the copy c_2 = a_3
    assert not pl.is_sat(pub, bad)
    with pytest.raises(AssertionError):
        plonk.prove(pub, bad)
    bad = list(priv)
    bad[3 * 7 + 1] = 5                                  # b_7 is a free slot with zero selectors: still satisfiable
    assert pl.is_sat(pub, bad) and plonk.verify(plonk.prove(pub, bad), pub)
    wrong_pub = {k: (v + 1) % cv.r for k, v in pub.items()}
    with pytest.raises(AssertionError, match="gate constraints"):
        plonk.prove(wrong_pub, priv)
    with pytest.raises(AssertionError, match="ProvingKey"):
        Plonk(pl, "BN254").prove(pub, priv)


@pytest.mark.parametrize("curve", ["BN254", "BLS12_381"])
@pytest.mark.parametrize("n", [1, 5, 16, 1000, 40000])
def test_device_scans_against_host_helpers(gpu, curve, n):
    """zk_plonk_grand_product_dev / zk_poly_div_linear_dev / zk_poly_eval_dev / zk_vec_axpby_dev against the sequential
    host recurrences (zk_fr_*) and, for small n, Python integers"""
    import numpy as np
    from zksnake_amd import workloads as W
    from zksnake_amd.frvec import DevVec, FrOps
    cv = pyref.curve_by_name(curve)
    V = FrOps(cv.r)
    ints = W.field_stream(0xABC0 + n, 3 * n + 2, cv.r)[1]
    num, den, root, x = V.limbs(ints[:n]), V.limbs([v or 1 for v in ints[n:2 * n]]), ints[-1], ints[-2]
    d_num, d_den, out = V.d_from(num), V.d_from(den), DevVec(n + 1, zero=False)
    V.d_grand_product(n, d_num.ptr(), d_den.ptr(), out.ptr())
    assert (out.download() == V.grand_product(num, den)).all()
    q = DevVec(max(n - 1, 1))
    rem = V.d_div_linear(n, d_num.ptr(), root, q.ptr())
    hq, hrem = V.div_linear(num, root)
    assert rem == hrem == V.eval(num, root) == V.d_eval(n, d_num.ptr(), root)
    assert (q.download(n - 1) == hq).all()
    assert V.d_div_linear(n, d_num.ptr(), 0, q.ptr()) == ints[0] and (q.download(n - 1) == num[1:]).all()
    V.d_axpy(n, d_den.ptr(), x, d_num.ptr())
    exp = den.copy()
    V.scale_add(exp, num, x)
    assert (d_den.download() == exp).all()
    if n <= 16:
        acc, want = 1, [1]
        for a, b in zip(ints[:n], [v or 1 for v in ints[n:2 * n]]):
            acc = acc * a * pow(b, -1, cv.r) % cv.r
            want.append(acc)
        assert V.ints(out.download()) == want
        assert rem == sum(c * pow(root, k, cv.r) for k, c in enumerate(ints[:n])) % cv.r
    zero_den = den.copy()
    zero_den[n // 2] = 0
    with pytest.raises(Exception, match="zero denominator"):
        V.d_grand_product(n, d_num.ptr(), V.d_from(zero_den).ptr(), out.ptr())
