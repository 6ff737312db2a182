"""CPU: the PlonK oracle is self-consistent (prover + verifier + pairing), the transcript's byte encodings are the
reference's (transcript.py:40-71), Plonkish.is_sat follows plonkish.py:94-123."""

import hashlib

import pytest

from oracle import plonk_ref as PR
from oracle import pyref


def test_oracle_plonk_proves_and_verifies():
    cv = pyref.BN254
    gates, perm, pub, priv = PR.chain_gates(4, cv.r, inp=3)
    pk, vk = PR.setup(gates, perm, 4, cv, tau=0x1234567)
    proof = PR.prove(pk, pub, priv, cv, [0x1000 + i for i in range(11)])
    assert len(PR.proof_bytes(proof, cv)) == 9 * 32 + 6 * 32
    assert PR.verify(vk, proof, pub, cv)
    forged = {"points": proof["points"], "scalars": [(proof["scalars"][0] + 1) % cv.r] + proof["scalars"][1:]}
    assert not PR.verify(vk, forged, pub, cv)
    assert not PR.verify(vk, proof, {k: (v + 1) % cv.r for k, v in pub.items()}, cv)
    with pytest.raises(AssertionError, match="Copy constraints"):
        bad = list(priv)
        bad[1] = (bad[1] + 1) % cv.r  # b_0 no longer equals the other copies of `inp`
        PR.prove(pk, pub, bad, cv, [1] * 11)


def test_golden_plonk_vectors():
    """the committed oracle proofs (tests/golden/plonk_vectors.json) are reproduced"""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "plonk_vectors.json")) as f:
        gold = json.load(f)
    cv = pyref.BN254
    g = gold["BN254"]["4"]
    gates, perm, pub, priv = PR.chain_gates(4, cv.r, inp=gold["inp"])
    assert perm == g["permutation"] and [str(v) for v in priv] == g["private"]
    pk, _ = PR.setup(gates, perm, 4, cv, gold["tau"])
    assert {k: pyref.compress(cv, 1, pk["tau_Q"][k]).hex() for k in "LROMC"} == g["selector_commitments"]
    assert PR.proof_bytes(PR.prove(pk, pub, priv, cv, gold["blinding"]), cv).hex() == g["proof_bytes"]


def test_transcript_encodings():
    from zksnake_amd.ecc import EllipticCurve
    from zksnake_amd.transcript import FiatShamirTranscript
    E = EllipticCurve("BN254")
    P = E.G1() * 7
    t = FiatShamirTranscript(field=E.order)
    t.append(P)
    t.append(0x1234)
    t.append([5, 0])
    t.append(b"xy")
    t.append("z")
    h = hashlib.blake2b(b"")
    h.update(bytes(P.to_bytes()))
    h.update((0x1234).to_bytes(13, "big"))       # bit_length() BYTES: the reference's quirk
    h.update((5).to_bytes(3, "big") + b"" + b"xy" + b"z")
    d1 = h.digest()
    assert t.get_challenge_scalar() == int.from_bytes(d1, "big") % E.order
    t.append(1)
    h2 = hashlib.blake2b(d1)
    h2.update(b"\x01")
    assert t.get_challenge() == h2.digest()
    with pytest.raises(TypeError):
        t.append(1.5)
    # same bytes as the oracle's transcript
    o = PR.Transcript(pyref.BN254)
    o.point(pyref.G1(pyref.BN254).mul(pyref.BN254.g1, 7))
    o.scalar(0x1234)
    t2 = FiatShamirTranscript(field=E.order)
    t2.append(P)
    t2.append(0x1234)
    assert o.challenge() == t2.get_challenge_scalar()


def test_plonkish_container():
    from zksnake_amd.arithmetization import Plonkish
    r = pyref.BN254.r
    gates, perm, pub, priv = PR.chain_gates(8, r, inp=3)
    pl = Plonkish.from_gates(gates["L"][:7] + [1], gates["R"], gates["O"], gates["M"], gates["C"], perm)
    assert pl.length == 8 and pl.unpadded_length == 8 and pl.p == r
    assert pl.is_sat(pub, priv)
    bad = list(priv)
    bad[4] = (bad[4] + 1) % r
    assert not pl.is_sat(pub, bad)
    assert not pl.is_sat({7: 1}, priv)
    short = Plonkish.from_gates([1, 1, 1], [0] * 3, [0] * 3, [0] * 3, [0] * 3, list(range(12)))
    assert short.length == 4 and short.unpadded_length == 3 and short.qL == [1, 1, 1, 0]
    with pytest.raises(ValueError):
        Plonkish.from_gates([1], [0], [0], [0], [0], [0, 0, 1])
    with pytest.raises(NotImplementedError):
        pl.compile()
