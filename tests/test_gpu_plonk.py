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
@pytest.mark.parametrize("n", [4, 8, 64])
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
