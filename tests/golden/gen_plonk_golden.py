#!/usr/bin/env python3
"""
Generates tests/golden/plonk_vectors.json from the PlonK oracle (oracle/plonk_ref.py, the coefficient-form
restatement of the reference prover):  python tests/golden/gen_plonk_golden.py

Per curve: the chain circuit with n = 4 and n = 8 gates, pinned tau and blinding scalars, the oracle's proof bytes,
the key commitments and the Fiat-Shamir challenges.  These pin the oracle and the product against silent drift; they
are NOT outputs of the reference binary (parity unpinned).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import plonk_ref as PR  # noqa: E402
from oracle import pyref as R  # noqa: E402

TAU = 0x1234567
BLIND = [0x1000 + 17 * i for i in range(11)]


def main():
    out = {"tau": TAU, "blinding": BLIND, "inp": 3}
    for name in ("BN254", "BLS12_381"):
        cv = R.curve_by_name(name)
        per = {}
        for n in (4, 8):
            gates, perm, pub, priv = PR.chain_gates(n, cv.r, inp=3)
            pk, vk = PR.setup(gates, perm, n, cv, TAU)
            proof = PR.prove(pk, pub, priv, cv, BLIND)
            assert PR.verify(vk, proof, pub, cv)
            per[str(n)] = {
                "public": {str(k): str(v) for k, v in pub.items()},
                "private": [str(v) for v in priv],
                "permutation": perm,
                "selector_commitments": {k: R.compress(cv, 1, pk["tau_Q"][k]).hex() for k in "LROMC"},
                "sigma_commitments": [R.compress(cv, 1, P).hex() for P in pk["tau_S"]],
                "proof_bytes": PR.proof_bytes(proof, cv).hex(),
            }
        out[name] = per
    # a size at which the commitments' MSMs (4096 .. 12288 points on fixed-base plans) go through the two-level sort:
    # proof bytes only (the inputs are chain_gates(4096, r, inp=3)); ~3 minutes of schoolbook products in Python
    path = os.path.join(HERE, "plonk_vectors.json")
    big = {}
    if os.path.exists(path):
        with open(path) as f:
            big = json.load(f).get("big", {})
    if "big" in sys.argv[1:] or not big:
        import hashlib
        cv = R.curve_by_name("BN254")
        n = 4096
        gates, perm, pub, priv = PR.chain_gates(n, cv.r, inp=3)
        pk, vk = PR.setup(gates, perm, n, cv, TAU)
        pb = PR.proof_bytes(PR.prove(pk, pub, priv, cv, BLIND), cv)
        big = {"BN254": {str(n): {"proof_bytes": pb.hex(), "sha256": hashlib.sha256(pb).hexdigest()}}}
    out["big"] = big
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote plonk_vectors.json")


if __name__ == "__main__":
    main()
