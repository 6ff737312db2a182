#!/usr/bin/env python3
"""
Generates tests/golden/groth16_vectors.json: the Groth16 proof of the benchmark chain circuit
(benchmarks/benchmark_groth16.py:7-27 shape, zksnake_amd/workloads.chain_circuit) at full BASELINE sizes, computed by
the definitional oracle as discrete logarithms -- no FFT, no MSM: oracle/pyref.groth16_closed_form with
the toxic waste and the blinding pinned to the SplitMix64 stream of seed 0x5EED0004.

    python tests/golden/gen_groth16_golden.py [CURVE:LOG_N ...]        (default: BN254:20 BLS12_381:20 BLS12_381:22)

Pure Python big integers, a few minutes for the large sizes.  The stored vector is inputs' description + the 128/192
proof bytes and their sha256; tests/test_gpu_groth16.py and bench.py compare the GPU prover's bytes with it.
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyref  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402

OUT = os.path.join(HERE, "groth16_vectors.json")


def closed_form_chain(n, cv, toxic, rs):
    """pyref.groth16_closed_form specialised to the chain circuit's three one-entry-per-row matrices (same formula:
    the generic function spends its time building Python tuples for 3 * 2^20 triplets)."""
    r = cv.r
    A, B, C, w, n_col = W.chain_circuit(n, r)
    tau, alpha, beta, gamma, delta = toxic
    rr, ss = rs
    lag = pyref.lagrange_at(n, tau, cv)
    (ar, ac, av), (br, bc, bv), (cr, cc, cvv) = A, B, C
    assert av == bv == cvv == [1] * n and ar == br == cr == list(range(n))
    U = sum(lag[i] * w[ac[i]] for i in range(n)) % r
    V = sum(lag[i] * w[bc[i]] for i in range(n)) % r
    Wv = sum(lag[i] * w[cc[i]] for i in range(n)) % r
    L, R, O = [0] * n_col, [0] * n_col, [0] * n_col
    for i in range(n):
        L[ac[i]] += lag[i]
        R[bc[i]] += lag[i]
        O[cc[i]] += lag[i]
    priv = sum(w[j] * ((L[j] * beta + R[j] * alpha + O[j]) % r) for j in range(2, n_col)) % r
    a = (alpha + U + rr * delta) % r
    b = (beta + V + ss * delta) % r
    c = ((U * V - Wv) + priv) % r * pow(delta, -1, r) % r
    c = (c + ss * a + rr * b - rr * ss % r * delta) % r
    return a, b, c


def main():
    jobs = sys.argv[1:] or ["BN254:20", "BLS12_381:20", "BLS12_381:22"]
    out = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            out = json.load(f)
    for job in jobs:
        curve, log_n = job.split(":")
        log_n = int(log_n)
        cv = pyref.curve_by_name(curve)
        n = 1 << log_n
        toxic = tuple(W.field_stream(W.SEED_PROVE, 5, cv.r)[1])
        blind = tuple(W.field_stream(W.SEED_PROVE, 2, cv.r, offset=5)[1])
        t0 = time.time()
        if log_n <= 8:  # cross-check of the specialised formula against the generic one
            A, B, C, w, n_col = W.chain_circuit(n, cv.r)
            trip = (list(zip(*A)), list(zip(*B)), list(zip(*C)), n, n_col, 2, w)
            assert closed_form_chain(n, cv, toxic, blind) == pyref.groth16_closed_form(*trip, cv, toxic, blind)
        a, b, c = closed_form_chain(n, cv, toxic, blind)
        g1, g2 = pyref.G1(cv), pyref.G2(cv)
        pb = pyref.proof_bytes(cv, (g1.mul(g1.gen, a), g2.mul(g2.gen, b), g1.mul(g1.gen, c)))
        out.setdefault(curve, {})[str(log_n)] = {
            "circuit": "workloads.chain_circuit(2^%d, r, inp=2): n_public = 2" % log_n,
            "toxic_and_blinding": "workloads.field_stream(SEED_PROVE = 0x5EED0004, 7, r): tau, alpha, beta, gamma, delta, r, s",
            "proof_hex": pb.hex(), "sha256": hashlib.sha256(pb).hexdigest(),
        }
        print(job, "done in %.1f s" % (time.time() - t0), out[curve][str(log_n)]["sha256"], flush=True)
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
