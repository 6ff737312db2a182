#!/usr/bin/env python3
"""Groth16 prove timing on the benchmark chain circuit (BASELINE config 4 / 5 shape).

  python tools/prove_bench.py --log-n 20 [--curve BN254] [--reps 3]

Setup (key generation) is untimed; the timed region is Groth16.prove() with the witness given as limb
arrays on the host, matching benchmarks/benchmark_groth16.py:43-46 of the reference ("Prove time").
The proof bytes are printed (`proof_hex`); tests/test_gpu_groth16.py compares them with the committed closed-form proofs."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from zksnake_amd import _native as N  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.arithmetization import R1CS  # noqa: E402
from zksnake_amd.groth16 import Groth16  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="BN254")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n = 1 << args.log_n
    r = W.scalar_field(args.curve)
    t0 = time.time()
    A, B, C, w, n_col = W.chain_circuit(n, r)
    r1cs = R1CS.from_triplets(A, B, C, n, n_col, 2, args.curve)
    g = Groth16(r1cs, args.curve)
    toxic = tuple(x for x in W.field_stream(W.SEED_PROVE, 5, r)[1])
    blind = tuple(x for x in W.field_stream(W.SEED_PROVE, 2, r, offset=5)[1])
    g._toxic, g._blinding = toxic, blind
    t1 = time.time()
    g.setup()
    t2 = time.time()
    pub, prv = N.ints_to_limbs(w[:2]), N.ints_to_limbs(w[2:])
    times = []
    proof = None
    for _ in range(args.reps + 1):
        N.load().zk_dev_synchronize()
        s = time.perf_counter()
        proof = g.prove(pub, prv)
        times.append(time.perf_counter() - s)
    out = {"curve": args.curve, "log_n": args.log_n, "build_s": round(t1 - t0, 2), "setup_s": round(t2 - t1, 2),
           "prove_first_ms": round(times[0] * 1e3, 2), "prove_ms": round(min(times[1:]) * 1e3, 2),
           "prove_ms_all": [round(t * 1e3, 2) for t in times[1:]], "timeline_ms": {k: round(v, 3) for k, v in g.last_timings.items()}, "verifies": bool(g.verify(proof, w[:2])),
           "proof_hex": proof.to_bytes().hex()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
