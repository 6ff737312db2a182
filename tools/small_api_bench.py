#!/usr/bin/env python3
"""latency of the reference-shaped API on SMALL inputs (where fixed overheads, not throughput, decide):
  python tools/small_api_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zksnake_amd import _native as N, workloads as W  # noqa: E402
from zksnake_amd.arithmetization import R1CS  # noqa: E402
from zksnake_amd.ecc import EllipticCurve  # noqa: E402
from zksnake_amd.groth16 import Groth16  # noqa: E402
from zksnake_amd.polynomial import fft, ifft  # noqa: E402


def timed(label, fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"{label:46s} {min(ts):8.3f} ms (best of {reps})")


N.ensure_gpu()
E = EllipticCurve("BN254")
r = E.order
G = E.G1()
pts = [G * (i + 1) for i in range(64)]
sc = list(range(3, 67))
timed("PointG1 + PointG1", lambda: pts[3] + pts[5])
timed("PointG1 * int", lambda: pts[3] * 0x1234567890ABCDEF1234567890ABCDEF)
timed("multiexp(64 PointG1, 64 ints)", lambda: E.multiexp(pts, sc))
timed("batch_mul(G1, 64 ints)", lambda: E.batch_mul(G, sc))
vals = list(range(1, 1025))
timed("fft(1024 ints)", lambda: fft(vals, r))
timed("ifft(fft(1024 ints))", lambda: ifft(fft(vals, r), r))
for n in (8, 256, 4096):
    A, B, C, w, n_col = W.chain_circuit(n, r)
    g = Groth16(R1CS.from_triplets(A, B, C, n, n_col, 2, "BN254"), "BN254")
    t = time.perf_counter(); g.setup(); ts = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); proof = g.prove(w[:2], w[2:]); t1 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); proof = g.prove(w[:2], w[2:]); t2 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); ok = g.verify(proof, w[:2]); tv = (time.perf_counter() - t) * 1e3
    print(f"Groth16 chain circuit n={n:5d}: setup {ts:7.1f} ms, first prove {t1:6.2f} ms, prove {t2:6.2f} ms, verify {tv:6.1f} ms ({ok})")
