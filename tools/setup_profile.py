#!/usr/bin/env python3
"""where Groth16.setup and the first proof spend their time:  python tools/setup_profile.py [LOG_N] [CURVE]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zksnake_amd import _native as N, workloads as W  # noqa: E402
from zksnake_amd.arithmetization import R1CS  # noqa: E402
from zksnake_amd.groth16 import Groth16  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
curve = sys.argv[2] if len(sys.argv) > 2 else "BN254"
n = 1 << log_n
r = W.scalar_field(curve)
A, B, C, w, n_col = W.chain_circuit(n, r)
g = Groth16(R1CS.from_triplets(A, B, C, n, n_col, 2, curve), curve)
g._toxic = tuple(W.field_stream(W.SEED_PROVE, 5, r)[1])
g._blinding = tuple(W.field_stream(W.SEED_PROVE, 2, r, offset=5)[1])
N.ensure_gpu()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
g.setup()
pr.disable()
print(f"setup {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pub, prv = N.ints_to_limbs(w[:2]), N.ints_to_limbs(w[2:])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
g.prove(pub, prv)
pr.disable()
print(f"first prove {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
t0 = time.perf_counter()
g.prove(pub, prv)
print(f"second prove {time.perf_counter() - t0:.4f} s", g.last_timings)
