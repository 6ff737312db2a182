"""where the reference call shape prove(list[int], list[int]) spends its extra time over the limb-array call (GPU box)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from zksnake_amd import _native as N, workloads as W
from zksnake_amd.arithmetization import R1CS
from zksnake_amd.groth16 import Groth16
from zksnake_amd.device import PinnedArray

n = 1 << 20
r = W.scalar_field("BN254")
A, B, C, w, n_col = W.chain_circuit(n, r)
g = Groth16(R1CS.from_triplets(A, B, C, n, n_col, 2, "BN254"), "BN254")
g._toxic = tuple(W.field_stream(W.SEED_PROVE, 5, r)[1]); g._blinding = tuple(W.field_stream(W.SEED_PROVE, 2, r, offset=5)[1])
g.setup()
pin = PinnedArray((n_col, 4))
pageable = np.empty((n_col, 4), dtype=np.uint64)
for label, out in (("pinned", pin.array), ("pageable", pageable)):
    ts = []
    for _ in range(6):
        t = time.perf_counter(); N.ints_to_limbs(w, 4, r, out=out); ts.append((time.perf_counter() - t) * 1e3)
    print("ints_to_limbs ->", label, [round(x, 2) for x in ts], "threads env", os.environ.get("ZKMI_PACK_THREADS"))
import gc; gc.collect(); gc.freeze()
pub, prv = N.ints_to_limbs(w[:2]), N.ints_to_limbs(w[2:])
pl, vl = w[:2], w[2:]
for label, a, b in (("limbs", pub, prv), ("lists", pl, vl)):
    ts = []
    for _ in range(8):
        N.load().zk_dev_synchronize()
        t = time.perf_counter(); g.prove(a, b); ts.append((time.perf_counter() - t) * 1e3)
    print("prove", label, "median", round(statistics.median(ts[1:]), 2), [round(x, 2) for x in ts], {k: round(v, 2) for k, v in g.last_timings.items()})
