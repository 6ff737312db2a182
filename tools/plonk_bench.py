#!/usr/bin/env python3
"""PlonK prove timing on the chain circuit.

  python tools/plonk_bench.py --log-n 14 [--curve BN254] [--reps 3]

Setup is untimed; the timed region is Plonk.prove() with the witness as a host limb array.  Every proof is
checked with Plonk.verify()."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from zksnake_amd import _native as N  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.arithmetization import Plonkish  # noqa: E402
from zksnake_amd.plonk import Plonk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=14)
    ap.add_argument("--curve", default="BN254")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n = 1 << args.log_n
    r = W.scalar_field(args.curve)
    t0 = time.time()
    gates, perm, pub, priv = W.plonk_chain_gates(n, r)
    pl = Plonkish.from_gates(gates["L"], gates["R"], gates["O"], gates["M"], gates["C"], perm, args.curve)
    plonk = Plonk(pl, args.curve)
    plonk._tau = W.field_stream(W.SEED_PROVE, 1, r)[1][0]
    t1 = time.time()
    plonk.setup()
    t2 = time.time()
    witness = N.ints_to_limbs(priv)
    times = []
    for _ in range(args.reps + 1):
        N.load().zk_dev_synchronize()
        s = time.perf_counter()
        proof = plonk.prove(pub, witness)
        times.append(time.perf_counter() - s)
    ok = plonk.verify(proof, pub)
    print(json.dumps({"curve": args.curve, "log_n": args.log_n, "build_s": round(t1 - t0, 2), "setup_s": round(t2 - t1, 2),
                      "prove_first_ms": round(times[0] * 1e3, 2), "prove_ms": round(min(times[1:]) * 1e3, 2),
                      "prove_ms_all": [round(t * 1e3, 2) for t in times[1:]], "verifies": bool(ok)}))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
