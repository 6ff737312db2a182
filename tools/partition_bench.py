#!/usr/bin/env python3
"""Single-GPU projection of the sharded Groth16 prove: every rank of a `world`-rank partition is SIMULATED on the one GPU,
one after the other -- its share of prove() runs up to the proof's collective and is timed there (the collective itself is
replaced by a recorder, as in tests/test_gpu_groth16.py::test_simulated_ranks_of_the_task_partition; a second pass with the
gathered rows checks that every rank assembles the same, verifying proof).  The slowest rank's time is what an N-GPU run costs
before communication.  Prints one JSON object.

usage: partition_bench.py [--curve BN254] [--log-n 20] [--world 8] [--partition task|window|both] [--reps 5]"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np  # noqa: E402

from zksnake_amd import _native as N  # noqa: E402
from zksnake_amd import parallel  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.arithmetization import R1CS  # noqa: E402
from zksnake_amd.groth16 import Groth16  # noqa: E402


class _Stop(Exception):
    pass


def run(curve, log_n, world, partition, reps):
    import gc
    n = 1 << log_n
    r = W.scalar_field(curve)
    A, B, C, w, n_col = W.chain_circuit(n, r)
    toxic = tuple(W.field_stream(W.SEED_PROVE, 5, r)[1])
    blinding = tuple(W.field_stream(W.SEED_PROVE, 2, r, offset=5)[1])
    pub, prv = N.ints_to_limbs(w[:2]), N.ints_to_limbs(w[2:])
    rows, state = {}, {"rank": 0, "gathered": None, "t_stop": 0.0}
    real = parallel.all_gather_limbs

    def fake(mine, device=None):
        if state["gathered"] is None:
            state["t_stop"] = time.perf_counter()
            rows[state["rank"]] = np.array(mine, dtype=np.uint64, copy=True)
            raise _Stop()
        return state["gathered"]

    parallel.all_gather_limbs = fake
    per_rank, proofs = [], []
    r1cs = R1CS.from_triplets(A, B, C, n, n_col, 2, curve)
    try:
        provers = []
        for rank in range(world):
            g = Groth16(r1cs, curve)
            g._toxic, g._blinding = toxic, blinding
            g._shard, g._partition = (rank, world, None), partition
            t0 = time.perf_counter()
            g.setup()
            provers.append((g, time.perf_counter() - t0))
        gc.collect()
        gc.freeze()
        lib = N.load()
        for rank, (g, setup_s) in enumerate(provers):
            state["rank"] = rank
            times, qaps = [], []
            for _ in range(reps + 1):
                lib.zk_dev_synchronize()
                t0 = time.perf_counter()
                try:
                    g.prove(pub, prv)
                except _Stop:
                    pass
                times.append((state["t_stop"] - t0) * 1e3)
                qaps.append(g.last_timings.get("qap_ms", 0.0))
            mine = g._my_tasks()
            per_rank.append({"rank": rank, "tasks": {t: list(v) for t, v in mine.items()}, "qap_outputs": sorted(g._qap_needs()),
                             "ms_to_collective": round(statistics.median(times[1:]), 3), "ms_min": round(min(times[1:]), 3),
                             "upload_and_qap_ms": round(statistics.median(qaps[1:]), 3),
                             "msm_phase_ms": round(statistics.median(times[1:]) - statistics.median(qaps[1:]), 3),
                             "projected_ms": g.projected_ms[rank] if g.projected_ms else None, "setup_s": round(setup_s, 2)})
        state["gathered"] = np.stack([rows[k] for k in range(world)])
        for rank, (g, _) in enumerate(provers):
            state["rank"] = rank
            proofs.append(g.prove(pub, prv))
        same = len({p.to_bytes() for p in proofs}) == 1
        verifies = bool(provers[0][0].verify(proofs[0], w[:2]))
    finally:
        parallel.all_gather_limbs = real
        gc.unfreeze()
    out = {"curve": curve, "log_n": log_n, "world": world, "partition": partition,
           "slowest_rank_ms": max(p["ms_to_collective"] for p in per_rank),
           "slowest_msm_phase_ms": max(p["msm_phase_ms"] for p in per_rank),
           "sum_over_ranks_ms": round(sum(p["ms_to_collective"] for p in per_rank), 3),
           "all_ranks_same_proof": same, "verifies": verifies, "per_rank": per_rank}
    gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "groth16_vectors.json")
    if os.path.exists(gpath):
        with open(gpath) as f:
            gold = json.load(f).get(curve, {}).get(str(log_n))
        if gold is not None:
            out["matches_committed_closed_form"] = proofs[0].to_bytes().hex() == gold["proof_hex"]
    del provers
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="BN254")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--partition", default="both")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    N.ensure_gpu()
    res = {}
    for part in (("task", "window") if args.partition == "both" else (args.partition,)):
        res[part] = run(args.curve, args.log_n, args.world, part, args.reps)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
