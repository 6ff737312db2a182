#!/usr/bin/env python3
"""BN254 / BLS12-381 Fr NTT timing on resident data:  python tools/ntt_bench.py [--log-n 22] [--curve BN254] [--reps 20]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np  # noqa: E402

from zksnake_amd import _native as N  # noqa: E402
from zksnake_amd import workloads as W  # noqa: E402
from zksnake_amd.device import DeviceBuffer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--curve", default="BN254")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    lib = N.ensure_gpu()
    cid = N.curve_id(args.curve)
    m = 1 << args.log_n
    limbs = W.splitmix64(W.SEED_NTT, 4 * m).reshape(m, 4)
    limbs[:, 3] &= np.uint64((1 << 60) - 1)
    d = DeviceBuffer.from_numpy(limbs)
    N.check(lib.zk_ntt_dev(cid, 0, args.log_n, d.ptr, None))
    N.check(lib.zk_ntt_dev(cid, 1, args.log_n, d.ptr, None))
    assert (d.download((m, 4)) == limbs).all(), "iNTT(NTT(x)) != x"
    lib.zk_dev_synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        N.check(lib.zk_ntt_dev(cid, 0, args.log_n, d.ptr, None))
    lib.zk_dev_synchronize()
    ms = (time.perf_counter() - t0) / args.reps * 1e3
    print(json.dumps({"curve": args.curve, "log_n": args.log_n, "ms": round(ms, 4), "Melem/s": round(m / ms / 1e3, 1),
                      "algorithmic_GB/s": round(64 * m / ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
