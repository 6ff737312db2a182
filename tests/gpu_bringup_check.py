#!/usr/bin/env python3
"""Quick GPU bring-up check (a script, not collected by pytest): libzkmi vs the oracle on small/medium sizes, with
timings.  Lives under tests/ because it uses the oracle as its checker.  python tests/gpu_bringup_check.py"""
import sys, os, time, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from zksnake_amd import _native as N
from oracle import pyref as R, corc as C

lib = N.ensure_gpu()
print("devices:", lib.zk_device_count())
rng = np.random.default_rng(7)
random.seed(7)

def rand_fr(n, r):
    v = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    # reduce below r via python for small n, else clear top bits
    v[:, 3] &= np.uint64((1 << 61) - 1)
    return v

ok = True
for name, cid in (("BN254", 0), ("BLS12_381", 1)):
    cv = R.curve_by_name(name)
    for log_n in (0, 1, 3, 9, 10, 11, 14, 18):
        n = 1 << log_n
        a = rand_fr(n, cv.r)
        for inv in (0, 1):
            out = np.zeros_like(a)
            t = time.time()
            N.check(lib.zk_ntt(cid, inv, 0, n, N.u64p(a), n, N.u64p(out)))
            dt = time.time() - t
            exp = C.ntt(cid, a, inverse=bool(inv))
            good = (out == exp).all()
            ok &= bool(good)
            print(f"{name} ntt log_n={log_n} inv={inv} {'OK' if good else 'MISMATCH'} {dt*1e3:.1f} ms")
    n = 1000
    a = rand_fr(n, cv.r); b = rand_fr(n, cv.r)
    for op, nm in ((0, "mul"), (1, "add"), (2, "sub")):
        out = np.zeros_like(a)
        N.check(lib.zk_vec_op(cid, op, n, n, N.u64p(a), n, N.u64p(b), N.u64p(out)))
        good = (out == C.vec_op(cid, nm, a, b)).all(); ok &= bool(good)
        print(name, "vec", nm, "OK" if good else "MISMATCH")
    for grp in (1, 2):
        W = N.point_limbs(cid, grp)
        gen = np.zeros(W, dtype=np.uint64); N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
        for n in (1, 5, 300, 5000) + ((1 << 16,) if grp == 1 else ()):
            ks = rand_fr(n, cv.r)
            t = time.time()
            bases = np.zeros((n, W), dtype=np.uint64)
            N.check(lib.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
            dt_b = time.time() - t
            if n <= 5000:
                exp_b = C.batch_mul(cid, grp, ks, gen)
                good = (bases == exp_b).all(); ok &= bool(good)
                print(f"{name} G{grp} batch_mul n={n} {'OK' if good else 'MISMATCH'} {dt_b*1e3:.1f} ms")
            sc = rand_fr(n, cv.r)
            if n >= 5:
                sc[0] = 0; sc[1] = 0; sc[1, 0] = 1
                sc[2] = N.ints_to_limbs([cv.r - 1])[0]
                bases[4] = bases[3]
            out = np.zeros(W, dtype=np.uint64)
            t = time.time()
            N.check(lib.zk_msm(cid, grp, n, n, N.u64p(sc), N.u64p(bases), N.u64p(out)))
            dt = time.time() - t
            exp = C.msm(cid, grp, sc, bases, threads=8)
            good = (out == exp).all(); ok &= bool(good)
            print(f"{name} G{grp} msm n={n} {'OK' if good else 'MISMATCH'} {dt*1e3:.1f} ms (incl. plan+upload)")
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
