"""shared test helpers: seeded inputs and oracle-backed expectations (the oracle is only the checker)."""

import random

import numpy as np

from oracle import corc, pyref
from zksnake_amd import _native as N

CURVES = (("BN254", 0), ("BLS12_381", 1))


def rand_scalars(n, r, seed):
    rnd = random.Random(seed)
    vals = [rnd.randrange(r) for _ in range(n)]
    return vals, N.ints_to_limbs(vals, 4)


def rand_limbs(n, seed, top_bits=60):
    """(n,4) uint64 values below 2^(192+top_bits) -- cheap to generate at 2^20+ sizes"""
    rng = np.random.default_rng(seed)
    v = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << top_bits) - 1)
    return v


def generator_limbs(lib, cid, grp):
    out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
    N.check(lib.zk_point_generator(cid, grp, N.u64p(out)))
    return out


def oracle_bases(cid, grp, n, seed):
    """n points k_i * G from the C++ oracle (k_i seeded)"""
    cv = pyref.curve_by_name("BN254" if cid == 0 else "BLS12_381")
    ks, kl = rand_scalars(n, cv.r, seed)
    g = pyref.Group(cv, grp)
    pts = corc.batch_mul(cid, grp, kl, corc.points_to_limbs([g.gen], cid, grp)[0])
    return ks, pts
