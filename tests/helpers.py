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


# ---- multi-process tests: ranks as spawned children that report through a queue ---------------------------------------
def rank_entry(worker, rank, q, args):
    """child side: run worker(rank, *args) -> result; the result or the child's traceback goes back through the queue"""
    import traceback
    try:
        q.put(("ok", rank, worker(rank, *args)))
    except BaseException:  # noqa: BLE001 - reported to the parent, which fails the test with this text
        q.put(("error", rank, traceback.format_exc()))
        raise


def run_ranks(worker, world, args=(), timeout=300, poll=0.5):
    """spawn `world` children running worker(rank, *args) and return their results ordered by rank.  The parent polls the
    queue in short intervals and fails AS SOON AS a child has reported a traceback or has exited non-zero without a result
    (round 3: a child that died early left the parent in q.get(timeout=300) until pytest-timeout killed the run)."""
    import queue as _queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_entry, args=(worker, r, q, tuple(args))) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = {}, time.monotonic() + timeout
    try:
        while len(results) < world:
            try:
                kind, rank, payload = q.get(timeout=poll)
            except _queue.Empty:
                dead = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (None, 0) and r not in results]
                if dead:
                    # give a traceback that is already on its way a moment to arrive
                    try:
                        kind, rank, payload = q.get(timeout=2.0)
                    except _queue.Empty:
                        raise AssertionError(f"rank(s) {dead} exited (rank, code) without a result") from None
                elif time.monotonic() > deadline:
                    raise AssertionError(f"ranks {sorted(set(range(world)) - set(results))} did not report within {timeout} s") from None
                else:
                    continue
            if kind == "error":
                raise AssertionError(f"rank {rank} failed:\n{payload}")
            results[rank] = payload
        for p in procs:
            p.join(60)
            assert p.exitcode == 0, f"a rank exited with code {p.exitcode}"
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
                p.join(10)
    return [results[r] for r in range(world)]
