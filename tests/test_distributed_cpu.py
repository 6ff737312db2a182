"""CPU, world_size 2 over gloo: the multi-GPU path of the MSM (window sharding -> one partial point per rank ->
all_gather -> host point additions).  The per-rank partial here comes from the definition (oracle digits), the
exchange + combination code is the product's (zksnake_amd/parallel.py), exactly what bench.py runs over RCCL."""

import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _signed_digits(s, c, nwin):
    """the product's digit rule: s + sum 2^(c-1) 2^(wc), then c-bit fields minus 2^(c-1)"""
    biased = s + sum(1 << (w * c + c - 1) for w in range(nwin))
    return [((biased >> (w * c)) & ((1 << c) - 1)) - (1 << (c - 1)) for w in range(nwin)]


def _worker(rank, world, port, c, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import corc, pyref as R
    from zksnake_amd import _native as N
    from zksnake_amd.parallel import all_gather_sum, window_ranges

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        cv, cid, grp = R.BN254, 0, 1
        g = R.G1(cv)
        import random
        rnd = random.Random(99)
        n = 12
        pts = [g.mul(g.gen, rnd.randrange(cv.r)) for _ in range(n)]
        sc = [rnd.randrange(cv.r) for _ in range(n)]
        sc[0], sc[1] = 0, cv.r - 1
        nwin = (254 + 1 + c - 1) // c
        first, count = window_ranges(nwin, world)[rank]
        partial = None
        for P, s in zip(pts, sc):
            d = _signed_digits(s, c, nwin)
            assert sum(x << (w * c) for w, x in enumerate(d)) == s
            k = sum(d[w] << (w * c) for w in range(first, first + count))
            partial = g.add(partial, R.ec_mul(g.F, P, k))
        mine = corc.points_to_limbs([partial], cid, grp)[0]
        total = all_gather_sum(cid, grp, mine)
        exp = corc.points_to_limbs([g.msm(pts, sc)], cid, grp)[0]
        q.put((rank, bool((total == exp).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("c", [16, 13])
def test_window_sharded_msm_over_gloo(c):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, c, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, True), (1, True)]
