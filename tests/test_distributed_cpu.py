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


# ---- task x window partition of a proof (zksnake_amd/parallel.py: partition_proof) -------------------------------------
@pytest.mark.parametrize("curve_id,log_n", [(0, 20), (1, 23), (0, 10)])
def test_partition_covers_every_window_once(curve_id, log_n):
    sys.path.insert(0, ROOT)
    from zksnake_amd.parallel import PROOF_TASKS, partition_proof, window_partition
    nwin = {"k": 13, "u": 13, "v1": 13, "v2": 13, "h": 13} if log_n >= 20 else {"k": 16, "u": 16, "v1": 16, "v2": 16, "h": 16}
    one_gpu = partition_proof(1, nwin, curve_id, 1 << log_n)[1][0]
    for world in range(1, 18):
        assignment, projected = partition_proof(world, nwin, curve_id, 1 << log_n)
        assert len(assignment) == world == len(projected)
        for task in PROOF_TASKS:
            wins = [w for mine in assignment for t, (f, c) in mine.items() if t == task for w in range(f, f + c)]
            assert wins == list(range(nwin[task])), (world, task, wins)   # in rank order, contiguous, complete
        assert max(projected) <= one_gpu + 1e-9
        if world >= 2:
            assert max(projected) < one_gpu          # more ranks never leave the slowest rank where it was
        if world >= 8:
            # one MSM per rank (at most two pieces where the line is cut inside one), and the full QAP chain on few ranks
            assert all(len(mine) <= 2 for mine in assignment)
        ref = window_partition(world, nwin)
        # under its own cost model the task partition is never slower than "every MSM by window on every rank"
        from zksnake_amd.parallel import _segment_cost
        ref_cost = max(_segment_cost(curve_id, (1 << log_n) / float(1 << 20), {t: c for t, (f, c) in mine.items()}) for mine in ref if mine)
        assert max(projected) <= ref_cost + 1e-9, (world, max(projected), ref_cost)
        assert all(sorted(w for mine in ref for t, (f, c) in mine.items() if t == task for w in range(f, f + c)) == list(range(nwin[task]))
                   for task in PROOF_TASKS)
    # more ranks than (task, window) units: the surplus ranks hold nothing (they still take part in the proof's collective)
    assignment, projected = partition_proof(100, {t: 16 for t in PROOF_TASKS}, 0, 1 << 12)
    assert len(assignment) == 100 and sum(1 for mine in assignment if not mine) >= 20 and all(len(mine) <= 1 for mine in assignment)
    assert sorted(w for mine in assignment for t, (f, c) in mine.items() if t == "v2" for w in range(f, f + c)) == list(range(16))
    # a circuit without private wires has no <kdelta_1, w> MSM
    assignment, _ = partition_proof(4, {"k": 0, "u": 16, "v1": 16, "v2": 16, "h": 16}, 0, 1 << 12)
    assert all("k" not in mine for mine in assignment)


def test_eight_rank_partition_of_the_headline_prove():
    """BN254 at 2^20 on eight ranks: the witness MSM, <tau_1, u> and <tau_1, v> on a rank each, the G2 MSM over three ranks by
    window, <target_1, h> over two; the model's slowest rank well below the replicated-QAP layout of rounds 1-3"""
    sys.path.insert(0, ROOT)
    from zksnake_amd.parallel import partition_proof
    assignment, projected = partition_proof(8, {t: 13 for t in ("k", "u", "v1", "v2", "h")}, 0, 1 << 20)
    assert [sorted(m) for m in assignment[:3]] == [["k"], ["u"], ["v1"]]
    assert sum("v2" in m for m in assignment) == 3 and sum("h" in m for m in assignment) == 2
    assert max(projected) < 3.0   # ms; one GPU: 11.2 in the same model


def _flag_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from oracle import pyref
    from zksnake_amd import _native as N
    from zksnake_amd.arithmetization import R1CS
    from zksnake_amd.groth16 import Groth16
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        cv = pyref.BN254
        A, B, C, n_row, n_col, n_pub, w = pyref.readme_circuit(cv.r)
        unz = lambda m: tuple(list(t) for t in zip(*m))  # noqa: E731
        g = Groth16(R1CS.from_triplets(unz(A), unz(B), unz(C), n_row, n_col, n_pub, "BN254"), "BN254")
        g.shard_over_ranks(None)
        E = g.E
        from zksnake_amd._algebra import _points_to_limbs
        zero2 = np.zeros(N.point_limbs(0, 2), dtype=np.uint64)
        mine = _points_to_limbs([E.G1() * (rank + 2)], 0, 1)[0]
        # (1) all fine: the totals are the sums of the ranks' partial points, in the order they were given
        tot = g._exchange([(mine, 1), (zero2, 2), (mine, 1), (mine, 1), (mine, 1)])
        ok = tot[0] == E.G1() * sum(r + 2 for r in range(world)) and tot[1].is_zero()
        # (2) one rank reports a witness failure: every rank raises the reference's ValueError out of the collective
        witness_error = None
        try:
            if rank == world - 1:
                g._report_failure(g._FLAG_WITNESS)
                witness_error = "reported"
            else:
                g._exchange([(mine, 1), (zero2, 2), (mine, 1), (mine, 1), (mine, 1)])
        except ValueError as exc:
            witness_error = str(exc)
        # (3) any other failure: RuntimeError naming the rank
        other_error = None
        try:
            if rank == 0:
                g._report_failure(g._FLAG_ERROR)
                other_error = "reported"
            else:
                g._exchange([(mine, 1), (zero2, 2), (mine, 1), (mine, 1), (mine, 1)])
        except RuntimeError as exc:
            other_error = str(exc)
        return bool(ok), witness_error, other_error
    finally:
        dist.destroy_process_group()


def test_failure_flag_travels_in_the_proofs_collective():
    """round-3 advisor finding: a sharded rank that raised before the all_gather left the others blocked in it.  The partial points
    now travel with a flag word; host-side point arithmetic only, so this runs without a GPU"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import run_ranks
    results = run_ranks(_flag_worker, 2, (2, _free_port()), timeout=120)
    assert results[0] == (True, "Failed to evaluate with the given witness", "reported")
    assert results[1][0] is True and results[1][1] == "reported" and "rank(s) [0]" in results[1][2]


def test_run_ranks_fails_fast_when_a_child_dies():
    """the parent of a multi-process test notices a dead child within seconds, with the child's traceback (round 3: it sat in
    q.get(timeout=300) until pytest-timeout fired)"""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import run_ranks
    t0 = time.monotonic()
    with pytest.raises(AssertionError, match="rank 1 failed(.|\n)*ZeroDivisionError"):
        run_ranks(_dying_worker, 2, (), timeout=120)
    assert time.monotonic() - t0 < 60
    t0 = time.monotonic()
    with pytest.raises(AssertionError, match="exited"):
        run_ranks(_exiting_worker, 2, (), timeout=120)
    assert time.monotonic() - t0 < 60


def _dying_worker(rank):
    import time
    if rank == 1:
        return 1 // 0
    time.sleep(30)   # the healthy rank would have kept the old parent waiting
    return rank


def _exiting_worker(rank):
    import os as _os
    import time
    if rank == 0:
        _os._exit(3)   # no traceback, no result: only the exit code tells
    time.sleep(30)
    return rank


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no WORLD_SIZE: the shape of the driver's N = 1 command) must run TWO ranks --
    round 3 ran one and printed "n_gpus": 1.  --rendezvous-only stops after the ranks have formed their group and counted
    themselves, so the launcher is checked without a GPU; a mismatching WORLD_SIZE is refused."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-extra", "--rendezvous-only"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(x) for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_in_group"] == 2 and lines[0]["ranks_counted"] == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rendezvous-only"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr and "n_gpus" not in bad.stdout
