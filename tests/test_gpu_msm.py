"""GPU parity: Pippenger MSM and batch scalar multiplication through the C ABI, all four groups,
against the CPU oracle (bit-exact affine results), plus the 2^20 closed-form check of BASELINE config 2."""

import numpy as np
import pytest

from helpers import CURVES, generator_limbs, oracle_bases, rand_limbs, rand_scalars
from oracle import corc, pyref
from zksnake_amd import _native as N
from zksnake_amd import workloads as W

pytestmark = pytest.mark.gpu


def _msm(lib, cid, grp, sc, bases):
    out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
    N.check(lib.zk_msm(cid, grp, bases.shape[0], sc.shape[0], N.u64p(sc), N.u64p(bases), N.u64p(out)))
    return out


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
@pytest.mark.parametrize("n", [1, 2, 7, 33, 1000])
def test_msm_matches_oracle(gpu, name, cid, grp, n):
    cv = pyref.curve_by_name(name)
    _, bases = oracle_bases(cid, grp, n, 40 + n)
    _, sc = rand_scalars(n, cv.r, 50 + n)
    assert (_msm(gpu, cid, grp, sc, bases) == corc.msm(cid, grp, sc, bases, threads=8)).all()


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_msm_edge_cases(gpu, name, cid, grp):
    """zero / one / r-1 scalars, scalars >= r, duplicate bases, P and -P, the point at infinity as a base"""
    cv = pyref.curve_by_name(name)
    g = pyref.Group(cv, grp)
    n = 24
    ks, bases = oracle_bases(cid, grp, n, 60)
    pts = corc.limbs_to_points(bases, cid, grp)
    vals, _ = rand_scalars(n, cv.r, 61)
    vals[0], vals[1], vals[2], vals[3] = 0, 1, cv.r - 1, cv.r + 7     # r + 7 must act like 7
    pts[5] = pts[4]                                                   # duplicate base
    pts[7] = g.neg(pts[6]); vals[7] = vals[6]                         # cancels exactly
    pts[8] = None                                                     # infinity base
    pts[10] = pts[9]; vals[10] = vals[9]                              # same point, same scalar -> doubling in a bucket
    bases = corc.points_to_limbs(pts, cid, grp)
    sc = N.ints_to_limbs(vals, 4)
    exp = g.msm(pts, vals)
    got = corc.limbs_to_points(_msm(gpu, cid, grp, sc, bases), cid, grp)[0]
    assert got == exp
    # all scalars zero -> infinity ; everything cancels -> infinity
    zero = np.zeros((n, 4), dtype=np.uint64)
    assert not _msm(gpu, cid, grp, zero, bases).any()
    two = corc.points_to_limbs([pts[0], g.neg(pts[0])], cid, grp)
    assert not _msm(gpu, cid, grp, N.ints_to_limbs([5, 5]), two).any()


def test_msm_length_rules(gpu):
    bases = np.zeros((3, 8), dtype=np.uint64)
    sc = np.zeros((2, 4), dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    assert gpu.zk_msm(0, 1, 3, 2, N.u64p(sc), N.u64p(bases), N.u64p(out)) == N.ZK_ERR_LENGTH
    assert b"Number of points and scalars mismatch" in gpu.zk_last_error()
    assert gpu.zk_msm(0, 1, 0, 0, N.u64p(sc), N.u64p(bases), N.u64p(out)) == N.ZK_OK and not out.any()
    assert gpu.zk_msm(7, 1, 1, 1, N.u64p(sc), N.u64p(bases), N.u64p(out)) == N.ZK_ERR_ARG


@pytest.mark.parametrize("flags", [0, N.MSM_NO_GLV])
@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("c", [2, 3, 4, 7, 11, 13, 16])
def test_msm_window_sizes_and_sharding(gpu, name, cid, c, flags):
    """every window width gives the same point, with the scalars split by the endomorphism (the default for general G1
    plans: 2n entries in windows over 128 bits) and without; partial results of disjoint window ranges add up.
    Two-bit windows take one window more than ceil(bits / 2): the bias leaves only a third of the range above a positive
    value, and BLS12-381 half-scalars reach 0.673 * 2^127, its r 0.453 * 2^256 (round-2 advisor finding: ~0.9 % of random
    scalars overflowed the top digit of a split-scalar plan; the plain windows of BLS12-381 had the same hole)"""
    grp, cv = 1, pyref.curve_by_name(name)
    n = 3000
    _, bases = oracle_bases(cid, grp, n, 70)
    vals, _ = rand_scalars(n, cv.r, 71)
    vals[:6] = [0, 1, cv.r - 1, cv.r + 5, (1 << 256) - 1, cv.r // 2]   # reduced mod r like Fr::from(BigUint)
    sc = N.ints_to_limbs(vals, 4)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, c, h))
    try:
        cb, nw, ent = N._i(0), N._i(0), N._u64(0)
        N.check(gpu.zk_msm_plan_windows(h, cb, nw))
        N.check(gpu.zk_msm_plan_entries(h, ent))
        assert ent.value == (n if flags else 2 * n)
        assert cb.value == c and nw.value == -(-(cv.r.bit_length() + 1 if flags else 128) // c) + (1 if c == 2 else 0)
        lc, ln = N._i(0), N._i(0)
        N.check(gpu.zk_msm_window_layout(cid, grp, n, flags, c, lc, ln))
        assert (lc.value, ln.value) == (cb.value, nw.value)
        out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        assert (out == exp).all()
        from zksnake_amd.parallel import sum_points, window_ranges
        for world in (2, 3, 8):
            parts = []
            for first, count in window_ranges(nw.value, world):
                part = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
                if count:
                    N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(part), None))
                parts.append(part)
            assert (sum_points(cid, grp, parts) == exp).all()
        # fewer scalars than bases: the first k bases are used (ecc.py:118-119)
        N.check(gpu.zk_msm_plan_run(h, 100, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        assert (out == corc.msm(cid, grp, sc[:100], bases[:100], threads=8)).all()
        tm = (N.ctypes.c_float * 5)()
        assert gpu.zk_msm_plan_timings(h, tm, 5) == 5 and tm[1] > 0
    finally:
        N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("name,cid,grp,c", [("BN254", 0, 1, 0), ("BN254", 0, 1, 9), ("BN254", 0, 2, 0), ("BLS12_381", 1, 1, 0), ("BLS12_381", 1, 2, 6)])
def test_msm_precomputed_table(gpu, name, cid, grp, c):
    """ZK_MSM_PRECOMPUTE: fixed-base table 2^(c w) P_i with one shared bucket set gives the same point,
    also for window-range partials, truncated scalar vectors and the edge cases"""
    from zksnake_amd.parallel import sum_points, window_ranges
    cv = pyref.curve_by_name(name)
    g = pyref.Group(cv, grp)
    n = 700
    _, bases = oracle_bases(cid, grp, n, 90 + grp)
    pts = corc.limbs_to_points(bases, cid, grp)
    vals, _ = rand_scalars(n, cv.r, 91)
    vals[0], vals[1], vals[2] = 0, 1, cv.r - 1
    pts[5] = pts[4]; pts[7] = g.neg(pts[6]); vals[7] = vals[6]; pts[8] = None
    bases = corc.points_to_limbs(pts, cid, grp)
    sc = N.ints_to_limbs(vals, 4)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    PW = N.point_limbs(cid, grp)
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, N.MSM_PRECOMPUTE, c, h))
    try:
        out = np.zeros(PW, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        assert (out == exp).all()
        cb, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_plan_windows(h, cb, nw))
        parts = []
        for first, count in window_ranges(nw.value, 3):
            part = np.zeros(PW, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(part), None))
            parts.append(part)
        assert (sum_points(cid, grp, parts) == exp).all()
        N.check(gpu.zk_msm_plan_run(h, 50, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        assert (out == corc.msm(cid, grp, sc[:50], bases[:50], threads=8)).all()
        # asynchronous form on the plan's own stream
        N.check(gpu.zk_msm_plan_enqueue(h, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
        assert gpu.zk_msm_plan_enqueue(h, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN) == N.ZK_ERR_ARG  # one run in flight
        N.check(gpu.zk_msm_plan_finish(h, N.u64p(out)))
        assert (out == exp).all()
        assert gpu.zk_msm_plan_finish(h, N.u64p(out)) == N.ZK_ERR_ARG
    finally:
        N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_batch_mul(gpu, name, cid, grp):
    cv = pyref.curve_by_name(name)
    gen = generator_limbs(gpu, cid, grp)
    vals, sc = rand_scalars(50, cv.r, 80)
    vals[:4] = [0, 1, cv.r - 1, 2]
    sc = N.ints_to_limbs(vals, 4)
    out = np.zeros((50, N.point_limbs(cid, grp)), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, 50, N.u64p(sc), N.u64p(gen), 1, N.u64p(out)))
    exp = corc.batch_mul(cid, grp, sc, gen)
    assert (out == exp).all()
    # per-element bases
    out2 = np.zeros_like(out)
    N.check(gpu.zk_batch_mul(cid, grp, 50, N.u64p(sc), N.u64p(exp), 0, N.u64p(out2)))
    assert (out2 == corc.batch_mul(cid, grp, sc, exp)).all()


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_batch_mul_fixed_base_table(gpu, name, cid, grp):
    """one base, many scalars (what Groth16.setup does, protocol.py:81-97): the fixed-base path (16 signed 16-bit digits
    against a table of d * 2^(16 j) * G, batched inversion) against the oracle's double-and-add, edge scalars included;
    a second base must rebuild the cached table"""
    cv = pyref.curve_by_name(name)
    n = 2500
    gen = generator_limbs(gpu, cid, grp)
    vals, _ = rand_scalars(n, cv.r, 81 + grp)
    vals[:8] = [0, 1, cv.r - 1, 2, cv.r + 3, 1 << 15, (1 << 16) - 1, (1 << 255) - 1]   # >= r reduces like Fr::from
    vals[8] = sum(1 << (16 * j + 15) for j in range(15))                               # every digit at the carry boundary
    sc = N.ints_to_limbs(vals, 4)
    out = np.zeros((n, N.point_limbs(cid, grp)), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(sc), N.u64p(gen), 1, N.u64p(out)))
    exp = corc.batch_mul(cid, grp, sc, gen)
    assert (out == exp).all()
    other = exp[5].copy()                                                              # another base: 7-ish * G
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(sc), N.u64p(other), 1, N.u64p(out)))
    assert (out == corc.batch_mul(cid, grp, sc, other)).all()
    inf = np.zeros_like(gen)                                                           # the point at infinity as the base
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(sc), N.u64p(inf), 1, N.u64p(out)))
    assert not out.any()


@pytest.mark.parametrize("flags", [0, 1])
def test_plan_for_a_window_range(gpu, flags):
    """zk_msm_plan_create_range: a rank's plan holds only its windows (table rows, workspace); the partial points of the
    three ranks add up to the full MSM and a run outside the range is refused"""
    from zksnake_amd.parallel import sum_points, window_ranges
    cid, grp, cv = 0, 1, pyref.BN254
    n = 1500
    _, bases = oracle_bases(cid, grp, n, 95)
    _, sc = rand_scalars(n, cv.r, 96)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    lc, ln = N._i(0), N._i(0)
    N.check(gpu.zk_msm_window_layout(cid, grp, n, flags, 12, lc, ln))   # what a rank asks before it creates its plan
    nwin = ln.value
    assert lc.value == 12 and nwin == ((254 + 1 + 12 - 1) // 12 if flags else (128 + 12 - 1) // 12)
    parts = []
    for first, count in window_ranges(nwin, 3):
        h = N._u64(0)
        N.check(gpu.zk_msm_plan_create_range(cid, grp, n, bases.ctypes.data, 0, flags, 12, first, count, h))
        try:
            part = np.zeros(8, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(part), None))      # 0, 0 = the plan's range
            again = np.zeros(8, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(again), None))
            assert (part == again).all()
            if count > 1:
                a, b = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
                N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, 1, N.u64p(a), None))
                N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first + 1, count - 1, N.u64p(b), None))
                assert (sum_points(cid, grp, [a, b]) == part).all()
            outside = first - 1 if first > 0 else first + count
            assert gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, outside, 1, N.u64p(again), None) == N.ZK_ERR_ARG
            parts.append(part)
        finally:
            N.check(gpu.zk_msm_plan_destroy(h))
    assert (sum_points(cid, grp, parts) == exp).all()


def _known_dl_case(gpu, n, scalars_limbs, scalar_ints, cid=0, grp=1):
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    k_limbs, k_ints = W.field_stream(W.SEED_MSM_BASES, n, r)
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, N.point_limbs(cid, grp)), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(k_limbs), N.u64p(gen), 1, N.u64p(bases)))
    dot = sum(a * b for a, b in zip(scalar_ints, k_ints)) % r
    exp = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
    N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
    assert (_msm(gpu, cid, grp, scalars_limbs, bases) == exp).all()


def test_msm_2_20_closed_form(gpu):
    """BASELINE config 2 at full size: MSM(s, k_i G) == (sum s_i k_i mod r) G"""
    n = 1 << 20
    sc_limbs, sc_ints = W.field_stream(W.SEED_MSM_SCALARS, n, pyref.BN254.r)
    _known_dl_case(gpu, n, sc_limbs, sc_ints)


@pytest.mark.parametrize("cid,extra", [(0, 0), (0, 3), (1, 0)])
def test_msm_general_g1_at_the_split_scalar_limit(gpu, cid, extra):
    """2^22 points is the largest general G1 plan that splits its scalars with the endomorphism (2^23 entries per window,
    coarse bins above the LDS stage of the second sort level); three points more and the plan keeps the plain windows"""
    n = (1 << 22) + extra
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    sc_limbs, sc_ints = W.field_stream(W.SEED_MSM_SCALARS, n, r)
    lc, ln = N._i(0), N._i(0)
    N.check(gpu.zk_msm_window_layout(cid, 1, n, 0, 0, lc, ln))
    assert (lc.value, ln.value) == ((16, 8) if extra == 0 else (16, 16))
    _known_dl_case(gpu, n, sc_limbs, sc_ints, cid, 1)


@pytest.mark.parametrize("cid,grp", [(0, 1), (1, 2)])
def test_msm_largest_plan_2_26(gpu, cid, grp):
    """2^26 points is the largest plan the library accepts (one more is refused), here for the smallest and the largest point
    type (all four groups pass; the other two are left out for the suite's running time) and both plan modes:
    the general path with 16 plain windows (2^30 sort entries, no scalar split above 2^22 points) and the fixed-base path
    (13 x 2^26 table rows: 56 GB for BN254 G1, 167 GB for BLS12-381 G2; 20-bit windows, fine bucket bits beside the entries) --
    against the closed form (sum s_i k_i) G with the sum from the CPU oracle (corc.dot) and the final scalar multiplication
    from the oracle as well"""
    n = 1 << 26
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    PW = N.point_limbs(cid, grp)
    sc = W.splitmix64(0x26A + grp, 4 * n).reshape(n, 4)
    ks = W.splitmix64(0x26B + 8 * cid, 4 * n).reshape(n, 4)
    sc[:, 3] &= np.uint64((1 << 60) - 1)   # < 2^252 < r
    ks[:, 3] &= np.uint64((1 << 60) - 1)
    sc[:5] = N.ints_to_limbs([0, 1, r - 1, r - 2, 1 << 251])
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, PW), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
    assert (bases[-64:] == corc.batch_mul(cid, grp, ks[-64:], gen)).all()       # spot check of the inputs themselves
    dot = corc.dot(cid, sc, ks, threads=16)
    exp = corc.batch_mul(cid, grp, N.ints_to_limbs([dot]), gen)[0]
    del ks
    h = N._u64(0)
    assert gpu.zk_msm_plan_create(cid, grp, n + 1, bases.ctypes.data, 0, 0, 0, h) == N.ZK_ERR_ARG
    for flags, layout in ((0, (16, 16)), (N.MSM_PRECOMPUTE, (20, 13))):
        N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
        try:
            cb, nw = N._i(0), N._i(0)
            N.check(gpu.zk_msm_plan_windows(h, cb, nw))
            assert (cb.value, nw.value) == layout
            out = np.zeros(PW, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
            assert (out == exp).all(), flags
        finally:
            N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("cid,grp,log_n", [(0, 2, 16), (1, 1, 18), (1, 2, 15)])
def test_msm_closed_form_other_groups(gpu, cid, grp, log_n):
    """BN254 G2 and the BLS12-381 twins at sizes where the closed form is the only cheap oracle"""
    n = 1 << log_n
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    sc_limbs, sc_ints = W.field_stream(W.SEED_MSM_SCALARS, n, r)
    _known_dl_case(gpu, n, sc_limbs, sc_ints, cid, grp)


def _bases_from_library(gpu, cid, grp, n, seed):
    """n points k_i G made by zk_batch_mul (fast); they are only INPUTS here -- the expectation comes from the CPU oracle"""
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    ks = W.splitmix64(seed, 4 * n).reshape(n, 4)
    ks[:, 3] &= np.uint64((1 << 60) - 1)
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, N.point_limbs(cid, grp)), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(ks), N.u64p(gen), 1, N.u64p(bases)))
    assert (bases[:64] == corc.batch_mul(cid, grp, ks[:64], gen)).all()       # spot check of the inputs themselves
    return bases, r


@pytest.mark.parametrize("cid,grp,log_n", [(0, 1, 20), (0, 2, 19), (1, 1, 19), (1, 2, 19)])
def test_msm_full_size_against_cpu_oracle(gpu, cid, grp, log_n):
    """BASELINE config 2 (BN254 G1, 2^20 pairs) and the other three groups at 2^19, compared DIRECTLY with the CPU
    restatement of ark's Pippenger (oracle/zk_oracle.cpp, all host cores) -- not through the library's own
    batch_mul / point_mul closed form.  Both plan modes: the general path (two-level sort at n >= 2^19) and the fixed-base
    table with its shared bucket set (ZK_MSM_PRECOMPUTE, what every 2^20 proof runs), plus a window-range split of each."""
    import os
    from zksnake_amd.parallel import sum_points, window_ranges
    n = 1 << log_n
    bases, r = _bases_from_library(gpu, cid, grp, n, 0xB45E5 + 16 * cid + grp)
    sc = W.field_stream(W.SEED_MSM_SCALARS + 7 * cid + grp, n, r)[0]
    sc[:3] = N.ints_to_limbs([0, 1, r - 1])
    exp = corc.msm(cid, grp, sc, bases, threads=min(16, os.cpu_count() or 1))
    PW = N.point_limbs(cid, grp)
    for flags in (0, N.MSM_PRECOMPUTE):
        h = N._u64(0)
        N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
        try:
            out = np.zeros(PW, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
            assert (out == exp).all(), f"flags={flags}"
            cb, nw = N._i(0), N._i(0)
            N.check(gpu.zk_msm_plan_windows(h, cb, nw))
            parts = []
            for first, count in window_ranges(nw.value, 8):                  # the 8-rank split of BASELINE config 5
                part = np.zeros(PW, dtype=np.uint64)
                N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(part), None))
                parts.append(part)
            assert (sum_points(cid, grp, parts) == exp).all(), f"flags={flags} sharded"
        finally:
            N.check(gpu.zk_msm_plan_destroy(h))


def test_msm_skewed_scalars(gpu):
    """benchmark-witness-like single-bit scalars and one hot bucket (SURVEY 8d skew variant)"""
    n = 1 << 16
    limbs, ints = W.powers_of_two_scalars(n, pyref.BN254.r)
    _known_dl_case(gpu, n, limbs, ints)
    ones = [1] * n
    _known_dl_case(gpu, n, N.ints_to_limbs(ones), ones)
    same = [0xDEADBEEFCAFEBABE1234567] * n
    _known_dl_case(gpu, n, N.ints_to_limbs(same), same)


@pytest.mark.parametrize("flags", [0, 1])
def test_msm_skewed_scalars_full_size(gpu, flags):
    """2^20 pairs with degenerate scalar distributions, both plan modes (the fixed-base plan then runs 20-bit windows
    on one shared bucket set): every scalar equal (one bucket per window holds all 2^20 entries: workgroup-tier combine,
    unstaged level-B sort), all ones, and single-bit scalars 2^(i mod 254) like the benchmark circuit's witness"""
    cid, grp, r = 0, 1, pyref.BN254.r
    n = 1 << 20
    k_limbs, k_ints = W.field_stream(W.SEED_MSM_BASES + 5, n, r)
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, 8), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(k_limbs), N.u64p(gen), 1, N.u64p(bases)))
    ksum = sum(k_ints) % r
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
    try:
        out, exp = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        same = 0xDEADBEEFCAFEBABE1234567890ABCDEF0123456789ABCDEF
        for value in (same, 1, r - 1):
            sc = np.tile(N.ints_to_limbs([value]), (n, 1))
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
            N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([value * ksum % r])), N.u64p(exp)))
            assert (out == exp).all(), hex(value)
        limbs, ints = W.powers_of_two_scalars(n, r)
        dot = sum(a * b for a, b in zip(ints, k_ints)) % r
        N.check(gpu.zk_msm_plan_run(h, n, limbs.ctypes.data, 0, 0, 0, N.u64p(out), None))
        N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
        assert (out == exp).all()
    finally:
        N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("n", [12345, 70001, (1 << 18) + 3, (1 << 19) + 1])
def test_msm_odd_sizes_and_window_ranges(gpu, n):
    """sizes that are not powers of two (ragged last chunk / segment), as one run and as the three window ranges of a
    3-rank sharded run whose partial points are summed on the host (zk_point_sum)"""
    from zksnake_amd.parallel import sum_points, window_ranges
    cid, grp, r = 0, 1, pyref.BN254.r
    sc_limbs, sc_ints = W.field_stream(0xABCD + n, n, r)
    k_limbs, k_ints = W.field_stream(0x1234 + n, n, r)
    gen = np.zeros(8, dtype=np.uint64)
    N.check(gpu.zk_point_generator(cid, grp, N.u64p(gen)))
    bases = np.zeros((n, 8), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(k_limbs), N.u64p(gen), 1, N.u64p(bases)))
    dot = sum(a * b for a, b in zip(sc_ints, k_ints)) % r
    exp = np.zeros(8, dtype=np.uint64)
    N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, 0, 0, h))
    out = np.zeros(8, dtype=np.uint64)
    N.check(gpu.zk_msm_plan_run(h, n, sc_limbs.ctypes.data, 0, 0, 0, N.u64p(out), None))
    assert (out == exp).all()
    c, nwin = N._i(0), N._i(0)
    N.check(gpu.zk_msm_plan_windows(h, c, nwin))
    parts = []
    for first, count in window_ranges(nwin.value, 3):
        part = np.zeros(8, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc_limbs.ctypes.data, 0, first, count, N.u64p(part), None))
        parts.append(part)
    assert (sum_points(cid, grp, parts) == exp).all()
    # a shorter scalar vector against the same plan (multiexp's truncation rule): the first n - 5 bases only
    m = n - 5
    dot_m = sum(a * b for a, b in zip(sc_ints[:m], k_ints[:m])) % r
    N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot_m])), N.u64p(exp)))
    N.check(gpu.zk_msm_plan_run(h, m, sc_limbs.ctypes.data, 0, 0, 0, N.u64p(out), None))
    assert (out == exp).all()
    N.check(gpu.zk_msm_plan_destroy(h))


def test_fixed_base_plan_ragged_size_above_2_20(gpu):
    """(2^20 + 77) points: wide windows with a ragged last chunk / tile, a shorter scalar vector against the same plan and
    a clone of the plan (shared table, own workspace) running concurrently"""
    cid, grp, r = 0, 1, pyref.BN254.r
    n = (1 << 20) + 77
    sc_limbs, sc_ints = W.field_stream(0xF00D, n, r)
    k_limbs, k_ints = W.field_stream(0xBEEF, n, r)
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, 8), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(k_limbs), N.u64p(gen), 1, N.u64p(bases)))

    def expect(m):
        e = np.zeros(8, dtype=np.uint64)
        dot = sum(a * b for a, b in zip(sc_ints[:m], k_ints[:m])) % r
        N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(e)))
        return e

    h, h2 = N._u64(0), N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, N.MSM_PRECOMPUTE, 0, h))
    N.check(gpu.zk_msm_plan_clone(h, h2))
    try:
        cb, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_plan_windows(h, cb, nw))
        assert cb.value == 20 and nw.value == 13          # the wide-window layout is what this test is about
        a, b = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_enqueue(h, n, sc_limbs.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
        N.check(gpu.zk_msm_plan_enqueue(h2, n - 1000, sc_limbs.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
        N.check(gpu.zk_msm_plan_finish(h, N.u64p(a)))
        N.check(gpu.zk_msm_plan_finish(h2, N.u64p(b)))
        assert (a == expect(n)).all() and (b == expect(n - 1000)).all()
        N.check(gpu.zk_msm_plan_destroy(h))               # the clone keeps the shared table alive
        h = None
        N.check(gpu.zk_msm_plan_run(h2, n, sc_limbs.ctypes.data, 0, 0, 0, N.u64p(b), None))
        assert (b == a).all()
    finally:
        if h is not None:
            N.check(gpu.zk_msm_plan_destroy(h))
        N.check(gpu.zk_msm_plan_destroy(h2))


@pytest.mark.parametrize("cid,grp,n", [(0, 1, (1 << 21) + 5), (1, 1, 1 << 21), (0, 2, (1 << 20) + (1 << 19))])
def test_fixed_base_plan_above_2_20_keeps_20_bit_windows(gpu, cid, grp, n):
    """13 x n table rows no longer fit beside the fine bucket bits in a 32-bit sort entry: the fine bits travel in a byte
    array of their own and the plan keeps its 13 windows of 20 bits (round 2 fell back to 17-bit windows here)"""
    r = (pyref.BN254 if cid == 0 else pyref.BLS12_381).r
    sc_limbs, sc_ints = W.field_stream(0xF00D + n, n, r)
    k_limbs, k_ints = W.field_stream(0xBEEF + n, n, r)
    PW = N.point_limbs(cid, grp)
    gen = generator_limbs(gpu, cid, grp)
    bases = np.zeros((n, PW), dtype=np.uint64)
    N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(k_limbs), N.u64p(gen), 1, N.u64p(bases)))
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, N.MSM_PRECOMPUTE, 0, h))
    try:
        cb, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_plan_windows(h, cb, nw))
        assert (cb.value, nw.value) == (20, 13)
        for m in (n, n - 12345):
            out, exp = np.zeros(PW, dtype=np.uint64), np.zeros(PW, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, m, sc_limbs.ctypes.data, 0, 0, 0, N.u64p(out), None))
            dot = sum(a * b for a, b in zip(sc_ints[:m], k_ints[:m])) % r
            N.check(gpu.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([dot])), N.u64p(exp)))
            assert (out == exp).all()
    finally:
        N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("flags,n", [(1, 5000), (N.MSM_NO_GLV, 5000), (1, 1 << 20)])
def test_shared_sort_between_g1_and_g2_plans(gpu, flags, n):
    """zk_msm_plan_enqueue_shared: the G2 plan runs on the digits and the sorted entry list of the G1 plan's run in flight
    (Groth16's <tau_1, v> and <tau_2, v>); results equal the independently sorted runs, the lender can run again at once,
    and plans of different sizes refuse to share"""
    cid, r = 0, pyref.BN254.r
    sc = W.field_stream(0x5A5A, n, r)[0]
    sc2 = W.field_stream(0xA5A5, n, r)[0]
    plans, singles = {}, {}
    for grp in (1, 2):
        bases, _ = _bases_from_library(gpu, cid, grp, n, 0xC0FFEE + grp)
        h = N._u64(0)
        N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
        plans[grp] = h
        for tag, s_ in (("a", sc), ("b", sc2)):
            out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, s_.ctypes.data, 0, 0, 0, N.u64p(out), None))
            singles[(grp, tag)] = out
    try:
        for tag, s_ in (("a", sc), ("b", sc2), ("a", sc)):   # back to back: the lender's buffers are reused while lent
            o1, o2 = np.zeros(8, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_enqueue(plans[1], n, s_.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
            N.check(gpu.zk_msm_plan_enqueue_shared(plans[2], plans[1], N.STREAM_PLAN))
            N.check(gpu.zk_msm_plan_finish(plans[1], N.u64p(o1)))
            N.check(gpu.zk_msm_plan_finish(plans[2], N.u64p(o2)))
            assert (o1 == singles[(1, tag)]).all() and (o2 == singles[(2, tag)]).all()
        # nothing in flight on the lender -> refused; a plan of another size -> refused
        assert gpu.zk_msm_plan_enqueue_shared(plans[2], plans[1], N.STREAM_PLAN) == N.ZK_ERR_ARG
        small_bases, _ = _bases_from_library(gpu, cid, 2, 64, 0xC0FFEE)
        hs = N._u64(0)
        N.check(gpu.zk_msm_plan_create(cid, 2, 64, small_bases.ctypes.data, 0, flags, 0, hs))
        N.check(gpu.zk_msm_plan_enqueue(plans[1], n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
        assert gpu.zk_msm_plan_enqueue_shared(hs, plans[1], N.STREAM_PLAN) == N.ZK_ERR_ARG
        o1 = np.zeros(8, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_finish(plans[1], N.u64p(o1)))
        assert (o1 == singles[(1, "a")]).all()
        N.check(gpu.zk_msm_plan_destroy(hs))
        if flags == N.MSM_NO_GLV:
            # two split-scalar plans of different groups: each splits against its own eigenvalue, the digits do not mix
            b1, _ = _bases_from_library(gpu, cid, 1, n, 0xC0FFEE + 1)
            b2, _ = _bases_from_library(gpu, cid, 2, n, 0xC0FFEE + 2)
            hg1, hg2 = N._u64(0), N._u64(0)
            N.check(gpu.zk_msm_plan_create(cid, 1, n, b1.ctypes.data, 0, 0, 0, hg1))
            N.check(gpu.zk_msm_plan_create(cid, 2, n, b2.ctypes.data, 0, 0, 0, hg2))
            N.check(gpu.zk_msm_plan_enqueue(hg1, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
            assert gpu.zk_msm_plan_enqueue_shared(hg2, hg1, N.STREAM_PLAN) == N.ZK_ERR_ARG
            N.check(gpu.zk_msm_plan_finish(hg1, N.u64p(o1)))
            assert (o1 == singles[(1, "a")]).all()
            o2 = np.zeros(16, dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(hg2, n, sc.ctypes.data, 0, 0, 0, N.u64p(o2), None))
            assert (o2 == singles[(2, "a")]).all()       # the split-scalar G2 plan agrees with the plain one
            N.check(gpu.zk_msm_plan_destroy(hg1))
            N.check(gpu.zk_msm_plan_destroy(hg2))
            # a general G1 plan that splits its scalars with the endomorphism sorts 2n half-scalars: nothing a G2 plan could use
            bases1, _ = _bases_from_library(gpu, cid, 1, n, 0xC0FFEE + 1)
            hg = N._u64(0)
            N.check(gpu.zk_msm_plan_create(cid, 1, n, bases1.ctypes.data, 0, 0, 0, hg))
            N.check(gpu.zk_msm_plan_enqueue(hg, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
            assert gpu.zk_msm_plan_enqueue_shared(plans[2], hg, N.STREAM_PLAN) == N.ZK_ERR_ARG
            N.check(gpu.zk_msm_plan_finish(hg, N.u64p(o1)))
            assert (o1 == singles[(1, "a")]).all()
            N.check(gpu.zk_msm_plan_destroy(hg))
    finally:
        for h in plans.values():
            N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("flags", [0, N.MSM_PRECOMPUTE])
def test_plan_options_switch_paths_in_process(gpu, flags):
    """zk_msm_plan_set_option flips the tuning knobs of ONE live plan (they used to be process-wide environment latches, and
    this test had to run a child process): the chunked one-level sort (which otherwise only serves n > 2^24), one-step
    row / column sums, other lane counts, the accumulate kernel with and without its priority steps -- every combination gives
    the oracle's point; bad names and values are refused"""
    cid, grp, n = 0, 1, 1 << 19
    bases, r = _bases_from_library(gpu, cid, grp, n, 0x0F710)
    sc = W.splitmix64(0x5CA1A, 4 * n).reshape(n, 4)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    exp = corc.msm(cid, grp, sc, bases, threads=16)
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
    try:
        def run():
            out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
            tm = (N.ctypes.c_float * 5)()
            gpu.zk_msm_plan_timings(h, tm, 5)
            return out, list(tm)

        base, _ = run()
        assert (base == exp).all()
        for name, value in ((b"two_level_sort", 0), (b"sum_one_step", 1), (b"lanes_per_output", 16), (b"segment_lanes", 65536),
                            (b"priority_steps", 0), (b"two_level_sort", 1), (b"sum_one_step", 0), (b"lanes_per_output", 0),
                            (b"priority_steps", 1)):
            N.check(gpu.zk_msm_plan_set_option(h, name, value))
            out, _ = run()
            assert (out == exp).all(), (name, value)
        for name, value in ((b"no_such_option", 1), (b"fine_log", 7), (b"lanes_per_output", 48), (b"segment_lanes", 1 << 30), (b"segment_lanes", 3)):
            assert gpu.zk_msm_plan_set_option(h, name, value) == N.ZK_ERR_ARG, (name, value)
        # not while a run is in flight
        N.check(gpu.zk_msm_plan_enqueue(h, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
        assert gpu.zk_msm_plan_set_option(h, b"sum_one_step", 1) == N.ZK_ERR_ARG
        out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm_plan_finish(h, N.u64p(out)))
        assert (out == exp).all()
    finally:
        N.check(gpu.zk_msm_plan_destroy(h))
    assert gpu.zk_msm_plan_set_option(h, b"sum_one_step", 1) == N.ZK_ERR_ARG     # unknown handle


@pytest.mark.parametrize("flags", [0, 1])
def test_two_step_enqueue_orders_the_accumulate_kernels(gpu, flags):
    """zk_msm_plan_enqueue_sort + zk_msm_plan_enqueue_rest: three plans sort side by side, their accumulate kernels run in
    a chain (each after the previous plan's), the results equal the one-call runs; misuse is refused"""
    cid, r = 0, pyref.BN254.r
    n = 60000
    plans, expect, scal = [], [], []
    for k, grp in enumerate((1, 2, 1)):
        bases, _ = _bases_from_library(gpu, cid, grp, n, 0xAB + k)
        sc = W.field_stream(0x77 + k, n, r)[0]
        h = N._u64(0)
        N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
        out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        plans.append(h); expect.append(out); scal.append(sc)
    try:
        for _ in range(2):
            for h, sc in zip(plans, scal):
                N.check(gpu.zk_msm_plan_enqueue_sort(h, n, sc.ctypes.data, 0, 0, 0, N.STREAM_PLAN))
            tmp = np.zeros(16, dtype=np.uint64)
            assert gpu.zk_msm_plan_finish(plans[0], N.u64p(tmp)) == N.ZK_ERR_ARG          # accumulate not enqueued yet
            assert gpu.zk_msm_plan_enqueue_sort(plans[0], n, scal[0].ctypes.data, 0, 0, 0, N.STREAM_PLAN) == N.ZK_ERR_ARG
            N.check(gpu.zk_msm_plan_enqueue_rest(plans[1], 0))               # the G2 plan first
            N.check(gpu.zk_msm_plan_enqueue_rest(plans[0], plans[1]))
            N.check(gpu.zk_msm_plan_enqueue_rest(plans[2], plans[0]))
            assert gpu.zk_msm_plan_enqueue_rest(plans[2], 0) == N.ZK_ERR_ARG  # nothing sorted any more
            for h, e in zip(plans, expect):
                out = np.zeros(e.shape[0], dtype=np.uint64)
                N.check(gpu.zk_msm_plan_finish(h, N.u64p(out)))
                assert (out == e).all()
        assert gpu.zk_msm_plan_enqueue_rest(plans[0], plans[0]) == N.ZK_ERR_ARG
    finally:
        for h in plans:
            N.check(gpu.zk_msm_plan_destroy(h))


@pytest.mark.parametrize("grp", [1, 2])
@pytest.mark.parametrize("name,cid", CURVES)
def test_split_scalar_plan_on_the_decomposition_corner_cases(gpu, name, cid, grp):
    """scalars at and around lambda, r - lambda, the thirds of r, powers of two and their negatives: the places where the
    short-lattice rounding of the split-scalar digits kernel changes sign or carries"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gen_glv_params", os.path.join(os.path.dirname(__file__), "..", "tools", "gen_glv_params.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    cs = G.constants("Bn254" if cid == 0 else "Bls381") if grp == 1 else G.constants_g2("Bn254G2" if cid == 0 else "Bls381G2")
    cv = pyref.curve_by_name(name)
    r, lam = cv.r, cs["lam"]
    vals = [0, 1, 2, r - 1, r - 2, r // 2, r // 2 + 1, lam, lam + 1, lam - 1, r - lam, r - lam + 1, (r - 1) // 3, 2 * (r - 1) // 3,
            lam * lam % r, (lam * lam + 1) % r, cs["a1"] % r, (-cs["b1"]) % r, cs["a2"] % r, cs["b2"] % r]
    vals += [(1 << b) % r for b in range(0, 256, 7)] + [(r - (1 << b)) % r for b in range(0, 254, 9)]
    n = len(vals)
    _, bases = oracle_bases(cid, grp, n, 123)
    sc = N.ints_to_limbs(vals, 4)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    for flags in (0, N.MSM_NO_GLV):
        for c in (0, 16, 13):
            h = N._u64(0)
            N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, c, h))
            ent = N._u64(0)
            N.check(gpu.zk_msm_plan_entries(h, ent))
            assert ent.value == (n if flags else 2 * n)      # every group splits its scalars unless told not to
            out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
            N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
            N.check(gpu.zk_msm_plan_destroy(h))
            assert (out == exp).all(), (flags, c)
    # every scalar alone against one base: a wrong half-scalar cannot hide in a sum
    for i in range(n):
        one = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm(cid, grp, 1, 1, N.u64p(sc[i:i + 1].copy()), N.u64p(bases[i:i + 1].copy()), N.u64p(one)))
        assert (one == corc.msm(cid, grp, sc[i:i + 1], bases[i:i + 1], threads=1)).all(), vals[i]


def test_point_array_plan_cache_follows_the_window_layout(gpu):
    """PointArray.plan() caches per (slot, mode); a later call with another window width, or after window_range changed,
    must not get the old plan back (round-2 advisor finding) -- and the results stay the same point"""
    from zksnake_amd._algebra import PointArray
    cid, grp, n = 0, 1, 700
    cv = pyref.BN254
    _, bases = oracle_bases(cid, grp, n, 31)
    _, sc = rand_scalars(n, cv.r, 32)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    arr = PointArray(cid, grp, bases)

    def run(h, first=0, count=0):
        out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(out), None))
        return out

    def windows(h):
        c, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_plan_windows(h, c, nw))
        return c.value, nw.value

    h0 = arr.plan(0, precompute=True, window_bits=9)
    assert windows(h0)[0] == 9 and (run(h0) == exp).all()
    assert arr.plan(0, precompute=True) == h0                     # 0 = "whatever the plans have"
    h1 = arr.plan(1, precompute=True, window_bits=9)              # a clone of the same layout
    assert h1 != h0 and windows(h1) == windows(h0)
    h2 = arr.plan(0, precompute=True, window_bits=12)             # another width: both slots are rebuilt
    assert windows(h2)[0] == 12 and (run(h2) == exp).all()
    assert windows(arr.plan(1, precompute=True))[0] == 12
    # a window range (a sharded rank's share): the full-range plans go, the partial results add up
    from zksnake_amd.parallel import sum_points
    nw = windows(h2)[1]
    parts = []
    for first, count in ((0, nw // 2), (nw // 2, nw - nw // 2)):
        arr.window_range = (first, count)
        parts.append(run(arr.plan(0, precompute=True, window_bits=12), first, count))
    assert (sum_points(cid, grp, parts) == exp).all()
    arr.release()


@pytest.mark.parametrize("cid,grp", [(0, 1), (1, 2)])
def test_slots_with_their_own_window_ranges_and_wide_windows(gpu, cid, grp):
    """a rank of the task-partitioned prover may hold <tau_1, u> whole and a window range of <tau_1, v> (two slots of ONE key
    vector with different layouts), and from 2^21 constraints on its range plans take the 20-bit windows of an unsharded plan:
    per-slot ranges in PointArray, range plans with an explicit width, partial results that add up to the oracle's MSM"""
    from zksnake_amd._algebra import PointArray
    from zksnake_amd.parallel import sum_points
    cv = pyref.BN254 if cid == 0 else pyref.BLS12_381
    n = 1500
    _, bases = oracle_bases(cid, grp, n, 41 + cid)
    _, sc = rand_scalars(n, cv.r, 42 + cid)
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    arr = PointArray(cid, grp, bases)

    def run(h, first=0, count=0):
        out = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, first, count, N.u64p(out), None))
        return out

    def windows(h):
        c, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_plan_windows(h, c, nw))
        return c.value, nw.value

    whole = arr.plan(0, precompute=True, window_bits=20)
    c, nw = windows(whole)
    assert c == 20 and nw == (cv.r.bit_length() + 1 + 19) // 20 and (run(whole) == exp).all()
    parts = []
    cuts = [0, 4, 8, nw]
    for first, last in zip(cuts, cuts[1:]):
        arr.slot_ranges[1] = (first, last - first)
        h = arr.plan(1, precompute=True, window_bits=20)
        assert h != whole and windows(h) == (20, nw)
        parts.append(run(h, first, last - first))
        assert arr.plan(0, precompute=True, window_bits=20) == whole      # the other slot keeps its plan
    assert (sum_points(cid, grp, parts) == exp).all()
    # a range that covers everything shares the whole plan's table (a clone), and dropping the range brings it back too
    arr.slot_ranges.pop(1)
    h = arr.plan(1, precompute=True, window_bits=20)
    assert h != whole and (run(h) == exp).all()
    # concurrent=True reaches a plan that an earlier, non-concurrent caller created (round-3 advisor finding)
    assert arr._plan_concurrent[whole] is False
    assert arr.plan(0, precompute=True, window_bits=20, concurrent=True) == whole and arr._plan_concurrent[whole] is True
    arr.release()
    assert not arr._plans and not arr._plan_layout


def test_window_layouts_of_sharded_and_unsharded_plans(gpu):
    """zk_msm_window_layout_ex tells a rank of the task-partitioned prover both layouts of an MSM before any plan exists: the
    one zk_msm_plan_create_range takes (16 windows of 16 bits for fixed-base plans, which split evenly) and the one
    zk_msm_plan_create takes (13 of 20 bits from 2^20 points on); zk_msm_window_layout is the former"""
    def layout(cid, grp, n, flags, all_windows, bits=0):
        c, nw = N._i(0), N._i(0)
        N.check(gpu.zk_msm_window_layout_ex(cid, grp, n, flags, bits, all_windows, c, nw))
        return c.value, nw.value
    for cid, scalar_bits in ((0, 254), (1, 255)):
        for grp in (1, 2):
            assert layout(cid, grp, 1 << 20, N.MSM_PRECOMPUTE, 0) == (16, 16)
            assert layout(cid, grp, 1 << 20, N.MSM_PRECOMPUTE, 1) == (20, (scalar_bits + 1 + 19) // 20)
            assert layout(cid, grp, 1 << 19, N.MSM_PRECOMPUTE, 1) == (16, 16)        # below 2^20 points the wide windows do not pay
            assert layout(cid, grp, 1 << 23, N.MSM_PRECOMPUTE, 0, bits=20) == (20, (scalar_bits + 1 + 19) // 20)   # explicit width: ranges too
            c, nw = N._i(0), N._i(0)
            N.check(gpu.zk_msm_window_layout(cid, grp, 1 << 20, N.MSM_PRECOMPUTE, 0, c, nw))
            assert (c.value, nw.value) == layout(cid, grp, 1 << 20, N.MSM_PRECOMPUTE, 0)
    # the layout a plan reports is the one the query promised
    from zksnake_amd._algebra import PointArray
    _, bases = oracle_bases(0, 1, 300, 77)
    arr = PointArray(0, 1, bases)
    c, nw = N._i(0), N._i(0)
    N.check(gpu.zk_msm_plan_windows(arr.plan(0, precompute=True), c, nw))
    assert (c.value, nw.value) == layout(0, 1, 300, N.MSM_PRECOMPUTE, 1)
    arr.release()
    assert gpu.zk_msm_window_layout_ex(2, 1, 100, 0, 0, 0, c, nw) == N.ZK_ERR_ARG


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("flags", [0, N.MSM_PRECOMPUTE])
def test_pair_split_g2_accumulate_kernel_equals_the_one_lane_kernel(gpu, cid, flags):
    """the G2 accumulate step with every Fp2 value split by component over a lane pair (fp2_split.hip.h; default for BLS12-381 G2,
    selectable for BN254 G2) against the one-lane kernel and the oracle: random scalars, and the cases that leave the main path --
    the point at infinity as a base, P and -P in one bucket (cancellation), the same point twice with equal scalars (the doubling
    branch), scalars 0 / 1 / r - 1, a bucket that receives one entry only"""
    grp = 2
    cv = pyref.BN254 if cid == 0 else pyref.BLS12_381
    n = 2500
    _, bases = oracle_bases(cid, grp, n, 51 + cid)
    _, sc = rand_scalars(n, cv.r, 52 + cid)
    PW = N.point_limbs(cid, grp)
    # infinity, P / -P with equal scalars, duplicates with equal scalars, edge scalars
    bases[5] = 0
    neg = np.zeros(PW, dtype=np.uint64)
    N.check(gpu.zk_point_neg(cid, grp, N.u64p(bases[10]), N.u64p(neg)))
    bases[11] = neg
    sc[11] = sc[10]
    bases[21] = bases[20]
    sc[21] = sc[20]
    bases[31] = bases[30]
    bases[32] = bases[30]
    sc[31] = sc[30]
    sc[32] = sc[30]
    sc[40] = 0
    sc[41] = N.ints_to_limbs([1])[0]
    sc[42] = N.ints_to_limbs([cv.r - 1])[0]
    exp = corc.msm(cid, grp, sc, bases, threads=8)
    h = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, grp, n, bases.ctypes.data, 0, flags, 0, h))
    outs = {}
    for mode in (0, 1, -1):
        N.check(gpu.zk_msm_plan_set_option(h, b"split_pairs", mode))
        out = np.zeros(PW, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, n, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        outs[mode] = out
        assert (out == exp).all(), f"split_pairs = {mode}"
    # a short MSM: most buckets hold a single entry (the accumulator is replaced, never added to)
    for mode in (0, 1):
        N.check(gpu.zk_msm_plan_set_option(h, b"split_pairs", mode))
        out = np.zeros(PW, dtype=np.uint64)
        N.check(gpu.zk_msm_plan_run(h, 7, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        assert (out == corc.msm(cid, grp, sc[:7], bases[:7], threads=1)).all()
    assert gpu.zk_msm_plan_set_option(h, b"split_pairs", 2) == N.ZK_ERR_ARG
    N.check(gpu.zk_msm_plan_destroy(h))
    # the base-field groups have no such kernel
    _, b1 = oracle_bases(cid, 1, 16, 3)
    h1 = N._u64(0)
    N.check(gpu.zk_msm_plan_create(cid, 1, 16, b1.ctypes.data, 0, 0, 0, h1))
    assert gpu.zk_msm_plan_set_option(h1, b"split_pairs", 1) == N.ZK_ERR_ARG
    N.check(gpu.zk_msm_plan_set_option(h1, b"split_pairs", 0))
    N.check(gpu.zk_msm_plan_destroy(h1))
