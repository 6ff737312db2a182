"""GPU parity: batched point (de)compression (zk_points_compress / zk_points_decompress) against the CPU oracle's
per-point encodings (oracle/pyref.py compress / decompress, pinned by the reference's golden points), all four groups,
with the error behaviour of the reference's from_hex (ark deserialize_compressed) on damaged input."""

import numpy as np
import pytest

from helpers import CURVES, oracle_bases
from oracle import corc, pyref
from zksnake_amd import _native as N
from zksnake_amd._algebra import PointArray

pytestmark = pytest.mark.gpu


def _compress(lib, cid, grp, limbs):
    nb = lib.zk_point_bytes(cid, grp)
    out = np.zeros(limbs.shape[0] * nb, dtype=np.uint8)
    bad = N._u64(0)
    N.check(lib.zk_points_compress(cid, grp, limbs.shape[0], N.u64p(limbs), N.u8p(out), bad))
    return out


def _decompress(lib, cid, grp, raw, n):
    out = np.zeros((n, N.point_limbs(cid, grp)), dtype=np.uint64)
    bad = N._u64(0)
    rc = lib.zk_points_decompress(cid, grp, n, N.u8p(raw), N.u64p(out), bad)
    return rc, bad.value, out


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_batch_codec_matches_oracle(gpu, name, cid, grp):
    cv = pyref.curve_by_name(name)
    n = 300
    _, bases = oracle_bases(cid, grp, n, 7 + grp)
    bases[5] = 0      # the point at infinity
    bases[n - 1] = 0
    pts = corc.limbs_to_points(bases, cid, grp)
    want = b"".join(pyref.compress(cv, grp, p) for p in pts)
    got = _compress(gpu, cid, grp, bases)
    assert got.tobytes() == want
    rc, _, back = _decompress(gpu, cid, grp, got, n)
    assert rc == 0 and (back == bases).all()
    # the per-point host codec agrees
    nb = gpu.zk_point_bytes(cid, grp)
    one = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
    for i in (0, 5, 17):
        N.check(gpu.zk_point_decompress(cid, grp, N.u8p(got[i * nb:(i + 1) * nb].copy()), N.u64p(one)))
        assert (one == bases[i]).all()


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_batch_decompress_rejects_like_from_hex(gpu, name, cid, grp):
    """first offending point decides; same messages as the per-point path"""
    n = 200
    _, bases = oracle_bases(cid, grp, n, 11)
    good = _compress(gpu, cid, grp, bases)
    nb = gpu.zk_point_bytes(cid, grp)
    one = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)

    def damaged(index, mutate):
        raw = good.copy()
        mutate(raw[index * nb:(index + 1) * nb])
        return raw

    def all_ones(p):
        p[:] = 0xFF if cid == N.CURVE_BN254 else p
        if cid != N.CURVE_BN254:
            p[0] &= 0x7F  # BLS: clear the "compressed" bit

    def x_out_of_range(p):
        if cid == N.CURVE_BN254:
            p[:] = 0xFF
            p[nb - 1] = 0x3F
            if grp == 2:
                p[nb // 2 - 1] = 0x3F
        else:
            p[:] = 0xFF
            p[0] = 0x9F

    def infinity_with_x(p):
        if cid == N.CURVE_BN254:
            p[nb - 1] |= 0x40
            p[nb - 1] &= 0x7F
        else:
            p[0] = (p[0] | 0x40) & 0xDF

    for index, mutate in ((3, all_ones), (150, x_out_of_range), (199, infinity_with_x)):
        raw = damaged(index, mutate)
        rc, bad, _ = _decompress(gpu, cid, grp, raw, n)
        assert rc == N.ZK_ERR_POINT and bad == index
        msg = gpu.zk_last_error()
        assert gpu.zk_point_decompress(cid, grp, N.u8p(raw[index * nb:(index + 1) * nb].copy()), N.u64p(one)) == N.ZK_ERR_POINT
        assert gpu.zk_last_error() == msg
    # two damaged points: the lower index is reported
    raw = damaged(150, x_out_of_range)
    all_ones(raw[20 * nb:21 * nb])
    rc, bad, _ = _decompress(gpu, cid, grp, raw, n)
    assert rc == N.ZK_ERR_POINT and bad == 20


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_batch_decompress_checks_curve_and_subgroup(gpu, name, cid, grp):
    """an x with no y on the curve, and (where the cofactor is not 1) a curve point outside the r-torsion"""
    n = 100
    _, bases = oracle_bases(cid, grp, n, 13)
    good = _compress(gpu, cid, grp, bases)
    nb = gpu.zk_point_bytes(cid, grp)
    one = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
    # walk x = 1, 2, ... (c1 = 0 in G2): collect one x off the curve and one on the curve but outside the subgroup,
    # as classified by the per-point host codec
    off_curve = not_subgroup = None
    fb = nb // grp
    x = 1
    while off_curve is None or (not_subgroup is None and not (cid == N.CURVE_BN254 and grp == 1)):
        if cid == N.CURVE_BN254:
            enc = x.to_bytes(fb, "little") + bytes(nb - fb)
        else:
            enc = bytearray(bytes(nb - fb) + x.to_bytes(fb, "big"))
            enc[0] |= 0x80
        enc = np.frombuffer(bytes(enc), dtype=np.uint8).copy()
        rc = gpu.zk_point_decompress(cid, grp, N.u8p(enc), N.u64p(one))
        msg = gpu.zk_last_error().decode()
        if rc and "not on the curve" in msg and off_curve is None:
            off_curve = (enc, msg)
        if rc and "subgroup" in msg and not_subgroup is None:
            not_subgroup = (enc, msg)
        x += 1
        assert x < 200
    for case in (off_curve, not_subgroup):
        if case is None:
            continue
        raw = good.copy()
        raw[42 * nb:43 * nb] = case[0]
        rc, bad, _ = _decompress(gpu, cid, grp, raw, n)
        assert rc == N.ZK_ERR_POINT and bad == 42 and gpu.zk_last_error().decode() == case[1]


def test_batch_compress_rejects_off_curve_point(gpu):
    _, bases = oracle_bases(N.CURVE_BN254, 1, 100, 17)
    bases[31, 0] ^= 1
    out = np.zeros(100 * 32, dtype=np.uint8)
    bad = N._u64(0)
    assert gpu.zk_points_compress(N.CURVE_BN254, 1, 100, N.u64p(bases), N.u8p(out), bad) == N.ZK_ERR_POINT
    assert bad.value == 31 and b"not on the curve" in gpu.zk_last_error()


@pytest.mark.parametrize("name,cid", CURVES)
def test_key_sized_round_trip(gpu, name, cid):
    """2^16 distinct points (k * G for random k through the fixed-base batch multiplication) through PointArray"""
    cv = pyref.curve_by_name(name)
    n = 1 << 16
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= (1 << 60) - 1
    for grp in (1, 2):
        gen = np.zeros(N.point_limbs(cid, grp), dtype=np.uint64)
        N.check(gpu.zk_point_generator(cid, grp, N.u64p(gen)))
        pts = np.zeros((n, N.point_limbs(cid, grp)), dtype=np.uint64)
        N.check(gpu.zk_batch_mul(cid, grp, n, N.u64p(sc), N.u64p(gen), 1, N.u64p(pts)))
        arr = PointArray(cid, grp, pts)
        raw = arr.to_bytes()
        assert len(raw) == n * gpu.zk_point_bytes(cid, grp)
        back = PointArray.from_compressed(cid, grp, raw, n)
        assert (back.limbs == pts).all()
        # spot-check against the oracle's encoding
        for i in (0, 1, n - 1):
            p = corc.limbs_to_points(pts[i:i + 1], cid, grp)[0]
            nb = gpu.zk_point_bytes(cid, grp)
            assert raw[i * nb:(i + 1) * nb] == pyref.compress(cv, grp, p)
