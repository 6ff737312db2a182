"""GPU parity: NTT / iNTT / coset / vector ops / vanishing division / fused QAP pipeline through the C ABI,
against the CPU oracle on the same seeded inputs (bit-exact), plus size-independent properties at 2^22."""

import numpy as np
import pytest

from helpers import CURVES, rand_limbs, rand_scalars
from oracle import corc, pyref
from zksnake_amd import _native as N

pytestmark = pytest.mark.gpu


def _ntt(lib, cid, a, size, inverse=0, coset=0):
    n = 1 if size <= 1 else 1 << (size - 1).bit_length()
    out = np.zeros((n, 4), dtype=np.uint64)
    N.check(lib.zk_ntt(cid, inverse, coset, a.shape[0], N.u64p(a), size, N.u64p(out)))
    return out


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 12, 13, 14, 15, 16])
def test_ntt_matches_oracle(gpu, name, cid, log_n):
    cv = pyref.curve_by_name(name)
    n = 1 << log_n
    _, a = rand_scalars(n, cv.r, 100 + log_n)
    assert (_ntt(gpu, cid, a, n) == corc.ntt(cid, a)).all()
    assert (_ntt(gpu, cid, a, n, inverse=1) == corc.ntt(cid, a, inverse=True)).all()


@pytest.mark.parametrize("name,cid", CURVES)
def test_ntt_kat_and_definition(gpu, name, cid):
    cv = pyref.curve_by_name(name)
    # SURVEY Appendix A known-answer vector
    got = N.limbs_to_ints(_ntt(gpu, cid, N.ints_to_limbs([1, 2, 3, 4]), 4))
    assert got == pyref.ntt([1, 2, 3, 4], 4, cv)
    vals, a = rand_scalars(64, cv.r, 5)
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 64)) == pyref.dft_naive(vals, 64, cv)
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 64, inverse=1)) == pyref.dft_naive(vals, 64, cv, inverse=True)


@pytest.mark.parametrize("name,cid", CURVES)
def test_ntt_padding_truncation_reduction(gpu, name, cid):
    cv = pyref.curve_by_name(name)
    vals, a = rand_scalars(5, cv.r, 6)
    # shorter input is zero-padded, size rounded up to a power of two
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 6)) == pyref.ntt(vals, 8, cv)
    # longer input is truncated to the domain (ark-poly's fft_in_place resizes the vector; parity unpinned)
    vals, a = rand_scalars(11, cv.r, 7)
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 4)) == pyref.ntt(vals, 4, cv) == pyref.ntt(vals[:4], 4, cv)
    # values >= r are reduced on entry (Fr::from(BigUint))
    big = [cv.r + 5, 2 * cv.r + 1, (1 << 256) - 1, 0]
    assert N.limbs_to_ints(_ntt(gpu, cid, N.ints_to_limbs(big), 4)) == pyref.ntt([b % cv.r for b in big], 4, cv)
    # empty input
    assert N.limbs_to_ints(_ntt(gpu, cid, np.zeros((0, 4), dtype=np.uint64), 4)) == [0, 0, 0, 0]


@pytest.mark.parametrize("name,cid", CURVES)
def test_coset_ntt(gpu, name, cid):
    cv = pyref.curve_by_name(name)
    vals, a = rand_scalars(32, cv.r, 8)
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 32, coset=1)) == pyref.coset_ntt(vals, 32, cv)
    assert N.limbs_to_ints(_ntt(gpu, cid, a, 32, inverse=1, coset=1)) == pyref.coset_ntt(vals, 32, cv, inverse=True)


def test_domain_too_large(gpu):
    a = np.zeros((1, 4), dtype=np.uint64)
    out = np.zeros((1, 4), dtype=np.uint64)
    assert gpu.zk_ntt(0, 0, 0, 1, N.u64p(a), 1 << 29, N.u64p(out)) == N.ZK_ERR_DOMAIN  # BN254 two-adicity 28
    assert b"too large" in gpu.zk_last_error()


@pytest.mark.parametrize("name,cid", CURVES)
def test_vec_ops(gpu, name, cid):
    cv = pyref.curve_by_name(name)
    va, a = rand_scalars(777, cv.r, 9)
    vb, b = rand_scalars(500, cv.r, 10)
    for op, fn in ((0, lambda x, y: x * y), (1, lambda x, y: x + y), (2, lambda x, y: x - y)):
        out = np.zeros((777, 4), dtype=np.uint64)
        N.check(gpu.zk_vec_op(cid, op, 777, 777, N.u64p(a), 500, N.u64p(b), N.u64p(out)))
        exp = [fn(x, vb[i] if i < 500 else 0) % cv.r for i, x in enumerate(va)]
        assert N.limbs_to_ints(out) == exp


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("n,ln", [(8, 5), (8, 8), (8, 13), (8, 16), (8, 27), (16, 31)])
def test_divide_by_vanishing(gpu, name, cid, n, ln):
    cv = pyref.curve_by_name(name)
    vals, a = rand_scalars(ln, cv.r, 11 + ln)
    q = np.zeros((max(ln - n, 1), 4), dtype=np.uint64)
    rem = np.zeros((min(ln, n), 4), dtype=np.uint64)
    flag = N._i(0)
    N.check(gpu.zk_poly_div_vanishing(cid, n, ln, N.u64p(a), N.u64p(q), N.u64p(rem), flag))
    eq, er = pyref.divide_by_vanishing(vals, n, cv.r)
    assert pyref.strip_zeros(N.limbs_to_ints(q[: max(ln - n, 0)])) == eq
    assert pyref.strip_zeros(N.limbs_to_ints(rem)) == er
    assert flag.value == (0 if er else 1)


def _ntt_dev_roundtrip(gpu, cid, a, log_n):
    """forward and inverse transform of `a` on the device: (forward, inverse-of-forward)"""
    import ctypes
    d = ctypes.c_void_p()
    N.check(gpu.zk_dev_alloc(a.nbytes, ctypes.byref(d)))
    try:
        N.check(gpu.zk_dev_upload(d, a.ctypes.data, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 0, log_n, d, None))
        fwd = np.empty_like(a)
        N.check(gpu.zk_dev_download(fwd.ctypes.data, d, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 1, log_n, d, None))
        back = np.empty_like(a)
        N.check(gpu.zk_dev_download(back.ctypes.data, d, a.nbytes))
    finally:
        gpu.zk_dev_free(d)
    return fwd, back


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("log_n", [17, 18, 19, 20, 21, 22])
def test_ntt_three_pass_sizes_match_oracle_elementwise(gpu, name, cid, log_n):
    """the 3-pass plan (2^17 .. 2^24: BASELINE config 3 and every full-size proof) element by element against the CPU
    oracle, forward and inverse (round-2 verdict: only spot values were checked above 2^16); every size up to 2^22, because
    the stages per pass differ from size to size (6+5+6, 6+6+6, 7+6+6, 8+6+6, 8+6+7, 8+6+8)"""
    n = 1 << log_n
    a = rand_limbs(n, 300 + log_n)   # below 2^252 < r for both fields
    fwd, back = _ntt_dev_roundtrip(gpu, cid, a, log_n)
    assert (fwd == corc.ntt(cid, a, threads=8)).all()
    assert (back == a).all()
    # the inverse transform of fresh data, on its own
    import ctypes
    d = ctypes.c_void_p()
    N.check(gpu.zk_dev_alloc(a.nbytes, ctypes.byref(d)))
    try:
        N.check(gpu.zk_dev_upload(d, a.ctypes.data, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 1, log_n, d, None))
        inv = np.empty_like(a)
        N.check(gpu.zk_dev_download(inv.ctypes.data, d, a.nbytes))
    finally:
        gpu.zk_dev_free(d)
    assert (inv == corc.ntt(cid, a, inverse=True, threads=8)).all()


def test_ntt_four_pass_size_2_25(gpu):
    """2^25 is the first size with a 4-pass plan (ntt_dev_impl: ceil(25 / 8) passes): element-wise against the CPU oracle,
    round trip, and the two spot values that are plain sums"""
    cid, cv, log_n = 0, pyref.BN254, 25
    n = 1 << log_n
    a = rand_limbs(n, 25)
    fwd, back = _ntt_dev_roundtrip(gpu, cid, a, log_n)
    assert (back == a).all()
    assert (fwd == corc.ntt(cid, a, threads=16)).all()
    # out[0] = sum a_j, out[n/2] = alternating sum: column sums of the 64-bit limbs as Python integers
    def total(x):
        return sum(int(x[:, k].astype(object).sum()) << (64 * k) for k in range(4))
    r = cv.r
    assert N.limbs_to_ints(fwd[0:1])[0] == total(a) % r
    assert N.limbs_to_ints(fwd[n // 2:n // 2 + 1])[0] == (total(a[0::2]) - total(a[1::2])) % r


def test_ntt_largest_bn254_domain_2_28(gpu):
    """2^28 is the largest domain BN254's Fr has (two-adicity 28; src/bn254/polynomial.rs:520,541 refuse anything above): four
    passes of seven stages over 8 GiB, offsets beyond 32 bits everywhere.  The full CPU transform would take minutes, so single
    outputs are checked against the definition (corc.ntt_spot, O(n) each; pinned against the full transform on the CPU),
    forward and inverse, plus the round trip of the whole vector"""
    import ctypes
    cid, log_n = 0, 28
    n = 1 << log_n
    a = np.random.default_rng(28).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] >>= 3   # below 2^252 < r
    d = ctypes.c_void_p()
    N.check(gpu.zk_dev_alloc(a.nbytes, ctypes.byref(d)))
    try:
        N.check(gpu.zk_dev_upload(d, a.ctypes.data, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 0, log_n, d, None))
        fwd = np.empty_like(a)
        N.check(gpu.zk_dev_download(fwd.ctypes.data, d, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 1, log_n, d, None))
        back = np.empty_like(a)
        N.check(gpu.zk_dev_download(back.ctypes.data, d, a.nbytes))
        assert np.array_equal(back, a)
        del back
        # the inverse transform of fresh data: a is read as a vector of evaluations
        N.check(gpu.zk_dev_upload(d, a.ctypes.data, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 1, log_n, d, None))
        inv = np.empty_like(a)
        N.check(gpu.zk_dev_download(inv.ctypes.data, d, a.nbytes))
    finally:
        gpu.zk_dev_free(d)
    # indices that touch every digit of the four passes: ends, halves, a tile boundary of each pass, arbitrary ones
    for k in (0, 1, n // 2, n - 1, (1 << 21) + 1, (1 << 14) - 1, 0x9E3779B, 0x5A5A5A5):
        assert (fwd[k] == corc.ntt_spot(cid, a, k, threads=16)).all(), k
    for k in (0, n - 1, 0x3C6EF37, 0xDEADBEE):
        assert (inv[k] == corc.ntt_spot(cid, a, k, inverse=True, threads=16)).all(), k


def test_ntt_2_22_properties(gpu):
    """BASELINE config 3 size: round trip, linearity, and spot evaluation against Horner"""
    import ctypes
    cid, cv, log_n = 0, pyref.BN254, 22
    n = 1 << log_n
    a = rand_limbs(n, 22)
    d = ctypes.c_void_p()
    N.check(gpu.zk_dev_alloc(n * 32, ctypes.byref(d)))
    try:
        N.check(gpu.zk_dev_upload(d, a.ctypes.data, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 0, log_n, d, None))
        fwd = np.empty_like(a)
        N.check(gpu.zk_dev_download(fwd.ctypes.data, d, a.nbytes))
        N.check(gpu.zk_ntt_dev(cid, 1, log_n, d, None))
        back = np.empty_like(a)
        N.check(gpu.zk_dev_download(back.ctypes.data, d, a.nbytes))
    finally:
        gpu.zk_dev_free(d)
    assert (back == a).all()
    # spot values: out[k] = sum_j a_j w^(jk); k = 0 is the plain sum, k = n/2 the alternating sum
    ints = N.limbs_to_ints(a)
    r = cv.r
    assert N.limbs_to_ints(fwd[0:1])[0] == sum(ints) % r
    assert N.limbs_to_ints(fwd[n // 2:n // 2 + 1])[0] == (sum(ints[0::2]) - sum(ints[1::2])) % r
    k = 123457
    assert N.limbs_to_ints(fwd[k:k + 1])[0] == pyref.poly_eval(ints, pow(cv.root_of_unity(n), k, r), r)


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("n", [2, 16, 1024])
def test_qap_pipeline_matches_oracle(gpu, name, cid, n):
    from zksnake_amd.device import DeviceBuffer
    cv = pyref.curve_by_name(name)
    A, B, C, n_row, n_col, n_pub, w = pyref.chain_circuit(n, cv.r)
    a = N.ints_to_limbs(pyref.sparse_dot(A, n_row, w, cv.r))
    b = N.ints_to_limbs(pyref.sparse_dot(B, n_row, w, cv.r))
    c = N.ints_to_limbs(pyref.sparse_dot(C, n_row, w, cv.r))
    eu, ev, eh = corc.qap_h(cid, a, b, c)
    da, db, dc = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b), DeviceBuffer.from_numpy(c)
    dh, dw = DeviceBuffer(n * 32), DeviceBuffer(4 * n * 32)
    ok = N._i(0)
    N.check(gpu.zk_qap_h_dev(cid, n.bit_length() - 1, da.ptr, db.ptr, dc.ptr, dh.ptr, dw.ptr, ok, None))
    assert ok.value == 1
    assert (da.download((n, 4)) == eu).all() and (db.download((n, 4)) == ev).all() and (dh.download((n, 4)) == eh).all()
    # a wrong witness leaves a remainder
    bad = a.copy()
    bad[0, 0] ^= np.uint64(1)
    da.upload(bad); db.upload(b)
    N.check(gpu.zk_qap_h_dev(cid, n.bit_length() - 1, da.ptr, db.ptr, dc.ptr, dh.ptr, dw.ptr, ok, None))
    assert ok.value == 0


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("n", [8, 4096, 1 << 13])
def test_qap_uv_only_and_two_step_form(gpu, name, cid, n):
    """zk_qap_uv_dev (what a rank of the task-partitioned prover runs when its MSM reads u or v only) gives the u and v of the
    whole chain, for either vector alone as well; zk_qap_h_dev_begin / _end (the form prove() uses, with the fused passes of
    round 4: coset shift on the inverse transforms' last pass, the quotient formed by the last transform's first pass) agree
    with the one-call form and the oracle, in single-pass (n <= 2048) and multi-pass transforms"""
    from zksnake_amd.device import DeviceBuffer
    cv = pyref.curve_by_name(name)
    A, B, C, n_row, n_col, n_pub, w = pyref.chain_circuit(n, cv.r)
    a = N.ints_to_limbs(pyref.sparse_dot(A, n_row, w, cv.r))
    b = N.ints_to_limbs(pyref.sparse_dot(B, n_row, w, cv.r))
    c = N.ints_to_limbs(pyref.sparse_dot(C, n_row, w, cv.r))
    eu, ev, eh = corc.qap_h(cid, a, b, c)
    log_n = n.bit_length() - 1
    da, db = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b)
    ev_ptr = N._vp()
    N.check(gpu.zk_qap_uv_dev(cid, log_n, da.ptr, db.ptr, None, N.ctypes.byref(ev_ptr)))
    gpu.zk_dev_synchronize()
    assert ev_ptr.value and (da.download((n, 4)) == eu).all() and (db.download((n, 4)) == ev).all()
    da.upload(a); db.upload(b)
    N.check(gpu.zk_qap_uv_dev(cid, log_n, da.ptr, None, None, None))      # u alone, no event asked for
    N.check(gpu.zk_qap_uv_dev(cid, log_n, None, db.ptr, None, None))      # v alone
    gpu.zk_dev_synchronize()
    assert (da.download((n, 4)) == eu).all() and (db.download((n, 4)) == ev).all()
    # two-step form on a stream of its own; c must come through untouched (w is transformed straight out of it)
    st = N._vp()
    N.check(gpu.zk_stream_create(0, N.ctypes.byref(st)))
    da.upload(a); db.upload(b)
    dc, dh, dw = DeviceBuffer.from_numpy(c), DeviceBuffer(n * 32), DeviceBuffer(4 * n * 32)
    N.check(gpu.zk_qap_h_dev_begin(cid, log_n, da.ptr, db.ptr, dc.ptr, dh.ptr, dw.ptr, st, N.ctypes.byref(ev_ptr)))
    ok = N._i(0)
    N.check(gpu.zk_qap_h_dev_end(cid, log_n, dw.ptr, ok, st))
    assert ok.value == 1
    assert (da.download((n, 4)) == eu).all() and (db.download((n, 4)) == ev).all() and (dh.download((n, 4)) == eh).all()
    assert (dc.download((n, 4)) == c).all()
    N.check(gpu.zk_stream_destroy(st))


@pytest.mark.parametrize("name,cid", CURVES)
def test_spmv(gpu, name, cid):
    from zksnake_amd.array import SparseArray
    from zksnake_amd.device import DeviceBuffer
    import random
    cv = pyref.curve_by_name(name)
    rnd = random.Random(3)
    n_row, n_col = 37, 23
    trip = [(rnd.randrange(n_row), rnd.randrange(n_col), rnd.randrange(cv.r)) for _ in range(200)]
    m = SparseArray.from_triplets(*zip(*trip), n_row, n_col, cv.r)
    w = [rnd.randrange(cv.r) for _ in range(n_col)]
    rp, cl, vl = m.to_csr()
    bufs = [DeviceBuffer.from_numpy(x) for x in (rp, cl, vl, N.ints_to_limbs(w))]
    out = DeviceBuffer(n_row * 32)
    N.check(gpu.zk_spmv_dev(cid, n_row, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, out.ptr, 0, None))
    assert N.limbs_to_ints(out.download((n_row, 4))) == m.dot(w) == pyref.sparse_dot(trip, n_row, w, cv.r)


@pytest.mark.parametrize("name,cid", CURVES)
def test_spmv_long_rows(gpu, name, cid):
    """rows far beyond the lane-per-row limit (the constant-one wire / input wires of a transposed R1CS): one row with
    10000 entries (three work items), one with 65 (just over the limit), one with 64 (just under), empty rows, against
    SparseArray.dot; then the transpose used by Groth16.setup"""
    import random
    from zksnake_amd.array import SparseArray
    from zksnake_amd.device import DeviceBuffer
    from zksnake_amd.spmv import ITEM, LONG_ROW, DeviceCsr
    cv = pyref.curve_by_name(name)
    rnd = random.Random(11)
    n_row, n_col = 9, 12000
    trip = [(0, c, rnd.randrange(cv.r)) for c in range(10000)]
    trip += [(3, rnd.randrange(n_col), rnd.randrange(cv.r)) for _ in range(LONG_ROW + 1)]
    trip += [(5, rnd.randrange(n_col), rnd.randrange(cv.r)) for _ in range(LONG_ROW)]
    trip += [(7, 1, cv.r - 1), (7, 0, 5)]
    rnd.shuffle(trip)
    m = SparseArray.from_triplets(*zip(*trip), n_row, n_col, cv.r)
    w = [rnd.randrange(cv.r) for _ in range(n_col)]
    csr = DeviceCsr(cid, *m.to_csr())
    assert csr.n_long == 2 and csr.n_items == -(-10000 // ITEM) + 1
    dw, out = DeviceBuffer.from_numpy(N.ints_to_limbs(w)), DeviceBuffer(n_row * 32)
    out.upload(np.full((n_row, 4), 7, dtype=np.uint64))            # stale data must not survive in empty rows
    csr.apply(dw.ptr, out.ptr)
    assert N.limbs_to_ints(out.download((n_row, 4))) == m.dot(w)
    # transpose: 12000 rows of 0..2 entries each
    x = [rnd.randrange(cv.r) for _ in range(n_row)]
    csc = DeviceCsr(cid, *m.to_csc())
    dx, out_t = DeviceBuffer.from_numpy(N.ints_to_limbs(x)), DeviceBuffer(n_col * 32)
    csc.apply(dx.ptr, out_t.ptr)
    exp = [0] * n_col
    for r, c, v in trip:
        exp[c] = (exp[c] + v * x[r]) % cv.r
    assert N.limbs_to_ints(out_t.download((n_col, 4))) == exp


def test_spmv_of_an_empty_matrix_clears_the_output_on_the_given_stream(gpu):
    """a matrix without entries: the output is cleared with a memset ordered on the caller's (non-blocking) stream, not
    on the legacy default stream (round-2 advisor finding)"""
    import ctypes
    from zksnake_amd.device import DeviceBuffer
    from zksnake_amd.spmv import DeviceCsr
    n_row = 1 << 16
    csr = DeviceCsr(0, np.zeros(n_row + 1, dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=np.uint64))
    st = ctypes.c_void_p()
    N.check(gpu.zk_stream_create(0, ctypes.byref(st)))
    try:
        out = DeviceBuffer.from_numpy(np.full((n_row, 4), 9, dtype=np.uint64))
        dw = DeviceBuffer.from_numpy(np.ones((4, 4), dtype=np.uint64))
        N.check(gpu.zk_debug_spin_dev(st, 2000))     # the clear must queue up behind work already on the stream
        csr.apply(dw.ptr, out.ptr, st)
        N.check(gpu.zk_stream_synchronize(st))
        assert not out.download((n_row, 4)).any()
    finally:
        N.check(gpu.zk_stream_destroy(st))
