"""
QAP witness evaluation (reference python/zksnake/groth16/qap.py:11-71) on the GPU.

The reference computes, through ~25 Python<->Rust crossings,
    a, b, c = A.w, B.w, C.w ; u, v, w = ifft(a), ifft(b), ifft(c)
    uv = ifft(fft(u) * fft(v)) over the doubled domain ; h = (uv - w) / (X^n - 1), remainder must be 0.
Here the witness goes up once, three CSR mat-vecs (zk_spmv_dev) produce a, b, c in HBM, and
zk_qap_h_dev runs the whole transform chain without leaving the device.  `evaluate_witness`
keeps the reference's return shape (four Polynomial objects); the prover uses the device-resident form.
"""

import numpy as np

from .. import _native as N
from ..constant import BN254_SCALAR_FIELD
from ..device import DeviceBuffer, PinnedArray
from ..polynomial import POLY_OBJECT
from ..utils import next_power_of_two


class DeviceQapResult:
    """u, v, h coefficient vectors (n canonical Fr elements each) living in HBM"""

    def __init__(self, n, u, v, h, witness):
        self.n = n
        self.u = u
        self.v = v
        self.h = h
        self.witness = witness  # full witness vector on the device (n_col elements)


class QAP:
    def __init__(self, p=None):
        self.a = []
        self.b = []
        self.c = []
        self.n_public = 0
        self.p = p or BN254_SCALAR_FIELD
        self._dev = None
        self._ws = None
        self._stream = None

    def from_r1cs(self, r1cs):
        assert r1cs.A is not None, "R1CS is not compiled"
        self.n_public = r1cs.n_public
        n = next_power_of_two(r1cs.A.n_row)
        self.a, self.b, self.c = r1cs.A, r1cs.B, r1cs.C
        self.a.n_row = self.b.n_row = self.c.n_row = n
        for m in (self.a, self.b, self.c):
            m._csr = None
            m._csc = None
        self._dev = None

    # ---- device-resident evaluation -----------------------------------------------------------
    def _curve_id(self):
        return POLY_OBJECT[self.p].curve_id

    def _device_matrices(self, needs=None):
        """the CSR matrices A, B, C in HBM (built on first use).  needs: the outputs the caller will ask for (a subset of
        {"u", "v", "h"}; None = all): a rank that only ever needs u uploads A alone."""
        if self._dev is None:
            self._dev = [None, None, None]
        want = (True, True, True) if needs is None or "h" in needs else ("u" in needs, "v" in needs, False)
        from ..spmv import DeviceCsr
        for k, m in enumerate((self.a, self.b, self.c)):
            if want[k] and self._dev[k] is None:
                self._dev[k] = DeviceCsr(self._curve_id(), *m.to_csr())
        return self._dev

    def _workspace(self, n, n_col):
        """device buffers reused across proofs (hipMalloc / hipFree are synchronous and slow)"""
        key = (n, n_col)
        if self._ws is None or self._ws[0] != key:
            if self._ws is not None:
                self._ws[1]["w_host"].free()   # page-locked staging of the previous size
            eb = 32
            self._ws = (key, dict(w=DeviceBuffer(n_col * eb), a=DeviceBuffer(n * eb), b=DeviceBuffer(n * eb),
                                  c=DeviceBuffer(n * eb), h=DeviceBuffer(n * eb), work=DeviceBuffer(4 * n * eb),
                                  w_host=PinnedArray((n_col, 4))))
        return self._ws[1]

    def _qap_stream(self):
        """the QAP chain runs on a stream of its own (non-blocking w.r.t. the default stream), so that an MSM that
        only needs the witness can run beside it on its plan's stream"""
        if self._stream is None:
            st = N._vp()
            N.check(N.ensure_gpu().zk_stream_create(0, N.ctypes.byref(st)))
            self._stream = st
        return self._stream

    def evaluate_witness_device(self, witness, after_upload=None, after_uv=None, needs=None) -> DeviceQapResult:
        """witness: list of ints or (n_col, 4) uint64 limbs.  Raises ValueError when the witness does
        not satisfy the constraints (non-zero remainder), like the reference.  The returned buffers belong
        to this QAP object and are overwritten by the next call.  `after_upload(witness_buffer)` is called once the
        canonical witness is resident in HBM, before the transform chain is enqueued.  `after_uv(event, u, v)` is called
        with the chain in flight: `event` fires when the coefficient vectors u and v (device buffers) are final, a third
        of the way into the chain, so that work that needs only them can be queued behind it.
        needs (a rank of a task-partitioned prover, Groth16.shard_over_ranks): the outputs that will be read, a subset of
        {"u", "v", "h"}; None or anything with "h" runs the whole chain.  Without "h" only the sparse products and inverse
        transforms behind u and / or v run (the other fields of the result are None) and the witness is NOT checked against the
        constraints here -- the ranks that hold h do that."""
        lib = N.ensure_gpu()
        cid = self._curve_id()
        n = self.a.n_row
        if n < 2:
            raise ValueError("the QAP needs at least 2 rows")
        log_n = n.bit_length() - 1
        # the witness may come as ints, as one limb array, or as (public, private) limb arrays (uploaded piecewise:
        # concatenating 2^20 x 32 B on the host would cost more than the upload itself)
        parts = list(witness) if isinstance(witness, tuple) else [witness]
        parts = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) if isinstance(p, np.ndarray) else p for p in parts]
        if sum(len(p) for p in parts) != self.a.n_col:
            raise ValueError("witness length does not match the number of R1CS columns")
        ws = self._workspace(n, self.a.n_col)
        st = self._qap_stream()
        row = 0
        staged_async = False
        try:
            for part in parts:
                if len(part) and not isinstance(part, np.ndarray):
                    # ints (the reference's call shape) are repacked into the page-locked staging rows, reduced mod r, and every
                    # finished chunk goes up on the QAP stream while the worker threads convert the next one
                    stage = ws["w_host"].array[row:row + len(part)]
                    base = ws["w"].ptr + 32 * row

                    def ship(b, e, stage=stage, base=base):
                        N.check(lib.zk_dev_upload_async(base + 32 * b, stage[b:e].ctypes.data, 32 * (e - b), st))

                    staged_async = True   # before the conversion: a chunk may have been shipped when a later one raises
                    N.ints_to_limbs(part, 4, self.p, out=stage, chunk_done=ship)
                elif len(part):
                    ws["w"].upload(part, offset=32 * row)
                row += len(part)
        finally:
            # also when the conversion fails half way (a negative int in a later chunk): copies out of the shared staging buffer
            # may still be in flight, and the next prove() rewrites that buffer (round-3 advisor finding)
            if staged_async:
                N.check(lib.zk_stream_synchronize(st))   # the witness is resident before anything is queued against it
        if any(isinstance(p, np.ndarray) for p in (witness if isinstance(witness, tuple) else (witness,))):
            # int lists were reduced mod r on the host (Fr::from); limb arrays are reduced here, one cheap pass
            N.check(lib.zk_vec_canon_dev(cid, self.a.n_col, ws["w"].ptr, st))
            N.check(lib.zk_stream_synchronize(st))
        if after_upload is not None:
            after_upload(ws["w"])
        if needs is not None and "h" not in needs:
            mats = self._device_matrices(needs)
            d_u = ws["a"] if "u" in needs else None
            d_v = ws["b"] if "v" in needs else None
            if d_u is not None:
                mats[0].apply(ws["w"].ptr, d_u.ptr, st)
            if d_v is not None:
                mats[1].apply(ws["w"].ptr, d_v.ptr, st)
            if d_u is not None or d_v is not None:
                event = N._vp()
                N.check(lib.zk_qap_uv_dev(cid, log_n, d_u.ptr if d_u else None, d_v.ptr if d_v else None, st, N.ctypes.byref(event)))
                if after_uv is not None:
                    after_uv(event, d_u, d_v)
                N.check(lib.zk_stream_synchronize(st))
            return DeviceQapResult(n, d_u, d_v, None, ws["w"])
        for mat, dst in zip(self._device_matrices(), (ws["a"], ws["b"], ws["c"])):
            mat.apply(ws["w"].ptr, dst.ptr, st)
        ok = N._i(0)
        if after_uv is None:
            N.check(lib.zk_qap_h_dev(cid, log_n, ws["a"].ptr, ws["b"].ptr, ws["c"].ptr, ws["h"].ptr, ws["work"].ptr, ok, st))
        else:
            event = N._vp()
            N.check(lib.zk_qap_h_dev_begin(cid, log_n, ws["a"].ptr, ws["b"].ptr, ws["c"].ptr, ws["h"].ptr, ws["work"].ptr, st,
                                           N.ctypes.byref(event)))
            after_uv(event, ws["a"], ws["b"])
            N.check(lib.zk_qap_h_dev_end(cid, log_n, ws["work"].ptr, ok, st))
        if not ok.value:
            raise ValueError("(U * V - W) did not divided by Z to zero")
        return DeviceQapResult(n, ws["a"], ws["b"], ws["h"], ws["w"])

    # ---- reference-shaped API ---------------------------------------------------------------------
    def evaluate_witness(self, witness: list):
        """returns the polynomials (U, V, W, H) like the reference (W is recomputed from C.w)."""
        mod = POLY_OBJECT[self.p]
        res = self.evaluate_witness_device(witness)
        n = res.n
        u = N.limbs_to_ints(res.u.download((n, 4)))
        v = N.limbs_to_ints(res.v.download((n, 4)))
        h = N.limbs_to_ints(res.h.download((n, 4)))
        w_ints = witness if not isinstance(witness, np.ndarray) else N.limbs_to_ints(witness)
        w = mod.ifft(self.c.dot(w_ints), n)
        mk = lambda coeffs: mod.Polynomial(1, [(c, ()) for c in coeffs], n)  # noqa: E731
        return mk(u), mk(v), mk(w), mk(h)
