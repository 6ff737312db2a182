"""
Groth16 setup / prove / verify with the reference's class surface
(python/zksnake/groth16/protocol.py:14-186).  prove() keeps everything between the witness upload
and the five MSM results on the GPU:

    witness -> [SpMV x3] -> a, b, c -> [iNTT x3, NTT(2n) x2, mul, iNTT(2n), fold] -> u, v, h   (qap.py)
    A  = <tau_1, u> + alpha_1 + r delta_1          B1 = <tau_1, v> + beta_1 + s delta_1
    B2 = <tau_2, v> + beta_2 + s delta_2           C  = <target_1, h> + <kdelta_1, w_priv> + s A + r B1 - r s delta_1

The proving key vectors are `PointArray`s whose Montgomery-form bases stay resident in HBM
(MSM plans), so a prove moves one witness up and three points down.

Multi-GPU (BASELINE config 5): after `shard_over_ranks()` the five MSMs -- the reference's five independent
`multiexp` calls, protocol.py:133-155 -- are laid on one line of (task, window) units that is cut into one contiguous piece
per rank (zksnake_amd/parallel.py: partition_proof): a rank runs one MSM, or a window range of the G2 one, and evaluates only
the part of the QAP its scalars need (nothing for <kdelta_1, w>, one sparse product + one inverse transform for u or v, the
whole chain only for h).  The partial points of all ranks and an ok flag travel in ONE all_gather (RCCL) and every rank
assembles the same proof.  `partition="window"` keeps the layout of rounds 1-3 (every rank a window range of every MSM and
the whole QAP) for comparison.
"""

import os

import numpy as np

from .. import _native as N
from ..arithmetization.r1cs import R1CS
from ..device import DeviceBuffer
from ..ecc import EllipticCurve, PointArray
from ..frvec import DevVec, FrOps
from ..polynomial import POLY_OBJECT
from ..spmv import DeviceCsr
from ..utils import get_random_int
from .qap import QAP
from .serialization import Proof, ProvingKey, VerifyingKey


def _as_array(E, pts, group):
    """lists coming from ProvingKey.from_bytes become device-backed arrays on first use"""
    if isinstance(pts, PointArray):
        return pts
    from .._algebra import _points_to_limbs
    cid = E.curve.curve_id
    return PointArray(cid, group, _points_to_limbs(pts, cid, group))


def _ptr(buf):
    """device address of a QAP output, None when this rank did not compute it (its MSM then has no window here either)"""
    return buf.ptr if buf is not None else None


class Groth16:
    def __init__(self, r1cs: R1CS, curve: str = "BN254"):
        self.E = EllipticCurve(curve)
        self.order = self.E.order
        self.qap = QAP(self.order)
        self.qap.from_r1cs(r1cs)
        self.proving_key = None
        self.verifying_key = None
        self._toxic = None      # tests may pin (tau, alpha, beta, gamma, delta)
        self._blinding = None   # tests may pin (r, s)
        self.precompute_keys = True  # proving-key MSM plans use the fixed-base table (ZK_MSM_PRECOMPUTE)
        self.last_timings = {}
        self._collective_ms = 0.0
        self._live = []         # MSM plan handles with a run in flight (prove() cancels them when it fails half way)
        self._shard = None      # (rank, world, torch device or None) once shard_over_ranks() was called
        self._partition = "task"
        self._assignment = None  # per rank {task: (first window, count)}, computed when the key sizes are known
        self._n_windows = {}     # task -> window count of its MSM in the sharded layout
        self._task_bits = {}     # task -> explicit window width of its plans (empty: the library's choice)
        self.projected_ms = None  # the partition's cost model, per rank (parallel.partition_proof)
        self._exchanged = True   # False between the first enqueue and the collective of a sharded prove()

    # ------------------------------------------------------------------------------------------
    def setup(self, prepare_prover=True):
        """trusted setup: ProvingKey and VerifyingKey from fresh (or pinned) toxic waste.
        prepare_prover (not in the reference's signature): also build what the first prove() would otherwise build -- the
        device-resident fixed-base plans of the four key vectors and the CSR matrices -- on a worker thread, beside the
        host-side matrix work of the setup, so that the first proof costs what every later one does."""
        q = self.order
        G1, G2 = self.E.G1(), self.E.G2()
        if self._toxic is not None:
            tau, alpha, beta, gamma, delta = self._toxic
        else:
            tau, alpha, beta, gamma, delta = (get_random_int(q - 1) for _ in range(5))
        inv_gamma, inv_delta = pow(gamma, -1, q), pow(delta, -1, q)

        n = self.qap.a.n_row
        n_wires = self.qap.a.n_col
        n_pub = self.qap.n_public
        mod = POLY_OBJECT[q]
        V = FrOps(q)
        lib, cid = N.ensure_gpu(), self.E.curve.curve_id
        # Lagrange basis at tau (host C++, one inversion), then everything per-wire on the GPU:
        # L = A^T lag, R = B^T lag, O = C^T lag (transposed CSR SpMV), K = beta L + alpha R + O
        # L_i(tau) = (1/n) sum_j tau^j w^(-ij): the inverse transform of the powers of tau (one 0.2 ms iNTT instead of
        # n sequential host products and an inversion, evaluate_lagrange_coefficients of polynomial.rs:645-652)
        d_pow = V.d_powers(tau, n)
        powers = d_pow.download()
        d_lag = DevVec(n, zero=False)
        V.d_copy(n, d_pow.ptr(), d_lag.ptr())
        V.d_ntt(d_lag, n, inverse=True)

        # the key vectors that depend on the toxic waste only come first: their fixed-base plans are then built on a worker
        # thread (GPU: chains of doublings; the calls release the GIL) while this thread transposes the matrices (numpy)
        t = mod.evaluate_vanishing_polynomial(n, tau)
        d_shift = DevVec(n, zero=False)
        N.check(lib.zk_vec_axpby_dev(cid, n, N.u64p(V.one(t * inv_delta % q)), d_pow.ptr(), None, None, None, d_shift.ptr(), None))
        shifted = d_shift.download()
        tau_G1 = self.E.batch_mul(G1, powers, as_array=True)
        tau_G2 = self.E.batch_mul(G2, powers, as_array=True)
        target_G1 = self.E.batch_mul(G1, shifted, as_array=True)
        worker = self._start_plan_worker([(tau_G1, "u"), (tau_G2, "v2"), (tau_G1, "v1"), (target_G1, "h")]) if prepare_prover else None

        sums = []
        for mat in (self.qap.a, self.qap.b, self.qap.c):
            out = DevVec(n_wires)
            csc = DeviceCsr(cid, *mat.to_csc())   # the transpose: one row per wire, the input wires' rows are long
            csc.apply(d_lag.ptr(), out.ptr(), None)
            lib.zk_dev_synchronize()
            sums.append(out)
        L, R, O = sums
        K = DevVec(n_wires, zero=False)
        N.check(lib.zk_vec_axpby_dev(cid, n_wires, N.u64p(V.one(beta)), L.ptr(), N.u64p(V.one(alpha)), R.ptr(), None, K.ptr(), None))
        V.d_add(n_wires, K.ptr(), O.ptr(), K.ptr())
        k_gamma, k_delta = DevVec(max(n_pub, 1)), DevVec(max(n_wires - n_pub, 1))
        if n_pub:
            N.check(lib.zk_vec_axpby_dev(cid, n_pub, N.u64p(V.one(inv_gamma)), K.ptr(), None, None, None, k_gamma.ptr(), None))
        if n_wires > n_pub:
            N.check(lib.zk_vec_axpby_dev(cid, n_wires - n_pub, N.u64p(V.one(inv_delta)), K.ptr(n_pub), None, None, None, k_delta.ptr(), None))

        k_gamma_G1 = self.E.batch_mul(G1, k_gamma.download(n_pub), as_array=True) if n_pub else []
        k_delta_G1 = self.E.batch_mul(G1, k_delta.download(n_wires - n_pub), as_array=True) if n_wires > n_pub else []
        if worker is not None:
            worker.join()
            if worker.error is not None:
                raise worker.error
            if n_wires > n_pub:
                self._build_plan(k_delta_G1, "k")
            self.qap._device_matrices(self._qap_needs())
            self.qap._workspace(n, n_wires)
            self.qap._qap_stream()

        alpha_G1, beta_G1, delta_G1 = G1 * alpha, G1 * beta, G1 * delta
        beta_G2, gamma_G2, delta_G2 = G2 * beta, G2 * gamma, G2 * delta
        self.proving_key = ProvingKey(alpha_G1, beta_G1, beta_G2, delta_G1, delta_G2, tau_G1, tau_G2, target_G1, k_delta_G1)
        self.verifying_key = VerifyingKey(alpha_G1, beta_G2, gamma_G2, delta_G2, k_gamma_G1)

    # ------------------------------------------------------------------------------------------
    # task -> (slot of its PointArray, stream priority): <tau_1, u> and <tau_1, v> are two slots over one key vector
    _TASK_SLOT = {"k": (0, False), "u": (0, False), "v1": (1, False), "v2": (0, True), "h": (0, False)}

    def _task_sizes(self):
        """points per MSM of prove(): from the key when there is one, else from the QAP's shape (what setup() will build)"""
        pk = self.proving_key
        n, n_wires, n_pub = self.qap.a.n_row, self.qap.a.n_col, self.qap.n_public
        if pk is not None:
            return {"k": len(pk.kdelta_1), "u": min(n, len(pk.tau_1)), "v1": min(n, len(pk.tau_1)), "v2": min(n, len(pk.tau_2)),
                    "h": min(n, len(pk.target_1))}
        return {"k": n_wires - n_pub, "u": n, "v1": n, "v2": n, "h": n}

    def _my_tasks(self):
        """{task: (first window, count)} of this rank, or None without sharding.  Every rank computes the same table."""
        if self._shard is None or self._shard[1] <= 1:
            return None
        if self._assignment is None:
            from ..parallel import TASK_GROUP, partition_proof, window_partition
            lib, cid = N.load(), self.E.curve.curve_id
            flags = N.MSM_PRECOMPUTE if self.precompute_keys else 0
            sizes = self._task_sizes()
            n_windows, whole, bits = {}, {}, {}
            # From 2^21 constraints on every plan of the task partition, split or whole, takes the layout of an unsharded plan
            # (13 windows of 20 bits: 19 % fewer bucket additions than 16 of 16; the larger shared bucket set costs 0.2-0.6 ms per
            # rank, less than the additions saved).  At 2^20 and below only ranks that hold an MSM whole do (_task_range).
            wide = self._partition == "task" and self.qap.a.n_row >= (1 << 21)
            for task, count in sizes.items():
                if count > 0:
                    c, nwin = N.ctypes.c_int(0), N.ctypes.c_int(0)
                    N.check(lib.zk_msm_window_layout_ex(cid, TASK_GROUP[task], count, flags, 0, 1, c, nwin))
                    whole[task] = nwin.value
                    if wide:
                        n_windows[task], bits[task] = nwin.value, c.value
                    else:
                        N.check(lib.zk_msm_window_layout_ex(cid, TASK_GROUP[task], count, flags, 0, 0, c, nwin))
                        n_windows[task] = nwin.value
            self._n_windows, self._task_bits = n_windows, bits
            if self._partition == "window":
                self._assignment = window_partition(self._shard[1], n_windows)
                self.projected_ms = None
            else:
                self._assignment, self.projected_ms = partition_proof(self._shard[1], n_windows, cid, self.qap.a.n_row, whole, wide)
        return self._assignment[self._shard[0]]

    def _task_range(self, task):
        """this rank's (first, count) windows of `task` (count 0: none); None = the whole MSM in the layout of an unsharded plan
        (no sharding, or a rank of the task partition that holds this MSM whole: 13 wide windows instead of 16 from 2^20 points
        on)"""
        mine = self._my_tasks()
        if mine is None:
            return None
        rng = mine.get(task, (0, 0))
        if self._partition == "task" and rng[1] and rng[1] == self._n_windows.get(task):
            return None
        return rng

    def _build_plan(self, arr, task):
        slot, high_priority = self._TASK_SLOT[task]
        rng = self._task_range(task)
        if rng is not None:
            if rng[1] == 0:
                return None
            arr.slot_ranges[slot] = rng
        else:
            arr.slot_ranges.pop(slot, None)
        return arr.plan(slot, precompute=self.precompute_keys, high_priority=high_priority, concurrent=True,
                        window_bits=self._task_bits.get(task, 0))

    def prepare_prover(self):
        """build what the first prove() would otherwise build: the fixed-base MSM plans of the proving key and the QAP's
        device-resident matrices and workspace.  setup() does this by default; call it after loading a key from bytes
        (`groth16.proving_key = ProvingKey.from_bytes(...)`) to pay the one-off cost before the first proof.  Not in the
        reference's API."""
        assert self.proving_key, "ProvingKey has not been generated"
        pk = self.proving_key
        pk.tau_1, pk.tau_2 = _as_array(self.E, pk.tau_1, 1), _as_array(self.E, pk.tau_2, 2)
        pk.target_1, pk.kdelta_1 = _as_array(self.E, pk.target_1, 1), _as_array(self.E, pk.kdelta_1, 1)
        for arr, task in ((pk.tau_1, "u"), (pk.tau_2, "v2"), (pk.tau_1, "v1"), (pk.target_1, "h"), (pk.kdelta_1, "k")):
            if len(arr):
                self._build_plan(arr, task)
        n, n_wires = self.qap.a.n_row, self.qap.a.n_col
        self.qap._device_matrices(self._qap_needs())
        self.qap._workspace(n, n_wires)
        self.qap._qap_stream()

    def _qap_needs(self):
        """which of the QAP's outputs this rank's MSMs read: a subset of {"u", "v", "h"}; None = everything (no sharding)"""
        mine = self._my_tasks()
        if mine is None:
            return None
        from ..parallel import TASK_NEEDS
        return {TASK_NEEDS[t] for t in mine} - {"w"}

    def _start_plan_worker(self, jobs):
        import threading

        def run():
            try:
                N.bind_thread()
                for arr, task in jobs:
                    self._build_plan(arr, task)
            except Exception as exc:  # noqa: BLE001 - re-raised by the caller after join()
                worker.error = exc

        worker = threading.Thread(target=run, name="zkmi-plan-builder")
        worker.error = None
        worker.start()
        return worker

    def shard_over_ranks(self, device=None, partition="task"):
        """split prove() over the ranks of the default torch.distributed group (one process per GPU; backend nccl = RCCL, or
        gloo).  partition="task" (default): the task x window partition of parallel.partition_proof -- a rank runs one MSM (or a
        window range of a long one) and only the part of the QAP that MSM needs; "window": every rank runs a window range of every
        MSM and the whole QAP (rounds 1-3).  `device` is where the gathered partial points are staged (the rank's GPU for RCCL,
        None for gloo).  Call it BEFORE setup() so that the plans built there cover this rank's share only; plans that already
        exist for other windows are dropped and rebuilt on the next prove()."""
        import torch.distributed as dist
        if partition not in ("task", "window"):
            raise ValueError("partition must be 'task' or 'window'")
        self._shard = (dist.get_rank(), dist.get_world_size(), device)
        self._partition = partition
        self._assignment = None
        if self._shard[1] > 1 and self.proving_key is not None:
            pk = self.proving_key
            for arr in (pk.tau_1, pk.tau_2, pk.target_1, pk.kdelta_1):
                if isinstance(arr, PointArray):
                    arr.release()   # rebuilt for this rank's windows by the next prove() / prepare_prover()
                    arr.slot_ranges = {}

    def _enqueue_msm(self, bases, task, d_scalars, count, share_sort_of=None, sort_only=False, wait_event=None):
        """start <bases[:count], scalars> (scalars already in HBM) on the plan's own stream; with sharding only
        this rank's windows of `task`.  Returns (array, handle); handle None = this rank has no window of that MSM.
        sort_only: digits and sort only -- the accumulate kernel and the reduction follow with _enqueue_rest."""
        from ..parallel import TASK_GROUP
        lib = N.load()
        arr = _as_array(self.E, bases, TASK_GROUP[task])
        slot, high_priority = self._TASK_SLOT[task]
        first, cnt = 0, 0  # 0, 0 = all windows
        rng = self._task_range(task)
        if rng is not None:
            # this rank's windows are known before the plan exists, so the plan (fixed-base table rows, workspace) is
            # created for that range only: 1/8 of the table memory and build time on 8 ranks
            first, cnt = rng
            if cnt == 0:
                return arr, None
            arr.slot_ranges[slot] = (first, cnt)
        else:
            arr.slot_ranges.pop(slot, None)
        handle = arr.plan(slot, precompute=self.precompute_keys, high_priority=high_priority, concurrent=True,
                          window_bits=self._task_bits.get(task, 0))
        if wait_event is not None:
            N.check(lib.zk_msm_plan_wait_event(handle, wait_event))   # the scalars are still being produced on another stream
        if share_sort_of is not None and count == len(arr) and not os.environ.get("ZKMI_NO_SHARED_SORT"):
            # <tau_1, v> is already in flight with the same scalars: B2 = <tau_2, v> reuses its digits and sorted entries
            # (refused, and sorted normally, when the two plans hold different window ranges)
            if lib.zk_msm_plan_enqueue_shared(handle, share_sort_of, N.STREAM_PLAN) == N.ZK_OK:
                self._live.append(handle)
                return arr, handle
        enqueue = lib.zk_msm_plan_enqueue_sort if sort_only else lib.zk_msm_plan_enqueue
        N.check(enqueue(handle, count, d_scalars, 1, first, cnt, N.STREAM_PLAN))
        self._live.append(handle)
        return arr, handle

    @staticmethod
    def _enqueue_rest(handle, after):
        """accumulate kernel, reduction and D2H of a plan whose sort is in flight, the accumulate kernel not before that of
        `after` (a handle or None) has finished"""
        if handle is not None:
            N.check(N.load().zk_msm_plan_enqueue_rest(handle, after or 0))

    def _finish_msm(self, handle, group):
        """affine limbs of the (partial) MSM result; all-zero = infinity"""
        lib = N.load()
        out = np.zeros(N.point_limbs(self.E.curve.curve_id, group), dtype=np.uint64)
        if handle is not None:
            N.check(lib.zk_msm_plan_finish(handle, N.u64p(out)))
            if handle in self._live:
                self._live.remove(handle)
        return out

    def _points(self, parts):
        from .._algebra import _point_class
        cid = self.E.curve.curve_id
        return [_point_class(cid, group)._from_limbs(t) for t, group in parts]

    # flag word that travels with the partial points: 0 = fine, 1 = this rank's witness check failed (a_i b_i != c_i somewhere),
    # 2 = this rank failed otherwise.  Every rank learns of a failure in the SAME collective it would have waited in, and
    # raises too, instead of hanging in all_gather while the failing rank has left prove() (round-3 advisor finding).
    _FLAG_OK, _FLAG_WITNESS, _FLAG_ERROR = 0, 1, 2

    def _exchange(self, parts, flag=0):
        """parts: [(limbs, group)] partial points of this rank (zeros for MSMs it holds no window of) -> totals over all
        ranks; one all_gather.  Raises on every rank when any rank reports a failure."""
        import time
        from ..parallel import all_gather_limbs, sum_points
        cid = self.E.curve.curve_id
        flat = np.concatenate([p for p, _ in parts] + [np.array([flag], dtype=np.uint64)])
        t0 = time.perf_counter()
        self._exchanged = True
        gathered = all_gather_limbs(flat, self._shard[2])  # (world, len)
        self._collective_ms += (time.perf_counter() - t0) * 1e3
        flags = gathered[:, -1]
        if (flags == self._FLAG_WITNESS).any():
            raise ValueError("Failed to evaluate with the given witness")
        if flags.any():
            raise RuntimeError(f"Groth16.prove failed on rank(s) {[int(r) for r in np.nonzero(flags)[0]]}")
        totals, off = [], 0
        for p, group in parts:
            totals.append((sum_points(cid, group, list(gathered[:, off:off + p.shape[0]])), group))
            off += p.shape[0]
        return self._points(totals)

    def _report_failure(self, flag):
        """a sharded rank that fails before the collective still takes part in it, with its flag set (points all zero)"""
        cid = self.E.curve.curve_id
        zeros = [(np.zeros(N.point_limbs(cid, g), dtype=np.uint64), g) for g in (1, 2, 1, 1, 1)]
        try:
            self._exchange(zeros, flag)
        except (ValueError, RuntimeError):
            pass   # the flag we just sent; the caller re-raises the original exception

    def prove(self, public_witness, private_witness) -> Proof:
        """public_witness / private_witness: lists of ints (reference API) or (k, 4) uint64 limb arrays."""
        import time
        assert self.proving_key, "ProvingKey has not been generated"
        pk = self.proving_key
        assert len(pk.kdelta_1) == len(private_witness), "Length of kdelta_1 and private_witness must be equal"
        q = self.order
        t_start = time.perf_counter()
        if self._blinding is not None:
            r, s = self._blinding
        else:
            r, s = get_random_int(q - 1), get_random_int(q - 1)
            if self._shard is not None and self._shard[1] > 1:  # every rank assembles the same proof
                import torch.distributed as dist
                box = [(r, s)]
                dist.broadcast_object_list(box, src=0)
                r, s = box[0]

        # limb arrays go up as they are; lists of ints (the reference's call shape) are repacked by the QAP straight into its
        # page-locked staging rows -- no concatenated copy of the two lists, no fresh 32 MB array per proof
        witness = tuple(x if isinstance(x, (np.ndarray, list, tuple)) else list(x) for x in (public_witness, private_witness))
        n_pub = self.qap.n_public
        n_priv = len(private_witness)
        early = {}

        def start_witness_msm(d_witness):
            # <kdelta_1, w_priv> needs the witness only: it starts on its plan's stream as soon as the witness is in
            # HBM and runs beside the QAP transform chain
            if n_priv > 0:
                pk.kdelta_1, early["k"] = self._enqueue_msm(pk.kdelta_1, "k", d_witness.ptr + 32 * n_pub, n_priv)

        ordered = not os.environ.get("ZKMI_UNORDERED_MSMS")
        n_rows = self.qap.a.n_row

        def start_uv_sorts(event, d_u, d_v):
            # u and v are final a third of the way into the QAP chain: the sorts of <tau_1, v> and <tau_1, u> run beside
            # the rest of it instead of in a phase of their own afterwards
            pk.tau_1, early["v1"] = self._enqueue_msm(pk.tau_1, "v1", _ptr(d_v), min(n_rows, len(pk.tau_1)), sort_only=True, wait_event=event)
            pk.tau_1, early["u"] = self._enqueue_msm(pk.tau_1, "u", _ptr(d_u), min(n_rows, len(pk.tau_1)), sort_only=True, wait_event=event)

        # A plan accepts one run at a time: whatever goes wrong between the first enqueue and the last finish (a witness that
        # fails the divisibility check, an allocation failure, a HIP error, KeyboardInterrupt), the runs still in flight are
        # cancelled before the exception leaves, or every later prove() on this key would fail with "already has a run in flight"
        self._live = []
        sharded = self._shard is not None and self._shard[1] > 1
        self._exchanged = not sharded
        try:
            if sharded and not self._my_tasks():
                # more ranks than the partition has pieces: this rank holds no window of any MSM.  It uploads nothing and
                # only takes part in the proof's collective (zero points, flag 0), then assembles the same proof as the others
                from .qap import DeviceQapResult
                res = DeviceQapResult(n_rows, None, None, None, None)
            else:
                try:
                    res = self.qap.evaluate_witness_device(witness, after_upload=start_witness_msm, after_uv=start_uv_sorts if ordered else None,
                                                           needs=self._qap_needs())
                except ValueError as exc:
                    raise ValueError("Failed to evaluate with the given witness") from exc
            return self._prove_msms(pk, res, early, ordered, r, s, q, t_start)
        except BaseException as exc:
            lib = N.load()
            for handle in self._live:
                lib.zk_msm_plan_cancel(handle)   # status ignored: the original exception is the one to report
            if not self._exchanged:
                # the other ranks are (or will be) waiting in the proof's one collective: meet them there with the flag set
                self._report_failure(self._FLAG_WITNESS if isinstance(exc, ValueError) else self._FLAG_ERROR)
            raise
        finally:
            self._live = []
            self._exchanged = True

    def _prove_msms(self, pk, res, early, ordered, r, s, q, t_start):
        import time
        t_qap = time.perf_counter()
        self.last_timings = {"qap_ms": (t_qap - t_start) * 1e3}   # completed below; a sharded rank that stops at the collective keeps this
        self._collective_ms = 0.0
        n = res.n
        # The four MSMs over u, v, h run on their plans' own streams.  An accumulate kernel fills every wave slot of the
        # chip until it ends: left to themselves two of them only slow each other down and the sorts of the plans behind
        # them starve.  So the three sorts go first, side by side, and the accumulate kernels run one after the other,
        # the G2 one (three times the work of a G1 one, and the longest reduction tail) at the head: every reduction but
        # the last overlaps the next plan's accumulate kernel.
        if ordered:
            h_v1, h_u = early.get("v1"), early.get("u")   # sorts enqueued beside the QAP chain
            pk.target_1, h_h = self._enqueue_msm(pk.target_1, "h", _ptr(res.h),
                                                 min(n, len(pk.target_1)), sort_only=True)
        else:
            pk.tau_1, h_v1 = self._enqueue_msm(pk.tau_1, "v1", _ptr(res.v), min(n, len(pk.tau_1)))
        pk.tau_2, h_v2 = self._enqueue_msm(pk.tau_2, "v2", _ptr(res.v), min(n, len(pk.tau_2)), share_sort_of=h_v1)
        if ordered:
            self._enqueue_rest(h_v1, h_v2)
            self._enqueue_rest(h_u, h_v1 or h_v2)
            self._enqueue_rest(h_h, h_u or h_v1 or h_v2)
        else:
            pk.tau_1, h_u = self._enqueue_msm(pk.tau_1, "u", _ptr(res.u), min(n, len(pk.tau_1)))
            pk.target_1, h_h = self._enqueue_msm(pk.target_1, "h", _ptr(res.h), min(n, len(pk.target_1)))
        h_k = early.get("k")
        t_enq = time.perf_counter()
        # the blinding terms depend on the key and (r, s) only: the host computes them (four scalar multiplications,
        # ~1 ms) while the GPU works through the MSMs, instead of after them
        a_fixed = pk.alpha_1 + pk.delta_1 * r
        b1_fixed = pk.beta_1 + pk.delta_1 * s
        b2_fixed = pk.beta_2 + pk.delta_2 * s
        c_fixed = (-pk.delta_1) * (r * s % q)
        # collected in the order the GPU completes them, so that only the last host tail is left once the GPU is idle
        f_k = self._finish_msm(h_k, 1)
        f_v2 = self._finish_msm(h_v2, 2)
        f_v1 = self._finish_msm(h_v1, 1)
        f_u = self._finish_msm(h_u, 1)
        if self._shard is None or self._shard[1] <= 1:
            # A, B and the two scalar multiplications of C need <tau, u> and <tau, v> only: done on the host while the GPU
            # still works on <target_1, h>, the last MSM of the chain
            msm_u, msm_v2, msm_v1, sum_delta_witness = self._points([(f_u, 1), (f_v2, 2), (f_v1, 1), (f_k, 1)])
            A = msm_u + a_fixed
            B1 = msm_v1 + b1_fixed
            B2 = msm_v2 + b2_fixed
            c_partial = sum_delta_witness + A * s + B1 * r + c_fixed
            f_h = self._finish_msm(h_h, 1)
            t_fin = time.perf_counter()
            (HZ,) = self._points([(f_h, 1)])
            C = HZ + c_partial
        else:
            f_h = self._finish_msm(h_h, 1)
            t_fin = time.perf_counter()
            msm_u, msm_v2, msm_v1, HZ, sum_delta_witness = self._exchange([(f_u, 1), (f_v2, 2), (f_v1, 1), (f_h, 1), (f_k, 1)])
            A = msm_u + a_fixed
            B1 = msm_v1 + b1_fixed
            B2 = msm_v2 + b2_fixed
            C = HZ + sum_delta_witness + A * s + B1 * r + c_fixed
        t_end = time.perf_counter()
        # qap_ms is the part a window-sharded prover REPLICATES on every rank (witness upload, three sparse products, six
        # transforms); msm_* is what the ranks share; collective_ms the all_gather of the partial points inside the exchange
        self.last_timings = {"qap_ms": (t_qap - t_start) * 1e3, "msm_enqueue_ms": (t_enq - t_qap) * 1e3,
                             "msm_finish_ms": (t_fin - t_enq) * 1e3, "exchange_assemble_ms": (t_end - t_fin) * 1e3,
                             "collective_ms": self._collective_ms}
        return Proof(A, B2, C)

    # ------------------------------------------------------------------------------------------
    def verify(self, proof: Proof, public_witness: list) -> bool:
        assert self.verifying_key, "VerifyingKey has not been generated"
        vk = self.verifying_key
        assert len(vk.ic) == len(public_witness), "Length of IC and public_witness must be equal"
        if isinstance(public_witness, np.ndarray):
            public_witness = N.limbs_to_ints(public_witness)
        sum_gamma_witness = self.E.multiexp(vk.ic, public_witness)
        # e(A, B) == e(alpha, beta) * e(sum_gamma_witness, gamma) * e(C, delta)  (protocol.py:176-186), checked as
        # e(-A, B) * e(alpha, beta) * e(sum_gamma_witness, gamma) * e(C, delta) == 1: the same predicate through ONE
        # multi-pairing, i.e. one final exponentiation (the expensive half of the host-side pairing) instead of two
        prod = self.E.multi_pairing([-proof.A, vk.alpha_1, sum_gamma_witness, proof.C],
                                    [proof.B, vk.beta_2, vk.gamma_2, vk.delta_2])
        return prod.is_one()
