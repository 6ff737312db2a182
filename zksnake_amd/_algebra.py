"""
zksnake_amd._algebra -- the drop-in for the reference's native extension `zksnake._algebra`
(pyo3 module built in reference src/lib.rs:6-185).  Same submodule names and callables:

    ec_bn254, ec_bls12_381:
        PointG1, PointG2, g1(), g2(), batch_multi_scalar_g1/g2, multiscalar_mul_g1/g2,
        pairing, multi_pairing                                (src/bn254/curve.rs)
    polynomial_bn254, polynomial_bls12_381:
        Polynomial, fft, ifft, coset_fft, coset_ifft, add/mul_over_evaluation_domain,
        evaluate_vanishing_polynomial, evaluate_lagrange_coefficients,
        get_evaluation_point, get_all_evaluation_points       (src/bn254/polynomial.rs)

Every vector / multi-point operation goes through the C ABI of libzkmi.so (HIP kernels on gfx950);
single-point arithmetic and the codecs are the library's host functions.  Nothing here computes
field or curve arithmetic in Python except the small setup/verify-side scalar helpers
(vanishing polynomial, naive polynomial algebra that the reference also runs on the CPU).

On top of the reference's list-of-int interface every vector entry point also accepts numpy
limb arrays ((n, 4) uint64, canonical little-endian limbs) and `PointArray` objects, so that
2^20-element inputs do not pay Python big-int marshalling.
"""

import types

import numpy as np

from . import _native as N
from . import constant
from . import pairing as _pairing

_MODULUS = {0: constant.BN254_SCALAR_FIELD, 1: constant.BLS12_381_SCALAR_FIELD}
_FIELD = {0: constant.BN254_MODULUS, 1: constant.BLS12_381_MODULUS}
_TWO_ADICITY = {0: 28, 1: 32}


def _scalar_limbs(values, cid):
    """list[int] | (n,4) uint64 -> (n,4) uint64 canonical (values reduced mod r like Fr::from)."""
    if isinstance(values, np.ndarray):
        arr = np.ascontiguousarray(values, dtype=np.uint64)
        if arr.ndim != 2 or arr.shape[1] != 4:
            raise TypeError("scalar limb arrays must have shape (n, 4)")
        return arr
    return N.ints_to_limbs(values, 4, _MODULUS[cid])


# =====================================================================================
# points
# =====================================================================================

class PointArray:
    """n affine points of one group as an (n, limbs) uint64 array (canonical coordinates,
    all-zero row = infinity).  Behaves like a read-only list of points; keeps a lazily built
    device-resident MSM plan so repeated multiexps against the same key stay in HBM."""

    def __init__(self, curve_id, group, limbs):
        self.curve_id = curve_id
        self.group = group
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, N.point_limbs(curve_id, group))
        self._plans = {}
        self._plan_layout = {}    # (slot, precompute flag) -> (window_bits, window range) the plan of that key was created with
        self._plan_concurrent = {}  # plan handle -> the "runs beside other plans" setting it carries
        self.window_range = None  # (first, count): plans cover only these windows (a rank of a window-sharded MSM) ...
        self.slot_ranges = {}     # ... unless the slot has a range of its own (a rank that holds different windows of <tau_1, u>
        #                           and <tau_1, v>, Groth16.shard_over_ranks)

    def __len__(self):
        return self.limbs.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return PointArray(self.curve_id, self.group, self.limbs[i])
        cls = _point_class(self.curve_id, self.group)
        return cls._from_limbs(self.limbs[i].copy())

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def to_list(self):
        return list(self)

    # below this many points the per-point host codec is used (no GPU needed: a verifying key's few ic points)
    BATCH_CODEC_MIN = 64

    def to_bytes(self):
        """the points' compressed encodings back to back (what `b"".join(p.to_bytes() for p in points)` gives)"""
        lib = N.load()
        nb = lib.zk_point_bytes(self.curve_id, self.group)
        out = np.zeros(len(self) * nb, dtype=np.uint8)
        if len(self) >= self.BATCH_CODEC_MIN:
            N.ensure_gpu()
            N.check(lib.zk_points_compress(self.curve_id, self.group, len(self), N.u64p(self.limbs), N.u8p(out), None))
        else:
            for i, row in enumerate(self.limbs):
                N.check(lib.zk_point_compress(self.curve_id, self.group, N.u64p(row), N.u8p(out[i * nb:(i + 1) * nb])))
        return out.tobytes()

    @classmethod
    def from_compressed(cls, curve_id, group, data, count):
        """`count` compressed points read from `data`, validated like from_hex (flags, range, curve, subgroup)"""
        lib = N.load()
        nb = lib.zk_point_bytes(curve_id, group)
        raw = np.frombuffer(data, dtype=np.uint8, count=count * nb)
        limbs = np.zeros((count, N.point_limbs(curve_id, group)), dtype=np.uint64)
        if count >= cls.BATCH_CODEC_MIN:
            N.ensure_gpu()
            N.check(lib.zk_points_decompress(curve_id, group, count, N.u8p(raw), N.u64p(limbs), None))
        else:
            for i in range(count):
                N.check(lib.zk_point_decompress(curve_id, group, N.u8p(raw[i * nb:(i + 1) * nb]), N.u64p(limbs[i])))
        return cls(curve_id, group, limbs)

    def plan(self, slot=0, precompute=False, high_priority=False, window_bits=0, concurrent=False):
        """device-resident bases + workspace (created on first use).  A second slot gives an independent
        workspace so two MSMs over the same key (tau_1 with u and with v) can be in flight together.
        precompute=True builds the fixed-base table 2^(c w) P_i (ZK_MSM_PRECOMPUTE): right for proving keys,
        which are reused by every proof.  high_priority puts the plan's own stream at the top priority level;
        window_bits 0 = the library's choice for len(self) points.  concurrent=True says that the plan runs beside other plans
        (a prover's pipelines): a NEW plan then keeps one issue priority in its accumulate kernel (plan option "priority_steps" = 0;
        the steps pay only when the kernel has the GPU to itself, include/zkmi.h)."""
        lib = N.load()
        pre = bool(precompute)
        key = (slot, pre)
        # A plan is cached per (slot, mode) together with the layout it was created with.  A call that asks for another window
        # width, or comes after the slot's window range changed, drops that plan and builds anew instead of silently handing
        # back a plan with the old layout (round-2 advisor finding).  window_bits 0 = "whatever the mode's plans have".
        rng = self.slot_ranges.get(slot, self.window_range)
        have = self._plan_layout.get(key)
        if have is not None and (have[1] != rng or (window_bits and have[0] != int(window_bits))):
            self._drop(key)
            have = None
        if window_bits == 0:
            # another slot of the mode fixes the width (clones share one table)
            window_bits = next((lay[0] for (s, p), lay in self._plan_layout.items() if p == pre and lay[0]), 0)
        elif have is None:
            # a new width for the mode: the other slots' plans were built over the old table, they go too
            for k in [k for k, lay in self._plan_layout.items() if k[1] == pre and lay[0] != int(window_bits)]:
                self._drop(k)
        if key not in self._plans:
            lib = N.ensure_gpu()
            h = N._u64(0)
            other = next((self._plans[k] for k, lay in self._plan_layout.items()
                          if k[1] == pre and lay == (int(window_bits), rng) and k in self._plans), None)
            if other is not None:
                # further slots over the same windows share the first plan's bases (and its fixed-base table): only the
                # workspace is new
                N.check(lib.zk_msm_plan_clone(other, h))
            else:
                flags = (N.MSM_PRECOMPUTE if precompute else 0) | (N.MSM_HIGH_PRIORITY if high_priority else 0)
                if rng is not None:
                    first, count = rng
                    N.check(lib.zk_msm_plan_create_range(self.curve_id, self.group, len(self), self.limbs.ctypes.data, 0, flags,
                                                         window_bits, first, count, h))
                else:
                    N.check(lib.zk_msm_plan_create(self.curve_id, self.group, len(self), self.limbs.ctypes.data, 0, flags,
                                                   window_bits, h))
            self._plans[key] = h.value
            self._plan_layout[key] = (int(window_bits), rng)
            self._plan_concurrent[h.value] = False
        handle = self._plans[key]
        if concurrent and not self._plan_concurrent.get(handle):
            # also for a plan an earlier, non-concurrent caller created (round-3 advisor finding: a direct pk.tau_1.plan() left
            # the prover's accumulate kernels with their priority steps).  Refused while a run is in flight: tried again by the
            # next call.
            if lib.zk_msm_plan_set_option(handle, b"priority_steps", 0) == N.ZK_OK:
                self._plan_concurrent[handle] = True
        return handle

    def _drop(self, key):
        """destroy the plan of one (slot, mode); a run still in flight on it is cancelled first (waits for the plan's stream)"""
        handle = self._plans.pop(key, None)
        self._plan_layout.pop(key, None)
        if handle is not None:
            lib = N.load()
            lib.zk_msm_plan_cancel(handle)   # status ignored: nothing in flight is the normal case
            lib.zk_msm_plan_destroy(handle)
            self._plan_concurrent.pop(handle, None)

    def release(self):
        for key in list(self._plans):
            self._drop(key)
        self._plans = {}
        self._plan_layout = {}
        self._plan_concurrent = {}

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def _points_to_limbs(points, cid, group):
    if isinstance(points, PointArray):
        return points.limbs
    w = N.point_limbs(cid, group)
    out = np.empty((len(points), w), dtype=np.uint64)
    for i, p in enumerate(points):
        out[i] = p._limbs
    return out


class _PointBase:
    """shared behaviour of PointG1 / PointG2 (reference src/bn254/curve.rs:25-186, 194-324)."""

    CURVE = 0
    GROUP = 1
    __slots__ = ("_limbs",)

    @classmethod
    def _from_limbs(cls, limbs):
        obj = object.__new__(cls)
        obj._limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        return obj

    @classmethod
    def _coords_to_limbs(cls, coords):
        fq = N.fq_limbs(cls.CURVE)
        p = _FIELD[cls.CURVE]
        buf = b"".join((int(c) % p).to_bytes(8 * fq, "little") if int(c) >= 0 else _neg_error() for c in coords)
        return np.frombuffer(buf, dtype=np.uint64).copy()

    def _coords(self):
        fq = N.fq_limbs(self.CURVE)
        raw = self._limbs.tobytes()
        return [int.from_bytes(raw[8 * fq * k:8 * fq * (k + 1)], "little") for k in range(2 * self.GROUP)]

    def _check_on_curve(self):
        lib = N.load()
        if lib.zk_point_on_curve(self.CURVE, self.GROUP, N.u64p(self._limbs)) != 1:
            # ark's G1Affine::new panics on an off-curve point (curve.rs:29)
            raise ValueError("point is not on the curve")

    def _binary(self, other, negate_other=False):
        if type(other) is not type(self):
            raise TypeError(f"cannot combine {type(self).__name__} with {type(other).__name__}")
        lib = N.load()
        rhs = other._limbs
        if negate_other:
            tmp = np.zeros_like(rhs)
            N.check(lib.zk_point_neg(self.CURVE, self.GROUP, N.u64p(rhs), N.u64p(tmp)))
            rhs = tmp
        out = np.zeros_like(self._limbs)
        N.check(lib.zk_point_add(self.CURVE, self.GROUP, N.u64p(self._limbs), N.u64p(rhs), N.u64p(out)))
        return self._from_limbs(out)

    def __add__(self, other):
        return self._binary(other)

    def __radd__(self, other):
        return self._binary(other)

    def __sub__(self, other):
        return self._binary(other, negate_other=True)

    def __rsub__(self, other):
        # the reference computes self - other here too (curve.rs:93-95)
        return self._binary(other, negate_other=True)

    def __neg__(self):
        lib = N.load()
        out = np.zeros_like(self._limbs)
        N.check(lib.zk_point_neg(self.CURVE, self.GROUP, N.u64p(self._limbs), N.u64p(out)))
        return self._from_limbs(out)

    def __mul__(self, scalar):
        if not isinstance(scalar, int) or isinstance(scalar, bool):
            return NotImplemented
        if scalar < 0:
            raise OverflowError("can't convert negative int to unsigned")
        lib = N.load()
        k = N.ints_to_limbs([scalar % _MODULUS[self.CURVE]], 4)
        out = np.zeros_like(self._limbs)
        N.check(lib.zk_point_mul(self.CURVE, self.GROUP, N.u64p(self._limbs), N.u64p(k), N.u64p(out)))
        return self._from_limbs(out)

    __rmul__ = __mul__

    def __eq__(self, other):
        return type(other) is type(self) and bool((self._limbs == other._limbs).all())

    def __hash__(self):
        return hash((self.CURVE, self.GROUP, self._limbs.tobytes()))

    def is_zero(self):
        return not self._limbs.any()

    def to_bytes(self):
        """compressed encoding as a list of ints (the reference returns Vec<u8> -> list)"""
        lib = N.load()
        nb = lib.zk_point_bytes(self.CURVE, self.GROUP)
        buf = np.zeros(nb, dtype=np.uint8)
        N.check(lib.zk_point_compress(self.CURVE, self.GROUP, N.u64p(self._limbs), N.u8p(buf)))
        return buf.tolist()

    def to_hex(self):
        return bytes(self.to_bytes()).hex()

    @classmethod
    def from_bytes(cls, data):
        lib = N.load()
        nb = lib.zk_point_bytes(cls.CURVE, cls.GROUP)
        data = bytes(data)
        if len(data) != nb:
            raise ValueError(f"Cannot deserialize point: expected {nb} bytes, got {len(data)}")
        buf = np.frombuffer(data, dtype=np.uint8).copy()
        out = np.zeros(N.point_limbs(cls.CURVE, cls.GROUP), dtype=np.uint64)
        st = lib.zk_point_decompress(cls.CURVE, cls.GROUP, N.u8p(buf), N.u64p(out))
        if st != N.ZK_OK:
            raise ValueError(lib.zk_last_error().decode())
        return cls._from_limbs(out)

    @property
    def generator(self):
        lib = N.load()
        out = np.zeros_like(self._limbs)
        N.check(lib.zk_point_generator(self.CURVE, self.GROUP, N.u64p(out)))
        return self._from_limbs(out)

    def __repr__(self):
        return self.__str__()


def _neg_error():
    raise OverflowError("can't convert negative int to unsigned")


_POINT_CLASSES = {}


def _point_class(cid, group):
    return _POINT_CLASSES[(cid, group)]


def _make_point_classes(cid):
    class PointG1(_PointBase):
        CURVE = cid
        GROUP = 1
        __slots__ = ()

        def __init__(self, x, y):
            self._limbs = self._coords_to_limbs([x, y])
            if x or y:
                self._check_on_curve()

        @property
        def x(self):
            return self._coords()[0]

        @property
        def y(self):
            return self._coords()[1]

        @classmethod
        def identity(cls):
            return cls._from_limbs(np.zeros(N.point_limbs(cid, 1), dtype=np.uint64))

        def __str__(self):
            if self.is_zero():
                return "infinity"
            c = self._coords()
            return f"({c[0]}, {c[1]})"

    class PointG2(_PointBase):
        CURVE = cid
        GROUP = 2
        __slots__ = ()

        def __init__(self, x1, x2, y1, y2):
            self._limbs = self._coords_to_limbs([x1, x2, y1, y2])
            if x1 or x2 or y1 or y2:
                self._check_on_curve()

        @property
        def x(self):
            return self._coords()[0:2]

        @property
        def y(self):
            return self._coords()[2:4]

        def __str__(self):
            c = self._coords()
            return f"([{c[0]}, {c[1]}], [{c[2]}, {c[3]}])"

    PointG1.__qualname__ = PointG1.__name__ = "PointG1"
    PointG2.__qualname__ = PointG2.__name__ = "PointG2"
    _POINT_CLASSES[(cid, 1)] = PointG1
    _POINT_CLASSES[(cid, 2)] = PointG2
    return PointG1, PointG2


def _make_ec(cid):
    PointG1, PointG2 = _make_point_classes(cid)

    def _gen(group):
        lib = N.load()
        out = np.zeros(N.point_limbs(cid, group), dtype=np.uint64)
        N.check(lib.zk_point_generator(cid, group, N.u64p(out)))
        return _point_class(cid, group)._from_limbs(out)

    def g1():
        return _gen(1)

    def g2():
        return _gen(2)

    def _batch(group, points, scalars, as_array=False):
        """batch_multi_scalar_g1/g2 (curve.rs:326-354): out[i] = scalars[i] * points[i]"""
        lib = N.ensure_gpu()
        sc = _scalar_limbs(scalars, cid)
        n = sc.shape[0]
        if isinstance(points, _PointBase):
            base, broadcast = points._limbs, 1
        else:
            base = _points_to_limbs(points, cid, group)
            broadcast = 0
            if base.shape[0] != n:
                # the reference zips (&points, &scalars): the shorter one bounds the result
                n = min(n, base.shape[0])
                sc, base = sc[:n], base[:n]
        out = np.zeros((n, N.point_limbs(cid, group)), dtype=np.uint64)
        if n:
            sc = np.ascontiguousarray(sc)
            base = np.ascontiguousarray(base)
            N.check(lib.zk_batch_mul(cid, group, n, N.u64p(sc), N.u64p(base), broadcast, N.u64p(out)))
        arr = PointArray(cid, group, out)
        return arr if as_array else arr.to_list()

    def batch_multi_scalar_g1(points, scalars, as_array=False):
        return _batch(1, points, scalars, as_array)

    def batch_multi_scalar_g2(points, scalars, as_array=False):
        return _batch(2, points, scalars, as_array)

    def _msm(group, points, scalars):
        """multiscalar_mul_g1/g2 (curve.rs:356-392)"""
        lib = N.ensure_gpu()
        sc = _scalar_limbs(scalars, cid)
        n_s = sc.shape[0]
        n_p = len(points)
        if n_s != n_p:
            raise ValueError("Number of points and scalars mismatch")
        out = np.zeros(N.point_limbs(cid, group), dtype=np.uint64)
        if isinstance(points, PointArray) and n_p:
            N.check(lib.zk_msm_plan_run(points.plan(), n_s, sc.ctypes.data, 0, 0, 0, N.u64p(out), None))
        else:
            base = _points_to_limbs(points, cid, group)
            st = lib.zk_msm(cid, group, n_p, n_s, N.u64p(sc), N.u64p(base), N.u64p(out))
            if st == N.ZK_ERR_LENGTH:
                raise ValueError("Number of points and scalars mismatch")
            N.check(st)
        return _point_class(cid, group)._from_limbs(out)

    def multiscalar_mul_g1(points, scalars):
        return _msm(1, points, scalars)

    def multiscalar_mul_g2(points, scalars):
        return _msm(2, points, scalars)

    def _affine_ints(pt, group):
        if pt.is_zero():
            return None
        c = pt._coords()
        return (c[0], c[1]) if group == 1 else ((c[0], c[1]), (c[2], c[3]))

    def pairing(a, b):
        return _pairing.pairing(cid, _affine_ints(a, 1), _affine_ints(b, 2))

    def multi_pairing(a, b):
        return _pairing.multi_pairing(cid, [_affine_ints(p, 1) for p in a], [_affine_ints(q, 2) for q in b])

    mod = types.SimpleNamespace(
        PointG1=PointG1, PointG2=PointG2, PointG12=_pairing.GT, g1=g1, g2=g2,
        batch_multi_scalar_g1=batch_multi_scalar_g1, batch_multi_scalar_g2=batch_multi_scalar_g2,
        multiscalar_mul_g1=multiscalar_mul_g1, multiscalar_mul_g2=multiscalar_mul_g2,
        pairing=pairing, multi_pairing=multi_pairing, curve_id=cid,
    )
    return mod


# =====================================================================================
# polynomials over the scalar field
# =====================================================================================

def _next_pow2(n):
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


def _make_poly(cid):
    r = _MODULUS[cid]

    def _domain_check(size):
        n = _next_pow2(size)
        if n.bit_length() - 1 > _TWO_ADICITY[cid]:
            # EvaluationDomain::new(size).unwrap() panics in the reference (polynomial.rs:48)
            raise ValueError("Domain size is too large")
        return n

    def _ntt(vals, size, inverse, coset, as_limbs=False):
        lib = N.ensure_gpu()
        a = _scalar_limbs(vals, cid)
        n = _domain_check(size)
        out = np.zeros((n, 4), dtype=np.uint64)
        st = lib.zk_ntt(cid, inverse, coset, a.shape[0], N.u64p(a), size, N.u64p(out))
        if st == N.ZK_ERR_DOMAIN:
            raise ValueError("Domain size is too large")
        N.check(st)
        return out if as_limbs else N.limbs_to_ints(out)

    def fft(coeffs, size, as_limbs=False):
        return _ntt(coeffs, size, 0, 0, as_limbs)

    def ifft(evals, size, as_limbs=False):
        return _ntt(evals, size, 1, 0, as_limbs)

    def coset_fft(coeffs, size, as_limbs=False):
        return _ntt(coeffs, size, 0, 1, as_limbs)

    def coset_ifft(evals, size, as_limbs=False):
        return _ntt(evals, size, 1, 1, as_limbs)

    def _vec(op, size, a, b, as_limbs=False):
        lib = N.ensure_gpu()
        la, lb = _scalar_limbs(a, cid), _scalar_limbs(b, cid)
        out = np.zeros((size, 4), dtype=np.uint64)
        if size:
            N.check(lib.zk_vec_op(cid, op, size, la.shape[0], N.u64p(la), lb.shape[0], N.u64p(lb), N.u64p(out)))
        return out if as_limbs else N.limbs_to_ints(out)

    def mul_over_evaluation_domain(size, a, b, as_limbs=False):
        return _vec(0, size, a, b, as_limbs)

    def add_over_evaluation_domain(size, a, b, as_limbs=False):
        if len(a) < size or len(b) < size:
            raise IndexError("index out of range")  # the reference indexes a[i], b[i] for i < size
        return _vec(1, size, a, b, as_limbs)

    def evaluate_vanishing_polynomial(n, tau):
        m = _next_pow2(n)
        if m.bit_length() - 1 > _TWO_ADICITY[cid]:
            raise ValueError("Domain size is too large")
        return (pow(int(tau) % r, m, r) - 1) % r

    def evaluate_lagrange_coefficients(n, tau, as_limbs=False):
        lib = N.load()
        m = _next_pow2(n)
        if m.bit_length() - 1 > _TWO_ADICITY[cid]:
            raise ValueError("Domain size is too large")
        out = np.zeros((m, 4), dtype=np.uint64)
        t = N.ints_to_limbs([int(tau)], 4, r)
        N.check(lib.zk_fr_lagrange_coeffs(cid, m, N.u64p(t), N.u64p(out)))
        return out if as_limbs else N.limbs_to_ints(out)

    def _omega(domain):
        lib = N.load()
        out = np.zeros(4, dtype=np.uint64)
        st = lib.zk_fr_root_of_unity(cid, domain, N.u64p(out))
        if st == N.ZK_ERR_DOMAIN:
            raise ValueError("Domain size is too large")
        N.check(st)
        return N.limbs_to_ints(out.reshape(1, 4))[0]

    def get_evaluation_point(domain, i):
        return pow(_omega(domain), i, r)

    def get_all_evaluation_points(domain):
        w = _omega(domain)
        out, cur = [], 1
        for _ in range(_next_pow2(domain)):
            out.append(cur)
            cur = cur * w % r
        return out

    class Polynomial:
        """Dense univariate / sparse multivariate polynomial with an attached evaluation domain
        (reference src/bn254/polynomial.rs:17-516).  Constructor signature of the pyclass:
        Polynomial(num_vars, [(coeff, [(var, power), ...]), ...], domain_size)."""

        def __init__(self, num_vars, coeff_terms, size):
            self.num_vars = num_vars
            self.domain = _domain_check(size if size else 1)
            if num_vars > 1:
                terms = {}
                for coeff, term in coeff_terms:
                    key = tuple(sorted((v, pw) for v, pw in term if pw))
                    terms[key] = (terms.get(key, 0) + coeff) % r
                self.terms = {k: v for k, v in terms.items() if v}
                self.c = None
            else:
                self.terms = None
                self.c = self._strip([int(c) % r for c, _ in coeff_terms])

        # -- helpers --
        @staticmethod
        def _strip(c):
            while c and c[-1] == 0:
                c.pop()
            return c

        @classmethod
        def _uni(cls, coeffs, domain):
            p = cls.__new__(cls)
            p.num_vars, p.domain, p.terms = 1, domain, None
            p.c = cls._strip(list(coeffs))
            return p

        @classmethod
        def _multi(cls, num_vars, terms, domain):
            p = cls.__new__(cls)
            p.num_vars, p.domain, p.c = num_vars, domain, None
            p.terms = {k: v % r for k, v in terms.items() if v % r}
            return p

        def is_univariate(self):
            return self.c is not None

        def coeffs(self):
            if self.c is not None:
                return list(self.c)
            out = {}
            for key, v in self.terms.items():
                exps = [0] * self.num_vars
                for var, pw in key:
                    exps[var] = pw
                out[tuple(exps)] = v
            return out

        def degree(self):
            if self.c is not None:
                return max(len(self.c) - 1, 0)
            return max((sum(pw for _, pw in k) for k in self.terms), default=0)

        def is_zero(self):
            return len(self.c) == 0 if self.c is not None else len(self.terms) == 0

        def __eq__(self, other):
            if not isinstance(other, Polynomial):
                return False
            if (self.c is None) != (other.c is None):
                return False
            return self.c == other.c if self.c is not None else self.terms == other.terms

        def __hash__(self):
            return hash(tuple(self.c)) if self.c is not None else hash(frozenset(self.terms.items()))

        # -- arithmetic --
        def _combine(self, other, sign):
            if isinstance(other, int):
                k = other * sign
                if self.c is not None:
                    c = list(self.c) or [0]
                    c[0] = (c[0] + k) % r
                    return self._uni(c, self.domain)
                t = dict(self.terms)
                t[()] = (t.get((), 0) + k) % r
                return self._multi(self.num_vars, t, self.domain)
            if not isinstance(other, Polynomial):
                return NotImplemented
            if (self.c is None) != (other.c is None):
                raise TypeError("Can only add same n-variate polynomial")
            if self.c is not None:
                # poly +/- poly is on the Groth16 path (qap.py:66 `uv - w`): element-wise on the GPU
                n = max(len(self.c), len(other.c))
                if n == 0:
                    return self._uni([], self.domain)
                out = _vec(1 if sign > 0 else 2, n, N.ints_to_limbs(self.c, 4), N.ints_to_limbs(other.c, 4))
                return self._uni(out, self.domain)
            t = dict(self.terms)
            for k, v in other.terms.items():
                t[k] = (t.get(k, 0) + sign * v) % r
            return self._multi(self.num_vars, t, self.domain)

        def __add__(self, other):
            return self._combine(other, 1)

        __radd__ = __add__

        def __sub__(self, other):
            return self._combine(other, -1)

        def __neg__(self):
            return self * (r - 1)

        def __mul__(self, other):
            if isinstance(other, int):
                k = other % r
                if self.c is not None:
                    return self._uni([x * k % r for x in self.c], self.domain)
                return self._multi(self.num_vars, {t: v * k for t, v in self.terms.items()}, self.domain)
            if not isinstance(other, Polynomial):
                return NotImplemented
            if (self.c is None) != (other.c is None):
                raise TypeError("Can only multiply same n-variate polynomial")
            if self.c is not None:
                # naive product, like the reference (polynomial.rs:354-358)
                if not self.c or not other.c:
                    return self._uni([], self.domain)
                out = [0] * (len(self.c) + len(other.c) - 1)
                for i, x in enumerate(self.c):
                    if x:
                        for j, y in enumerate(other.c):
                            out[i + j] += x * y
                return self._uni([v % r for v in out], self.domain)
            out = {}
            for ka, va in self.terms.items():
                for kb, vb in other.terms.items():
                    exps = dict(ka)
                    for var, pw in kb:
                        exps[var] = exps.get(var, 0) + pw
                    key = tuple(sorted(exps.items()))
                    out[key] = (out.get(key, 0) + va * vb) % r
            return self._multi(self.num_vars, out, self.domain)

        __rmul__ = __mul__

        def __truediv__(self, other):
            if not isinstance(other, Polynomial) or self.c is None or other.c is None:
                raise TypeError("Can only divide same n-variate polynomial")
            if not other.c:
                raise ZeroDivisionError("division by the zero polynomial")
            rem = list(self.c)
            dl = len(other.c)
            if len(rem) < dl:
                return [self._uni([], self.domain), self._uni(rem, self.domain)]
            inv = pow(other.c[-1], -1, r)
            q = [0] * (len(rem) - dl + 1)
            for i in range(len(q) - 1, -1, -1):
                coef = rem[i + dl - 1] * inv % r
                q[i] = coef
                if coef:
                    for j, d in enumerate(other.c):
                        rem[i + j] = (rem[i + j] - coef * d) % r
            return [self._uni(q, self.domain), self._uni(rem[: dl - 1], self.domain)]

        def multiply_by_vanishing_poly(self):
            if self.c is None:
                raise TypeError("Can only multiply univariate polynomial")
            n = self.domain
            out = [0] * (len(self.c) + n)
            for i, x in enumerate(self.c):
                out[i] = (out[i] - x) % r
                out[i + n] = (out[i + n] + x) % r
            return self._uni(out, self.domain)

        def divide_by_vanishing_poly(self):
            """[q, rem] with self = q * (X^n - 1) + rem (polynomial.rs:466-489); O(n) fold on the GPU."""
            if self.c is None:
                raise TypeError("Can only divide univariate polynomial")
            n = self.domain
            ln = len(self.c)
            if ln == 0:
                return [self._uni([], n), self._uni([], n)]
            lib = N.ensure_gpu()
            la = N.ints_to_limbs(self.c, 4)
            qlen, top = max(ln - n, 0), min(ln, n)
            q = np.zeros((max(qlen, 1), 4), dtype=np.uint64)
            rem = np.zeros((top, 4), dtype=np.uint64)
            flag = N._i(0)
            N.check(lib.zk_poly_div_vanishing(cid, n, ln, N.u64p(la), N.u64p(q), N.u64p(rem), flag))
            return [self._uni(N.limbs_to_ints(q[:qlen]) if qlen else [], n), self._uni(N.limbs_to_ints(rem), n)]

        def __call__(self, point):
            if self.c is not None:
                if not isinstance(point, int):
                    raise TypeError("Univariate polynomial evaluation only accept int")
                acc = 0
                for c in reversed(self.c):
                    acc = (acc * point + c) % r
                return acc
            if not isinstance(point, (list, tuple)):
                raise TypeError("Multivariate polynomial evaluation only accept list of int")
            acc = 0
            for key, v in self.terms.items():
                t = v
                for var, pw in key:
                    t = t * pow(point[var], pw, r) % r
                acc = (acc + t) % r
            return acc

        def __str__(self):
            if self.c is not None:
                parts = []
                for e in range(len(self.c) - 1, -1, -1):
                    c = self.c[e]
                    if c:
                        parts.append(f"{c}x^{e}" if e > 1 else (f"{c}x" if e == 1 else f"{c}"))
                return " + ".join(parts)
            return " + ".join(f"{v}*" + "*".join(f"x_{var}^{pw}" for var, pw in k) if k else f"{v}"
                              for k, v in sorted(self.terms.items()))

        __repr__ = __str__

    Polynomial.__qualname__ = "Polynomial"

    return types.SimpleNamespace(
        Polynomial=Polynomial, fft=fft, ifft=ifft, coset_fft=coset_fft, coset_ifft=coset_ifft,
        add_over_evaluation_domain=add_over_evaluation_domain, mul_over_evaluation_domain=mul_over_evaluation_domain,
        evaluate_vanishing_polynomial=evaluate_vanishing_polynomial,
        evaluate_lagrange_coefficients=evaluate_lagrange_coefficients,
        get_evaluation_point=get_evaluation_point, get_all_evaluation_points=get_all_evaluation_points,
        curve_id=cid, modulus=r,
    )


ec_bn254 = _make_ec(0)
ec_bls12_381 = _make_ec(1)
polynomial_bn254 = _make_poly(0)
polynomial_bls12_381 = _make_poly(1)
