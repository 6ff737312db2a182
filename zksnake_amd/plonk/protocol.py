"""
Vanilla PlonK setup / prove / verify with the reference's class surface
(python/zksnake/plonk/protocol.py:20-647): same transcript, same proof elements, same byte layouts, so with
the blinding scalars pinned the proof bytes equal the reference prover's.

How the prover is organised here (the reference multiplies coefficient-form polynomials pairwise through
2n..8n FFTs and divides by X^n - 1 in coefficient form, protocol.py:213-385):

  * every polynomial is a limb array; transforms, element-wise products and the nine commitments (MSMs over a
    device-resident tau_g1 plan) run on the GPU through libzkmi.so;
  * the quotient T is obtained on the coset g*H_4n: all inputs are evaluated there once (the key's selector /
    permutation / L1 columns are cached), the numerator
        gate + alpha (prod(w_j + beta id_j + gamma) z - prod(w_j + beta sigma_j + gamma) z(omega X)) + alpha^2 (z - 1) L1
    is formed element-wise and multiplied by 1 / (X^n - 1) -- which takes only four distinct values on that coset --
    and one inverse coset transform gives T.  T is the unique quotient, so it equals the reference's;
    (the reference's own `coset_fft` offsets by the domain generator, which maps H_4n onto itself, so the coset here is
    formed explicitly: coefficients are scaled by g^i, g the field's multiplicative generator, before a plain NTT);
  * z(omega X) on the coset is z's evaluation vector rotated by 4n / n = four places; no second transform;
  * vectors stay in HBM for the whole proof (the witness goes up once; only single coefficients and field elements
    come down): the grand product is two product scans + one inversion, evaluations at zeta are a device reduction,
    the linearisation a chain of device multiply-adds, division by X - zeta a rescale + suffix-sum scan + rescale,
    and the commitments read their scalars in place.
"""

import numpy as np

from .. import _native as N
from ..arithmetization.plonkish import Plonkish
from ..ecc import EllipticCurve, PointArray
from ..frvec import DevVec, FrOps
from ..transcript import FiatShamirTranscript
from ..utils import get_random_int
from .serialization import SELECTORS, Proof, ProvingKey, VerifyingKey

K1, K2 = 2, 3  # coset representatives of the wire columns b and c (protocol.py:65-66)
COSET_SHIFT = {N.CURVE_BN254: 5, N.CURVE_BLS12_381: 7}  # multiplicative generators of the scalar fields


class Plonk:
    def __init__(self, constraints: Plonkish, curve: str = "BN254"):
        self.E = EllipticCurve(curve)
        self.order = self.E.order
        self.constraints = constraints
        self.G1_tau = None
        self.G2_tau = None
        self.label = "PlonK"
        self.proving_key = None
        self.verifying_key = None
        self._ops = FrOps(self.order)
        self._tau = None       # tests may pin the trapdoor
        self._blinding = None  # tests may pin the 11 blinding scalars, in the order the reference draws them
        self.last_timings = {}

    # ------------------------------------------------------------------------------------------
    def _tau_plan(self, slot=0):
        """a resident fixed-base MSM plan over the proving key's tau_g1 (all commitments of a proof use these); slots are
        independent workspaces, so the commitments of one round can be in flight together"""
        pk, cid = self.proving_key, self.E.curve.curve_id
        if not isinstance(pk.tau_g1, PointArray):
            from .._algebra import _points_to_limbs
            pk.tau_g1 = PointArray(cid, 1, _points_to_limbs(pk.tau_g1, cid, 1))
        # tau_g1 holds ~4n points but a commitment multiplies n + 2 .. 3n + 5 of them: below 2^20 gates the 16-bit windows
        # win (the wider windows the library would pick for the table's size pay only from ~2^20 scalars per MSM)
        return pk.tau_g1.plan(slot, precompute=True, window_bits=16 if pk.n < (1 << 20) else 0, concurrent=True)

    def _commit_many(self, jobs, meanwhile=None):
        """jobs: [(device vector, count, offset)] -> commitments; the MSMs of one round run concurrently on their plans'
        own streams (the latency-bound reduction of one hides behind the accumulation of the others).
        meanwhile: host work that does not need the results, done while the GPU runs them."""
        lib, cid = N.load(), self.E.curve.curve_id
        from .._algebra import _point_class
        handles = []
        for slot, (vec, count, offset) in enumerate(jobs):
            assert count <= len(self.proving_key.tau_g1), "Constraints are too big for the given g1_tau"
            h = self._tau_plan(slot)
            N.check(lib.zk_msm_plan_enqueue(h, count, vec.ptr(offset), 1, 0, 0, N.STREAM_PLAN))
            handles.append(h)
        if meanwhile is not None:
            meanwhile()
        points = []
        for h in handles:
            out = np.zeros(N.point_limbs(cid, 1), dtype=np.uint64)
            N.check(lib.zk_msm_plan_finish(h, N.u64p(out)))
            points.append(_point_class(cid, 1)._from_limbs(out))
        return points

    def _run_plan(self, count, scalars_ptr, on_device):
        cid = self.E.curve.curve_id
        assert count <= len(self.proving_key.tau_g1), "Constraints are too big for the given g1_tau"
        out = np.zeros(N.point_limbs(cid, 1), dtype=np.uint64)
        if count:
            N.check(N.load().zk_msm_plan_run(self._tau_plan(), count, scalars_ptr, on_device, 0, 0, N.u64p(out), None))
        from .._algebra import _point_class
        return _point_class(cid, 1)._from_limbs(out)

    def _commit(self, coeffs):
        """<tau_g1[:len], coeffs> for a coefficient vector in host memory"""
        coeffs = np.ascontiguousarray(coeffs)
        return self._run_plan(coeffs.shape[0], coeffs.ctypes.data, 0)

    def _commit_dev(self, vec, count, offset=0):
        """<tau_g1[:count], vec[offset:offset+count]> with the scalars read in place from HBM"""
        return self._run_plan(count, vec.ptr(offset), 1)

    def setup(self, g1_tau=None, g2_tau=None, prepare_prover=True):
        """universal setup (or reuse of given powers of tau) + circuit preprocessing (protocol.py:39-155).
        prepare_prover: also build what every proof reuses (the key's coset evaluations in HBM, the workspaces of the
        commitments that run side by side), so that the first proof costs what every later one does."""
        V, r = self._ops, self.order
        n = self.constraints.length
        if not g1_tau:
            tau = self._tau if self._tau is not None else get_random_int(r - 1)
            powers = V.d_powers(tau, n + 6).download(n + 6)   # (1, tau, tau^2, ..) as limbs, computed on the device
            self.G1_tau = self.E.batch_mul(self.E.G1(), powers, as_array=True)
            self.G2_tau = self.E.G2() * tau
        else:
            assert len(g1_tau) >= n + 6, "Constraints are too big for the given g1_tau"
            self.G1_tau, self.G2_tau = g1_tau, g2_tau

        # identity / permuted slot labels: omega^i, k1 omega^i, k2 omega^i addressed through the copy permutation
        roots = V.ntt(V.limbs([0, 1]) if n > 1 else V.limbs([1]), n)   # NTT of X = (omega^i)
        id2 = V.add(roots, roots)
        ids = np.concatenate([roots, id2, V.add(id2, roots)])
        sigma = ids[np.asarray(self.constraints.permutation, dtype=np.int64)].reshape(3, n, 4)
        c = self.constraints
        selector = {k: V.ntt(V.limbs(col), n, inverse=True) for k, col in zip(SELECTORS, (c.qL, c.qR, c.qO, c.qM, c.qC))}
        perm_poly = [V.ntt(np.ascontiguousarray(sigma[j]), n, inverse=True) for j in range(3)]
        id_poly = [V.ntt(np.ascontiguousarray(ids[j * n:(j + 1) * n]), n, inverse=True) for j in range(3)]
        selector_eval = {k: V.ntt(selector[k], 4 * n) for k in SELECTORS}
        l1 = V.const(pow(n, -1, r), n)                                 # iNTT of (1, 0, .., 0)
        lagrange_evals = V.ntt(l1, 4 * n)

        pk = ProvingKey(n, self.G1_tau, selector, selector_eval, perm_poly, id_poly, {}, [], lagrange_evals, self.E.name)
        self.proving_key = pk
        pk.tau_selector_poly = {k: self._commit(selector[k]) for k in SELECTORS}
        pk.tau_permutation_poly = [self._commit(p) for p in perm_poly]
        pk._cache["sigma_evals"] = sigma
        self.verifying_key = VerifyingKey(n, self.G2_tau, pk.tau_selector_poly, pk.tau_permutation_poly, self.E.name)
        if prepare_prover:
            self._key_columns()
            for slot in range(1, 3):   # a round commits to at most three polynomials at once
                self._tau_plan(slot)

    # ------------------------------------------------------------------------------------------
    def _quotient_domain(self):
        """size of the coset the quotient (degree 3n + 5) is interpolated on: 4n, or 8n / 16n for the tiniest circuits"""
        n = self.proving_key.n
        return max(4 * n, 1 << (3 * n + 5).bit_length())

    def _key_columns(self):
        """per-key vectors the prover reuses across proofs, resident in HBM: coefficient vectors, the n-domain
        columns and labels, and the coset evaluations the quotient kernel reads"""
        pk, V, n, r = self.proving_key, self._ops, self.proving_key.n, self.order
        cache = pk._cache
        if "dev" not in cache:
            m = self._quotient_domain()
            g = COSET_SHIFT[V.cid]
            assert pow(g, m, r) != 1
            roots = V.ntt(V.limbs([0, 1]) if n > 1 else V.limbs([1]), n)
            id2 = V.add(roots, roots)
            sigma_evals = cache.get("sigma_evals")
            if sigma_evals is None:  # key came from bytes
                sigma_evals = [V.ntt(p, n) for p in pk._permutation]
            dev = {
                "m": m, "omega": V.int_at(roots, 1) if n > 1 else 1,
                "shift": (V.d_from(V.powers(g, m)), V.d_from(V.powers(pow(g, -1, r), m))),
                "id_labels": [V.d_from(v) for v in (roots, id2, V.add(id2, roots))],
                "sigma_labels": [V.d_from(np.ascontiguousarray(v)) for v in sigma_evals],
                "q_coeffs": {k: V.d_from(pk._selector[k], n) for k in SELECTORS},
                "sigma_coeffs": [V.d_from(p, n) for p in pk._permutation],
                "q_columns": {k: V.d_from(self._column(k)) for k in SELECTORS},
            }
            cache["dev"] = dev
            x = self._to_coset(V.d_from(V.limbs([0, 1])), 2)                # the coset points g * omega_m^i
            xs = x.download(m // n)
            dev["zh_inv"] = V.limbs([pow(pow(V.int_at(xs, i), n, r) - 1, -1, r) for i in range(m // n)])  # period m / n
            dev["x"] = x
            dev["l1"] = self._to_coset(V.d_from(V.const(pow(n, -1, r), n)), n)
            dev["q"] = {k: self._to_coset(dev["q_coeffs"][k], n) for k in SELECTORS}
            dev["sigma"] = [self._to_coset(p, n) for p in dev["sigma_coeffs"]]
        return cache["dev"]

    def _to_coset(self, coeffs, count):
        """evaluations on g * H_m of the polynomial held in the first `count` entries of a device vector"""
        V, dev = self._ops, self.proving_key._cache["dev"]
        out = DevVec(dev["m"])
        V.d_mul(count, coeffs.ptr(), dev["shift"][0].ptr(), out.ptr())
        V.d_ntt(out, dev["m"])
        return out

    def _blind(self, vec, n, scalars):
        """vec += (s0 + s1 X + ..)(X^n - 1) on a device coefficient vector: one launch for its four to six coefficient updates"""
        self._ops.d_lincomb(vec, at=[(i, -s) for i, s in enumerate(scalars)] + [(n + i, s) for i, s in enumerate(scalars)])

    def prove(self, public_witness: dict, private_witness) -> Proof:
        """public_witness: {row: value}; private_witness: flat [a0, b0, c0, a1, ...] as ints or a (3k, 4) limb array."""
        assert self.proving_key, "ProvingKey has not been generated"
        pk, V, r = self.proving_key, self._ops, self.order
        n = pk.n
        dev = self._key_columns()
        m = dev["m"]
        blind = list(self._blinding) if self._blinding is not None else [get_random_int(r - 1) for _ in range(11)]

        wit = V.limbs(private_witness)
        rows = -(-wit.shape[0] // 3)
        assert rows <= n, "witness is longer than the circuit"

        transcript = FiatShamirTranscript(field=r)
        for k in SELECTORS:
            transcript.append(pk.tau_selector_poly[k])
        for point in pk.tau_permutation_poly:
            transcript.append(point)
        for _, v in public_witness.items():
            transcript.append(v)

        # the flat witness goes up once and is de-interleaved on the GPU into the three n-domain columns (gate check, grand
        # product); the coefficient vectors are m entries long so the same buffers later feed the coset transforms
        wit_d = V.d_from(wit, 3 * rows)
        if isinstance(private_witness, np.ndarray):  # int lists are reduced by V.limbs; limb arrays here (Fr::from)
            N.check(N.load().zk_vec_canon_dev(V.cid, 3 * rows, wit_d.ptr(), None))
        col_d = []
        for j in range(3):
            col = DevVec(n, zero=rows < n)
            # `rows` entries for EVERY column: wit_d is zero-padded to 3 * rows, so a witness whose length is not a
            # multiple of three gets zeros in the missing b / c slots of its last row, as the reference pads a, b, c
            # (plonk/protocol.py:167-169) -- never stale data of a pooled buffer
            V.d_gather(rows, wit_d.ptr(), 3, j, col.ptr())
            col_d.append(col)
        pi_d = DevVec(n)
        for k, v in public_witness.items():
            pi_d.upload(V.one(v), k)
        qc = dev["q_columns"]
        # the gate identity has to hold on H itself, else the "quotient" is not a polynomial (the reference asserts a
        # zero remainder, protocol.py:347)
        g1, g2 = DevVec(n, zero=False), DevVec(n, zero=False)
        V.d_mul(n, col_d[0].ptr(), qc["L"].ptr(), g1.ptr())
        for wire, sel in ((col_d[1], qc["R"]), (col_d[2], qc["O"])):
            V.d_mul(n, wire.ptr(), sel.ptr(), g2.ptr())
            V.d_add(n, g1.ptr(), g2.ptr(), g1.ptr())
        V.d_mul(n, col_d[0].ptr(), col_d[1].ptr(), g2.ptr())
        V.d_mul(n, g2.ptr(), qc["M"].ptr(), g2.ptr())
        V.d_add(n, g1.ptr(), g2.ptr(), g1.ptr())
        V.d_add(n, g1.ptr(), qc["C"].ptr(), g1.ptr())
        V.d_add(n, g1.ptr(), pi_d.ptr(), g1.ptr())
        assert V.d_is_zero(n, g1.ptr()), "gate constraints are not satisfied"

        # -- round 1: blinded wire polynomials and their commitments ---------------------------------
        wires = []
        for j in range(3):
            w = DevVec(m)
            V.d_copy(n, col_d[j].ptr(), w.ptr())
            V.d_ntt(w, n, inverse=True)
            self._blind(w, n, blind[2 * j:2 * j + 2])
            wires.append(w)
        pi_c = DevVec(m)
        V.d_copy(n, pi_d.ptr(), pi_c.ptr())
        V.d_ntt(pi_c, n, inverse=True)
        tau_w = self._commit_many([(w, n + 2, 0) for w in wires])
        for point in tau_w:
            transcript.append(point)

        # -- round 2: permutation grand product z --------------------------------------------------
        beta = transcript.get_challenge_scalar()
        gamma = transcript.get_challenge_scalar()
        wire_ptrs = [c.ptr() for c in col_d]
        V.d_perm_terms(n, wire_ptrs, [v.ptr() for v in dev["id_labels"]], beta, gamma, g1.ptr())
        V.d_perm_terms(n, wire_ptrs, [v.ptr() for v in dev["sigma_labels"]], beta, gamma, g2.ptr())
        acc = DevVec(n + 1, zero=False)
        V.d_grand_product(n, g1.ptr(), g2.ptr(), acc.ptr())
        assert V.int_at(acc.download(1, n), 0) == 1, "Copy constraints are not satisfied"
        z = DevVec(m)
        V.d_copy(n, acc.ptr(), z.ptr())
        V.d_ntt(z, n, inverse=True)
        self._blind(z, n, blind[6:9])
        tau_z = self._commit_dev(z, n + 3)
        transcript.append(tau_z)

        # -- round 3: quotient on the coset g * H_m ------------------------------------------------
        alpha = transcript.get_challenge_scalar()
        a_e, b_e, c_e = (self._to_coset(w, n + 2) for w in wires)
        z_e = self._to_coset(z, n + 3)
        pi_e = self._to_coset(pi_c, n)
        q, sg = dev["q"], dev["sigma"]
        t = DevVec(m, zero=False)
        V.d_quotient(m, n, [a_e.ptr(), b_e.ptr(), c_e.ptr(), z_e.ptr(), pi_e.ptr(), q["L"].ptr(), q["R"].ptr(), q["O"].ptr(), q["M"].ptr(),
                            q["C"].ptr(), sg[0].ptr(), sg[1].ptr(), sg[2].ptr(), dev["x"].ptr(), dev["l1"].ptr()],
                     dev["zh_inv"], beta, gamma, alpha, t.ptr())
        V.d_ntt(t, m, inverse=True)
        V.d_mul(m, t.ptr(), dev["shift"][1].ptr(), t.ptr())
        assert m == 3 * n + 6 or V.d_is_zero(m - (3 * n + 6), t.ptr(3 * n + 6)), "quotient has a remainder"
        # T = t_lo + X^n t_mid + X^2n t_hi with t_lo += b9 X^n, t_mid += -b9 + b10 X^n, t_hi += -b10: the blinding terms
        # are added on the group side (tau_g1[0] = G, tau_g1[n] = tau^n G)
        G, Gn = pk.tau_g1[0], pk.tau_g1[n]
        fixed = []   # the four blinding multiples need the key and the blinding scalars only: computed while the MSMs run

        def blinding_terms():
            fixed.extend([Gn * blind[9], G * ((-blind[9]) % r) + Gn * blind[10], G * ((-blind[10]) % r)])

        t_lo, t_mid, t_hi = self._commit_many([(t, n, 0), (t, n, n), (t, n + 6, 2 * n)], meanwhile=blinding_terms)
        tau_t = [t_lo + fixed[0], t_mid + fixed[1], t_hi + fixed[2]]
        for point in tau_t:
            transcript.append(point)

        # -- round 4: openings at zeta and the linearisation polynomial ----------------------------
        zeta = transcript.get_challenge_scalar()
        omega = dev["omega"]
        sc = dev["sigma_coeffs"]
        za, zb, zc, zs1, zs2, zzw, pi_zeta = V.d_eval_many(
            [(n + 2, w.ptr(), zeta) for w in wires] + [(n, sc[0].ptr(), zeta), (n, sc[1].ptr(), zeta),
                                                       (n + 3, z.ptr(), zeta * omega % r), (n, pi_c.ptr(), zeta)])
        zeta_n = pow(zeta, n, r)
        zh_zeta = (zeta_n - 1) % r
        l1_zeta = zh_zeta * pow(n * (zeta - 1) % r, -1, r) % r
        f1 = (za + beta * zeta + gamma) * (zb + beta * K1 * zeta + gamma) * (zc + beta * K2 * zeta + gamma) % r
        f2 = (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) * zzw % r
        a2l1 = alpha * alpha * l1_zeta % r

        lin = DevVec(n + 6)
        w_mid, w_hi = zh_zeta * zeta_n % r, zh_zeta * zeta_n * zeta_n % r
        # the ten terms of the linearisation polynomial in ONE launch (they were ten launches of a few microseconds each, with the
        # GPU idle between them: the section was launch-bound)
        V.d_lincomb(lin, [(n, weight, dev["q_coeffs"][k].ptr()) for k, weight in zip(SELECTORS, (za, zb, zc, za * zb % r, 1))]
                    + [(n + 3, alpha * f1 + a2l1, z.ptr()), (n, -alpha * f2 * beta, sc[2].ptr()),
                       (n, -zh_zeta, t.ptr()), (n, -w_mid, t.ptr(n)), (n + 6, -w_hi, t.ptr(2 * n))])
        for value in (za, zb, zc, zs1, zs2, zzw):
            transcript.append(value)

        # -- round 5: opening proofs --------------------------------------------------------------------
        v = transcript.get_challenge_scalar()
        vk_pow, shift, opening_terms = 1, 0, []
        for poly, count, value in ((wires[0], n + 2, za), (wires[1], n + 2, zb), (wires[2], n + 2, zc), (sc[0], n, zs1), (sc[1], n, zs2)):
            vk_pow = vk_pow * v % r
            opening_terms.append((count, vk_pow, poly.ptr()))
            shift += vk_pow * value
        # ... and with them the constant and X^n corrections: PI(zeta), the sigma_3 / L1 constants, the quotient blinding, the
        # opening values -- one launch for the five terms and the two coefficients
        V.d_lincomb(lin, opening_terms,
                    at=[(0, pi_zeta - alpha * f2 * (zc + gamma) - a2l1 + w_mid * blind[9] + w_hi * blind[10] - shift),
                        (n, -zh_zeta * blind[9] - w_mid * blind[10])])
        quot, quot_w = DevVec(n + 5, zero=False), DevVec(n + 2, zero=False)
        assert V.d_div_linear(n + 6, lin.ptr(), zeta, quot.ptr()) == 0
        V.d_add_at(z, 0, -zzw)
        assert V.d_div_linear(n + 3, z.ptr(), zeta * omega % r, quot_w.ptr()) == 0
        tau_w_zeta, tau_w_zeta_omega = self._commit_many([(quot, n + 5, 0), (quot_w, n + 2, 0)])

        return Proof(tau_w[0], tau_w[1], tau_w[2], tau_z, tau_t[0], tau_t[1], tau_t[2], tau_w_zeta, tau_w_zeta_omega,
                     za, zb, zc, zs1, zs2, zzw)

    def _column(self, k):
        """selector column k on the n-domain (from the circuit when present, else from the key's polynomial)"""
        if self.constraints is not None and self.constraints.qL is not None:
            return self._ops.limbs(getattr(self.constraints, "q" + k))
        return self._ops.ntt(self.proving_key._selector[k], self.proving_key.n)

    # ------------------------------------------------------------------------------------------
    def _recompute_challenges(self, proof: Proof, public_input: dict):
        vk = self.verifying_key
        transcript = FiatShamirTranscript(field=self.order)
        for k in SELECTORS:
            transcript.append(vk.tau_selector_poly[k])
        for point in vk.tau_permutation_poly:
            transcript.append(point)
        for _, v in public_input.items():
            transcript.append(v)
        for point in (proof.tau_a, proof.tau_b, proof.tau_c):
            transcript.append(point)
        beta = transcript.get_challenge_scalar()
        gamma = transcript.get_challenge_scalar()
        transcript.append(proof.tau_z)
        alpha = transcript.get_challenge_scalar()
        for point in (proof.tau_t_lo, proof.tau_t_mid, proof.tau_t_hi):
            transcript.append(point)
        zeta = transcript.get_challenge_scalar()
        for value in (proof.zeta_a, proof.zeta_b, proof.zeta_c, proof.zeta_sigma1, proof.zeta_sigma2, proof.zeta_omega):
            transcript.append(value)
        v = transcript.get_challenge_scalar()
        transcript.append(proof.tau_W_zeta)
        transcript.append(proof.tau_W_zeta_omega)
        u = transcript.get_challenge_scalar()
        return beta, gamma, alpha, zeta, v, u

    def verify(self, proof: Proof, public_input: dict):
        """the verifier of protocol.py:540-647: one batched KZG opening check with two pairings"""
        assert self.verifying_key, "VerifyingKey has not been generated"
        from ..polynomial import barycentric_eval, get_evaluation_point
        vk, r = self.verifying_key, self.order
        n = vk.n
        beta, gamma, alpha, zeta, v, u = self._recompute_challenges(proof, public_input)
        omega = get_evaluation_point(n, 1, r)
        zeta_n = pow(zeta, n, r)
        zh_zeta = (zeta_n - 1) % r
        l1_zeta = barycentric_eval(n, {0: 1}, zeta, r)
        pi_zeta = barycentric_eval(n, public_input, zeta, r)
        za, zb, zc, zs1, zs2, zzw = proof.zeta_a, proof.zeta_b, proof.zeta_c, proof.zeta_sigma1, proof.zeta_sigma2, proof.zeta_omega
        a2l1 = alpha * alpha * l1_zeta % r
        s12 = (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) % r
        r0 = (pi_zeta - a2l1 - s12 * (zc + gamma) * zzw * alpha) % r

        Q, S = vk.tau_selector_poly, vk.tau_permutation_poly
        z_weight = ((za + beta * zeta + gamma) * (zb + beta * K1 * zeta + gamma) * (zc + beta * K2 * zeta + gamma) * alpha + a2l1 + u) % r
        # F - E as one multi-scalar sum; scalars reduced mod r, negatives as r - x
        terms = [
            (Q["M"], za * zb), (Q["L"], za), (Q["R"], zb), (Q["O"], zc), (Q["C"], 1),
            (proof.tau_z, z_weight), (S[2], -s12 * alpha * beta * zzw),
            (proof.tau_t_lo, -zh_zeta), (proof.tau_t_mid, -zh_zeta * zeta_n), (proof.tau_t_hi, -zh_zeta * zeta_n * zeta_n),
            (proof.tau_a, v), (proof.tau_b, pow(v, 2, r)), (proof.tau_c, pow(v, 3, r)), (S[0], pow(v, 4, r)), (S[1], pow(v, 5, r)),
            (self.E.G1(), r0 - (v * za + pow(v, 2, r) * zb + pow(v, 3, r) * zc + pow(v, 4, r) * zs1 + pow(v, 5, r) * zs2 + u * zzw)),
            (proof.tau_W_zeta, zeta), (proof.tau_W_zeta_omega, u * zeta * omega),
        ]
        rhs_point = terms[0][0] * (terms[0][1] % r)
        for point, scalar in terms[1:]:
            rhs_point = rhs_point + point * (scalar % r)
        # e(W_zeta + u W_zeta_omega, tau G2) == e(rhs, G2)  (protocol.py:640-647), checked as a product equal to one:
        # one multi-pairing, one final exponentiation
        lhs_point = proof.tau_W_zeta + proof.tau_W_zeta_omega * u
        return self.E.multi_pairing([lhs_point, -rhs_point], [vk.tau_g2, self.E.G2()]).is_one()
