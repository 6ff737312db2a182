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
  * the O(n) sequential pieces (grand product, Horner evaluations, division by X - zeta, the linearisation's
    multiply-adds) are host C++ (zk_fr_*), never per-coefficient Python.
"""

import numpy as np

from .. import _native as N
from ..arithmetization.plonkish import Plonkish
from ..ecc import EllipticCurve, PointArray
from ..frvec import FrOps
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
    def _commit(self, coeffs):
        """<tau_g1[:len], coeffs> on the resident plan of the proving key's tau_g1"""
        pk = self.proving_key
        if not isinstance(pk.tau_g1, PointArray):
            from .._algebra import _points_to_limbs
            pk.tau_g1 = PointArray(self.E.curve.curve_id, 1, _points_to_limbs(pk.tau_g1, self.E.curve.curve_id, 1))
        assert len(coeffs) <= len(pk.tau_g1), "Constraints are too big for the given g1_tau"
        return self.E.multiexp(pk.tau_g1, np.ascontiguousarray(coeffs))

    def setup(self, g1_tau=None, g2_tau=None):
        """universal setup (or reuse of given powers of tau) + circuit preprocessing (protocol.py:39-155)"""
        V, r = self._ops, self.order
        n = self.constraints.length
        if not g1_tau:
            tau = self._tau if self._tau is not None else get_random_int(r - 1)
            powers = [1] * (n + 6)
            for i in range(1, n + 6):
                powers[i] = powers[i - 1] * tau % r
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

    # ------------------------------------------------------------------------------------------
    def _key_columns(self):
        """per-key vectors the prover reuses across proofs: the n-domain labels and the 4n-coset evaluations"""
        pk, V, n, r = self.proving_key, self._ops, self.proving_key.n, self.order
        cache = pk._cache
        if "coset" not in cache:
            roots = V.ntt(V.limbs([0, 1]) if n > 1 else V.limbs([1]), n)
            id2 = V.add(roots, roots)
            cache["id_evals"] = [roots, id2, V.add(id2, roots)]
            if "sigma_evals" not in cache:  # key came from bytes
                cache["sigma_evals"] = [V.ntt(p, n) for p in pk._permutation]
            m = self._quotient_domain()
            g = COSET_SHIFT[V.cid]
            assert pow(g, m, r) != 1
            cache["shift"] = (V.powers(g, m), V.powers(pow(g, -1, r), m))
            x = self._to_coset(V.limbs([0, 1]))                        # the coset points g * omega_4n^i
            # 1 / (x^n - 1) on the coset has period m / n
            xn = [pow(V.int_at(x, i), n, r) for i in range(m // n)]
            zh_inv = np.tile(V.limbs([pow(v - 1, -1, r) for v in xn]), (n, 1))
            cache["coset"] = {
                "x": x, "zh_inv": zh_inv, "l1": self._to_coset(V.const(pow(n, -1, r), n)),
                "q": {k: self._to_coset(pk._selector[k]) for k in SELECTORS},
                "sigma": [self._to_coset(p) for p in pk._permutation],
            }
        return cache

    def _quotient_domain(self):
        """size of the coset the quotient (degree 3n + 5) is interpolated on: 4n, or 8n for the tiniest circuits"""
        n = self.proving_key.n
        return max(4 * n, 1 << (3 * n + 5).bit_length())

    def _to_coset(self, coeffs):
        """evaluations of a coefficient vector on g * H_m"""
        V, m = self._ops, self._quotient_domain()
        fwd = self.proving_key._cache["shift"][0]
        k = coeffs.shape[0]
        return V.ntt(V.mul(np.ascontiguousarray(coeffs), np.ascontiguousarray(fwd[:k])), m)

    def _from_coset(self, evals):
        V, m = self._ops, self._quotient_domain()
        return V.mul(V.ntt(evals, m, inverse=True), self.proving_key._cache["shift"][1])

    def _blind(self, coeffs, n, scalars):
        """coeffs += (s0 + s1 X + ..)(X^n - 1), in place"""
        for i, s in enumerate(scalars):
            self._ops.add_at(coeffs, i, -s)
            self._ops.add_at(coeffs, n + i, s)

    def prove(self, public_witness: dict, private_witness) -> Proof:
        """public_witness: {row: value}; private_witness: flat [a0, b0, c0, a1, ...] as ints or a (3k, 4) limb array."""
        assert self.proving_key, "ProvingKey has not been generated"
        pk, V, r = self.proving_key, self._ops, self.order
        n = pk.n
        m = self._quotient_domain()
        cache = self._key_columns()
        cos = cache["coset"]
        blind = list(self._blinding) if self._blinding is not None else [get_random_int(r - 1) for _ in range(11)]

        wit = V.limbs(private_witness)
        cols = []
        for j in range(3):
            col = V.zeros(n)
            part = wit[j::3]
            col[:part.shape[0]] = part
            cols.append(col)
        pi_evals = V.zeros(n)
        for k, v in public_witness.items():
            pi_evals[k] = V.one(v)[0]

        transcript = FiatShamirTranscript(field=r)
        for k in SELECTORS:
            transcript.append(pk.tau_selector_poly[k])
        for point in pk.tau_permutation_poly:
            transcript.append(point)
        for _, v in public_witness.items():
            transcript.append(v)

        # -- round 1: blinded wire polynomials and their commitments ---------------------------------
        wires = []
        for j in range(3):
            coeffs = V.zeros(n + 2)
            coeffs[:n] = V.ntt(cols[j], n, inverse=True)
            self._blind(coeffs, n, blind[2 * j:2 * j + 2])
            wires.append(coeffs)
        pi_coeffs = V.ntt(pi_evals, n, inverse=True)
        tau_w = [self._commit(w) for w in wires]
        for point in tau_w:
            transcript.append(point)

        # -- round 2: permutation grand product z --------------------------------------------------
        beta = transcript.get_challenge_scalar()
        gamma = transcript.get_challenge_scalar()
        beta_n, gamma_n = V.const(beta, n), V.const(gamma, n)

        def column_product(labels):
            acc = None
            for col, lab in zip(cols, labels):
                term = V.add(V.add(col, V.mul(beta_n, np.ascontiguousarray(lab))), gamma_n)
                acc = term if acc is None else V.mul(acc, term)
            return acc

        acc = V.grand_product(column_product(cache["id_evals"]), column_product(cache["sigma_evals"]))
        assert V.int_at(acc, n) == 1, "Copy constraints are not satisfied"
        z = V.zeros(n + 3)
        z[:n] = V.ntt(np.ascontiguousarray(acc[:n]), n, inverse=True)
        self._blind(z, n, blind[6:9])
        tau_z = self._commit(z)
        transcript.append(tau_z)

        # -- round 3: quotient on the coset g * H_4n -----------------------------------------------
        alpha = transcript.get_challenge_scalar()
        a_e, b_e, c_e = (self._to_coset(w) for w in wires)
        z_e = self._to_coset(z)
        zw_e = np.ascontiguousarray(np.roll(z_e, -(m // n), axis=0))
        pi_e = self._to_coset(pi_coeffs)
        q = cos["q"]
        # the gate identity also has to hold on H itself, else the "quotient" is not a polynomial (the reference
        # asserts a zero remainder, protocol.py:347); checked on the n-domain where it costs 1/4 of a coset pass
        gate_h = V.add(V.add(V.add(V.mul(cols[0], V.limbs(self._column("L"))), V.mul(cols[1], V.limbs(self._column("R")))),
                             V.add(V.mul(cols[2], V.limbs(self._column("O"))), V.mul(V.mul(cols[0], cols[1]), V.limbs(self._column("M"))))),
                       V.add(V.limbs(self._column("C")), pi_evals))
        assert not gate_h.any(), "gate constraints are not satisfied"

        gate = V.add(V.add(V.add(V.mul(a_e, q["L"]), V.mul(b_e, q["R"])), V.add(V.mul(c_e, q["O"]), V.mul(V.mul(a_e, b_e), q["M"]))),
                     V.add(q["C"], pi_e))
        beta_m, gamma_m = V.const(beta, m), V.const(gamma, m)
        bx = V.mul(beta_m, cos["x"])
        bx2 = V.add(bx, bx)
        ag, bg, cg = V.add(a_e, gamma_m), V.add(b_e, gamma_m), V.add(c_e, gamma_m)
        left = V.mul(V.mul(V.mul(V.add(ag, bx), V.add(bg, bx2)), V.add(cg, V.add(bx2, bx))), z_e)
        s = cos["sigma"]
        right = V.mul(V.mul(V.mul(V.add(ag, V.mul(beta_m, s[0])), V.add(bg, V.mul(beta_m, s[1]))), V.add(cg, V.mul(beta_m, s[2]))), zw_e)
        boundary = V.mul(V.sub(z_e, V.const(1, m)), cos["l1"])
        numer = V.add(V.add(gate, V.mul(V.const(alpha, m), V.sub(left, right))), V.mul(V.const(alpha * alpha % r, m), boundary))
        t = self._from_coset(V.mul(numer, cos["zh_inv"]))
        assert not t[3 * n + 6:].any(), "quotient has a remainder"
        t_lo, t_mid, t_hi = V.zeros(n + 1), V.zeros(n + 1), V.zeros(n + 6)
        t_lo[:n], t_mid[:n], t_hi[:] = t[:n], t[n:2 * n], t[2 * n:3 * n + 6]
        V.add_at(t_lo, n, blind[9])
        V.add_at(t_mid, 0, -blind[9])
        V.add_at(t_mid, n, blind[10])
        V.add_at(t_hi, 0, -blind[10])
        tau_t = [self._commit(p) for p in (t_lo, t_mid, t_hi)]
        for point in tau_t:
            transcript.append(point)

        # -- round 4: openings at zeta and the linearisation polynomial ----------------------------
        zeta = transcript.get_challenge_scalar()
        omega = V.int_at(cache["id_evals"][0], 1) if n > 1 else 1
        za, zb, zc = (V.eval(w, zeta) for w in wires)
        zs1, zs2 = V.eval(pk._permutation[0], zeta), V.eval(pk._permutation[1], zeta)
        zzw = V.eval(z, zeta * omega % r)
        zeta_n = pow(zeta, n, r)
        zh_zeta = (zeta_n - 1) % r
        l1_zeta = zh_zeta * pow(n * (zeta - 1) % r, -1, r) % r
        pi_zeta = V.eval(pi_coeffs, zeta)
        f1 = (za + beta * zeta + gamma) * (zb + beta * K1 * zeta + gamma) * (zc + beta * K2 * zeta + gamma) % r
        f2 = (za + beta * zs1 + gamma) * (zb + beta * zs2 + gamma) * zzw % r
        a2l1 = alpha * alpha * l1_zeta % r

        lin = V.zeros(n + 6)
        for k, weight in zip(SELECTORS, (za, zb, zc, za * zb % r, 1)):
            V.scale_add(lin, pk._selector[k], weight)
        V.scale_add(lin, z, (alpha * f1 + a2l1) % r)
        V.scale_add(lin, pk._permutation[2], -alpha * f2 * beta % r)
        for part, weight in ((t_lo, 1), (t_mid, zeta_n), (t_hi, zeta_n * zeta_n % r)):
            V.scale_add(lin, part, -zh_zeta * weight % r)
        V.add_at(lin, 0, pi_zeta - alpha * f2 * (zc + gamma) - a2l1)
        for value in (za, zb, zc, zs1, zs2, zzw):
            transcript.append(value)

        # -- round 5: opening proofs --------------------------------------------------------------------
        v = transcript.get_challenge_scalar()
        vk_pow, shift = 1, 0
        for poly, value in ((wires[0], za), (wires[1], zb), (wires[2], zc), (pk._permutation[0], zs1), (pk._permutation[1], zs2)):
            vk_pow = vk_pow * v % r
            V.scale_add(lin, poly, vk_pow)
            shift += vk_pow * value
        V.add_at(lin, 0, -shift)
        w_zeta, rem = V.div_linear(lin, zeta)
        assert rem == 0
        z_shift = z.copy()
        V.add_at(z_shift, 0, -zzw)
        w_zeta_omega, rem = V.div_linear(z_shift, zeta * omega % r)
        assert rem == 0
        tau_w_zeta, tau_w_zeta_omega = self._commit(w_zeta), self._commit(w_zeta_omega)

        return Proof(tau_w[0], tau_w[1], tau_w[2], tau_z, tau_t[0], tau_t[1], tau_t[2], tau_w_zeta, tau_w_zeta_omega,
                     za, zb, zc, zs1, zs2, zzw)

    def _column(self, k):
        """selector column k on the n-domain (from the circuit when present, else from the key's polynomial)"""
        cache = self.proving_key._cache
        key = "col_" + k
        if key not in cache:
            if self.constraints is not None and self.constraints.qL is not None:
                cache[key] = self._ops.limbs(getattr(self.constraints, "q" + k))
            else:
                cache[key] = self._ops.ntt(self.proving_key._selector[k], self.proving_key.n)
        return cache[key]

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
        lhs = self.E.pairing(proof.tau_W_zeta + proof.tau_W_zeta_omega * u, vk.tau_g2)
        rhs = self.E.pairing(rhs_point, self.E.G2())
        return lhs == rhs
