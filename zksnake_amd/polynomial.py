"""
Polynomial front end with the reference's function names (python/zksnake/polynomial.py:17-237),
dispatching on the field modulus to the BN254 / BLS12-381 scalar-field module of `_algebra`.
All transforms and element-wise products run on the GPU through libzkmi.so.
"""

from . import _algebra
from .constant import BLS12_381_SCALAR_FIELD, BN254_SCALAR_FIELD
from .utils import next_power_of_two

POLY_OBJECT = {
    BN254_SCALAR_FIELD: _algebra.polynomial_bn254,
    BLS12_381_SCALAR_FIELD: _algebra.polynomial_bls12_381,
}


def Polynomial(coeffs, p, domain_size=None):
    """list -> dense univariate (coeffs[i] * x^i); dict {exponent tuple: coeff} -> multivariate."""
    mod = POLY_OBJECT[p]
    size = domain_size or len(coeffs)
    if isinstance(coeffs, list):
        return mod.Polynomial(1, [(c, ()) for c in coeffs], size)
    if isinstance(coeffs, dict):
        num_vars = len(next(iter(coeffs)))
        terms = [(c, [(v, pw) for v, pw in enumerate(exps) if pw]) for exps, c in coeffs.items()]
        return mod.Polynomial(num_vars, terms, size)
    raise TypeError("Coefficients must be in list or dict")


def get_evaluation_point(domain, i, p):
    return 1 if i == 0 else POLY_OBJECT[p].get_evaluation_point(domain, i)


def get_all_evaluation_points(domain, p):
    return POLY_OBJECT[p].get_all_evaluation_points(domain)


def fft(coeffs, p, size=None):
    return POLY_OBJECT[p].fft(coeffs, size or len(coeffs))


def ifft(evals, p, size=None):
    return POLY_OBJECT[p].ifft(evals, size or len(evals))


def coset_fft(coeffs, p, size=None):
    return POLY_OBJECT[p].coset_fft(coeffs, size or len(coeffs))


def coset_ifft(evals, p, size=None):
    return POLY_OBJECT[p].coset_ifft(evals, size or len(evals))


def _pad_coeffs(a, b):
    """zero-pad two coefficient lists to the common length the reference uses before fft
    (polynomial.py:126-148): longer degree d -> both end up with d + 1 + next_pow2(d) entries."""
    da, db = len(a) - 1, len(b) - 1
    extra = next_power_of_two(max(da, db))
    target = max(da, db) + 1 + extra
    return a + [0] * (target - len(a)), b + [0] * (target - len(b))


def mul_over_fft(domain, a, b, p, return_poly=True):
    """product of two coefficient-form polynomials through NTT -> pointwise -> iNTT"""
    pa, pb = _pad_coeffs(a.coeffs(), b.coeffs())
    fa, fb = fft(pa, p), fft(pb, p)
    prod = mul_over_evaluation_domain(len(fa), fa, fb, p)
    if not return_poly:
        return prod
    return Polynomial(ifft(prod, p), p, domain)


def add_over_evaluation_domain(domain, evals, p):
    mod = POLY_OBJECT[p]
    acc = evals[0]
    for nxt in evals[1:]:
        acc = mod.add_over_evaluation_domain(domain, acc, nxt)
    return acc


def mul_over_evaluation_domain(domain, a, b, p):
    return POLY_OBJECT[p].mul_over_evaluation_domain(domain, a, b)


def evaluate_vanishing_polynomial(domain, x, p):
    return POLY_OBJECT[p].evaluate_vanishing_polynomial(domain, x)


def evaluate_lagrange_coefficients(domain, x, p):
    return POLY_OBJECT[p].evaluate_lagrange_coefficients(domain, x)


def barycentric_eval(domain, sparse_eval, x, p):
    omega = get_evaluation_point(domain, 1, p)
    total = 0
    for i, value in sparse_eval.items():
        w_i = pow(omega, i, p)
        total += value * w_i * pow(x - w_i, -1, p)
    return (pow(x, domain, p) - 1) * pow(domain, -1, p) * total % p


def lagrange_interpolation(x, y, p):
    poly = Polynomial([0], p)
    for j, (xj, yj) in enumerate(zip(x, y)):
        term = Polynomial([yj], p)
        for k, xk in enumerate(x):
            if k != j:
                inv = pow(xj - xk, -1, p)
                term = term * Polynomial([(-xk) * inv % p, inv % p], p)
        poly = poly + term
    return poly
