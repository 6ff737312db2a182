"""
Deterministic synthetic inputs for the hot-path configurations of BASELINE.json:
SplitMix64 streams for scalars / base discrete logs, and the benchmark "chain" circuit of the
reference's benchmarks/benchmark_groth16.py:7-27 built directly as sparse matrices.

Pure numpy / Python ints; no GPU work here.
"""

import numpy as np

from . import constant

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB

SEED_MSM_SCALARS = 0x5EED0002
SEED_MSM_BASES = 0x5EED1002
SEED_NTT = 0x5EED0003
SEED_PROVE = 0x5EED0004


def splitmix64(seed, count, offset=0):
    """`count` consecutive SplitMix64 outputs (as uint64) starting `offset` outputs into the stream."""
    with np.errstate(over="ignore"):
        idx = np.arange(1 + offset, 1 + offset + count, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def field_stream(seed, n, modulus, offset=0):
    """n field elements: 4 consecutive outputs as little-endian limbs, reduced mod `modulus`.
    Returns (limbs (n,4) uint64 canonical < modulus, list of Python ints)."""
    raw = splitmix64(seed, 4 * n, 4 * offset).reshape(n, 4)
    data = raw.tobytes()
    ints = [int.from_bytes(data[32 * i:32 * i + 32], "little") % modulus for i in range(n)]
    buf = b"".join(v.to_bytes(32, "little") for v in ints)
    limbs = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).copy()
    return limbs, ints


def powers_of_two_scalars(n, modulus):
    """skew variant of SURVEY 8d: scalars 2^(i mod 254) (benchmark-witness-like single-bit scalars)."""
    ints = [pow(2, i % 254, modulus) for i in range(n)]
    buf = b"".join(v.to_bytes(32, "little") for v in ints)
    return np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).copy(), ints


def chain_circuit(n, modulus, inp=2):
    """The reference benchmark circuit with n constraints (n >= 2):
         v0 = inp*inp, v_i = v_{i-1}*inp (i < n-1), out = v_{n-2}
       wires [1, out, inp, v0 .. v_{n-2}], n_public = 2.
       Returns (A, B, C) as (rows, cols, vals) int64/object arrays in triplet form, the full
       witness as a list of ints, n_col."""
    nv = n - 1
    rows = np.arange(n, dtype=np.int64)
    a_cols = np.empty(n, dtype=np.int64)
    a_cols[0] = 2
    a_cols[1:nv] = 3 + np.arange(nv - 1)
    a_cols[n - 1] = 3 + nv - 1
    b_cols = np.full(n, 2, dtype=np.int64)
    b_cols[n - 1] = 0
    c_cols = np.empty(n, dtype=np.int64)
    c_cols[:nv] = 3 + np.arange(nv)
    c_cols[n - 1] = 1
    w = [1, 0, inp % modulus]
    cur = inp % modulus
    for _ in range(nv):
        cur = cur * inp % modulus
        w.append(cur)
    w[1] = cur
    ones = [1] * n
    A = (rows.tolist(), a_cols.tolist(), ones)
    B = (rows.tolist(), b_cols.tolist(), ones)
    C = (rows.tolist(), c_cols.tolist(), ones)
    return A, B, C, w, 3 + nv


def plonk_chain_gates(n, modulus, inp=2):
    """PlonK form of the same chain (n a power of two >= 2): rows i < n-1 are a*b - c = 0 with a_0 = b_i = inp,
    a_{i+1} = c_i; row n-1 is a - out = 0 with `out` public.  Returns (selector columns dict L,R,O,M,C,
    slot permutation over [a | b | c], public dict {row: -out}, flat private witness [a0, b0, c0, a1, ..])."""
    assert n >= 2 and n & (n - 1) == 0
    r = modulus
    vals, cur = [], inp % r
    for _ in range(n - 1):
        cur = cur * inp % r
        vals.append(cur)
    a = [inp % r] + vals[:-1] + [vals[-1]]
    b = [inp % r] * (n - 1) + [0]
    c = vals + [0]
    gates = {"L": [0] * (n - 1) + [1], "R": [0] * n, "O": [r - 1] * (n - 1) + [0], "M": [1] * (n - 1) + [0], "C": [0] * n}
    perm = np.arange(3 * n, dtype=np.int64)
    ring = np.concatenate([[0], n + np.arange(n - 1)])        # a_0 and every live b_i carry `inp`
    perm[ring] = np.roll(ring, -1)
    i = np.arange(n - 1)
    perm[2 * n + i], perm[i + 1] = i + 1, 2 * n + i           # c_i <-> a_{i+1}
    private = [x for row in zip(a, b, c) for x in row]
    return gates, perm.tolist(), {n - 1: (-vals[-1]) % r}, private


def scalar_field(curve):
    return constant.BN254_SCALAR_FIELD if curve in ("BN254", "BN128", "ALT_BN128") else constant.BLS12_381_SCALAR_FIELD
