"""Small helpers with the reference's names (python/zksnake/utils.py:6-78)."""

import os
import secrets
import time


def get_random_int(n_max):
    """uniform integer in [1, n_max] from the OS entropy source"""
    return 1 + secrets.randbelow(n_max)


def get_n_jobs():
    """ZKSNAKE_PARALLEL_CPU knob of the reference; the GPU backend has no use for it but keeps the parse."""
    value = os.environ.get("ZKSNAKE_PARALLEL_CPU")
    return int(value) if value else 1


def split_list(data, n):
    return [data[i:i + n] for i in range(0, len(data), n)]


def next_power_of_two(n):
    return 1 << (n - 1).bit_length()


def is_power_of_two(n):
    return n & (n - 1) == 0


def inner_product(a, b, p):
    return sum(x * y for x, y in zip(a, b)) % p


def batch_modinv(a, m):
    """inverses of all a[i] modulo m with a single modular inversion (Montgomery's trick)"""
    n = len(a)
    prefix = [1] * (n + 1)
    for i, x in enumerate(a):
        prefix[i + 1] = prefix[i] * x % m
    inv = pow(prefix[n], -1, m)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * prefix[i] % m
        inv = inv * a[i] % m
    return out


class Timer:
    def __init__(self, name):
        self.name = name
        self.start_time = self.end_time = 0.0

    def __enter__(self):
        self.start_time = time.time()
        return self

    def __exit__(self, exc_type, exc, tb):
        self.end_time = time.time()
        print(f"{self.name}: {self.end_time - self.start_time:.2f} seconds")
