"""Groth16 (https://eprint.iacr.org/2016/260) with the reference's surface
(python/zksnake/groth16/__init__.py:5-6): Groth16, Proof, ProvingKey, VerifyingKey."""

from .protocol import Groth16
from .serialization import Proof, ProvingKey, VerifyingKey

__all__ = ["Groth16", "Proof", "ProvingKey", "VerifyingKey"]
