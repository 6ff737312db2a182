"""Vanilla PlonK (https://eprint.iacr.org/2019/953) with the reference's surface
(python/zksnake/plonk/__init__.py): Plonk, Proof, ProvingKey, VerifyingKey."""

from .protocol import Plonk
from .serialization import Proof, ProvingKey, VerifyingKey

__all__ = ["Plonk", "Proof", "ProvingKey", "VerifyingKey"]
