"""R1CS container of the proving path (reference python/zksnake/arithmetization/__init__.py:9-11 re-exports
R1CS; the symbolic constraint DSL `Var`/`ConstraintSystem` of the reference is a front end outside the
accelerated path)."""

from .plonkish import Plonkish
from .r1cs import R1CS

__all__ = ["R1CS", "Plonkish"]
