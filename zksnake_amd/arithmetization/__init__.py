"""R1CS container of the proving path (reference python/zksnake/arithmetization/__init__.py:9-11 re-exports
R1CS and Var; the symbolic constraint DSL of the reference -- expressions over `Var`, `ConstraintSystem` -- is a front end
outside the accelerated path, see SURVEY.md section 2: `Var` here only names a variable, e.g. the target of a hint)."""

from .plonkish import Plonkish
from .r1cs import R1CS, Var

__all__ = ["R1CS", "Plonkish", "Var"]
