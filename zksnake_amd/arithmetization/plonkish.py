"""
PlonK gate/copy-constraint container with the fields of the reference's `Plonkish`
(python/zksnake/arithmetization/plonkish.py:10-136): selector columns qL, qR, qO, qM, qC padded to a power
of two, the copy permutation over the 3n wire slots [a | b | c], `is_sat`.

The reference fills these from its Rust symbolic front end (`ConstraintSystem.compile_to_plonkish`), which is
outside this backend's path; here the columns are given directly with `from_gates`.
"""

from ..ecc import EllipticCurve
from ..utils import next_power_of_two


class Plonkish:
    def __init__(self, cs=None, curve: str = "BN254"):
        if cs is not None:
            raise NotImplementedError("compiling a symbolic ConstraintSystem is not part of this backend; "
                                      "build the gate columns with Plonkish.from_gates")
        self.constraint_system = None
        self.unpadded_length = 0
        self.length = 0
        self.qL = self.qR = self.qO = self.qM = self.qC = None
        self.witness_map = []
        self.permutation = []
        self.curve = curve
        self.p = EllipticCurve(curve).order

    @classmethod
    def from_gates(cls, qL, qR, qO, qM, qC, permutation, curve: str = "BN254"):
        """columns of equal length (one entry per gate) and the slot permutation of length 3 * next_pow2(len):
        slot i of [a | b | c] must carry the same value as slot permutation[i]"""
        self = cls(None, curve)
        size = len(qL)
        if not (len(qR) == len(qO) == len(qM) == len(qC) == size):
            raise ValueError("selector columns differ in length")
        self.unpadded_length = size
        self.length = next_power_of_two(size)
        pad = lambda col: [int(v) % self.p for v in col] + [0] * (self.length - size)  # noqa: E731
        self.qL, self.qR, self.qO, self.qM, self.qC = pad(qL), pad(qR), pad(qO), pad(qM), pad(qC)
        if sorted(permutation) != list(range(3 * self.length)):
            raise ValueError("permutation must be a bijection of the 3n wire slots")
        self.permutation = list(permutation)
        return self

    def compile(self):
        raise NotImplementedError("no symbolic front end here; use Plonkish.from_gates")

    def solve(self, inputs: dict) -> dict:
        raise NotImplementedError("no symbolic front end here; supply the witness")

    def is_sat(self, public_witness: dict, private_witness: list):
        """gate and copy constraints against the flat witness [a0, b0, c0, a1, ...] (plonkish.py:94-123)"""
        a, b, c = list(private_witness[::3]), list(private_witness[1::3]), list(private_witness[2::3])
        for i in range(self.unpadded_length):
            pi = public_witness.get(i, None) or 0
            g = self.qL[i] * a[i] + self.qR[i] * b[i] + self.qM[i] * (a[i] * b[i]) + self.qO[i] * c[i] + (self.qC[i] + pi)
            if g % self.p != 0:
                return False
        flat = []
        for col in (a, b, c):
            flat += col + [0] * (self.length - len(col))
        return all(flat[src] == flat[dst] for src, dst in enumerate(self.permutation))

    def to_bytes(self):
        raise NotImplementedError

    @classmethod
    def from_bytes(cls, data):
        raise NotImplementedError
