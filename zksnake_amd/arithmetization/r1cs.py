"""
R1CS object with the shape the prover consumes (reference python/zksnake/arithmetization/r1cs.py:11-122):
attributes A, B, C (SparseArray), n_public, p; `is_sat`; `from_file` for circom `.r1cs` binaries.

The reference builds A, B, C by compiling its symbolic ConstraintSystem (Rust, O(n^2), out of scope here);
this class is filled directly from matrices (`from_matrices`), from triplet lists, or from a circom file
whose matrices are kept verbatim in circom wire order (proofs do not depend on the wire order).  File circuits get a
`constraint_system` facade with the reference's `unsafe_assign` / `solve` (hint callbacks for wires that propagation alone
cannot determine: examples/example_bitify_circom.py).
"""

from ..array import SparseArray
from ..ecc import EllipticCurve


class Var:
    """A named circuit variable: the one role of the reference's `Var` (src/arithmetization/symbolic.rs `Field` with
    `Gate::Input(name)`) that file circuits need -- naming the target of `unsafe_assign`.  Building expressions with it is
    the reference's symbolic front end, which is outside the accelerated path."""

    __slots__ = ("name",)

    def __init__(self, name: str):
        if not isinstance(name, str):
            raise TypeError("Invalid assignment expression")
        self.name = name

    def __repr__(self):
        return f"Var({self.name!r})"


class FileConstraintSystem:
    """`r1cs.constraint_system` of a circuit loaded with `R1CS.from_file`: the two calls the reference's examples make on it
    (examples/example_bitify_circom.py:11-24) -- `unsafe_assign(Var(name), func, args)` and `solve(inputs)` -- backed by the
    matrices of the file instead of the reference's symbolic ConstraintSystem."""

    def __init__(self, r1cs):
        self._r1cs = r1cs
        self.hints = []   # (target wire, callable, argument names) in registration order

    def unsafe_assign(self, target, func, args):
        """Non-deterministic assignment (symbolic.rs:634-650): once every variable named in `args` is known, `target`
        takes `func(**{name: value})`, which must be an int.  `target` must be a plain variable."""
        if not isinstance(target, Var):
            raise TypeError("Invalid assignment expression")
        r = self._r1cs
        try:
            wire = r.wire_names.index(target.name, 1)
        except ValueError:
            raise KeyError(f"unknown variable {target.name}") from None
        # the argument names are looked up when the hint is evaluated, as in the reference (symbolic.rs:760 "Argument not exist")
        self.hints.append((wire, func, tuple(args)))

    def solve(self, inputs: dict) -> dict:
        return self._r1cs.solve(inputs)


class R1CS:
    def __init__(self, cs=None, curve: str = "BN254"):
        self.A = None
        self.B = None
        self.C = None
        self.constraint_system = cs
        self.curve = curve
        self.n_public = (len(cs.public_vars) + 1) if cs is not None else 1
        self.p = EllipticCurve(curve).order
        self.wire_names = None   # from_file: wire index -> variable name (.sym labels, or out1/pub1/priv1/v1 ...)
        self._wire_of = {}       # from_file: variable name -> wire index
        self.input_wires = ()    # from_file: wires of the declared public and private inputs

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_triplets(cls, A, B, C, n_row, n_col, n_public, curve="BN254"):
        """A, B, C: (rows, cols, vals) triples of equal-length sequences."""
        self = cls(None, curve)
        self.A = SparseArray.from_triplets(*A, n_row, n_col, self.p)
        self.B = SparseArray.from_triplets(*B, n_row, n_col, self.p)
        self.C = SparseArray.from_triplets(*C, n_row, n_col, self.p)
        self.n_public = n_public
        return self

    @classmethod
    def from_matrices(cls, A, B, C, n_public, curve="BN254"):
        """dense row lists (tests / tiny circuits)"""
        self = cls(None, curve)
        n_row, n_col = len(A), len(A[0])
        self.A = SparseArray(A, n_row, n_col, self.p)
        self.B = SparseArray(B, n_row, n_col, self.p)
        self.C = SparseArray(C, n_row, n_col, self.p)
        for m in (self.A, self.B, self.C):
            m.triplets = [(r, c, v % self.p) for r, c, v in m.triplets]
        self.n_public = n_public
        return self

    def compile(self):
        """compile the attached constraint system (must expose the reference's compile_to_r1cs() /
        num_constraints() / num_witness() protocol, arithmetization/r1cs.py:21-40)."""
        if self.constraint_system is None or isinstance(self.constraint_system, FileConstraintSystem):
            if self.A is None:
                raise ValueError("no constraint system attached")
            return   # matrices read from a file are used as they are
        cs = self.constraint_system
        rows, cols = cs.num_constraints(), cs.num_witness() + 1
        self.A = SparseArray([[]], rows, cols, self.p)
        self.B = SparseArray([[]], rows, cols, self.p)
        self.C = SparseArray([[]], rows, cols, self.p)
        for a, b, c in cs.compile_to_r1cs():
            self.A.append(a)
            self.B.append(b)
            self.C.append(c)

    def solve(self, inputs: dict) -> dict:
        """{variable name: value} for every wire, from the declared inputs (reference r1cs.py:42-55).  Circuits loaded
        with `from_file` are solved by constraint propagation over the matrices (`solve_wires`); as in the reference
        only the declared inputs are read (symbolic.rs:654-663) and a missing one is an error."""
        if self.constraint_system is not None and not isinstance(self.constraint_system, FileConstraintSystem):
            return self.constraint_system.solve(inputs)
        if self.wire_names is None:
            raise NotImplementedError("no symbolic constraint system attached; supply the witness directly")
        known = {}
        for wire in self.input_wires:
            name = self.wire_names[wire]
            if name not in inputs:
                raise ValueError(f"missing value for input variable {name}")
            known[wire] = inputs[name]
        hints = self.constraint_system.hints if isinstance(self.constraint_system, FileConstraintSystem) else ()
        w = self.solve_wires(known, hints)
        return {self.wire_names[i]: w[i] for i in range(1, len(w))}

    def generate_witness(self, solve_result: dict):
        if self.constraint_system is None or isinstance(self.constraint_system, FileConstraintSystem):
            if self.wire_names is None:
                raise NotImplementedError("no symbolic constraint system attached; supply the witness directly")
            # circom wire order: [1, public outputs, public inputs, private inputs, intermediates]
            w = [1] + [solve_result[self.wire_names[i]] % self.p for i in range(1, self.A.n_col)]
            return w[: self.n_public], w[self.n_public:]
        w = []
        for v in self.constraint_system.get_witness_vector():
            if v == "0":
                w.append(1)
            elif isinstance(v, str):
                w.append(solve_result[v] % self.p)
            else:
                w.append(v % self.p)
        return w[: self.n_public], w[self.n_public:]

    def solve_wires(self, known: dict, hints=()) -> list:
        """Witness by constraint propagation for matrix-only systems (circom files): starting from the wires in
        `known` ({wire index: value}; wire 0 = 1 is implied), repeatedly take a constraint <A,w>*<B,w> = <C,w> in
        which exactly one wire is unknown and appears linearly, and solve for it.  This is the job the reference's
        symbolic solver does for file circuits (src/arithmetization/symbolic.rs:652-806, one-unknown isolation).
        `hints`: (target wire, callable, argument names) rows of `unsafe_assign`; a hint fires as soon as all its
        arguments are known (symbolic.rs:748-782) and its result must be an int ("Non deterministic result must be Integer")."""
        assert self.A is not None, "R1CS is not compiled"
        p = self.p
        n_col = self.A.n_col
        w = [None] * n_col
        w[0] = 1
        for k, v in known.items():
            w[int(k)] = int(v) % p
        rows = {}
        for name, m in (("a", self.A), ("b", self.B), ("c", self.C)):
            for r, c, v in m.triplets:
                rows.setdefault(r, {"a": [], "b": [], "c": []})[name].append((c, v % p))

        def split(terms):
            acc, unk = 0, []
            for c, v in terms:
                if w[c] is None:
                    unk.append((c, v))
                else:
                    acc = (acc + v * w[c]) % p
            return acc, unk

        def fire_hints(todo):
            fired, rest = False, []
            for wire, func, args in todo:
                for a in args:
                    if a not in self._wire_of:
                        raise KeyError(f"Argument not exist: {a}")
                vals = [w[self._wire_of[a]] for a in args]
                if any(v is None for v in vals):
                    rest.append((wire, func, args))
                    continue
                out = func(**dict(zip(args, vals)))
                if isinstance(out, bool) or not isinstance(out, int):
                    raise TypeError("Non deterministic result must be Integer")
                if out < 0:
                    raise OverflowError("can't convert negative int to unsigned")   # the reference parses the result as BigUint
                # A hint may confirm a wire that an input or the propagation has already fixed, never change it: the reference
                # overwrites blindly (symbolic.rs:771-773), which lets a hint silently replace a declared input (round-3 advisor finding)
                if w[wire] is not None and w[wire] != out % p:
                    raise ValueError(f"hint for wire {wire} ({self.wire_names[wire]}) contradicts the value it already has")
                w[wire] = out % p
                fired = True
            return fired, rest

        hints_left = list(hints)
        pending = [rows[i] for i in sorted(rows)]
        progress = True
        while (pending or hints_left) and progress:
            progress, hints_left = fire_hints(hints_left)
            rest = []
            for row in pending:
                (sa, ua), (sb, ub), (sc, uc) = split(row["a"]), split(row["b"]), split(row["c"])
                n_unknown = len({c for c, _ in ua + ub + uc})
                if n_unknown == 0:
                    continue
                solved = False
                if n_unknown == 1:
                    if uc and not ua and not ub and len(uc) == 1:
                        c, v = uc[0]
                        w[c] = (sa * sb - sc) * pow(v, -1, p) % p
                        solved = True
                    elif ua and not ub and not uc and len(ua) == 1 and sb % p:
                        c, v = ua[0]
                        w[c] = (sc * pow(sb, -1, p) - sa) * pow(v, -1, p) % p
                        solved = True
                    elif ub and not ua and not uc and len(ub) == 1 and sa % p:
                        c, v = ub[0]
                        w[c] = (sc * pow(sa, -1, p) - sb) * pow(v, -1, p) % p
                        solved = True
                if solved:
                    progress = True
                else:
                    rest.append(row)
            pending = rest
        if any(x is None for x in w):
            raise ValueError("constraint propagation could not determine every wire from the given inputs")
        if hints:
            # a hint is taken on trust when it fires; the reference asserts every constraint once all its variables are
            # known (symbolic.rs:684-696), so a wrong hint must not pass silently here either
            for i, row in sorted(rows.items()):
                (sa, _), (sb, _), (sc, _) = split(row["a"]), split(row["b"]), split(row["c"])
                if sa * sb % p != sc:
                    raise AssertionError(f"constraint {i} is not satisfied by the hinted assignment")
        return w

    def is_sat(self, public_witness: list, private_witness: list):
        assert self.A is not None, "R1CS is not compiled"
        w = list(public_witness) + list(private_witness)
        az, bz, cz = self.A.dot(w), self.B.dot(w), self.C.dot(w)
        return [x * y % self.p for x, y in zip(az, bz)] == cz

    def to_bytes(self):
        raise NotImplementedError

    @classmethod
    def from_bytes(cls, data):
        raise NotImplementedError

    # ---- circom .r1cs (binary format v1) --------------------------------------------------------
    @classmethod
    def from_file(cls, r1csfile: str, symfile: str = None, curve: str = "BN254"):
        """Load A, B, C from a circom `.r1cs` file.  Sections: 1 header, 2 constraints, 3 wire map.
        Wires: [1, public outputs, public inputs, private inputs, intermediates]; n_public counts the
        constant wire plus the public outputs and inputs (reference r1cs.py:94-122, parser.py:10-219).
        Variable names come from the `.sym` file (rows `label,index,component,name`; only index > 0 is a wire,
        parser.py:29-35,170-177) or, without one, are out<i>, pub<i>, priv<i>, v<i> (parser.py:178-201); they drive
        `solve({"main.a": 1, ...})` and `generate_witness`.  The matrices are kept verbatim in circom wire order
        (the reference re-compiles them through its symbolic system into a HashMap order; proofs do not depend on it)."""
        with open(r1csfile, "rb") as f:
            data = f.read()
        if data[:4] != b"r1cs":
            raise AssertionError(f"Invalid magic bytes: {data[:4]}")
        version = int.from_bytes(data[4:8], "little")
        if version != 1:
            raise AssertionError(f"Unsupported r1cs file version: {version}")
        n_sections = int.from_bytes(data[8:12], "little")
        pos = 12
        sections = {}
        for _ in range(n_sections):
            kind = int.from_bytes(data[pos:pos + 4], "little")
            size = int.from_bytes(data[pos + 4:pos + 12], "little")
            sections[kind] = data[pos + 12:pos + 12 + size]
            pos += 12 + size
        hdr = sections[1]
        fs = int.from_bytes(hdr[0:4], "little")
        prime = int.from_bytes(hdr[4:4 + fs], "little")
        o = 4 + fs
        n_wires, n_pub_out, n_pub_in, n_priv_in = (int.from_bytes(hdr[o + 4 * k:o + 4 * k + 4], "little") for k in range(4))
        m_constraints = int.from_bytes(hdr[o + 24:o + 28], "little")
        self = cls(None, curve)
        if prime != self.p:
            raise ValueError("the .r1cs file is over a different field than the selected curve")
        body = sections[2]
        pos = 0
        mats = ([], [], [])
        for row in range(m_constraints):
            for k in range(3):
                nterms = int.from_bytes(body[pos:pos + 4], "little")
                pos += 4
                for _ in range(nterms):
                    wire = int.from_bytes(body[pos:pos + 4], "little")
                    val = int.from_bytes(body[pos + 4:pos + 4 + fs], "little")
                    pos += 4 + fs
                    if wire >= n_wires:
                        raise IndexError(f"constraint {row} references wire {wire}, but the file declares {n_wires} wires")
                    if val:
                        mats[k].append((row, wire, val))
        for name, trip in zip("ABC", mats):
            rows, cols, vals = (list(t) for t in zip(*trip)) if trip else ([], [], [])
            setattr(self, name, SparseArray.from_triplets(rows, cols, vals, m_constraints, n_wires, self.p))
        self.n_public = 1 + n_pub_out + n_pub_in
        first_in = 1 + n_pub_out
        self.input_wires = tuple(range(first_in, first_in + n_pub_in + n_priv_in))
        names = [None] * n_wires
        if symfile:
            import csv
            with open(symfile, "r", encoding="utf-8") as f:
                for rec in csv.reader(f, delimiter=","):
                    if len(rec) != 4:
                        continue
                    index = int(rec[1])
                    if 0 < index < n_wires:   # -1: optimised away; 0 is the constant wire
                        names[index] = rec[3]
        else:
            layout = (("out", n_pub_out), ("pub", n_pub_in), ("priv", n_priv_in),
                      ("v", n_wires - 1 - n_pub_out - n_pub_in - n_priv_in))
            wire = 1
            for prefix, count in layout:
                for k in range(count):
                    names[wire] = f"{prefix}{k + 1}"
                    wire += 1
        missing = [i for i in range(1, n_wires) if names[i] is None]
        if missing:
            raise ValueError(f"the symbol file names no variable for wires {missing[:8]}{'...' if len(missing) > 8 else ''}")
        names[0] = "0"
        self.wire_names = names
        self._wire_of = {name: i for i, name in enumerate(names) if i > 0}
        self.constraint_system = FileConstraintSystem(self)
        self.header = dict(n_wires=n_wires, n_pub_out=n_pub_out, n_pub_in=n_pub_in, n_priv_in=n_priv_in,
                           m_constraints=m_constraints, prime=prime)
        return self
