"""
Sparse matrix in triplet form, the input contract of the proving path
(same attributes as the reference's python/zksnake/array.py:4-44).

`dot` keeps the reference's host-side semantics for small systems; `to_csr` produces the arrays
the GPU SpMV kernel (zk_spmv_dev) consumes so that A.w, B.w, C.w of a 2^20-row system never
loop in Python.
"""

import numpy as np

from . import _native as N


def _stable_order(keys):
    """argsort(keys, kind="stable") for non-negative integer keys below 2^32 and fewer than 2^32 entries: the
    position rides in the low half of a 64-bit sort key, so numpy's (vectorised, unstable) integer sort can be used --
    several times faster than its stable merge sort on 2^20 and more entries"""
    n = keys.shape[0]
    if n == 0 or n >= (1 << 32) or int(keys.max()) >= (1 << 32):
        return np.argsort(keys, kind="stable")
    packed = (keys.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    packed.sort()
    return (packed & np.uint64(0xFFFFFFFF)).astype(np.int64)


class SparseArray:
    def __init__(self, matrix, n_row, n_col, p):
        self.p = p
        self.n_row = n_row
        self.n_col = n_col
        self._triplets = []
        self._columns = None  # (rows, cols, vals) as given to from_triplets, until someone asks for `triplets`
        self.triplets_map = {}
        self._csr = None
        self._csc = None
        for i, row in enumerate(matrix):
            for j, value in enumerate(row):
                if value != 0:
                    self._triplets.append((i, j, value))

    @classmethod
    def from_triplets(cls, rows, cols, vals, n_row, n_col, p):
        """bulk constructor: the three columns are kept as given (lists or arrays); the reference's list of
        (row, col, value) tuples is only built when `triplets` is read"""
        self = cls([], n_row, n_col, p)
        self._columns = (rows, cols, vals)
        return self

    @property
    def triplets(self):
        if self._columns is not None:
            rows, cols, vals = self._columns
            self._triplets = list(zip((int(r) for r in rows), (int(c) for c in cols), (int(v) for v in vals)))
            self._columns = None
        return self._triplets

    @triplets.setter
    def triplets(self, value):
        self._columns = None
        self._triplets = value

    def append(self, triplets):
        current = self.triplets  # materialises the tuples when the matrix came from from_triplets
        for row, col, value in triplets:
            if value != 0:
                self.triplets_map.setdefault(row, []).append((col, value))
                current.append((row, col, value))
        self._csr = None
        self._csc = None

    def rows_map(self):
        """row -> [(col, value)] (built on demand when the matrix came from from_triplets)"""
        if len(self.triplets_map) == 0 and self.triplets:
            m = {}
            for row, col, value in self.triplets:
                m.setdefault(row, []).append((col, value))
            self.triplets_map = m
        return self.triplets_map

    def dot(self, vector):
        out = [0] * self.n_row
        for row, col, value in self.triplets:
            out[row] += vector[col] * value
        return [x % self.p for x in out]

    def to_csr(self):
        """(row_ptr uint32[n_row+1], cols uint32[nnz], vals uint64[nnz,4]) with rows padded to n_row"""
        if self._csr is None:
            if self._columns is not None:
                rows = np.asarray(self._columns[0], dtype=np.int64)
                cols = np.asarray(self._columns[1], dtype=np.int64)
                vals = self._value_limbs(self._columns[2])
            else:
                nnz = len(self._triplets)
                rows = np.fromiter((t[0] for t in self._triplets), dtype=np.int64, count=nnz)
                cols = np.fromiter((t[1] for t in self._triplets), dtype=np.int64, count=nnz)
                vals = self._value_limbs([t[2] for t in self._triplets])
            # the GPU SpMV reads w[cols[k]] unchecked: reject out-of-range indices here, as the reference's dot() does
            # with an IndexError (array.py:36-44)
            if rows.size and (int(rows.min()) < 0 or int(rows.max()) >= self.n_row):
                raise IndexError(f"row index out of range for a matrix with {self.n_row} rows")
            if cols.size and (int(cols.min()) < 0 or int(cols.max()) >= self.n_col):
                raise IndexError(f"column index out of range for a matrix with {self.n_col} columns")
            counts = np.bincount(rows, minlength=self.n_row)
            row_ptr = np.zeros(self.n_row + 1, dtype=np.uint32)
            np.cumsum(counts, out=row_ptr[1:])
            if rows.size < 2 or bool((rows[1:] >= rows[:-1]).all()):
                # triplets already in row order (how circuits are usually emitted): nothing to permute
                self._csr = (row_ptr, cols.astype(np.uint32), np.ascontiguousarray(vals))
            else:
                order = _stable_order(rows)
                self._csr = (row_ptr, cols[order].astype(np.uint32), np.ascontiguousarray(vals[order]))
        return self._csr

    def _value_limbs(self, vals):
        """matrix values -> (nnz, 4) uint64 limbs reduced mod p; small non-negative values (the common case: +-1
        coefficients arrive as 1 and p - 1) take a vectorised path"""
        if isinstance(vals, np.ndarray) and vals.ndim == 2:
            return np.ascontiguousarray(vals, dtype=np.uint64)
        if isinstance(vals, np.ndarray):
            # numpy casts silently: a signed -1 would wrap to 2^64 - 1 instead of p - 1, a float would truncate.
            # Only unsigned integers, or signed ones that are all non-negative, take the vectorised path.
            fast = vals.dtype.kind == "u" or (vals.dtype.kind == "i" and (vals.size == 0 or int(vals.min()) >= 0))
            if vals.dtype.kind not in "uiO":
                raise TypeError(f"matrix values must be integers, got dtype {vals.dtype}")
            if fast:
                out = np.zeros((vals.shape[0], 4), dtype=np.uint64)
                out[:, 0] = vals.astype(np.uint64)
                return out
            return N.ints_to_limbs([int(v) % self.p for v in vals], 4)
        try:
            small = np.asarray(vals, dtype=np.uint64)  # Python ints: raises OverflowError for values >= 2^64 or negative
            out = np.zeros((small.shape[0], 4), dtype=np.uint64)
            out[:, 0] = small
            return out
        except (OverflowError, TypeError, ValueError):
            p = self.p
            return N.ints_to_limbs([int(v) % p for v in vals], 4)

    def to_csc(self):
        """the transpose in CSR form: (col_ptr uint32[n_col+1], rows uint32[nnz], vals uint64[nnz,4]) -- what
        zk_spmv_dev needs for A^T x (the per-wire sums of Groth16.setup)"""
        if self._csc is None or self._csr is None:
            row_ptr, cols, vals = self.to_csr()
            rows = np.repeat(np.arange(self.n_row, dtype=np.uint32), np.diff(row_ptr.astype(np.int64)))
            order = _stable_order(cols.astype(np.int64))
            col_ptr = np.zeros(self.n_col + 1, dtype=np.uint32)
            np.cumsum(np.bincount(cols, minlength=self.n_col), out=col_ptr[1:])
            self._csc = (col_ptr, np.ascontiguousarray(rows[order]), np.ascontiguousarray(vals[order]))
        return self._csc
