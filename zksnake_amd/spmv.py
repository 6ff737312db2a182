"""
Device-resident CSR matrix for the witness products of the proving path (A.w, B.w, C.w of QAP.evaluate_witness,
python/zksnake/groth16/qap.py:42-49 -> array.py:36-44) and for the per-wire sums of Groth16.setup (the transposes).

Rows are split by length once per matrix: short rows take one GPU lane each (zk_spmv_dev); a long row -- the constant-one
wire and the input wires of a real circuit reach 2^20 entries in the transposed matrices -- is cut into work items of at
most ITEM entries that one workgroup each sums up (zk_spmv_long_dev).
"""

import numpy as np

from . import _native as N
from .device import DeviceBuffer

LONG_ROW = 64     # rows with more entries leave the lane-per-row kernel
ITEM = 4096       # entries per work item of a long row


class DeviceCsr:
    def __init__(self, curve_id, row_ptr, cols, vals):
        self.cid = curve_id
        self.n_rows = int(row_ptr.shape[0] - 1)
        self.row_ptr = DeviceBuffer.from_numpy(np.ascontiguousarray(row_ptr, dtype=np.uint32))
        self.cols = DeviceBuffer.from_numpy(np.ascontiguousarray(cols, dtype=np.uint32)) if len(cols) else None
        self.vals = DeviceBuffer.from_numpy(np.ascontiguousarray(vals, dtype=np.uint64)) if len(cols) else None
        lengths = np.diff(row_ptr.astype(np.int64))
        long_rows = np.nonzero(lengths > LONG_ROW)[0]
        self.n_long = int(long_rows.shape[0])
        self.n_items = 0
        if self.n_long:
            per_row = (lengths[long_rows] + ITEM - 1) // ITEM
            item_ptr = np.zeros(self.n_long + 1, dtype=np.uint32)
            np.cumsum(per_row, out=item_ptr[1:])
            self.n_items = int(item_ptr[-1])
            row_of = np.repeat(np.arange(self.n_long), per_row)
            k_in_row = np.arange(self.n_items) - item_ptr[:-1].astype(np.int64)[row_of]
            start = row_ptr[long_rows].astype(np.int64)[row_of] + k_in_row * ITEM
            end = np.minimum(start + ITEM, row_ptr[long_rows + 1].astype(np.int64)[row_of])
            items = np.stack([start, end], axis=1).astype(np.uint32)
            self.long_rows = DeviceBuffer.from_numpy(long_rows.astype(np.uint32))
            self.item_ptr = DeviceBuffer.from_numpy(item_ptr)
            self.items = DeviceBuffer.from_numpy(np.ascontiguousarray(items))
            self.partials = DeviceBuffer(self.n_items * 32)

    def apply(self, d_w, d_out, stream=None):
        """d_out[row] = sum_k vals[k] * d_w[cols[k]] (device pointers); rows without entries give 0"""
        lib = N.load()
        if self.cols is None:
            N.check(lib.zk_dev_memset(d_out, 0, self.n_rows * 32))
            return
        N.check(lib.zk_spmv_dev(self.cid, self.n_rows, self.row_ptr.ptr, self.cols.ptr, self.vals.ptr, d_w, d_out,
                                LONG_ROW if self.n_long else 0, stream))
        if self.n_long:
            N.check(lib.zk_spmv_long_dev(self.cid, self.n_long, self.long_rows.ptr, self.item_ptr.ptr, self.n_items, self.items.ptr,
                                         self.cols.ptr, self.vals.ptr, d_w, self.partials.ptr, d_out, stream))
