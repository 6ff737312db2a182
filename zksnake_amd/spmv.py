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


def long_row_items(row_ptr, long_row=LONG_ROW, item=ITEM):
    """(long_rows, item_ptr, items): the rows with more than `long_row` entries, and their entry ranges cut into work
    items [k0, k1) of at most `item` entries; items[item_ptr[j]:item_ptr[j + 1]] belong to long_rows[j].  Pure numpy."""
    rp = np.asarray(row_ptr).astype(np.int64)
    lengths = np.diff(rp)
    long_rows = np.nonzero(lengths > long_row)[0]
    per_row = (lengths[long_rows] + item - 1) // item
    item_ptr = np.zeros(long_rows.shape[0] + 1, dtype=np.uint32)
    np.cumsum(per_row, out=item_ptr[1:])
    n_items = int(item_ptr[-1])
    row_of = np.repeat(np.arange(long_rows.shape[0]), per_row)
    k_in_row = np.arange(n_items) - item_ptr[:-1].astype(np.int64)[row_of]
    start = rp[long_rows][row_of] + k_in_row * item
    end = np.minimum(start + item, rp[long_rows + 1][row_of])
    items = np.stack([start, end], axis=1).astype(np.uint32).reshape(-1, 2)
    return long_rows.astype(np.uint32), item_ptr, np.ascontiguousarray(items)


class DeviceCsr:
    def __init__(self, curve_id, row_ptr, cols, vals):
        self.cid = curve_id
        self.n_rows = int(row_ptr.shape[0] - 1)
        self.row_ptr = DeviceBuffer.from_numpy(np.ascontiguousarray(row_ptr, dtype=np.uint32))
        self.cols = DeviceBuffer.from_numpy(np.ascontiguousarray(cols, dtype=np.uint32)) if len(cols) else None
        self.vals = DeviceBuffer.from_numpy(np.ascontiguousarray(vals, dtype=np.uint64)) if len(cols) else None
        long_rows, item_ptr, items = long_row_items(row_ptr)
        self.n_long = int(long_rows.shape[0])
        self.n_items = int(items.shape[0])
        if self.n_long:
            self.long_rows = DeviceBuffer.from_numpy(long_rows)
            self.item_ptr = DeviceBuffer.from_numpy(item_ptr)
            self.items = DeviceBuffer.from_numpy(items)
            self.partials = DeviceBuffer(self.n_items * 32)

    def apply(self, d_w, d_out, stream=None):
        """d_out[row] = sum_k vals[k] * d_w[cols[k]] (device pointers); rows without entries give 0"""
        lib = N.load()
        if self.cols is None:
            N.check(lib.zk_dev_memset_async(d_out, 0, self.n_rows * 32, stream))   # ordered with the kernels that read d_out on `stream`
            return
        N.check(lib.zk_spmv_dev(self.cid, self.n_rows, self.row_ptr.ptr, self.cols.ptr, self.vals.ptr, d_w, d_out,
                                LONG_ROW if self.n_long else 0, stream))
        if self.n_long:
            N.check(lib.zk_spmv_long_dev(self.cid, self.n_long, self.long_rows.ptr, self.item_ptr.ptr, self.n_items, self.items.ptr,
                                         self.cols.ptr, self.vals.ptr, d_w, self.partials.ptr, d_out, stream))
