"""Thin RAII wrapper over the library's device allocator (zk_dev_* in include/zkmi.h)."""

import ctypes

import numpy as np

from . import _native as N


class DeviceBuffer:
    def __init__(self, nbytes):
        lib = N.ensure_gpu()
        ptr = ctypes.c_void_p()
        N.check(lib.zk_dev_alloc(nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value
        self.nbytes = nbytes

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        buf.upload(arr)
        return buf

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        N.check(N.load().zk_dev_upload(self.ptr + offset, arr.ctypes.data, arr.nbytes))

    def download(self, shape, dtype=np.uint64, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        N.check(N.load().zk_dev_download(out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        N.check(N.load().zk_dev_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            N.load().zk_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class _PinnedBlock:
    """one zk_host_alloc allocation; freed when the last holder (the PinnedArray, or a numpy view's base buffer) lets go"""

    def __init__(self, nbytes):
        ptr = ctypes.c_void_p()
        N.check(N.ensure_gpu().zk_host_alloc(nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value

    def release(self):
        if self.ptr:
            N.load().zk_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class PinnedArray:
    """page-locked host memory viewed as a numpy array (`.array`): staging for vectors that cross the link every proof.
    Views of `.array` keep the allocation alive; `free()` releases it (do not use views afterwards).  The ctypes buffer
    under the array refers to the allocation block only, never back to this object: no reference cycle, so the page-locked
    memory goes away by reference counting as soon as the array and its views do (round-2 advisor finding)."""

    def __init__(self, shape, dtype=np.uint64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._block = _PinnedBlock(self.nbytes)
        self.ptr = self._block.ptr
        raw = (ctypes.c_uint8 * max(1, self.nbytes)).from_address(self.ptr)
        raw._owner = self._block
        self.array = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self._block.release()
            self.ptr = None
