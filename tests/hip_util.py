"""A device buffer for tests, on the HIP runtime libstn.so itself runs on (ctypes on libamdhip64 — the copy the library was linked
against is already in the process once binding.load() has run; no PyTorch needed for a destination pointer)."""
import ctypes

import numpy as np

from supertonic_amd import binding


def _hip():
    binding.load()
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise OSError("libamdhip64 not found")


class DeviceBuffer:
    """hipMalloc'd copy of a numpy array; .ptr for the C ABI, .to_host() reads it back (hipMemcpy synchronises)."""

    def __init__(self, arr):
        self._hip = _hip()
        self._hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self._hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self._hip.hipFree.argtypes = [ctypes.c_void_p]
        arr = np.ascontiguousarray(arr)
        self.shape, self.dtype, self.nbytes = arr.shape, arr.dtype, arr.nbytes
        p = ctypes.c_void_p()
        assert self._hip.hipMalloc(ctypes.byref(p), self.nbytes) == 0
        self.ptr = p.value
        assert self._hip.hipMemcpy(self.ptr, arr.ctypes.data, self.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        assert self._hip.hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            self._hip.hipFree(self.ptr)
            self.ptr = None
