"""Caller-owned device memory for the GPU tests: a few lines of ctypes over the HIP runtime libmgx itself is
linked against (/opt/rocm/lib/libamdhip64.so - the test process then runs ONE ROCm stack, the one the library
was built with; the torch wheel bundles another copy of the runtime with the same SONAME, which is why the GPU
tests no longer hold their device buffers in torch tensors).  DevArray mimics the little of the tensor
interface the tests use: row slices are views, clone() / numpy() copy."""
import ctypes as C
import os

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        # the copy of the runtime this process already has (libmgx's, if the library was loaded first), else /opt/rocm's
        path = "/opt/rocm/lib/libamdhip64.so"
        try:
            for line in open("/proc/self/maps"):
                if "libamdhip64" in line and "/" in line:
                    path = line[line.index("/"):].strip()
                    break
        except OSError:
            pass
        _hip = C.CDLL(path if os.path.exists(path) else "libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
    return _hip


def device_count() -> int:
    try:
        n = C.c_int(0)
        return n.value if hip().hipGetDeviceCount(C.byref(n)) == 0 else 0
    except OSError:
        return 0


def synchronize():
    assert hip().hipDeviceSynchronize() == 0


class _Owner:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), max(int(nbytes), 1)) == 0, "hipMalloc failed"

    def __del__(self):
        try:
            if self.ptr:
                hip().hipFree(self.ptr)
        except Exception:
            pass


class DevArray:
    """a C-contiguous array in device memory (or a view of whole leading rows of one)"""

    def __init__(self, shape, dtype, owner=None, offset=0):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._owner = owner or _Owner(self.nbytes)
        self._offset = offset

    def data_ptr(self) -> int:
        return (self._owner.ptr.value or 0) + self._offset

    def __getitem__(self, key):
        assert isinstance(key, slice) and key.step in (None, 1), "row slices only"
        lo, hi, _ = key.indices(self.shape[0])
        row = int(np.prod(self.shape[1:])) * self.dtype.itemsize
        return DevArray((hi - lo,) + self.shape[1:], self.dtype, self._owner, self._offset + lo * row)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        synchronize()
        assert hip().hipMemcpy(out.ctypes.data, self.data_ptr(), self.nbytes, 2) == 0      # device to host
        return out

    def cpu(self):
        return self

    def clone(self):
        out = DevArray(self.shape, self.dtype)
        assert hip().hipMemcpy(out.data_ptr(), self.data_ptr(), self.nbytes, 3) == 0       # device to device
        return out

    def contiguous(self):
        return self         # whole leading rows of a C-contiguous array are contiguous

    def item(self):
        assert int(np.prod(self.shape)) == 1
        return self.numpy().reshape(-1)[0].item()


def from_numpy(a: np.ndarray) -> DevArray:
    a = np.ascontiguousarray(a)
    out = DevArray(a.shape, a.dtype)
    assert hip().hipMemcpy(out.data_ptr(), a.ctypes.data, a.nbytes, 1) == 0                # host to device
    return out


def zeros(shape, dtype) -> DevArray:
    shape = (shape,) if isinstance(shape, (int, np.integer)) else shape
    out = DevArray(shape, dtype)
    assert hip().hipMemset(out.data_ptr(), 0, out.nbytes) == 0
    return out


def zeros_like(t: DevArray) -> DevArray:
    return zeros(t.shape, t.dtype)


def ones_like(t: DevArray) -> DevArray:
    return from_numpy(np.ones(t.shape, dtype=t.dtype))


def empty_like(t: DevArray) -> DevArray:
    return DevArray(t.shape, t.dtype)
