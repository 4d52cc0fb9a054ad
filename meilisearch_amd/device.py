"""Context (one per process and GPU) and raw device buffers (torch = plumbing)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class Context:
    """msi_ctx: owns the HIP stream every libmsi call of this process uses."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        check(lib().msi_ctx_create(device, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    @property
    def device(self):
        return lib().msi_ctx_device(self._h)

    @property
    def stream_ptr(self):
        return lib().msi_ctx_stream(self._h)

    def torch_stream(self):
        """The context's HIP stream as a torch ExternalStream (for events)."""
        import torch
        return torch.cuda.ExternalStream(self.stream_ptr, device=torch.device("cuda", self.device))

    def set_profiling(self, enable):
        check(lib().msi_ctx_set_profiling(self._h, 1 if enable else 0))

    def synchronize(self):
        check(lib().msi_ctx_synchronize(self._h))

    def close(self):
        if self._h:
            lib().msi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """A torch CUDA tensor viewed as a raw device pointer."""

    def __init__(self, tensor):
        assert _lib.on_device(tensor) and tensor.is_contiguous()
        self.tensor = tensor

    @property
    def ptr(self):
        return C.c_void_p(self.tensor.data_ptr())


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def as_u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)
