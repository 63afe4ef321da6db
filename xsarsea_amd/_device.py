"""Device-resident rasters at the drop-in boundary.

`invert_from_model`, `sigma0_detrend` and `nesz_flattening` accept rasters that already live in HBM -- torch CUDA(=HIP)
tensors, or any object exposing `__cuda_array_interface__` -- and then return torch tensors on the same device: nothing
crosses PCIe, the kernels run on torch's current stream (asynchronously, ordered with the caller's other work on it).
PyTorch is plumbing here: it owns the device memory and the stream, the work is libxsw's.
"""
import numpy as np

from . import _lib


def _torch():
    import sys
    return sys.modules.get("torch")  # never imported on behalf of a numpy caller


def is_device_array(a):
    if a is None or np.isscalar(a) or isinstance(a, np.ndarray):
        return False
    torch = _torch()
    if torch is not None and isinstance(a, torch.Tensor):
        return a.is_cuda
    return hasattr(a, "__cuda_array_interface__")


def any_device_array(*arrays):
    return any(is_device_array(a) for a in arrays)


def as_tensor(a, device, dtype=None):
    """torch view of a device array (zero copy), or an upload of a host array / scalar raster, on `device`."""
    import torch
    if isinstance(a, torch.Tensor):
        t = a if a.device == device else a.to(device)
    elif hasattr(a, "__cuda_array_interface__"):
        t = torch.as_tensor(a, device=device)
    else:
        t = torch.as_tensor(np.asarray(a)).to(device)
    return t if dtype is None or t.dtype == dtype else t.to(dtype)


def device_of(*arrays):
    import torch
    for a in arrays:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    for a in arrays:
        if is_device_array(a):
            return torch.as_tensor(a, device="cuda").device
    raise ValueError("no device array among the arguments")


class on_current_stream:
    """Runs the context's launches on torch's current stream of `device` (so they are ordered with the caller's other work,
    asynchronously), then hands the context back to its own stream; both hand-overs are device-side event waits."""

    def __init__(self, ctx, device):
        import torch
        self.ctx, self.handle = ctx, torch.cuda.current_stream(device).cuda_stream

    def __enter__(self):
        self.ctx.lock.acquire()
        self.ctx.set_stream(self.handle)
        return self.ctx

    def __exit__(self, *exc):
        try:
            self.ctx.use_own_stream()
        finally:
            self.ctx.lock.release()
        return False


def xsw_dtype(t):
    import torch
    if t.dtype in (torch.float32, torch.complex64):
        return _lib.XSW_F32
    if t.dtype in (torch.float64, torch.complex128):
        return _lib.XSW_F64
    raise TypeError(f"raster dtype must be float32/float64 (complex64/complex128), not {t.dtype}")
