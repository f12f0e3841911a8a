"""Tensor plumbing between the caller's arrays and the C ABI's (pointer, brn_mem) pairs.

numpy arrays and CPU torch tensors travel as BRN_MEM_HOST; torch tensors on an AMD GPU travel as BRN_MEM_DEVICE on
torch's current stream.  PyTorch is plumbing here (device memory + streams), never arithmetic.
"""
import ctypes as C
import numpy as np

from . import _ffi


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def as_arg(x, shape=None):
    """-> (pointer:int, loc, keepalive, kind).  Enforces fp32 + contiguity (candle: Tensor::from_vec of f32)."""
    if x is None:
        return None, None, None, None
    if _is_torch(x):
        import torch
        if x.dtype != torch.float32:
            raise TypeError("expected an f32 tensor")
        t = x.contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        loc = _ffi.BRN_MEM_DEVICE if t.is_cuda else _ffi.BRN_MEM_HOST
        return t.data_ptr(), loc, t, "torch"
    a = np.ascontiguousarray(x, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a.ctypes.data, _ffi.BRN_MEM_HOST, a, "numpy"


def host_ptr(x):
    """weights are always host buffers at the boundary (the library copies them)."""
    if x is None:
        return None, None
    if _is_torch(x):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a.ctypes.data, a


def alloc_like(x, shape):
    if _is_torch(x):
        import torch
        return torch.empty(tuple(shape), dtype=torch.float32, device=x.device)
    return np.empty(tuple(shape), dtype=np.float32)


def stream_of(x):
    if _is_torch(x) and x.is_cuda:
        import torch
        return torch.cuda.current_stream(x.device).cuda_stream
    return None


def device_of(x, default=0):
    if _is_torch(x) and x.is_cuda:
        return x.device.index if x.device.index is not None else 0
    return default


def ptr_of(x):
    if _is_torch(x):
        return x.data_ptr()
    return x.ctypes.data
