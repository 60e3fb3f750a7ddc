"""Helpers for the GPU parity tests: layout / dtype conversion between the oracle's
NCHW float32 numpy arrays and the library's NHWC device tensors."""
import numpy as np
import torch

from gdl import _lib as L

DEV = "cuda:0"


def bf16_round(a):
    """float32 numpy -> nearest bf16, returned as float32 (what a bf16 tensor holds)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def quant(a, dt):
    return bf16_round(a) if dt == L.GDL_BF16 else np.ascontiguousarray(a, dtype=np.float32)


def to_nhwc(x_nchw, dt):
    t = torch.from_numpy(np.ascontiguousarray(x_nchw, dtype=np.float32)).to(DEV)
    return t.permute(0, 2, 3, 1).contiguous().to(L.torch_dtype(dt))


def from_nhwc(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu().numpy()


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype).contiguous()


def empty(shape, dt):
    return torch.empty(shape, device=DEV, dtype=L.torch_dtype(dt))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def tol(dt, f32, bf16):
    return bf16 if dt == L.GDL_BF16 else f32


def pack_weight(w, dt):
    """float32 numpy [K,C,R,S] -> (w_krsc, w_crsk) device tensors via the library."""
    K, C, R, S = w.shape
    wd = dev(w)
    krsc = empty((K, R, S, C), dt)
    crsk = empty((C, R, S, K), dt)
    L.call("gdl_pack_weight", dt, L.ptr(wd), L.ptr(krsc), L.ptr(crsk), K, C, R, S, L.cur_stream())
    return krsc, crsk


def gather_table(mode, dt, N, H, W, C, K, R, S, stride, pad):
    """Build the gather table of one convolution geometry (returns the owning uint8 tensor)."""
    nbytes = L.load().gdl_conv_table_bytes(mode, N, H, W, R, S, stride, pad)
    t = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, S, stride, pad, L.ptr(t), L.cur_stream())
    return t
