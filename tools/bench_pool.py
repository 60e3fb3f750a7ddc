#!/usr/bin/env python3
"""Stem pooling kernels alone at the CREMA-D B=64 shapes: bn+relu+maxpool forward, maxpool backward."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

lib = L.load()
dt = L.dtype_code("bf16")
dev = "cuda:0"
st = L.cur_stream()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, H, W in (("visual", 192, 112, 112), ("audio", 64, 129, 94)):
    C = 64
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.randn(N, H, W, C, device=dev).bfloat16()
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    out = torch.empty(N, P, Q, C, device=dev, dtype=torch.bfloat16)
    idx = torch.empty(N, P, Q, C, device=dev, dtype=torch.uint8)
    dout = torch.randn(N, P, Q, C, device=dev).bfloat16()
    dx = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
    ym = torch.empty_like(out)
    mu, rs, ga, coef = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, \
        torch.randn(2 * C, device=dev) * 0.01
    f = timeit(lambda: L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(idx), None, N, H, W, C, st))
    f2 = timeit(lambda: L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(idx), L.ptr(ym), N, H, W, C, st))
    b = timeit(lambda: L.call("gdl_maxpool_bwd", dt, L.ptr(dout), L.ptr(idx), L.ptr(dx), N, H, W, C, st))
    b2 = timeit(lambda: L.call("gdl_maxpool_bn_bwd_apply", dt, L.ptr(dout), L.ptr(idx), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(mu),
                               L.ptr(rs), L.ptr(ga), L.ptr(coef), L.ptr(dx), N, H, W, C, st))
    fb = (y.numel() * 2 + out.numel() * 3) / 1e6
    bb = (dx.numel() * 2 + out.numel() * 3) / 1e6
    print(f"{name}: fwd {f:.1f} us ({fb:.0f} MB, {fb / f * 1e3:.0f} GB/s), +ymax {f2:.1f} us   bwd {b:.1f} us ({bb:.0f} MB, "
          f"{bb / b * 1e3:.0f} GB/s)   fused bwd+bn apply {b2:.1f} us ({bb + dx.numel() * 2 / 1e6:.0f} MB, "
          f"{(bb + dx.numel() * 2 / 1e6) / b2 * 1e3:.0f} GB/s)")
