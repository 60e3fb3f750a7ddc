#!/usr/bin/env python3
"""Standalone bandwidth of the BatchNorm / elementwise kernels at the CREMA-D B=64 shapes."""
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
    return e0.elapsed_time(e1) / iters


for name, M, C in (("visual l1", 64 * 3 * 56 * 56, 64), ("visual l2", 64 * 3 * 28 * 28, 128), ("visual l3", 64 * 3 * 14 * 14, 256),
                   ("visual l4", 64 * 3 * 7 * 7, 512), ("audio l1", 64 * 65 * 47, 64), ("audio l2", 64 * 33 * 24, 128)):
    y = torch.randn(M, C, device=dev).bfloat16()
    g = torch.randn(M, C, device=dev).bfloat16()
    res = torch.randn(M, C, device=dev).bfloat16()
    out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    sc, sh, mean, rstd, gamma = (torch.rand(C, device=dev) + 0.5 for _ in range(5))
    nb = lib.gdl_bn_bwd_blocks(M, C)
    part = torch.empty(nb, C, 2, device=dev)
    coef = torch.rand(2, C, device=dev)
    nbytes = M * C * 2
    p = lambda t: t.data_ptr()
    t1 = timeit(lambda: L.call("gdl_bn_act", dt, p(y), p(sc), p(sh), None, None, None, 1, p(out), M, C, st))
    t2 = timeit(lambda: L.call("gdl_bn_act", dt, p(y), p(sc), p(sh), p(res), None, None, 1, p(out), M, C, st))
    t3 = timeit(lambda: L.call("gdl_bn_bwd_reduce", dt, p(g), p(y), p(sc), p(sh), p(mean), p(rstd), 1, p(part), M, C, st))
    t4 = timeit(lambda: L.call("gdl_bn_bwd_apply", dt, p(g), p(y), p(sc), p(sh), p(mean), p(rstd), p(gamma), p(coef), 1, p(out),
                               M, C, st))
    t5 = timeit(lambda: out.copy_(y))
    gb = lambda k, t: k * nbytes / t / 1e6
    print(f"{name:10s} M={M:7d} C={C:3d} {nbytes / 1e6:6.1f} MB | bn_act {t1 * 1e3:6.1f} us {gb(2, t1):5.0f} GB/s | +res {t2 * 1e3:6.1f} us "
          f"{gb(3, t2):5.0f} | bwd_reduce {t3 * 1e3:6.1f} us {gb(2, t3):5.0f} | bwd_apply {t4 * 1e3:6.1f} us {gb(3, t4):5.0f} | "
          f"torch copy {t5 * 1e3:6.1f} us {gb(2, t5):5.0f}")
