#!/usr/bin/env python3
"""3x3 stride-1 data gradients of the >= 128-channel layers at the B = 64 shapes, alone: bare (gdl_conv_dgrad), and as the engine
launches them -- with ReLU bits and the BatchNorm-backward sums against one partner tensor (conv2's), or with the addend and two
partners as well (conv1's).  usage (GPU box, repo root): [GDL_LIB=...] python3 tools/bench_dgrad_bn.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

SHAPES = [("visual L2", 192, 128, 28, 28), ("visual L3", 192, 256, 14, 14), ("visual L4", 192, 512, 7, 7),
          ("audio L2", 64, 128, 33, 24), ("audio L3", 64, 256, 17, 12), ("audio L4", 64, 512, 9, 6)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    lib = L.load()
    dt = L.dtype_code("bf16")
    dev = "cuda:0"
    st = L.cur_stream()
    print(f"{'layer':10s} {'bare us':>8s} {'bits+bn1':>9s} {'+add+bn2':>9s} {'forward':>8s}")
    for name, N, C, H, W in SHAPES:
        K = C
        dy = torch.randn(N, H, W, K, device=dev).to(torch.bfloat16)
        x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
        y1 = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
        y2 = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
        add = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
        wc = (torch.randn(C, 3, 3, K, device=dev) * 0.05).to(torch.bfloat16)
        wk = (torch.randn(K, 3, 3, C, device=dev) * 0.05).to(torch.bfloat16)
        dx = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
        yo = torch.empty(N, H, W, K, device=dev, dtype=torch.bfloat16)
        bits = torch.randint(0, 256, (N * H * W * C // 8,), device=dev, dtype=torch.uint8)
        mean, rstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        tabs = []
        for mode in (0, 1):
            t = torch.empty(lib.gdl_conv_table_bytes(mode, N, H, W, 3, 3, 1, 1), dtype=torch.uint8, device=dev)
            L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, 3, 3, 1, 1, t.data_ptr(), st)
            tabs.append(t)
        tiles = lib.gdl_conv_dgrad_bn_tiles(dt, N, H, W, C, K, 3, 3, 1, 1)
        p1, p2 = torch.empty(tiles, C, 2, device=dev), torch.empty(tiles, C, 2, device=dev)
        ft = lib.gdl_conv_bn_tiles(dt, N, H, W, C, K, 3, 3, 1, 1)
        fp = torch.empty(ft, K, 2, device=dev)

        def bare():
            L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, tabs[1].data_ptr(), N, H, W, C, K, 3, 3, 1, 1, st)

        def lean():
            L.call("gdl_conv_dgrad_bn", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, bits.data_ptr(), tabs[1].data_ptr(), N, H,
                   W, C, K, 3, 3, 1, 1, y1.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p1.data_ptr(), None, None, None, None, st)

        def rich():
            L.call("gdl_conv_dgrad_bn", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), add.data_ptr(), bits.data_ptr(),
                   tabs[1].data_ptr(), N, H, W, C, K, 3, 3, 1, 1, y1.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p1.data_ptr(),
                   y2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p2.data_ptr(), st)

        def fwd():
            L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), yo.data_ptr(), fp.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, 3, 3,
                   1, 1, st)

        print(f"{name:10s} {timeit(bare):8.1f} {timeit(lean):9.1f} {timeit(rich):9.1f} {timeit(fwd):8.1f}", flush=True)


if __name__ == "__main__":
    main()
