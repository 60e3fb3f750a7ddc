#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels at the CREMA-D B=64 shapes.
Prints achieved TFLOP/s for forward / dgrad / wgrad of every distinct conv shape."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

SHAPES = [  # enc, Nimg/B, C, H, W, K, R, stride, pad, count
    ("audio", 1, 64, 65, 47, 64, 3, 1, 1, 4), ("audio", 1, 64, 65, 47, 128, 3, 2, 1, 1),
    ("audio", 1, 128, 33, 24, 128, 3, 1, 1, 3), ("audio", 1, 64, 65, 47, 128, 1, 2, 0, 1),
    ("audio", 1, 128, 33, 24, 256, 3, 2, 1, 1), ("audio", 1, 256, 17, 12, 256, 3, 1, 1, 3),
    ("audio", 1, 256, 17, 12, 512, 3, 2, 1, 1), ("audio", 1, 512, 9, 6, 512, 3, 1, 1, 3),
    ("visual", 3, 64, 56, 56, 64, 3, 1, 1, 4), ("visual", 3, 64, 56, 56, 128, 3, 2, 1, 1),
    ("visual", 3, 128, 28, 28, 128, 3, 1, 1, 3), ("visual", 3, 64, 56, 56, 128, 1, 2, 0, 1),
    ("visual", 3, 128, 28, 28, 256, 3, 2, 1, 1), ("visual", 3, 256, 14, 14, 256, 3, 1, 1, 3),
    ("visual", 3, 256, 14, 14, 512, 3, 2, 1, 1), ("visual", 3, 512, 7, 7, 512, 3, 1, 1, 3),
]


def timeit(fn, iters):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--miopen", action="store_true",
                    help="also time torch.nn.functional.conv2d (MIOpen, channels_last, the same dtype) forward / input gradient / "
                         "weight gradient per shape: a comparator column, never the product path")
    a = ap.parse_args()
    dt = L.dtype_code(a.dtype)
    td = L.torch_dtype(dt)
    dev = "cuda:0"
    st = L.cur_stream()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    mtot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    if a.miopen:
        torch.backends.cudnn.benchmark = True  # MIOpen find mode: the library's best kernel per shape
    print(f"{'shape':46s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}"
          + ("   || MIOpen fwd / dgrad / wgrad ms" if a.miopen else ""))
    for enc, mul, C, H, W, K, R, stride, pad, cnt in SHAPES:
        N = a.batch * mul
        P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(N, H, W, C, device=dev).to(td)
        dy = torch.randn(N, P, Q, K, device=dev).to(td)
        wk = torch.randn(K, R, R, C, device=dev).to(td)
        wc = torch.randn(C, R, R, K, device=dev).to(td)
        y = torch.empty(N, P, Q, K, device=dev, dtype=td)
        dx = torch.empty(N, H, W, C, device=dev, dtype=td)
        dw = torch.empty(K, C, R, R, device=dev)
        tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
        part = torch.empty(tiles, K, 2, device=dev)
        nb = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        tabs = []
        for mode in (0, 1):
            t = torch.empty(L.load().gdl_conv_table_bytes(mode, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
            L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
            tabs.append(t)
        tf, td_ = tabs[0].data_ptr(), tabs[1].data_ptr()
        gf = 2.0 * N * P * Q * K * C * R * R / 1e9
        t_f = timeit(lambda: L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tf, N, H,
                                    W, C, K, R, R, stride, pad, st), a.iters)
        t_d = timeit(lambda: L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, td_, N, H, W, C,
                                    K, R, R, stride, pad, st), a.iters)
        t_w = timeit(lambda: L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tf, N, H, W, C, K, R, R,
                                    stride, pad, ws.data_ptr(), nb, st), a.iters)
        tot["fwd"] += t_f * cnt
        tot["dgrad"] += t_d * cnt
        tot["wgrad"] += t_w * cnt
        name = f"{enc} {C}x{H}x{W}->{K} {R}x{R}/{stride} x{cnt}"
        line = f"{name:46s} {gf:8.2f} | {t_f:8.3f} {gf / t_f:7.1f} | {t_d:8.3f} {gf / t_d:7.1f} | {t_w:8.3f} {gf / t_w:7.1f}"
        if a.miopen:
            import torch.nn.functional as F
            from torch.nn import grad as G

            xm = x.permute(0, 3, 1, 2)   # NCHW view of the NHWC storage = channels_last
            wm = wk.permute(0, 3, 1, 2)  # [K][C][R][S] view of [K][R][S][C] = channels_last
            dym = dy.permute(0, 3, 1, 2)
            m_f = timeit(lambda: F.conv2d(xm, wm, stride=stride, padding=pad), a.iters)
            m_d = timeit(lambda: G.conv2d_input(xm.shape, wm, dym, stride=stride, padding=pad), a.iters)
            m_w = timeit(lambda: G.conv2d_weight(xm, wm.shape, dym, stride=stride, padding=pad), a.iters)
            mtot["fwd"] += m_f * cnt
            mtot["dgrad"] += m_d * cnt
            mtot["wgrad"] += m_w * cnt
            line += f"   || {m_f:7.3f} {m_d:7.3f} {m_w:7.3f}"
        print(line, flush=True)
    print("totals (ms, weighted by layer count):", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
    if a.miopen:
        print("MIOpen totals (ms):", {k: round(v, 3) for k, v in mtot.items()}, "sum", round(sum(mtot.values()), 3))


if __name__ == "__main__":
    main()
