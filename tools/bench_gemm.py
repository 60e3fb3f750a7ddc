#!/usr/bin/env python3
"""The Swin-T Linears (the library's 1x1 gdl_conv_fwd_bias / gdl_conv_dgrad / gdl_conv_wgrad) one shape at a time at
B x T = 192 frames: ms, TFLOP/s and GB/s against the bytes each GEMM has to move (HBM floor) -- most of them are HBM-bound
(K = 96 ... 768), so the roofline that matters is bytes / 6 TB/s, not the MFMA peak."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402


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
    ap.add_argument("--frames", type=int, default=192)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dt, td, dev, st = L.GDL_BF16, torch.bfloat16, "cuda:0", L.cur_stream()
    lib = L.load()
    ld = lambda c: (c + 63) // 64 * 64
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    floor = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'shape (M x K -> N)':34s} x n | {'fwd ms':>7s} {'TF/s':>6s} {'GB/s':>6s} | {'dgrad':>7s} {'TF/s':>6s} {'GB/s':>6s} | {'wgrad':>7s} {'TF/s':>6s} {'GB/s':>6s}")
    for i, depth in enumerate((2, 2, 6, 2)):
        C = 96 << i
        M = a.frames * (56 >> i) ** 2
        for name, K, N, extra_out, res in (("qkv", ld(C), 3 * ld(C), 0, 0), ("proj", ld(C), ld(C), 0, 1), ("fc1", ld(C), ld(4 * C), 1, 0),
                                           ("fc2", ld(4 * C), ld(C), 0, 1)):
            x = torch.randn(M, K, device=dev).to(td)
            w = torch.randn(N, K, device=dev).to(td)
            wT = torch.randn(K, N, device=dev).to(td)
            y = torch.empty(M, N, device=dev, dtype=td)
            y2 = torch.empty(M, N, device=dev, dtype=td) if extra_out else None
            r = torch.randn(M, N, device=dev).to(td) if res else None
            b = torch.zeros(N, device=dev)
            dx = torch.empty(M, K, device=dev, dtype=td)
            dw = torch.empty(N, K, device=dev)
            tabs = []
            for mode, ch in ((0, K), (1, N)):
                t = torch.empty(lib.gdl_conv_table_bytes(mode, M, 1, 1, 1, 1, 1, 0), dtype=torch.uint8, device=dev)
                L.call("gdl_conv_build_table", mode, dt, M, 1, 1, K, N, 1, 1, 1, 0, L.ptr(t), st)
                tabs.append(t)
            nb = lib.gdl_conv_wgrad_workspace_bytes(dt, M, 1, 1, K, N, 1, 1, 1, 0)
            ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
            t_f = timeit(lambda: L.call("gdl_conv_fwd_bias", dt, L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(b), L.ptr(r) if res else None,
                                        L.ptr(y2) if extra_out else None, L.ptr(tabs[0]), M, 1, 1, K, N, 1, 1, 1, 0, st), a.iters)
            t_d = timeit(lambda: L.call("gdl_conv_dgrad", dt, L.ptr(y), L.ptr(wT), L.ptr(dx), None, L.ptr(tabs[1]), M, 1, 1, K, N, 1, 1,
                                        1, 0, st), a.iters)
            t_w = timeit(lambda: L.call("gdl_conv_wgrad", dt, L.ptr(y), L.ptr(x), L.ptr(dw), L.ptr(tabs[0]), M, 1, 1, K, N, 1, 1, 1, 0,
                                        L.ptr(ws), nb, st), a.iters)
            gf = 2.0 * M * K * N / 1e9
            by_f = 2.0 * M * (K + N * (1 + extra_out + res)) / 1e9
            by_d = 2.0 * M * (K + N) / 1e9
            by_w = 2.0 * M * (K + N) / 1e9
            print(f"s{i} {name:5s} {M:7d} x {K:4d} -> {N:4d}  x{depth:2d} | {t_f:7.3f} {gf / t_f:6.0f} {by_f / t_f * 1e3:6.0f} | "
                  f"{t_d:7.3f} {gf / t_d:6.0f} {by_d / t_d * 1e3:6.0f} | {t_w:7.3f} {gf / t_w:6.0f} {by_w / t_w * 1e3:6.0f}", flush=True)
            for k, t, by in (("fwd", t_f, by_f), ("dgrad", t_d, by_d), ("wgrad", t_w, by_w)):
                tot[k] += t * depth
                floor[k] += by / 6.0 * depth  # ms at 6 TB/s
    print("totals ms:", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
    print("HBM floor at 6 TB/s, ms:", {k: round(v, 2) for k, v in floor.items()}, "sum", round(sum(floor.values()), 2))


if __name__ == "__main__":
    main()
