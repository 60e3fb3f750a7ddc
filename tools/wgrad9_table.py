#!/usr/bin/env python3
"""The 9-tap weight gradient launch by launch: for each of the 26 launches of a CREMA-D B = 64 step -- 8
distinct shapes -- the decomposition (tiles, pixel slices, blocks per CU, stages per slice and their imbalance), the kernel's and
its fold's stand-alone time against the MFMA and HBM floors, the bytes of partials, and -- with a -DGDL_TIMING build
(GDL_LIB=.../build_timing/libgdl_hip.so) -- the per-wave cycle split prologue / per-stage (barrier, DMA issue + first reads,
compute) / epilogue.

    python3 tools/wgrad9_table.py                       # times (normal build)
    GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so python3 tools/wgrad9_table.py --cycles
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

# (encoder, layer, images, channels, H, W, launches per step): backbone.py:141-156 at B = 64, T = 3
SHAPES = [("audio", 1, 64, 64, 65, 47, 4), ("audio", 2, 64, 128, 33, 24, 3), ("audio", 3, 64, 256, 17, 12, 3), ("audio", 4, 64, 512, 9, 6, 3),
          ("visual", 1, 192, 64, 56, 56, 4), ("visual", 2, 192, 128, 28, 28, 3), ("visual", 3, 192, 256, 14, 14, 3), ("visual", 4, 192, 512, 7, 7, 3)]
MFMA, HBM = 2.5e15, 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", action="store_true", help="per-wave cycle split (needs the -DGDL_TIMING build)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    dt = L.dtype_code("bf16")
    dev = "cuda:0"
    st = L.cur_stream()
    print(f"# conv_wgrad9 per launch, B = 64 CREMA-D shapes; {torch.cuda.get_device_name(0)}; library {L.SO_PATH}")
    hdr = (f"{'layer':10s} {'n':>2s} {'M':>7s} {'C=K':>4s} {'tiles':>5s} {'slices':>6s} {'blocks':>6s} {'blk/CU':>6s} {'stg/slice':>10s} "
           f"{'imbal':>6s} {'kernel us':>9s} {'fold us':>8s} {'mfma us':>8s} {'hbm us':>7s} {'frac':>6s} {'partials MB':>11s} {'x alg.':>6s}")
    if a.cycles:
        hdr += f" | {'prologue':>8s} {'barrier/st':>10s} {'issue/st':>9s} {'compute/st':>10s} {'epilogue':>8s} {'steady %':>8s}"
    print(hdr)
    tot_k = tot_f = 0.0
    for enc, layer, N, C, H, W, n in SHAPES:
        K = C
        M = N * H * W
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        dy = torch.randn(N, H, W, K, device=dev).bfloat16()
        dw = torch.empty(K, C, 3, 3, device=dev)
        nb = lib.gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, 3, 3, 1, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        t = torch.empty(lib.gdl_conv_table_bytes(0, N, H, W, 3, 3, 1, 1), dtype=torch.uint8, device=dev)
        L.call("gdl_conv_build_table", 0, dt, N, H, W, C, K, 3, 3, 1, 1, t.data_ptr(), st)

        def run():
            L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), t.data_ptr(), N, H, W, C, K, 3, 3, 1, 1,
                   ws.data_ptr(), nb, st)

        for _ in range(5):
            run()
        torch.cuda.synchronize()
        lib.gdl_prof_set_filter(None)
        lib.gdl_prof_enable(1)
        for _ in range(a.reps):
            run()
        torch.cuda.synchronize()
        lib.gdl_prof_enable(0)
        ns = lib.gdl_prof_nslots()
        n_l, n_ms, n_w = (ctypes.c_int64 * ns)(), (ctypes.c_double * ns)(), (ctypes.c_double * ns)()
        L.call("gdl_prof_collect", n_l, n_ms, n_w)
        us = {}
        for s in range(ns):
            if n_l[s]:
                us[lib.gdl_prof_slot_name(s).decode()] = n_ms[s] / n_l[s] * 1e3
        k_us = us.get("gdl::conv_wgrad9_kernel", float("nan"))
        f_us = us.get("gdl::wgrad9_reduce_kernel", float("nan"))
        tiles = (K // 64) * (C // 64)
        # csrc/conv_wgrad9.hip plan_w9 restated (256 blocks aimed at, at least MIN_ST stages per slice: 56 since round 6; the
        # GDL_WGRAD9_MINST tuning knob is honoured like the library honours it); the workspace query above is the larger of this
        # and the per-tap kernel's need
        min_st = int(os.environ.get("GDL_WGRAD9_MINST", "56")) if os.environ.get("GDL_TUNING") == "1" else 56
        ns = max(1, min((256 + tiles - 1) // tiles, (M + min_st * 64 - 1) // (min_st * 64)))
        chunk = ((M + ns - 1) // ns + 63) // 64 * 64
        slices = (M + chunk - 1) // chunk
        stages = (M + 63) // 64
        per = chunk // 64
        last = stages - per * (slices - 1)
        blocks = tiles * slices
        flops = 2.0 * M * K * C * 9
        alg = 2.0 * M * (K + C) + 4.0 * K * C * 9
        part = slices * K * 9 * C * 4
        row = (f"{enc + ' L' + str(layer):10s} {n:2d} {M:7d} {C:4d} {tiles:5d} {slices:6d} {blocks:6d} {blocks / 256:6.2f} {per:4d}/{last:<5d} "
               f"{per / max(1, stages / slices):6.2f} {k_us:9.1f} {f_us:8.1f} {flops / MFMA * 1e6:8.1f} {alg / HBM * 1e6:7.1f} "
               f"{flops / MFMA * 1e6 / k_us:6.3f} {part / 1e6:11.1f} {(alg + 2 * part) / alg:6.2f}")
        tot_k += n * k_us
        tot_f += n * f_us
        if a.cycles:
            dbg = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
            rc = lib.gdl_debug_timing_buffer(dbg.data_ptr())
            if rc != 0:
                raise SystemExit("--cycles needs the -DGDL_TIMING build (GDL_LIB=...)")
            run()
            torch.cuda.synchronize()
            lib.gdl_debug_timing_buffer(None)
            d = dbg.cpu().numpy().reshape(-1, 8, 8)[:, :4].reshape(-1, 8)
            d = d[d[:, 0] != 0]
            nst = d[:, 6].mean()
            pro, bar, iss, cmp_, epi = (d[:, i].mean() for i in (1, 2, 3, 4, 5))
            life = pro + bar + iss + d[:, 7].mean() + cmp_ + epi
            row += (f" | {pro:8.0f} {bar / nst:10.0f} {(iss + d[:, 7].mean()) / nst:9.0f} {cmp_ / nst:10.0f} {epi:8.0f} "
                    f"{100 * (bar + iss + d[:, 7].mean() + cmp_) / life:7.1f}%")
        print(row)
    print(f"# per step: kernels {tot_k / 1e3:.3f} ms + folds {tot_f / 1e3:.3f} ms stand-alone (26 + 26 launches)")
    print("# stg/slice: stages (64 pixels) per slice / stages of the last slice; imbal: longest slice over the mean; x alg.: (algorithmic bytes +")
    print("# partials written + read back) / algorithmic bytes; frac: time at the MFMA peak / measured kernel time; cycles: means over the worker waves")


if __name__ == "__main__":
    main()
