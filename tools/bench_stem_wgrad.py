#!/usr/bin/env python3
"""The stem forward convolution and weight gradient alone at the CREMA-D B=64 shapes (visual 192x224x224x3, audio 64x257x188x1)."""
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
for name, n_img, H, W, Cin in (("visual", 192, 224, 224, 3), ("audio", 64, 257, 188, 1)):
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = torch.randn(lib.gdl_stem_pad_bytes(dt, n_img, H, W) // 2, device=dev).bfloat16()
    dy = torch.randn(n_img * P * Q, 64, device=dev).bfloat16()
    tab = torch.empty(lib.gdl_stem_table_bytes(n_img, H, W), dtype=torch.uint8, device=dev)
    L.call("gdl_stem_build_table", dt, n_img, H, W, L.ptr(tab), st)
    nb = lib.gdl_stem_conv_wgrad_workspace_bytes(n_img, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dw = torch.empty(64, Cin, 7, 7, device=dev)

    def run():
        L.call("gdl_stem_conv_wgrad", dt, L.ptr(dy), L.ptr(xp), L.ptr(dw), L.ptr(tab), n_img, H, W, Cin, L.ptr(ws), nb, st)

    wp = torch.randn(lib.gdl_stem_weight_bytes(dt) // 2, device=dev).bfloat16()
    y = torch.empty(n_img * P * Q, 64, device=dev, dtype=torch.bfloat16)
    part = torch.empty(lib.gdl_stem_conv_bn_tiles(dt, n_img, H, W), 64, 2, device=dev)

    def fwd():
        L.call("gdl_stem_conv_fwd", dt, L.ptr(xp), L.ptr(wp), L.ptr(y), None if os.environ.get("NOSTATS") else L.ptr(part), L.ptr(tab), n_img, H, W, Cin, st)

    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(20):
        fwd()
    f1.record()
    torch.cuda.synchronize()
    fus = f0.elapsed_time(f1) / 20 * 1e3
    print(f"{name}: forward {fus:.1f} us, y {y.numel() * 2 / 1e6:.0f} MB -> {y.numel() * 2 / fus / 1e3:.0f} GB/s written")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * n_img * P * Q * 64 * 49 * Cin / 1e9
    print(f"{name}: {us:.1f} us (kernel + reduce), {gf / us * 1e3:.0f} TFLOP/s algorithmic, dy {dy.numel() * 2 / 1e6:.0f} MB -> {dy.numel() * 2 / us / 1e3:.0f} GB/s")
