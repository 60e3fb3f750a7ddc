#!/usr/bin/env python3
"""The fused stem backward (gdl_stem_bwd_fused) alone at the CREMA-D B = 64 shapes: time per launch (kernel + fold) and, on the
-DGDL_TIMING build (GDL_LIB=.../build_timing/libgdl_hip.so, --cycles), the per-wave cycle split of the kernel: prologue, per
stage wait (memory + barrier) / emit + issue / multiply.  Random operands (timing only; parity: tests/test_ops_gpu.py)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cycles", action="store_true")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = L.load()
dt = L.dtype_code("bf16")
dev = "cuda:0"
st = L.cur_stream()
print(f"# gdl_stem_bwd_fused alone; {torch.cuda.get_device_name(0)}; library {L.SO_PATH}")
for name, n_img, H, W, Cin in (("visual", 192, 224, 224, 3), ("audio", 64, 257, 188, 1)):
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PP, QQ = (P - 1) // 2 + 1, (Q - 1) // 2 + 1
    C = 64
    xp = torch.randn(lib.gdl_stem_pad_bytes(dt, n_img, H, W) // 2, device=dev).bfloat16()
    y = torch.randn(n_img, P, Q, C, device=dev).bfloat16()
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    mean, rstd, gamma = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    coef = torch.randn(2 * C, device=dev) * 0.01
    out, ym = torch.empty(n_img, PP, QQ, C, device=dev, dtype=torch.bfloat16), torch.empty(n_img, PP, QQ, C, device=dev, dtype=torch.bfloat16)
    ix = torch.empty(n_img, PP, QQ, C, dtype=torch.uint8, device=dev)
    L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(ix), L.ptr(ym), n_img, P, Q, C, st)
    dz = torch.randn(n_img, PP, QQ, C, device=dev).bfloat16()
    nb = lib.gdl_stem_conv_wgrad_workspace_bytes(n_img, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dw = torch.empty(64, Cin, 7, 7, device=dev)

    def run():
        L.call("gdl_stem_bwd_fused", dt, L.ptr(dz), L.ptr(ix), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
               L.ptr(coef), L.ptr(xp), L.ptr(dw), n_img, H, W, Cin, L.ptr(ws), nb, st)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    mb = (y.numel() * 2 + dz.numel() * 3 + xp.numel() * 2) / 1e6
    row = f"{name:7s} {n_img} x {P} x {Q}: {us:7.1f} us per launch (kernel + fold), operands {mb:.0f} MB -> {mb / us * 1e3 / 1e3:.2f} TB/s"
    if a.cycles:
        dbg = torch.zeros(1 << 14, 8, dtype=torch.int64, device=dev)
        if lib.gdl_debug_timing_buffer(dbg.data_ptr()) != 0:
            raise SystemExit("--cycles needs the -DGDL_TIMING build (GDL_LIB=...)")
        run()
        torch.cuda.synchronize()
        lib.gdl_debug_timing_buffer(None)
        d = dbg.cpu().numpy()
        d = d[d[:, 0] != 0]
        nst = d[:, 6].mean()
        row += (f" | waves {len(d)}, stages/wave {nst:.1f}: prologue {d[:, 1].mean():.0f} clk, per stage wait {d[:, 2].mean() / nst:.0f} "
                f"emit+issue {d[:, 3].mean() / nst:.0f} multiply {d[:, 4].mean() / nst:.0f}, loop {d[:, 5].mean():.0f} clk")
    print(row)
