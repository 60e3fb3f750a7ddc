#!/usr/bin/env python3
"""LayerNorm forward / backward of the Swin branch per stage shape (bf16, 192 frames of 224 x 224): time and bytes/s of each
launch alone (HIP events, 30 launches after 5 warm-ups).  Knobs (with GDL_TUNING=1): GDL_SW_LN_CAP, GDL_SW_PBLOCKS."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

dev = "cuda:0"
lib = L.load()
dc = L.dtype_code("bf16")
st = L.cur_stream()
SHAPES = [("s0 norm", 602112, 96, 128), ("s1 norm", 150528, 192, 192), ("s2 norm", 37632, 384, 384), ("s3 norm", 9408, 768, 768),
          ("merge 0", 150528, 384, 384), ("merge 1", 37632, 768, 768), ("merge 2", 9408, 1536, 1536)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':10s} {'M':>7s} {'ld':>5s} | fwd us  TB/s | bwd us  TB/s | bwd+colsum us  TB/s")
for name, M, C, ld in SHAPES:
    x = torch.randn(M, ld, device=dev).bfloat16()
    dy = torch.randn(M, ld, device=dev).bfloat16()
    add = torch.randn(M, ld, device=dev).bfloat16()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    g, b = torch.ones(ld, device=dev), torch.zeros(ld, device=dev)
    stats = torch.empty(M, 2, device=dev)
    dgb = torch.empty(3, ld, device=dev)
    part = torch.empty(lib.gdl_swin_partial_bytes(2 * ld), dtype=torch.uint8, device=dev)
    tf = timed(lambda: L.call("gdl_swin_ln_fwd", dc, L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), M, C, ld, st))
    tb = timed(lambda: L.call("gdl_swin_ln_bwd", dc, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(g), L.ptr(add), L.ptr(dx), L.ptr(dgb),
                              L.ptr(part), M, C, ld, st))
    row = f"{name:10s} {M:7d} {ld:5d} | {tf:6.1f} {M * ld * 4 / tf / 1e6:5.2f} | {tb:6.1f} {M * ld * 8 / tb / 1e6:5.2f} |"
    if 3 * ld <= 2 * 3072 and 4 * (64 // min(64, max(16, 1 << (ld // 8 - 1).bit_length()))) * 3 * ld * 4 <= 65536:
        tc = timed(lambda: L.call("gdl_swin_ln_bwd_colsum", dc, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(g), L.ptr(add), L.ptr(dx),
                                  L.ptr(dgb), L.ptr(part), M, C, ld, st))
        row += f" {tc:6.1f} {M * ld * 8 / tc / 1e6:5.2f}"
    print(row)
