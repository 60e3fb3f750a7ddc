#!/usr/bin/env python3
"""gdl_linear_bwd (both gradients of a Linear from one pass over dy) against the pair gdl_conv_dgrad + gdl_conv_wgrad at the
stage-1 shape of config 5 (602 112 x 128 -> 384): us and TB/s of the bytes each form has to move."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdl import _lib as L  # noqa: E402
from gpu_util import gather_table  # noqa: E402

dev = "cuda:0"
lib = L.load()
dc = L.dtype_code("bf16")
st = L.cur_stream()
M, K, N = 602112, 128, 384
dy = (torch.randn(M, N, device=dev) * 0.5).bfloat16()
x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
wT = (torch.randn(K, N, device=dev) * 0.1).bfloat16()
dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
dw = torch.empty(N, K, device=dev)
nb = lib.gdl_linear_bwd_workspace_bytes(M, K, N)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
tab_d = gather_table(L.GATHER_DGRAD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
tab_f = gather_table(L.GATHER_FWD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
wsb = lib.gdl_conv_wgrad_workspace_bytes(dc, M, 1, 1, K, N, 1, 1, 1, 0)
ws2 = torch.empty(wsb, dtype=torch.uint8, device=dev)


x[:, 96:] = 0
wT[96:] = 0
db = torch.empty(N, device=dev)
part = torch.empty(lib.gdl_swin_partial_bytes(N), dtype=torch.uint8, device=dev)


def fused(kreal=128, bias=False):
    L.call("gdl_linear_bwd", dc, L.ptr(dy), L.ptr(x), L.ptr(wT), L.ptr(dx), L.ptr(dw), L.ptr(db) if bias else None, L.ptr(ws), nb, M, K,
           kreal, N, st)


def colsum():
    L.call("gdl_swin_colsum", dc, L.ptr(dy), None, L.ptr(db), L.ptr(part), M, N, st)


def pair():
    L.call("gdl_conv_dgrad", dc, L.ptr(dy), L.ptr(wT), L.ptr(dx), None, L.ptr(tab_d), M, 1, 1, K, N, 1, 1, 1, 0, st)
    L.call("gdl_conv_wgrad", dc, L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(tab_f), M, 1, 1, K, N, 1, 1, 1, 0, L.ptr(ws2), wsb, st)


for name, fn, nbytes in (("fused", fused, 2 * M * (N + 2 * K)), ("pair", pair, 2 * M * (2 * N + 2 * K)), ("fused", fused, 2 * M * (N + 2 * K)),
                         ("fused, 96 real columns", lambda: fused(96), 2 * M * (N + 2 * K)),
                         ("fused, 96 + bias gradient", lambda: fused(96, True), 2 * M * (N + 2 * K)), ("column sums alone", colsum, 2 * M * N)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:26s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:.0f} MB")
