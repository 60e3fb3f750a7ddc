#!/usr/bin/env python3
"""Phase breakdown of swin_attn_bwd_mfma_kernel (needs a -DGDL_TIMING build whose kernel accumulates s_memtime deltas per phase
into the debug buffer -- the instrumented source is not kept in the tree).  Stage-1 shape by default."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

dev = "cuda:0"
lib = L.load()
dt = L.dtype_code("bf16")
N, r, ws, nh, ld, shift = 192, 56, 7, 3, 128, 3
M = N * r * r
qkv = (torch.randn(M, 3 * ld, device=dev) * 0.5).bfloat16()
do = (torch.randn(M, ld, device=dev) * 0.5).bfloat16()
dq = torch.empty_like(qkv)
table = torch.randn((2 * ws - 1) ** 2, nh, device=dev) * 0.02
dtab = torch.empty_like(table)
tpart = torch.empty(lib.gdl_swin_attn_bwd_workspace_bytes(N, r, r, ws, nh), dtype=torch.uint8, device=dev)
dbg = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
st = L.cur_stream()


def run():
    L.call("gdl_swin_attn_bwd", dt, L.ptr(qkv), L.ptr(table), L.ptr(do), L.ptr(dq), L.ptr(dtab), L.ptr(tpart), N, r, r, ws, shift, nh, ld, st)


for _ in range(3):
    run()
torch.cuda.synchronize()
L.call("gdl_debug_timing_buffer", dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
L.call("gdl_debug_timing_buffer", None)
d = dbg.cpu().numpy()
d = d[d[:, 0] > 0]
tot = d[:, :5].sum(1)
print(f"{e0.elapsed_time(e1) * 1e3:.0f} us; {len(d)} blocks stamped; per block (all its windows), mean clk: total {tot.mean():.0f}")
for k, nm in enumerate(("bookkeeping + Q/K/V/dO loads", "S, softmax, dS (strip loop)", "P^T + dO^T + dV product", "dS tile + K^T + dQ product",
                        "dS^T + Q^T + dK product + d(table)")):
    print(f"  {nm:32s} {d[:, k].mean():9.0f}  {100 * d[:, k].sum() / tot.sum():5.1f} %")
