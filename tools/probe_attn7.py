#!/usr/bin/env python3
"""Phase breakdown of swin_attn7_bwd_kernel (needs a -DGDL_TIMING build: `make BUILD=build_timing EXTRA=-DGDL_TIMING`, run with
GDL_LIB=.../build_timing/libgdl_hip.so): every wave sums s_memtime deltas per phase over its windows.  Stage-1 shape by default;
argument: stage 0..3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

dev = "cuda:0"
lib = L.load()
dt = L.dtype_code("bf16")
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shift = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r, nh, ld = ((56, 3, 128), (28, 6, 192), (14, 12, 384), (7, 24, 768))[stage]
N, ws = 192, 7
M = N * r * r
qkv = (torch.randn(M, 3 * ld, device=dev) * 0.5).bfloat16()
do = (torch.randn(M, ld, device=dev) * 0.5).bfloat16()
dq = torch.empty_like(qkv)
table = torch.randn((2 * ws - 1) ** 2, nh, device=dev) * 0.02
dtab = torch.empty_like(table)
tpart = torch.empty(lib.gdl_swin_attn_bwd_workspace_bytes(N, r, r, ws, nh), dtype=torch.uint8, device=dev)
dbg = torch.zeros(4096 * 4, 8, dtype=torch.int64, device=dev)
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
d = d[d[:, 6] > 0]
nwin = d[:, 6].sum()
print(f"stage {stage} shift {shift}: {e0.elapsed_time(e1) * 1e3:.0f} us; {len(d)} waves stamped, {nwin / len(d):.1f} windows each; "
      f"wave life {d[:, 5].mean():.0f} s_memtime ticks")
names = ("offsets + loads issued + landed, dO rows -> LDS", "four strips: S, dP, softmax, dS, P -> tile", "dV = P^T dO (+ stores)",
         "dS -> tile, K rows, dQ = dS K, D += dS . I", "Q rows, dK = dS^T Q")
tot = d[:, :5].sum()
for k, nm in enumerate(names):
    print(f"  {nm:52s} {d[:, k].sum() / nwin:8.1f} ticks / window  {100 * d[:, k].sum() / tot:5.1f} %")
print(f"  sum {tot / nwin:.1f} ticks / window; outside the windows (bias tile, loop, fold): {(d[:, 5].sum() - tot) / len(d):.0f} ticks / wave")
