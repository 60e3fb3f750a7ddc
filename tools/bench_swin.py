#!/usr/bin/env python3
"""Swin-T visual encoder alone (forward + backward, B x T frames at 224 x 224): time per pass and the per-kernel table
of the library's measurement tap."""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, ROOT)
from gdl import _lib as L  # noqa: E402
from gdl.swin import SwinEngine  # noqa: E402
from oracle import fixtures as fx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = "cuda:0"
lib = L.load()
cfg = fx.SWIN_T
eng = SwinEngine(cfg, a.dtype, a.batch, a.frames, dev)
if os.environ.get("GDL_NOGRAPH"):
    eng.use_graph = False
params = [torch.randn(s, device=dev) * 0.02 for _, s in eng.param_shapes()]
for (n, _), p in zip(eng.param_shapes(), params):
    if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("norm.weight"):
        p.fill_(1.0)
eng.set_params(params)
grads = [torch.empty_like(p) for p in params]
x = torch.randn(a.batch, 3, a.frames, 224, 224, device=dev)
df = torch.randn(a.batch * a.frames, 768, device=dev)
for _ in range(2):
    eng.forward(x)
    eng.backward(df, grads)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    eng.forward(x)
torch.cuda.synchronize()
tf = (time.perf_counter() - t0) / a.iters * 1e3
t0 = time.perf_counter()
for _ in range(a.iters):
    eng.forward(x)
    eng.backward(df, grads)
torch.cuda.synchronize()
tfb = (time.perf_counter() - t0) / a.iters * 1e3
n_img = a.batch * a.frames
gf = 4.5 * n_img  # ~4.5 GFLOP per 224 x 224 image forward (Swin-T)
print(f"Swin-T {a.dtype} {n_img} frames: forward {tf:.2f} ms, forward+backward {tfb:.2f} ms "
      f"({n_img / tfb * 1e3:.0f} frames/s, ~{3 * gf / tfb:.0f} TFLOP/s of model arithmetic)")
lib.gdl_prof_set_filter(None)
lib.gdl_prof_enable(1)
eng.forward(x)
eng.backward(df, grads)
torch.cuda.synchronize()
lib.gdl_prof_enable(0)
ns = lib.gdl_prof_nslots()
n_l, n_ms, n_w = (ctypes.c_int64 * ns)(), (ctypes.c_double * ns)(), (ctypes.c_double * ns)()
L.call("gdl_prof_collect", n_l, n_ms, n_w)
rows = sorted(((n_ms[s], n_l[s], lib.gdl_prof_slot_name(s).decode()) for s in range(ns) if n_l[s]), reverse=True)
tot = sum(r[0] for r in rows)
print(f"kernel time {tot:.2f} ms in {sum(r[1] for r in rows)} launches")
for ms, n, name in rows[:int(os.environ.get("GDL_BENCH_SWIN_ROWS", "14"))]:
    print(f"  {name[:66]:66s} n={n:4d}  {ms:7.3f} ms")
print("graphs:", {k[0]: (v["n"], v["g"] is not None) for k, v in eng._graphs.items()}, "use_graph", eng.use_graph)
