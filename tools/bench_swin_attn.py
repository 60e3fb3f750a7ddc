#!/usr/bin/env python3
"""WindowAttention forward / backward alone at the four Swin-T stage shapes of config 5 (192 frames): time per launch and the
rate against the algorithmic bytes (forward: q, k, v read + out written = 4 M ld elements; backward: q, k, v, dout read + dqkv
written = 7 M ld).  `--check` compares every output of the library against a second call (run-to-run) and prints checksums, so
that two builds can be compared bit for bit."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=192)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
lib = L.load()
dt = L.dtype_code("bf16")
st = L.cur_stream()
N, ws = a.frames, 7
tot = {"fwd": 0.0, "bwd": 0.0}
floor = {"fwd": 0.0, "bwd": 0.0}
print("stage  res heads  ld shift |  fwd us   TB/s |  bwd us   TB/s")
for stage, (r, nh, ld, depth) in enumerate(((56, 3, 128, 2), (28, 6, 192, 2), (14, 12, 384, 6), (7, 24, 768, 2))):
    M = N * r * r
    g = torch.Generator(device=dev).manual_seed(stage)
    qkv = (torch.randn(M, 3 * ld, device=dev, generator=g) * 0.5).bfloat16()
    do = (torch.randn(M, ld, device=dev, generator=g) * 0.5).bfloat16()
    out = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
    dq = torch.empty_like(qkv)
    table = torch.randn((2 * ws - 1) ** 2, nh, device=dev, generator=g) * 0.02
    dtab = torch.empty_like(table)
    tpart = torch.empty(lib.gdl_swin_attn_bwd_workspace_bytes(N, r, r, ws, nh), dtype=torch.uint8, device=dev)
    for shift in (0, 3) if r > ws else (0,):
        def fwd():
            L.call("gdl_swin_attn_fwd", dt, L.ptr(qkv), L.ptr(table), L.ptr(out), N, r, r, ws, shift, nh, ld, st)

        def bwd():
            L.call("gdl_swin_attn_bwd", dt, L.ptr(qkv), L.ptr(table), L.ptr(do), L.ptr(dq), L.ptr(dtab), L.ptr(tpart), N, r, r, ws,
                   shift, nh, ld, st)

        res = {}
        for name, fn, units in (("fwd", fwd, 4), ("bwd", bwd, 7)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            res[name] = (us, units * M * ld * 2 / us / 1e6)
            nl = depth / 2 if r > ws else depth  # half of a stage's blocks are shifted
            tot[name] += us * nl / 1e3
            floor[name] += units * M * ld * 2 / 6e12 * 1e3 * nl
        print(f"s{stage}    {r:4d} {nh:5d} {ld:4d} {shift:5d} | {res['fwd'][0]:7.1f} {res['fwd'][1]:6.2f} | {res['bwd'][0]:7.1f} {res['bwd'][1]:6.2f}")
        if a.check:
            sums = [float(t.float().abs().sum()) for t in (out, dq, dtab)]
            o1, d1, t1 = out.clone(), dq.clone(), dtab.clone()
            fwd()
            bwd()
            torch.cuda.synchronize()
            same = bool((o1 == out).all()) and bool((d1 == dq).all()) and bool((t1 == dtab).all())
            print(f"       checksums out {sums[0]:.6e} dqkv {sums[1]:.6e} dtable {sums[2]:.6e}  run-to-run identical: {same}")
print(f"per step (12 blocks): forward {tot['fwd']:.3f} ms, backward {tot['bwd']:.3f} ms; at 6 TB/s: {floor['fwd']:.3f} / {floor['bwd']:.3f} ms")
