#!/usr/bin/env python3
"""Experiment (round 5): the Swin-T branch as SEVERAL concurrent chains.  The Swin encoder has no batch statistics, so the 192
frames of a B = 64, T = 3 step can run as k engines of 192 / k frames on k streams (parameter gradients add up).  One chain keeps
~1.3 kernels in flight (profiles/r04_trace_vggsound_swin_step.txt); does overlapping k chains shorten forward + backward?
    python3 tools/bench_swin_split.py [--splits 1 2 3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, ROOT)
from gdl.swin import SwinEngine  # noqa: E402
from oracle import fixtures as fx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = "cuda:0"
cfg = fx.SWIN_T
for k in a.splits:
    if a.batch % k:
        continue
    b = a.batch // k
    engs = [SwinEngine(cfg, "bf16", b, a.frames, dev) for _ in range(k)]
    params = [torch.randn(s, device=dev) * 0.02 for _, s in engs[0].param_shapes()]
    for (n, _), p in zip(engs[0].param_shapes(), params):
        if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("norm.weight"):
            p.fill_(1.0)
    for e in engs:
        e.set_params(params)
    grads = [[torch.empty_like(p) for p in params] for _ in range(k)]
    xs = [torch.randn(b, 3, a.frames, 224, 224, device=dev) for _ in range(k)]
    dfs = [torch.randn(b * a.frames, 768, device=dev) for _ in range(k)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]

    def step():
        cur = torch.cuda.current_stream()
        for i in range(k):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                engs[i].forward(xs[i])
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                engs[i].backward(dfs[i], grads[i])
        for i in range(k):
            cur.wait_stream(streams[i])
        for i in range(1, k):  # the partial gradients add up
            torch._foreach_add_(grads[0], grads[i])

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    print(f"Swin-T bf16, {a.batch * a.frames} frames as {k} chain(s) of {b * a.frames}: forward + backward {ms:.2f} ms")
    del engs, grads, xs, dfs, params
    torch.cuda.empty_cache()
