#!/usr/bin/env python3
"""When each encoder chain starts and ends inside a step (events on the chains' own streams, no profiler):
how much of the step the two chains really overlap."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, ROOT)
from gdl.trainer import DGLTrainer  # noqa: E402
from models.basic_model import AVClassifier_DGL  # noqa: E402

dev = torch.device("cuda:0")
args = argparse.Namespace(fusion_method="concat", dataset="CREMAD", modality="full", batch_size=64)
model = AVClassifier_DGL(args).to(dev)
tr = DGLTrainer(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, alpha=4.0, dtype="bf16")
B = 64
spec = torch.randn(B, 257, 188, device=dev)
image = torch.randn(B, 3, 3, 224, 224, device=dev)
label = torch.randint(0, 6, (B,), device=dev)
for _ in range(5):
    tr.step(spec, image, label)
marks = []
orig_fwd_v, orig_fwd_a = tr.eng_v.forward, tr.eng_a.forward
orig_bwd_v, orig_bwd_a = tr.eng_v.backward, tr.eng_a.backward


def wrap(fn, name):
    def inner(*a, **k):
        s = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        r = fn(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(s)
        marks.append((name, e0, e1))
        return r
    return inner


tr.eng_v.forward, tr.eng_a.forward = wrap(orig_fwd_v, "visual fwd"), wrap(orig_fwd_a, "audio fwd")
tr.eng_v.backward, tr.eng_a.backward = wrap(orig_bwd_v, "visual bwd"), wrap(orig_bwd_a, "audio bwd")
for it in range(3):
    marks.clear()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record(torch.cuda.current_stream())
    tr.step(spec, image, label)
    t1 = torch.cuda.Event(enable_timing=True)
    t1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    print(f"step {it}: {t0.elapsed_time(t1):.3f} ms")
    for name, e0, e1 in marks:
        print(f"   {name:11s} {t0.elapsed_time(e0):7.3f} -> {t0.elapsed_time(e1):7.3f} ms")
