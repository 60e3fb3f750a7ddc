#!/usr/bin/env python3
"""Host-side enqueue time of one DGLTrainer.step vs its device time (is the step launch-bound?)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl.trainer import DGLTrainer  # noqa: E402
from models.basic_model import AVClassifier_DGL  # noqa: E402
from utils.utils import setup_seed, weight_init  # noqa: E402

dev = torch.device("cuda", 0)
setup_seed(0)
args = argparse.Namespace(fusion_method="concat", dataset="CREMAD", modality="full", batch_size=64)
model = AVClassifier_DGL(args)
model.apply(weight_init)
model.to(dev).train()
tr = DGLTrainer(model, lr=2e-3, alpha=4.0, momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype="bf16")
B = 64
spec = torch.randn(B, 257, 188, device=dev)
image = torch.randn(B, 3, 3, 224, 224, device=dev)
label = torch.randint(0, 6, (B,), device=dev)
for _ in range(5):
    tr.step(spec, image, label)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter()
    tr.step(spec, image, label)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# host time with an idle device (sync before every step): pure enqueue cost
idle = []
for _ in range(10):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.step(spec, image, label)
    idle.append(time.perf_counter() - a)
torch.cuda.synchronize()
print(f"20 steps: enqueue loop {t_enq * 1e3 / 20:.2f} ms/step, wall {t_all * 1e3 / 20:.2f} ms/step; "
      f"enqueue with idle device {sum(idle) / len(idle) * 1e3:.2f} ms/step (min {min(idle) * 1e3:.2f})")
