#!/usr/bin/env python3
"""How long each encoder's forward + backward takes when it has the GPU to itself (same engines, same batch as
bench.py) -- compare with the step time to see how much the concurrent streams cost each other."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, ROOT)
from gdl.trainer import DGLTrainer  # noqa: E402
from models.basic_model import AVClassifier_DGL  # noqa: E402
import argparse  # noqa: E402

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
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    tr.step(spec, image, label)
torch.cuda.synchronize()
print(f"full step: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
nf = tr.nf
for name, eng, x, g, df, feat in (("visual", tr.eng_v, image, tr.gviews[nf + 60:nf + 120], tr.dfv, tr.fv),
                                  ("audio", tr.eng_a, spec.unsqueeze(1), tr.gviews[nf:nf + 60], tr.dfa, tr.fa)):
    for phase in ("fwd", "bwd", "fwd+bwd"):
        def run():
            if "fwd" in phase:
                eng.forward(x, True, feat_out=feat)
            if "bwd" in phase:
                eng.backward(g, dfeat=df)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        print(f"{name} {phase} alone: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
