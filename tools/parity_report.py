#!/usr/bin/env python3
"""Worst observed deviations of DGLTrainer from every reference golden (tests/golden/*.npz), per case and dtype:
the numbers the tolerances in tests/test_step_gpu.py are set against.  Run on the GPU box from the repo root."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_step_gpu as T  # noqa: E402
from gdl.trainer import DGLTrainer  # noqa: E402

for name in T.STEP_CASES:
    g = T._gold(name)
    cfg = json.loads(str(g["config"]))
    for dtype in ("f32", "bf16"):
        model = T._make_model(cfg, dtype)
        model.train()
        tr = DGLTrainer(model, lr=cfg["lr"], alpha=cfg["alpha"], mode=cfg["mode"])
        for st in range(cfg["steps"]):
            spec, image, label = T._batch(cfg, st)
            tr.step(spec, image, label)
            r = tr.read()
            pre = f"s{st}."
            lg = max(float(np.abs(r[k] - g[pre + k]).max()) for k in ("out", "out_a", "out_v") if pre + k in g.files and k in r)
            ls = max(abs(r[k] - float(g[pre + k])) for k in ("loss_f", "loss_a", "loss_v") if pre + k in g.files and k in r)
            tn = abs(r["total_norm"] - float(g[pre + "total_norm"])) / float(g[pre + "total_norm"])
            names = [str(n) for n in g[pre + "grad_names"]]
            gn, isnone = g[pre + "grad_norm"], g[pre + "grad_is_none"]
            tot = float(g[pre + "total_norm"])
            clip = min(1.0, 40.0 / (tot + 1e-6))
            rel = [(abs(r["grad_norm"][n] - gn[i]) / max(gn[i], 1e-4 * clip * tot), n) for i, n in enumerate(names) if not isnone[i]]
            worst = max(rel)
            print(f"{name:20s} {dtype:4s} step {st}: logits {lg:.2e}  loss {ls:.2e}  total_norm {tn:.2e}  per-param {worst[0]:.2e} ({worst[1]})")
