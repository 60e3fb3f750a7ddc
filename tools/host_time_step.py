#!/usr/bin/env python3
"""Host time of DGLTrainer.step() against the device time of the step (N = 1): how long the single host thread needs to ENQUEUE a
step (per part: the two encoders' passes, everything else) and whether the device ever waits for it.
usage: python3 tools/host_time_step.py [workload]   (default vggsound_swin)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
import bench  # noqa: E402
from gdl.trainer import DGLTrainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "vggsound_swin"
wl = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
B = 64
model, _ = bench.build_model(wl, B, dev)
tr = DGLTrainer(model, lr=2e-3, alpha=wl["alpha"], momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype="bf16")
g = torch.Generator(device="cpu").manual_seed(99)
data = [(torch.randn(B, *wl["spec"], generator=g).to(dev), torch.randn(B, 3, 3, 224, 224, generator=g).to(dev),
         torch.randint(0, wl["n_classes"], (B,), generator=g).to(dev)) for _ in range(4)]
acc = {}


def timed(obj, meth, key):
    f = getattr(obj, meth)

    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, meth, w)


tr.step(*data[0])  # (the engines are built by the first step)
for o, n in ((tr.eng_v, "visual"), (tr.eng_a, "audio")):
    timed(o, "forward", n + " forward")
    timed(o, "backward", n + " backward")
for i in range(15):
    tr.step(*data[i % 4])
torch.cuda.synchronize()
acc.clear()
K = 40
t0 = time.perf_counter()
host = 0.0
for i in range(K):
    t = time.perf_counter()
    tr.step(*data[i % 4])
    host += time.perf_counter() - t
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{name}: {wall / K * 1e3:.3f} ms per step on the device, {host / K * 1e3:.3f} ms of host time per step() call "
      f"(all {K} calls returned after {t_enq * 1e3:.1f} ms, the device finished after {wall * 1e3:.1f} ms)")
for k, v in sorted(acc.items()):
    print(f"  {k:18s} {v / K * 1e3:7.3f} ms per step")
print(f"  {'the rest':18s} {(host - sum(acc.values())) / K * 1e3:7.3f} ms per step")

# where the step ends on the device: the visual chain's last kernel against the optimizer's (events behind a step's enqueues)
tr.phase_events = []
rows = []
for i in range(12):
    tr.phase_events.clear()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    tr.step(*data[i % 4])
    ev, ea, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    ev.record(tr.s_v)
    ea.record(tr.s_a)
    ec.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    marks = {n: e0.elapsed_time(e) for n, e in tr.phase_events}
    rows.append((e0.elapsed_time(ev), e0.elapsed_time(ea), e0.elapsed_time(ec), marks))
v, a_, c, marks = rows[-1]
print(f"one step alone (device idle before it): visual chain done at {v:.3f} ms, audio / main chain at {a_:.3f}, caller's stream at {c:.3f}")
print("  marks on main:", ", ".join(f"{k} {t:.3f}" for k, t in marks.items()))
