#!/usr/bin/env python3
"""The data-parallel schedule variants of DGLTrainer with a ONE-rank RCCL group on one GPU (no collective kernel runs: a proxy for
what the schedules cost by themselves; `bench.py --gpus N` times the same variants with real traffic, comm.schedule_variants_ms)."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl.trainer import DGLTrainer  # noqa: E402
from models.basic_model import AVClassifier_DGL  # noqa: E402
from utils.utils import setup_seed, weight_init  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--no-group", action="store_true", help="the same variants without a process group")
ap.add_argument("--emulate-traffic", type=int, default=0,
                help="N > 0: every bucket's (no-op, one rank) all-reduce is followed by N in-place passes over the bucket on a stream of "
                     "its own, behind the producer's event and in front of the optimizer -- a busy fifth hardware queue, ~N x 10-15 us "
                     "per 33 MB bucket; what a real collective's kernels do to the schedule can only be measured on N > 1 GPUs")
ap.add_argument("--traffic-on-caller", action="store_true",
                help="with --emulate-traffic: the emulated collectives run on the stream step() is called on (the lane that idles in "
                     "the data-parallel mode) instead of a stream of their own -- VERDICT r4 next #7: no fifth hardware queue, so the "
                     "visual side stream and the early backward can stay")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
pg = None
if not a.no_group:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1, device_id=dev)
    pg = dist.group.WORLD
setup_seed(0)
args = argparse.Namespace(fusion_method="concat", dataset="CREMAD", modality="full", batch_size=64)
model = AVClassifier_DGL(args)
model.apply(weight_init)
model.to(dev).train()
tr = DGLTrainer(model, lr=2e-3, alpha=4.0, momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype="bf16", process_group=pg)
B = 64
data = [(torch.randn(B, 257, 188, device=dev), torch.randn(B, 3, 3, 224, 224, device=dev), torch.randint(0, 6, (B,), device=dev))
        for _ in range(4)]


def timed(n):
    for i in range(5):
        tr.step(*data[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(*data[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tr.step(*data[0])
if a.emulate_traffic and tr.reducer is not None:
    red = tr.reducer
    red.force_comm = True
    cs = torch.cuda.current_stream(dev) if a.traffic_on_caller else torch.cuda.Stream(device=dev)
    launch0, wait0 = red.launch, red.wait_all
    evs = []

    def launch(name):
        launch0(name)
        lo, hi = red.buckets[name]
        cs.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cs):
            seg = red.flat[lo:hi]
            for _ in range(a.emulate_traffic):
                seg.mul_(1.0)
        evs.append(cs.record_event())

    def wait_all():
        wait0()
        for e in evs:
            torch.cuda.current_stream(dev).wait_event(e)
        evs.clear()

    red.launch, red.wait_all = launch, wait_all
print("default:", "visual", "caller" if tr.visual_on_caller else tr.eng_v.lane(), "early", tr.early_backward, "audio lane", tr.audio_on_caller,
      f"{timed(a.steps):.3f} ms")
for rnd in range(2):
    for vis in ("off", "owned", "caller"):
        for early in (False, True):
            for lane in (False, True):
                if vis == "caller" and lane:
                    continue
                tr.visual_on_caller = vis == "caller"
                tr.eng_v.side_stream(vis == "owned")
                tr.early_backward = early
                tr.audio_on_caller = lane
                print(f"round {rnd}  visual_wgrad_{vis:6s} early_backward_{'on ' if early else 'off'} audio_lane_{'on ' if lane else 'off'}  {timed(a.steps):.3f} ms")
if pg is not None:
    torch.cuda.synchronize()
    dist.destroy_process_group()
