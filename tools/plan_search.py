#!/usr/bin/env python3
"""Coordinate search over the slab tile of each stride-1 3x3 layer shape, measured inside the step (GDL_PLAN, GDL_TUNING=1).
Run on the GPU box from the repo root: python tools/plan_search.py [steps]"""
import json
import os
import subprocess
import sys

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 64
SHAPES = [("v1", B * 3 * 56 * 56, 64), ("v2", B * 3 * 28 * 28, 128), ("v3", B * 3 * 14 * 14, 256), ("v4", B * 3 * 7 * 7, 512),
          ("a1", B * 65 * 47, 64), ("a2", B * 33 * 24, 128), ("a3", B * 17 * 12, 256), ("a4", B * 9 * 6, 512)]
OPTS = [(128, 64), (256, 64), (128, 128), (192, 128), (256, 128)]


def run(plan):
    env = dict(os.environ, GDL_TUNING="1", GDL_PLAN=",".join(f"{m}:{oc}:{bm}:{bn}" for (m, oc), (bm, bn) in plan.items()))
    best = 1e9
    for _ in range(2):
        out = subprocess.run([sys.executable, "bench.py", "--steps", str(steps), "--warmup", "8", "--no-cpu-baseline", "--no-f32",
                              "--no-prof"], env=env, capture_output=True, text=True).stdout
        for l in out.splitlines():
            if l.startswith("{"):
                best = min(best, json.loads(l)["ms_per_step"])
    return best


plan = {}
base = run(plan)
print(f"default plan: {base:.3f} ms", flush=True)
for name, m, oc in SHAPES:
    res = {}
    for bm, bn in OPTS:
        if oc % bn:
            continue
        trial = dict(plan)
        trial[(m, oc)] = (bm, bn)
        res[(bm, bn)] = run(trial)
    bo = min(res, key=res.get)
    print(name, {f"{k[0]}x{k[1]}": round(v, 3) for k, v in res.items()}, "->", bo, flush=True)
    if res[bo] < base - 0.01:
        plan[(m, oc)] = bo
        base = res[bo]
print("final plan:", {f"{k[0]}:{k[1]}": v for k, v in plan.items()}, f"{base:.3f} ms")
