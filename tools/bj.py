#!/usr/bin/env python3
"""Print value / ms_per_step (and optionally the kernel table) of bench.py JSON logs: tools/bj.py [-k] file..."""
import json
import sys

kt = "-k" in sys.argv
for f in [a for a in sys.argv[1:] if a != "-k"]:
    try:
        line = [x for x in open(f) if x.startswith("{")][-1]
    except (OSError, IndexError):
        print(f, "no JSON line")
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(f"{f}: {d['value']:.0f} samples/s  {d['ms_per_step']:.3f} ms/step  dominant {r.get('kernel')} frac {r.get('frac')} "
          f"avg {r.get('avg_launch_us')} us")
    if kt and d.get("kernels"):
        tot = 0.0
        for k in d["kernels"]:
            tot += k["ms_per_step"]
            print(f"   {k['kernel'][:62]:62s} n={k['launches_per_step']:5.1f} avg={k['avg_us']:7.1f} ms={k['ms_per_step']:.3f} frac={k['frac']:.3f}")
        print(f"   sum of kernel time per step {tot:.2f} ms")
