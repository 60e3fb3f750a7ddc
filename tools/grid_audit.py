#!/usr/bin/env python3
"""Chip-fill audit of one step from a rocprofv3 kernel trace: per (kernel, grid, workgroup) the launches per step, mean duration,
workgroups, workgroups per CU and what share of the step's kernel time the launches with fewer than one workgroup per CU hold.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline ...
    python3 tools/grid_audit.py gpurun_out/kt --steps 6
"""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=1, help="steps the trace covers (launch counts are divided by it)")
    ap.add_argument("--cus", type=int, default=256)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + a.dir)
    agg = collections.defaultdict(lambda: [0, 0.0])
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            key = (r["Kernel_Name"], grid // max(1, wg), wg)
            agg[key][0] += 1
            agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            meta[key] = (r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("Scratch_Size", ""))
    tot = sum(v[1] for v in agg.values())
    print(f"# {len(files)} trace file(s), {sum(v[0] for v in agg.values())} launches, {tot / a.steps / 1e3:.3f} ms of kernel time per step")
    print(f"{'us/step':>9s} {'n/step':>7s} {'us each':>8s} {'blocks':>7s} {'blk/CU':>6s} {'threads':>7s} {'LDS':>6s} {'VGPR':>5s} {'AGPR':>5s}  kernel")
    under = 0.0
    for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        name, blocks, wg = key
        lds, vg, ag, _ = meta[key]
        flag = " <" if blocks < a.cus else ""
        if blocks < a.cus:
            under += us
        if us / a.steps < 2.0:
            continue
        print(f"{us / a.steps:9.1f} {n / a.steps:7.2f} {us / n:8.1f} {blocks:7d} {blocks / a.cus:6.2f} {wg:7d} {lds:>6s} {vg:>5s} {ag:>5s}  {name[:110]}{flag}")
    print(f"# launches with fewer blocks than CUs: {under / a.steps / 1e3:.3f} ms of kernel time per step ({100 * under / tot:.1f} %)")


if __name__ == "__main__":
    main()
