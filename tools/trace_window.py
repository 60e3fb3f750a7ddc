#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: the launches between the last two `sgd_kernel` launches (one steady-state step), per queue in
start order: start us (from the window's begin), duration, blocks, name.  --grep keeps the rows whose name matches (with their
predecessor and successor on the queue)."""
import argparse
import csv
import glob
import os
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--grep", default=None)
    ap.add_argument("--anchor", default="sgd_kernel")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    anch = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
    lo, hi = anch[-2] + 1, anch[-1] + 1
    win = rows[lo:hi]
    t0 = int(win[0]["Start_Timestamp"])
    print(f"# window: {len(win)} launches, {(int(win[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; columns of the trace: {list(rows[0].keys())}")
    byq = {}
    for r in win:
        byq.setdefault((r.get("Queue_Id", "?"), r.get("Stream_Id", "?")), []).append(r)
    for q, rs in byq.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"## queue {q}: {len(rs)} launches, {busy:.1f} us busy")
        for i, r in enumerate(rs):
            if a.grep:
                near = any(re.search(a.grep, rs[j]["Kernel_Name"]) for j in (i - 1, i, i + 1) if 0 <= j < len(rs))
                if not near:
                    continue
            wg = int(r["Workgroup_Size_X"])
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} "
                  f"{int(r['Grid_Size_X']) // max(1, wg):6d}  {r['Kernel_Name'][:100]}")


if __name__ == "__main__":
    main()
