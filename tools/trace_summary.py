#!/usr/bin/env python3
"""Per-kernel and per-queue summary of ONE iteration out of a rocprofv3 kernel trace (the rocpd SQLite database that
`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 <program>` writes as DIR/NAME_results.db): every dispatch of the iteration,
including the kernels that carry no event pair of the library's own tap (round 4 found 0.9 ms of the Swin step in three of those).

usage: python3 tools/trace_summary.py DB --marker SUBSTRING [--iteration -2] [--gaps US]
  --marker     a kernel that runs exactly once per iteration (e.g. swin_patch_gather, sgd_kernel): iteration k = the dispatches from
               its k-th occurrence up to the next one
  --iteration  which one (negative: from the end; the last complete one is -2)
  --gaps US    also list the idle gaps longer than US microseconds per hardware queue
Tracing serialises the host (every launch is intercepted): per-kernel durations are trustworthy, the overlap between queues
and the start times are NOT those of an untraced run (tools/host_time_step.py measures those with events)."""
import argparse
import re
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--marker", required=True)
ap.add_argument("--iteration", type=int, default=-2)
ap.add_argument("--gaps", type=float, default=0.0)
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
rows = list(sqlite3.connect(a.db).execute("select name, start, end, queue_id from kernels order by start"))
cut = re.compile(r"\(.*")
idx = [i for i, r in enumerate(rows) if a.marker in r[0]]
if len(idx) < 2:
    raise SystemExit(f"marker {a.marker!r}: {len(idx)} dispatches in {len(rows)}")
k = a.iteration if a.iteration >= 0 else len(idx) + a.iteration
if not 0 <= k < len(idx) - 1:
    raise SystemExit(f"iteration {a.iteration} of {len(idx) - 1} complete ones")
seg = rows[idx[k]:idx[k + 1]]
t0 = seg[0][1]
print(f"iteration {k} of {len(idx) - 1}: {len(seg)} dispatches, {(max(r[2] for r in seg) - t0) / 1e6:.3f} ms from the first start to the last end, "
      f"{sum(r[2] - r[1] for r in seg) / 1e6:.3f} ms of kernel time")
per_q = defaultdict(list)
for n, s, e, q in seg:
    per_q[q].append((s, e, cut.sub("", n).replace("void ", "").replace("gdl::", "")))
for q, v in sorted(per_q.items()):
    print(f"  queue {q}: {len(v):4d} dispatches, busy {sum(e - s for s, e, _ in v) / 1e6:7.3f} ms, from {(v[0][0] - t0) / 1e6:7.3f} to "
          f"{(max(e for _, e, _ in v) - t0) / 1e6:7.3f} ms")
    if a.gaps > 0:
        last = None
        for s, e, n in v:
            if last is not None and (s - last) / 1e3 > a.gaps:
                print(f"      idle {(s - last) / 1e3:8.1f} us before {n[:60]} at {(s - t0) / 1e6:.3f} ms")
            last = e if last is None else max(last, e)
agg = defaultdict(lambda: [0, 0])
for n, s, e, q in seg:
    key = cut.sub("", n).replace("void ", "")
    agg[key][0] += 1
    agg[key][1] += e - s
print(f"{'kernel':86s} {'n':>4s} {'us':>9s} {'avg us':>8s}")
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{key[:86]:86s} {n:4d} {t / 1e3:9.1f} {t / n / 1e3:8.1f}")
