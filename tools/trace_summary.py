#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv or csv.gz): per-step wall time, per-queue busy time and gaps."""
import collections, csv, gzip, sys
f = sys.argv[1]
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = sg[-2] + 1, sg[-1] + 1
step = rows[a:b]
s0, s1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"last step: wall {(s1 - s0) / 1e3:.1f} us, {len(step)} launches")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    gaps = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"])) / 1e3 for i in range(len(rs) - 1)]
    big = sorted(gaps, reverse=True)[:3]
    print(f"  queue {q}: {len(rs)} launches, busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, first start +{(int(rs[0]['Start_Timestamp']) - s0) / 1e3:.1f} us, largest gaps {['%.1f' % g for g in big]}")
    import statistics
    hist = collections.Counter(min(int(g // 2) * 2, 20) for g in gaps)
    print("     gap histogram (us, 2-us bins, 20 = >=20):", dict(sorted(hist.items())), f"median {statistics.median(gaps):.1f} sum {sum(gaps) / 1:.0f} us")
    if "-v" in sys.argv:
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for i, r in enumerate(rs):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if i + 1 < len(rs):
                agg[k][2] += gaps[i]
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"       {k:60s} n={v[0]:3d} busy {v[1]:8.1f} us  gap-after {v[2]:7.1f} us")
