#!/usr/bin/env python3
"""Utilisation timeline of ONE step of the native runner: which kernels are resident per 100 us window,
how many lanes (streams) carry work, and how many algorithmic TFLOP/s and GB/s flow -- from the library's own timing tap
(`gdl_prof_timeline`: HIP events stamped with each kernel's begin / end, no tracer, the host is not serialised).

    python3 tools/utilisation_timeline.py [--workload cremad] [--window 100] [--steps 5] [--out profiles/r05_utilisation_timeline.txt]

What a tapped step is: every launch carries an event pair; an event-stamped launch does not pipeline with its neighbours on the
stream the way a plain one does, so the tapped step is a few per cent longer than the untapped one (both are printed).  The
tapped step is enqueued WITHOUT a synchronisation in front of it, behind `--lead` untapped steps and in front of two more: the
host is ahead of the device as in a training run (a step tapped right behind a synchronisation starts its second chain ~0.4 ms
late -- the host needs that long to enqueue the first chain's forward -- which no free-running step does).  The
flow columns price each launch by its ALGORITHMIC work (flops of the convolution; bytes its operands + result need once) spread
evenly over its own duration.  A window is flagged `<40%` when it is under 40 % of BOTH roofs (2.5 PFLOP/s dense bf16 MFMA,
8 TB/s HBM3E)."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-gdl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def short(name):
    n = name.replace("gdl::", "").replace("_kernel", "")
    n = n.replace("conv3x3_slab<bf16, ", "slab<").replace("conv_igemm<bf16, ", "igemm<").replace("<bf16>", "")
    return n.replace("<bf16, ", "<")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cremad")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--window", type=float, default=100.0, help="window length, us")
    ap.add_argument("--steps", type=int, default=5, help="tapped steps (the one with the median span is printed)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lead", type=int, default=6, help="untapped steps enqueued in front of each tapped one (no synchronisation between)")
    ap.add_argument("--out", default="")
    ap.add_argument("--launches", action="store_true", help="also list every launch (lane, start, end, kernel)")
    a = ap.parse_args()

    import bench
    from gdl import _lib as L
    from gdl.trainer import DGLTrainer

    lib = L.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.WORKLOADS[a.workload]
    a.dtype = "bf16"
    model, _ = bench.build_model(wl, a.batch, dev)
    tr = DGLTrainer(model, lr=2e-3, alpha=wl["alpha"], momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype="bf16")
    g = torch.Generator(device="cpu").manual_seed(1234)
    B = a.batch
    data = [(torch.randn(B, *wl["spec"], generator=g).to(dev), torch.randn(B, 3, 3, 224, 224, generator=g).to(dev),
             torch.randint(0, wl["n_classes"], (B,), generator=g).to(dev)) for _ in range(4)]
    for i in range(a.warmup):
        tr.step(*data[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        tr.step(*data[i % 4])
    torch.cuda.synchronize()
    untapped_ms = (time.perf_counter() - t0) / 40 * 1e3

    cap = 4096
    runs = []
    for s in range(a.steps):
        # free-running: the host is several steps ahead of the device (as in the timed region of bench.py) when the tapped
        # step is enqueued -- no synchronisation in front of it, untapped steps before and behind it
        for i in range(a.lead):
            tr.step(*data[i % 4])
        lib.gdl_prof_set_filter(None)
        lib.gdl_prof_enable(1)
        tr.step(*data[s % 4])
        lib.gdl_prof_enable(0)
        for i in range(2):
            tr.step(*data[i % 4])
        torch.cuda.synchronize()
        slot = (ctypes.c_int32 * cap)()
        lane = (ctypes.c_int32 * cap)()
        st = (ctypes.c_double * cap)()
        en = (ctypes.c_double * cap)()
        wk = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        n = lib.gdl_prof_timeline(cap, slot, lane, st, en, wk, by)
        if n < 0:
            raise SystemExit(f"gdl_prof_timeline: {L.last_error()}")
        n = min(n, cap)
        recs = []
        for i in range(n):
            name = lib.gdl_prof_slot_name(slot[i]).decode()
            mfma = lib.gdl_prof_slot_bound(slot[i]) == 1
            flops = wk[i] if mfma else 0.0
            nbytes = by[i] if mfma else wk[i]
            recs.append((lane[i], st[i] * 1e3, en[i] * 1e3, name, flops, nbytes))  # us
        ns = lib.gdl_prof_nslots()
        dummy = [(ctypes.c_int64 * max(ns, 1))(), (ctypes.c_double * max(ns, 1))(), (ctypes.c_double * max(ns, 1))()]
        L.call("gdl_prof_collect", *dummy)  # clears the records
        runs.append(recs)
    spans = [max(r[2] for r in recs) for recs in runs]
    pick = sorted(range(len(runs)), key=lambda i: spans[i])[len(runs) // 2]
    recs = runs[pick]
    span = spans[pick]

    out = []
    P = out.append
    P(f"# utilisation timeline of one tapped step -- workload {a.workload}, B = {B}, bf16; {torch.cuda.get_device_name(0)}")
    P(f"# untapped step {untapped_ms:.3f} ms (40 steps); tapped steps' spans (first kernel start -> last kernel end): "
      + ", ".join(f"{s / 1e3:.3f}" for s in spans) + f" ms; printed: the median one ({span / 1e3:.3f} ms, {len(recs)} launches)")
    lanes = sorted(set(r[0] for r in recs))
    P("# lanes (streams in order of first use by the step):")
    for ln in lanes:
        v = [r for r in recs if r[0] == ln]
        busy = sum(r[2] - r[1] for r in v)
        tf = sum(r[4] for r in v)
        P(f"#   lane {ln}: {len(v):3d} launches, from {min(r[1] for r in v) / 1e3:6.3f} to {max(r[2] for r in v) / 1e3:6.3f} ms, "
          f"busy {busy / 1e3:6.3f} ms, {tf / 1e9:8.1f} GFLOP; first {short(v[0][3])[:40]}, last {short(v[-1][3])[:40]}")
    tot_k = sum(r[2] - r[1] for r in recs)
    P(f"# sum of kernel durations {tot_k / 1e3:.3f} ms = {tot_k / span:.2f} kernels in flight on average; "
      f"{sum(r[4] for r in recs) / 1e12:.3f} TFLOP, {sum(r[5] for r in recs) / 1e9:.2f} GB algorithmic")
    P(f"# {'window ms':>11s} {'lanes':>5s} {'kern':>5s} {'TFLOP/s':>8s} {'%mfma':>6s} {'GB/s':>7s} {'%hbm':>5s}  flag   resident kernels (share of the window)")
    W = a.window
    nwin = int(span / W) + 1
    low = []
    for w in range(nwin):
        w0, w1 = w * W, (w + 1) * W
        fl = by = 0.0
        res = {}
        lanes_on = {}
        for ln, s, e, name, f, b in recs:
            ov = min(e, w1) - max(s, w0)
            if ov <= 0:
                continue
            d = max(e - s, 1e-3)
            fl += f * ov / d
            by += b * ov / d
            res[short(name)] = res.get(short(name), 0.0) + ov / W
            lanes_on[ln] = lanes_on.get(ln, 0.0) + ov / W
        tfs = fl / (W * 1e-6) / 1e12
        gbs = by / (W * 1e-6) / 1e9
        pm, ph = tfs / 2500.0, gbs / 8000.0
        flag = "<40%" if (pm < 0.4 and ph < 0.4) else ""
        names = ", ".join(f"{k} {v:.2f}" for k, v in sorted(res.items(), key=lambda kv: -kv[1])[:5])
        nl = sum(1 for v in lanes_on.values() if v > 0.05)
        P(f"  {w0 / 1e3:5.2f}-{w1 / 1e3:5.2f} {nl:5d} {sum(res.values()):5.2f} {tfs:8.1f} {100 * pm:5.1f}% {gbs:7.0f} {100 * ph:4.0f}%  {flag:5s}  {names}")
        if flag:
            low.append((w0, res, lanes_on, pm, ph))
    P(f"# {len(low)} of {nwin} windows under 40 % of both roofs")
    # what ran in the low windows, aggregated
    agg = {}
    for w0, res, lanes_on, pm, ph in low:
        for k, v in res.items():
            agg[k] = agg.get(k, 0.0) + v * W
    P("# kernel time inside the flagged windows (us), largest first:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]:
        P(f"#   {k:60s} {v:8.1f}")
    # mean per-phase flows: forward / backward boundary = the first data gradient or BatchNorm-backward launch on any lane
    if a.launches:
        P("# every launch: lane start_us end_us dur_us kernel")
        for ln, s, e, name, f, b in sorted(recs, key=lambda r: r[1]):
            P(f"L {ln} {s:9.1f} {e:9.1f} {e - s:8.1f} {short(name)}")
    text = "\n".join(out)
    print(text)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
