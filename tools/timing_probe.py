#!/usr/bin/env python3
"""Per-block s_memtime stamps of one conv launch (needs a -DGDL_TIMING build of libgdl_hip.so).

usage: python tools/timing_probe.py --op fwd --shape N,C,H,W,K,R,stride,pad
Prints, per segment between consecutive stamps, the mean / p10 / p90 cycles over blocks, the block
lifetime, and how many blocks were resident per CU on average.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="192,64,56,56,64,3,1,1")
    ap.add_argument("--op", default="fwd")
    ap.add_argument("--names", default="entry,issued,landed,kloop_end,kloop_end2,end",
                    help="conv3x3_pslab_kernel also leaves stamps 6 / 7 (behind the first tile's epilogue rounds / behind its fold): "
                         "--names entry,issued,landed,kloop_end,kloop_end2,end,rounds,fold prints them relative to kloop_end2")
    ap.add_argument("--c64", action="store_true", help="the 64-channel persistent kernel's per-tile phases (use --names entry,loop,loop_end,end)")
    a = ap.parse_args()
    N, C, H, W, K, R, stride, pad = [int(v) for v in a.shape.split(",")]
    lib = L.load()
    dt = L.dtype_code("bf16")
    td = torch.bfloat16
    dev = "cuda:0"
    st = L.cur_stream()
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, W, C, device=dev).to(td)
    dy = torch.randn(N, P, Q, K, device=dev).to(td)
    wk = torch.randn(K, R, R, C, device=dev).to(td)
    wc = torch.randn(C, R, R, K, device=dev).to(td)
    y = torch.empty(N, P, Q, K, device=dev, dtype=td)
    dx = torch.empty(N, H, W, C, device=dev, dtype=td)
    dw = torch.empty(K, C, R, R, device=dev)
    tiles = lib.gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    part = torch.empty(tiles, K, 2, device=dev)
    nb = lib.gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    tabs = []
    for mode in (0, 1):
        t = torch.empty(lib.gdl_conv_table_bytes(mode, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
        L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
        tabs.append(t)
    nblk = 1 << 16
    dbg = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)

    def run():
        if a.op == "fwd":
            L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tabs[0].data_ptr(), N, H, W,
                   C, K, R, R, stride, pad, st)
        elif a.op == "dgrad":
            L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, tabs[1].data_ptr(), N, H, W, C, K,
                   R, R, stride, pad, st)
        else:
            L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R,
                   stride, pad, ws.data_ptr(), nb, st)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.call("gdl_debug_timing_buffer", dbg.data_ptr())
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    L.call("gdl_debug_timing_buffer", None)
    d = dbg.cpu().numpy().astype(np.int64)
    names = a.names.split(",")
    ns = len(names)
    full = d
    d = d[: 1 << 15]
    live = d[:, 0] != 0
    d = d[live]
    print(f"{a.op} {a.shape}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(d)} blocks stamped")
    # s_memtime counters are per XCD (block b runs on XCD b % 8) and not synchronised with each other: spans per XCD
    bid = np.nonzero(live)[0]
    last = 5 if ns == 8 else ns - 1
    life = d[:, last] - d[:, 0]
    spans, resid = [], []
    for x in range(8):
        sel = (bid & 7) == x
        if not sel.any():
            continue
        sp = d[sel, last].max() - d[sel, 0].min()
        spans.append(sp)
        resid.append(life[sel].sum() / sp)
    span = float(np.mean(spans))
    wall_us = e0.elapsed_time(e1) * 1e3
    print(f"kernel span per XCD: mean {span:.0f} ticks (min {min(spans)}, max {max(spans)}) for {wall_us:.1f} us of wall time "
          f"-> {span / wall_us / 1e3:.2f} ticks/ns; blocks resident per XCD: {np.mean(resid):.1f} (= {np.mean(resid) / 32:.2f} per CU)")
    t0 = d[:, 0].min()
    print(f"block life: mean {life.mean():.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f}")
    for i in range(last):
        seg = d[:, i + 1] - d[:, i]
        print(f"  {names[i]:>10s} -> {names[i + 1]:<10s} mean {seg.mean():8.0f}  p10 {np.percentile(seg, 10):8.0f}  p90 "
              f"{np.percentile(seg, 90):8.0f}")
    if ns == 8:  # (stamps 6 / 7 were taken between stamps 4 and 5)
        for nm, i in (("kloop_end2 -> rounds done", 6), ("rounds done -> fold done", 7)):
            seg = d[:, i] - d[:, 4 if i == 6 else 6]
            print(f"  {nm:>28s} mean {seg.mean():8.0f}  p10 {np.percentile(seg, 10):8.0f}  p90 {np.percentile(seg, 90):8.0f}")
    seg = full[(1 << 15):][live]
    if a.c64:  # conv3x3_c64_kernel: per-tile phase times of wave 0
        nt = seg[:, 7].sum()
        print(f"  tiles per block: mean {seg[:, 7].mean():.1f} max {seg[:, 7].max()}")
        for i, nm in enumerate(("slab DMA issue", "K-loop", "barrier A", "staging + barrier B", "wait vmcnt(0)", "row loads / stores / stats",
                                "barrier C")):
            print(f"  per tile {nm:>28s}: {seg[:, i].sum() / max(nt, 1):7.0f}")
    elif seg[:, 5].max() > 0:
        nk = seg[:, 5].mean()
        names5 = ("wait+barrier", "dma issue", "reads->1st data", "mfma0+2nd data", "mfma1")
        if (seg[:, 6] == 0x5053).any():  # conv3x3_pslab_kernel (round 6): stamps consumed a step later, nothing waits for them
            names5 = ("wait for F0", "[A] 24 MFMA+reads", "wait + barrier", "[C] 24 MFMA+DMA+rd", "tail (slab, loop)")
        for i, nm in enumerate(names5):
            print(f"  per K-step {nm:>20s}: {seg[:, i].mean() / nk:7.0f}")
    start = d[:, 0] - t0
    print("start-time quantiles:", [int(np.percentile(start, q)) for q in (0, 25, 50, 75, 100)])


if __name__ == "__main__":
    main()
