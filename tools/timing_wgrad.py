#!/usr/bin/env python3
"""wgrad per-block wait / issue / compute cycle totals (needs the -DGDL_TIMING build, GDL_LIB=...)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402

lib = L.load()
dt = L.dtype_code("bf16")
dev = "cuda:0"
st = L.cur_stream()
for shape in sys.argv[1:]:
    N, C, H, W, K, R, stride, pad = [int(v) for v in shape.split(",")]
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    dy = torch.randn(N, P, Q, K, device=dev).bfloat16()
    dw = torch.empty(K, C, R, R, device=dev)
    nb = lib.gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    t = torch.empty(lib.gdl_conv_table_bytes(0, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
    L.call("gdl_conv_build_table", 0, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
    dbg = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)

    def run():
        L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), t.data_ptr(), N, H, W, C, K, R, R, stride,
               pad, ws.data_ptr(), nb, st)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.call("gdl_debug_timing_buffer", dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    L.call("gdl_debug_timing_buffer", None)
    dall = dbg.cpu().numpy()
    # the 9-tap kernel leaves one record per wave ([block][8 waves]: 4 workers, 4 loaders), the per-tap one per block
    groups = [("all", dall)]
    if R == 3 and stride == 1 and os.environ.get("GDL_WGRAD9", "1") != "0":
        groups = [("worker waves", dall.reshape(-1, 8, 8)[:, :4].reshape(-1, 8)), ("loader waves", dall.reshape(-1, 8, 8)[:, 4:].reshape(-1, 8))]
    print(f"wgrad {shape}: {e0.elapsed_time(e1) * 1e3:.1f} us (kernel+reduce)")
    for gname, d in groups:
        d = d[d[:, 0] != 0]
        if len(d) == 0:  # (the four-wave form has no loader waves)
            continue
        nst = d[:, 6].mean()
        span = float(d[:, 0].max() - d[:, 0].min())  # s_memtime ticks = shader cycles
        print(f"  {gname}: {len(d)} records, {nst:.1f} stages each, entry spread {span:.0f} clk")
        for i, nm in ((1, "prologue"), (2, "wait+barrier"), (3, "issue"), (7, "masks"), (4, "compute"), (5, "epilogue")):
            v = d[:, i]
            per = f" ({v.mean() / nst:7.1f} / stage)" if i in (2, 3, 4, 7) else ""
            print(f"   {nm:>13s}: mean {v.mean():9.0f} p10 {np.percentile(v, 10):9.0f} p90 {np.percentile(v, 90):9.0f}{per}")
