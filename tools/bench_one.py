#!/usr/bin/env python3
"""Run ONE conv shape (fwd / dgrad / wgrad) a few times -- target for rocprofv3 --pmc runs."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="192,64,56,56,64,3,1,1", help="N,C,H,W,K,R,stride,pad")
    ap.add_argument("--op", default="fwd")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    N, C, H, W, K, R, stride, pad = [int(v) for v in a.shape.split(",")]
    dt = L.dtype_code(a.dtype)
    td = L.torch_dtype(dt)
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
    tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    part = torch.empty(tiles, K, 2, device=dev)
    nb = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    tabs = []
    for mode in (0, 1):
        t = torch.empty(L.load().gdl_conv_table_bytes(mode, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
        L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
        tabs.append(t)
    tf, td_ = tabs[0].data_ptr(), tabs[1].data_ptr()
    for _ in range(a.iters):
        if a.op == "fwd":
            L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tf, N, H, W, C, K, R, R,
                   stride, pad, st)
        elif a.op == "dgrad":
            L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, td_, N, H, W, C, K, R, R, stride,
                   pad, st)
        else:
            L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tf, N, H, W, C, K, R, R, stride, pad,
                   ws.data_ptr(), nb, st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
