#!/usr/bin/env python3
"""Race screen: run each conv op of the CREMA-D net several times on identical inputs (garbage in the
outputs / workspaces between runs) and require bit-identical results."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gdl import _lib as L  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    dt = L.GDL_BF16
    td = torch.bfloat16
    dev = "cuda:0"
    st = L.cur_stream()
    B = int(os.environ.get("B", "64"))
    bad = 0
    for enc, mul, C, H, W, K, R, stride, pad, cnt in SHAPES:
        N = B * mul
        P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(N, H, W, C, device=dev).to(td)
        dy = torch.randn(N, P, Q, K, device=dev).to(td)
        wk = torch.randn(K, R, R, C, device=dev).to(td)
        wc = torch.randn(C, R, R, K, device=dev).to(td)
        tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
        nb = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
        tabs = []
        for mode in (0, 1):
            t = torch.empty(L.load().gdl_conv_table_bytes(mode, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
            L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
            tabs.append(t)
        res = {"fwd": [], "stats": [], "dgrad": [], "wgrad": []}
        for rep in range(4):
            y = torch.full((N, P, Q, K), float("nan"), device=dev, dtype=td)
            dx = torch.full((N, H, W, C), float("nan"), device=dev, dtype=td)
            dw = torch.full((K, C, R, R), float("nan"), device=dev)
            part = torch.full((tiles, K, 2), float("nan"), device=dev)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev).random_()
            L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tabs[0].data_ptr(), N, H, W,
                   C, K, R, R, stride, pad, st)
            L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, tabs[1].data_ptr(), N, H, W, C, K,
                   R, R, stride, pad, st)
            L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R,
                   stride, pad, ws.data_ptr(), nb, st)
            torch.cuda.synchronize()
            res["fwd"].append(y.view(torch.int16).clone())
            res["stats"].append(part.view(torch.int32).clone())
            res["dgrad"].append(dx.view(torch.int16).clone())
            res["wgrad"].append(dw.view(torch.int32).clone())
        if os.environ.get("DETAIL") and any(int((res["stats"][0] != t).sum()) for t in res["stats"][1:]):
            # where do the statistics differ?  rows = M-tiles, columns = (channel, sum | sum of squares)
            a = res["stats"][0].view(torch.float32)
            for i, t in enumerate(res["stats"][1:], 1):
                b = t.view(torch.float32)
                d = (res["stats"][0] != t).nonzero()
                if len(d) == 0:
                    continue
                tl, ch, w = d[:, 0], d[:, 1], d[:, 2]
                print(f"   run {i} vs 0: {len(d)} words differ of {a.numel()}; tiles {tl.min().item()}..{tl.max().item()} "
                      f"({len(tl.unique())} of {a.shape[0]}), channels {sorted(ch.unique().tolist())[:24]}, "
                      f"which {sorted(w.unique().tolist())}, nan a/b {int(torch.isnan(a).sum())}/{int(torch.isnan(b).sum())}")
                for j in range(min(6, len(d))):
                    x, y_ = a[tl[j], ch[j], w[j]].item(), b[tl[j], ch[j], w[j]].item()
                    print(f"      tile {tl[j].item()} ch {ch[j].item()} {'sum' if w[j] == 0 else 'sq '}: {x!r} vs {y_!r}  rel {abs(x - y_) / max(abs(x), 1e-30):.2e}")
        line = f"{enc} {C}x{H}x{W}->{K} {R}x{R}/{stride}:"
        for k, v in res.items():
            nd = sum(int((v[0] != t).sum()) for t in v[1:])
            nan = int(torch.isnan(res[k][0].view(td if k in ('fwd', 'dgrad') else torch.float32)).sum())
            line += f" {k}={'OK' if nd == 0 and nan == 0 else f'DIFF({nd}) NAN({nan})'}"
            bad += (nd != 0) + (nan != 0)
        print(line)
    print("RACES FOUND" if bad else "all deterministic")


if __name__ == "__main__":
    main()
