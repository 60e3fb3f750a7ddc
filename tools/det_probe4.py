import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L
dt = L.GDL_BF16; td = torch.bfloat16; dev = "cuda:0"; st = L.cur_stream()
N, C, H, W, K, R, stride, pad = 64, 64, 65, 47, 128, 3, 2, 1
P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
x = torch.randn(N, H, W, C, device=dev).to(td); wk = torch.randn(K, R, R, C, device=dev).to(td)
dy = torch.randn(N, P, Q, K, device=dev).to(td); wc = torch.randn(C, R, R, K, device=dev).to(td)
tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
tabs = []
for mode in (0, 1):
    t = torch.empty(L.load().gdl_conv_table_bytes(mode, N, H, W, R, R, stride, pad), dtype=torch.uint8, device=dev)
    L.call("gdl_conv_build_table", mode, dt, N, H, W, C, K, R, R, stride, pad, t.data_ptr(), st)
    tabs.append(t)
nb = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
BM = 256
for rep in range(4):
    y = torch.full((N, P, Q, K), float("nan"), device=dev, dtype=td)
    dx = torch.full((N, H, W, C), float("nan"), device=dev, dtype=td)
    dw = torch.full((K, C, R, R), float("nan"), device=dev)
    part = torch.full((tiles, K, 2), float("nan"), device=dev)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev).random_()
    L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R, stride, pad, st)
    L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, tabs[1].data_ptr(), N, H, W, C, K, R, R, stride, pad, st)
    L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R, stride, pad, ws.data_ptr(), nb, st)
    torch.cuda.synchronize()
    yf = y.double().view(-1, K)
    M = yf.shape[0]
    pad_rows = tiles * BM - M
    yp = torch.cat([yf, torch.zeros(pad_rows, K, device=dev, dtype=torch.float64)]).view(tiles, BM, K)
    ref = torch.stack([yp.sum(1), (yp * yp).sum(1)], -1)
    err = (part.double() - ref).abs() / (ref.abs() + 1.0)
    bad = (err > 1e-3).nonzero()
    print('rep', rep, 'bad entries', bad.shape[0], 'of', part.numel(), ' w-values:', sorted(set(bad[:, 2].tolist())), 'chan mod 8:', sorted(set((bad[:, 1] % 8).tolist())))
    for idx in bad[:3].tolist():
        ti, c, w = idx
        resid = (part[ti, c, w].double() - ref[ti, c, w]).item()
        col = yp[ti, :, c] if w == 0 else yp[ti, :, c] ** 2
        # which residue class of rows (row % 32) explains the residual?
        best = None
        for r in range(32):
            s = col[r::32].sum().item()
            for sign in (+1, -1):
                d = abs(resid - sign * s)
                if best is None or d < best[0]:
                    best = (d, r, sign, s)
        # or a wave's rows: rows with (row%32)//8 == wave
        wbest = None
        for wv in range(4):
            rows = [r for r in range(BM) if (r % 32) // 8 == wv]
            s = col[rows].sum().item()
            for sign in (+1, -1):
                d = abs(resid - sign * s)
                if wbest is None or d < wbest[0]:
                    wbest = (d, wv, sign, s)
        print('   tile', ti, 'ch', c, 'w', w, 'got', part[ti, c, w].item(), 'ref', ref[ti, c, w].item(), 'resid', resid, '| best thread-rows', best, '| best wave', wbest)
