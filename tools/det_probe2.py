import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))
from gdl import _lib as L
dt = L.GDL_BF16; td = torch.bfloat16; dev = "cuda:0"; st = L.cur_stream()
N, C, H, W, K, R, stride, pad = 64, 64, 65, 47, 128, 1, 2, 0
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
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
dx = torch.empty(N, H, W, C, device=dev, dtype=td); dw = torch.empty(K, C, R, R, device=dev)
for seq in ("fwd", "fwd+dgrad", "fwd+wgrad", "fwd+fill"):
    outs = []
    for rep in range(6):
        y = torch.full((N, P, Q, K), float("nan"), device=dev, dtype=td)
        part = torch.full((tiles, K, 2), float("nan"), device=dev)
        L.call("gdl_conv_fwd", dt, x.data_ptr(), wk.data_ptr(), y.data_ptr(), part.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R, stride, pad, st)
        if "dgrad" in seq:
            L.call("gdl_conv_dgrad", dt, dy.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, tabs[1].data_ptr(), N, H, W, C, K, R, R, stride, pad, st)
        if "wgrad" in seq:
            L.call("gdl_conv_wgrad", dt, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tabs[0].data_ptr(), N, H, W, C, K, R, R, stride, pad, ws.data_ptr(), nb, st)
        if "fill" in seq:
            ws.random_()
        torch.cuda.synchronize()
        outs.append((part.clone(), y.clone()))
    nd = [int((outs[r][0].view(torch.int32) != outs[0][0].view(torch.int32)).sum()) for r in range(1, 6)]
    ny = [int((outs[r][1].view(torch.int16) != outs[0][1].view(torch.int16)).sum()) for r in range(1, 6)]
    print(seq, 'stats diffs', nd, 'y diffs', ny)
    if sum(nd):
        r = max(range(1, 6), key=lambda r: nd[r - 1])
        ne = (outs[r][0].view(torch.int32) != outs[0][0].view(torch.int32)).nonzero()
        for idx in ne[:6].tolist():
            ti, c, w = idx
            print('    tile', ti, 'ch', c, 'w', w, outs[0][0][ti, c, w].item(), outs[r][0][ti, c, w].item())
        print('    tiles:', sorted(set(ne[:, 0].tolist()))[:40])
        print('    chans:', sorted(set(ne[:, 1].tolist()))[:70])
