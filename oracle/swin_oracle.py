"""TEST INFRASTRUCTURE -- CPU restatement of the Swin visual encoder ("next" row N4 of SURVEY 8(f)).

Written from scratch on torch tensor arithmetic (none of the reference's files): the same function the reference's
`SwinTransformer.forward` computes with `args.pe = 0`, `ape = False`, `patch_norm = True`, all dropouts and the stochastic
depth at 0 -- /root/reference/models/swin_transformer.py:596-634 (model), :256-295 (block), :124-157 (window attention),
:330-353 (patch merging), :478-486 (patch embedding).  Gradients come from torch autograd over this restatement.
Only tests/ (and, for the Swin workload, bench.py's cpu_baseline leg) may import it; it is pinned against the golden
vectors `tests/golden/swin_*.npz` captured from the imported reference (tests/golden/make_golden.py::run_swin_case).

The formulation differs from the reference's on purpose, so that it also pins the index arithmetic the HIP kernels use:
windows are NOT materialised by roll / view / permute -- every (shifted) window is a list of token indices into the
un-rolled [H*W] token grid (`window_tokens`), the attention mask comes from region ids of the shifted coordinates
(`window_regions`), and the attention output is scattered back through the same index list.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def window_tokens(H, W, ws, shift):
    """[nW, ws*ws] int64: token index (h*W + w, un-rolled grid) of slot (i, j) of every window of the grid rolled by
    -shift (swin_transformer.py:266-276: slot (r, c) of the rolled grid holds token ((r+shift)%H, (c+shift)%W))."""
    r = (np.arange(H) + shift) % H
    c = (np.arange(W) + shift) % W
    grid = r[:, None] * W + c[None, :]  # rolled position -> original token
    g = grid.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    return torch.from_numpy(np.ascontiguousarray(g)).long()


def window_regions(H, W, ws, shift):
    """[nW, ws*ws] region id of every window slot in ROLLED coordinates (swin_transformer.py:222-240): 3 x 3 regions cut
    at H-ws and H-shift; tokens of different regions inside one window must not attend to each other (-100)."""
    def reg(n):
        a = np.zeros(n, np.int64)
        a[n - ws:n - shift] = 1
        a[n - shift:] = 2
        return a

    g = reg(H)[:, None] * 3 + reg(W)[None, :]
    g = g.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    return torch.from_numpy(np.ascontiguousarray(g)).long()


def relative_index(ws):
    """[ws*ws, ws*ws] index into the (2ws-1)^2 bias table (swin_transformer.py:103-113)."""
    i = np.arange(ws * ws)
    h, w = i // ws, i % ws
    dh = h[:, None] - h[None, :] + ws - 1
    dw = w[:, None] - w[None, :] + ws - 1
    return torch.from_numpy(dh * (2 * ws - 1) + dw).long()


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def block(x, P, pre, H, W, nh, ws, shift, drop=None):
    """x [N, H*W, C] -> same (swin_transformer.py:256-295).  drop: None (DropPath is the identity: eval mode or rate 0) or two
    [N] tensors of per-frame scales (0 or 1 / keep_prob) for the attention and the Mlp branch -- what timm's
    `drop_path(x, p, training, scale_by_keep=True)` multiplies the branch with (:290, :293)."""
    N, L, C = x.shape
    hd = C // nh
    idx = window_tokens(H, W, ws, shift)  # [nW, T]
    nW, T = idx.shape
    h = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    hw = h[:, idx.reshape(-1)].reshape(N, nW, T, C)
    qkv = hw @ P[pre + "attn.qkv.weight"].t() + P[pre + "attn.qkv.bias"]
    qkv = qkv.reshape(N, nW, T, 3, nh, hd)
    q, k, v = (qkv[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))  # [N, nW, nh, T, hd]
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = P[pre + "attn.relative_position_bias_table"][relative_index(ws).reshape(-1)].reshape(T, T, nh).permute(2, 0, 1)
    s = s + bias
    if shift > 0:
        reg = window_regions(H, W, ws, shift)
        mask = torch.where(reg[:, :, None] != reg[:, None, :], torch.tensor(-100.0), torch.tensor(0.0))  # [nW, T, T]
        s = s + mask[None, :, None]
    o = torch.softmax(s, dim=-1) @ v  # [N, nW, nh, T, hd]
    o = o.permute(0, 1, 3, 2, 4).reshape(N, nW * T, C)
    o = o @ P[pre + "attn.proj.weight"].t() + P[pre + "attn.proj.bias"]
    back = torch.zeros_like(x).index_add(1, idx.reshape(-1), o)  # idx is a permutation of the tokens
    x = x + (back if drop is None else back * drop[0][:, None, None])
    m = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
    m = gelu(m @ P[pre + "mlp.fc1.weight"].t() + P[pre + "mlp.fc1.bias"])
    m = m @ P[pre + "mlp.fc2.weight"].t() + P[pre + "mlp.fc2.bias"]
    return x + (m if drop is None else m * drop[1][:, None, None])


def merge(x, P, pre, H, W):
    """[N, H*W, C] -> [N, H/2*W/2, 2C] (swin_transformer.py:330-353: channel blocks (even h, even w), (odd h, even w),
    (even h, odd w), (odd h, odd w))."""
    N, L, C = x.shape
    g = x.reshape(N, H // 2, 2, W // 2, 2, C)
    cat = torch.cat([g[:, :, 0, :, 0], g[:, :, 1, :, 0], g[:, :, 0, :, 1], g[:, :, 1, :, 1]], -1).reshape(N, L // 4, 4 * C)
    cat = layer_norm(cat, P[pre + "norm.weight"], P[pre + "norm.bias"])
    return cat @ P[pre + "reduction.weight"].t()


def forward(x, P, cfg, drop=None):
    """x [B, 3, T, img, img] -> pooled features [B*T, C_last]   (swin_transformer.py:596-634, args.pe = 0).
    drop: None or a [blocks][2][B*T] tensor of DropPath scales, blocks in network order (see `block`)."""
    B, Cin, T, Hi, Wi = x.shape
    p, E = cfg["patch"], cfg["embed"]
    x = x.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, Hi, Wi)
    N, H, W = B * T, Hi // p, Wi // p
    pat = x.reshape(N, Cin, H, p, W, p).permute(0, 2, 4, 1, 3, 5).reshape(N, H * W, Cin * p * p)
    t = pat @ P["patch_embed.proj.weight"].reshape(E, -1).t() + P["patch_embed.proj.bias"]
    t = layer_norm(t, P["patch_embed.norm.weight"], P["patch_embed.norm.bias"])
    nl = len(cfg["depths"])
    kb = 0
    for i, (depth, nh) in enumerate(zip(cfg["depths"], cfg["heads"])):
        ws = min(cfg["window"], H)
        for j in range(depth):
            shift = 0 if (j % 2 == 0 or H <= cfg["window"]) else cfg["window"] // 2
            t = block(t, P, f"layers.{i}.blocks.{j}.", H, W, nh, ws, shift, None if drop is None else drop[kb])
            kb += 1
        if i < nl - 1:
            t = merge(t, P, f"layers.{i}.downsample.", H, W)
            H, W = H // 2, W // 2
    t = layer_norm(t, P["norm.weight"], P["norm.bias"])
    return t.mean(1)


def drop_path_scales(cfg, rate, n_frames, generator=None):
    """[blocks][2][n_frames] float32 DropPath scales as the reference's training forward draws them: block k of sum(depths) drops
    with probability linspace(0, rate, sum(depths))[k] (swin_transformer.py:546), a fresh Bernoulli(keep) / keep per branch and
    frame, attention branch first; a block with probability 0 is nn.Identity (:218) and draws nothing."""
    nb = sum(cfg["depths"])
    dpr = torch.linspace(0, rate, nb).tolist()
    out = torch.ones((nb, 2, n_frames), dtype=torch.float32)
    for k, p in enumerate(dpr):
        if p > 0:
            for br in range(2):
                out[k, br] = torch.empty(n_frames).bernoulli_(1.0 - p, generator=generator) / (1.0 - p)
    return out


def forward_backward(x, params, cfg, dy, drop=None):
    """numpy in / numpy out: (y, {name: grad}) for the loss sum(y * dy)."""
    P = {k: torch.from_numpy(np.array(v)).clone().requires_grad_(True) for k, v in params.items()}
    y = forward(torch.from_numpy(np.asarray(x)), P, cfg, None if drop is None else torch.from_numpy(np.asarray(drop, dtype=np.float32)))
    (y * torch.from_numpy(np.asarray(dy))).sum().backward()
    return y.detach().numpy(), {k: v.grad.numpy() for k, v in P.items()}
