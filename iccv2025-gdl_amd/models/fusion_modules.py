"""Drop-in mirror of the concat fusion heads of /root/reference/models/fusion_modules.py.

`ConcatFusion_DGL` (:45-59) and `ConcatFusion` (:33-42): same constructor arguments, parameter
names (`fc_out`, and the never-used `fc_auxi` of the DGL head) and return order.
The three Linear calls, two zero fills and three cats of the reference collapse into one
gfx950 kernel per direction (csrc/head.hip).
"""
import torch
import torch.nn as nn

from gdl import _lib as L


def _f32c(t):
    return t.float().contiguous()


class _ConcatDGLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, W, b):
        x, y, W, b = _f32c(x), _f32c(y), _f32c(W), _f32c(b)
        B, n = x.shape[0], W.shape[0]
        if W.shape[1] != x.shape[1] + y.shape[1]:
            raise RuntimeError("gdl: ConcatFusion_DGL: fc_out must be as wide as the two feature vectors together")
        out, x_out, y_out = (torch.empty((B, n), device=x.device) for _ in range(3))
        ctx.xy = None if (x.shape[1] == 512 and y.shape[1] == 512) else (x.shape[1], y.shape[1])
        if ctx.xy is None:
            L.call("gdl_head_concat_fwd", L.ptr(x), L.ptr(y), L.ptr(W), L.ptr(b), L.ptr(out), L.ptr(x_out), L.ptr(y_out), B,
                   n, L.cur_stream())
        else:  # unequal widths (the Swin composition, 512 + 768)
            L.call("gdl_head_concat_xy_fwd", L.ptr(x), L.ptr(y), L.ptr(W), L.ptr(b), L.ptr(out), L.ptr(x_out), L.ptr(y_out), B,
                   n, ctx.xy[0], ctx.xy[1], L.cur_stream())
        ctx.save_for_backward(x, y, W)
        # undefined upstream gradients stay None (not zero tensors): the reference's second backward,
        # loss_f.backward() (main_dgl.py:122), reaches this node through `output` only, which was computed from
        # detached features -- returning None for dx / dy then keeps autograd from re-running both encoder backwards
        ctx.set_materialize_grads(False)
        return x_out, y_out, out

    @staticmethod
    def backward(ctx, g_x_out, g_y_out, g_out):
        x, y, W = ctx.saved_tensors
        B, n = x.shape[0], W.shape[0]
        gx = _f32c(g_x_out) if g_x_out is not None else None
        gy = _f32c(g_y_out) if g_y_out is not None else None
        go = _f32c(g_out) if g_out is not None else None
        if gx is None and gy is None and go is None:
            return None, None, None, None
        to_feat = gx is not None or gy is not None
        dx, dy = (torch.empty_like(x), torch.empty_like(y)) if to_feat else (None, None)
        dW, db = torch.empty_like(W), torch.empty(n, device=x.device)
        # `output` was computed from cat(x, y).detach() (fusion_modules.py:53-56): it never reaches x / y
        if ctx.xy is None:
            L.call("gdl_head_concat_bwd", L.ptr(x), L.ptr(y), L.ptr(W), L.ptr(gx), L.ptr(gy), L.ptr(go), 0, 1, L.ptr(dx),
                   L.ptr(dy), L.ptr(dW), L.ptr(db), B, n, L.cur_stream())
        else:
            L.call("gdl_head_concat_xy_bwd", L.ptr(x), L.ptr(y), L.ptr(W), L.ptr(gx), L.ptr(gy), L.ptr(go), 0, 1, L.ptr(dx),
                   L.ptr(dy), L.ptr(dW), L.ptr(db), B, n, ctx.xy[0], ctx.xy[1], L.cur_stream())
        return dx, dy, dW, db


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, W, b):
        x, y, W, b = _f32c(x), _f32c(y), _f32c(W), _f32c(b)
        B, n = x.shape[0], W.shape[0]
        out = torch.empty((B, n), device=x.device)
        L.call("gdl_head_concat_fwd", L.ptr(x), L.ptr(y), L.ptr(W), L.ptr(b), L.ptr(out), None, None, B, n,
               L.cur_stream())
        ctx.save_for_backward(x, y, W)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, y, W = ctx.saved_tensors
        B, n = x.shape[0], W.shape[0]
        go = _f32c(g_out)
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        dW, db = torch.empty_like(W), torch.empty(n, device=x.device)
        L.call("gdl_head_concat_bwd", L.ptr(x), L.ptr(y), L.ptr(W), None, None, L.ptr(go), 1, 0, L.ptr(dx), L.ptr(dy),
               L.ptr(dW), L.ptr(db), B, n, L.cur_stream())
        return dx, dy, dW, db


class _SumDGLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, Wx, bx, Wy, by):
        x, y, Wx, bx, Wy, by = (_f32c(t) for t in (x, y, Wx, bx, Wy, by))
        B, n = x.shape[0], Wx.shape[0]
        if x.shape[1] != 512 or y.shape[1] != 512 or Wx.shape[1] != 512 or Wy.shape[1] != 512:
            raise RuntimeError("gdl: SumFusion_DGL expects 512-d audio and visual features")
        out, x_out, y_out = (torch.empty((B, n), device=x.device) for _ in range(3))
        L.call("gdl_head_sum_fwd", L.ptr(x), L.ptr(y), L.ptr(Wx), L.ptr(bx), L.ptr(Wy), L.ptr(by), L.ptr(out), L.ptr(x_out),
               L.ptr(y_out), B, n, L.cur_stream())
        ctx.save_for_backward(x, y, Wx, Wy)
        ctx.set_materialize_grads(False)  # see _ConcatDGLFn
        return x_out, y_out, out

    @staticmethod
    def backward(ctx, g_x_out, g_y_out, g_out):
        x, y, Wx, Wy = ctx.saved_tensors
        B, n = x.shape[0], Wx.shape[0]
        gx = _f32c(g_x_out) if g_x_out is not None else None
        gy = _f32c(g_y_out) if g_y_out is not None else None
        go = _f32c(g_out) if g_out is not None else None
        if gx is None and gy is None and go is None:
            return None, None, None, None, None, None
        to_feat = gx is not None or gy is not None
        dx, dy = (torch.empty_like(x), torch.empty_like(y)) if to_feat else (None, None)
        dWx, dWy = torch.empty_like(Wx), torch.empty_like(Wy)
        dbx, dby = torch.empty(n, device=x.device), torch.empty(n, device=x.device)
        # `output` was computed from x.detach() / y.detach() (fusion_modules.py:27-29): it never reaches x / y
        L.call("gdl_head_sum_bwd", L.ptr(x), L.ptr(y), L.ptr(Wx), L.ptr(Wy), L.ptr(gx), L.ptr(gy), L.ptr(go), 0, 1, L.ptr(dx),
               L.ptr(dy), L.ptr(dWx), L.ptr(dbx), L.ptr(dWy), L.ptr(dby), B, n, L.cur_stream())
        return dx, dy, dWx, dbx, dWy, dby


class SumFusion_DGL(nn.Module):
    """fusion_modules.py:16-30: outx = fc_x(x), outy = fc_y(y), output = fc_x(x.detach()) + fc_y(y.detach())."""

    def __init__(self, input_dim=512, output_dim=100):
        super(SumFusion_DGL, self).__init__()
        self.fc_x = nn.Linear(input_dim, output_dim)
        self.fc_y = nn.Linear(input_dim, output_dim)

    def forward(self, x, y):
        outx, outy, output = _SumDGLFn.apply(x, y, self.fc_x.weight, self.fc_x.bias, self.fc_y.weight, self.fc_y.bias)
        return outx, outy, output


class _GatedDGLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, W1, b1, W2, b2, Wo, bo):
        x, y, W1, b1, W2, b2, Wo, bo = (_f32c(t) for t in (x, y, W1, b1, W2, b2, Wo, bo))
        B, n = x.shape[0], Wo.shape[0]
        if x.shape[1] != 512 or y.shape[1] != 512 or W1.shape != (512, 512) or W2.shape != (512, 512) or Wo.shape[1] != 512:
            raise RuntimeError("gdl: GatedFusion_DGL expects 512-d features and dim = 512")
        hx, hy = torch.empty((B, 512), device=x.device), torch.empty((B, 512), device=x.device)
        out, x_out, y_out = (torch.empty((B, n), device=x.device) for _ in range(3))
        L.call("gdl_head_gated_fwd", L.ptr(x), L.ptr(y), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(Wo), L.ptr(bo),
               L.ptr(hx), L.ptr(hy), L.ptr(out), L.ptr(x_out), L.ptr(y_out), B, n, L.cur_stream())
        ctx.save_for_backward(x, y, hx, hy, W1, W2, Wo)
        ctx.set_materialize_grads(False)  # see _ConcatDGLFn
        return x_out, y_out, out

    @staticmethod
    def backward(ctx, g_x_out, g_y_out, g_out):
        x, y, hx, hy, W1, W2, Wo = ctx.saved_tensors
        B, n = x.shape[0], Wo.shape[0]
        gx = _f32c(g_x_out) if g_x_out is not None else None
        gy = _f32c(g_y_out) if g_y_out is not None else None
        go = _f32c(g_out) if g_out is not None else None
        if gx is None and gy is None and go is None:
            return (None,) * 8
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        dW1, dW2, dWo = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(Wo)
        db1, db2, dbo = torch.empty(512, device=x.device), torch.empty(512, device=x.device), torch.empty(n, device=x.device)
        ws = torch.empty(2 * B * 512, device=x.device)
        L.call("gdl_head_gated_bwd", L.ptr(x), L.ptr(y), L.ptr(hx), L.ptr(hy), L.ptr(W1), L.ptr(W2), L.ptr(Wo), L.ptr(gx),
               L.ptr(gy), L.ptr(go), 1, L.ptr(dx), L.ptr(dy), L.ptr(dW1), L.ptr(db1), L.ptr(dW2), L.ptr(db2), L.ptr(dWo),
               L.ptr(dbo), L.ptr(ws), B, n, L.cur_stream())
        if gx is None and gy is None:
            # `output` alone (detached hidden vectors, fusion_modules.py:237-243) leaves fc_x / fc_y without a gradient:
            # autograd must see None, not zeros (SGD skips grad-less parameters, weight decay included)
            return None, None, None, None, None, None, dWo, dbo
        return dx, dy, dW1, db1, dW2, db2, dWo, dbo


class GatedFusion_DGL(nn.Module):
    """fusion_modules.py:213-250 (x_gate=True, the only setting basic_model.py:38 constructs)."""

    def __init__(self, input_dim=512, dim=512, output_dim=100, x_gate=True):
        super(GatedFusion_DGL, self).__init__()
        if not x_gate:
            raise NotImplementedError("gdl: GatedFusion_DGL is built for x_gate=True (basic_model.py:38)")
        self.fc_x = nn.Linear(input_dim, dim)
        self.fc_y = nn.Linear(input_dim, dim)
        self.fc_out = nn.Linear(dim, output_dim)
        self.x_gate = x_gate
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, y):
        out_x, out_y, output = _GatedDGLFn.apply(x, y, self.fc_x.weight, self.fc_x.bias, self.fc_y.weight, self.fc_y.bias,
                                                 self.fc_out.weight, self.fc_out.bias)
        return out_x, out_y, output


class _FiLMDGLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, Wfc, bfc, Wo, bo):
        x, y, Wfc, bfc, Wo, bo = (_f32c(t) for t in (x, y, Wfc, bfc, Wo, bo))
        B, n = x.shape[0], Wo.shape[0]
        if x.shape[1] != 512 or y.shape[1] != 512 or Wfc.shape != (512, 512 * 512) or Wo.shape[1] != 512:
            raise RuntimeError("gdl: FiLM_DGL expects 512-d features and dim = 512")
        if B > 512:  # (gdl_head_film_workspace_bytes returns 0 beyond; csrc/head_film.hip walks sample groups of 64)
            raise RuntimeError("gdl: FiLM_DGL handles at most 512 samples per call")
        nb = L.load().gdl_head_film_workspace_bytes(B)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)  # keeps W_k v_b for the backward
        hidden = torch.empty((3, B, 512), device=x.device)
        out, x_out, y_out = (torch.empty((B, n), device=x.device) for _ in range(3))
        L.call("gdl_head_film_fwd", L.ptr(x), L.ptr(y), L.ptr(Wfc), L.ptr(bfc), L.ptr(Wo), L.ptr(bo), L.ptr(hidden), L.ptr(out),
               L.ptr(x_out), L.ptr(y_out), B, n, L.ptr(ws), nb, L.cur_stream())
        ctx.save_for_backward(x, y, Wfc, Wo, hidden, ws)
        ctx.set_materialize_grads(False)  # see _ConcatDGLFn
        return x_out, y_out, out

    @staticmethod
    def backward(ctx, g_x_out, g_y_out, g_out):
        x, y, Wfc, Wo, hidden, ws = ctx.saved_tensors
        B, n = x.shape[0], Wo.shape[0]
        gx = _f32c(g_x_out) if g_x_out is not None else None
        gy = _f32c(g_y_out) if g_y_out is not None else None
        go = _f32c(g_out) if g_out is not None else None
        if gx is None and gy is None and go is None:
            return (None,) * 6
        uni_only_out = gx is None and gy is None  # `output` alone never reaches x / y (detached, fusion_modules.py:150-158)
        dx = None if uni_only_out else torch.empty_like(x)
        dy = None if uni_only_out else torch.empty_like(y)
        dWfc, dbfc = torch.empty_like(Wfc), torch.empty(512, device=x.device)
        dWo, dbo = torch.empty_like(Wo), torch.empty(n, device=x.device)
        L.call("gdl_head_film_bwd", L.ptr(x), L.ptr(y), L.ptr(Wfc), L.ptr(Wo), L.ptr(hidden), L.ptr(gx), L.ptr(gy), L.ptr(go), 1,
               L.ptr(dx), L.ptr(dy), L.ptr(dWfc), L.ptr(dbfc), L.ptr(dWo), L.ptr(dbo), B, n, L.ptr(ws), ws.numel(),
               L.cur_stream())
        return dx, dy, dWfc, dbfc, dWo, dbo


class FiLM_DGL(nn.Module):
    """fusion_modules.py:126-178: fc = Linear(dim*dim, dim) applied to flattened outer products (a bilinear form per
    output), fc_out = Linear(dim, n).  134 M parameters.  `x_film` is accepted and, as in the reference, unused."""

    def __init__(self, input_dim=512, dim=512, output_dim=100, x_film=True):
        super(FiLM_DGL, self).__init__()
        if input_dim != 512 or dim != 512:
            raise NotImplementedError("gdl: FiLM_DGL is built for input_dim = dim = 512 (basic_model.py:36)")
        self.fc = nn.Linear(dim * dim, dim)
        self.fc_out = nn.Linear(dim, output_dim)
        self.x_film = x_film

    def forward(self, x, y):
        z_x, z_y, output = _FiLMDGLFn.apply(x, y, self.fc.weight, self.fc.bias, self.fc_out.weight, self.fc_out.bias)
        return z_x, z_y, output


class ConcatFusion(nn.Module):
    def __init__(self, input_dim=1024, output_dim=100):
        super(ConcatFusion, self).__init__()
        self.fc_out = nn.Linear(input_dim, output_dim)

    def forward(self, x, y):
        output = _ConcatFn.apply(x, y, self.fc_out.weight, self.fc_out.bias)
        return x, y, output


class ConcatFusion_DGL(nn.Module):
    def __init__(self, input_dim=512 * 2, output_dim=100):
        super(ConcatFusion_DGL, self).__init__()
        self.fc_out = nn.Linear(input_dim, output_dim)
        self.fc_auxi = nn.Linear(input_dim, output_dim)  # registered but unused, as in the reference

    def forward(self, x, y):
        x_out, y_out, output = _ConcatDGLFn.apply(x, y, self.fc_out.weight, self.fc_out.bias)
        return x_out, y_out, output
