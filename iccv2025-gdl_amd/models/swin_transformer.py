"""Drop-in mirror of /root/reference/models/swin_transformer.py (`SwinTransformer`, :486-674) on libgdl_hip.

Same constructor signature, the same sub-module tree and therefore the same `state_dict()` keys (patch_embed.proj /
.norm, layers.i.blocks.j.{norm1, attn.{relative_position_bias_table, relative_position_index, qkv, proj}, norm2,
mlp.{fc1, fc2}}, attn_mask of the shifted blocks, layers.i.downsample.{reduction, norm}, norm), the same initialisation
(:568-576: trunc_normal(.02) Linear weights and bias tables, zero biases, unit LayerNorms; PyTorch's default for the patch
convolution) and the same forward contract: `forward(x [B, 3, T, H, W]) -> [B*T, num_features]` pooled features.  The
`torch.nn` layers are parameter containers; the arithmetic is one autograd node over `gdl.swin.SwinEngine`.

Supported: `args.pe = 0` (the DUL branch of :577-586 is not built), `ape = False`, `patch_norm = True`, dropouts 0, head
dimension 32 (every published Swin size); anything else raises.  Stochastic depth (`drop_path_rate`, default 0.1 like the
reference's): the identity in eval mode; a training forward draws the per-frame masks with torch's generator on the input's
device, in the reference's order (`drop_path_scales`), and the engine applies them (`gdl_swin_drop_path`).
"""
import numpy as np
import torch
import torch.nn as nn

from gdl import _lib as L
from gdl.swin import SwinEngine


def _rel_index(ws):
    i = np.arange(ws * ws)
    h, w = i // ws, i % ws
    return torch.from_numpy((h[:, None] - h[None, :] + ws - 1) * (2 * ws - 1) + (w[:, None] - w[None, :] + ws - 1)).long()


def _shift_mask(H, W, ws, shift):
    def reg(n):
        a = np.zeros(n, np.int64)
        a[n - ws:n - shift] = 1
        a[n - shift:] = 2
        return a

    g = (reg(H)[:, None] * 3 + reg(W)[None, :]).reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    return torch.from_numpy(np.where(g[:, None, :] != g[:, :, None], -100.0, 0.0).astype(np.float32))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(0.)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (window_size, window_size), num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _rel_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(0.)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio, qkv_bias):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer("attn_mask", _shift_mask(*input_resolution, self.window_size, self.shift_size)
                             if self.shift_size > 0 else None)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, downsample):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias) for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)


def drop_path_scales(cfg, rate, n_frames, device, generator=None):
    """float32 [blocks][2][n_frames]: what timm's `drop_path(x, p, training=True, scale_by_keep=True)` multiplies the attention /
    Mlp branch of every block with in a training forward (swin_transformer.py:218, 290, 293): block k of sum(depths) drops a frame
    with probability linspace(0, rate, sum(depths))[k] (:546) -- Bernoulli(keep) / keep per frame, a fresh draw per branch,
    attention branch first, blocks in network order; a block of probability 0 is nn.Identity and draws nothing."""
    nb = sum(cfg["depths"])
    out = torch.ones((nb, 2, n_frames), dtype=torch.float32, device=device)
    for k, p in enumerate(torch.linspace(0, rate, nb).tolist()):
        if p > 0:
            for br in range(2):
                m = torch.empty(n_frames, dtype=torch.float32, device=device).bernoulli_(1.0 - p, generator=generator)
                out[k, br] = m / (1.0 - p) if p < 1.0 else m
    return out


class _SwinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, pooled, x, *params):
        eng = net._engine(x)
        eng.set_params([p.detach() for p in params])
        drop = None
        if net.training and net.drop_path_rate > 0:
            drop = net.drop_scales_override if net.drop_scales_override is not None else \
                drop_path_scales(net.cfg, net.drop_path_rate, x.shape[0] * x.shape[2], x.device)
            net.last_drop_scales = drop
        feat = eng.forward(x.float().contiguous(), pool_frames=pooled, drop_scales=drop).clone()
        ctx.net, ctx.eng, ctx.n = net, eng, len(params)
        ctx.serial = eng.serial  # the engine is shared by every forward of this shape (train and eval): see backward
        ctx.set_materialize_grads(False)
        return feat

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None) + (None,) * ctx.n
        grads = [torch.empty_like(p) for p in ctx.eng._params]
        # raises if another forward went through the engine in between (as _ResNetFn does for the ResNet18 engine)
        ctx.eng.backward(g.float().contiguous(), grads, serial=ctx.serial)
        return (None, None, None) + tuple(grads)


class SwinTransformer(nn.Module):
    def __init__(self, args, modality, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=128,
                 depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, fused_window_process=False, **kwargs):
        super().__init__()
        if getattr(args, "pe", 0):
            raise NotImplementedError("gdl: SwinTransformer with args.pe (the DUL branch) is not built")
        if ape or not patch_norm or qk_scale is not None or drop_rate or attn_drop_rate or in_chans != 3 or \
                norm_layer is not nn.LayerNorm or not qkv_bias or modality != 'visual':
            raise NotImplementedError("gdl: SwinTransformer supports modality='visual', ape=False, patch_norm=True, qkv_bias=True, "
                                      "in_chans=3 and drop_rate = attn_drop_rate = 0")
        # Stochastic depth (the reference constructor's default is 0.1, swin_transformer.py:516): the identity in eval mode; a
        # training forward draws per-frame masks (drop_path_scales).  `drop_scales_override`: a [blocks][2][frames] tensor used
        # instead of a fresh draw (tests, replaying a recorded run); `last_drop_scales`: what the last training forward used.
        self.drop_path_rate = float(drop_path_rate)
        if not 0.0 <= self.drop_path_rate < 1.0:
            raise ValueError("gdl: drop_path_rate must be in [0, 1)")
        self.drop_scales_override, self.last_drop_scales = None, None
        if any(embed_dim * 2 ** i != 32 * h for i, h in enumerate(num_heads)) or mlp_ratio != int(mlp_ratio):
            raise NotImplementedError("gdl: SwinTransformer needs head dimension 32 and an integer mlp_ratio")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.ape, self.patch_norm, self.mlp_ratio, self.modality = ape, patch_norm, mlp_ratio, modality
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=0.)
        res = self.patches_resolution[0]
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), (res // 2 ** i, res // 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                       qkv_bias, i < self.num_layers - 1) for i in range(self.num_layers)])
        self.norm = nn.LayerNorm(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.apply(self._init_weights)
        self.args = args
        self.cfg = dict(img=img_size, patch=patch_size, embed=embed_dim, depths=tuple(depths), heads=tuple(num_heads),
                        window=window_size, mlp=int(mlp_ratio))
        self.gdl_dtype = None  # None: GDL_DTYPE / bf16
        self._eng = {}

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def _engine(self, x):
        import os

        if not x.is_cuda:
            raise RuntimeError("gdl: SwinTransformer runs on the GPU only (no CPU fallback)")
        if x.dim() != 5 or x.shape[1] != 3 or x.shape[3] != self.cfg["img"] or x.shape[4] != self.cfg["img"]:
            raise RuntimeError(f"gdl: SwinTransformer expects [B, 3, T, {self.cfg['img']}, {self.cfg['img']}] frames")
        dtype = self.gdl_dtype or os.environ.get("GDL_DTYPE", "bf16")
        key = (x.shape[0], x.shape[2], str(x.device), dtype)
        if key not in self._eng:
            self._eng[key] = SwinEngine(self.cfg, dtype, x.shape[0], x.shape[2], x.device)
        return self._eng[key]

    def forward_features(self, x):
        raise NotImplementedError("gdl: the token map is internal to the engine; use forward()")

    def forward(self, x):
        return _SwinFn.apply(self, False, x, *self.parameters())

    def forward_pooled(self, x):
        """[B, 3, T, H, W] -> [B, num_features]: the features averaged over the T frames of a sample (not a method of the
        reference class; the counterpart of the pooling basic_model.py:77-80 applies to the ResNet branch)."""
        return _SwinFn.apply(self, x.shape[2] > 1, x, *self.parameters())
