"""Swin visual encoder on libgdl_hip: forward + backward of the function the reference's
`SwinTransformer.forward` computes with `args.pe = 0`, `ape = False`, `patch_norm = True` and every dropout / the
stochastic depth at 0 (/root/reference/models/swin_transformer.py:596-634; block :256-295; window attention :124-157;
patch merging :330-353; patch embedding :478-486).

The host side here only plans buffers and enqueues kernels (PyTorch = device allocator + streams):
  * every nn.Linear and the 4x4/4 patch convolution is a 1x1 convolution of the library -- `gdl_conv_fwd`, `gdl_conv_dgrad`,
    `gdl_conv_wgrad` (MFMA implicit GEMM; weights zero-padded so that every width is a multiple of 64: 96 -> 128 in stage 1,
    a QKV row is three such segments);
  * LayerNorm, bias / GELU / residual, window attention (shift, mask, relative-position bias), patch merging and the final
    token mean are the `gdl_swin_*` kernels of csrc/swin.hip.
Everything saved for the backward is kept (nothing recomputed except the attention probabilities).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib as L


class _PackDesc(ctypes.Structure):  # csrc/swin.hip::SwinPackDesc
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dstT", ctypes.c_void_p)] + \
        [(n, ctypes.c_int32) for n in ("n", "k", "nseg", "nseg_pad", "kseg", "kseg_pad", "np", "kp", "dt", "blk0")]


assert ctypes.sizeof(_PackDesc) == 64


class _RedDesc(ctypes.Structure):  # csrc/swin.hip::SwinRedDesc
    _fields_ = [("partial", ctypes.c_void_p), ("out", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in ("rows", "width", "blk0", "stride")]


assert ctypes.sizeof(_RedDesc) == 32


def _ld(c):
    return (c + 63) // 64 * 64


class _Linear:
    """One nn.Linear (weight [n][k], optional bias) in the kernels' padded layouts."""

    def __init__(self, eng, w_idx, b_idx, n, k, nseg=None, kseg=None):
        self.eng, self.w_idx, self.b_idx, self.n, self.k = eng, w_idx, b_idx, n, k
        self.nseg, self.kseg = nseg or n, kseg or k
        self.nseg_pad, self.kseg_pad = _ld(self.nseg), _ld(self.kseg)
        self.np, self.kp = n // self.nseg * self.nseg_pad, k // self.kseg * self.kseg_pad
        dev, td = eng.device, eng.tdtype
        self.w = torch.empty((self.np, self.kp), dtype=td, device=dev)   # [out][in]  (forward, weight gradient)
        self.wT = torch.empty((self.kp, self.np), dtype=td, device=dev)  # [in][out]  (data gradient)
        self.b = torch.zeros(self.np, dtype=torch.float32, device=dev) if b_idx is not None else None
        self.dw = torch.empty((self.np, self.kp), dtype=torch.float32, device=dev)
        self.db = torch.empty(self.np, dtype=torch.float32, device=dev) if b_idx is not None else None

    def pack_descs(self, params):
        """(src, dst, dstT, n, k, nseg, nseg_pad, kseg, kseg_pad, dtype) records of gdl_swin_pack_batched"""
        e = self.eng
        d = [(params[self.w_idx], self.w, self.wT, self.n, self.k, self.nseg, self.nseg_pad, self.kseg, self.kseg_pad, e.dt)]
        if self.b is not None:
            d.append((params[self.b_idx], self.b, None, self.n, 1, self.nseg, self.nseg_pad, 1, 1, L.GDL_F32))
        return d

    def unpack_descs(self, grads):
        d = [(self.dw, grads[self.w_idx], None, self.n, self.k, self.nseg, self.nseg_pad, self.kseg, self.kseg_pad, L.GDL_F32)]
        if self.b is not None:
            d.append((self.db, grads[self.b_idx], None, self.n, 1, self.nseg, self.nseg_pad, 1, 1, L.GDL_F32))
        return d

    # y[M][np] = x[M][kp] . w^T
    def fwd(self, x, y, M, st):
        e = self.eng
        L.call("gdl_conv_fwd", e.dt, L.ptr(x), L.ptr(self.w), L.ptr(y), None, L.ptr(e.table(L.GATHER_FWD, M, self.kp, self.np)), M, 1,
               1, self.kp, self.np, 1, 1, 1, 0, st)

    # y = x . w^T + b (+ res): bias and residual in the GEMM's epilogue
    def fwd_bias(self, x, y, res, M, st, gelu_out=None):
        e = self.eng
        L.call("gdl_conv_fwd_bias", e.dt, L.ptr(x), L.ptr(self.w), L.ptr(y), L.ptr(self.b), L.ptr(res) if res is not None else None,
               L.ptr(gelu_out) if gelu_out is not None else None, L.ptr(e.table(L.GATHER_FWD, M, self.kp, self.np)), M, 1, 1, self.kp,
               self.np, 1, 1, 1, 0, st)

    # dx[M][kp] = dy[M][np] . w
    def dgrad(self, dy, dx, M, st):
        e = self.eng
        L.call("gdl_conv_dgrad", e.dt, L.ptr(dy), L.ptr(self.wT), L.ptr(dx), None, L.ptr(e.table(L.GATHER_DGRAD, M, self.kp, self.np)),
               M, 1, 1, self.kp, self.np, 1, 1, 1, 0, st)

    # dx[M][kp] = (dy[M][np] . w) * gelu'(u), column sums of dx added to the fixed-point accumulators `acc` ([kp][2] int64)
    def dgrad_gelu(self, dy, dx, u, acc, scale, M, st):
        e = self.eng
        L.call("gdl_conv_dgrad_gelu", e.dt, L.ptr(dy), L.ptr(self.wT), L.ptr(dx), L.ptr(u), L.ptr(acc), scale,
               L.ptr(e.table(L.GATHER_DGRAD, M, self.kp, self.np)), M, 1, 1, self.kp, self.np, 1, 1, 1, 0, st)

    # both gradients from one pass over dy (csrc/linear_bwd.hip) where the shape is one of its: dx = dy . w, dw = dy^T . x
    # bias=True: db = column sums of dy as well -- from the fused kernel where the input width leaves it a padding tile, else by
    # gdl_swin_colsum
    def bwd_pair(self, dy, x, dx, M, st, bias=False):
        e = self.eng
        if e.lib.gdl_linear_bwd_ok(e.dt, M, self.kp, self.np) and self.kseg == self.k:
            ws = e.linear_bwd_ws(M, self.kp, self.np)
            db_in = bias and self.k <= 96
            L.call("gdl_linear_bwd", e.dt, L.ptr(dy), L.ptr(x), L.ptr(self.wT), L.ptr(dx), L.ptr(self.dw), L.ptr(self.db) if db_in else None,
                   L.ptr(ws), ws.numel(), M, self.kp, self.k, self.np, st)
            if bias and not db_in:
                e.colsum(dy, None, self, M, self.np, st)
        else:
            if bias:
                e.colsum(dy, None, self, M, self.np, st)
            self.wgrad(dy, x, M, st)
            self.dgrad(dy, dx, M, st)

    # dw[np][kp] = dy^T . x
    def wgrad(self, dy, x, M, st):
        e = self.eng
        ws = e.wgrad_ws(M, self.kp, self.np)
        L.call("gdl_conv_wgrad", e.dt, L.ptr(dy), L.ptr(x), L.ptr(self.dw), L.ptr(e.table(L.GATHER_FWD, M, self.kp, self.np)), M, 1, 1,
               self.kp, self.np, 1, 1, 1, 0, L.ptr(ws), ws.numel(), st)


class _Norm:
    def __init__(self, eng, w_idx, b_idx, c):
        self.eng, self.w_idx, self.b_idx, self.c, self.ld = eng, w_idx, b_idx, c, _ld(c)
        dev = eng.device
        self.g = torch.zeros(self.ld, dtype=torch.float32, device=dev)
        self.b = torch.zeros(self.ld, dtype=torch.float32, device=dev)
        self.dgb = torch.empty((3, self.ld), dtype=torch.float32, device=dev)  # d gamma, d beta, (bwd(colsum=True)) column sums of dx
        self.partial = None  # (deferred folds: allocated by the first backward)

    def pack_descs(self, params):
        return [(params[src], dst, None, self.c, 1, self.c, self.ld, 1, 1, L.GDL_F32) for src, dst in ((self.w_idx, self.g), (self.b_idx, self.b))]

    def unpack_descs(self, grads):
        return [(self.dgb[row], grads[dst], None, self.c, 1, self.c, self.ld, 1, 1, L.GDL_F32) for row, dst in ((0, self.w_idx), (1, self.b_idx))]

    def fwd(self, x, y, stats, M, st):
        L.call("gdl_swin_ln_fwd", self.eng.dt, L.ptr(x), L.ptr(self.g), L.ptr(self.b), L.ptr(y), L.ptr(stats), M, self.c, self.ld, st)

    def bwd(self, dy, x, stats, add, dx, M, st, colsum=False):
        """colsum: dgb[2] = column sums of dx -- the bias gradient of the Linear whose output gradient dx is"""
        e = self.eng
        name = "gdl_swin_ln_bwd_colsum" if colsum else "gdl_swin_ln_bwd"
        if not e.defer_folds:
            L.call(name, e.dt, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(self.g), L.ptr(add) if add is not None else None, L.ptr(dx),
                   L.ptr(self.dgb), L.ptr(e.partial), M, self.c, self.ld, st)
            return
        # the partial rows stay in a buffer of this LayerNorm's own; SwinEngine._fold_all folds every call's rows in one launch
        if self.partial is None:
            self.rows, self.M = e.lib.gdl_swin_ln_bwd_rows(e.dt, M, self.ld), M
            if self.rows <= 0:
                raise L.GdlError("gdl_swin_ln_bwd_rows: bad arguments")
            self.partial = torch.empty(self.rows * 3 * self.ld, dtype=torch.float32, device=e.device)
        assert M == self.M
        L.call(name, e.dt, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(self.g), L.ptr(add) if add is not None else None, L.ptr(dx), None,
               L.ptr(self.partial), M, self.c, self.ld, st)
        nr = 3 if colsum else 2
        e.defer_fold(self.partial, self.dgb, self.rows, nr * self.ld, nr * self.ld)


class SwinEngine:
    """A planned Swin encoder for fixed (cfg, dtype, B, T).  `cfg`: dict(img, patch, embed, depths, heads, window, mlp).
    Parameters / gradients are lists of float32 CUDA tensors in the reference's named_parameters() order
    (`param_shapes()`); forward(x [B,3,T,img,img] f32) -> float32 [B*T, C_last]; backward(dfeat, grads)."""

    def __init__(self, cfg, dtype, B, T, device):
        self.lib = L.load()
        self.cfg, self.B, self.T = dict(cfg), B, T
        self.N = B * T
        self.dt = L.dtype_code(dtype)
        self.tdtype = L.torch_dtype(self.dt)
        self.device = torch.device(device)
        self._tables, self._wg = {}, None
        # HIP-graph replay of the forward / backward launch sequences (see _run); GDL_NOGRAPH=1 (with GDL_TUNING=1) keeps every
        # pass eager -- the PMC scripts set it, so that counters are collected over ordinary launches
        import os

        self._graphs = {}
        self.use_graph = not (os.environ.get("GDL_TUNING") == "1" and os.environ.get("GDL_NOGRAPH") == "1")
        self.serial = 0  # forward counter: a backward must follow the forward whose activations are in the buffers
        E, p = cfg["embed"], cfg["patch"]
        assert cfg["img"] % p == 0 and 3 * p * p <= 64
        res = cfg["img"] // p
        names = []
        idx = {}

        def reg(name, shape):
            idx[name] = len(names)
            names.append((name, tuple(shape)))
            return idx[name]

        dev, td, N = self.device, self.tdtype, self.N

        def buf(rows, cols, dtype=None):
            return torch.empty((rows, cols), dtype=dtype or td, device=dev)

        # ---- patch embedding
        self.pe = _Linear(self, reg("patch_embed.proj.weight", (E, 3, p, p)), reg("patch_embed.proj.bias", (E,)), E, 3 * p * p)
        self.pe_norm = _Norm(self, reg("patch_embed.norm.weight", (E,)), reg("patch_embed.norm.bias", (E,)), E)
        M0 = N * res * res
        self.pe_rows = buf(M0, 64)
        self.pe_out = buf(M0, _ld(E))
        self.pe_stats = buf(M0, 2, torch.float32)
        self.x0 = buf(M0, _ld(E))
        # ---- stages
        self.stages = []
        nl = len(cfg["depths"])
        maxw, maxld = 0, 64
        x_prev = None
        for i, (depth, nh) in enumerate(zip(cfg["depths"], cfg["heads"])):
            C, r = E << i, res >> i
            assert C == nh * 32, "head dimension must be 32"
            ws = min(cfg["window"], r)
            M, ld, hid = N * r * r, _ld(C), cfg["mlp"] * C
            blocks = []
            for j in range(depth):
                pre = f"layers.{i}.blocks.{j}."
                b = {"shift": 0 if (j % 2 == 0 or r <= cfg["window"]) else cfg["window"] // 2, "k": sum(cfg["depths"][:i]) + j}
                b["norm1"] = _Norm(self, reg(pre + "norm1.weight", (C,)), reg(pre + "norm1.bias", (C,)), C)
                b["table_idx"] = reg(pre + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, nh))
                b["qkv"] = _Linear(self, reg(pre + "attn.qkv.weight", (3 * C, C)), reg(pre + "attn.qkv.bias", (3 * C,)), 3 * C, C, nseg=C)
                b["proj"] = _Linear(self, reg(pre + "attn.proj.weight", (C, C)), reg(pre + "attn.proj.bias", (C,)), C, C)
                b["norm2"] = _Norm(self, reg(pre + "norm2.weight", (C,)), reg(pre + "norm2.bias", (C,)), C)
                b["fc1"] = _Linear(self, reg(pre + "mlp.fc1.weight", (hid, C)), reg(pre + "mlp.fc1.bias", (hid,)), hid, C)
                b["fc2"] = _Linear(self, reg(pre + "mlp.fc2.weight", (C, hid)), reg(pre + "mlp.fc2.bias", (C,)), C, hid)
                # saved for the backward
                b["x_in"] = None  # set below (the previous block's output)
                b["stats1"], b["h"], b["qkv_a"], b["attn"] = buf(M, 2, torch.float32), buf(M, ld), buf(M, 3 * ld), buf(M, ld)
                b["x_mid"], b["stats2"], b["m"] = buf(M, ld), buf(M, 2, torch.float32), buf(M, ld)
                b["u"], b["a"], b["x_out"] = buf(M, _ld(hid)), buf(M, _ld(hid)), buf(M, ld)
                blocks.append(b)
                maxld = max(maxld, 3 * ld, _ld(hid))
            st = {"C": C, "r": r, "ws": ws, "nh": nh, "M": M, "ld": ld, "hid": hid, "blocks": blocks}
            if i < nl - 1:
                pre = f"layers.{i}.downsample."
                st["red"] = _Linear(self, reg(pre + "reduction.weight", (2 * C, 4 * C)), None, 2 * C, 4 * C)
                st["mnorm"] = _Norm(self, reg(pre + "norm.weight", (4 * C,)), reg(pre + "norm.bias", (4 * C,)), 4 * C)
                st["cat"], st["catn"], st["mstats"] = buf(M // 4, 4 * C), buf(M // 4, 4 * C), buf(M // 4, 2, torch.float32)
                st["merged"] = buf(M // 4, _ld(2 * C))
                maxld = max(maxld, 4 * C)
            self.stages.append(st)
            maxw = max(maxw, self.lib.gdl_swin_attn_bwd_workspace_bytes(N, r, r, ws, nh))
        Cl = E << (nl - 1)
        self.out_norm = _Norm(self, reg("norm.weight", (Cl,)), reg("norm.bias", (Cl,)), Cl)
        last = self.stages[-1]
        self.out_stats, self.out_ln = buf(last["M"], 2, torch.float32), buf(last["M"], last["ld"])
        self.feat = torch.empty((N, Cl), dtype=torch.float32, device=dev)
        self.feat_b = torch.empty((B, Cl), dtype=torch.float32, device=dev)  # averaged over the T frames of a sample
        self.C_out, self.L_out = Cl, last["r"] * last["r"]
        self.names = names
        # scratch: LayerNorm / column-sum partials, attention table partials, gradient buffers
        self.partial = torch.empty(self.lib.gdl_swin_partial_bytes(maxld), dtype=torch.uint8, device=dev)
        # Round 6: the folds of the LayerNorm backwards' and column-sum passes' partial rows (42 launches per Swin-T backward, each
        # on the branch's only chain, each producing a parameter gradient nobody reads before the optimizer) wait until the end of
        # the backward and run as ONE launch (gdl_swin_partial_reduce_batched; a partial buffer per call site instead of the shared
        # one; bit-identical results).  GDL_SWIN_DEFER_FOLDS=0 (tuning aid): a fold launch behind every pass, as before.
        self.defer_folds = not (os.environ.get("GDL_TUNING") == "1" and os.environ.get("GDL_SWIN_DEFER_FOLDS") == "0")
        self._fold_jobs, self._fold_key, self._fold_tabs = None, None, {}
        self.tpart = torch.empty(max(maxw, 4), dtype=torch.uint8, device=dev)
        M1, ld1 = self.stages[0]["M"], self.stages[0]["ld"]
        wide = max(max(3 * s["ld"], _ld(s["hid"])) * s["M"] for s in self.stages)
        self.g_a = torch.empty(M1 * ld1, dtype=td, device=dev)   # gradient of the residual stream (ping)
        self.g_b = torch.empty(M1 * ld1, dtype=td, device=dev)   # (pong)
        self.g_c = torch.empty(M1 * ld1, dtype=td, device=dev)   # branch gradient at token width
        self.g_w = torch.empty(wide, dtype=td, device=dev)       # branch gradient at QKV / hidden width
        self.g_cat = torch.empty(max([s["M"] // 4 * 4 * s["C"] for s in self.stages[:-1]] + [1]), dtype=td, device=dev)
        self.g_cat2 = torch.empty_like(self.g_cat)
        # fc1 bias gradients: fixed-point column-sum accumulators of the fused fc2 data gradient (gdl_conv_dgrad_gelu), one arena
        # for all blocks (zeroed / converted once per backward phase); the float results ARE the Linears' db buffers (views)
        self.fuse_gelu = not (os.environ.get("GDL_TUNING") == "1" and os.environ.get("GDL_SWIN_FUSE_GELU") == "0")
        # (tuning aid, OFF by default: measured 26.8 ms either way at 192 frames -- the Linears' three GEMMs are all HBM-bound, beside
        # the weight gradients the data gradients slow down by what the weight gradients would have taken: 4.2 -> 6.4 ms)
        tot = 0
        self.fc1_off = []
        for s in self.stages:
            for b in s["blocks"]:
                b["fc1_off"] = tot
                tot += b["fc1"].np
            self.fc1_off.append(tot)  # end offset of the stage
        self.fc1_acc = torch.zeros((tot, 2), dtype=torch.int64, device=dev)
        self.fc1_db = torch.zeros(tot, dtype=torch.float32, device=dev)
        if self.fuse_gelu:
            for s in self.stages:
                for b in s["blocks"]:
                    b["fc1"].db = self.fc1_db[b["fc1_off"]:b["fc1_off"] + b["fc1"].np]
        # Bias gradients of the Linears fed by the residual stream's gradient (fc2, proj, the patch embedding) = a third result
        # row of the LayerNorm backward that produced that gradient; `cs_dx`: this block's fc2 gets it from the norm1 backward
        # of the NEXT block (or the final norm's) -- not at a stage's last block below the last stage (patch merging in between)
        self.fuse_ln = not (os.environ.get("GDL_TUNING") == "1" and os.environ.get("GDL_SWIN_FUSE_LN") == "0")
        for si, s in enumerate(self.stages):
            for j, b in enumerate(s["blocks"]):
                b["cs_dx"] = None
                if not self.fuse_ln:
                    continue
                b["proj"].db = b["norm2"].dgb[2]
                if j + 1 < len(s["blocks"]):
                    b["cs_dx"] = s["blocks"][j + 1]["norm1"]
                    s["blocks"][j + 1]["cs_prev"] = True
                elif si == len(self.stages) - 1:
                    b["cs_dx"] = self.out_norm
                if b["cs_dx"] is not None:
                    b["fc2"].db = b["cs_dx"].dgb[2]
        if self.fuse_ln:
            self.pe.db = self.pe_norm.dgb[2]
        # |mean over the rows| < 2^7 for every column of a hidden-width gradient; one scale for all stages (the widest M)
        self.fc1_scale = 2.0 ** (62 - 7 - max(1, (M1 - 1).bit_length()))
        # the weight gradients' split-K workspace at its final size (it is shared by launches on two streams: no regrowth later)
        need = [self.pe_rows.shape[0], self.pe.kp, self.pe.np]
        nb = self.lib.gdl_conv_wgrad_workspace_bytes(self.dt, *need[:1], 1, 1, *need[1:], 1, 1, 1, 0)
        for s in self.stages:
            for b in s["blocks"]:
                for k in ("qkv", "proj", "fc1", "fc2"):
                    nb = max(nb, self.lib.gdl_conv_wgrad_workspace_bytes(self.dt, s["M"], 1, 1, b[k].kp, b[k].np, 1, 1, 1, 0))
            if "red" in s:
                nb = max(nb, self.lib.gdl_conv_wgrad_workspace_bytes(self.dt, s["M"] // 4, 1, 1, s["red"].kp, s["red"].np, 1, 1, 1, 0))
        self._wg = torch.empty(nb, dtype=torch.uint8, device=dev)
        # ... and the partials of the fused data + weight gradient (csrc/linear_bwd.hip; 0 bytes for the shapes it does not take)
        lb = max([self.lib.gdl_linear_bwd_workspace_bytes(s["M"], b[k].kp, b[k].np) for s in self.stages for b in s["blocks"]
                  for k in ("qkv", "fc1")] + [16])
        self._lbws = torch.empty(lb, dtype=torch.uint8, device=dev)
        self._params = None
        self.have_fwd = False
        # stochastic depth (swin_transformer.py:218, 290, 293): per-frame scales of the two residual branches of every block, copied
        # here from the caller's tensor by a forward that is given one (fixed address: the launch sequences may be graph replays)
        self.nblocks = sum(cfg["depths"])
        self.drop_buf = torch.ones((self.nblocks, 2, N), dtype=torch.float32, device=dev)
        self._drop = False  # the last forward ran with DropPath: its backward scales the branch gradients the same way

    # ------------------------------------------------------------------ plumbing
    def param_shapes(self):
        return list(self.names)

    def table(self, mode, M, C, K):
        key = (mode, M, C if mode == L.GATHER_FWD else K)
        t = self._tables.get(key)
        if t is None:
            nb = self.lib.gdl_conv_table_bytes(mode, M, 1, 1, 1, 1, 1, 0)
            t = torch.empty(nb, dtype=torch.uint8, device=self.device)
            L.call("gdl_conv_build_table", mode, self.dt, M, 1, 1, C, K, 1, 1, 1, 0, L.ptr(t), L.cur_stream())
            self._tables[key] = t
        return t

    def linear_bwd_ws(self, M, K, N):
        need = self.lib.gdl_linear_bwd_workspace_bytes(M, K, N)
        if getattr(self, "_lbws", None) is None or self._lbws.numel() < need:
            self._lbws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._lbws

    def wgrad_ws(self, M, C, K):
        nb = self.lib.gdl_conv_wgrad_workspace_bytes(self.dt, M, 1, 1, C, K, 1, 1, 1, 0)
        if self._wg is None or self._wg.numel() < nb:
            self._wg = torch.empty(nb, dtype=torch.uint8, device=self.device)
        return self._wg

    def set_params(self, params):
        if len(params) != len(self.names):
            raise L.GdlError(f"SwinEngine.set_params: {len(params)} tensors, expected {len(self.names)}")
        for p, (n, shape) in zip(params, self.names):
            if p.dtype != torch.float32 or not p.is_contiguous() or tuple(p.shape) != shape or p.device != self.device:
                raise L.GdlError(f"SwinEngine.set_params: {n} must be a contiguous float32 tensor of shape {shape} on {self.device}")
        self._params = list(params)

    def _desc_table(self, recs, unpack):
        """device array of SwinPackDesc + its block count (1024 elements per block)"""
        arr = (_PackDesc * len(recs))()
        blk = 0
        for d, (src, dst, dstT, n, k, nseg, nseg_pad, kseg, kseg_pad, dt) in zip(arr, recs):
            d.src, d.dst, d.dstT = src.data_ptr(), dst.data_ptr(), (dstT.data_ptr() if dstT is not None else None)
            d.n, d.k, d.nseg, d.nseg_pad, d.kseg, d.kseg_pad = n, k, nseg, nseg_pad, kseg, kseg_pad
            d.np, d.kp, d.dt, d.blk0 = n // nseg * nseg_pad, k // kseg * kseg_pad, dt, blk
            blk += ((n * k if unpack else d.np * d.kp) + 1023) // 1024
        host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
        return host.to(self.device), len(recs), blk

    def _pack_all(self, st):
        key = tuple(p.data_ptr() for p in self._params)
        if getattr(self, "_pack_key", None) != key:
            recs = [r for o in self._linears_norms() for r in o.pack_descs(self._params)]
            self._pack_tab, self._pack_key = self._desc_table(recs, False), key
        t, n, blk = self._pack_tab
        L.call("gdl_swin_pack_batched", L.ptr(t), n, blk, 0, st)

    def _unpack_all(self, grads, st, part=None):
        key = tuple(g.data_ptr() for g in grads)
        cache = self.__dict__.setdefault("_unpack_tabs", {})
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
        if part not in cache:
            recs = [r for o in self._linears_norms(part) for r in o.unpack_descs(grads)]
            cache[part] = self._desc_table(recs, True)
        t, n, blk = cache[part]
        L.call("gdl_swin_pack_batched", L.ptr(t), n, blk, 1, st)

    # ------------------------------------------------------------------ deferred folds of partial rows
    def colsum(self, g, u, lin, M, ld, st):
        """lin.db = column sums of g [M][ld] (u given: g <- g * gelu'(u) first)"""
        if not self.defer_folds:
            L.call("gdl_swin_colsum", self.dt, L.ptr(g), L.ptr(u) if u is not None else None, L.ptr(lin.db), L.ptr(self.partial), M, ld, st)
            return
        if getattr(lin, "cs_partial", None) is None:
            lin.cs_rows, lin.cs_M = self.lib.gdl_swin_colsum_rows(self.dt, M, ld), M
            if lin.cs_rows <= 0:
                raise L.GdlError("gdl_swin_colsum_rows: bad arguments")
            lin.cs_partial = torch.empty(lin.cs_rows * ld, dtype=torch.float32, device=self.device)
        assert M == lin.cs_M
        L.call("gdl_swin_colsum", self.dt, L.ptr(g), L.ptr(u) if u is not None else None, None, L.ptr(lin.cs_partial), M, ld, st)
        self.defer_fold(lin.cs_partial, lin.db, lin.cs_rows, ld, ld)

    def _fold_begin(self, key):
        """start of a backward body: its fold jobs are collected once per (phase, DropPath) and kept as a device table"""
        self._fold_key = key
        self._fold_jobs = None if key in self._fold_tabs else []

    def defer_fold(self, partial, out, rows, width, stride):
        if self._fold_jobs is None:  # this body's table exists already (the launch sequence of a body is fixed)
            return
        o0, o1 = out.data_ptr(), out.data_ptr() + 4 * width
        for j in self._fold_jobs:  # a later pass that writes into an earlier job's result takes those columns over (DropPath: the
            j0, j1 = j[1], j[1] + 4 * j[3]  # column sums of the scaled branch gradient replace the LayerNorm backward's third row)
            if o0 < j1 and o1 > j0:
                if not (o0 > j0 and o1 == j1):
                    raise L.GdlError("SwinEngine: overlapping fold results that are not a replaced last row")
                j[3] = (o0 - j0) // 4
        self._fold_jobs.append([partial.data_ptr(), out.data_ptr(), rows, width, stride])

    def _fold_all(self, st):
        """one launch: every deferred fold of this backward body (before its gradients are unpacked)"""
        if not self.defer_folds:
            return
        tab = self._fold_tabs.get(self._fold_key)
        if tab is None:
            jobs = self._fold_jobs
            arr = (_RedDesc * len(jobs))()
            blk = 0
            for d, (partial, out, rows, width, stride) in zip(arr, jobs):
                d.partial, d.out, d.rows, d.width, d.blk0, d.stride = partial, out, rows, width, blk, stride
                blk += (width + 15) // 16
            host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
            tab = self._fold_tabs[self._fold_key] = (host.to(self.device), len(jobs), blk)
            self._fold_jobs = None
        t, n, blk = tab
        if n:
            L.call("gdl_swin_partial_reduce_batched", L.ptr(t), n, blk, st)

    def _linears_norms(self, part=None):
        """part None: every Linear / LayerNorm; "last": the last stage's blocks and the final norm (the parameters whose
        gradients backward phase 1 completes -- DGLTrainer's visual_l4 bucket); "rest": the others."""
        last = len(self.stages) - 1
        if part != "last":
            yield self.pe
            yield self.pe_norm
        for si, s in enumerate(self.stages):
            if part is None or (part == "last") == (si == last):
                for b in s["blocks"]:
                    for k in ("norm1", "qkv", "proj", "norm2", "fc1", "fc2"):
                        yield b[k]
            if "red" in s and part != "last":
                yield s["red"]
                yield s["mnorm"]
        if part != "rest":
            yield self.out_norm

    def _v(self, flat, rows, cols):
        return flat[:rows * cols].view(rows, cols)

    # ------------------------------------------------------------------ forward
    def forward(self, x, pool_frames=False, out=None, drop_scales=None):
        """-> float32 [B*T, C_last] (the reference's contract), or [B, C_last] averaged over the frames of a sample
        (pool_frames: what basic_model.py:77-80 does for the ResNet branch -- the samples' frames are consecutive rows).
        drop_scales: None (DropPath is the identity: eval mode, or drop_path_rate = 0) or a float32 [blocks][2][B*T] tensor on
        the device -- the factor each frame's attention / Mlp branch is multiplied with before it joins the residual stream, 0
        or 1 / keep_prob (timm's drop_path with scale_by_keep; the caller draws them: `models.swin_transformer.drop_path_scales`)."""
        if self._params is None:
            raise L.GdlError("SwinEngine.forward: parameters not set")
        cfg, dt, N, st = self.cfg, self.dt, self.N, L.cur_stream()
        B, T = self.B, self.T
        if tuple(x.shape) != (B, 3, T, cfg["img"], cfg["img"]) or x.dtype != torch.float32 or not x.is_contiguous() or \
                x.device != self.device:
            raise L.GdlError("SwinEngine.forward: x must be a contiguous float32 [B, 3, T, img, img] tensor on the engine's device")
        if out is None:
            out = self.feat_b if pool_frames else self.feat
        elif tuple(out.shape) != ((B if pool_frames else N), self.C_out) or out.dtype != torch.float32 or not out.is_contiguous():
            raise L.GdlError("SwinEngine.forward: `out` must be a contiguous float32 [B*T or B, C_last] tensor")
        drop = drop_scales is not None
        if drop:
            if tuple(drop_scales.shape) != tuple(self.drop_buf.shape) or drop_scales.dtype != torch.float32 or \
                    drop_scales.device != self.device:
                raise L.GdlError(f"SwinEngine.forward: drop_scales must be a float32 {list(self.drop_buf.shape)} tensor on the engine's device")
            self.drop_buf.copy_(drop_scales, non_blocking=True)
        self._drop = drop
        L.call("gdl_swin_patch_gather", dt, L.ptr(x), L.ptr(self.pe_rows), B, T, cfg["img"], cfg["img"], cfg["patch"], st)
        key = ("f", out.data_ptr(), bool(pool_frames), drop) + tuple(p.data_ptr() for p in self._params)
        self._run(key, lambda: self._forward_body(out, pool_frames, drop))
        self.have_fwd = True
        self.serial += 1
        return out

    def _run(self, key, body):
        """Run `body` (a fixed launch sequence over fixed buffers) -- eagerly the first two times and whenever the
        measurement tap is recording, afterwards as a replay of its captured HIP graph: ~400 launches per pass enqueued from
        Python would otherwise leave the device waiting for the host."""
        if not self.use_graph or self.lib.gdl_prof_enabled():
            return body()
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 8:  # callers that pass fresh buffers every time (autograd) never repeat a key
                self._graphs = {k: v for k, v in self._graphs.items() if v["g"] is not None}
                if len(self._graphs) >= 8:
                    return body()
            ent = self._graphs[key] = {"n": 0, "g": None}
        if ent["g"] is not None:
            return ent["g"].replay()
        ent["n"] += 1
        if ent["n"] <= 2:
            return body()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                body()
        except RuntimeError as e:  # capture not possible on this stack: stay eager from here on
            self.use_graph = False
            torch.cuda.synchronize(self.device)
            if torch.cuda.is_current_stream_capturing():
                raise L.GdlError(f"SwinEngine: graph capture failed and the stream is still capturing: {e}") from e
            return body()
        ent["g"] = g
        g.replay()

    def _drop_path(self, y, res, k, branch, out, M, ld, st):
        """out = (res or 0) + drop_buf[k][branch][frame of the row] * y"""
        L.call("gdl_swin_drop_path", self.dt, L.ptr(y), L.ptr(res) if res is not None else None, L.ptr(self.drop_buf[k, branch]),
               L.ptr(out), M, M // self.N, ld, st)

    def _forward_body(self, out, pool_frames, drop=False):
        cfg, dt, N, st = self.cfg, self.dt, self.N, L.cur_stream()
        B, T = self.B, self.T
        P = self._params
        self._pack_all(st)  # float32 masters -> kernel layouts (one launch per step, like the encoder's weight pack)
        M0 = self.pe_rows.shape[0]
        self.pe.fwd_bias(self.pe_rows, self.pe_out, None, M0, st)
        xcur = self.x0
        self.pe_norm.fwd(self.pe_out, xcur, self.pe_stats, M0, st)
        for s in self.stages:
            M, ld, r = s["M"], s["ld"], s["r"]
            for b in s["blocks"]:
                b["x_in"] = xcur
                b["norm1"].fwd(xcur, b["h"], b["stats1"], M, st)
                b["qkv"].fwd_bias(b["h"], b["qkv_a"], None, M, st)
                L.call("gdl_swin_attn_fwd", dt, L.ptr(b["qkv_a"]), L.ptr(P[b["table_idx"]]), L.ptr(b["attn"]), N, r, r, s["ws"],
                       b["shift"], s["nh"], ld, st)
                if drop:  # the branch without the residual, then x_mid = x_in + scale[frame] * branch (g_c: free during a forward)
                    br = self._v(self.g_c, M, ld)
                    b["proj"].fwd_bias(b["attn"], br, None, M, st)
                    self._drop_path(br, xcur, b["k"], 0, b["x_mid"], M, ld, st)
                else:
                    b["proj"].fwd_bias(b["attn"], b["x_mid"], xcur, M, st)
                b["norm2"].fwd(b["x_mid"], b["m"], b["stats2"], M, st)
                b["fc1"].fwd_bias(b["m"], b["u"], None, M, st, gelu_out=b["a"])  # u = fc1(m) + b, a = gelu(u)
                if drop:
                    b["fc2"].fwd_bias(b["a"], br, None, M, st)
                    self._drop_path(br, b["x_mid"], b["k"], 1, b["x_out"], M, ld, st)
                else:
                    b["fc2"].fwd_bias(b["a"], b["x_out"], b["x_mid"], M, st)
                xcur = b["x_out"]
            if "red" in s:
                L.call("gdl_swin_merge", dt, L.ptr(xcur), L.ptr(s["cat"]), N, r, r, s["C"], ld, 0, st)
                s["mnorm"].fwd(s["cat"], s["catn"], s["mstats"], M // 4, st)
                s["red"].fwd(s["catn"], s["merged"], M // 4, st)
                xcur = s["merged"]
        last = self.stages[-1]
        self.x_last = xcur
        self.out_norm.fwd(xcur, self.out_ln, self.out_stats, last["M"], st)
        L.call("gdl_swin_token_mean", dt, L.ptr(self.out_ln), L.ptr(out), B if pool_frames else N,
               self.L_out * (T if pool_frames else 1), self.C_out, last["ld"], st)

    # ------------------------------------------------------------------ backward
    def backward(self, dfeat, grads, serial=None, phase=0):
        """dfeat float32 [B*T, C_last] (or [B, C_last] for a pool_frames forward); grads: float32 tensors of the
        parameter shapes (overwritten).  serial: the engine's `serial` right after the forward being differentiated -- if
        another forward has run through the engine since (a validation pass of the same shape, two forwards before one
        backward), its activations are gone and this raises instead of returning gradients of the wrong pass.
        phase 0: the whole backward.  phase 1: the upstream gradient, the final norm and the LAST stage -- afterwards the gradients
        of `layers.<last>.*` and `norm.*` (half of Swin-T's parameters) are final on the stream, so a data-parallel caller
        can start their all-reduce (DGLTrainer's visual_l4 bucket) while phase 2 (the other stages and the patch embedding;
        same `dfeat` / `grads` arguments) runs -- the counterpart of gdl_encoder_backward_phase."""
        if not self.have_fwd:
            raise L.GdlError("SwinEngine.backward: no forward to differentiate")
        if serial is not None and serial != self.serial:
            raise L.GdlError("SwinEngine.backward: the engine has run another forward since the one being differentiated "
                             f"(forward {serial}, now {self.serial}): its saved activations were overwritten")
        if len(grads) != len(self.names):
            raise L.GdlError("SwinEngine.backward: wrong number of gradient tensors")
        N = self.N
        pooled = dfeat.shape[0] == self.B and self.T > 1
        if tuple(dfeat.shape) != ((self.B if pooled else N), self.C_out) or dfeat.dtype != torch.float32 or not dfeat.is_contiguous():
            raise L.GdlError("SwinEngine.backward: dfeat must be a contiguous float32 [B*T or B, C_last] tensor")
        for gten, (n, shape) in zip(grads, self.names):
            if gten.dtype != torch.float32 or not gten.is_contiguous() or tuple(gten.shape) != shape:
                raise L.GdlError(f"SwinEngine.backward: the gradient of {n} must be a contiguous float32 tensor of shape {shape}")
        if phase not in (0, 1, 2):
            raise L.GdlError("SwinEngine.backward: phase must be 0, 1 or 2")
        if phase == 2 and getattr(self, "_bw_phase1_serial", None) != self.serial:
            raise L.GdlError("SwinEngine.backward: phase 2 without phase 1 of the same forward")
        if phase == 1:
            self._bw_phase1_serial = self.serial
        grads = list(grads)
        key = ("b", phase, dfeat.data_ptr(), pooled, self._drop) + tuple(t.data_ptr() for t in grads) + tuple(p.data_ptr() for p in self._params)
        self._run(key, lambda: self._backward_body(dfeat, grads, pooled, phase, self._drop))
        if phase == 2:
            self._bw_phase1_serial = None

    def _backward_body(self, dfeat, grads, pooled, phase=0, drop=False):
        dt, N, st = self.dt, self.N, L.cur_stream()
        P = self._params
        last = self.stages[-1]
        nst = len(self.stages)
        M, ld = last["M"], last["ld"]
        ga, gb = self._v(self.g_a, M, ld), self._v(self.g_b, M, ld)
        self._fold_begin((phase, bool(drop), bool(pooled)))
        if phase != 2:
            L.call("gdl_swin_token_mean_bwd", dt, L.ptr(dfeat), L.ptr(ga), self.B if pooled else N,
                   self.L_out * (self.T if pooled else 1), self.C_out, ld, st)
            self.out_norm.bwd(ga, self.x_last, self.out_stats, None, gb, M, st, colsum=self.fuse_ln)
        # dx: gradient of the current stage's output tokens.  (Every block hands dx back in the buffer it got it in, so after
        # the last stage it sits in g_b again: phase 2 picks it up there without any state from phase 1 -- both phases may
        # be HIP-graph replays.)
        dx, spare = gb, ga
        first_si = nst - 2 if phase == 2 else nst - 1
        last_si = nst - 1 if phase == 1 else 0
        fuse = self.fuse_gelu
        # (round 3 also had the Linears' weight gradients on a side stream with per-buffer reuse events: 26.8 ms either way -- all
        # three GEMMs of a Linear are HBM-bound -- removed in round 4, tools/experiments/r4_pruned_knobs.diff.txt)
        def wg(lin, dy, x, rows, tag):
            return lin.wgrad(dy, x, rows, st)

        def need(tag=None):
            return None

        acc_lo = self.fc1_off[last_si - 1] if last_si > 0 else 0
        acc_hi = self.fc1_off[first_si]
        if fuse:
            self.fc1_acc[acc_lo:acc_hi].zero_()
        for si in range(first_si, last_si - 1, -1):
            s = self.stages[si]
            M, ld, r = s["M"], s["ld"], s["r"]
            if "red" in s:  # dx is the gradient of the merged tokens [M/4][ld(2C)]
                need()  # (the buffers change roles here: every pending weight gradient first)
                M4, C4 = M // 4, 4 * s["C"]
                gc, gc2 = self._v(self.g_cat, M4, C4), self._v(self.g_cat2, M4, C4)
                s["red"].wgrad(dx, s["catn"], M4, st)
                s["red"].dgrad(dx, gc, M4, st)
                s["mnorm"].bwd(gc, s["cat"], s["mstats"], None, gc2, M4, st)
                dx = self._v(self.g_a, M, ld)
                spare = self._v(self.g_b, M, ld)
                L.call("gdl_swin_merge", dt, L.ptr(gc2), L.ptr(dx), N, r, r, s["C"], ld, 1, st)
            else:
                dx, spare = self._v(dx.reshape(-1), M, ld), self._v(spare.reshape(-1), M, ld)
            gtok = self._v(self.g_c, M, ld)
            for b in reversed(s["blocks"]):
                hid_ld = b["fc1"].np
                gw = self._v(self.g_w, M, hid_ld)
                # x_out = x_mid + fc2(gelu(fc1(norm2(x_mid)))) ; dx = d x_out
                # DropPath: the Mlp branch sees g2 = scale[frame] * dx (in gtok until d m is written there); its bias gradient is a
                # column sum of g2 -- the row the LayerNorm backward left for it sums the unscaled dx
                g2 = dx
                if drop:
                    g2 = gtok
                    self._drop_path(dx, None, b["k"], 1, g2, M, ld, st)
                if b["cs_dx"] is None or drop:
                    self.colsum(g2, None, b["fc2"], M, ld, st)
                wg(b["fc2"], g2, b["a"], M, "fc2")
                need("qkv")  # (the previous block's: it reads g_w)
                if fuse:  # d u = (dx . W2) * gelu'(u) and fc1's bias gradient in the GEMM's epilogue
                    b["fc2"].dgrad_gelu(g2, gw, b["u"], self.fc1_acc[b["fc1_off"]:], self.fc1_scale, M, st)
                else:
                    b["fc2"].dgrad(g2, gw, M, st)                                 # d a
                    self.colsum(gw, b["u"], b["fc1"], M, hid_ld, st)  # d u
                b["fc1"].bwd_pair(gw, b["m"], gtok, M, st)                        # fc1's weight gradient and d m
                need("proj")  # (the previous block's: it reads `spare`)
                b["norm2"].bwd(gtok, b["x_mid"], b["stats2"], dx, spare, M, st, colsum=self.fuse_ln)   # spare = d x_mid
                # x_mid = x_in + proj(attn(qkv(norm1(x_in))))
                g1 = spare
                if drop:  # g1 = scale[frame] * d x_mid, in g_w (free between fc1's data gradient and the attention backward)
                    g1 = self._v(self.g_w, M, ld)
                    self._drop_path(spare, None, b["k"], 0, g1, M, ld, st)
                if not self.fuse_ln or drop:
                    self.colsum(g1, None, b["proj"], M, ld, st)
                wg(b["proj"], g1, b["attn"], M, "proj")
                b["proj"].dgrad(g1, gtok, M, st)                               # d attention output
                gq = self._v(self.g_w, M, 3 * ld)
                need("fc1")
                L.call("gdl_swin_attn_bwd", dt, L.ptr(b["qkv_a"]), L.ptr(P[b["table_idx"]]), L.ptr(gtok), L.ptr(gq),
                       L.ptr(grads[b["table_idx"]]), L.ptr(self.tpart), N, r, r, s["ws"], b["shift"], s["nh"], ld, st)
                b["qkv"].bwd_pair(gq, b["h"], gtok, M, st, bias=True)            # qkv's weight and bias gradients and d h
                need("fc2")
                b["norm1"].bwd(gtok, b["x_in"], b["stats1"], spare, dx, M, st, colsum=b.get("cs_prev", False))    # dx = d x_in
        need()
        if fuse:
            L.call("gdl_acc_to_float", L.ptr(self.fc1_acc[acc_lo:]), acc_hi - acc_lo, 1.0 / self.fc1_scale, L.ptr(self.fc1_db[acc_lo:]), st)
        if phase == 1:
            self._fold_all(st)
            self._unpack_all(grads, st, "last")  # the last stage's and the final norm's gradients -> the parameters' shapes
            return
        # patch embedding: x0 = norm(conv(x) + b); no input gradient
        M0 = self.pe_rows.shape[0]
        g0 = self._v(spare.reshape(-1), M0, self.pe.np)
        self.pe_norm.bwd(dx, self.pe_out, self.pe_stats, None, g0, M0, st, colsum=self.fuse_ln)
        if not self.fuse_ln:
            self.colsum(g0, None, self.pe, M0, self.pe.np, st)
        self.pe.wgrad(g0, self.pe_rows, M0, st)
        self._fold_all(st)
        self._unpack_all(grads, st, "rest" if phase == 2 else None)  # padded float32 gradients -> the parameters' shapes, one launch
