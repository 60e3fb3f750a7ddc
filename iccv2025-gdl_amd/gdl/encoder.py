"""Python handle of the C++ ResNet18 encoder engine (csrc/encoder.cpp)."""
import ctypes

import torch

from . import _lib as L


class EncoderEngine:
    """One planned ResNet18 encoder for a fixed (modality, dtype, B, T, H, W).

    Owns its workspace (a torch uint8 tensor: PyTorch is the device allocator) and an opaque
    gdl_encoder_t.  forward()/backward() enqueue kernels on the CURRENT torch stream.
    """

    def __init__(self, modality, dtype, B, T, H, W, device):
        self.lib = L.load()
        self.modality = L.GDL_AUDIO if modality in ("audio", L.GDL_AUDIO) else L.GDL_VISUAL
        self.dtype = L.dtype_code(dtype)
        self.shape = (B, T, H, W)
        self.device = torch.device(device)
        h = ctypes.c_void_p()
        L.call("gdl_encoder_create", ctypes.byref(h), self.modality, self.dtype, B, T, H, W)
        self.h = h
        self.ws_bytes = self.lib.gdl_encoder_workspace_bytes(self.h)
        with torch.cuda.device(self.device):
            self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        assert self.ws.data_ptr() % 256 == 0
        L.call("gdl_encoder_bind", self.h, self.ws.data_ptr(), self.ws_bytes)
        numel = (ctypes.c_int64 * L.ENC_NPARAMS)()
        L.call("gdl_encoder_param_numel", self.h, numel)
        self.param_numel = list(numel)
        n, hh, ww = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.call("gdl_encoder_out_shape", self.h, ctypes.byref(n), ctypes.byref(hh), ctypes.byref(ww))
        self.out_shape = (n.value, 512, hh.value, ww.value)
        self._bound = None

    def side_stream(self, enable=True):
        """Fork the weight gradients of backward() onto an engine-owned side stream (see include/gdl_hip.h)."""
        if self.lane() is not None and (not enable or self.lane() != "owned"):
            L.call("gdl_encoder_side_stream", self.h, 0)
            self._lane = None
        if enable and self.lane() is None:
            L.call("gdl_encoder_side_stream", self.h, 1)
            self._lane = "owned"

    def borrow_side_stream(self, stream_handle):
        """The weight gradients of backward() on a stream of the caller's (gdl_encoder_borrow_side_stream; 0 = the null stream);
        None: back to no side lane.  An engine-owned side stream is replaced."""
        if stream_handle is None:
            if self.lane() is not None:
                L.call("gdl_encoder_side_stream", self.h, 0)
                self._lane = None
            return
        if self.lane() == "owned":
            L.call("gdl_encoder_side_stream", self.h, 0)
            self._lane = None
        if self.lane() != ("borrowed", stream_handle):
            L.call("gdl_encoder_borrow_side_stream", self.h, stream_handle if stream_handle else None)
            self._lane = ("borrowed", stream_handle)

    def lane(self):
        """None, "owned" or ("borrowed", stream handle): where backward() puts the weight gradients"""
        return getattr(self, "_lane", None)

    def has_side_stream(self):
        return self.lane() is not None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.gdl_encoder_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_params(self, params, running_mean, running_var, num_batches_tracked):
        """Lists of CUDA tensors in the reference's named_parameters()/BatchNorm order."""
        assert len(params) == L.ENC_NPARAMS and len(running_mean) == L.ENC_NBN
        key = tuple(p.data_ptr() for p in params) + tuple(t.data_ptr() for t in running_mean) + \
            tuple(t.data_ptr() for t in running_var) + tuple(t.data_ptr() for t in num_batches_tracked)
        if key == self._bound:
            return
        for p, n in zip(params, self.param_numel):
            if p.dtype != torch.float32 or not p.is_contiguous() or p.numel() != n or p.device != self.device:
                raise L.GdlError("encoder.set_params: parameters must be contiguous float32 tensors of the reference "
                                 f"shapes on {self.device}")
        P = (ctypes.c_void_p * L.ENC_NPARAMS)(*[p.data_ptr() for p in params])
        RM = (ctypes.c_void_p * L.ENC_NBN)(*[t.data_ptr() for t in running_mean])
        RV = (ctypes.c_void_p * L.ENC_NBN)(*[t.data_ptr() for t in running_var])
        NB = (ctypes.c_void_p * L.ENC_NBN)(*[t.data_ptr() for t in num_batches_tracked])
        L.call("gdl_encoder_set_params", self.h, P, RM, RV, NB)
        self._bound = key

    def forward(self, x, training, want_feat=True, want_fmap=False, feat_out=None):
        B, T, H, W = self.shape
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        cin = 1 if self.modality == L.GDL_AUDIO else 3
        if x.numel() != B * cin * T * H * W:
            raise L.GdlError(f"encoder.forward: input has {x.numel()} elements, engine was planned for "
                             f"B={B} Cin={cin} T={T} H={H} W={W}")
        feat = None
        if want_feat:
            feat = feat_out if feat_out is not None else torch.empty((B, 512), device=self.device)
        fmap = torch.empty(self.out_shape, device=self.device) if want_fmap else None
        L.call("gdl_encoder_forward", self.h, x.data_ptr(), 1 if training else 0, L.ptr(feat), L.ptr(fmap),
               L.cur_stream())
        self._keep = x  # the kernels read x asynchronously
        return feat, fmap

    @property
    def serial(self):
        return self.lib.gdl_encoder_forward_serial(self.h)

    def bn_overflow(self):
        """BatchNorms of the last training forward whose fixed-point statistics exceeded their headroom (0 = fine).  Synchronises."""
        n = self.lib.gdl_encoder_bn_overflow(self.h, L.cur_stream())
        if n < 0:
            raise L.GdlError(f"gdl_encoder_bn_overflow: error {-n}")
        return n

    def backward(self, grads, dfeat=None, dfmap=None, phase=0):
        """grads: 60 float32 CUDA tensors (overwritten).  phase 0 = everything; 1 = upstream gradient + layer4 (the last
        15 gradient tensors are then final: their all-reduce can start), 2 = the rest (no dfeat / dfmap)."""
        if dfeat is not None:
            dfeat = dfeat.float().contiguous()
        if dfmap is not None:
            dfmap = dfmap.float().contiguous()
        G = (ctypes.c_void_p * L.ENC_NPARAMS)(*[g.data_ptr() for g in grads])
        if phase == 0:
            L.call("gdl_encoder_backward", self.h, L.ptr(dfeat), L.ptr(dfmap), G, L.cur_stream())
        else:
            L.call("gdl_encoder_backward_phase", self.h, phase, L.ptr(dfeat), L.ptr(dfmap), G, L.cur_stream())
        if phase != 2:
            self._keep_b = (dfeat, dfmap)
