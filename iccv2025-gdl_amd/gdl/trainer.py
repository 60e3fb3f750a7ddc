"""Native DGL training step: the whole body of /root/reference/main_dgl.py:97-154 as one
sequence of gfx950 kernels on two HIP streams (+ a side stream for the visual weight gradients), without an autograd tape.

Per step (all asynchronous, nothing is read back unless `read()` is called).  Default form for the concat / sum DGL heads
(the Swin composition's 512 + 768 concat head included) without a process group ("early backward", bit-identical to the
junction form below):
  stream V: visual forward -> gdl_head_uni_dfeat (alpha * dCE(v_out)/d feature) -> visual backward
  stream A: audio  forward -> gdl_head_uni_dfeat (alpha * dCE(a_out)/d feature) -> audio  backward
            -> fusion head forward (all three logit sets), 3x cross-entropy, gradient of fc_out from loss_f alone
  then on A: fused grad statistics (total norm -> clip coefficient, per-encoder sum mean|g|)  (main_dgl.py:129-143)
             fused clip + SGD(momentum, weight decay) over the flat parameter arena            (:154)
Junction form (other heads, the non-DGL step, and -- until a multi-GPU run has validated the early form's collective order --
every data-parallel run):
  audio encoder forward  (stream A)  ||  visual encoder forward (stream V)
  fusion head forward, 3x cross-entropy, head backward with the DGL truncation   (stream A)
      - encoders receive only alpha * d(CE(a_out) + CE(v_out))      (main_dgl.py:108-110)
      - fc_out receives only d CE(out) on detached features          (:114-122)
      - fc_auxi never receives a gradient and is skipped by SGD      
  audio encoder backward (stream A)  ||  visual encoder backward (stream V)
      [+ per-bucket RCCL all-reduce as soon as a bucket is final]
  grad statistics, clip + SGD as above

The result is numerically the reference's two-phase backward: SURVEY section 0 shows the
single-pass form is bit-identical in exact arithmetic, and tests/test_step_gpu.py checks it
against golden vectors of the reference.
"""
import ctypes
import os

import torch

from . import _lib as L
from .encoder import EncoderEngine


class _SwinAdapter:
    """gdl.swin.SwinEngine behind the EncoderEngine calls the trainer makes (features averaged over the T frames of a
    sample; no BatchNorm state; phase 1 = upstream gradient + final norm + last stage, phase 2 = the rest)."""

    def __init__(self, cfg, dtype, B, T, device, net=None):
        from .swin import SwinEngine

        self.eng = SwinEngine(cfg, dtype, B, T, device)
        self.net = net  # the mirror module: drop_path_rate, drop_scales_override / last_drop_scales (stochastic depth)

    def set_params(self, params):
        self.eng.set_params(params)

    def forward(self, x, training, feat_out=None):
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise L.GdlError("DGLTrainer: frames must be a contiguous float32 [B, 3, T, H, W] tensor")
        drop = None
        if training and self.net is not None and self.net.drop_path_rate > 0:  # a training step draws the DropPath masks
            import sys

            drop_path_scales = sys.modules[type(self.net).__module__].drop_path_scales  # (models.swin_transformer, as imported by the caller)
            drop = self.net.drop_scales_override if self.net.drop_scales_override is not None else \
                drop_path_scales(self.eng.cfg, self.net.drop_path_rate, self.eng.N, self.eng.device)
            self.net.last_drop_scales = drop
        return self.eng.forward(x, pool_frames=True, out=feat_out, drop_scales=drop)

    def backward(self, grads, dfeat=None, phase=0):
        if phase != 2:
            self._dfeat = dfeat
        self.eng.backward(self._dfeat, list(grads), phase=phase)


_CHAIN_STREAMS = {}


def _chain_streams(device):
    """(audio-chain stream, visual-chain stream) of `device`, created once per process"""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _CHAIN_STREAMS:
        _CHAIN_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _CHAIN_STREAMS[key]


class DGLTrainer:
    def __init__(self, model, lr, alpha=4.0, momentum=0.9, weight_decay=1e-4, max_norm=40.0, mode="dgl", dtype=None,
                 process_group=None, comm_backend="torch", visual_side_stream=None, early_backward=None):
        """comm_backend: "torch" -- torch.distributed all_reduce on `process_group` (nccl = RCCL); "abi" -- the library's own
        RCCL communicator (gdl_comm_*), bootstrapped through `process_group`."""
        self.lib = L.load()
        self.model = model
        self.mode = mode
        self.lr, self.alpha, self.mu, self.wd, self.max_norm = float(lr), float(alpha), float(momentum), \
            float(weight_decay), float(max_norm)
        self.pg = process_group
        # visual_side_stream: None = the visual encoder's weight gradients get a stream of their own unless a process group is
        # given (the collective's stream is then the fourth); True / False force it.  Measured with a ONE-rank RCCL group on one
        # MI355X (bench.py, GDL_BENCH_FORCE_PG=1): 6.05 ms without, 5.80 ms with it (5.78 ms without a group) -- but a one-rank
        # all-reduce launches no kernel, so whether five streams hold up beside real RCCL traffic is for the first multi-GPU run
        # to tell (bench.py --side-stream on).
        self.visual_side_stream = visual_side_stream
        # early_backward: None = on where it applies (DGL step with the concat (512 + 512) or sum head): each encoder's feature
        # gradient comes from ITS auxiliary loss alone (main_dgl.py:110-122), so gdl_head_uni_dfeat computes it on the encoder's
        # own stream right behind its forward and the backward starts without waiting for the other encoder; the fusion
        # head (logits of all three sets, the losses, fc_out's gradient) follows on the audio stream behind the audio backward.
        # Same numbers bit for bit (tests/test_step_gpu.py::test_early_backward_identical), no forward -> head -> backward junction.
        # With a process group the early form issues the collectives in the order audio_l4, visual_l4, audio_rest, fusion,
        # visual_rest (over two streams; the same host code, hence the same order, on every rank).  Round 5: it is the default
        # there too -- the one-rank proxy reads 5.85 against 5.92 ms (round 4), with emulated collective traffic on a stream of
        # its own 6.18-6.19 against 6.30-6.31 ms (tools/pg_variants.py --emulate-traffic 16, profiles/r05_pg_variants.txt); the
        # two- and four-rank step tests (tests/test_ddp_gpu.py) run both forms against the oracle.  `bench.py --gpus N` still
        # times every variant on first contact with real RCCL traffic (comm.schedule_variants_ms); early_backward=False opts out.
        self.early_backward = early_backward
        if early_backward is None and os.environ.get("GDL_TUNING") == "1" and os.environ.get("GDL_EARLY_BWD") == "0":
            self.early_backward = False  # tuning aid (A/B)
        self.dtype = dtype if dtype is not None else model.audio_net.gdl_dtype
        head = model.fusion_module
        # head kind: concat (fc_out [n,1024]; ConcatFusion / ConcatFusion_DGL) or sum (fc_x, fc_y [n,512]; SumFusion_DGL)
        # or gated (fc_x, fc_y [512,512] + fc_out [n,512]; GatedFusion_DGL -- the step never gives fc_x / fc_y a
        # gradient (main_dgl.py:114-122 drops phase 1's, loss_f sees detached hidden vectors), so like fc_auxi they stay
        # outside the optimised arena)
        # or film (fc [512, 262144] + fc_out [n,512]; FiLM_DGL: all four tensors are trained by loss_f)
        self.head = ("gated" if hasattr(head, "fc_out") else "sum") if hasattr(head, "fc_x") else \
            ("film" if hasattr(head, "fc") else "concat")
        first = head.fc_x if self.head == "sum" else head.fc_out
        self.device = first.weight.device
        if self.device.type != "cuda":
            raise L.GdlError("DGLTrainer: the model must live on an MI355X (cuda) device; there is no CPU path")
        if self.head != "concat" and mode != "dgl":
            raise L.GdlError("DGLTrainer: the sum / gated / film heads are built for the DGL step only")
        if self.head == "gated" and not getattr(head, "x_gate", True):
            raise L.GdlError("DGLTrainer: GatedFusion_DGL is built for x_gate=True (basic_model.py:38)")
        self.n_classes = (head.fc_out if self.head == "gated" else first).weight.shape[0]
        # ---- flat arenas: [trained fusion-head tensors | audio_net (60) | visual_net (60)]
        # (ConcatFusion_DGL's fc_auxi never receives a gradient, SURVEY G1: it stays outside the arena)
        if self.head == "film":
            named = [("fusion_module.fc.weight", head.fc.weight), ("fusion_module.fc.bias", head.fc.bias),
                     ("fusion_module.fc_out.weight", head.fc_out.weight), ("fusion_module.fc_out.bias", head.fc_out.bias)]
        elif self.head == "sum":
            named = [("fusion_module.fc_x.weight", head.fc_x.weight), ("fusion_module.fc_x.bias", head.fc_x.bias),
                     ("fusion_module.fc_y.weight", head.fc_y.weight), ("fusion_module.fc_y.bias", head.fc_y.bias)]
        else:
            named = [("fusion_module.fc_out.weight", head.fc_out.weight), ("fusion_module.fc_out.bias", head.fc_out.bias)]
        nf = self.nf = len(named)
        named += [("audio_net." + n, p) for n, p in model.audio_net.named_parameters()]
        named += [("visual_net." + n, p) for n, p in model.visual_net.named_parameters()]
        # the visual branch: ResNet18 (60 tensors, 512 features) or the Swin composition of SURVEY row N4
        # (models.basic_model.AVClassifier_DGL_Swin: gdl.swin.SwinEngine, num_features wide)
        self.vis_swin = hasattr(model.visual_net, "cfg") and hasattr(model.visual_net, "num_features")
        self.nv = len(named) - nf - 60
        self.dv = int(model.visual_net.num_features) if self.vis_swin else 512
        if self.vis_swin and (self.head != "concat" or mode != "dgl"):
            raise L.GdlError("DGLTrainer: the Swin visual branch is built for the concat DGL head")
        if not self.vis_swin and self.nv != 60:
            raise L.GdlError("DGLTrainer: visual_net must be the ResNet18 mirror or the SwinTransformer mirror")
        self.names = [n for n, _ in named]
        offs, o = [0], 0
        for _, p in named:
            o += p.numel()
            offs.append(o)
        self.offsets = offs
        self.total = o
        group = [0] * nf + [1] * 60 + [2] * self.nv
        self.params = torch.empty(o, device=self.device)
        self.grads = torch.zeros(o, device=self.device)
        self.momentum = torch.zeros(o, device=self.device)
        self.pviews, self.gviews = [], []
        for i, (_, p) in enumerate(named):
            v = self.params[offs[i]:offs[i + 1]].view(p.shape)
            v.copy_(p.data)
            p.data = v  # the module now aliases the arena: state_dict / eval see the trained weights
            self.pviews.append(v)
            self.gviews.append(self.grads[offs[i]:offs[i + 1]].view(p.shape))
        # all-reduce buckets (ranges of the flat gradient arena).  layer4 = the last 15 tensors of an encoder, 8.4 M
        # of its 11.2 M parameters, is final right after the first two blocks of the backward: its own bucket lets
        # three quarters of the exchange overlap the rest of the backward.
        a0, v0 = nf, nf + 60
        # (Swin: the last stage + final norm -- the tensors whose gradients the backward finishes first, SwinEngine.backward
        # phase 1 -- play layer4's part)
        vsplit = v0 + 45 if not self.vis_swin else \
            v0 + next(i for i, (n, _) in enumerate(named[v0:]) if n.startswith("visual_net.layers.%d." % (model.visual_net.num_layers - 1)))
        self.bucket = {"fusion": (0, offs[nf]),
                       "audio_l4": (offs[a0 + 45], offs[a0 + 60]), "audio_rest": (offs[a0], offs[a0 + 45]),
                       "visual_l4": (offs[vsplit], offs[v0 + self.nv]), "visual_rest": (offs[v0], offs[vsplit])}
        self.reducer = None
        self.world = 1
        if process_group is not None:
            from .ddp import BucketReducer

            self.reducer = BucketReducer(self.grads, self.bucket, process_group, backend=comm_backend)
            self.world = self.reducer.world
            # Replica state follows rank 0 (parameters, momentum, BatchNorm running statistics / counters): a seed
            # that differs between ranks or a rank-0-only checkpoint load must not diverge silently.  fc_auxi (and the
            # gated head's fc_x / fc_y) live outside the arena: they are never updated but still part of the state.
            self.reducer.sync_state([self.params, self.momentum] + self._replica_buffers())
        h = ctypes.c_void_p()
        so = (ctypes.c_int64 * len(offs))(*offs)
        sg = (ctypes.c_int32 * len(group))(*group)
        L.call("gdl_optim_create", ctypes.byref(h), so, sg, len(group))
        self.opt = h
        self.opt_ws_bytes = self.lib.gdl_optim_workspace_bytes(h)
        self.opt_ws = torch.empty(max(self.opt_ws_bytes, 8), dtype=torch.uint8, device=self.device)
        # (explicit: a caching allocator may hand out a block at the address of an earlier trainer's workspace)
        L.call("gdl_optim_bind_workspace", self.opt, L.ptr(self.opt_ws), self.opt_ws_bytes, L.cur_stream())
        self.stats = torch.zeros(self.lib.gdl_optim_stats_len(h), device=self.device)
        self.losses = torch.zeros(3, device=self.device)  # loss_f, loss_a, loss_v
        # The two chain streams are shared by every trainer of a process on this device: torch hands out a NEW pool stream per
        # torch.cuda.Stream() call and never retires one, and once more distinct streams have carried work than the runtime has
        # hardware queues (four), two chains can end up time-slicing one queue -- the third trainer built in a process ran its step
        # 15 % slower than the same trainer in a fresh process (bench.py's `extra_workloads.ks`: 7.33 vs 6.36 ms, round 4).
        self.s_a, self.s_v = _chain_streams(self.device)
        self.eng_a = self.eng_v = None
        self.steps = 0
        self.phase_events = None  # set to [] to record (name, event) marks on the main stream per step
        # measuring tap (bench.py): a [n, 2] device tensor + a position; while the position is not None every step copies its
        # (total norm, clip coefficient) into the next row, device to device on the step's stream
        self.stats_log = None
        self.stats_log_pos = None

    def _replica_buffers(self):
        """Every tensor of the replica that is not in the flat arenas: BatchNorm running statistics and counters of
        both encoders, and the fusion-head parameters the step never trains."""
        m = self.model
        in_arena = {v.data_ptr() for v in getattr(self, "pviews", [])}
        extra = [p.data for p in m.fusion_module.parameters() if p.data.data_ptr() not in in_arena]
        return [b for net in (m.audio_net, m.visual_net) for b in net.buffers()] + extra

    def sync_replicas(self):
        """Broadcast rank 0's BatchNorm buffers (before eval / checkpoint: 'replica 0 persists', main_dgl.py:244 --
        BatchNorm statistics are per rank during training, as with the reference's nn.DataParallel replicas)."""
        if self.reducer is not None:
            self.reducer.broadcast_buffers(self._replica_buffers())

    def state_dict(self):
        """Optimizer-side state a reference checkpoint keeps besides model.state_dict() (optimizer.state_dict() /
        scheduler, main_dgl.py:372): momentum arena, learning rate, step count.  In data-parallel mode the BatchNorm
        buffers are first made rank 0's, so model.state_dict() taken next is the reference's replica-0 state."""
        self.sync_replicas()
        return {"momentum": self.momentum.detach().clone(), "lr": self.lr, "steps": self.steps, "names": list(self.names),
                "offsets": list(self.offsets), "mu": self.mu, "weight_decay": self.wd}

    def load_state_dict(self, sd):
        if list(sd["offsets"]) != list(self.offsets) or list(sd["names"]) != list(self.names):
            raise L.GdlError("DGLTrainer.load_state_dict: the checkpoint's parameter layout differs from this model's")
        self.momentum.copy_(sd["momentum"].to(self.device))
        self.lr, self.steps = float(sd["lr"]), int(sd["steps"])
        if self.reducer is not None:  # the model's parameters alias the arena: whatever rank 0 loaded is the truth
            self.reducer.sync_state([self.params, self.momentum] + self._replica_buffers())

    def close(self):
        """Releases what the trainer owns outside PyTorch's allocator: the optimizer descriptor and, for
        comm_backend="abi", the RCCL communicator (ncclCommDestroy).  Idempotent; also run by __del__."""
        if getattr(self, "reducer", None) is not None:
            self.reducer.close()
        if getattr(self, "opt", None):
            self.lib.gdl_optim_destroy(self.opt)
            self.opt = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ setup per batch shape
    def _prepare(self, spec, image):
        B, F_, T_ = spec.shape
        Bv, C, T, H, W = image.shape
        if Bv != B or C != 3:
            raise L.GdlError("DGLTrainer.step: spec must be [B,F,T'] and image [B,3,T,H,W]")
        key = (B, F_, T_, T, H, W)
        if getattr(self, "_key", None) == key:
            return
        self._key = key
        self.eng_a = EncoderEngine("audio", self.dtype, B, 1, F_, T_, self.device)
        if self.vis_swin:
            cfg = self.model.visual_net.cfg
            if H != cfg["img"] or W != cfg["img"]:
                raise L.GdlError(f"DGLTrainer.step: the Swin branch was built for {cfg['img']} x {cfg['img']} frames")
            self.eng_v = _SwinAdapter(cfg, self.dtype, B, T, self.device, self.model.visual_net)
        else:
            self.eng_v = EncoderEngine("visual", self.dtype, B, T, H, W, self.device)
        # A fourth stream for the visual (critical-path) encoder's weight gradients -- but never a fifth:
        # with a process group the collective's stream is the fourth (see gdl_encoder_side_stream).
        # (tuning aid, honoured only with GDL_TUNING=1: GDL_SIDE_STREAM 0 = off, 1 = both engines)
        side = os.environ.get("GDL_SIDE_STREAM") if os.environ.get("GDL_TUNING") == "1" else None
        if side == "1":
            self.eng_a.side_stream(True)
        # The audio engine's weight gradients ride on the stream step() is called on: that stream only orders the step before and
        # behind, so its hardware queue idles for the whole step -- a fourth lane without a fifth queue (round 4: 5.63 -> 5.43 ms;
        # the audio chain, a quarter of the arithmetic in ~100 small launches, had been the last to finish: 5.36 ms against the
        # visual chain's 5.18, tools/chain_timeline.py).  Bound per step (the caller's current stream may change): step().
        # Not with a process group by default: there the backward runs in two phases without the early start and the visual
        # engine keeps its weight gradients on its chain, and the one-rank proxy (GDL_BENCH_FORCE_PG=1) reads 5.87 ms with the
        # borrowed lane against 5.78 without (5.61 with the visual side stream alone) -- `bench.py --gpus N` times the variants on
        # first contact with real RCCL traffic (`comm.schedule_variants_ms`); `trainer.audio_on_caller` may be set at any time.
        # (tuning aid: GDL_SIDE_STREAM=2 = the round-3 layout, the visual engine's side stream alone; 3 = the borrowed lane forced)
        self.audio_on_caller = side == "3" or (side not in ("0", "1", "2", "4") and self.reducer is None)
        # The borrowed lane for the VISUAL engine instead (tuning aid "4"; a variant `bench.py --gpus N` times): without a group it
        # equals an owned side stream (5.57 ms), with a one-rank group it reads 6.57 ms -- not a default anywhere.
        self.visual_on_caller = side == "4"
        assert not (self.audio_on_caller and self.visual_on_caller), "the caller's stream carries ONE engine's weight gradients"
        want_v = self.visual_side_stream if self.visual_side_stream is not None else self.reducer is None
        if (side in ("1", "2") or (side in (None, "3") and want_v)) and not self.vis_swin:
            self.eng_v.side_stream(True)
        n, d = self.n_classes, self.device
        self.fa, self.fv = torch.empty((B, 512), device=d), torch.empty((B, self.dv), device=d)
        self.dfa, self.dfv = torch.empty((B, 512), device=d), torch.empty((B, self.dv), device=d)
        self.dscr_a, self.dscr_v = torch.empty((B, 512), device=d), torch.empty((B, self.dv), device=d)
        if self.head == "film":
            if B > 512:
                raise L.GdlError("DGLTrainer: the FiLM head handles at most 512 samples per step")
            self.hidden = torch.empty((3, B, 512), device=d)
            self.head_ws = torch.empty(self.lib.gdl_head_film_workspace_bytes(B), dtype=torch.uint8, device=d)
        if self.head == "gated":  # hidden vectors (saved for the backward) + its scratch
            self.hx, self.hy = torch.empty((B, 512), device=d), torch.empty((B, 512), device=d)
            self.head_ws = torch.empty(2 * B * 512, device=d)
        self.out, self.out_a, self.out_v = (torch.empty((B, n), device=d) for _ in range(3))
        self.g_f, self.g_a, self.g_v = (torch.empty((B, n), device=d) for _ in range(3))
        self.B = B

    def _bind(self):
        m = self.model
        if self.vis_swin:
            self.eng_v.set_params([p.data for p in m.visual_net.parameters()])
        for eng, net in ((self.eng_a, m.audio_net),) + (() if self.vis_swin else ((self.eng_v, m.visual_net),)):
            bns = net._bn_layers()
            eng.set_params([p.data for p in net.parameters()], [b.running_mean for b in bns],
                           [b.running_var for b in bns], [b.num_batches_tracked for b in bns])

    def _check_label(self, label):
        """The loss kernels index logits by label: require the reference's dtype and shape (a class index outside
        [0, n) raises a device assert in the reference; here the loss kernel skips the sample and poisons the loss)."""
        if label.dtype != torch.int64 or label.dim() != 1 or label.shape[0] != self.B or label.device != self.device:
            raise L.GdlError(f"DGLTrainer: label must be an int64 [B={self.B}] tensor on {self.device}, got "
                             f"{label.dtype} {tuple(label.shape)} on {label.device}")

    # ------------------------------------------------------------------ the step
    def step(self, spec, image, label):
        """spec [B,F,T'] float, image [B,3,T,H,W] float, label [B] int64 -- all resident on the device."""
        self._prepare(spec, image)
        self._check_label(label)
        self._bind()
        # The head, the losses and the optimizer run on the audio chain's stream rather than on the caller's: one stream
        # (hardware queue) less in play measured +1 % (tools: 9 660 -> 9 760 samples/s).  The caller's stream is ordered
        # before the step and behind it, so the call keeps ordinary stream semantics.
        caller = torch.cuda.current_stream(self.device)
        main = self.s_a
        # the borrowed lane (the stream this call came in on): the audio engine's weight gradients, or the visual engine's
        if self.audio_on_caller:
            self.eng_a.borrow_side_stream(caller.cuda_stream)
        elif isinstance(self.eng_a.lane(), tuple):
            self.eng_a.borrow_side_stream(None)
        if not self.vis_swin:
            if self.visual_on_caller and not self.audio_on_caller:
                self.eng_v.borrow_side_stream(caller.cuda_stream)
            elif isinstance(self.eng_v.lane(), tuple):
                self.eng_v.borrow_side_stream(None)
        main.wait_stream(caller)
        with torch.cuda.stream(main):
            self._step_on(main, spec, image, label)
        caller.wait_stream(main)

    def _step_on(self, main, spec, image, label):
        audio = spec.unsqueeze(1)  # main_dgl.py:100
        label = label.contiguous()
        B, n = self.B, self.n_classes
        nf = self.nf
        self._mark(main, "start")
        ev = main.record_event()
        self.s_a.wait_event(ev)
        self.s_v.wait_event(ev)
        # the visual encoder is the critical path (3x the audio work): enqueue it first so the single
        # host thread's ~100 launches per encoder pass do not delay it
        dgl = self.mode == "dgl"
        early = (dgl and self.head in ("concat", "sum") and n <= 512 and self.early_backward is not False
                 and (self.dv == 512 or (self.head == "concat" and self.dv in (768, 1024))))
        red = self.reducer
        gv, ga = self.gviews[nf + 60:nf + 60 + self.nv], self.gviews[nf:nf + 60]
        if early:
            pv = self.pviews
            if self.head == "concat":  # fc_out [n][512 + dv] + one bias
                wa, wv, ldw, ba, bv = L.ptr(pv[0]), pv[0].data_ptr() + 512 * 4, 512 + self.dv, L.ptr(pv[1]), L.ptr(pv[1])
            else:  # fc_x, fc_y [n][512] with their biases
                wa, wv, ldw, ba, bv = L.ptr(pv[0]), L.ptr(pv[2]), 512, L.ptr(pv[1]), L.ptr(pv[3])
            with torch.cuda.stream(self.s_v):
                self.eng_v.forward(image, True, feat_out=self.fv)
                L.call("gdl_head_uni_dfeat_w", L.ptr(self.fv), wv, ldw, bv, L.ptr(label), self.alpha, L.ptr(self.dfv), B, n,
                       self.dv, self.s_v.cuda_stream)
                ev_v = self.s_v.record_event()
            with torch.cuda.stream(self.s_a):
                self.eng_a.forward(audio, True, feat_out=self.fa)
                L.call("gdl_head_uni_dfeat", L.ptr(self.fa), wa, ldw, ba, L.ptr(label), self.alpha, L.ptr(self.dfa), B, n,
                       self.s_a.cuda_stream)
            # (host order: both forwards are enqueued before either backward, so neither chain waits for the host)
            with torch.cuda.stream(self.s_v):
                if red is None:
                    self.eng_v.backward(gv, dfeat=self.dfv)
                else:
                    self.eng_v.backward(gv, dfeat=self.dfv, phase=1)
            with torch.cuda.stream(self.s_a):
                if red is None:
                    self.eng_a.backward(ga, dfeat=self.dfa)
                else:
                    # collectives of one communicator run in issue order, identical on every rank
                    self.eng_a.backward(ga, dfeat=self.dfa, phase=1)
                    red.launch("audio_l4")
            if red is not None:
                with torch.cuda.stream(self.s_v):
                    red.launch("visual_l4")
                    self.eng_v.backward(gv, phase=2)
                with torch.cuda.stream(self.s_a):
                    self.eng_a.backward(ga, phase=2)
                    red.launch("audio_rest")
            # the fusion head on the audio stream (= main), behind the audio backward: all three logit sets, the losses, the
            # gradient of fc_out (/ fc_x, fc_y) from loss_f alone; its feature gradients go to scratch (the encoders have theirs)
            main.wait_event(ev_v)
            st = main.cuda_stream
            self._head_forward(True, st)
            L.call("gdl_softmax_ce3", L.ptr(self.out), L.ptr(self.out_a), L.ptr(self.out_v), L.ptr(label), 1.0, self.alpha,
                   self.alpha, self.losses.data_ptr(), L.ptr(self.g_f), L.ptr(self.g_a), L.ptr(self.g_v), B, n, st)
            gvw = self.gviews
            if self.head == "sum":
                L.call("gdl_head_sum_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[2]), L.ptr(self.g_a),
                       L.ptr(self.g_v), L.ptr(self.g_f), 0, 0, L.ptr(self.dscr_a), L.ptr(self.dscr_v), L.ptr(gvw[0]), L.ptr(gvw[1]),
                       L.ptr(gvw[2]), L.ptr(gvw[3]), B, n, st)
            elif self.dv != 512:
                L.call("gdl_head_concat_xy_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(self.g_a), L.ptr(self.g_v),
                       L.ptr(self.g_f), 0, 0, L.ptr(self.dscr_a), L.ptr(self.dscr_v), L.ptr(gvw[0]), L.ptr(gvw[1]), B, n, 512,
                       self.dv, st)
            else:
                L.call("gdl_head_concat_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(self.g_a), L.ptr(self.g_v),
                       L.ptr(self.g_f), 0, 0, L.ptr(self.dscr_a), L.ptr(self.dscr_v), L.ptr(gvw[0]), L.ptr(gvw[1]), B, n, st)
            if red is not None:
                red.launch("fusion")
                with torch.cuda.stream(self.s_v):
                    red.launch("visual_rest")
            self._finish_step(main, st)
            return
        with torch.cuda.stream(self.s_v):
            self.eng_v.forward(image, True, feat_out=self.fv)
        with torch.cuda.stream(self.s_a):
            self.eng_a.forward(audio, True, feat_out=self.fa)
        main.wait_stream(self.s_a)
        main.wait_stream(self.s_v)
        self._mark(main, "fwd_done")
        st = main.cuda_stream
        self._head_forward(dgl, st)
        lp = self.losses.data_ptr()
        if dgl:  # loss_f, alpha*loss_a, alpha*loss_v (main_dgl.py:102-108) in one launch
            L.call("gdl_softmax_ce3", L.ptr(self.out), L.ptr(self.out_a), L.ptr(self.out_v), L.ptr(label), 1.0, self.alpha,
                   self.alpha, lp, L.ptr(self.g_f), L.ptr(self.g_a), L.ptr(self.g_v), B, n, st)
        else:
            L.call("gdl_softmax_ce", L.ptr(self.out), L.ptr(label), 1.0, lp, L.ptr(self.g_f), B, n, st)
        if dgl:
            # DGL truncation: `out` is computed from detached features (flag 0) and the head gradients of the
            # unimodal losses are dropped before loss_f.backward() (flag 0)   (main_dgl.py:110-122)
            if self.head == "film":
                pv, gv = self.pviews, self.gviews
                L.call("gdl_head_film_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[2]), L.ptr(self.hidden),
                       L.ptr(self.g_a), L.ptr(self.g_v), L.ptr(self.g_f), 0, L.ptr(self.dfa), L.ptr(self.dfv), L.ptr(gv[0]),
                       L.ptr(gv[1]), L.ptr(gv[2]), L.ptr(gv[3]), B, n, L.ptr(self.head_ws), self.head_ws.numel(), st)
            elif self.head == "gated":
                fm = self.model.fusion_module
                L.call("gdl_head_gated_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(self.hx), L.ptr(self.hy),
                       L.ptr(fm.fc_x.weight), L.ptr(fm.fc_y.weight), L.ptr(self.pviews[0]), L.ptr(self.g_a), L.ptr(self.g_v),
                       L.ptr(self.g_f), 0, L.ptr(self.dfa), L.ptr(self.dfv), None, None, None, None, L.ptr(self.gviews[0]),
                       L.ptr(self.gviews[1]), L.ptr(self.head_ws), B, n, st)
            elif self.head == "sum":
                pv, gv = self.pviews, self.gviews
                L.call("gdl_head_sum_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[2]), L.ptr(self.g_a),
                       L.ptr(self.g_v), L.ptr(self.g_f), 0, 0, L.ptr(self.dfa), L.ptr(self.dfv), L.ptr(gv[0]), L.ptr(gv[1]),
                       L.ptr(gv[2]), L.ptr(gv[3]), B, n, st)
            elif self.dv != 512:  # 512 + num_features wide fc_out (the Swin composition)
                L.call("gdl_head_concat_xy_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(self.pviews[0]), L.ptr(self.g_a),
                       L.ptr(self.g_v), L.ptr(self.g_f), 0, 0, L.ptr(self.dfa), L.ptr(self.dfv), L.ptr(self.gviews[0]),
                       L.ptr(self.gviews[1]), B, n, 512, self.dv, st)
            else:
                L.call("gdl_head_concat_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(self.pviews[0]), L.ptr(self.g_a),
                       L.ptr(self.g_v), L.ptr(self.g_f), 0, 0, L.ptr(self.dfa), L.ptr(self.dfv), L.ptr(self.gviews[0]),
                       L.ptr(self.gviews[1]), B, n, st)
        else:  # BASELINE config 1: ConcatFusion + one CE loss (main.py:161-175)
            L.call("gdl_head_concat_bwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(self.pviews[0]), None, None, L.ptr(self.g_f),
                   1, 0, L.ptr(self.dfa), L.ptr(self.dfv), L.ptr(self.gviews[0]), L.ptr(self.gviews[1]), B, n, st)
        self._mark(main, "head_done")
        red = self.reducer
        if red is not None:
            red.launch("fusion")
        ev2 = main.record_event()
        self.s_a.wait_event(ev2)
        self.s_v.wait_event(ev2)
        gv, ga = self.gviews[nf + 60:nf + 60 + self.nv], self.gviews[nf:nf + 60]
        if red is None:
            with torch.cuda.stream(self.s_v):
                self.eng_v.backward(gv, dfeat=self.dfv)
            with torch.cuda.stream(self.s_a):
                self.eng_a.backward(ga, dfeat=self.dfa)
        else:
            # Data parallel: each encoder's backward in two phases so that the layer4 bucket (75 % of the bytes) is
            # exchanged while layer3 .. stem are still being differentiated.  Collectives of one communicator run in
            # issue order, identical on every rank: audio before visual (the audio passes are the shorter ones).
            with torch.cuda.stream(self.s_v):
                self.eng_v.backward(gv, dfeat=self.dfv, phase=1)
            with torch.cuda.stream(self.s_a):
                self.eng_a.backward(ga, dfeat=self.dfa, phase=1)
                red.launch("audio_l4")
            with torch.cuda.stream(self.s_v):
                red.launch("visual_l4")
                self.eng_v.backward(gv, phase=2)
            with torch.cuda.stream(self.s_a):
                self.eng_a.backward(ga, phase=2)
                red.launch("audio_rest")
            with torch.cuda.stream(self.s_v):
                red.launch("visual_rest")
        self._finish_step(main, st)

    def _finish_step(self, main, st):
        """Joins the chains (and the collectives), then gradient statistics + clip + SGD on `main`."""
        red = self.reducer
        main.wait_stream(self.s_a)
        main.wait_stream(self.s_v)
        if red is not None:
            red.wait_all()
        self._mark(main, "bwd_done")
        gs = 1.0 / self.world
        L.call("gdl_optim_grad_stats", self.opt, L.ptr(self.grads), self.max_norm, gs, L.ptr(self.stats),
               L.ptr(self.opt_ws), self.opt_ws_bytes, st)
        L.call("gdl_optim_sgd_step", self.opt, L.ptr(self.params), L.ptr(self.grads), L.ptr(self.momentum),
               L.ptr(self.stats), gs, self.lr, self.mu, self.wd, st)
        self._mark(main, "end")
        if self.stats_log is not None and self.stats_log_pos is not None and self.stats_log_pos < self.stats_log.shape[0]:
            self.stats_log[self.stats_log_pos].copy_(self.stats[:2], non_blocking=True)
            self.stats_log_pos += 1
        self.steps += 1

    def _head_forward(self, dgl, st):
        """(out, out_a, out_v) from the pooled features self.fa / self.fv."""
        pv, B, n = self.pviews, self.B, self.n_classes
        oa, ov = (L.ptr(self.out_a), L.ptr(self.out_v)) if dgl else (None, None)
        if self.head == "film":  # fusion_modules.py:140-178
            L.call("gdl_head_film_fwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[1]), L.ptr(pv[2]), L.ptr(pv[3]),
                   L.ptr(self.hidden), L.ptr(self.out), oa, ov, B, n, L.ptr(self.head_ws), self.head_ws.numel(), st)
        elif self.head == "gated":  # fusion_modules.py:232-250
            fm = self.model.fusion_module
            L.call("gdl_head_gated_fwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(fm.fc_x.weight), L.ptr(fm.fc_x.bias),
                   L.ptr(fm.fc_y.weight), L.ptr(fm.fc_y.bias), L.ptr(pv[0]), L.ptr(pv[1]), L.ptr(self.hx), L.ptr(self.hy),
                   L.ptr(self.out), oa, ov, B, n, st)
        elif self.head == "sum":  # fusion_modules.py:22-30
            L.call("gdl_head_sum_fwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[1]), L.ptr(pv[2]), L.ptr(pv[3]),
                   L.ptr(self.out), oa, ov, B, n, st)
        elif self.dv != 512:
            L.call("gdl_head_concat_xy_fwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[1]), L.ptr(self.out), oa, ov,
                   B, n, 512, self.dv, st)
        else:  # fusion_modules.py:38-42 / 51-59
            L.call("gdl_head_concat_fwd", L.ptr(self.fa), L.ptr(self.fv), L.ptr(pv[0]), L.ptr(pv[1]), L.ptr(self.out), oa, ov,
                   B, n, st)

    def _mark(self, stream, name):
        if self.phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            self.phase_events.append((name, e))

    # ------------------------------------------------------------------ validation (main_dgl.py:168-222)
    def valid(self, batches):
        """The reference's valid(): eval-mode forward (BatchNorm running statistics) of every (spec, image, label)
        batch, arg-max of the three logit sets and per-class counters -- all on the device, one host copy at the
        end instead of 3*B per batch.  Returns (acc, acc_a, acc_v) = sum(acc*)/sum(num) as main_dgl.py:222."""
        n = self.n_classes
        cnt = torch.zeros((4, n), dtype=torch.int64, device=self.device)
        main = torch.cuda.current_stream(self.device)
        self.sync_replicas()  # data-parallel: evaluate with rank 0's running statistics (SURVEY 8(e) "Buffers")
        for spec, image, label in batches:
            self._prepare(spec, image)
            self._check_label(label)
            self._bind()
            audio = spec.unsqueeze(1)
            label = label.contiguous()
            ev = main.record_event()
            self.s_a.wait_event(ev)
            self.s_v.wait_event(ev)
            with torch.cuda.stream(self.s_v):
                self.eng_v.forward(image, False, feat_out=self.fv)
            with torch.cuda.stream(self.s_a):
                self.eng_a.forward(audio, False, feat_out=self.fa)
            main.wait_stream(self.s_a)
            main.wait_stream(self.s_v)
            st = main.cuda_stream
            dgl = self.mode == "dgl"
            self._head_forward(dgl, st)
            L.call("gdl_eval_count", L.ptr(self.out), L.ptr(self.out_a) if dgl else None, L.ptr(self.out_v) if dgl else None,
                   L.ptr(label), self.B, n, cnt[0].data_ptr(), cnt[1].data_ptr(), cnt[2].data_ptr() if dgl else None,
                   cnt[3].data_ptr() if dgl else None, st)
        c = cnt.cpu().numpy().astype("float64")
        self.valid_counts = c
        tot = max(c[0].sum(), 1.0)
        return c[1].sum() / tot, c[2].sum() / tot, c[3].sum() / tot

    # ------------------------------------------------------------------ results (host sync)
    def read(self):
        """Synchronises and returns the quantities the reference prints / logs per step."""
        torch.cuda.synchronize(self.device)
        s = self.stats.cpu().numpy()
        ls = self.losses.cpu().numpy()
        nseg = len(self.names)
        r = {"loss_f": float(ls[0]), "loss_a": float(ls[1]), "loss_v": float(ls[2]), "total_norm": float(s[0]),
             "clip_coef": float(s[1]), "audio_grad_sum": float(s[2]), "visual_grad_sum": float(s[3]),
             "grad_norm": dict(zip(self.names, s[4:4 + nseg].tolist())),
             "grad_absmean": dict(zip(self.names, s[4 + nseg:4 + 2 * nseg].tolist())),
             "out": self.out.cpu().numpy()}
        if self.mode == "dgl":
            r["out_a"] = self.out_a.cpu().numpy()
            r["out_v"] = self.out_v.cpu().numpy()
        # a diverged BatchNorm (statistics beyond the fixed-point headroom, csrc/bnacc.h) must be as loud as the reference's
        # inf / NaN: ReLU turns the NaN statistics' outputs into zeros, so the logits alone may look sane
        bad = sum(e.bn_overflow() for e in (self.eng_a, self.eng_v) if e is not None and hasattr(e, "bn_overflow"))
        if bad:
            raise FloatingPointError(f"gdl: the statistics of {bad} BatchNorm layer(s) overflowed in the last training forward "
                                     "(activations of mean magnitude beyond 8192: the run has diverged)")
        return r

    def grad(self, name):
        return self.gviews[self.names.index(name)]
