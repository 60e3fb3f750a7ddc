"""TEST INFRASTRUCTURE -- CPU restatement of the DGL training step of BASELINE config 5's composition: ResNet18 audio
encoder + Swin visual encoder + ConcatFusion_DGL over 512 + C features.

The reference never instantiates this composition; it is assembled here exactly as the golden generator
assembles it from the reference's parts (tests/golden/make_golden.py::_SwinDGL): audio branch and pooling of
/root/reference/models/basic_model.py:73-75 (oracle.ResNet18 / avgpool, the C restatement), the Swin encoder of
models/swin_transformer.py:596-634 (oracle/swin_oracle.py, torch autograd on the CPU) with its per-frame features averaged
over the T frames of a sample, ConcatFusion_DGL (models/fusion_modules.py:45-59) and the step body of main_dgl.py:97-154
(three cross-entropies, backward of the unimodal losses, drop of the fusion-head gradients, backward of loss_f on detached
features, clip_grad_norm_(40), SGD(momentum 0.9, weight decay 1e-4)).  Pinned against tests/golden/dgl_swin_tiny_b4.npz in
tests/test_oracle_golden.py.  Only tests/ may import it.
"""
import numpy as np
import torch

from . import oracle as orc
from . import swin_oracle as so


class SwinAVModel:
    def __init__(self, params, buffers, swin_cfg):
        """params / buffers: ordered dicts of float32 arrays in the composition's state_dict naming
        (oracle.fixtures.swin_dgl_state); updated in place by train_step."""
        self.P, self.B, self.cfg = params, buffers, dict(swin_cfg)
        self.audio = orc.ResNet18(params, buffers, "audio_net.", "audio")
        self.mom = {}

    def _visual(self, image, drop=None):
        """-> (features [B, C] float32 numpy, closure dfeat -> {name: grad}).  drop: the training forward's DropPath scales
        ([blocks][2][B*T], swin_oracle.block) or None"""
        pre = "visual_net."
        Pt = {k[len(pre):]: torch.from_numpy(np.array(v)).clone().requires_grad_(True) for k, v in self.P.items()
              if k.startswith(pre)}
        B, T = image.shape[0], image.shape[2]
        y = so.forward(torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)), Pt, self.cfg,
                       None if drop is None else torch.from_numpy(np.asarray(drop, dtype=np.float32)))  # [B*T, C]
        fv = y.view(B, T, -1).mean(1)

        def backward(dfv):
            (fv * torch.from_numpy(np.ascontiguousarray(dfv, dtype=np.float32))).sum().backward()
            return {pre + k: v.grad.numpy() for k, v in Pt.items()}

        return fv.detach().numpy().astype(np.float32), backward

    def train_step(self, spec, image, label, alpha, lr, momentum=0.9, wd=1e-4, max_norm=40.0, drop=None):
        P = self.P
        B = image.shape[0]
        audio = np.ascontiguousarray(spec[:, None].astype(np.float32))
        a = self.audio.forward(audio, True)
        fa = orc.avgpool_fwd(a, B, 1)
        fv, vis_bwd = self._visual(image, drop)
        W, b = P["fusion_module.fc_out.weight"], P["fusion_module.fc_out.bias"]
        out_a, out_v, out = orc.concat_dgl_fwd(fa, fv, W, b)
        loss_v, g_v = orc.softmax_ce(out_v, label, alpha)  # main_dgl.py:102,108
        loss_a, g_a = orc.softmax_ce(out_a, label, alpha)  # :103,108
        loss_f, g_f = orc.softmax_ce(out, label, 1.0)      # :104
        dfa, dfv, _, _ = orc.concat_dgl_bwd(fa, fv, W, g_a, g_v, None)   # :110, head gradients dropped (:114-119)
        _, _, dW, db = orc.concat_dgl_bwd(fa, fv, W, None, None, g_f)    # :122
        G = {"fusion_module.fc_out.weight": dW, "fusion_module.fc_out.bias": db}
        G.update(self.audio.backward(orc.avgpool_bwd(dfa, a.shape, B, 1)))
        G.update(vis_bwd(dfv))
        total = float(np.sqrt(sum(orc.sumsq(g) for g in G.values())))
        coef = min(1.0, max_norm / (total + 1e-6))
        for k in G:
            G[k] = (G[k] * np.float32(coef)).astype(np.float32)
        r = {"out": out, "out_a": out_a, "out_v": out_v, "loss_f": loss_f, "loss_a": loss_a, "loss_v": loss_v,
             "total_norm": total,
             "audio_grad_sum": sum(orc.abs_mean(G[k]) for k in P if k.startswith("audio_net.")),
             "visual_grad_sum": sum(orc.abs_mean(G[k]) for k in P if k.startswith("visual_net.")),
             "grad_norm": {k: float(np.sqrt(orc.sumsq(g))) for k, g in G.items()}}
        for k in P:
            if k not in G:
                continue  # fc_auxi: grad None -> SGD skips it
            first = k not in self.mom
            if first:
                self.mom[k] = np.zeros_like(P[k])
            orc.sgd_(P[k], G[k], self.mom[k], lr, momentum, wd, first)
        return r
