"""TEST INFRASTRUCTURE -- the DGL training step of BASELINE config 5's composition (ResNet18 audio + Swin visual +
ConcatFusion_DGL over 512 + C) restated with PyTorch CPU operators in a chosen precision -- float64 by default: the ARBITER of
the config-5 parity tests at the benchmark's own size (B = 64, T = 3: 192 frames, 602 112 stage-1 tokens), where the plain
float32 oracles are too slow or too noisy (the bench-size Swin tests compared HIP with HIP).

Written from scratch (none of the reference's files): the audio branch is oracle/torch_step.py's ResNet18 restatement
(/root/reference/models/backbone.py:75-201), the visual branch oracle/swin_oracle.py's index-list restatement of
/root/reference/models/swin_transformer.py:596-634 with the per-frame features averaged over a sample's T frames, the head and
the step body are /root/reference/models/fusion_modules.py:45-59 and /root/reference/main_dgl.py:97-154 (three
cross-entropies, backward of the unimodal losses with the graph retained, the fusion head's gradients dropped, backward of
loss_f on detached features, clip_grad_norm_(40), SGD with momentum 0.9 and weight decay 1e-4).  The Swin encoder has no batch
statistics (LayerNorm is per token), so its passes run over CHUNKS of samples: a no-grad forward gives the features, the head
and the losses give the feature gradient, and every chunk is then recomputed with autograd and back-propagated with its slice of
that gradient -- parameter gradients add up over the chunks, peak memory is one chunk's.

Pinned in tests/test_oracle_golden.py against the goldens captured from the imported reference (`swin_t_b1.npz` through
`swin_features_and_grads`, `dgl_swin_tiny_b4.npz` through `TorchSwinStep.train_step`).  Only tests/ may import it.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import swin_oracle as so
from .torch_step import encoder as resnet_encoder


def swin_features_and_grads(x, params, cfg, dy, dtype=torch.float64, chunk=16):
    """The Swin encoder alone at any frame count: x [B, 3, T, img, img], dy [B*T, C] -> (y [B*T, C], {name: grad}) for the loss
    sum(y * dy), computed in `dtype` over chunks of `chunk` frames (numpy float64 out)."""
    P = {k: torch.from_numpy(np.array(v)).to(dtype).requires_grad_(True) for k, v in params.items()}
    x = torch.from_numpy(np.asarray(x))
    B, Cin, T, H, W = x.shape
    frames = x.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, 1, H, W)  # every frame a one-frame sample
    dyt = torch.from_numpy(np.asarray(dy)).to(dtype)
    ys = []
    for f0 in range(0, B * T, chunk):
        xc = frames[f0:f0 + chunk].to(dtype)
        y = so.forward(xc, P, cfg)
        (y * dyt[f0:f0 + chunk]).sum().backward()
        ys.append(y.detach())
    return torch.cat(ys).numpy(), {k: v.grad.numpy() for k, v in P.items()}


class TorchSwinStep:
    def __init__(self, params, buffers, swin_cfg, dtype=torch.float64, chunk_samples=4, threads=None):
        """params / buffers: name -> numpy array in the composition's state_dict naming (oracle.fixtures.swin_dgl_state)."""
        if threads:
            torch.set_num_threads(int(threads))
        self.cfg, self.dtype, self.chunk = dict(swin_cfg), dtype, int(chunk_samples)
        self.P = {k: torch.from_numpy(np.array(v)).to(dtype).requires_grad_(True) for k, v in params.items()}
        self.Bf = {k: (torch.from_numpy(np.array(v)).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.from_numpy(np.array(v)))
                   for k, v in buffers.items()}
        self.mom = {}

    def _swin_params(self):
        pre = "visual_net."
        return {k[len(pre):]: v for k, v in self.P.items() if k.startswith(pre)}

    def train_step(self, spec, image, label, alpha, lr, momentum=0.9, wd=1e-4, max_norm=40.0):
        P, Bf, dt = self.P, self.Bf, self.dtype
        spec = torch.as_tensor(np.asarray(spec)).to(dt)
        image = torch.as_tensor(np.asarray(image))
        label = torch.as_tensor(np.asarray(label)).long()
        B, T = image.shape[0], image.shape[2]
        for p in P.values():
            p.grad = None
        # audio branch: one autograd graph over the whole batch (BatchNorm statistics are the batch's)
        a = resnet_encoder(spec.unsqueeze(1), P, Bf, "audio_net", True)
        fa = torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)
        # visual branch, pass 1: features without a graph, chunk by chunk
        Pv = self._swin_params()
        with torch.no_grad():
            fv = torch.cat([so.forward(image[b0:b0 + self.chunk].to(dt), Pv, self.cfg).view(-1, T, self.feat_dim()).mean(1)
                            for b0 in range(0, B, self.chunk)])
        fv = fv.detach().requires_grad_(True)
        W_, b_ = P["fusion_module.fc_out.weight"], P["fusion_module.fc_out.bias"]
        out = F.linear(torch.cat((fa, fv), 1).detach(), W_, b_)                 # fusion_modules.py:53-56
        out_a = F.linear(torch.cat((fa, torch.zeros_like(fv)), 1), W_, b_)       # :57
        out_v = F.linear(torch.cat((torch.zeros_like(fa), fv), 1), W_, b_)       # :58
        loss_v, loss_a, loss_f = F.cross_entropy(out_v, label), F.cross_entropy(out_a, label), F.cross_entropy(out, label)
        ((loss_a + loss_v) * alpha).backward(retain_graph=True)                  # main_dgl.py:108-110
        for k, p in P.items():
            if k.startswith("fusion_module."):
                p.grad = None                                                    # :114-119
        loss_f.backward()                                                        # :122
        dfv = fv.grad.detach()
        # visual branch, pass 2: every chunk again, with a graph, back-propagated with its slice of the feature gradient
        for b0 in range(0, B, self.chunk):
            fvc = so.forward(image[b0:b0 + self.chunk].to(dt), Pv, self.cfg).view(-1, T, self.feat_dim()).mean(1)
            (fvc * dfv[b0:b0 + self.chunk]).sum().backward()
        with_grad = [p for p in P.values() if p.grad is not None]
        total = float(torch.nn.utils.clip_grad_norm_(with_grad, max_norm))       # :129
        res = {"out": out.detach().numpy(), "out_a": out_a.detach().numpy(), "out_v": out_v.detach().numpy(),
               "loss_f": loss_f.item(), "loss_a": loss_a.item(), "loss_v": loss_v.item(), "total_norm": total,
               "audio_grad_sum": float(sum(p.grad.abs().mean() for k, p in P.items() if k.startswith("audio_net."))),
               "visual_grad_sum": float(sum(p.grad.abs().mean() for k, p in P.items() if k.startswith("visual_net."))),
               "grad_norm": {k: float(p.grad.double().norm()) for k, p in P.items() if p.grad is not None}}
        with torch.no_grad():                                                    # :154 (torch.optim.SGD, first step buf = g)
            for k, p in P.items():
                if p.grad is None:
                    continue
                g = p.grad + wd * p
                if k not in self.mom:
                    self.mom[k] = g.clone()
                else:
                    self.mom[k].mul_(momentum).add_(g)
                p.add_(self.mom[k], alpha=-lr)
        return res

    def feat_dim(self):
        return self.cfg["embed"] << (len(self.cfg["depths"]) - 1)
