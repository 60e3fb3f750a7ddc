"""TEST INFRASTRUCTURE -- the DGL training step restated with PyTorch CPU operators.

Written from scratch on torch.nn.functional (none of the reference's files): it is the "PyTorch-op restatement" of
BASELINE.md section 3 / SURVEY 8(d), timed by bench.py's `cpu_baseline` beside the C port (oracle/gdl_oracle.c) so that
the reported CPU baseline is the arithmetic the reference actually runs on a CPU (ATen / oneDNN kernels), not only a
naive port.  Only tests/, __graft_entry__.smoke() and bench.py's baseline legs (cpu_baseline, and the same restatement on the
GPU as the diagnostic `comparators.torch_rocm`) may import it.

What it follows: ResNet18 without pool / fc (/root/reference/models/backbone.py:75-201: 7x7/2 stem, BN, ReLU,
MaxPool 3/2/1, four layers of two BasicBlocks with a 1x1/2 conv + BN shortcut on the first block of layers 2-4), the
pooling glue and ConcatFusion_DGL head of models/basic_model.py:65-86 and models/fusion_modules.py:45-59, and the step
body of main_dgl.py:97-154 (three cross-entropies, two backward passes with the fusion-head gradients dropped in between,
clip_grad_norm_(40), SGD with momentum 0.9 and weight decay 1e-4).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _bn(x, P, Bf, name, training):
    return F.batch_norm(x, Bf[name + ".running_mean"], Bf[name + ".running_var"], P[name + ".weight"], P[name + ".bias"],
                        training=training, momentum=0.1, eps=1e-5)


def _block(x, P, Bf, pre, stride, has_ds, training):
    out = F.relu(_bn(F.conv2d(x, P[pre + ".conv1.weight"], stride=stride, padding=1), P, Bf, pre + ".bn1", training))
    out = _bn(F.conv2d(out, P[pre + ".conv2.weight"], stride=1, padding=1), P, Bf, pre + ".bn2", training)
    if has_ds:
        x = _bn(F.conv2d(x, P[pre + ".downsample.0.weight"], stride=stride), P, Bf, pre + ".downsample.1", training)
    return F.relu(out + x)


def encoder(x, P, Bf, pre, training):
    """x [N, Cin, H, W] -> [N, 512, h, w]   (ResNet.forward without the permute, backbone.py:166-201)"""
    x = F.relu(_bn(F.conv2d(x, P[pre + ".conv1.weight"], stride=2, padding=3), P, Bf, pre + ".bn1", training))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li in range(1, 5):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _block(x, P, Bf, f"{pre}.layer{li}.{bi}", stride, li > 1 and bi == 0, training)
    return x


class TorchStep:
    def __init__(self, params, buffers, threads=None, device="cpu", autocast=None, channels_last=False, dtype=None):
        """params / buffers: name -> numpy array (oracle.fixtures.model_state).
        device / autocast / channels_last: the SAME restatement on another device -- bench.py's `comparators.torch_rocm`
        runs it on the MI355X with stock PyTorch-ROCm operators (MIOpen / rocBLAS, bf16 autocast, channels_last weights):
        a diagnostic number beside `cpu_baseline`, never the product path and never a parity reference.
        dtype=torch.float64: the step in double precision on the CPU -- the tests' measure of the fp32 oracle's OWN rounding
        error (tests/test_step_gpu.py::test_full_size_oracle_parity)."""
        if threads:
            torch.set_num_threads(int(threads))
        self.dev = torch.device(device)
        self.dtype = dtype
        self.autocast = autocast
        self.cl = bool(channels_last)

        def put(v, grad):
            t = torch.from_numpy(np.array(v)).clone().to(self.dev)
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            if self.cl and t.dim() == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            return t.requires_grad_(True) if grad else t

        self.P = {k: put(v, True) for k, v in params.items()}
        self.Bf = {k: put(v, False) for k, v in buffers.items()}
        self.mom = {}

    def forward(self, spec, image, training=True):
        if self.autocast is not None:
            with torch.autocast(self.dev.type, dtype=self.autocast):
                out, out_a, out_v = self._forward(spec, image, training)
            return out.float(), out_a.float(), out_v.float()
        return self._forward(spec, image, training)

    def _forward(self, spec, image, training=True):
        P, Bf = self.P, self.Bf
        B, _, T, H, W = image.shape
        xa = spec.unsqueeze(1)
        xv = image.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)
        if self.cl:
            xa = xa.contiguous(memory_format=torch.channels_last)
            xv = xv.contiguous(memory_format=torch.channels_last)
        a = encoder(xa, P, Bf, "audio_net", training)
        v = encoder(xv, P, Bf, "visual_net", training)
        v = v.view(B, T, 512, v.shape[-2], v.shape[-1]).permute(0, 2, 1, 3, 4)
        fa = torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)
        fv = torch.flatten(F.adaptive_avg_pool3d(v, 1), 1)
        W_, b_ = P["fusion_module.fc_out.weight"], P["fusion_module.fc_out.bias"]
        z = torch.zeros_like(fa)
        out = F.linear(torch.cat((fa, fv), 1).detach(), W_, b_)
        out_a = F.linear(torch.cat((fa, z), 1), W_, b_)
        out_v = F.linear(torch.cat((z, fv), 1), W_, b_)
        return out, out_a, out_v

    def train_step(self, spec, image, label, alpha, lr, momentum=0.9, wd=1e-4, max_norm=40.0):
        spec, image = torch.as_tensor(spec).to(self.dev), torch.as_tensor(image).to(self.dev)
        if self.dtype is not None:
            spec, image = spec.to(self.dtype), image.to(self.dtype)
        label = torch.as_tensor(label).long().to(self.dev)
        P = self.P
        for p in P.values():
            p.grad = None
        out, out_a, out_v = self.forward(spec, image, True)
        loss_v, loss_a, loss_f = F.cross_entropy(out_v, label), F.cross_entropy(out_a, label), F.cross_entropy(out, label)
        ((loss_a + loss_v) * alpha).backward(retain_graph=True)
        for k, p in P.items():
            if k.startswith("fusion_module."):
                p.grad = None
        loss_f.backward()
        with_grad = [p for p in P.values() if p.grad is not None]
        total = torch.nn.utils.clip_grad_norm_(with_grad, max_norm)
        if self.dev.type == "cpu":
            total = float(total)
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is None:
                    continue
                g = p.grad + wd * p
                if k not in self.mom:
                    self.mom[k] = g.clone()
                else:
                    self.mom[k].mul_(momentum).add_(g)
                p.add_(self.mom[k], alpha=-lr)
        if self.dev.type != "cpu":  # timing use: no host read-back per step
            return {"out": out.detach(), "out_a": out_a.detach(), "out_v": out_v.detach(), "loss_f": loss_f.detach(),
                    "loss_a": loss_a.detach(), "loss_v": loss_v.detach(), "total_norm": total}
        return {"out": out.detach().numpy(), "out_a": out_a.detach().numpy(), "out_v": out_v.detach().numpy(),
                "loss_f": loss_f.item(), "loss_a": loss_a.item(), "loss_v": loss_v.item(), "total_norm": total,
                # post-clip norm of every gradient tensor (what DGLTrainer.read()["grad_norm"] reports)
                "grad_norm": {k: float(p.grad.double().norm()) for k, p in P.items() if p.grad is not None}}
