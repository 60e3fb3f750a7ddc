#!/usr/bin/env python3
"""Which bf16 rounding of the DGL step produces the deviation of the worst BatchNorm-parameter gradients?   (CPU only)

The step of oracle/torch_step.py in float64 with bf16 rounding inserted at ONE class of storage points at a time -- the points
where the HIP path keeps a bf16 tensor -- against the same step without any rounding:
    w    convolution weights (the packed bf16 copies; masters stay exact)
    in   the network inputs
    y    raw convolution outputs (the saved BatchNorm inputs)                         forward value
    a    BatchNorm(+ReLU)(+add) outputs: a1, block outputs, the pooled stem output   forward value
    dy   gradients at the convolution outputs (what bn_bwd_apply stores)             backward value
    dx   gradients at the activations (what the data gradients store)                backward value
    all  everything above
Prints the logits' deviation and, for the gradient tensors that were worst in tests/test_step_gpu.py::
test_full_size_bf16_against_fp64, |norm - norm64| / norm64 per source, plus the worst tensor of each source.

usage: python3 tools/parity_sources.py [--batch 16] [--threads 8] [--sources w,y,a,dy,dx,all]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fixtures as fx  # noqa: E402

ON = set()


def rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


class Q(torch.autograd.Function):
    """identity with optional bf16 rounding of the value (forward) and of the gradient (backward)"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rb(x) if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


def q(x, f, b):
    f, b = f in ON, b in ON
    return Q.apply(x, f, b) if (f or b) else x


def wq(w):
    return Q.apply(w, True, False) if "w" in ON else w


def bn(x, P, Bf, name):
    return F.batch_norm(x, Bf[name + ".running_mean"].clone(), Bf[name + ".running_var"].clone(), P[name + ".weight"],
                        P[name + ".bias"], training=True, momentum=0.1, eps=1e-5)


def block(x, P, Bf, pre, stride, has_ds):
    y1 = q(F.conv2d(x, wq(P[pre + ".conv1.weight"]), stride=stride, padding=1), "y", "dy")
    a1 = q(F.relu(bn(y1, P, Bf, pre + ".bn1")), "a", "dx")
    y2 = q(F.conv2d(a1, wq(P[pre + ".conv2.weight"]), stride=1, padding=1), "y", "dy")
    out = bn(y2, P, Bf, pre + ".bn2")
    if has_ds:
        yd = q(F.conv2d(x, wq(P[pre + ".downsample.0.weight"]), stride=stride), "y", "dy")
        x = bn(yd, P, Bf, pre + ".downsample.1")
    return q(F.relu(out + x), "a", "dx")


def encoder(x, P, Bf, pre):
    x = q(x, "in", None)
    y = q(F.conv2d(x, wq(P[pre + ".conv1.weight"]), stride=2, padding=3), "y", "dy")
    x = q(F.max_pool2d(F.relu(bn(y, P, Bf, pre + ".bn1")), kernel_size=3, stride=2, padding=1), "a", "dx")
    for li in range(1, 5):
        for bi in range(2):
            x = block(x, P, Bf, f"{pre}.layer{li}.{bi}", 2 if (li > 1 and bi == 0) else 1, li > 1 and bi == 0)
    return x


def step(P0, Bf0, spec, image, label, alpha, want_grads=False):
    P = {k: torch.from_numpy(np.array(v)).double().requires_grad_(True) for k, v in P0.items()}
    Bf = {k: torch.from_numpy(np.array(v)).double() if np.array(v).dtype.kind == "f" else torch.from_numpy(np.array(v)) for k, v in Bf0.items()}
    B, _, T, H, W = image.shape
    a = encoder(spec.unsqueeze(1), P, Bf, "audio_net")
    v = encoder(image.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W), P, Bf, "visual_net")
    v = v.view(B, T, 512, v.shape[-2], v.shape[-1]).permute(0, 2, 1, 3, 4)
    fa = torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)
    fv = torch.flatten(F.adaptive_avg_pool3d(v, 1), 1)
    W_, b_ = P["fusion_module.fc_out.weight"], P["fusion_module.fc_out.bias"]
    z = torch.zeros_like(fa)
    out = F.linear(torch.cat((fa, fv), 1).detach(), W_, b_)
    out_a = F.linear(torch.cat((fa, z), 1), W_, b_)
    out_v = F.linear(torch.cat((z, fv), 1), W_, b_)
    loss = (F.cross_entropy(out_a, label) + F.cross_entropy(out_v, label)) * alpha
    loss.backward()
    g = {k: float(p.grad.norm()) for k, p in P.items() if p.grad is not None and not k.startswith("fusion_module.")}
    lo = torch.cat((out, out_a, out_v), 1).detach().numpy()
    if want_grads:
        return lo, g, {k: p.grad.numpy() for k, p in P.items() if p.grad is not None and not k.startswith("fusion_module.")}
    return lo, g


def two_steps(P0, Bf0, batches, alpha, lr=2e-3, mom=0.9, wd=1e-4, max_norm=40.0):
    """two consecutive steps (encoder parameters only: SGD with momentum + weight decay + clip, main_dgl.py:129,154,249);
    BatchNorm running statistics do not enter a training forward"""
    P = {k: np.array(v, dtype=np.float64) for k, v in P0.items()}
    res = []
    buf = {}
    for spec, image, label in batches:
        lo, g, grads = step(P, Bf0, spec, image, label, alpha, want_grads=True)
        res.append((lo, g))
        total = float(np.sqrt(sum(float((v * v).sum()) for v in grads.values())))
        clip = min(1.0, max_norm / (total + 1e-6))
        for k, gv in grads.items():
            d = gv * clip + wd * P[k]
            buf[k] = d if k not in buf else mom * buf[k] + d
            P[k] = P[k] - lr * buf[k]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 64))
    ap.add_argument("--sources", default="w,in,y,a,dy,dx,all")
    ap.add_argument("--two-steps", action="store_true", help="two consecutive steps with every source on: the deviation of the SECOND step")
    ap.add_argument("--tiny-encoders", action="store_true",
                    help="the tiny encoder fixtures of tests/test_step_gpu.py::test_encoder_golden: element-wise relative error of every "
                         "gradient tensor with all bf16 storage points on (what BF16_ENC_GRAD_TOL bounds)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    P0, Bf0 = fx.model_state(6, "concat_dgl")
    spec, image, label = fx.make_batch(0, args.batch, [257, 188], 3, [224, 224], 6)
    spec, image, label = torch.from_numpy(spec).double(), torch.from_numpy(image).double(), torch.from_numpy(label).long()
    if args.tiny_encoders:
        for name, cin in (("enc_audio_tiny", 1), ("enc_visual_tiny", 3)):
            g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            Pn = fx.make_state(fx.resnet18_param_shapes("", cin))
            Bn = fx.make_state(fx.resnet18_buffer_shapes(""))
            x = torch.from_numpy(g["x"]).double()
            if x.dim() == 5:  # visual fixture [B, 3, T, H, W] -> frames
                B, C, T, H, W = x.shape
                x = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
            res = {}
            for tag, on in (("none", []), ("all", ["w", "in", "y", "a", "dy", "dx"])):
                ON.clear()
                ON.update(on)
                P = {"n." + k: torch.from_numpy(np.array(v)).double().requires_grad_(True) for k, v in Pn.items()}
                Bf = {"n." + k: (torch.from_numpy(np.array(v)).double() if np.array(v).dtype.kind == "f" else torch.from_numpy(np.array(v)))
                      for k, v in Bn.items()}
                y = encoder(x, P, Bf, "n")
                dy = torch.from_numpy(g["dy"]).double()
                if dy.shape != y.shape:  # the visual golden's feature map is [B, 512, T, h, w]
                    B_, C_, T_, h_, w_ = dy.shape
                    dy = dy.permute(0, 2, 1, 3, 4).reshape(B_ * T_, C_, h_, w_)
                y.backward(dy)
                res[tag] = (y.detach().numpy(), {k: p.grad.numpy() for k, p in P.items()})
            rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
            r = {k: rel(res["all"][1][k], res["none"][1][k]) for k in res["none"][1]}
            wk = max(r, key=r.get)
            print(f"{name}: features relerr {rel(res['all'][0], res['none'][0]):.3e}; gradient tensors, element-wise relerr: worst {r[wk]:.3f} "
                  f"({wk}), median {np.median(list(r.values())):.3f}", flush=True)
        return
    if args.two_steps:
        b = [(spec, image, label)]
        s2, i2, l2 = fx.make_batch(1, args.batch, [257, 188], 3, [224, 224], 6)
        b.append((torch.from_numpy(s2).double(), torch.from_numpy(i2).double(), torch.from_numpy(l2).long()))
        ON.clear()
        ref = two_steps(P0, Bf0, b, 4.0)
        ON.update(["w", "in", "y", "a", "dy", "dx"])
        got = two_steps(P0, Bf0, b, 4.0)
        for st, ((lo0, g0), (lo, g)) in enumerate(zip(ref, got)):
            rel = {k: abs(g[k] - g0[k]) / g0[k] for k in g0}
            wk = max(rel, key=rel.get)
            tn0, tn = np.sqrt(sum(v * v for v in g0.values())), np.sqrt(sum(v * v for v in g.values()))
            print(f"B = {args.batch} step {st}, all bf16 storage points on vs none: logits {np.abs(lo - lo0).max():.2e}, total norm "
                  f"{abs(tn - tn0) / tn0:.2e}, per-tensor norms worst {rel[wk]:.4f} ({wk}), median {np.median(list(rel.values())):.4f}", flush=True)
        return
    ON.clear()
    lo0, g0 = step(P0, Bf0, spec, image, label, 4.0)
    watch = ["visual_net.layer1.1.bn1.bias", "audio_net.layer1.1.bn2.bias", "visual_net.layer1.1.bn1.weight", "visual_net.layer1.0.bn2.bias",
             "visual_net.layer1.1.conv1.weight", "visual_net.layer4.1.conv2.weight"]
    print(f"B = {args.batch}: relative deviation of gradient-tensor norms from the un-rounded float64 step, per bf16 rounding source")
    print(f"{'source':>6s}  {'logits':>9s}  " + "  ".join(f"{w.replace('visual_net', 'v').replace('audio_net', 'a'):>22s}" for w in watch) + "   worst tensor")
    for src in args.sources.split(","):
        ON.clear()
        ON.update(["w", "in", "y", "a", "dy", "dx"] if src == "all" else [src])
        lo, g = step(P0, Bf0, spec, image, label, 4.0)
        rel = {k: abs(g[k] - g0[k]) / g0[k] for k in g0}
        wk = max(rel, key=rel.get)
        print(f"{src:>6s}  {np.abs(lo - lo0).max():9.2e}  " + "  ".join(f"{rel[w]:22.4f}" for w in watch) + f"   {wk} {rel[wk]:.4f}", flush=True)


if __name__ == "__main__":
    main()
