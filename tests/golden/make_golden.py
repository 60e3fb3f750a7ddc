#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference
(`/root/reference`, PyTorch CPU fp32) in the BUILD container.

Run once here:  python tests/golden/make_golden.py
The reference never travels to the GPU box; only the .npz outputs do.  Inputs
and weights are regenerated on both sides by `oracle/fixtures.py`.

What is restated around the imported model is the step body of
`main_dgl.py:97-154` (the script itself hard-codes cuda:0 and imports
librosa/torchvision, so it cannot run here): zero_grad, forward, three CE
losses, `((loss_a+loss_v)*alpha).backward(retain_graph=True)`, drop of the
fusion-head grads, `loss_f.backward()`, `clip_grad_norm_(.., 40, 2)`, the two
logged `sum(mean(abs(grad)))`, `optimizer.step()` with SGD(momentum .9, wd 1e-4).
"""
import argparse
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import fixtures as fx  # noqa: E402

REF = "/root/reference"


def _import_reference():
    # timm is absent here; models/swin_transformer.py:11 needs three symbols.
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        """timm 0.x `DropPath` / `drop_path(x, drop_prob, training, scale_by_keep=True)` restated (timm is not in this image):
        identity in eval mode or at probability 0; otherwise one Bernoulli(keep) per sample of dim 0, divided by keep.  Every
        drawn mask is appended to `DropPath.record` so that a fixture can hold what the reference's forward multiplied with."""
        record = []

        def __init__(self, p=0.0):
            super().__init__()
            self.p = float(p)

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1.0 - self.p
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0:
                m.div_(keep)
            DropPath.record.append(m.reshape(-1).clone())
            return x * m

    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: (x, x)
    tl.trunc_normal_ = nn.init.trunc_normal_
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = tm
    sys.modules["timm.models.layers"] = tl
    sys.path.insert(0, REF)
    import models.basic_model as bm  # noqa
    import models.backbone as bb  # noqa
    import models.fusion_modules as fm  # noqa

    return bm, bb, fm


def _load(model, n_classes, fusion):
    ps, bs = fx.model_state(n_classes, fusion)
    sd = {k: torch.from_numpy(v.copy()) for k, v in {**ps, **bs}.items()}
    missing = model.load_state_dict(sd, strict=True)
    return missing


class _ConcatAV(nn.Module):
    """BASELINE config 1: the reference's `AVClassifier` (non-DGL) class is
    missing from models/basic_model.py; its math is the two
    imported encoders + imported `ConcatFusion` with the pooling glue of
    basic_model.py:73-82 and a single CE loss (main.py:161-175)."""

    def __init__(self, bb, fm, n_classes):
        super().__init__()
        self.fusion_module = fm.ConcatFusion(output_dim=n_classes)
        self.audio_net = bb.resnet18(modality="audio", args=None)
        self.visual_net = bb.resnet18(modality="visual", args=None)

    def forward(self, audio, visual):
        import torch.nn.functional as F

        a = self.audio_net(audio)
        v = self.visual_net(visual)
        (_, C, H, W) = v.size()
        B = a.size()[0]
        v = v.view(B, -1, C, H, W).permute(0, 2, 1, 3, 4)
        a = torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)
        v = torch.flatten(F.adaptive_avg_pool3d(v, 1), 1)
        _, _, out = self.fusion_module(a, v)
        return out, a, v


class _SwinDGL(nn.Module):
    """BASELINE config 5 (a composition the reference never instantiates) from the reference's own parts:
    imported `resnet18('audio')` + pooling of basic_model.py:73-75, imported `SwinTransformer` (pooled [B*T, C] features,
    averaged over the T frames of a sample like basic_model.py:77-80 does for the ResNet branch) and imported
    `ConcatFusion_DGL` at input_dim = 512 + C."""

    def __init__(self, bb, fm, n_classes, cfg):
        super().__init__()
        import models.swin_transformer as sw

        feat = cfg["embed"] << (len(cfg["depths"]) - 1)
        self.fusion_module = fm.ConcatFusion_DGL(input_dim=512 + feat, output_dim=n_classes)
        self.audio_net = bb.resnet18(modality="audio", args=None)
        self.visual_net = sw.SwinTransformer(argparse.Namespace(pe=0), "visual", img_size=cfg["img"], patch_size=cfg["patch"],
                                             in_chans=3, embed_dim=cfg["embed"], depths=list(cfg["depths"]),
                                             num_heads=list(cfg["heads"]), window_size=cfg["window"],
                                             mlp_ratio=float(cfg["mlp"]), drop_path_rate=0.0)

    def forward(self, audio, visual):
        import torch.nn.functional as F

        a = torch.flatten(F.adaptive_avg_pool2d(self.audio_net(audio), 1), 1)
        B, T = visual.shape[0], visual.shape[2]
        v = self.visual_net(visual).view(B, T, -1).mean(1)
        a_out, v_out, out = self.fusion_module(a, v)
        return out, a_out, v_out


def _summ(t):
    a = t.detach().double().flatten()
    return np.array([a.sum().item(), a.abs().sum().item()], dtype=np.float64)


def run_step_case(name, bm, bb, fm, dataset, spec_hw, frames, image_hw, batch, alpha, steps, mode="dgl", seed=0,
                  lr=2e-3, fusion="concat", swin_cfg=None):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    n_classes = fx.N_CLASSES[dataset]
    if swin_cfg is not None:
        model = _SwinDGL(bb, fm, n_classes, swin_cfg)
        ps, bs = fx.swin_dgl_state(n_classes, swin_cfg)
        assert [n for n, _ in model.named_parameters()] == list(ps)
        res = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in {**ps, **bs}.items()}, strict=False)
        assert not res.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in res.missing_keys)
    elif mode == "dgl":
        args = argparse.Namespace(fusion_method=fusion, dataset=dataset, modality="full", batch_size=batch)
        model = bm.AVClassifier_DGL(args)
        _load(model, n_classes, fusion + "_dgl")
    else:
        model = _ConcatAV(bb, fm, n_classes)
        _load(model, n_classes, "concat")
    # main_dgl.py:244 wraps in DataParallel, which supplies the 'module.' prefix
    # that the grad-drop filter (main_dgl.py:114-119) keys on; on CPU we mimic the prefix.
    named = [("module." + n, p) for n, p in model.named_parameters()]
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)  # main_dgl.py:249
    crit = nn.CrossEntropyLoss()  # main_dgl.py:71
    out_d = {}
    cfg = dict(name=name, dataset=dataset, n_classes=n_classes, spec_hw=list(spec_hw), frames=frames,
               image_hw=list(image_hw), batch=batch, alpha=alpha, steps=steps, mode=mode, seed=seed, lr=lr,
               momentum=0.9, weight_decay=1e-4, max_norm=40.0, torch=torch.__version__, fusion=fusion)
    if swin_cfg is not None:
        cfg["swin"] = dict(swin_cfg)
    out_d["config"] = np.array(json.dumps(cfg))
    model.train()
    for st in range(steps):
        spec, image, label = fx.make_batch(seed + st, batch, spec_hw, frames, image_hw, n_classes)
        spec, image, label = torch.from_numpy(spec), torch.from_numpy(image), torch.from_numpy(label)
        opt.zero_grad()  # main_dgl.py:97
        pre = f"s{st}."
        if mode == "dgl":
            out, out_a, out_v = model(spec.unsqueeze(1).float(), image.float())  # :100
            loss_v = crit(out_v, label)
            loss_a = crit(out_a, label)
            loss_f = crit(out, label)
            loss_unimodal = (loss_a + loss_v) * alpha  # :108
            loss_unimodal.backward(retain_graph=True)  # :110
            dropped = 0.0
            for n, p in named:  # :114-119
                layer = str(n).split(".")[1]
                if "fusion" in layer:
                    if p.grad is not None:
                        dropped += float(p.grad.double().pow(2).sum())
                    p.grad = None
            loss_f.backward()  # :122
            out_d[pre + "out_a"] = out_a.detach().numpy()
            out_d[pre + "out_v"] = out_v.detach().numpy()
            out_d[pre + "loss_a"] = np.float64(loss_a.item())
            out_d[pre + "loss_v"] = np.float64(loss_v.item())
            out_d[pre + "dropped_head_gradnorm"] = np.float64(dropped ** 0.5)
        else:
            out, fa, fv = model(spec.unsqueeze(1).float(), image.float())
            loss_f = crit(out, label)
            loss_f.backward()
        out_d[pre + "out"] = out.detach().numpy()
        out_d[pre + "loss_f"] = np.float64(loss_f.item())
        total = nn.utils.clip_grad_norm_(model.parameters(), max_norm=40, norm_type=2)  # :129
        out_d[pre + "total_norm"] = np.float64(total.item())
        a_sum = sum(torch.abs(p.grad).mean().item() for p in model.audio_net.parameters())  # :132-137
        v_sum = sum(torch.abs(p.grad).mean().item() for p in model.visual_net.parameters())  # :139-143
        out_d[pre + "audio_grad_sum"] = np.float64(a_sum)
        out_d[pre + "visual_grad_sum"] = np.float64(v_sum)
        names, gn, gam, gh8, isnone = [], [], [], [], []
        for n, p in model.named_parameters():
            names.append(n)
            if p.grad is None:
                isnone.append(1)
                gn.append(0.0)
                gam.append(0.0)
                gh8.append(np.zeros(8, np.float32))
                continue
            isnone.append(0)
            g = p.grad.detach()
            gn.append(float(g.double().norm()))
            gam.append(float(g.abs().double().mean()))
            h = g.flatten()[:8].numpy()
            gh8.append(np.pad(h, (0, 8 - len(h))))
            if g.numel() <= 10000:
                out_d[pre + "grad." + n] = g.numpy().copy()
        out_d[pre + "grad_names"] = np.array(names)
        out_d[pre + "grad_norm"] = np.array(gn)
        out_d[pre + "grad_absmean"] = np.array(gam)
        out_d[pre + "grad_head8"] = np.stack(gh8)
        out_d[pre + "grad_is_none"] = np.array(isnone, dtype=np.int8)
        opt.step()  # :154
        ps, ph8, ms = [], [], []
        for n, p in model.named_parameters():
            ps.append(_summ(p))
            h = p.detach().flatten()[:8].numpy()
            ph8.append(np.pad(h, (0, 8 - len(h))))
            buf = opt.state.get(p, {}).get("momentum_buffer", None)
            ms.append(_summ(buf) if buf is not None else np.zeros(2))
        out_d[pre + "param_sums"] = np.stack(ps)
        out_d[pre + "param_head8"] = np.stack(ph8)
        out_d[pre + "momentum_sums"] = np.stack(ms)
        for n, b in model.named_buffers():
            out_d[pre + "buf." + n] = b.detach().numpy().copy()
    # eval-mode forward after the last step (valid(), main_dgl.py:185-204)
    model.eval()
    with torch.no_grad():
        spec, image, label = fx.make_batch(seed + 1000, batch, spec_hw, frames, image_hw, n_classes)
        o = model(torch.from_numpy(spec).unsqueeze(1).float(), torch.from_numpy(image).float())
        out_d["eval.out"] = o[0].numpy()
        if mode == "dgl":
            out_d["eval.out_a"] = o[1].numpy()
            out_d["eval.out_v"] = o[2].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out_d)
    print(name, "loss_f", out_d[f"s{steps-1}.loss_f"], "total_norm", out_d[f"s{steps-1}.total_norm"])


def run_encoder_case(name, bb, modality, in_shape, seed=0):
    """Full feature map + parameter grads of one imported `resnet18`
    (backbone.py:255) on a tiny input, for a fixed upstream gradient."""
    net = bb.resnet18(modality=modality, args=None)
    prefix = ""
    ps = fx.make_state(fx.resnet18_param_shapes(prefix, 1 if modality == "audio" else 3))
    bs = fx.make_state(fx.resnet18_buffer_shapes(prefix))
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in {**ps, **bs}.items()}, strict=True)
    r = np.random.default_rng([77, seed])
    x = r.standard_normal(in_shape, dtype=np.float32)
    net.train()
    y = net(torch.from_numpy(x))
    dy = np.random.default_rng([78, seed]).standard_normal(tuple(y.shape), dtype=np.float32)
    y.backward(torch.from_numpy(dy))
    d = {"x": x, "y": y.detach().numpy(), "dy": dy}
    for n, p in net.named_parameters():
        g = p.grad.numpy()
        if g.size <= 10000:
            d["grad." + n] = g.copy()
        else:  # big conv weights: L2 norm, mean|g| and a strided sample (every 997th element)
            d["gradstat." + n] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum()), np.abs(g).mean()])
            d["gradsample." + n] = g.reshape(-1)[::997].copy()
    for n, b in net.named_buffers():
        d["buf." + n] = b.detach().numpy().copy()
    net.eval()
    with torch.no_grad():
        d["y_eval"] = net(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, d["y"].shape, float(np.abs(d["y"]).mean()))


def run_swin_case(name, cfg, batch, frames, seed=0, drop_path_rate=0.0):
    """Pooled features + parameter gradients of the imported `SwinTransformer` (swin_transformer.py:486-674, through the
    3-symbol timm stub above; drop_path_rate = 0 makes the identity DropPath exact) for a fixed upstream gradient.
    drop_path_rate > 0: the training forward under torch.manual_seed(seed); the fixture keeps the masks the blocks drew
    (`drop_scales` [blocks][2][frames]; a block of probability 0 is nn.Identity and keeps scale 1)."""
    import models.swin_transformer as sw
    from timm.models.layers import DropPath

    args = argparse.Namespace(pe=0)
    net = sw.SwinTransformer(args, "visual", img_size=cfg["img"], patch_size=cfg["patch"], in_chans=3, embed_dim=cfg["embed"],
                             depths=list(cfg["depths"]), num_heads=list(cfg["heads"]), window_size=cfg["window"],
                             mlp_ratio=float(cfg["mlp"]), drop_path_rate=float(drop_path_rate))
    ps = fx.make_state(fx.swin_param_shapes(cfg))
    assert [n for n, _ in net.named_parameters()] == list(ps), "parameter order differs from the reference's"
    missing = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in ps.items()}, strict=False)
    assert not missing.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in missing.missing_keys)
    x = fx.swin_input(cfg, batch, frames, seed)
    net.train()
    DropPath.record.clear()
    torch.manual_seed(seed)
    y = net(torch.from_numpy(x))
    dy = np.random.default_rng([79, seed]).standard_normal(tuple(y.shape), dtype=np.float32)
    y.backward(torch.from_numpy(dy))
    d = {"config": np.array(json.dumps(dict(cfg, batch=batch, frames=frames, seed=seed))), "y": y.detach().numpy(), "dy": dy}
    if drop_path_rate > 0:
        nb, nf = sum(cfg["depths"]), batch * frames
        probs = torch.linspace(0, drop_path_rate, nb).tolist()  # swin_transformer.py:546
        scales = np.ones((nb, 2, nf), np.float32)
        rec = iter(DropPath.record)
        for k, pk in enumerate(probs):
            if pk > 0:  # (:218: probability 0 -> nn.Identity, nothing drawn)
                scales[k, 0], scales[k, 1] = next(rec).numpy(), next(rec).numpy()
        assert next(rec, None) is None and (scales == 0).any(), "the masks of this seed drop nothing: pick another"
        d["drop_scales"] = scales
        d["config"] = np.array(json.dumps(dict(cfg, batch=batch, frames=frames, seed=seed, drop_path_rate=drop_path_rate)))
    for n, p in net.named_parameters():
        g = p.grad.numpy()
        d["gradstat." + n] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum()), np.abs(g).mean()])
        if g.size <= 10000:
            d["grad." + n] = g.copy()
        else:
            d["gradsample." + n] = g.reshape(-1)[::997].copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, d["y"].shape, float(np.abs(d["y"]).mean()))


def run_head_case(name, fm, cls, n_classes, batch=4):
    """Isolated fusion head (fusion_modules.py:33-59) forward + both backward
    phases of main_dgl.py:110-122."""
    head = getattr(fm, cls)(output_dim=n_classes)
    shapes = {"fc_out.weight": (n_classes, 1024), "fc_out.bias": (n_classes,)}
    if cls == "ConcatFusion_DGL":
        shapes.update({"fc_auxi.weight": (n_classes, 1024), "fc_auxi.bias": (n_classes,)})
    st = fx.make_state({"fusion_module." + k: v for k, v in shapes.items()})
    head.load_state_dict({k[len("fusion_module."):]: torch.from_numpy(v.copy()) for k, v in st.items()})
    r = np.random.default_rng([91, n_classes])
    x = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    y = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    g = [r.standard_normal((batch, n_classes), dtype=np.float32) for _ in range(3)]
    o = head(x, y)
    d = {"x": x.detach().numpy(), "y": y.detach().numpy()}
    if cls == "ConcatFusion_DGL":
        x_out, y_out, out = o
        d.update(x_out=x_out.detach().numpy(), y_out=y_out.detach().numpy(), out=out.detach().numpy())
        (x_out * torch.from_numpy(g[0])).sum().add((y_out * torch.from_numpy(g[1])).sum()).backward(retain_graph=True)
        d.update(g_x_out=g[0], g_y_out=g[1], g_out=g[2], dx=x.grad.numpy().copy(), dy=y.grad.numpy().copy(),
                 dW_uni=head.fc_out.weight.grad.numpy().copy(), db_uni=head.fc_out.bias.grad.numpy().copy())
        head.fc_out.weight.grad = None
        head.fc_out.bias.grad = None
        x.grad = None
        y.grad = None
        (out * torch.from_numpy(g[2])).sum().backward()
        d.update(dW_f=head.fc_out.weight.grad.numpy().copy(), db_f=head.fc_out.bias.grad.numpy().copy(),
                 dx_after_f=np.zeros(1) if x.grad is None else x.grad.numpy().copy(),
                 auxi_grad_is_none=np.int8(head.fc_auxi.weight.grad is None))
    else:
        _, _, out = o
        d.update(out=out.detach().numpy(), g_out=g[2])
        (out * torch.from_numpy(g[2])).sum().backward()
        d.update(dx=x.grad.numpy().copy(), dy=y.grad.numpy().copy(), dW=head.fc_out.weight.grad.numpy().copy(),
                 db=head.fc_out.bias.grad.numpy().copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "ok")


def run_sum_head_case(name, fm, n_classes, batch=4):
    """SumFusion_DGL (fusion_modules.py:16-30): forward + both backward phases of main_dgl.py:110-122."""
    head = fm.SumFusion_DGL(output_dim=n_classes)
    shapes = {"fc_x.weight": (n_classes, 512), "fc_x.bias": (n_classes,), "fc_y.weight": (n_classes, 512),
              "fc_y.bias": (n_classes,)}
    st = fx.make_state({"fusion_module." + k: v for k, v in shapes.items()})
    head.load_state_dict({k[len("fusion_module."):]: torch.from_numpy(v.copy()) for k, v in st.items()})
    r = np.random.default_rng([92, n_classes])
    x = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    y = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    g = [r.standard_normal((batch, n_classes), dtype=np.float32) for _ in range(3)]
    x_out, y_out, out = head(x, y)
    d = {"x": x.detach().numpy(), "y": y.detach().numpy(), "x_out": x_out.detach().numpy(), "y_out": y_out.detach().numpy(),
         "out": out.detach().numpy(), "g_x_out": g[0], "g_y_out": g[1], "g_out": g[2]}
    (x_out * torch.from_numpy(g[0])).sum().add((y_out * torch.from_numpy(g[1])).sum()).backward(retain_graph=True)
    d.update(dx=x.grad.numpy().copy(), dy=y.grad.numpy().copy())
    for n, p in head.named_parameters():
        d["uni." + n] = p.grad.numpy().copy()
        p.grad = None
    x.grad = None
    y.grad = None
    (out * torch.from_numpy(g[2])).sum().backward()
    for n, p in head.named_parameters():
        d["f." + n] = p.grad.numpy().copy()
    d["dx_after_f_is_none"] = np.int8(x.grad is None)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "ok")


def run_gated_head_case(name, fm, n_classes, batch=4):
    """GatedFusion_DGL(x_gate=True) (fusion_modules.py:213-250): forward + both backward phases of main_dgl.py:110-122."""
    head = fm.GatedFusion_DGL(output_dim=n_classes, x_gate=True)
    shapes = {"fc_x.weight": (512, 512), "fc_x.bias": (512,), "fc_y.weight": (512, 512), "fc_y.bias": (512,),
              "fc_out.weight": (n_classes, 512), "fc_out.bias": (n_classes,)}
    st = fx.make_state({"fusion_module." + k: v for k, v in shapes.items()})
    head.load_state_dict({k[len("fusion_module."):]: torch.from_numpy(v.copy()) for k, v in st.items()})
    r = np.random.default_rng([93, n_classes])
    x = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    y = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    g = [r.standard_normal((batch, n_classes), dtype=np.float32) for _ in range(3)]
    x_out, y_out, out = head(x, y)
    d = {"x": x.detach().numpy(), "y": y.detach().numpy(), "x_out": x_out.detach().numpy(), "y_out": y_out.detach().numpy(),
         "out": out.detach().numpy(), "g_x_out": g[0], "g_y_out": g[1], "g_out": g[2]}
    (x_out * torch.from_numpy(g[0])).sum().add((y_out * torch.from_numpy(g[1])).sum()).backward(retain_graph=True)
    d.update(dx=x.grad.numpy().copy(), dy=y.grad.numpy().copy())
    def store(key, a):  # the 512x512 gradients: L2 norm + every 97th element keep the fixture small
        if a.size <= 10000:
            d[key] = a.copy()
        else:
            d[key + ".norm"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
            d[key + ".sample97"] = a.reshape(-1)[::97].copy()

    for n, p in head.named_parameters():
        store("uni." + n, p.grad.numpy())
        p.grad = None
    x.grad = None
    y.grad = None
    (out * torch.from_numpy(g[2])).sum().backward()
    for n, p in head.named_parameters():
        d["f_is_none." + n] = np.int8(p.grad is None)
        if p.grad is not None:
            store("f." + n, p.grad.numpy())
    d["dx_after_f_is_none"] = np.int8(x.grad is None)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "ok", {n: int(d["f_is_none." + n]) for n, _ in head.named_parameters()})


def run_seeded_init_case(name, bm):
    """`setup_seed(0)` -> `AVClassifier_DGL(args)` -> `.apply(weight_init)` (main_dgl.py:226-238) for every fusion head
    this build provides: per-tensor (sum, sum|.|) of the resulting state_dict.  Pins constructor order, constructor-time
    RNG consumption and weight_init of the drop-in modules."""
    import utils.utils as ut  # the reference's (REF is first on sys.path)

    d = {}
    for fusion in ("concat", "sum", "gated", "film"):
        ut.setup_seed(0)
        args = argparse.Namespace(fusion_method=fusion, dataset="CREMAD", modality="full", batch_size=2)
        model = bm.AVClassifier_DGL(args)
        model.apply(ut.weight_init)
        sd = model.state_dict()
        d[fusion + ".keys"] = np.array(list(sd.keys()))
        d[fusion + ".sums"] = np.stack([_summ(v.float()) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "ok")


def run_film_head_case(name, fm, n_classes, batch=3):
    """FiLM_DGL (fusion_modules.py:126-178): forward + both backward phases of main_dgl.py:110-122."""
    head = fm.FiLM_DGL(output_dim=n_classes, x_film=True)
    shapes = {"fc.weight": (512, 512 * 512), "fc.bias": (512,), "fc_out.weight": (n_classes, 512), "fc_out.bias": (n_classes,)}
    st = fx.make_state({"fusion_module." + k: v for k, v in shapes.items()})
    head.load_state_dict({k[len("fusion_module."):]: torch.from_numpy(v.copy()) for k, v in st.items()})
    r = np.random.default_rng([94, n_classes])
    x = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    y = torch.from_numpy(r.standard_normal((batch, 512), dtype=np.float32)).requires_grad_()
    g = [r.standard_normal((batch, n_classes), dtype=np.float32) for _ in range(3)]
    x_out, y_out, out = head(x, y)
    d = {"x": x.detach().numpy(), "y": y.detach().numpy(), "x_out": x_out.detach().numpy(), "y_out": y_out.detach().numpy(),
         "out": out.detach().numpy(), "g_x_out": g[0], "g_y_out": g[1], "g_out": g[2]}

    def store(key, a):
        if a.size <= 10000:
            d[key] = a.copy()
        else:
            d[key + ".norm"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
            d[key + ".sample97"] = a.reshape(-1)[::9973].copy() if a.size > 10 ** 7 else a.reshape(-1)[::97].copy()

    (x_out * torch.from_numpy(g[0])).sum().add((y_out * torch.from_numpy(g[1])).sum()).backward(retain_graph=True)
    d.update(dx=x.grad.numpy().copy(), dy=y.grad.numpy().copy())
    for n, p in head.named_parameters():
        store("uni." + n, p.grad.numpy())
        p.grad = None
    x.grad = None
    y.grad = None
    (out * torch.from_numpy(g[2])).sum().backward()
    for n, p in head.named_parameters():
        store("f." + n, p.grad.numpy())
    d["dx_after_f_is_none"] = np.int8(x.grad is None)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "ok")


def run_input_pipeline_case(name):
    """Input pipeline.  The reference's datasets need librosa and torchvision, neither of which is
    installed here, so these vectors come from the independent implementations that ARE: torch.stft with librosa's
    defaults spelled out (periodic Hann of n_fft samples, center=True, pad_mode constant / reflect) and the torch
    operations torchvision's ToTensor / Normalize are documented to perform."""
    g = torch.Generator().manual_seed(20250824)
    out = {}
    for tag, n_fft, hop, n in (("cremad", 512, 353, 3000), ("ks", 256, 128, 2000)):
        wave = torch.randn(2, n, generator=g) * 0.6  # some samples exceed +-1: the clip matters
        out[f"{tag}.wave"] = wave.numpy()
        for pm in ("constant", "reflect"):
            X = torch.stft(wave.clamp(-1.0, 1.0).double(), n_fft, hop_length=hop, win_length=n_fft,
                           window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True, pad_mode=pm,
                           return_complex=True)
            out[f"{tag}.{pm}"] = torch.log(X.abs() + 1e-7).float().numpy()
    frames = torch.randint(0, 256, (2, 3, 12, 10, 3), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    t = frames.permute(0, 1, 4, 2, 3).float().div(255)                      # ToTensor
    t = t.sub(mean.view(1, 1, 3, 1, 1)).div(std.view(1, 1, 3, 1, 1))        # Normalize
    out["frames.u8"] = frames.numpy()
    out["frames.norm"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {len(out)} arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if not a.only or a.only in "input_pipeline":
        run_input_pipeline_case("input_pipeline")  # (needs no reference import)
        if a.only:
            return
    bm, bb, fm = _import_reference()
    cases = {
        "head_dgl_c6": lambda: run_head_case("head_dgl_c6", fm, "ConcatFusion_DGL", 6),
        "head_dgl_c34": lambda: run_head_case("head_dgl_c34", fm, "ConcatFusion_DGL", 34),
        "head_concat_c6": lambda: run_head_case("head_concat_c6", fm, "ConcatFusion", 6),
        "enc_audio_tiny": lambda: run_encoder_case("enc_audio_tiny", bb, "audio", (2, 1, 65, 47)),
        "enc_visual_tiny": lambda: run_encoder_case("enc_visual_tiny", bb, "visual", (2, 3, 2, 64, 64)),
        "dgl_tiny_b4": lambda: run_step_case("dgl_tiny_b4", bm, bb, fm, "CREMAD", (65, 47), 2, (64, 64), 4, 4.0, 2),
        "dgl_tiny_t1_b2": lambda: run_step_case("dgl_tiny_t1_b2", bm, bb, fm, "CREMAD", (65, 47), 1, (64, 64), 2, 5.0,
                                                1),
        "dgl_cremad_b2": lambda: run_step_case("dgl_cremad_b2", bm, bb, fm, "CREMAD", (257, 188), 3, (224, 224), 2,
                                               4.0, 1),
        "dgl_ks_b2": lambda: run_step_case("dgl_ks_b2", bm, bb, fm, "KineticSound", (129, 626), 3, (224, 224), 2, 2.0,
                                           1),
        "seeded_init": lambda: run_seeded_init_case("seeded_init", bm),
        "head_sum_dgl_c6": lambda: run_sum_head_case("head_sum_dgl_c6", fm, 6),
        "dgl_sum_tiny_b4": lambda: run_step_case("dgl_sum_tiny_b4", bm, bb, fm, "CREMAD", (65, 47), 2, (64, 64), 4, 4.0, 2,
                                                 fusion="sum"),
        "dgl_film_tiny_b4": lambda: run_step_case("dgl_film_tiny_b4", bm, bb, fm, "CREMAD", (65, 47), 2, (64, 64), 4, 4.0, 2,
                                                  fusion="film"),
        "head_film_dgl_c6": lambda: run_film_head_case("head_film_dgl_c6", fm, 6),
        "head_gated_dgl_c6": lambda: run_gated_head_case("head_gated_dgl_c6", fm, 6),
        "dgl_gated_tiny_b4": lambda: run_step_case("dgl_gated_tiny_b4", bm, bb, fm, "CREMAD", (65, 47), 2, (64, 64), 4, 4.0, 2,
                                                   fusion="gated"),
        "dgl_swin_tiny_b4": lambda: run_step_case("dgl_swin_tiny_b4", bm, bb, fm, "CREMAD", (65, 47), 2, (56, 56), 4, 4.0, 2,
                                                  swin_cfg=fx.SWIN_TINY2),
        "swin_tiny2_b2": lambda: run_swin_case("swin_tiny2_b2", fx.SWIN_TINY2, 2, 2),
        "swin_t_b1": lambda: run_swin_case("swin_t_b1", fx.SWIN_T, 1, 2),
        "swin_tiny2_drop_b3": lambda: run_swin_case("swin_tiny2_drop_b3", fx.SWIN_TINY2, 3, 2, seed=1, drop_path_rate=0.3),
        "concat_cremad_b2": lambda: run_step_case("concat_cremad_b2", bm, bb, fm, "CREMAD", (257, 188), 3, (224, 224),
                                                  2, 0.0, 1, mode="concat"),
    }
    for k, f in cases.items():
        if a.only and a.only not in k:
            continue
        f()


if __name__ == "__main__":
    main()
