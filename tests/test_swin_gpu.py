"""Swin visual encoder on the MI355X: the HIP engine (gdl/swin.py over csrc/swin.hip + the library's 1x1
convolutions) against the golden vectors captured from the imported reference SwinTransformer and against the CPU oracle
(oracle/swin_oracle.py).  Every call goes through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV, L

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def _run(cfg, B, T, seed, dy, dtype, drop=None):
    from gdl.swin import SwinEngine

    eng = SwinEngine(cfg, dtype, B, T, DEV)
    P = fx.make_state(fx.swin_param_shapes(cfg))
    assert [n for n, _ in eng.param_shapes()] == list(P)
    params = [torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    x = torch.from_numpy(fx.swin_input(cfg, B, T, seed)).to(DEV)
    y = eng.forward(x, drop_scales=None if drop is None else torch.from_numpy(np.asarray(drop, np.float32)).to(DEV)).clone()
    grads = [torch.full_like(p, float("nan")) for p in params]
    eng.backward(torch.from_numpy(dy).to(DEV), grads)
    torch.cuda.synchronize()
    return y.cpu().numpy(), {k: g.cpu().numpy() for k, g in zip(P, grads)}, eng


def _relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name,cfg_name", [("swin_tiny2_b2", "SWIN_TINY2"), ("swin_t_b1", "SWIN_T"), ("swin_tiny2_drop_b3", "SWIN_TINY2")])
def test_swin_engine_golden(golden_dir, name, cfg_name, dtype):
    """`swin_tiny2_drop_b3`: the reference's TRAINING forward / backward with drop_path_rate = 0.3 and the DropPath masks its blocks
    drew (the fixture holds them): stochastic depth through gdl_swin_drop_path, forward and backward."""
    cfg = getattr(fx, cfg_name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    c = json.loads(str(g["config"]))
    y, grads, _ = _run(cfg, c["batch"], c["frames"], c["seed"], g["dy"], dtype, g["drop_scales"] if "drop_scales" in g.files else None)
    f32 = dtype == "f32"
    ey = _relerr(y, g["y"])
    worst, worst_k = 0.0, None
    bad = [k for k, v in grads.items() if not np.isfinite(v).all()]
    assert not bad, (len(bad), bad[:6])
    for k, v in grads.items():
        want_norm = g["gradstat." + k][0]
        en = abs(np.sqrt((v.astype(np.float64) ** 2).sum()) - want_norm) / want_norm
        if "grad." + k in g.files:
            ee = _relerr(v, g["grad." + k])
        else:
            ee = _relerr(v.reshape(-1)[::997], g["gradsample." + k])
        if max(en, ee) > worst:
            worst, worst_k = max(en, ee), k
    print(f"swin {name} {dtype}: features {ey:.2e}, worst gradient {worst:.2e} ({worst_k})")
    # f32: exact-f32 MFMA + fp32 elementwise against the fp32 reference; bf16: storage rounding through 4 / 24 LayerNorm'd
    # residual blocks (element-wise relative error of whole tensors, as for the ResNet encoders)
    # measured: f32 4e-7 / 2.7e-6, bf16 6.7e-3 / 2.4e-2 (Swin-T, 224 x 224), both worst on a relative-position bias table
    assert ey < (5e-6 if f32 else 2e-2), ey
    assert worst < (3e-5 if f32 else 6e-2), (worst_k, worst)


def test_swin_drop_path_graph_replays_and_mirror(golden_dir):
    """Stochastic depth beyond the first eager passes: (i) the engine's forward / backward become HIP-graph replays from the third
    call on -- new masks must still take effect (they are copied into an engine-owned buffer outside the graph); (ii) the drop-in
    module in training mode draws masks itself (`last_drop_scales`), the same module with those masks pinned reproduces the
    pass bit for bit, and its gradients agree with the CPU oracle run on the SAME masks; eval mode ignores the rate."""
    import argparse

    from models.swin_transformer import SwinTransformer
    from oracle import swin_oracle as so

    cfg = fx.SWIN_TINY2
    g = np.load(os.path.join(golden_dir, "swin_tiny2_drop_b3.npz"))
    c = json.loads(str(g["config"]))
    B, T = c["batch"], c["frames"]
    from gdl.swin import SwinEngine

    eng = SwinEngine(cfg, "f32", B, T, DEV)
    P = fx.make_state(fx.swin_param_shapes(cfg))
    params = [torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    x = torch.from_numpy(fx.swin_input(cfg, B, T, c["seed"])).to(DEV)
    dy = torch.from_numpy(g["dy"]).to(DEV)
    grads = [torch.empty_like(p) for p in params]
    ones = torch.ones_like(torch.from_numpy(g["drop_scales"])).to(DEV)
    rec = torch.from_numpy(g["drop_scales"]).to(DEV)
    outs = []
    for it in range(6):  # eager, eager, captured, replays -- alternating all-ones masks and the recorded ones
        sc = rec if it % 2 else ones
        y = eng.forward(x, drop_scales=sc).clone()
        eng.backward(dy, grads)
        outs.append((y.cpu().numpy(), grads[0].cpu().numpy().copy()))
    for it in (3, 5):
        assert _relerr(outs[it][0], g["y"]) < 5e-6
        np.testing.assert_array_equal(outs[it][0], outs[1][0])
        np.testing.assert_array_equal(outs[it][1], outs[1][1])
    y_plain = eng.forward(x).clone().cpu().numpy()  # no DropPath at all = all-ones scales
    np.testing.assert_allclose(outs[4][0], y_plain, rtol=0, atol=2e-6 * np.abs(y_plain).max())
    assert _relerr(outs[4][0], g["y"]) > 1e-2
    # (ii) the mirror
    args = argparse.Namespace(pe=0)
    net = SwinTransformer(args, "visual", img_size=cfg["img"], patch_size=cfg["patch"], embed_dim=cfg["embed"], depths=list(cfg["depths"]),
                          num_heads=list(cfg["heads"]), window_size=cfg["window"], mlp_ratio=float(cfg["mlp"]), drop_path_rate=0.5).to(DEV)
    net.gdl_dtype = "f32"
    res = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=False)
    assert not res.unexpected_keys
    net.train()
    torch.manual_seed(11)
    y1 = net(x)
    sc1 = net.last_drop_scales.clone()
    assert tuple(sc1.shape) == (4, 2, B * T) and bool((sc1[0] == 1).all()) and bool((sc1 == 0).any())
    (y1 * dy).sum().backward()
    g1 = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    net.drop_scales_override = sc1
    y2 = net(x)
    (y2 * dy).sum().backward()
    assert torch.equal(y1, y2) and all(torch.equal(g1[n], p.grad) for n, p in net.named_parameters())
    yo, go = so.forward_backward(fx.swin_input(cfg, B, T, c["seed"]), P, cfg, g["dy"], drop=sc1.cpu().numpy())
    assert _relerr(y1.detach().cpu().numpy(), yo) < 5e-6
    worst = max(_relerr(g1[n].cpu().numpy(), go[n]) for n in go)
    assert worst < 3e-5, worst
    net.eval()
    net.drop_scales_override = None
    with torch.no_grad():
        ye = net(x)
    np.testing.assert_allclose(ye.cpu().numpy(), y_plain, rtol=0, atol=2e-6 * np.abs(y_plain).max())


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_swin_batched_pack_equals_single_matrix_calls(dtype):
    """gdl_swin_pack_batched (one launch per step for every Linear / LayerNorm parameter: 32 x 32 tiles, the transposed copy through
    LDS) against gdl_swin_pack_matrix per matrix, bit for bit: the [out][in] copy, the transposed copy, the padded biases; and
    the gradients' way back (gdl_swin_pack_batched dir 1) against gdl_swin_unpack_matrix."""
    from gdl.swin import SwinEngine, _Linear

    cfg = fx.SWIN_TINY2
    eng = SwinEngine(cfg, dtype, 2, 2, DEV)
    P = fx.make_state(fx.swin_param_shapes(cfg))
    params = [torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    st = L.cur_stream()
    eng._pack_all(st)
    torch.cuda.synchronize()
    seen = 0
    for o in eng._linears_norms():
        if not isinstance(o, _Linear):
            continue
        w = torch.full_like(o.w, float("nan"))
        wT = torch.full_like(o.wT, float("nan"))
        L.call("gdl_swin_pack_matrix", eng.dt, L.ptr(params[o.w_idx]), L.ptr(w), L.ptr(wT), o.n, o.k, o.nseg, o.nseg_pad, o.kseg, o.kseg_pad, st)
        torch.cuda.synchronize()
        assert torch.equal(w.view(torch.uint8), o.w.view(torch.uint8)) and torch.equal(wT.view(torch.uint8), o.wT.view(torch.uint8))
        assert torch.equal(o.wT, o.w.t())
        if o.b is not None:
            want = torch.zeros_like(o.b).view(-1, o.nseg_pad)
            want[:, :o.nseg] = params[o.b_idx].view(-1, o.nseg)
            assert torch.equal(o.b, want.view(-1))
        # the way back: a padded float32 gradient -> the parameter's shape
        o.dw.copy_(torch.randn(o.dw.shape, device=DEV))
        seen += 1
    grads = [torch.full_like(p, float("nan")) for p in params]
    for o in eng._linears_norms():  # (every bias / LayerNorm gradient buffer defined)
        for t in (getattr(o, "db", None), getattr(o, "dgb", None)):
            if t is not None:
                t.zero_()
    eng._unpack_all(grads, st)
    torch.cuda.synchronize()
    for o in eng._linears_norms():
        if not isinstance(o, _Linear):
            continue
        rt = torch.empty_like(params[o.w_idx])
        L.call("gdl_swin_unpack_matrix", L.ptr(o.dw), L.ptr(rt), o.n, o.k, o.nseg, o.nseg_pad, o.kseg, o.kseg_pad, st)
        torch.cuda.synchronize()
        assert torch.equal(rt, grads[o.w_idx])
    assert seen >= 9


def test_swin_engine_deterministic_and_rebindable():
    cfg = fx.SWIN_TINY2
    dy = np.random.default_rng(5).standard_normal((4, 192), dtype=np.float32)
    y1, g1, eng = _run(cfg, 2, 2, 3, dy, "bf16")
    y2, g2, _ = _run(cfg, 2, 2, 3, dy, "bf16")
    np.testing.assert_array_equal(y1, y2)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k], err_msg=k)
    with pytest.raises(L.GdlError):
        eng.set_params([torch.zeros(1, device=DEV)])


def test_swin_dgl_dropin_step_golden(golden_dir):
    """BASELINE config 5's composition (ResNet18 audio + Swin visual + ConcatFusion_DGL over 512 + C) -- built in the golden
    generator from the reference's own classes -- with the reference's step body (main_dgl.py:97-154) run unchanged on the
    drop-in modules: forward, three CE losses, two-phase backward with the head-gradient drop, clip, SGD."""
    import argparse

    import torch.nn as nn
    from models.basic_model import AVClassifier_DGL_Swin
    from test_step_gpu import _batch, _dropin_checks

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    sc = cfg["swin"]
    args = argparse.Namespace(fusion_method="concat", dataset=cfg["dataset"], modality="full", batch_size=cfg["batch"], pe=0)
    model = AVClassifier_DGL_Swin(args, swin_kwargs=dict(img_size=sc["img"], patch_size=sc["patch"], embed_dim=sc["embed"],
                                                         depths=list(sc["depths"]), num_heads=list(sc["heads"]),
                                                         window_size=sc["window"], mlp_ratio=float(sc["mlp"]),
                                                         drop_path_rate=0.))
    P, Bf = fx.swin_dgl_state(cfg["n_classes"], sc)
    assert [n for n, _ in model.named_parameters()] == list(P)
    res = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in {**P, **Bf}.items()}, strict=False)
    assert not res.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in res.missing_keys)
    model = model.to(DEV)
    model.audio_net.gdl_dtype = model.visual_net.gdl_dtype = "f32"
    model = nn.DataParallel(model, device_ids=[0])  # main_dgl.py:244
    optimizer = torch.optim.SGD(model.parameters(), lr=cfg["lr"], momentum=0.9, weight_decay=1e-4)
    criterion = nn.CrossEntropyLoss()
    model.train()
    spec, image, label = _batch(cfg, 0)
    optimizer.zero_grad()
    out, out_a, out_v = model(spec.unsqueeze(1).float(), image.float())
    loss_v, loss_a, loss_f = criterion(out_v, label), criterion(out_a, label), criterion(out, label)
    ((loss_a + loss_v) * cfg["alpha"]).backward(retain_graph=True)
    for nme, parms in model.named_parameters():
        if 'fusion' in str(nme).split('.')[1]:
            parms.grad = None
    loss_f.backward()
    np.testing.assert_allclose(out_v.detach().cpu().numpy(), g["s0.out_v"], rtol=5e-4, atol=5e-4)
    _dropin_checks(model, optimizer, g, cfg, out, loss_f, loss_a)


def _swin_dgl_model(cfg, dtype):
    import argparse

    from models.basic_model import AVClassifier_DGL_Swin

    sc = cfg["swin"]
    args = argparse.Namespace(fusion_method="concat", dataset=cfg["dataset"], modality="full", batch_size=cfg["batch"], pe=0)
    model = AVClassifier_DGL_Swin(args, swin_kwargs=dict(img_size=sc["img"], patch_size=sc["patch"], embed_dim=sc["embed"],
                                                         depths=list(sc["depths"]), num_heads=list(sc["heads"]),
                                                         window_size=sc["window"], mlp_ratio=float(sc["mlp"]),
                                                         drop_path_rate=0.))
    P, Bf = fx.swin_dgl_state(cfg["n_classes"], sc)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in {**P, **Bf}.items()}, strict=False)
    model = model.to(DEV)
    model.audio_net.gdl_dtype = model.visual_net.gdl_dtype = dtype
    return model


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_swin_early_backward_identical(golden_dir, dtype):
    """The Swin composition's default step form (each encoder's feature gradient from ITS unimodal loss in one launch on its own
    stream -- gdl_head_uni_dfeat_w for the 768-wide Swin features -- and the backward right behind the forward; the fusion head
    off the critical path) leaves the SAME parameters, losses, logits and statistics, bit for bit, as forward -> head -> backward
    (main_dgl.py:97-154)."""
    from gdl.trainer import DGLTrainer
    from test_step_gpu import _batch

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    res = []
    for early in (False, True):
        model = _swin_dgl_model(cfg, dtype)
        model.train()
        tr = DGLTrainer(model, lr=cfg["lr"], alpha=cfg["alpha"], mode="dgl", dtype=dtype, early_backward=early)
        for st in range(3):
            spec, image, label = _batch(cfg, st)
            tr.step(spec, image, label)
        torch.cuda.synchronize()
        r = tr.read()
        res.append((tr.params.clone(), tr.losses.clone(), r["total_norm"], tr.out_a.clone(), tr.out_v.clone(), tr.out.clone(),
                    tr.dfv.clone()))
    for a, b in zip(res[0], res[1]):
        if torch.is_tensor(a):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        else:
            assert a == b


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_swin_dgl_native_step_golden(golden_dir, dtype):
    """DGLTrainer (flat arenas, fused head / losses / optimizer, two chain streams) with the Swin visual branch, two steps,
    against the golden of the composed reference parts."""
    from gdl.trainer import DGLTrainer
    from test_step_gpu import _batch

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    model = _swin_dgl_model(cfg, dtype)
    model.train()
    tr = DGLTrainer(model, lr=cfg["lr"], alpha=cfg["alpha"], mode="dgl", dtype=dtype)
    f32 = dtype == "f32"
    for st in range(cfg["steps"]):
        spec, image, label = _batch(cfg, st)
        tr.step(spec, image, label)
        r = tr.read()
        pre = f"s{st}."
        if st > 0 and not f32:  # (second bf16 step of a B=4 BatchNorm fixture: finiteness only, as for the ResNet fixtures)
            assert np.isfinite(r["out"]).all() and np.isfinite(r["total_norm"])
            continue
        lt = (1e-2 if st else 5e-4) if f32 else 0.2
        for k in ("out", "out_a", "out_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt, atol=lt, err_msg=k)
        for k in ("loss_f", "loss_a", "loss_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt if f32 else 5e-2, atol=lt if f32 else 5e-2, err_msg=k)
        nt = (2e-2 if st else 3e-3) if f32 else 4e-2
        np.testing.assert_allclose(r["total_norm"], g[pre + "total_norm"], rtol=nt)
        np.testing.assert_allclose(r["visual_grad_sum"], g[pre + "visual_grad_sum"], rtol=2 * nt)
        names = [str(n) for n in g[pre + "grad_names"]]
        gn, isnone = g[pre + "grad_norm"], g[pre + "grad_is_none"]
        gt = (6e-2 if st else 1e-2) if f32 else 0.3
        tn = float(g[pre + "total_norm"])
        for i, n in enumerate(names):
            if isnone[i]:
                assert n not in r["grad_norm"]
                continue
            assert abs(r["grad_norm"][n] - gn[i]) <= gt * gn[i] + 1e-5 * min(1.0, 40.0 / tn) * tn, (n, r["grad_norm"][n], gn[i])
    if f32:
        last = f"s{cfg['steps'] - 1}."
        ps, sd = g[last + "param_sums"], model.state_dict()
        for i, n in enumerate(str(x) for x in g[last + "grad_names"]):
            np.testing.assert_allclose(sd[n].double().abs().sum().item(), ps[i][1], rtol=1e-3, err_msg=n)
    acc = tr.valid([_batch(cfg, 1000)])
    assert all(0.0 <= a <= 1.0 for a in acc)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_swin_engine_vs_oracle_16_frames(dtype):
    """Swin-T on 16 frames (B = 8, T = 2) against the CPU oracle: 50 176 stage-1 tokens, so the LayerNorm / column-sum
    kernels run their capped grids with several rows per lane and the reductions fold 512 partial rows -- paths the
    2-frame goldens do not reach.  Also the frame-pooled form ([B, 768] features) the DGL composition uses."""
    from oracle import swin_oracle as so

    cfg, B, T = fx.SWIN_T, 8, 2
    P = fx.make_state(fx.swin_param_shapes(cfg))
    x = fx.swin_input(cfg, B, T, seed=7)
    dy = np.random.default_rng(11).standard_normal((B * T, 768), dtype=np.float32)
    want_y, want_g = so.forward_backward(x, P, cfg, dy)
    y, grads, eng = _run(cfg, B, T, 7, dy, dtype)
    f32 = dtype == "f32"
    ey = _relerr(y, want_y)
    worst, worst_k = 0.0, None
    for k, v in grads.items():
        e = max(_relerr(v, want_g[k]), abs(np.linalg.norm(v.astype(np.float64)) - np.linalg.norm(want_g[k].astype(np.float64))) /
                np.linalg.norm(want_g[k].astype(np.float64)))
        if e > worst:
            worst, worst_k = e, k
    print(f"swin 16 frames {dtype}: features {ey:.2e}, worst gradient {worst:.2e} ({worst_k})")
    assert ey < (5e-6 if f32 else 2e-2), ey
    assert worst < (5e-5 if f32 else 8e-2), (worst_k, worst)
    # pooled over the frames of a sample: the mean of the per-frame features
    xp = torch.from_numpy(x).to(DEV)
    yp = eng.forward(xp, pool_frames=True).cpu().numpy()
    np.testing.assert_allclose(yp, y.reshape(B, T, -1).mean(1), rtol=0, atol=(1e-6 if f32 else 1e-6) * max(1.0, np.abs(y).max()))


def test_swin_engine_config5_size_vs_f64_arbiter():
    """Swin-T at config 5's own size (B = 64, T = 3: 192 frames, 602 112 stage-1 tokens -- every size-gated path of the engine: the
    streaming and the fused-backward Linears, the pipelined LayerNorm, the larger attention chunks) against the float64 CPU
    arbiter (oracle/torch_swin_step.py::swin_features_and_grads: the Swin oracle in double precision over chunks of 16 frames,
    itself pinned to the reference's goldens in tests/test_oracle_golden.py) -- BOTH the float32 exact-parity engine and the
    bf16 one (at this size the two engines used to be compared with each other only).  Features and every
    parameter gradient, element-wise relative to the tensor's largest value and norm against norm; the bf16 run twice
    (graph-free, bit-identical)."""
    from oracle.torch_swin_step import swin_features_and_grads

    cfg, B, T = fx.SWIN_T, 64, 3
    dy = np.random.default_rng(13).standard_normal((B * T, 768), dtype=np.float32)
    P = fx.make_state(fx.swin_param_shapes(cfg))
    want_y, want_g = swin_features_and_grads(fx.swin_input(cfg, B, T, 9), P, cfg, dy, chunk=16)
    res = {}
    for dtype in ("f32", "bf16"):
        y, g, _ = _run(cfg, B, T, 9, dy, dtype)
        torch.cuda.empty_cache()
        ey = _relerr(y, want_y)
        worst, worst_k = 0.0, None
        for k, v in g.items():
            w = want_g[k]
            e = max(_relerr(v, w), abs(np.linalg.norm(v.astype(np.float64)) - np.linalg.norm(w)) / np.linalg.norm(w))
            if e > worst:
                worst, worst_k = e, k
        print(f"swin 192 frames {dtype} vs float64: features {ey:.2e}, worst gradient {worst:.2e} ({worst_k})")
        res[dtype] = (y, g, ey, worst, worst_k)
    # f32: exact-f32 MFMA products summed in fp32 over 192 frames against double precision; bf16: storage rounding
    assert res["f32"][2] < 1e-5 and res["f32"][3] < 2e-4, res["f32"][2:]
    assert res["bf16"][2] < 2e-2 and res["bf16"][3] < 8e-2, res["bf16"][2:]
    y16b, g16b, _ = _run(cfg, B, T, 9, dy, "bf16")
    np.testing.assert_array_equal(res["bf16"][0], y16b)
    for k in g16b:
        np.testing.assert_array_equal(res["bf16"][1][k], g16b[k], err_msg=k)


def test_swin_dgl_step_config5_size_vs_f64_arbiter():
    """BASELINE config 5 as bench.py --workload vggsound_swin times it (B = 64, 129 x 626 spectrograms, 3 frames of 224 x 224,
    309 classes, ResNet18 audio + Swin-T visual + the 512 + 768 concat DGL head): two DGLTrainer steps in the float32
    exact-parity mode AND in bf16 against the same two steps of the float64 CPU arbiter (oracle/torch_swin_step.py::TorchSwinStep,
    pinned to the reference-generated golden in tests/test_oracle_golden.py; main_dgl.py:97-154) on the same seeded weights and
    batches -- logits, the three losses, the clip's total norm, the per-encoder gradient sums, every gradient tensor's norm.
    (this test compared bf16 with the HIP f32 engine only.)  Bounds: float32 logits / losses 5e-4, norms 3e-3
    (SURVEY 8(c)'s fp32 class); bf16 as the ResNet full-size test (logits 3e-2 + 1 % of |logit|, losses 1e-2, total norm 1e-2 --
    doubled in the second step, whose weights already differ by the first step's rounding)."""
    import bench
    from gdl.trainer import DGLTrainer
    from oracle.torch_swin_step import TorchSwinStep

    wl = bench.WORKLOADS["vggsound_swin"]
    B = 64
    g = torch.Generator(device="cpu").manual_seed(5)
    host = [(torch.randn(B, *wl["spec"], generator=g), torch.randn(B, 3, 3, 224, 224, generator=g),
             torch.randint(0, wl["n_classes"], (B,), generator=g)) for _ in range(2)]
    res = {}
    ref_state = None
    for dt in ("f32", "bf16"):
        model, _ = bench.build_model(wl, B, torch.device(DEV))
        if ref_state is None:  # (bench.build_model seeds the initialisation: both models start from these values)
            ref_state = ({k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()},
                         {k: v.detach().cpu().numpy().copy() for k, v in model.named_buffers() if k.startswith("audio_net.")})
        tr = DGLTrainer(model, lr=2e-3, alpha=wl["alpha"], momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype=dt)
        out = []
        for d in host:
            tr.step(*(t.to(DEV) for t in d))
            out.append(tr.read())
        res[dt] = out
        del tr, model
        torch.cuda.empty_cache()
    # (the arbiter's step costs ~3 min of host time at this size: it runs the FIRST step; the second step's bf16 results are held
    # against the float32 engine's, which the first step ties to the arbiter -- measured in round 5 with the arbiter on both steps:
    # f32 logits 3.0e-6 / 1.6e-4, worst gradient tensor 1.8e-3 / 1.2e-2; bf16 1.7e-2 / 3.1e-2, 0.115 / 0.061;
    # GDL_TEST_ARBITER_STEP2=1 reproduces that: the loop below then judges both steps, the second with doubled bounds)
    arb = TorchSwinStep(ref_state[0], ref_state[1], fx.SWIN_T, chunk_samples=4)
    want = [arb.train_step(host[0][0].numpy(), host[0][1].numpy(), host[0][2].numpy(), wl["alpha"], 2e-3)]
    if os.environ.get("GDL_TEST_ARBITER_STEP2") == "1":  # opt-in (3 more minutes of host time): the second step against the arbiter too
        want.append(arb.train_step(host[1][0].numpy(), host[1][1].numpy(), host[1][2].numpy(), wl["alpha"], 2e-3))
    a, b = res["f32"][1], res["bf16"][1]
    for k in ("out", "out_a", "out_v"):
        np.testing.assert_allclose(b[k], a[k], rtol=0, atol=6e-2 * max(1.0, float(np.abs(a[k]).max())), err_msg=f"step 1 {k}")
    for k in ("loss_f", "loss_a", "loss_v"):
        assert abs(b[k] - a[k]) < 3e-2 * max(1.0, abs(a[k])), (k, a[k], b[k])
    assert abs(b["total_norm"] - a["total_norm"]) < 4e-2 * a["total_norm"], (a["total_norm"], b["total_norm"])
    for k in ("audio_grad_sum", "visual_grad_sum"):
        assert abs(b[k] - a[k]) < 6e-2 * abs(a[k]), (k, a[k], b[k])
    for dt in ("f32", "bf16"):
        f32 = dt == "f32"
        for st, (a, b) in enumerate(zip(want, res[dt])):
            k2 = 1.0 if st == 0 else 2.0
            lt, ls, nt, gt = (5e-4, 5e-4, 3e-3, 1e-2) if f32 else (3e-2, 1e-2, 1e-2, 0.12)
            worst = {k: float((np.abs(b[k] - a[k]) / (1.0 + (0.0 if f32 else 1.0 / 3.0) * np.abs(a[k]))).max()) for k in ("out", "out_a", "out_v")}
            worst.update({k: abs(b[k] - a[k]) / max(1.0, abs(a[k])) for k in ("loss_f", "loss_a", "loss_v")})
            worst.update({k: abs(b[k] - a[k]) / abs(a[k]) for k in ("total_norm", "audio_grad_sum", "visual_grad_sum")})
            tn = a["total_norm"]
            rel = {n: abs(b["grad_norm"][n] - w) / max(w, 1e-6 * tn) for n, w in a["grad_norm"].items()}
            worst["grad_norm"], worst["grad_norm_median"] = max(rel.values()), float(np.median(list(rel.values())))
            print(f"config-5 step {st} {dt} vs float64: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
            assert set(b["grad_norm"]) == set(a["grad_norm"])
            for k in ("out", "out_a", "out_v"):
                assert worst[k] <= k2 * lt, (dt, st, k, worst[k])
            for k in ("loss_f", "loss_a", "loss_v"):
                assert worst[k] <= k2 * ls, (dt, st, k, worst[k])
            assert worst["total_norm"] <= k2 * nt and worst["audio_grad_sum"] <= 2 * k2 * nt and worst["visual_grad_sum"] <= 2 * k2 * nt, (dt, st, worst)
            assert worst["grad_norm"] <= k2 * gt, (dt, st, sorted(rel.items(), key=lambda kv: -kv[1])[:5])
            assert worst["grad_norm_median"] <= k2 * (1e-3 if f32 else 2e-2), (dt, st, worst)



def test_swin_trainer_graph_replay_equals_eager(golden_dir):
    """From the third step on SwinEngine replays captured HIP graphs of its forward / backward launch sequences: five steps
    with the replay must be bit-identical to five steps launched eagerly."""
    from gdl.trainer import DGLTrainer
    from test_step_gpu import _batch

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))

    def run(use_graph):
        model = _swin_dgl_model(cfg, "bf16")
        model.train()
        tr = DGLTrainer(model, lr=cfg["lr"], alpha=cfg["alpha"], mode="dgl", dtype="bf16")
        outs = []
        for st in range(5):
            spec, image, label = _batch(cfg, st % 2)
            if st == 0:
                tr._prepare(spec, image)
                tr.eng_v.eng.use_graph = use_graph
            tr.step(spec, image, label)
            r = tr.read()
            outs.append((r["out"].copy(), r["total_norm"], r["loss_f"]))
        captured = sum(1 for v in tr.eng_v.eng._graphs.values() if v["g"] is not None)
        return outs, tr.params.detach().cpu().numpy().copy(), captured

    a, pa, ca = run(True)
    b, pb, cb = run(False)
    assert cb == 0 and ca >= 2, (ca, cb)  # forward and backward graphs were captured and replayed
    for (oa, na, la), (ob, nb, lb) in zip(a, b):
        np.testing.assert_array_equal(oa, ob)
        assert na == nb and la == lb
    np.testing.assert_array_equal(pa, pb)


def test_swin_dropin_stale_forward_is_refused(golden_dir):
    """The drop-in SwinTransformer shares one engine between every forward of a shape (train and eval): a backward through
    an output whose activations a later forward has overwritten must raise (as the ResNet18 mirror does), never return the
    gradients of the other pass."""
    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    net = _swin_dgl_model(cfg, "bf16").visual_net
    net.train()
    sc = cfg["swin"]
    x = torch.randn(2, 3, 2, sc["img"], sc["img"], device=DEV)
    y1 = net(x)
    y1.sum().backward()  # the ordinary order works
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    y1 = net(x)
    with torch.no_grad():
        net(x + 1.0)  # e.g. a validation forward of the same shape in between
    with pytest.raises(L.GdlError, match="another forward"):
        y1.sum().backward()


@pytest.mark.parametrize("cfg_name", ["SWIN_TINY2", "SWIN_T"])
def test_swin_backward_phases_equal_whole(cfg_name):
    """SwinEngine.backward in two phases (1: upstream gradient + final norm + last stage, 2: the rest -- the data-parallel
    schedule that lets the last stage's bucket travel while the rest is differentiated) leaves bit for bit the gradients of the
    one-call backward, and the last stage's / final norm's are already final after phase 1."""
    from gdl.swin import SwinEngine

    cfg = getattr(fx, cfg_name)
    B, T = 2, 2
    eng = SwinEngine(cfg, "bf16", B, T, DEV)
    eng.use_graph = False
    P = fx.make_state(fx.swin_param_shapes(cfg))
    params = [torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    x = torch.from_numpy(fx.swin_input(cfg, B, T, 5)).to(DEV)
    dy = torch.from_numpy(np.random.default_rng(3).standard_normal((B * T, eng.C_out), dtype=np.float32)).to(DEV)
    eng.forward(x)
    whole = [torch.full_like(p, float("nan")) for p in params]
    eng.backward(dy, whole)
    torch.cuda.synchronize()
    eng.forward(x)
    two = [torch.full_like(p, float("nan")) for p in params]
    with pytest.raises(L.GdlError):
        eng.backward(dy, two, phase=2)  # no phase 1 of this forward yet
    eng.backward(dy, two, phase=1)
    torch.cuda.synchronize()
    last = f"layers.{len(cfg['depths']) - 1}."
    names = list(P)
    for n, a, b in zip(names, whole, two):
        if n.startswith(last) or n.startswith("norm."):
            assert torch.equal(a, b), n  # final after phase 1
        else:
            assert torch.isnan(b).all(), n  # untouched so far
    eng.backward(dy, two, phase=2)
    torch.cuda.synchronize()
    for n, a, b in zip(names, whole, two):
        assert torch.equal(a, b), n
