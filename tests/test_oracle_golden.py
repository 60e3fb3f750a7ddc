"""Pins the CPU oracle (oracle/) against golden vectors captured from the
imported reference (tests/golden/make_golden.py).  CPU-only."""
import json
import os

import numpy as np
import pytest

from oracle import fixtures as fx
from oracle import oracle as orc

RTOL, ATOL = 2e-4, 2e-5  # fp32 vs fp32, different summation orders


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name,ncls", [("head_dgl_c6", 6), ("head_dgl_c34", 34)])
def test_head_dgl(golden_dir, name, ncls):
    g = _load(golden_dir, name)
    st = fx.make_state({"fusion_module.fc_out.weight": (ncls, 1024), "fusion_module.fc_out.bias": (ncls,)})
    W, b = st["fusion_module.fc_out.weight"], st["fusion_module.fc_out.bias"]
    x_out, y_out, out = orc.concat_dgl_fwd(g["x"], g["y"], W, b)
    np.testing.assert_allclose(x_out, g["x_out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(y_out, g["y_out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)
    dx, dy, dW, db = orc.concat_dgl_bwd(g["x"], g["y"], W, g["g_x_out"], g["g_y_out"], None)
    np.testing.assert_allclose(dx, g["dx"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dy, g["dy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dW, g["dW_uni"], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(db, g["db_uni"], rtol=RTOL, atol=1e-4)
    dx2, dy2, dW2, db2 = orc.concat_dgl_bwd(g["x"], g["y"], W, None, None, g["g_out"])
    assert not dx2.any() and not dy2.any()  # detached input: no gradient reaches the encoders
    np.testing.assert_allclose(dW2, g["dW_f"], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(db2, g["db_f"], rtol=RTOL, atol=1e-4)
    assert int(g["auxi_grad_is_none"]) == 1


def test_head_sum_dgl(golden_dir):
    """SumFusion_DGL (fusion_modules.py:16-30): forward and both backward phases of main_dgl.py:110-122."""
    g = _load(golden_dir, "head_sum_dgl_c6")
    st = fx.make_state({"fusion_module.fc_x.weight": (6, 512), "fusion_module.fc_x.bias": (6,),
                        "fusion_module.fc_y.weight": (6, 512), "fusion_module.fc_y.bias": (6,)})
    Wx, bx, Wy, by = (st["fusion_module." + k] for k in ("fc_x.weight", "fc_x.bias", "fc_y.weight", "fc_y.bias"))
    x_out, y_out, out = orc.sum_dgl_fwd(g["x"], g["y"], Wx, bx, Wy, by)
    for a, k in ((x_out, "x_out"), (y_out, "y_out"), (out, "out")):
        np.testing.assert_allclose(a, g[k], rtol=RTOL, atol=ATOL)
    dx, dy, dWx, dbx, dWy, dby = orc.sum_dgl_bwd(g["x"], g["y"], Wx, Wy, g["g_x_out"], g["g_y_out"], None)
    for a, k in ((dx, "dx"), (dy, "dy"), (dWx, "uni.fc_x.weight"), (dbx, "uni.fc_x.bias"), (dWy, "uni.fc_y.weight"),
                 (dby, "uni.fc_y.bias")):
        np.testing.assert_allclose(a, g[k], rtol=RTOL, atol=1e-4)
    dx2, dy2, dWx, dbx, dWy, dby = orc.sum_dgl_bwd(g["x"], g["y"], Wx, Wy, None, None, g["g_out"])
    assert not dx2.any() and not dy2.any() and int(g["dx_after_f_is_none"]) == 1  # detached: nothing reaches the encoders
    for a, k in ((dWx, "f.fc_x.weight"), (dbx, "f.fc_x.bias"), (dWy, "f.fc_y.weight"), (dby, "f.fc_y.bias")):
        np.testing.assert_allclose(a, g[k], rtol=RTOL, atol=1e-4)


def _close_big(got, g, key, rtol=RTOL, atol=1e-4):
    """Compare with a golden stored whole (small tensors) or as norm + a strided sample (every 97th element; every
    9973rd for tensors above 10 M elements)."""
    if key in g.files:
        np.testing.assert_allclose(got, g[key], rtol=rtol, atol=atol, err_msg=key)
    else:
        step = 9973 if got.size > 10 ** 7 else 97
        np.testing.assert_allclose(np.sqrt((got.astype(np.float64) ** 2).sum()), float(g[key + ".norm"]), rtol=rtol, err_msg=key)
        np.testing.assert_allclose(got.reshape(-1)[::step], g[key + ".sample97"], rtol=rtol, atol=atol, err_msg=key)


def test_head_film_dgl(golden_dir):
    """FiLM_DGL (fusion_modules.py:126-178; a 134 M-parameter bilinear head): forward and both backward phases."""
    g = _load(golden_dir, "head_film_dgl_c6")
    names = ("fc.weight", "fc.bias", "fc_out.weight", "fc_out.bias")
    st = fx.make_state({"fusion_module." + k: s for k, s in zip(names, ((512, 512 * 512), (512,), (6, 512), (6,)))})
    Wfc, bfc, Wo, bo = (st["fusion_module." + k] for k in names)
    ox, oy, out, hidden = orc.film_dgl_fwd(g["x"], g["y"], Wfc, bfc, Wo, bo)
    for a, k in ((ox, "x_out"), (oy, "y_out"), (out, "out")):
        np.testing.assert_allclose(a, g[k], rtol=5e-4, atol=5e-4)
    dx, dy, G = orc.film_dgl_bwd(g["x"], g["y"], Wfc, Wo, hidden, g["g_x_out"], g["g_y_out"], None)
    np.testing.assert_allclose(dx, g["dx"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(dy, g["dy"], rtol=5e-4, atol=5e-4)
    for k in names:
        _close_big(G[k], g, "uni." + k, rtol=5e-4, atol=5e-4)
    dx2, dy2, G2 = orc.film_dgl_bwd(g["x"], g["y"], Wfc, Wo, hidden, None, None, g["g_out"])
    assert not dx2.any() and not dy2.any() and int(g["dx_after_f_is_none"]) == 1
    for k in names:
        _close_big(G2[k], g, "f." + k, rtol=5e-4, atol=5e-4)


def test_head_gated_dgl(golden_dir):
    """GatedFusion_DGL(x_gate=True) (fusion_modules.py:213-250): forward and both backward phases."""
    g = _load(golden_dir, "head_gated_dgl_c6")
    names = ("fc_x.weight", "fc_x.bias", "fc_y.weight", "fc_y.bias", "fc_out.weight", "fc_out.bias")
    st = fx.make_state({"fusion_module." + k: s for k, s in zip(names, ((512, 512), (512,), (512, 512), (512,), (6, 512), (6,)))})
    W1, b1, W2, b2, Wo, bo = (st["fusion_module." + k] for k in names)
    ox, oy, out, hx, hy = orc.gated_dgl_fwd(g["x"], g["y"], W1, b1, W2, b2, Wo, bo)
    for a, k in ((ox, "x_out"), (oy, "y_out"), (out, "out")):
        np.testing.assert_allclose(a, g[k], rtol=RTOL, atol=1e-4)
    dx, dy, G = orc.gated_dgl_bwd(g["x"], g["y"], hx, hy, W1, W2, Wo, g["g_x_out"], g["g_y_out"], None)
    np.testing.assert_allclose(dx, g["dx"], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(dy, g["dy"], rtol=RTOL, atol=1e-4)
    for k in names:
        _close_big(G[k], g, "uni." + k, atol=2e-4)
    dx2, dy2, G2 = orc.gated_dgl_bwd(g["x"], g["y"], hx, hy, W1, W2, Wo, None, None, g["g_out"])
    assert not dx2.any() and not dy2.any() and int(g["dx_after_f_is_none"]) == 1
    for k in names:
        if int(g["f_is_none." + k]):  # fc_x / fc_y: loss_f never reaches them (detached hidden vectors)
            assert not G2[k].any(), k
        else:
            _close_big(G2[k], g, "f." + k)


def test_head_concat(golden_dir):
    g = _load(golden_dir, "head_concat_c6")
    st = fx.make_state({"fusion_module.fc_out.weight": (6, 1024), "fusion_module.fc_out.bias": (6,)})
    W, b = st["fusion_module.fc_out.weight"], st["fusion_module.fc_out.bias"]
    np.testing.assert_allclose(orc.concat_fwd(g["x"], g["y"], W, b), g["out"], rtol=RTOL, atol=ATOL)
    dx, dy, dW, db = orc.concat_bwd(g["x"], g["y"], W, g["g_out"])
    for a, k in ((dx, "dx"), (dy, "dy"), (dW, "dW"), (db, "db")):
        np.testing.assert_allclose(a, g[k], rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize("name,modality", [("enc_audio_tiny", "audio"), ("enc_visual_tiny", "visual")])
def test_encoder(golden_dir, name, modality):
    g = _load(golden_dir, name)
    P = fx.make_state(fx.resnet18_param_shapes("", 1 if modality == "audio" else 3))
    Bf = fx.make_state(fx.resnet18_buffer_shapes(""))
    net = orc.ResNet18(P, Bf, "", modality)
    y = net.forward(g["x"], train=True)
    np.testing.assert_allclose(y, g["y"], rtol=1e-3, atol=1e-4)
    G = net.backward(g["dy"])
    for k in P:
        if "grad." + k in g.files:
            ref = g["grad." + k]
            # norm-wise.  Measured when this fixture was made: against an fp64 run of the reference
            # the oracle agrees to 5e-6 on every tensor, while the fp32 golden itself is 2.7e-3 off
            # for everything upstream of layer2.0.bn1 (one ReLU sign decided differently in fp32).
            # The bound therefore has to admit one such flip: 1 %.
            rel = np.linalg.norm((G[k] - ref).astype(np.float64)) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30)
            assert rel < 1e-2, (k, rel)
        else:
            st = g["gradstat." + k]
            got = np.array([np.sqrt((G[k].astype(np.float64) ** 2).sum()), np.abs(G[k]).mean()])
            np.testing.assert_allclose(got, st, rtol=1e-3, err_msg=k)
            ref = g["gradsample." + k]
            smp = G[k].reshape(-1)[::997]
            rel = np.linalg.norm((smp - ref).astype(np.float64)) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30)
            assert rel < 1e-2, (k, rel)
    for k in Bf:
        np.testing.assert_allclose(np.asarray(Bf[k], dtype=np.float64), g["buf." + k], rtol=1e-4, atol=1e-5, err_msg=k)
    net2 = orc.ResNet18(P, Bf, "", modality)
    np.testing.assert_allclose(net2.forward(g["x"], train=False), g["y_eval"], rtol=1e-3, atol=1e-4)


def check_step_against_golden(g, model_step, cfg, steps, rtol_logits=1e-3, rtol_norm=2e-3, rtol_gn=5e-3):
    """Shared by the oracle test here and the GPU parity tests: `model_step(st)` -> result dict."""
    base = (rtol_logits, rtol_norm, rtol_gn)
    for st in range(steps):
        r = model_step(st)
        pre = f"s{st}."
        # After an update the tiny fixtures (BatchNorm over 16-24 samples in layer4) amplify the
        # step-0 ReLU-flip noise chaotically (measured: logits 3e-3, grad norms 3e-2 at step 1 with
        # step 0 agreeing to 5e-6 / 2e-3); later steps only pin the update rule, loosely.
        rtol_logits, rtol_norm, rtol_gn = base if st == 0 else (1e-2, 1e-2, 6e-2)
        np.testing.assert_allclose(r["out"], g[pre + "out"], rtol=rtol_logits, atol=rtol_logits)
        np.testing.assert_allclose(r["loss_f"], g[pre + "loss_f"], rtol=rtol_logits, atol=rtol_logits)
        if cfg["mode"] == "dgl":
            np.testing.assert_allclose(r["out_a"], g[pre + "out_a"], rtol=rtol_logits, atol=rtol_logits)
            np.testing.assert_allclose(r["out_v"], g[pre + "out_v"], rtol=rtol_logits, atol=rtol_logits)
            np.testing.assert_allclose(r["loss_a"], g[pre + "loss_a"], rtol=rtol_logits, atol=rtol_logits)
            np.testing.assert_allclose(r["loss_v"], g[pre + "loss_v"], rtol=rtol_logits, atol=rtol_logits)
        np.testing.assert_allclose(r["total_norm"], g[pre + "total_norm"], rtol=rtol_norm)
        np.testing.assert_allclose(r["audio_grad_sum"], g[pre + "audio_grad_sum"], rtol=rtol_gn)
        np.testing.assert_allclose(r["visual_grad_sum"], g[pre + "visual_grad_sum"], rtol=rtol_gn)
        names = [str(n) for n in g[pre + "grad_names"]]
        gn, isnone = g[pre + "grad_norm"], g[pre + "grad_is_none"]
        for i, n in enumerate(names):
            if isnone[i]:
                assert n not in r["grads"], n  # fc_auxi: grad stays None
                continue
            got = float(np.sqrt((r["grads"][n].astype(np.float64) ** 2).sum()))
            assert abs(got - gn[i]) <= rtol_gn * gn[i] + 1e-6 * float(g[pre + "total_norm"]), (n, got, gn[i])


@pytest.mark.parametrize("name", ["dgl_tiny_b4", "dgl_tiny_t1_b2", "dgl_cremad_b2", "dgl_ks_b2", "concat_cremad_b2",
                                  "dgl_sum_tiny_b4", "dgl_gated_tiny_b4", "dgl_film_tiny_b4"])
def test_step(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = json.loads(str(g["config"]))
    fusion = cfg.get("fusion", "concat") + "_dgl" if cfg["mode"] == "dgl" else "concat"
    P, Bf = fx.model_state(cfg["n_classes"], fusion)
    model = orc.AVModel(P, Bf, cfg["mode"])

    def step(st):
        spec, image, label = fx.make_batch(cfg["seed"] + st, cfg["batch"], cfg["spec_hw"], cfg["frames"],
                                           cfg["image_hw"], cfg["n_classes"])
        return model.train_step(spec, image, label, cfg["alpha"], cfg["lr"])

    check_step_against_golden(g, step, cfg, cfg["steps"])
    last = f"s{cfg['steps'] - 1}."
    names = [str(n) for n in g[last + "grad_names"]]
    ps = g[last + "param_sums"]
    for i, n in enumerate(names):
        got = np.array([P[n].astype(np.float64).sum(), np.abs(P[n].astype(np.float64)).sum()])
        np.testing.assert_allclose(got[1], ps[i][1], rtol=1e-5 if cfg["steps"] == 1 else 5e-4, err_msg=n)
        np.testing.assert_allclose(P[n].reshape(-1)[:8], g[last + "param_head8"][i][:min(8, P[n].size)], rtol=1e-4,
                                   atol=5e-5 if cfg["steps"] == 1 else 2e-4, err_msg=n)
    for k in Bf:
        np.testing.assert_allclose(np.asarray(Bf[k], dtype=np.float64), g[last + "buf." + k], rtol=1e-3,
                                   atol=1e-5 if cfg["steps"] == 1 else 1e-3, err_msg=k)
    spec, image, label = fx.make_batch(cfg["seed"] + 1000, cfg["batch"], cfg["spec_hw"], cfg["frames"],
                                       cfg["image_hw"], cfg["n_classes"])
    o = model.forward(spec, image, train=False)
    tol = 1e-3 if cfg["steps"] == 1 else 1e-2
    np.testing.assert_allclose(o[0], g["eval.out"], rtol=tol, atol=tol)


def test_sgd_matches_torch():
    """orc_sgd against torch.optim.SGD (the third-party arithmetic behind main_dgl.py:249) over 3 steps."""
    import torch

    r = np.random.default_rng(5)
    p0 = r.standard_normal(1000).astype(np.float32)
    gs = [r.standard_normal(1000).astype(np.float32) for _ in range(3)]
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=2e-3, momentum=0.9, weight_decay=1e-4)
    p, buf = p0.copy(), np.zeros_like(p0)
    for i, g in enumerate(gs):
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        orc.sgd_(p, g, buf, 2e-3, 0.9, 1e-4, i == 0)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ input pipeline
@pytest.mark.parametrize("tag,n_fft,hop", [("cremad", 512, 353), ("ks", 256, 128)])
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_log_spectrogram_golden(golden_dir, tag, n_fft, hop, pad_mode):
    """The float64 restatement of librosa.stft + log against the committed torch.stft vectors (librosa itself is
    neither vendored in the reference nor installed here).  Tolerance: both sides are float64 rounded to float32."""
    g = np.load(os.path.join(golden_dir, "input_pipeline.npz"))
    got = orc.log_spectrogram(g[f"{tag}.wave"], n_fft, hop, pad_mode)
    want = g[f"{tag}.{pad_mode}"]
    assert got.shape == want.shape == (2, n_fft // 2 + 1, 1 + g[f"{tag}.wave"].shape[1] // hop)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_log_spectrogram_vs_torch():
    """Same check live, at the reference's full clip lengths (3 s at 22 050 Hz -> [257, 188]; 5 s at 16 kHz -> [129, 626])."""
    import torch

    rs = np.random.default_rng(5)
    for n_fft, hop, n, shape in ((512, 353, 22050 * 3, (257, 188)), (256, 128, 16000 * 5, (129, 626))):
        wave = (rs.standard_normal((1, n)) * 0.5).astype(np.float32)
        got = orc.log_spectrogram(wave, n_fft, hop, "constant")
        assert got.shape[1:] == shape  # the spectrogram shapes of BASELINE.json's configs
        X = torch.stft(torch.from_numpy(wave).clamp(-1, 1).double(), n_fft, hop_length=hop,
                       window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True, pad_mode="constant",
                       return_complex=True)
        np.testing.assert_allclose(got, torch.log(X.abs() + 1e-7).float().numpy(), rtol=0, atol=2e-6)


def test_normalize_frames_golden(golden_dir):
    """ToTensor + Normalize: bit-exact (the same fp32 operations in the same order)."""
    g = np.load(os.path.join(golden_dir, "input_pipeline.npz"))
    u8 = g["frames.u8"]
    got = orc.normalize_frames(u8.reshape(-1, *u8.shape[2:])).reshape(g["frames.norm"].shape)
    np.testing.assert_array_equal(got, g["frames.norm"])


def test_torch_restatement_matches_oracle():
    """oracle/torch_step.py (the PyTorch-operator restatement bench.py times as the second CPU baseline) against the
    C/numpy oracle on the tiny DGL fixture: same logits, losses and pre-clip gradient norm."""
    from oracle import fixtures as fx
    from oracle import oracle as orc
    from oracle.torch_step import TorchStep

    B, spec_hw, T, img_hw, ncls = 4, (65, 47), 2, (64, 64), 6
    spec, image, label = fx.make_batch(0, B, spec_hw, T, img_hw, ncls)
    P, Bf = fx.model_state(ncls, "concat_dgl")
    ref = orc.AVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, "dgl")
    r = ref.train_step(spec, image, label, 4.0, 2e-3)
    t = TorchStep(P, Bf).train_step(spec, image, label, 4.0, 2e-3)
    for k in ("out", "out_a", "out_v"):
        np.testing.assert_allclose(t[k], r[k], rtol=0, atol=1e-4, err_msg=k)
    for k in ("loss_f", "loss_a", "loss_v"):
        np.testing.assert_allclose(t[k], r[k], rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(t["total_norm"], r["total_norm"], rtol=1e-3)


@pytest.mark.parametrize("name,cfg_name", [("swin_tiny2_b2", "SWIN_TINY2"), ("swin_t_b1", "SWIN_T"), ("swin_tiny2_drop_b3", "SWIN_TINY2")])
def test_swin_oracle_matches_reference_golden(golden_dir, name, cfg_name):
    """oracle/swin_oracle.py (index-list windows, region-id masks; autograd) against features and parameter gradients of
    the imported reference SwinTransformer (swin_transformer.py:486-674; Swin-T settings and a two-stage 56x56 variant).
    `swin_tiny2_drop_b3`: the TRAINING forward with drop_path_rate = 0.3 -- the fixture holds the per-frame DropPath scales
    the reference's blocks drew (four of the six frames lose the second stage's first Mlp branch)."""
    import json

    from oracle import swin_oracle as so

    cfg = getattr(fx, cfg_name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    c = json.loads(str(g["config"]))
    P = fx.make_state(fx.swin_param_shapes(cfg))
    drop = g["drop_scales"] if "drop_scales" in g.files else None
    y, grads = so.forward_backward(fx.swin_input(cfg, c["batch"], c["frames"], c["seed"]), P, cfg, g["dy"], drop=drop)
    assert np.abs(y - g["y"]).max() <= 2e-6 * np.abs(g["y"]).max()
    if drop is not None:  # (the masks matter: without them the features are somewhere else)
        y0, _ = so.forward_backward(fx.swin_input(cfg, c["batch"], c["frames"], c["seed"]), P, cfg, g["dy"])
        assert np.abs(y0 - g["y"]).max() > 1e-2 * np.abs(g["y"]).max()
    assert set(grads) == {k[len("gradstat."):] for k in g.files if k.startswith("gradstat.")}
    for k, v in grads.items():
        want_norm = g["gradstat." + k][0]
        assert abs(np.sqrt((v.astype(np.float64) ** 2).sum()) - want_norm) <= 2e-5 * want_norm, k
        if "grad." + k in g.files:
            np.testing.assert_allclose(v, g["grad." + k], rtol=0, atol=2e-5 * np.abs(g["grad." + k]).max(), err_msg=k)
        else:
            w = g["gradsample." + k]
            np.testing.assert_allclose(v.reshape(-1)[::997], w, rtol=0, atol=2e-5 * np.abs(w).max(), err_msg=k)


def test_swin_step_oracle_matches_reference_golden(golden_dir):
    """oracle/swin_step.py (the composed DGL step: C-oracle ResNet18 audio + torch-autograd Swin oracle + ConcatFusion_DGL over
    512 + C + the step body of main_dgl.py:97-154) against the two-step golden of the composition assembled from the imported
    reference classes (tests/golden/make_golden.py::_SwinDGL).  The GPU tests use this oracle at B = 16 and at config 5's own
    shapes."""
    import json

    from oracle.swin_step import SwinAVModel

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    P, Bf = fx.swin_dgl_state(cfg["n_classes"], cfg["swin"])
    m = SwinAVModel(P, Bf, cfg["swin"])
    for st in range(cfg["steps"]):
        spec, image, label = fx.make_batch(cfg["seed"] + st, cfg["batch"], cfg["spec_hw"], cfg["frames"], cfg["image_hw"],
                                           cfg["n_classes"])
        r = m.train_step(spec, image, label, cfg["alpha"], cfg["lr"])
        pre = f"s{st}."
        lt, nt, gt = (2e-5, 2e-4, 2e-3) if st == 0 else (2e-3, 1e-3, 2e-2)  # (second step: fp32 ReLU-flip noise, as for ResNet)
        for k in ("out", "out_a", "out_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt, atol=lt, err_msg=k)
        for k in ("loss_f", "loss_a", "loss_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt * 10, atol=lt * 10, err_msg=k)
        for k in ("total_norm", "audio_grad_sum", "visual_grad_sum"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=nt, err_msg=k)
        names, gn, isnone = [str(n) for n in g[pre + "grad_names"]], g[pre + "grad_norm"], g[pre + "grad_is_none"]
        for i, n in enumerate(names):
            if isnone[i]:
                assert n not in r["grad_norm"]
            else:
                assert abs(r["grad_norm"][n] - gn[i]) <= gt * gn[i] + 1e-7, (n, r["grad_norm"][n], gn[i])


@pytest.mark.parametrize("name,cfg_name,chunk", [("swin_tiny2_b2", "SWIN_TINY2", 3), ("swin_t_b1", "SWIN_T", 1)])
def test_torch_swin_f64_arbiter_matches_reference_golden(golden_dir, name, cfg_name, chunk):
    """oracle/torch_swin_step.py::swin_features_and_grads -- the float64, frame-chunked form of the Swin oracle that arbitrates the
    config-5 GPU tests at 192 frames -- against features and parameter gradients of the imported reference SwinTransformer
    (swin_transformer.py:486-674).  The chunk size does not divide the frame count on purpose."""
    import json

    from oracle.torch_swin_step import swin_features_and_grads

    cfg = getattr(fx, cfg_name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    c = json.loads(str(g["config"]))
    P = fx.make_state(fx.swin_param_shapes(cfg))
    y, grads = swin_features_and_grads(fx.swin_input(cfg, c["batch"], c["frames"], c["seed"]), P, cfg, g["dy"], chunk=chunk)
    assert y.dtype == np.float64
    assert np.abs(y - g["y"]).max() <= 2e-6 * np.abs(g["y"]).max()  # (the golden is the reference's float32 run)
    for k, v in grads.items():
        want_norm = g["gradstat." + k][0]
        assert abs(np.sqrt((v ** 2).sum()) - want_norm) <= 2e-5 * want_norm, k
        if "grad." + k in g.files:
            np.testing.assert_allclose(v, g["grad." + k], rtol=0, atol=2e-5 * np.abs(g["grad." + k]).max(), err_msg=k)
        else:
            w = g["gradsample." + k]
            np.testing.assert_allclose(v.reshape(-1)[::997], w, rtol=0, atol=2e-5 * np.abs(w).max(), err_msg=k)


def test_torch_swin_step_f64_matches_reference_golden(golden_dir):
    """oracle/torch_swin_step.py::TorchSwinStep (float64; the Swin branch in chunks of 3 of the 4 samples: a no-grad feature pass,
    then every chunk recomputed with autograd) against the two-step golden of the composition assembled from the imported
    reference classes (tests/golden/make_golden.py::_SwinDGL; main_dgl.py:97-154)."""
    import json

    from oracle.torch_swin_step import TorchSwinStep

    g = np.load(os.path.join(golden_dir, "dgl_swin_tiny_b4.npz"))
    cfg = json.loads(str(g["config"]))
    P, Bf = fx.swin_dgl_state(cfg["n_classes"], cfg["swin"])
    m = TorchSwinStep(P, Bf, cfg["swin"], chunk_samples=3)
    for st in range(cfg["steps"]):
        spec, image, label = fx.make_batch(cfg["seed"] + st, cfg["batch"], cfg["spec_hw"], cfg["frames"], cfg["image_hw"],
                                           cfg["n_classes"])
        r = m.train_step(spec, image, label, cfg["alpha"], cfg["lr"])
        pre = f"s{st}."
        lt, nt, gt = (2e-5, 2e-4, 2e-3) if st == 0 else (2e-3, 1e-3, 2e-2)  # (second step: the golden's own fp32 ReLU-flip noise)
        for k in ("out", "out_a", "out_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt, atol=lt, err_msg=k)
        for k in ("loss_f", "loss_a", "loss_v"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=lt * 10, atol=lt * 10, err_msg=k)
        for k in ("total_norm", "audio_grad_sum", "visual_grad_sum"):
            np.testing.assert_allclose(r[k], g[pre + k], rtol=nt, err_msg=k)
        names, gn, isnone = [str(n) for n in g[pre + "grad_names"]], g[pre + "grad_norm"], g[pre + "grad_is_none"]
        for i, n in enumerate(names):
            if isnone[i]:
                assert n not in r["grad_norm"]
            else:
                assert abs(r["grad_norm"][n] - gn[i]) <= gt * gn[i] + 1e-7, (n, r["grad_norm"][n], gn[i])

