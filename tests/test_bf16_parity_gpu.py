"""bf16 parity at non-chaotic batch sizes: the sum / gated / film DGL heads and the Swin
composition through DGLTrainer at B = 16 against the CPU oracle's step on the same batch and weights -- BOTH steps compared
(the B = 2-4 golden fixtures can only check a second bf16 step for finiteness: their BatchNorms see 16-64 samples) -- and
BASELINE config 5 at its own shapes (VGGSound spectrogram 129 x 626, 309 logits, Swin-T at 224 x 224, T = 3) at the largest
batch whose CPU oracle step stays under a minute.  Reference semantics: /root/reference/main_dgl.py:97-154,
models/fusion_modules.py:16-30,126-178,213-250, models/swin_transformer.py:596-634."""
import argparse

import numpy as np
import pytest
import torch

from gpu_util import DEV, dev

from oracle import fixtures as fx
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _report(tag, r, ref):
    tn = ref["total_norm"]
    w = {k: float((np.abs(r[k] - ref[k]) / (1.0 + np.abs(ref[k]) / 3.0)).max()) for k in ("out", "out_a", "out_v")}
    w.update({k: abs(r[k] - ref[k]) / max(1.0, abs(ref[k])) for k in ("loss_f", "loss_a", "loss_v")})
    w.update({k: abs(r[k] - ref[k]) / ref[k] for k in ("total_norm", "audio_grad_sum", "visual_grad_sum")})
    assert set(r["grad_norm"]) == set(ref["grad_norm"]), set(r["grad_norm"]) ^ set(ref["grad_norm"])
    rel = {n: abs(r["grad_norm"][n] - want) / max(want, 1e-5 * tn) for n, want in ref["grad_norm"].items()}
    wk = max(rel, key=rel.get)
    w["grad_norm"], w["grad_norm_median"] = rel[wk], float(np.median(list(rel.values())))
    print(f"{tag}: " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()) + f" (worst tensor {wk})")
    return w


def _check(w, f32, later, b16=False):
    """f32: SURVEY 8(c)'s 5e-4 / 5e-4 / 3e-3 / 1e-2 (logits, losses, total norm, per-parameter norms).
    bf16, one step at config 5's shapes: SURVEY 8(c)'s bounds unmoved -- logits atol 3e-2 (+ 1 % of |logit|: `_report` scales
    by 1 + |ref| / 3), losses 1e-2, total norm rtol 1e-2, per-parameter norms 0.1 worst / 2e-2 median.
    bf16 at B = 16 (`b16`): SURVEY's numbers were probed at B = 64; a quarter of the samples per BatchNorm statistic costs
    about sqrt(4) in noise.  Measured on an MI355X (round 3, gpurun_out/r3c/parity2.log; sum / gated / film / Swin composition):
      step 0: logits 2.8e-2 / 1.3e-2 / 4.4e-2 / 2.4e-2, total norm <= 8e-3, worst per-parameter norm 0.113 / 0.083 / 0.123 / 0.084,
              median <= 1e-2;
      step 1: logits 8.1e-2 / 3.8e-2 / 9.2e-2 / 6.8e-2, total norm <= 1.5e-2, worst per-parameter norm 0.136 / 0.075 / 0.133 / 0.254,
              median <= 1.7e-2 (the worst tensors are BatchNorm weights / biases of layer 1: heavily cancelling sums of bf16 gradients).
    The bounds below are those with ~1.3x margin: regression guards for both steps where rounds 1-2 could only test the
    second step of their B = 2-4 fixtures for finiteness.
    What they are guards AGAINST (round 4): tools/parity_sources.py emulates an idealised bf16-storage
    implementation -- the float64 step with bf16 rounding at exactly the tensors this library stores in bf16 (weights, inputs,
    convolution outputs, activations, both kinds of gradient), nothing else -- at this batch size for two consecutive steps
    (profiles/r04_parity_two_steps_b16.txt): step 0 logits 2.3e-2, total norm 3.0e-3, worst tensor 0.086, median 6.4e-3; step 1
    logits 6.3e-2, total norm 3.4e-3, worst tensor 0.20 (audio_net.bn1.weight), median 7.3e-3.  The HIP path sits at 1.0-1.5x of
    that on the per-tensor and logit figures and at 2.7-4.5x on the total norm (the emulation keeps fp64 BatchNorm statistics
    and head): the deviations are the size ANY implementation with these storage points has -- they come from the FORWARD
    roundings (conv outputs / activations: ReLU decisions of near-zero pre-activations flip, each flip adds or removes a whole
    gradient element; per-source table at B = 16 / 64: profiles/r04_parity_sources_b16.txt / _b64.txt), not from the gradient
    storage (<= 0.002) -- and the bounds are 1.3-2x above the emulated figures, not a free parameter of last week's run."""
    if f32:
        lt, ls, nt, gt, gm = (1e-2, 1e-2, 2e-2, 6e-2, 1e-2) if later else (5e-4, 5e-4, 3e-3, 1e-2, 1e-3)
    elif b16:
        lt, ls, nt, gt, gm = (0.12, 1e-2, 2e-2, 0.33, 3e-2) if later else (6e-2, 1e-2, 1e-2, 0.16, 2e-2)
    else:
        lt, ls, nt, gt, gm = 3e-2, 1e-2, 1e-2, 0.1, 2e-2
    bad = [(n, w[n], lt) for n in ("out", "out_a", "out_v") if w[n] > lt]
    bad += [(n, w[n], ls) for n in ("loss_f", "loss_a", "loss_v") if w[n] > ls]
    bad += [(n, w[n], b) for n, b in (("total_norm", nt), ("audio_grad_sum", 2 * nt), ("visual_grad_sum", 2 * nt),
                                      ("grad_norm", gt), ("grad_norm_median", gm)) if w[n] > b]
    return bad


@pytest.mark.parametrize("fusion", ["sum", "gated", "film"])
def test_bf16_heads_b16_two_steps_vs_oracle(fusion):
    """B = 16 at CREMA-D's shapes (spectrogram 257 x 188, three frames of 224 x 224): two consecutive bf16 steps of DGLTrainer
    against two steps of the fp32 oracle (momentum and the updated weights included)."""
    from gdl.trainer import DGLTrainer
    from models.basic_model import AVClassifier_DGL
    from test_step_gpu import _load_state

    B, ncls, alpha, lr = 16, 6, 4.0, 2e-3
    shp = dict(spec_hw=(257, 188), frames=3, image_hw=(224, 224))  # CREMA-D's own shapes (BASELINE configs[1])
    P, Bf = fx.model_state(ncls, fusion + "_dgl")
    orc.set_num_threads(64)
    ref = orc.AVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, "dgl")
    args = argparse.Namespace(fusion_method=fusion, dataset="CREMAD", modality="full", batch_size=B)
    model = AVClassifier_DGL(args)
    _load_state(model, {**P, **Bf})
    model = model.to(DEV)
    model.audio_net.gdl_dtype = model.visual_net.gdl_dtype = "bf16"
    model.train()
    tr = DGLTrainer(model, lr=lr, alpha=alpha, dtype="bf16")
    bad = []
    for st in range(2):
        spec, image, label = fx.make_batch(100 + st, B, shp["spec_hw"], shp["frames"], shp["image_hw"], ncls)
        want = ref.train_step(spec, image, label, alpha, lr)
        want["grad_norm"] = {k: float(np.sqrt(orc.sumsq(g))) for k, g in want.pop("grads").items()}
        tr.step(dev(spec), dev(image), torch.from_numpy(label).to(DEV))
        w = _report(f"{fusion} head B=16 bf16 step {st}", tr.read(), want)
        bad += [(st,) + b for b in _check(w, False, st > 0, b16=True)]
    assert not bad, bad


def _swin_model(ncls, sc, dtype, B, drop_path_rate=0.):
    from models.basic_model import AVClassifier_DGL_Swin

    args = argparse.Namespace(fusion_method="concat", dataset="VGGSound" if ncls == 309 else "CREMAD", modality="full",
                              batch_size=B, pe=0)
    model = AVClassifier_DGL_Swin(args, swin_kwargs=dict(img_size=sc["img"], patch_size=sc["patch"], embed_dim=sc["embed"],
                                                         depths=list(sc["depths"]), num_heads=list(sc["heads"]),
                                                         window_size=sc["window"], mlp_ratio=float(sc["mlp"]),
                                                         drop_path_rate=drop_path_rate))
    P, Bf = fx.swin_dgl_state(ncls, sc)
    assert [n for n, _ in model.named_parameters()] == list(P)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in {**P, **Bf}.items()}, strict=False)
    model = model.to(DEV)
    model.audio_net.gdl_dtype = model.visual_net.gdl_dtype = dtype
    model.train()
    return model, P, Bf


def test_swin_composition_b16_bf16_two_steps_vs_oracle():
    """The Swin composition (ResNet18 audio at CREMA-D's 257 x 188 + two-stage Swin at 56 x 56 + ConcatFusion_DGL over 512 + 192) at B = 16, T = 2:
    two bf16 steps of DGLTrainer against oracle/swin_step.py (pinned to the reference golden in tests/test_oracle_golden.py)."""
    from gdl.trainer import DGLTrainer
    from oracle.swin_step import SwinAVModel

    B, ncls, alpha, lr, sc = 16, 6, 4.0, 2e-3, fx.SWIN_TINY2
    model, P, Bf = _swin_model(ncls, sc, "bf16", B)
    orc.set_num_threads(64)
    ref = SwinAVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, sc)
    tr = DGLTrainer(model, lr=lr, alpha=alpha, dtype="bf16")
    bad = []
    for st in range(2):
        spec, image, label = fx.make_batch(200 + st, B, (257, 188), 2, (sc["img"], sc["img"]), ncls)
        want = ref.train_step(spec, image, label, alpha, lr)
        tr.step(dev(spec), dev(image), torch.from_numpy(label).to(DEV))
        w = _report(f"Swin composition B=16 bf16 step {st}", tr.read(), want)
        bad += [(st,) + b for b in _check(w, False, st > 0, b16=True)]
    assert not bad, bad


def test_swin_composition_stochastic_depth_step_vs_oracle():
    """DGLTrainer on the Swin composition built with drop_path_rate = 0.4 (the reference's constructor default is 0.1): two f32
    training steps whose DropPath masks the trainer draws itself (`visual_net.last_drop_scales`), against oracle/swin_step.py
    run with the SAME masks -- SURVEY 8(c)'s f32 bounds; the masks must matter (the same steps without them are elsewhere)."""
    from gdl.trainer import DGLTrainer
    from oracle.swin_step import SwinAVModel

    B, ncls, alpha, lr, sc = 8, 6, 4.0, 2e-3, fx.SWIN_TINY2
    model, P, Bf = _swin_model(ncls, sc, "f32", B, drop_path_rate=0.4)
    orc.set_num_threads(64)
    ref = SwinAVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, sc)
    plain = SwinAVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, sc)
    tr = DGLTrainer(model, lr=lr, alpha=alpha, dtype="f32")
    torch.manual_seed(3)
    bad = []
    for st in range(2):
        spec, image, label = fx.make_batch(300 + st, B, (65, 47), 2, (sc["img"], sc["img"]), ncls)
        tr.step(dev(spec), dev(image), torch.from_numpy(label).to(DEV))
        r = tr.read()
        scales = model.visual_net.last_drop_scales.cpu().numpy()
        assert scales.shape == (4, 2, 2 * B) and (scales[0] == 1).all() and (scales == 0).any()
        want = ref.train_step(spec, image, label, alpha, lr, drop=scales)
        w = _report(f"Swin composition, stochastic depth, f32 step {st}", r, want)
        bad += [(st,) + b for b in _check(w, True, st > 0)]
        if st == 0:
            off = plain.train_step(spec, image, label, alpha, lr)
            assert np.abs(off["out_v"] - want["out_v"]).max() > 1e-3
    assert not bad, bad
    acc = tr.valid([(dev(spec), dev(image), torch.from_numpy(label).to(DEV))])  # eval: DropPath is the identity, nothing is drawn
    assert all(0.0 <= a <= 1.0 for a in acc)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_config5_full_shapes_vs_oracle(dtype):
    """BASELINE configs[4] at its own shapes: VGGSound spectrogram 129 x 626, 309 logits, Swin-T (embed 96, depths 2-2-6-2,
    heads 3-6-12-24, window 7) on T = 3 frames of 224 x 224, B = 8 (24 frames: 75 264 stage-1 tokens, every kernel form of
    bench.py --workload vggsound_swin except the batch count) through DGLTrainer against the CPU oracle's step: all three
    logit sets, the three losses, the pre-clip total norm, the logged sums and the post-clip norm of every gradient tensor."""
    from gdl.trainer import DGLTrainer
    from oracle.swin_step import SwinAVModel

    B, ncls, alpha, lr, sc = 8, 309, 2.0, 2e-3, fx.SWIN_T
    model, P, Bf = _swin_model(ncls, sc, dtype, B)
    orc.set_num_threads(64)
    torch.set_num_threads(64)
    ref = SwinAVModel({k: v.copy() for k, v in P.items()}, {k: np.array(v) for k, v in Bf.items()}, sc)
    spec, image, label = fx.make_batch(300, B, (129, 626), 3, (224, 224), ncls)
    want = ref.train_step(spec, image, label, alpha, lr)
    tr = DGLTrainer(model, lr=lr, alpha=alpha, dtype=dtype)
    tr.step(dev(spec), dev(image), torch.from_numpy(label).to(DEV))
    w = _report(f"config 5 shapes B=8 {dtype}", tr.read(), want)
    bad = _check(w, dtype == "f32", False)
    assert not bad, bad
