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


def _run(cfg, B, T, seed, dy, dtype):
    from gdl.swin import SwinEngine

    eng = SwinEngine(cfg, dtype, B, T, DEV)
    P = fx.make_state(fx.swin_param_shapes(cfg))
    assert [n for n, _ in eng.param_shapes()] == list(P)
    params = [torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    x = torch.from_numpy(fx.swin_input(cfg, B, T, seed)).to(DEV)
    y = eng.forward(x).clone()
    grads = [torch.full_like(p, float("nan")) for p in params]
    eng.backward(torch.from_numpy(dy).to(DEV), grads)
    torch.cuda.synchronize()
    return y.cpu().numpy(), {k: g.cpu().numpy() for k, g in zip(P, grads)}, eng


def _relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name,cfg_name", [("swin_tiny2_b2", "SWIN_TINY2"), ("swin_t_b1", "SWIN_T")])
def test_swin_engine_golden(golden_dir, name, cfg_name, dtype):
    cfg = getattr(fx, cfg_name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    c = json.loads(str(g["config"]))
    y, grads, _ = _run(cfg, c["batch"], c["frames"], c["seed"], g["dy"], dtype)
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
    assert ey < (2e-5 if f32 else 3e-2), ey
    assert worst < (2e-4 if f32 else 8e-2), (worst_k, worst)


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
