"""GPU parity tests, op level: every HIP kernel behind the C ABI against the CPU oracle
(oracle/) on the same seeded inputs.  For bf16 the inputs are pre-rounded to bf16 so that
the comparison isolates the kernel (fp32 accumulation + one output rounding)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

from gdl import _lib as L  # noqa: E402
from gpu_util import (DEV, bf16_round, dev, empty, from_nhwc, gather_table, pack_weight, quant, relerr, to_nhwc,  # noqa: E402
                      tol)

DTS = [L.GDL_F32, L.GDL_BF16]
rng = np.random.default_rng(2024)


def _conv_case(N, C, H, W, K, R, stride, pad, dt):
    x = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    w = quant((rng.standard_normal((K, C, R, R), dtype=np.float32) * np.sqrt(2.0 / (C * R * R))).astype(np.float32), dt)
    return x, w


CONV_SHAPES = [
    # N, C, H, W, K, R, stride, pad      -> tile config exercised (fwd)
    (2, 64, 17, 13, 64, 3, 1, 1),     # 64x64, ragged M tail, odd spatial dims
    (3, 64, 20, 18, 128, 3, 2, 1),    # 64x64, stride 2
    (2, 64, 9, 6, 128, 1, 2, 0),      # 1x1 stride 2 (downsample)
    (1, 128, 7, 5, 256, 3, 1, 1),     # 64x64, two K-steps per tap
    (10, 64, 33, 31, 128, 3, 1, 1),   # 128x64, ragged tail
    (3, 64, 120, 115, 64, 3, 1, 1),   # 256x64, ragged tail
    (4, 64, 112, 112, 128, 3, 1, 1),  # 256x64, two N-tiles
    (2, 256, 14, 14, 512, 3, 2, 1),   # stride 2, C=256
    (2, 64, 65, 47, 128, 3, 2, 1),    # stride 2 at odd dims (the audio layer-2 geometry): the 9-tap stride-2 weight gradient's edges
    (4, 64, 56, 56, 64, 3, 1, 1),     # 64 -> 64 channels: the weights-stationary persistent kernel (bf16), 98 tiles
    (3, 64, 65, 47, 64, 3, 1, 1),     # the same with odd spatial dims and a ragged last tile (M = 9165)
    (2, 64, 33, 157, 64, 3, 1, 1),    # the same on a wide image (Kinetics-Sounds audio layer 1): ONE slab buffer
    # xcd_linear (common.h) with fewer outer units than XCDs -- the stride-1 forward / data-gradient counterparts of the
    # weight-gradient shapes below, on the slab kernel (>= 128 channels) and on the flat kernel
    (24, 128, 17, 12, 128, 3, 1, 1),  # slab kernel: 26 M-tiles of 192 x one N-tile
    (40, 256, 9, 6, 256, 3, 1, 1),    # slab kernel: 12 M-tiles x 2 N-tiles (a tile's two N-tiles straddle an XCD boundary)
    (3, 256, 9, 6, 512, 3, 1, 1),     # slab kernel: ONE ragged M-tile x 4 N-tiles -- fewer tiles than XCDs
    (1, 128, 17, 12, 256, 1, 2, 0),   # flat kernel: one M-tile, four N-tiles
    (64, 128, 28, 28, 128, 3, 1, 1),  # slab kernel, several tiles per persistent block (round 6), 262 tiles
    (20, 256, 14, 14, 256, 3, 1, 1),  # the same with two N-tiles per M-tile and three channel chunks' slab reloads
    (7, 256, 13, 11, 256, 3, 1, 1),   # odd height and width, M = 1001 (ragged last tile), two N-tiles
    (5, 128, 31, 29, 128, 3, 1, 1),   # odd dims at the widest slab of the 192-row tile that still fits 32 KiB (W = 29: 252 rows)
]


def _check_shape_list(name, shapes):
    """Collection-time guard: a tuple that slipped into the trailing comment of the line above it is silently never run
    (round 5 lost a weight-gradient shape that way).  Every source line of the list must carry exactly one tuple, in code."""
    import inspect
    import re
    import sys
    body = inspect.getsource(sys.modules[__name__]).split(name + " = [", 1)[1].split("\n]\n", 1)[0]
    code_rows = 0
    for ln in body.splitlines():
        code, _, comment = ln.partition("#")
        code_rows += code.count("(")
        assert not re.search(r"\(\s*\d+\s*,\s*\d+\s*,\s*\d+\s*,", comment), f"{name}: a shape tuple inside a comment: {ln.strip()}"
    assert code_rows == len(shapes), f"{name}: {code_rows} tuples in the source, {len(shapes)} parsed"


_check_shape_list("CONV_SHAPES", CONV_SHAPES)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_fwd(shape, dt):
    N, C, H, W, K, R, stride, pad = shape
    x, w = _conv_case(N, C, H, W, K, R, stride, pad, dt)
    ref = orc.conv2d_fwd(x, w, stride, pad)
    P, Q = ref.shape[2], ref.shape[3]
    xd = to_nhwc(x, dt)
    krsc, _ = pack_weight(w, dt)
    y = empty((N, P, Q, K), dt)
    tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    part = torch.full((tiles, K, 2), float("nan"), device=DEV)
    tab = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, R, R, stride, pad)
    L.call("gdl_conv_fwd", dt, L.ptr(xd), L.ptr(krsc), L.ptr(y), L.ptr(part), L.ptr(tab), N, H, W, C, K, R, R, stride, pad,
           L.cur_stream())
    torch.cuda.synchronize()
    got = from_nhwc(y)
    assert relerr(got, ref) < tol(dt, 2e-6, 3e-3), relerr(got, ref)
    np.testing.assert_allclose(got, ref, rtol=tol(dt, 1e-4, 2e-2), atol=tol(dt, 1e-5, 2e-2))
    # BatchNorm partials: sums of the STORED values
    s = part.double().sum(0).cpu().numpy()
    g64 = got.astype(np.float64)
    np.testing.assert_allclose(s[:, 0], g64.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1], (g64 ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("with_addend", [False, True])
def test_conv_dgrad(shape, dt, with_addend):
    N, C, H, W, K, R, stride, pad = shape
    _, w = _conv_case(N, C, H, W, K, R, stride, pad, dt)
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    ref = orc.conv2d_bwd_data(dy, w, (N, C, H, W), stride, pad)
    add = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    if with_addend:
        ref = ref + add
    _, crsk = pack_weight(w, dt)
    dyd = to_nhwc(dy, dt)
    dx = to_nhwc(add, dt) if with_addend else empty((N, H, W, C), dt)
    tab = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, R, R, stride, pad)
    L.call("gdl_conv_dgrad", dt, L.ptr(dyd), L.ptr(crsk), L.ptr(dx), L.ptr(dx) if with_addend else None, L.ptr(tab), N, H,
           W, C, K, R, R, stride, pad, L.cur_stream())
    torch.cuda.synchronize()
    got = from_nhwc(dx)
    assert relerr(got, ref) < tol(dt, 2e-6, 4e-3), relerr(got, ref)


DS_SHAPES = [
    # N, C (block input), H, W, K (block output): conv1 3x3/2 pad 1 and the 1x1/2 shortcut of a downsample block
    (3, 64, 20, 18, 128),   # even dims
    (2, 64, 65, 47, 128),   # odd dims (the audio layer-2 geometry): the four parity classes differ in size
    (2, 256, 14, 14, 512),  # four K-steps per tap
    (5, 128, 7, 9, 256),    # tiny planes, ragged classes
]
_check_shape_list("DS_SHAPES", DS_SHAPES)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", DS_SHAPES)
@pytest.mark.parametrize("with_bits", [False, True])
def test_conv_dgrad_ds(shape, dt, with_bits):
    """dx = conv1^T(dy) + downsample^T(dy_ds) in one launch == the two data gradients of the oracle, summed."""
    N, C, H, W, K = shape
    _, w = _conv_case(N, C, H, W, K, 3, 2, 1, dt)
    wds = quant(rng.standard_normal((K, C, 1, 1), dtype=np.float32) * 0.2, dt)
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    dyd = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    ref = orc.conv2d_bwd_data(dy, w, (N, C, H, W), 2, 1) + orc.conv2d_bwd_data(dyd, wds, (N, C, H, W), 2, 0)
    _, crsk = pack_weight(w, dt)
    _, ck = pack_weight(wds, dt)
    dx = empty((N, H, W, C), dt)
    bits = None
    if with_bits:  # sign bits of a random "block input" (one byte per 16-byte vector)
        z = rng.standard_normal((N, C, H, W), dtype=np.float32)
        ref = ref * (z > 0)
        per = 16 // dx.element_size()
        zb = (torch.from_numpy(np.ascontiguousarray(z.transpose(0, 2, 3, 1))).reshape(-1, per) > 0).to(torch.int32)
        bits = (zb << torch.arange(per, dtype=torch.int32)).sum(1).to(torch.uint8).to(DEV)
    tab = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, 3, 3, 2, 1)
    dy_d, dyd_d = to_nhwc(dy, dt), to_nhwc(dyd, dt)
    L.call("gdl_conv_dgrad_ds", dt, L.ptr(dy_d), L.ptr(crsk), L.ptr(dyd_d), L.ptr(ck), L.ptr(dx),
           L.ptr(bits) if with_bits else None, L.ptr(tab), N, H, W, C, K, L.cur_stream())
    torch.cuda.synchronize()
    got = from_nhwc(dx)
    assert relerr(got, ref) < tol(dt, 2e-6, 4e-3), relerr(got, ref)


WGRAD_SHAPES = [
    (2, 64, 17, 13, 64, 3, 1, 1),    # 64x64 tiles
    (3, 64, 20, 18, 128, 3, 2, 1),   # TK=128, TC=64
    (2, 128, 9, 6, 64, 3, 1, 1),     # TK=64, TC=128
    (2, 128, 12, 10, 256, 3, 1, 1),  # 128x128
    (2, 64, 9, 6, 128, 1, 2, 0),     # 1x1 stride 2
    (3, 64, 56, 56, 64, 3, 1, 1),    # many splits; slab ring (W >= 24)
    (1, 64, 5, 24, 64, 3, 1, 1),     # narrowest ring geometry, M = 120 (ragged last stage)
    (1, 64, 5, 23, 64, 3, 1, 1),     # widest plain double buffer below it
    (2, 64, 3, 63, 128, 3, 1, 1),    # widest ring geometry (2W+2 = 128)
    (2, 64, 2, 64, 64, 3, 1, 1),     # just past it: plain
    (1, 128, 40, 47, 64, 3, 1, 1),   # audio layer-1 width: the ring wraps several times per slice
    (1, 64, 9, 157, 64, 3, 1, 1),    # Kinetics-Sounds audio layer-1 width: the 512-row ring
    (2, 64, 3, 191, 128, 3, 1, 1),   # widest 512-row ring geometry (2W+2 = 384)
    (1, 64, 3, 192, 64, 3, 1, 1),    # just past it: per-tap kernel
    (24, 128, 17, 12, 128, 3, 1, 1), # 3 pixel slices x 4 tiles: fewer slices than XCDs, a slice's tiles on two XCDs (xcd_linear)
    (40, 256, 9, 6, 256, 3, 1, 1),   # one slice of 16 tiles dealt to eight XCDs
    (24, 128, 17, 12, 256, 1, 2, 0), # per-tap kernel (1x1 stride 2), split count not a multiple of 8 (the round-up went in round 5)
    (6, 128, 17, 12, 256, 3, 2, 1),  # per-tap kernel (3x3 stride 2) with fewer pixel slices than XCDs
]
_check_shape_list("WGRAD_SHAPES", WGRAD_SHAPES)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_conv_wgrad(shape, dt):
    N, C, H, W, K, R, stride, pad = shape
    x, w = _conv_case(N, C, H, W, K, R, stride, pad, dt)
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    ref = orc.conv2d_bwd_weight(dy, x, w.shape, stride, pad)
    xd, dyd = to_nhwc(x, dt), to_nhwc(dy, dt)
    nbytes = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((K, C, R, R), float("nan"), device=DEV)
    tab = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, R, R, stride, pad)
    L.call("gdl_conv_wgrad", dt, L.ptr(dyd), L.ptr(xd), L.ptr(dw), L.ptr(tab), N, H, W, C, K, R, R, stride, pad, L.ptr(ws),
           nbytes, L.cur_stream())
    torch.cuda.synchronize()
    got = dw.cpu().numpy()
    assert relerr(got, ref) < 2e-5, relerr(got, ref)  # fp32 accumulation of (for bf16: exact) products


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [("audio", 2, 1, 1, 65, 47), ("visual", 2, 3, 2, 40, 36), ("visual_odd", 1, 3, 3, 33, 29),
                                  # output rows of >= 64 pixels: the bf16 weight gradient runs on the row-slab kernel
                                  ("row64", 1, 3, 2, 12, 128), ("row95_odd_h", 2, 1, 1, 9, 190), ("row135", 2, 3, 1, 7, 270)])
def test_stem_direct(case, dt):
    """7x7/2 stem as an implicit GEMM over the padded NHWC4 input (what the engine runs): forward with BatchNorm
    partials and weight gradient against the oracle's direct conv."""
    _, B, Cin, T, H, W = case
    lib = L.load()
    x = rng.standard_normal((B, Cin, T, H, W), dtype=np.float32)
    w = (rng.standard_normal((64, Cin, 7, 7), dtype=np.float32) * 0.1).astype(np.float32)
    xq, wq = quant(x, dt), quant(w, dt)  # the padded copy / the packed weights round to the storage type
    x4 = np.ascontiguousarray(xq.transpose(0, 2, 1, 3, 4)).reshape(B * T, Cin, H, W)
    ref = orc.conv2d_fwd(x4, wq, 2, 3)
    n_img, P, Q = B * T, ref.shape[2], ref.shape[3]
    M = n_img * P * Q
    st = L.cur_stream()
    xp = torch.empty(lib.gdl_stem_pad_bytes(dt, n_img, H, W), dtype=torch.uint8, device=DEV)
    wp = torch.empty(lib.gdl_stem_weight_bytes(dt), dtype=torch.uint8, device=DEV)
    tab = torch.empty(lib.gdl_stem_table_bytes(n_img, H, W), dtype=torch.uint8, device=DEV)
    y = empty((n_img, P, Q, 64), dt)
    tiles = lib.gdl_stem_conv_bn_tiles(dt, n_img, H, W)
    part = torch.full((tiles, 64, 2), float("nan"), device=DEV)
    xd, wd = dev(x), dev(w)
    L.call("gdl_stem_pad", dt, L.ptr(xd), L.ptr(xp), B, Cin, T, H, W, st)
    L.call("gdl_pack_stem_rows", dt, L.ptr(wd), L.ptr(wp), Cin, st)
    L.call("gdl_stem_build_table", dt, n_img, H, W, L.ptr(tab), st)
    L.call("gdl_stem_conv_fwd", dt, L.ptr(xp), L.ptr(wp), L.ptr(y), L.ptr(part), L.ptr(tab), n_img, H, W, Cin, st)
    torch.cuda.synchronize()
    got = from_nhwc(y)
    assert relerr(got, ref) < tol(dt, 2e-6, 3e-3), relerr(got, ref)
    # partials are the per-channel sum / sum of squares of the STORED values
    s = part.sum(0).cpu().numpy()
    gq = got.transpose(1, 0, 2, 3).reshape(64, -1).astype(np.float64)
    assert np.allclose(s[:, 0], gq.sum(1), rtol=1e-4, atol=1e-2) and np.allclose(s[:, 1], (gq * gq).sum(1), rtol=1e-4, atol=1e-2)
    # run to run: outputs and partial rows bit-identical (the persistent kernel's waves walk their stages in a fixed order)
    y2 = empty((n_img, P, Q, 64), dt)
    part2 = torch.full((tiles, 64, 2), float("nan"), device=DEV)
    L.call("gdl_stem_conv_fwd", dt, L.ptr(xp), L.ptr(wp), L.ptr(y2), L.ptr(part2), L.ptr(tab), n_img, H, W, Cin, st)
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.uint8), y.view(torch.uint8)) and torch.equal(part2.view(torch.int32), part.view(torch.int32))
    dy = quant(rng.standard_normal((n_img, 64, P, Q), dtype=np.float32), dt)
    refw = orc.conv2d_bwd_weight(dy, x4, (64, Cin, 7, 7), 2, 3)
    dyd = to_nhwc(dy, dt)
    nbytes = lib.gdl_stem_conv_wgrad_workspace_bytes(n_img, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, Cin, 7, 7), float("nan"), device=DEV)
    L.call("gdl_stem_conv_wgrad", dt, L.ptr(dyd), L.ptr(xp), L.ptr(dw), L.ptr(tab), n_img, H, W, Cin, L.ptr(ws), nbytes, st)
    torch.cuda.synchronize()
    assert relerr(dw.cpu().numpy(), refw) < 2e-5, relerr(dw.cpu().numpy(), refw)


@pytest.mark.parametrize("case", [("row64", 1, 3, 2, 12, 128), ("row95_odd_h", 2, 1, 1, 9, 190), ("row135", 2, 3, 1, 7, 270),
                                  ("cremad_audio_rows", 2, 1, 1, 33, 188), ("visual_rows", 1, 3, 2, 20, 224)])
def test_stem_bwd_fused(case):
    """gdl_stem_bwd_fused (round 5): max-pool gather + ReLU mask + BatchNorm-backward apply + the 7x7/2 stem's weight gradient in
    one launch, the gradient of the stem output never stored -- BIT-IDENTICAL to gdl_maxpool_bn_bwd_apply followed by
    gdl_stem_conv_wgrad (same element arithmetic and order, same stages, same fold), and against the oracle's
    maxpool_bwd -> relu_bwd -> bn_bwd -> conv2d_bwd_weight chain (/root/reference/models/backbone.py:97-106 differentiated)."""
    _, B, Cin, T, H, W = case
    dt = L.GDL_BF16
    lib = L.load()
    assert lib.gdl_stem_bwd_fused_ok(dt, W) == 1
    C = 64
    x = rng.standard_normal((B, Cin, T, H, W), dtype=np.float32)
    w = (rng.standard_normal((64, Cin, 7, 7), dtype=np.float32) * 0.1).astype(np.float32)
    xq, wq = quant(x, dt), quant(w, dt)
    x4 = np.ascontiguousarray(xq.transpose(0, 2, 1, 3, 4)).reshape(B * T, Cin, H, W)
    y = quant(orc.conv2d_fwd(x4, wq, 2, 3), dt)  # the stem output as stored
    n_img, P, Q = y.shape[0], y.shape[2], y.shape[3]
    gamma = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    gamma[::7] *= -1
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    rm, rv = np.zeros(C, np.float32), np.ones(C, np.float32)
    _, mean, invstd = orc.bn_fwd_train(y, gamma, beta, rm, rv)
    sc = (gamma * invstd).astype(np.float32)
    sh = (beta - mean * gamma * invstd).astype(np.float32)
    st = L.cur_stream()
    xp = torch.empty(lib.gdl_stem_pad_bytes(dt, n_img, H, W), dtype=torch.uint8, device=DEV)
    tab = torch.empty(lib.gdl_stem_table_bytes(n_img, H, W), dtype=torch.uint8, device=DEV)
    xd = dev(x)
    L.call("gdl_stem_pad", dt, L.ptr(xd), L.ptr(xp), B, Cin, T, H, W, st)
    L.call("gdl_stem_build_table", dt, n_img, H, W, L.ptr(tab), st)
    yd = to_nhwc(y, dt)
    scd, shd, smd, srd, gd = dev(sc), dev(sh), dev(mean.astype(np.float32)), dev(invstd.astype(np.float32)), dev(gamma)
    PP, QQ = (P - 1) // 2 + 1, (Q - 1) // 2 + 1
    out, ym = empty((n_img, PP, QQ, C), dt), empty((n_img, PP, QQ, C), dt)
    ix = torch.empty((n_img, PP, QQ, C), dtype=torch.uint8, device=DEV)
    L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(out), L.ptr(ix), L.ptr(ym), n_img, P, Q, C, st)
    dout = quant(rng.standard_normal((n_img, C, PP, QQ), dtype=np.float32), dt)
    doutd = to_nhwc(dout, dt)
    Mp, M = n_img * PP * QQ, n_img * P * Q
    blocks = lib.gdl_bn_bwd_blocks(Mp, C)
    bpart = torch.empty((blocks, C, 2), device=DEV)
    dg, db, coef = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2 * C, device=DEV)
    L.call("gdl_bn_bwd_reduce", dt, L.ptr(doutd), L.ptr(ym), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), 1, L.ptr(bpart),
           Mp, C, st)
    L.call("gdl_bn_bwd_finalize", L.ptr(bpart), blocks, C, float(M), L.ptr(dg), L.ptr(db), L.ptr(coef), st)
    # two launches: dy0 stored, then the weight gradient
    dyd = empty((n_img, P, Q, C), dt)
    L.call("gdl_maxpool_bn_bwd_apply", dt, L.ptr(doutd), L.ptr(ix), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd),
           L.ptr(gd), L.ptr(coef), L.ptr(dyd), n_img, P, Q, C, st)
    nbytes = lib.gdl_stem_conv_wgrad_workspace_bytes(n_img, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw2 = torch.full((64, Cin, 7, 7), float("nan"), device=DEV)
    L.call("gdl_stem_conv_wgrad", dt, L.ptr(dyd), L.ptr(xp), L.ptr(dw2), L.ptr(tab), n_img, H, W, Cin, L.ptr(ws), nbytes, st)
    # one launch
    ws1 = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw1 = torch.full((64, Cin, 7, 7), float("nan"), device=DEV)
    L.call("gdl_stem_bwd_fused", dt, L.ptr(doutd), L.ptr(ix), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), L.ptr(gd),
           L.ptr(coef), L.ptr(xp), L.ptr(dw1), n_img, H, W, Cin, L.ptr(ws1), nbytes, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dw1).all()
    assert torch.equal(dw1.view(torch.int32), dw2.view(torch.int32)), float((dw1 - dw2).abs().max())
    # run to run
    dw3 = torch.full((64, Cin, 7, 7), float("nan"), device=DEV)
    L.call("gdl_stem_bwd_fused", dt, L.ptr(doutd), L.ptr(ix), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), L.ptr(gd),
           L.ptr(coef), L.ptr(xp), L.ptr(dw3), n_img, H, W, Cin, L.ptr(ws1), nbytes, st)
    torch.cuda.synchronize()
    assert torch.equal(dw1.view(torch.int32), dw3.view(torch.int32))
    # the oracle's chain on the device's own arg-max codes (bf16 ties may pick another position than the oracle's first maximum)
    a = quant(np.maximum(y * sc[None, :, None, None] + sh[None, :, None, None], 0).astype(np.float32), dt)
    code = ix.cpu().numpy().astype(np.int64)  # [n][PP][QQ][C], code = r*3 + s inside the window
    d_a = np.zeros((n_img, P, Q, C), np.float32)
    nn_, pp_, qq_, cc_ = np.meshgrid(np.arange(n_img), np.arange(PP), np.arange(QQ), np.arange(C), indexing="ij")
    np.add.at(d_a, (nn_, 2 * pp_ - 1 + code // 3, 2 * qq_ - 1 + code % 3, cc_), np.transpose(dout, (0, 2, 3, 1)))
    d_a = np.ascontiguousarray(np.transpose(d_a, (0, 3, 1, 2)))
    dyref, _, _ = orc.bn_bwd(orc.relu_bwd(d_a, a), y, gamma, mean, invstd)
    refw = orc.conv2d_bwd_weight(quant(dyref.astype(np.float32), dt), x4, (64, Cin, 7, 7), 2, 3)
    assert relerr(dw1.cpu().numpy(), refw) < 2e-2, relerr(dw1.cpu().numpy(), refw)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C,N,H,W", [(64, 3, 9, 7), (128, 2, 17, 12), (512, 4, 3, 2)])
def test_bn_forward_backward(C, N, H, W, dt):
    x = quant(rng.standard_normal((N, C, H, W), dtype=np.float32) * 1.7 + 0.3, dt)
    gamma = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    rm = (0.05 * rng.standard_normal(C)).astype(np.float32)
    rv = (1 + 0.1 * np.abs(rng.standard_normal(C))).astype(np.float32)
    rm_ref, rv_ref = rm.copy(), rv.copy()
    yref, mean, invstd = orc.bn_fwd_train(x, gamma, beta, rm_ref, rv_ref)
    aref = np.maximum(yref, 0)
    M = N * H * W
    xd = to_nhwc(x, dt)
    lib = L.load()
    tiles = lib.gdl_bn_stats_tiles(M)
    part = torch.empty((tiles, C, 2), device=DEV)
    st = L.cur_stream()
    L.call("gdl_bn_stats", dt, L.ptr(xd), L.ptr(part), M, C, st)
    g_d, b_d, rm_d, rv_d = dev(gamma), dev(beta), dev(rm), dev(rv)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    sm, sr, sc, sh = (torch.empty(C, device=DEV) for _ in range(4))
    L.call("gdl_bn_finalize_train", L.ptr(part), tiles, C, float(M), L.ptr(g_d), L.ptr(b_d), 1e-5, 0.1, L.ptr(rm_d),
           L.ptr(rv_d), L.ptr(nbt), L.ptr(sm), L.ptr(sr), L.ptr(sc), L.ptr(sh), st)
    a = empty((N, H, W, C), dt)
    L.call("gdl_bn_act", dt, L.ptr(xd), L.ptr(sc), L.ptr(sh), None, None, None, 1, L.ptr(a), M, C, st)
    torch.cuda.synchronize()
    np.testing.assert_allclose(sm.cpu().numpy(), mean, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sr.cpu().numpy(), invstd, rtol=1e-4)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm_ref, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv_ref, rtol=1e-4)
    assert int(nbt.item()) == 1
    assert relerr(from_nhwc(a), aref) < tol(dt, 2e-6, 3e-3)
    # backward through relu + bn
    da = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    dxref, dgref, dbref = orc.bn_bwd(orc.relu_bwd(da, aref), x, gamma, mean, invstd)
    dad = to_nhwc(da, dt)
    blocks = lib.gdl_bn_bwd_blocks(M, C)
    bpart = torch.empty((blocks, C, 2), device=DEV)
    dg, db, coef = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2 * C, device=DEV)
    L.call("gdl_bn_bwd_reduce", dt, L.ptr(dad), L.ptr(xd), L.ptr(sc), L.ptr(sh), L.ptr(sm), L.ptr(sr), 1, L.ptr(bpart),
           M, C, st)
    L.call("gdl_bn_bwd_finalize", L.ptr(bpart), blocks, C, float(M), L.ptr(dg), L.ptr(db), L.ptr(coef), st)
    dx = empty((N, H, W, C), dt)
    L.call("gdl_bn_bwd_apply", dt, L.ptr(dad), L.ptr(xd), L.ptr(sc), L.ptr(sh), L.ptr(sm), L.ptr(sr), L.ptr(g_d),
           L.ptr(coef), 1, L.ptr(dx), M, C, st)
    torch.cuda.synchronize()
    # relu masks can differ where |bn output| ~ 0: allow a small norm-wise slack
    assert relerr(dg.cpu().numpy(), dgref) < 2e-3
    assert relerr(db.cpu().numpy(), dbref) < 2e-3
    assert relerr(from_nhwc(dx), dxref) < tol(dt, 2e-3, 6e-3)


@pytest.mark.parametrize("dt", DTS)
def test_bn_act_residual_and_relu_bwd(dt):
    N, C, H, W = 2, 128, 5, 7
    M = N * H * W
    y = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    r = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    sc, sh, rsc, rsh = (rng.standard_normal(C).astype(np.float32) for _ in range(4))
    yd, rd = to_nhwc(y, dt), to_nhwc(r, dt)
    scd, shd, rscd, rshd = dev(sc), dev(sh), dev(rsc), dev(rsh)
    st = L.cur_stream()
    bc = lambda v: v[None, :, None, None]
    for mode in (1, 2):
        ref = y * bc(sc) + bc(sh) + (r if mode == 1 else r * bc(rsc) + bc(rsh))
        ref = np.maximum(ref, 0)
        out = empty((N, H, W, C), dt)
        L.call("gdl_bn_act", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(rd), L.ptr(rscd) if mode == 2 else None,
               L.ptr(rshd) if mode == 2 else None, 1, L.ptr(out), M, C, st)
        torch.cuda.synchronize()
        assert relerr(from_nhwc(out), ref) < tol(dt, 2e-6, 3e-3)
    dz = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    dzd = to_nhwc(dz, dt)
    L.call("gdl_relu_bwd", dt, L.ptr(dzd), L.ptr(out), L.ptr(dzd), M * C, st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(from_nhwc(dzd), np.where(from_nhwc(out) > 0, dz, 0).astype(np.float32))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,H,W", [(2, 33, 24), (3, 16, 16), (1, 7, 9)])
def test_maxpool(N, H, W, dt):
    C = 64
    y = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    sc = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    sc[::7] *= -1  # negative scales: max and BN do not commute
    sh = (0.2 * rng.standard_normal(C)).astype(np.float32)
    a = np.maximum(y * sc[None, :, None, None] + sh[None, :, None, None], 0).astype(np.float32)
    a = quant(a, dt)
    pref, idx = orc.maxpool_fwd(a)
    P, Q = pref.shape[2], pref.shape[3]
    yd = to_nhwc(y, dt)
    out = empty((N, P, Q, C), dt)
    ix = torch.empty((N, P, Q, C), dtype=torch.uint8, device=DEV)
    st = L.cur_stream()
    scd, shd = dev(sc), dev(sh)  # keep alive: the caching allocator reuses freed temporaries at once
    ym = empty((N, P, Q, C), dt)
    L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(out), L.ptr(ix), L.ptr(ym), N, H, W, C, st)
    torch.cuda.synchronize()
    got = from_nhwc(out)
    np.testing.assert_allclose(got, pref, rtol=tol(dt, 1e-5, 1e-2), atol=tol(dt, 1e-6, 1e-2))
    # ymax = the raw y at the position the DEVICE chose (bit-exact gather)
    code = ix.cpu().numpy().astype(np.int64)  # [N,P,Q,C], r*3+s
    pp, qq = np.meshgrid(np.arange(P), np.arange(Q), indexing="ij")
    ih = np.clip(2 * pp[None, :, :, None] - 1 + code // 3, 0, H - 1)
    iw = np.clip(2 * qq[None, :, :, None] - 1 + code % 3, 0, W - 1)
    y_nhwc = np.transpose(y, (0, 2, 3, 1))
    want = y_nhwc[np.arange(N)[:, None, None, None], ih, iw, np.arange(C)[None, None, None, :]]
    np.testing.assert_array_equal(np.transpose(from_nhwc(ym), (0, 2, 3, 1)), want)
    out2, ix2 = empty((N, P, Q, C), dt), torch.empty_like(ix)  # without ymax: same values and indices
    L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(out2), L.ptr(ix2), None, N, H, W, C, st)
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.equal(ix2, ix)
    dout = quant(rng.standard_normal((N, C, P, Q), dtype=np.float32), dt)
    # reference backward with the argmax recomputed from the DEVICE output's own choice is not
    # available; where `a` has no exact ties inside a window both agree, so compare sums per
    # window owner: use the oracle on `a` (ties only at relu zeros, which carry zero gradient
    # after the relu mask applied below)
    dref = orc.maxpool_bwd(dout, idx, a.shape)
    dxd = empty((N, H, W, C), dt)
    doutd = to_nhwc(dout, dt)
    L.call("gdl_maxpool_bwd", dt, L.ptr(doutd), L.ptr(ix), L.ptr(dxd), N, H, W, C, st)
    torch.cuda.synchronize()
    mask = a > 0
    gotb = from_nhwc(dxd)
    if dt == L.GDL_F32:
        np.testing.assert_allclose(gotb * mask, dref * mask, rtol=1e-5, atol=1e-6)
    else:
        # bf16: ties between equal rounded neighbours may route to another position than the oracle's first maximum, so
        # the exact check is against the DEVICE's own routing: scatter dout by the stored window codes (r*3+s) in float64
        # and round once -- every input position must agree to one bf16 ulp (up to four gradients meet in one position)
        want = np.zeros((N, H, W, C), np.float64)
        dn = np.transpose(dout, (0, 2, 3, 1)).astype(np.float64)
        nn_, cc_ = np.arange(N)[:, None, None, None], np.arange(C)[None, None, None, :]
        ih_raw, iw_raw = 2 * pp[None, :, :, None] - 1 + code // 3, 2 * qq[None, :, :, None] - 1 + code % 3
        assert ih_raw.min() >= 0 and ih_raw.max() < H and iw_raw.min() >= 0 and iw_raw.max() < W  # codes point inside the image
        np.add.at(want, (np.broadcast_to(nn_, code.shape), ih_raw, iw_raw, np.broadcast_to(cc_, code.shape)), dn)
        got_nhwc = np.transpose(gotb, (0, 2, 3, 1)).astype(np.float64)
        np.testing.assert_allclose(got_nhwc, want, rtol=2.0 ** -7, atol=1e-6)
        np.testing.assert_allclose(gotb.sum((2, 3)), quant(dref, dt).sum((2, 3)), rtol=2e-2, atol=0.1)  # and the totals vs the oracle


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,H,W", [(2, 33, 24), (3, 16, 16), (1, 7, 9)])
def test_stem_pool_bn_backward(N, H, W, dt):
    """The stem's backward without the gathered gradient: bn_bwd_reduce over the pooled (dout, ymax), finalize with
    the stem-output count, maxpool_bn_bwd_apply -- against maxpool_bwd -> relu_bwd -> bn_bwd of the oracle
    (/root/reference/models/backbone.py:104-106 differentiated)."""
    C = 64
    y = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    gamma = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    gamma[::7] *= -1
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    rm, rv = np.zeros(C, np.float32), np.ones(C, np.float32)
    yref, mean, invstd = orc.bn_fwd_train(y, gamma, beta, rm, rv)
    sc = (gamma * invstd).astype(np.float32)
    sh = (beta - mean * gamma * invstd).astype(np.float32)
    a = quant(np.maximum(y * sc[None, :, None, None] + sh[None, :, None, None], 0).astype(np.float32), dt)
    pref, idx = orc.maxpool_fwd(a)
    P, Q = pref.shape[2], pref.shape[3]
    dout = quant(rng.standard_normal((N, C, P, Q), dtype=np.float32), dt)
    d_a = orc.maxpool_bwd(dout, idx, a.shape)
    dyref, dgref, dbref = orc.bn_bwd(orc.relu_bwd(d_a, a), y, gamma, mean, invstd)
    lib = L.load()
    st = L.cur_stream()
    yd, doutd = to_nhwc(y, dt), to_nhwc(dout, dt)
    scd, shd, smd, srd, gd = dev(sc), dev(sh), dev(mean.astype(np.float32)), dev(invstd.astype(np.float32)), dev(gamma)
    out, ym = empty((N, P, Q, C), dt), empty((N, P, Q, C), dt)
    ix = torch.empty((N, P, Q, C), dtype=torch.uint8, device=DEV)
    L.call("gdl_bn_relu_maxpool_fwd", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(out), L.ptr(ix), L.ptr(ym), N, H, W, C, st)
    Mp, M = N * P * Q, N * H * W
    blocks = lib.gdl_bn_bwd_blocks(Mp, C)
    bpart = torch.empty((blocks, C, 2), device=DEV)
    dg, db, coef = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2 * C, device=DEV)
    L.call("gdl_bn_bwd_reduce", dt, L.ptr(doutd), L.ptr(ym), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), 1, L.ptr(bpart),
           Mp, C, st)
    L.call("gdl_bn_bwd_finalize", L.ptr(bpart), blocks, C, float(M), L.ptr(dg), L.ptr(db), L.ptr(coef), st)
    dyd = empty((N, H, W, C), dt)
    L.call("gdl_maxpool_bn_bwd_apply", dt, L.ptr(doutd), L.ptr(ix), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd),
           L.ptr(gd), L.ptr(coef), L.ptr(dyd), N, H, W, C, st)
    torch.cuda.synchronize()
    # (bf16: ties between equal rounded neighbours may route to another position than the oracle's first maximum)
    assert relerr(dg.cpu().numpy(), dgref) < tol(dt, 1e-4, 2e-2)
    assert relerr(db.cpu().numpy(), dbref) < tol(dt, 1e-4, 2e-2)
    assert relerr(from_nhwc(dyd), dyref) < tol(dt, 1e-4, 0.15)
    # and against the unfused device kernels fed by the same device indices: gather -> reduce -> finalize -> apply
    g0 = empty((N, H, W, C), dt)
    L.call("gdl_maxpool_bwd", dt, L.ptr(doutd), L.ptr(ix), L.ptr(g0), N, H, W, C, st)
    b2 = lib.gdl_bn_bwd_blocks(M, C)
    bp2 = torch.empty((b2, C, 2), device=DEV)
    dg2, db2, coef2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2 * C, device=DEV)
    L.call("gdl_bn_bwd_reduce", dt, L.ptr(g0), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), 1, L.ptr(bp2), M, C, st)
    L.call("gdl_bn_bwd_finalize", L.ptr(bp2), b2, C, float(M), L.ptr(dg2), L.ptr(db2), L.ptr(coef2), st)
    dy2 = empty((N, H, W, C), dt)
    L.call("gdl_bn_bwd_apply", dt, L.ptr(g0), L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(smd), L.ptr(srd), L.ptr(gd), L.ptr(coef2),
           1, L.ptr(dy2), M, C, st)
    torch.cuda.synchronize()
    assert relerr(dg.cpu().numpy(), dg2.cpu().numpy()) < tol(dt, 1e-5, 5e-3)
    assert relerr(db.cpu().numpy(), db2.cpu().numpy()) < tol(dt, 1e-5, 5e-3)
    assert relerr(from_nhwc(dyd), from_nhwc(dy2)) < tol(dt, 1e-5, 6e-3)


@pytest.mark.parametrize("dt", DTS)
def test_avgpool(dt):
    B, T, C, H, W = 3, 2, 512, 3, 2
    x = quant(rng.standard_normal((B * T, C, H, W), dtype=np.float32), dt)
    ref = orc.avgpool_fwd(x, B, T)
    feat = torch.empty((B, C), device=DEV)
    st = L.cur_stream()
    xd = to_nhwc(x, dt)
    L.call("gdl_avgpool_fwd", dt, L.ptr(xd), L.ptr(feat), B, T, H * W, C, st)
    torch.cuda.synchronize()
    np.testing.assert_allclose(feat.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    df = rng.standard_normal((B, C), dtype=np.float32)
    dx = empty((B * T, H, W, C), dt)
    dfd = dev(df)
    L.call("gdl_avgpool_bwd", dt, L.ptr(dfd), L.ptr(dx), B, T, H * W, C, st)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(dx), orc.avgpool_bwd(df, x.shape, B, T)) < tol(dt, 1e-6, 3e-3)


def test_layout_roundtrip():
    for dt in DTS:
        x = quant(rng.standard_normal((3, 70, 5, 9), dtype=np.float32), dt)
        xd = dev(x)
        t = empty((3, 5, 9, 70), dt)
        back = torch.empty((3, 70, 5, 9), device=DEV)
        st = L.cur_stream()
        L.call("gdl_nchw_f32_to_nhwc", dt, L.ptr(xd), L.ptr(t), 3, 5, 9, 70, st)
        L.call("gdl_nhwc_to_nchw_f32", dt, L.ptr(t), L.ptr(back), 3, 5, 9, 70, st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(back.cpu().numpy(), x)
        np.testing.assert_array_equal(from_nhwc(t), x)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 64, 17, 13, 64, 3, 1, 1), (3, 64, 20, 18, 128, 3, 2, 1), (4, 64, 56, 56, 64, 3, 1, 1),
                                   (2, 64, 9, 6, 128, 1, 2, 0)])
def test_relu_bits_and_masked_dgrad(shape, dt):
    """gdl_bn_act_bits stores the sign bits of relu(bn(y) + res); gdl_conv_dgrad_relu zeroes the data gradient where
    they are 0: together the ReLU backward of a BasicBlock output (backbone.py:65-66) without a separate masking pass --
    against dgrad + relu_bwd of the oracle."""
    N, C, H, W, K, R, stride, pad = shape
    st = L.cur_stream()
    # the tensor whose gradient the dgrad produces: z = relu(y*sc + sh + res), NHWC [N,H,W,C]
    y = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    r = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    sc, sh = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    bc = lambda v: v[None, :, None, None]
    zref = np.maximum(y * bc(sc) + bc(sh) + r, 0)
    M = N * H * W
    epc = 8 if dt == L.GDL_BF16 else 4
    z = empty((N, H, W, C), dt)
    bits = torch.zeros(M * C // epc, dtype=torch.uint8, device=DEV)
    yd, rd, scd, shd = to_nhwc(y, dt), to_nhwc(r, dt), dev(sc), dev(sh)
    L.call("gdl_bn_act_bits", dt, L.ptr(yd), L.ptr(scd), L.ptr(shd), L.ptr(rd), None, None, L.ptr(z), L.ptr(bits), M, C, st)
    torch.cuda.synchronize()
    zg = from_nhwc(z)
    assert relerr(zg, zref) < tol(dt, 2e-6, 3e-3)
    want_bits = (np.transpose(zg, (0, 2, 3, 1)).reshape(-1, epc) > 0)
    got_bits = ((bits.cpu().numpy()[:, None] >> np.arange(epc)[None, :]) & 1).astype(bool)
    np.testing.assert_array_equal(got_bits, want_bits)
    # masked data gradient (+ addend) against the oracle
    w = quant((rng.standard_normal((K, C, R, R), dtype=np.float32) * np.sqrt(2.0 / (C * R * R))).astype(np.float32), dt)
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    add = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    ref = orc.conv2d_bwd_data(dy, w, (N, C, H, W), stride, pad) + add
    ref = np.where(zg > 0, ref, 0).astype(np.float32)
    _, crsk = pack_weight(w, dt)
    dyd = to_nhwc(dy, dt)
    dx = to_nhwc(add, dt)
    tab = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, R, R, stride, pad)
    L.call("gdl_conv_dgrad_relu", dt, L.ptr(dyd), L.ptr(crsk), L.ptr(dx), L.ptr(dx), L.ptr(bits), L.ptr(tab), N, H, W, C, K, R, R,
           stride, pad, st)
    torch.cuda.synchronize()
    got = from_nhwc(dx)
    assert relerr(got, ref) < tol(dt, 2e-6, 4e-3), relerr(got, ref)
    assert np.all(got[zg <= 0] == 0)


def test_conv_run_to_run_determinism():
    """Race screen (found a real one once: packed-f32 BatchNorm sums): every conv op of a few
    geometries, four runs each with fresh NaN-poisoned outputs and other kernels in between, must be
    bit-identical -- outputs, BatchNorm partials, data and weight gradients."""
    dt = L.GDL_BF16
    st = L.cur_stream()
    for (N, C, H, W, K, R, stride, pad) in [(16, 64, 65, 47, 128, 3, 2, 1), (16, 64, 65, 47, 128, 1, 2, 0),
                                             (24, 64, 56, 56, 64, 3, 1, 1), (48, 512, 7, 7, 512, 3, 1, 1)]:
        P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
        dy = torch.randn(N, P, Q, K, device=DEV).to(torch.bfloat16)
        wk = torch.randn(K, R, R, C, device=DEV).to(torch.bfloat16)
        wc = torch.randn(C, R, R, K, device=DEV).to(torch.bfloat16)
        tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
        nb = L.load().gdl_conv_wgrad_workspace_bytes(dt, N, H, W, C, K, R, R, stride, pad)
        tf = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, R, R, stride, pad)
        tdg = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, R, R, stride, pad)
        first = None
        for rep in range(4):
            y = torch.full((N, P, Q, K), float("nan"), device=DEV, dtype=torch.bfloat16)
            dx = torch.full((N, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
            dw = torch.full((K, C, R, R), float("nan"), device=DEV)
            part = torch.full((tiles, K, 2), float("nan"), device=DEV)
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV).random_()
            L.call("gdl_conv_fwd", dt, L.ptr(x), L.ptr(wk), L.ptr(y), L.ptr(part), L.ptr(tf), N, H, W, C, K, R, R, stride,
                   pad, st)
            L.call("gdl_conv_dgrad", dt, L.ptr(dy), L.ptr(wc), L.ptr(dx), None, L.ptr(tdg), N, H, W, C, K, R, R, stride,
                   pad, st)
            L.call("gdl_conv_wgrad", dt, L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(tf), N, H, W, C, K, R, R, stride, pad,
                   L.ptr(ws), nb, st)
            torch.cuda.synchronize()
            cur = (y.view(torch.int16).clone(), part.view(torch.int32).clone(), dx.view(torch.int16).clone(),
                   dw.view(torch.int32).clone())
            assert not torch.isnan(part).any() and not torch.isnan(dw).any()
            if first is None:
                first = cur
                # the partials must also be the exact sums of the stored outputs
                yf = y.double().view(-1, K)
                np.testing.assert_allclose(part.double().sum(0)[:, 0].cpu().numpy(), yf.sum(0).cpu().numpy(), rtol=1e-5,
                                           atol=1e-2)
            else:
                for a_, b_, nm in zip(first, cur, ("y", "bn partials", "dx", "dw")):
                    assert torch.equal(a_, b_), (nm, (N, C, H, W, K, R, stride))


@pytest.mark.parametrize("shape", [(192, 128, 28, 28, 256, 3, 2, 1), (192, 128, 28, 28, 128, 3, 1, 1),
                                   (192, 256, 14, 14, 256, 3, 1, 1), (192, 64, 56, 56, 128, 1, 2, 0)])
def test_bn_sums_determinism_full_size(shape):
    """The statistics of the convolution epilogues at the FULL visual shapes of the B = 64 step (192 images), six runs each:
    the forward's BatchNorm partial rows and the data gradient's BatchNorm-backward sums (ReLU bits, two partners where the tile
    has them) must be bit-identical.  These are the shapes (two convolution waves per SIMD) where a library built WITH the SLP
    vectoriser differs from run to run -- whole waves off by a few elements' worth -- because of one packed-f32 instruction form
    (csrc/Makefile, tools/check_isa.sh); the small shapes of test_conv_run_to_run_determinism do not show it.  The engine's
    int64-accumulator path adds these same per-tile sums and is covered by test_step_gpu.py::test_full_size_properties."""
    N, C, H, W, K, R, stride, pad = shape
    dt = L.GDL_BF16
    st = L.cur_stream()
    lib = L.load()
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(N, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(N, P, Q, K, device=DEV, generator=g).to(torch.bfloat16)
    wk = torch.randn(K, R, R, C, device=DEV, generator=g).to(torch.bfloat16)
    wc = torch.randn(C, R, R, K, device=DEV, generator=g).to(torch.bfloat16)
    y1 = torch.randn(N, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    y2 = torch.randn(N, H, W, C, device=DEV, generator=g).to(torch.bfloat16)
    bits = torch.randint(0, 256, (N * H * W * C // 8,), device=DEV, generator=g, dtype=torch.int32).to(torch.uint8)
    mean, rstd = torch.randn(C, device=DEV, generator=g) * 0.3, torch.rand(C, device=DEV, generator=g) + 0.5
    tiles = lib.gdl_conv_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    btiles = lib.gdl_conv_dgrad_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    tf = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, R, R, stride, pad)
    tdg = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, R, R, stride, pad)
    two = R == 3 and stride == 1
    first = None
    for rep in range(6):
        y = torch.full((N, P, Q, K), float("nan"), device=DEV, dtype=torch.bfloat16)
        dx = torch.full((N, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        part = torch.full((tiles, K, 2), float("nan"), device=DEV)
        bp1 = torch.full((btiles, C, 2), float("nan"), device=DEV)
        bp2 = torch.full((btiles, C, 2), float("nan"), device=DEV)
        L.call("gdl_conv_fwd", dt, L.ptr(x), L.ptr(wk), L.ptr(y), L.ptr(part), L.ptr(tf), N, H, W, C, K, R, R, stride, pad, st)
        L.call("gdl_conv_dgrad_bn", dt, L.ptr(dy), L.ptr(wc), L.ptr(dx), None, L.ptr(bits), L.ptr(tdg), N, H, W, C, K, R, R, stride,
               pad, L.ptr(y1), L.ptr(mean), L.ptr(rstd), L.ptr(bp1), L.ptr(y2) if two else None, L.ptr(mean) if two else None,
               L.ptr(rstd) if two else None, L.ptr(bp2) if two else None, st)
        torch.cuda.synchronize()
        assert not torch.isnan(part).any() and not torch.isnan(bp1).any() and (not two or not torch.isnan(bp2).any())
        cur = (y.view(torch.int16).clone(), part.view(torch.int32).clone(), dx.view(torch.int16).clone(),
               bp1.view(torch.int32).clone(), bp2.view(torch.int32).clone())
        if first is None:
            first = cur
            # and the forward partials are the sums of the stored outputs
            yf = y.double().view(-1, K)
            ps = part.double().sum(0)
            assert torch.allclose(ps[:, 0], yf.sum(0), rtol=1e-5, atol=0.5)
            assert torch.allclose(ps[:, 1], (yf * yf).sum(0), rtol=1e-5)
        else:
            for a_, b_, nm in zip(first, cur, ("y", "bn partials", "dx", "bn-backward sums", "bn-backward sums (second partner)")):
                if nm.endswith("(second partner)") and not two:
                    continue
                assert torch.equal(a_, b_), (nm, shape, int((a_ != b_).sum()))


BW_SHAPES = [
    # N, C, H, W, K, R, stride, pad, mask ("bits": ReLU through relu_bits, None), two partners
    (2, 64, 17, 13, 64, 3, 1, 1, "bits", False),      # small slab / flat tile, ragged last tile
    (24, 64, 56, 56, 64, 3, 1, 1, "bits", False),     # 64 -> 64 channels: the persistent kernel, one partial row per block
    (24, 64, 56, 56, 64, 3, 1, 1, None, False),
    (8, 128, 28, 28, 128, 3, 1, 1, "bits", True),     # 128-channel slab tiles, two BatchNorms fed by the same gradient
    (16, 64, 33, 24, 128, 3, 2, 1, "bits", False),    # permuted stride-2 data gradient (rows scattered through orow)
    (48, 512, 7, 7, 512, 3, 1, 1, "bits", True),      # the 8-wave 128 x 128 tile
    (64, 128, 28, 28, 128, 3, 1, 1, "bits", True),    # persistent slab kernel (round 6), 192-row tiles: 262 tiles, one N-tile
    (40, 256, 14, 14, 256, 3, 1, 1, "bits", False),   # the same with two N-tiles (a block keeps its channels) and four chunks
    (30, 128, 33, 24, 128, 3, 1, 1, None, False),     # its 128-row tiles, ragged last tile, no ReLU mask
    (7, 256, 13, 11, 256, 3, 1, 1, "bits", True),      # odd dims, ragged last tile, two partners (the RICH instantiation)
    (2, 64, 9, 6, 128, 1, 2, 0, None, False),         # 1x1 stride 2
]
_check_shape_list("BW_SHAPES", BW_SHAPES)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", BW_SHAPES)
def test_conv_dgrad_bn_sums(shape, dt):
    """gdl_conv_dgrad_bn: the data gradient whose epilogue also leaves the BatchNorm-backward sums of the stored gradient against
    the partner tensor(s) -- the stored tensor against the oracle's data gradient (+ mask), the sums against a float64
    restatement over the STORED values (what gdl_bn_bwd_reduce computes in its own pass, backbone.py:45-48,57 through autograd)."""
    N, C, H, W, K, R, stride, pad, mask, two = shape
    st = L.cur_stream()
    lib = L.load()
    _, w = _conv_case(N, C, H, W, K, R, stride, pad, dt)
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = quant(rng.standard_normal((N, K, P, Q), dtype=np.float32), dt)
    ref = orc.conv2d_bwd_data(dy, w, (N, C, H, W), stride, pad)
    y = quant(rng.standard_normal((N, C, H, W), dtype=np.float32) * 1.5 + 0.3, dt)
    y2 = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    mean, rstd = (rng.standard_normal(C) * 0.3).astype(np.float32), (0.5 + rng.random(C)).astype(np.float32)
    mean2, rstd2 = (rng.standard_normal(C) * 0.3).astype(np.float32), (0.5 + rng.random(C)).astype(np.float32)
    sc, sh = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    bc = lambda v: v[None, :, None, None]
    M = N * H * W
    epc = 8 if dt == L.GDL_BF16 else 4
    bits = None
    if mask == "bits":  # sign bits of relu(y * sc + sh), as gdl_bn_act_bits leaves them
        keep = (y * bc(sc) + bc(sh)) > 0
        kb = np.transpose(keep, (0, 2, 3, 1)).reshape(-1, epc)
        bits = torch.from_numpy((kb * (1 << np.arange(epc))[None, :]).sum(1).astype(np.uint8)).to(DEV)
        ref = np.where(keep, ref, 0)
    _, crsk = pack_weight(w, dt)
    dyd, yd, y2d = to_nhwc(dy, dt), to_nhwc(y, dt), to_nhwc(y2, dt)
    dx = empty((N, H, W, C), dt)
    tab = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, R, R, stride, pad)
    tiles = lib.gdl_conv_dgrad_bn_tiles(dt, N, H, W, C, K, R, R, stride, pad)
    part = torch.full((tiles, C, 2), float("nan"), device=DEV)
    part2 = torch.full((tiles, C, 2), float("nan"), device=DEV)
    md, rd, m2d, r2d = dev(mean), dev(rstd), dev(mean2), dev(rstd2)  # (kept alive across the launch)
    L.call("gdl_conv_dgrad_bn", dt, L.ptr(dyd), L.ptr(crsk), L.ptr(dx), None, L.ptr(bits) if bits is not None else None,
           L.ptr(tab), N, H, W, C, K, R, R, stride, pad, L.ptr(yd), L.ptr(md), L.ptr(rd), L.ptr(part),
           L.ptr(y2d) if two else None, L.ptr(m2d) if two else None, L.ptr(r2d) if two else None,
           L.ptr(part2) if two else None, st)
    torch.cuda.synchronize()
    got = from_nhwc(dx)
    assert relerr(got, ref) < tol(dt, 2e-6, 4e-3), relerr(got, ref)
    if mask:
        # (exactly where the bit is clear -- the reference itself can be an exact 0.0 elsewhere by cancellation)
        bad = np.argwhere(~keep & (got != 0))
        assert len(bad) == 0, (len(bad), bad[:6].tolist(), [float(got[tuple(b)]) for b in bad[:6]])
    # The sums: over the STORED gradient (what a separate gdl_bn_bwd_reduce pass would read back) -- or, round 4, where the tile is
    # staged as fp32 (bf16 launches of the 64-channel persistent kernel and of the flat kernel's 64-channel-wide tiles), over the
    # masked fp32 accumulators BEFORE their rounding to bf16, i.e. over the oracle's fp32 gradient.  Exactly one of the two must hold
    # at the tight tolerance (the two differ by the random walk of the bf16 roundings, ~1e-6 of sum |g|, well above it).
    s = part.double().sum(0).cpu().numpy()
    assert np.isfinite(s).all()
    s2 = part2.double().sum(0).cpu().numpy() if two else None

    def sums_ok(g64):
        want1 = g64.sum((0, 2, 3))
        want2 = (g64 * (y.astype(np.float64) - bc(mean)) * bc(rstd)).sum((0, 2, 3))
        scale_ = np.abs(g64).sum((0, 2, 3)).max()
        ok = np.allclose(s[:, 0], want1, rtol=1e-4, atol=2e-6 * scale_) and np.allclose(s[:, 1], want2, rtol=1e-4, atol=1e-5 * scale_)
        if two:
            want3 = (g64 * (y2.astype(np.float64) - bc(mean2)) * bc(rstd2)).sum((0, 2, 3))
            ok = ok and np.allclose(s2[:, 0], want1, rtol=1e-4, atol=2e-6 * scale_) and np.allclose(s2[:, 1], want3, rtol=1e-4, atol=1e-5 * scale_)
        return ok, float(np.abs(s[:, 0] - want1).max() / scale_), float(np.abs(s[:, 1] - want2).max() / scale_)

    stored, unrounded = sums_ok(got.astype(np.float64)), sums_ok(ref.astype(np.float64))
    assert stored[0] or unrounded[0], ("stored", stored, "fp32 accumulators", unrounded)
    if dt == L.GDL_BF16 and C == 64 and N * H * W >= 64 * 128 and R == 3 and stride == 1:
        assert unrounded[0], ("the 64-channel persistent kernel sums its fp32 accumulators", unrounded)


SPLIT_SHAPES = [
    # N, C, H, W, K  (3x3 stride 1 pad 1; bf16): few output tiles against a deep reduction -> the slab kernel splits K
    (64, 512, 9, 6, 512),     # audio layer 4 at B = 64: 27 x 4 tiles, 4-way
    (48, 512, 7, 7, 512),     # a visual layer-4 shape at B = 16
    (16, 256, 17, 12, 256),   # audio layer 3 at B = 16
    (5, 512, 9, 6, 512),      # ragged last M-tile
]
_check_shape_list("SPLIT_SHAPES", SPLIT_SHAPES)


@pytest.mark.parametrize("shape", SPLIT_SHAPES)
def test_conv_split_k(shape):
    """Split-K of the slab convolutions (gdl_conv_fwd_split / gdl_conv_dgrad_bn_split): the same results as the unsplit launches up
    to fp32 summation order -- output tensor, BatchNorm statistics rows, BatchNorm-backward sums with addend (in place) + ReLU bits +
    two partners -- the output also against the oracle, and bit-identical from run to run."""
    N, C, H, W, K = shape
    dt = L.GDL_BF16
    st = L.cur_stream()
    lib = L.load()
    need_f = lib.gdl_conv_split_workspace_bytes(dt, N, H, W, C, K, 3, 3, 1, 1, 0)
    need_d = lib.gdl_conv_split_workspace_bytes(dt, N, H, W, C, K, 3, 3, 1, 1, 1)
    assert need_f > 0 and need_d > 0, "these shapes are expected to split"
    assert lib.gdl_conv_split_workspace_bytes(dt, 192, 56, 56, 64, 64, 3, 3, 1, 1, 0) == 0  # a layer that fills the chip does not
    ws = torch.empty(max(need_f, need_d), dtype=torch.uint8, device=DEV)
    x, w = _conv_case(N, C, H, W, K, 3, 1, 1, dt)
    krsc, crsk = pack_weight(w, dt)
    xd = to_nhwc(x, dt)
    tab = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, 3, 3, 1, 1)
    tiles = lib.gdl_conv_bn_tiles(dt, N, H, W, C, K, 3, 3, 1, 1)
    outs, parts = [], []
    for split in (False, True, True):
        y = empty((N, H, W, K), dt)
        part = torch.full((tiles, K, 2), float("nan"), device=DEV)
        if split:
            L.call("gdl_conv_fwd_split", dt, L.ptr(xd), L.ptr(krsc), L.ptr(y), L.ptr(part), L.ptr(tab), N, H, W, C, K, 3, 3, 1, 1,
                   L.ptr(ws), ws.numel(), st)
        else:
            L.call("gdl_conv_fwd", dt, L.ptr(xd), L.ptr(krsc), L.ptr(y), L.ptr(part), L.ptr(tab), N, H, W, C, K, 3, 3, 1, 1, st)
        torch.cuda.synchronize()
        outs.append(y)
        parts.append(part)
    ref = orc.conv2d_fwd(x, w, 1, 1)
    assert relerr(from_nhwc(outs[1]), ref) < 4e-3
    assert torch.equal(outs[1], outs[2]) and torch.equal(parts[1], parts[2])  # run to run
    assert relerr(from_nhwc(outs[1]), from_nhwc(outs[0])) < 4e-3
    s0, s1 = parts[0].double().sum(0).cpu().numpy(), parts[1].double().sum(0).cpu().numpy()
    g = from_nhwc(outs[1]).astype(np.float64)
    np.testing.assert_allclose(s1[:, 0], g.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s1[:, 1], (g * g).sum((0, 2, 3)), rtol=1e-4)
    np.testing.assert_allclose(s1, s0, rtol=2e-2, atol=0.5)  # (different bf16 roundings of a few outputs)
    # ---- data gradient: addend in place, ReLU bits, two BatchNorm partners
    dy = quant(rng.standard_normal((N, K, H, W), dtype=np.float32), dt)
    add = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    yb = quant(rng.standard_normal((N, C, H, W), dtype=np.float32) * 1.5 + 0.3, dt)
    y2 = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    mean, rstd = (rng.standard_normal(C) * 0.3).astype(np.float32), (0.5 + rng.random(C)).astype(np.float32)
    mean2, rstd2 = (rng.standard_normal(C) * 0.3).astype(np.float32), (0.5 + rng.random(C)).astype(np.float32)
    bc = lambda v: v[None, :, None, None]
    keep = yb > 0.3
    kb = np.transpose(keep, (0, 2, 3, 1)).reshape(-1, 8)
    bits = torch.from_numpy((kb * (1 << np.arange(8))[None, :]).sum(1).astype(np.uint8)).to(DEV)
    dyd, yd, y2d = to_nhwc(dy, dt), to_nhwc(yb, dt), to_nhwc(y2, dt)
    tabd = gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, 3, 3, 1, 1)
    tl = lib.gdl_conv_dgrad_bn_tiles(dt, N, H, W, C, K, 3, 3, 1, 1)
    md, rd, m2d, r2d = dev(mean), dev(rstd), dev(mean2), dev(rstd2)
    res = []
    for split in (False, True, True):
        dx = to_nhwc(add, dt)  # the addend is accumulated in place
        p1 = torch.full((tl, C, 2), float("nan"), device=DEV)
        p2 = torch.full((tl, C, 2), float("nan"), device=DEV)
        args = [dt, L.ptr(dyd), L.ptr(crsk), L.ptr(dx), L.ptr(dx), L.ptr(bits), L.ptr(tabd), N, H, W, C, K, 3, 3, 1, 1, L.ptr(yd),
                L.ptr(md), L.ptr(rd), L.ptr(p1), L.ptr(y2d), L.ptr(m2d), L.ptr(r2d), L.ptr(p2)]
        if split:
            L.call("gdl_conv_dgrad_bn_split", *args, L.ptr(ws), ws.numel(), st)
        else:
            L.call("gdl_conv_dgrad_bn", *args, st)
        torch.cuda.synchronize()
        res.append((dx, p1, p2))
    want = np.where(keep, orc.conv2d_bwd_data(dy, w, (N, C, H, W), 1, 1) + add, 0)
    got = from_nhwc(res[1][0])
    assert relerr(got, want) < 6e-3 and np.all(got[~keep] == 0)
    assert all(torch.equal(a, b) for a, b in zip(res[1], res[2]))  # run to run
    g64 = got.astype(np.float64)
    sc_ = np.abs(g64).sum((0, 2, 3)).max()
    a1, a2 = res[1][1].double().sum(0).cpu().numpy(), res[1][2].double().sum(0).cpu().numpy()
    np.testing.assert_allclose(a1[:, 0], g64.sum((0, 2, 3)), rtol=1e-4, atol=2e-6 * sc_)
    np.testing.assert_allclose(a1[:, 1], (g64 * (yb.astype(np.float64) - bc(mean)) * bc(rstd)).sum((0, 2, 3)), rtol=1e-4, atol=1e-5 * sc_)
    np.testing.assert_allclose(a2[:, 0], g64.sum((0, 2, 3)), rtol=1e-4, atol=2e-6 * sc_)
    np.testing.assert_allclose(a2[:, 1], (g64 * (y2.astype(np.float64) - bc(mean2)) * bc(rstd2)).sum((0, 2, 3)), rtol=1e-4, atol=1e-5 * sc_)
    assert relerr(got, from_nhwc(res[0][0])) < 6e-3


@pytest.mark.parametrize("knobs", ["GDL_PSLAB_GRID=32", "GDL_PSLAB_GRID=64,GDL_PSLAB=2", "GDL_PSLAB=0"])
def test_persistent_slab_kernel_alternate_grids(knobs):
    """conv3x3_pslab_kernel (round 6) with SEVERAL tiles per block at test sizes -- the grid capped to 32 / 64 blocks, so the
    weight stream's wrap, the next tile's slab / mask prefetch, the per-block statistics rows and the N-tile-preserving stride
    are all exercised against the oracle -- and in place of the 8-wave tile (GDL_PSLAB=2); GDL_PSLAB=0 keeps the round-5 kernels
    under test.  Tuning knobs are read once per process, hence a child process running this file's convolution tests."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, GDL_TUNING="1", **dict(kv.split("=") for kv in knobs.split(",")))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "test_conv_fwd or test_conv_dgrad or test_relu_bits_and_masked_dgrad or test_conv_run_to_run_determinism",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout


_DUMP_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/iccv2025-gdl_amd"); sys.path.insert(0, {root!r} + "/tests")
from gdl import _lib as L
from gpu_util import DEV, empty, gather_table, pack_weight, quant, to_nhwc
rng = np.random.default_rng(7)
out = {{}}
for (N, C, H, W, K) in [(64, 128, 28, 28, 128), (40, 256, 14, 14, 256), (48, 512, 7, 7, 512)]:
    dt = L.GDL_BF16
    x = quant(rng.standard_normal((N, C, H, W), dtype=np.float32), dt)
    w = quant((rng.standard_normal((K, C, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (C * 9))).astype(np.float32), dt)
    dy = quant(rng.standard_normal((N, K, H, W), dtype=np.float32), dt)
    krsc, crsk = pack_weight(w, dt)
    xd, dyd = to_nhwc(x, dt), to_nhwc(dy, dt)
    y, dx = empty((N, H, W, K), dt), empty((N, H, W, C), dt)
    tiles = L.load().gdl_conv_bn_tiles(dt, N, H, W, C, K, 3, 3, 1, 1)
    part = torch.zeros((tiles, K, 2), device=DEV)
    tf, tb = gather_table(L.GATHER_FWD, dt, N, H, W, C, K, 3, 3, 1, 1), gather_table(L.GATHER_DGRAD, dt, N, H, W, C, K, 3, 3, 1, 1)
    L.call("gdl_conv_fwd", dt, L.ptr(xd), L.ptr(krsc), L.ptr(y), L.ptr(part), L.ptr(tf), N, H, W, C, K, 3, 3, 1, 1, L.cur_stream())
    L.call("gdl_conv_dgrad", dt, L.ptr(dyd), L.ptr(crsk), L.ptr(dx), None, L.ptr(tb), N, H, W, C, K, 3, 3, 1, 1, L.cur_stream())
    torch.cuda.synchronize()
    out["y%d" % C] = y.view(torch.int16).cpu().numpy()
    out["dx%d" % C] = dx.view(torch.int16).cpu().numpy()
    out["sum%d" % C] = part.double().sum(0).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_persistent_slab_kernel_bit_identical_to_round5_kernel(tmp_path):
    """conv3x3_pslab_kernel multiplies in the order conv3x3_slab_kernel does (chunk, tap, 32-channel half; fp32 MFMA accumulators,
    one rounding): its stored outputs must be BIT-identical to the round-5 kernel's (GDL_PSLAB=0) at all three channel counts, forward
    and data gradient -- only the per-channel statistics are summed in another order (per wave and block instead of per tile), and
    those must agree to fp32 rounding.  This is what separates "a different rounding order of the BatchNorm sums" from "a different
    convolution" when a step-level bf16 bound moves (tests/test_step_gpu.py::test_full_size_oracle_parity)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for knob in ("1", "0"):
        f = str(tmp_path / f"pslab{knob}.npz")
        env = dict(os.environ, GDL_TUNING="1", GDL_PSLAB=knob)
        r = subprocess.run([sys.executable, "-c", _DUMP_SNIPPET.format(root=root), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        files.append(np.load(f))
    new, old = files
    for k in new.files:
        if k.startswith("sum"):
            np.testing.assert_allclose(new[k], old[k], rtol=2e-6, atol=1e-3)
        else:
            assert np.array_equal(new[k], old[k]), (k, int((new[k] != old[k]).sum()))
