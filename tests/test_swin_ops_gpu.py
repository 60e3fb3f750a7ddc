"""The Swin operators of csrc/swin.hip one by one through the C ABI (gdl_swin_*, gdl_conv_fwd_bias, gdl_head_concat_xy_*)
against float64 restatements written here / oracle/swin_oracle.py's index helpers.  f32 storage: fp32-level agreement;
bf16 storage: inputs are rounded to bf16 first, so the bounds only carry the output rounding and fp32 accumulation."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, L, bf16_round

from oracle import swin_oracle as so

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(20260)
DTS = ["f32", "bf16"]


def _td(dt):
    return torch.float32 if dt == "f32" else torch.bfloat16


def _q(a, dt):
    return a.astype(np.float32) if dt == "f32" else bf16_round(a.astype(np.float32))


def _dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(_td(dt))


def _np(t):
    return t.float().cpu().numpy().astype(np.float64)


def _tol(dt, f32, bf16):
    return f32 if dt == "f32" else bf16


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,C,ld", [(37, 96, 128), (1000, 192, 192), (9000, 384, 384), (70, 768, 768), (33, 1536, 1536),
                                    (30011, 96, 128), (30001, 192, 192), (30005, 384, 384), (30002, 768, 768), (30003, 1536, 1536)])
def test_layernorm_fwd_bwd(dt, M, C, ld):
    """LayerNorm forward / backward (swin_transformer.py:203, 219, 340, 542) against float64; from 30 000 rows on the backward runs
    its pipelined form (the next iteration's rows requested before this one's arithmetic) -- one case per vectors-per-lane
    variant, row counts that are not multiples of a block's rows."""
    dc = L.dtype_code(dt)
    x = np.zeros((M, ld), np.float32)
    x[:, :C] = _q(rng.standard_normal((M, C)) * 1.5 + 0.3, dt)
    dy = np.zeros((M, ld), np.float32)
    dy[:, :C] = _q(rng.standard_normal((M, C)), dt)
    add = _q(rng.standard_normal((M, ld)), dt)
    add[:, C:] = 0
    g = np.zeros(ld, np.float32)
    b = np.zeros(ld, np.float32)
    g[:C] = 1 + 0.1 * rng.standard_normal(C)
    b[:C] = 0.1 * rng.standard_normal(C)
    xd, dyd, addd = _dev(x, dt), _dev(dy, dt), _dev(add, dt)
    gd, bd = torch.from_numpy(g).to(DEV), torch.from_numpy(b).to(DEV)
    y = torch.full((M, ld), float("nan"), device=DEV, dtype=_td(dt))
    stats = torch.empty((M, 2), device=DEV)
    st = L.cur_stream()
    L.call("gdl_swin_ln_fwd", dc, L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(y), L.ptr(stats), M, C, ld, st)
    x64 = x[:, :C].astype(np.float64)
    mu = x64.mean(1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(1, keepdims=True)
    rstd = 1 / np.sqrt(var + 1e-5)
    xh = (x64 - mu) * rstd
    want = xh * g[:C] + b[:C]
    got = _np(y)
    assert np.abs(got[:, :C] - want).max() < _tol(dt, 2e-5, 4e-2) and np.all(got[:, C:] == 0)
    np.testing.assert_allclose(_np(stats)[:, 0], mu[:, 0], atol=1e-5)
    np.testing.assert_allclose(_np(stats)[:, 1], rstd[:, 0], rtol=1e-4)
    dx = torch.full((M, ld), float("nan"), device=DEV, dtype=_td(dt))
    dgb = torch.empty((2, ld), device=DEV)
    part = torch.empty(L.load().gdl_swin_partial_bytes(ld), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_ln_bwd", dc, L.ptr(dyd), L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(addd), L.ptr(dx), L.ptr(dgb), L.ptr(part), M, C,
           ld, st)
    gg = dy[:, :C].astype(np.float64) * g[:C]
    want_dx = rstd * (gg - gg.mean(1, keepdims=True) - xh * (gg * xh).mean(1, keepdims=True)) + add[:, :C]
    gdx = _np(dx)
    assert np.abs(gdx[:, :C] - want_dx).max() < _tol(dt, 5e-5, 6e-2) * max(1.0, np.abs(want_dx).max())
    assert np.all(gdx[:, C:] == 0)
    dgam, dbet = (dy[:, :C].astype(np.float64) * xh).sum(0), dy[:, :C].astype(np.float64).sum(0)
    sc = max(1.0, np.sqrt(M))
    np.testing.assert_allclose(_np(dgb)[0, :C], dgam, atol=3e-5 * sc * 10)
    np.testing.assert_allclose(_np(dgb)[1, :C], dbet, atol=3e-5 * sc * 10)
    # the three-row form: same dx / d gamma / d beta bit for bit, row 2 = column sums of dx as stored
    vpr, lpr = ld // (4 if dt == "f32" else 8), 16
    while lpr < vpr and lpr < 64:
        lpr *= 2
    if 4 * (64 // lpr) * 3 * ld * 4 > 64 * 1024:
        return  # (wider than the Swin residual stream ever is: the third LDS row does not fit, the library refuses)
    dx3 = torch.full((M, ld), float("nan"), device=DEV, dtype=_td(dt))
    dgb3 = torch.empty((3, ld), device=DEV)
    part3 = torch.empty(L.load().gdl_swin_partial_bytes(2 * ld), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_ln_bwd_colsum", dc, L.ptr(dyd), L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(addd), L.ptr(dx3), L.ptr(dgb3),
           L.ptr(part3), M, C, ld, st)
    assert torch.equal(dx3, dx) and torch.equal(dgb3[:2], dgb)
    np.testing.assert_allclose(_np(dgb3)[2], _np(dx).sum(0), atol=3e-5 * sc * 10 * max(1.0, np.abs(want_dx).max()))


@pytest.mark.parametrize("dt", DTS)
def test_deferred_folds_bit_identical(dt):
    """Round 6: LayerNorm backwards / column-sum passes that leave their partial rows (result pointer NULL) + ONE
    gdl_swin_partial_reduce_batched over all of them == the per-call folds, bit for bit -- including a job whose third row is
    taken over by a later column-sum job (DropPath: width 2 * ld at a row pitch of 3 * ld) and row counts on both sides of the
    512-block cap."""
    import ctypes

    class Red(ctypes.Structure):
        _fields_ = [("partial", ctypes.c_void_p), ("out", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in ("rows", "width", "blk0", "stride")]

    lib, dc, st = L.load(), L.dtype_code(dt), L.cur_stream()
    jobs, want, got = [], [], []
    for M, C, ld, colsum, takeover in ((3000, 96, 128, True, False), (40000, 192, 192, True, True), (777, 384, 384, False, False)):
        x = _dev(_q(rng.standard_normal((M, ld)), dt), dt)
        x[:, C:] = 0
        dy = _dev(_q(rng.standard_normal((M, ld)), dt), dt)
        dy[:, C:] = 0
        g = torch.zeros(ld, device=DEV)
        g[:C] = 1.0
        b = torch.zeros(ld, device=DEV)
        y, stats = torch.empty_like(x), torch.empty((M, 2), device=DEV)
        L.call("gdl_swin_ln_fwd", dc, L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), M, C, ld, st)
        nr = 3 if colsum else 2
        name = "gdl_swin_ln_bwd_colsum" if colsum else "gdl_swin_ln_bwd"
        dx, ref = torch.empty_like(x), torch.empty((nr, ld), device=DEV)
        part = torch.empty(lib.gdl_swin_partial_bytes(2 * ld), dtype=torch.uint8, device=DEV)
        L.call(name, dc, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(g), None, L.ptr(dx), L.ptr(ref), L.ptr(part), M, C, ld, st)
        rows = lib.gdl_swin_ln_bwd_rows(dc, M, ld)
        assert 0 < rows <= 512
        own = torch.full((rows * nr * ld,), float("nan"), device=DEV)
        dx2, out = torch.empty_like(x), torch.full((nr, ld), float("nan"), device=DEV)
        L.call(name, dc, L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(g), None, L.ptr(dx2), None, L.ptr(own), M, C, ld, st)
        assert torch.equal(dx2, dx)
        jobs.append((own, out, rows, (2 if takeover else nr) * ld, nr * ld))
        if takeover:  # the column sums of another tensor replace the third row
            gq = _dev(_q(rng.standard_normal((M, ld)), dt), dt)
            L.call("gdl_swin_colsum", dc, L.ptr(gq), None, L.ptr(ref[2]), L.ptr(part), M, ld, st)
            crow = lib.gdl_swin_colsum_rows(dc, M, ld)
            assert 0 < crow <= 512
            cown = torch.full((crow * ld,), float("nan"), device=DEV)
            L.call("gdl_swin_colsum", dc, L.ptr(gq), None, None, L.ptr(cown), M, ld, st)
            jobs.append((cown, out[2], crow, ld, ld))
        want.append(ref)
        got.append(out)
    arr, blk = (Red * len(jobs))(), 0
    for d, (partial, out, rows, width, stride) in zip(arr, jobs):
        d.partial, d.out, d.rows, d.width, d.blk0, d.stride = partial.data_ptr(), out.data_ptr(), rows, width, blk, stride
        blk += (width + 15) // 16
    tab = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(DEV)
    L.call("gdl_swin_partial_reduce_batched", L.ptr(tab), len(jobs), blk, st)
    torch.cuda.synchronize()
    for w, o in zip(want, got):
        assert torch.equal(w, o)
    assert lib.gdl_swin_ln_bwd_rows(dc, 100, 100) == 0 and lib.gdl_swin_colsum_rows(99, 100, 128) == 0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,ld", [(50, 128), (3000, 384), (700, 3072)])
def test_bias_act_and_colsum(dt, M, ld):
    dc = L.dtype_code(dt)
    st = L.cur_stream()
    y0 = _q(rng.standard_normal((M, ld)), dt)
    bias = (0.3 * rng.standard_normal(ld)).astype(np.float32)
    res = _q(rng.standard_normal((M, ld)), dt)
    bd = torch.from_numpy(bias).to(DEV)
    # mode 2: y + b + res
    y = _dev(y0, dt)
    L.call("gdl_swin_bias_act", dc, L.ptr(y), L.ptr(bd), None, L.ptr(_keep(_dev(res, dt))), M, ld, 2, st)
    assert np.abs(_np(y) - (y0.astype(np.float64) + bias + res)).max() < _tol(dt, 1e-6, 3e-2)
    # mode 1: u = y + b stored, y = gelu(u)
    y, u = _dev(y0, dt), torch.empty((M, ld), device=DEV, dtype=_td(dt))
    L.call("gdl_swin_bias_act", dc, L.ptr(y), L.ptr(bd), L.ptr(u), None, M, ld, 1, st)
    uu = _np(u)
    assert np.abs(uu - (y0.astype(np.float64) + bias)).max() < _tol(dt, 1e-6, 2e-2)
    import math

    erf = np.vectorize(math.erf)
    assert np.abs(_np(y) - 0.5 * uu * (1 + erf(uu / math.sqrt(2)))).max() < _tol(dt, 2e-6, 2e-2)
    # column sums, with and without the GELU derivative
    gq = _q(rng.standard_normal((M, ld)), dt)
    gdv = _dev(gq, dt)
    db = torch.empty(ld, device=DEV)
    part = torch.empty(L.load().gdl_swin_partial_bytes(ld), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_colsum", dc, L.ptr(gdv), None, L.ptr(db), L.ptr(part), M, ld, st)
    np.testing.assert_allclose(_np(db), gq.astype(np.float64).sum(0), atol=2e-4 * np.sqrt(M))
    L.call("gdl_swin_colsum", dc, L.ptr(gdv), L.ptr(u), L.ptr(db), L.ptr(part), M, ld, st)
    dgelu = 0.5 * (1 + erf(uu / math.sqrt(2))) + uu * np.exp(-0.5 * uu * uu) / math.sqrt(2 * math.pi)
    want = gq.astype(np.float64) * dgelu
    assert np.abs(_np(gdv) - want).max() < _tol(dt, 3e-6, 3e-2)
    np.testing.assert_allclose(_np(db), _np(gdv).sum(0), atol=2e-4 * np.sqrt(M))


_KEEP = []


def _keep(t):
    _KEEP.append(t)
    del _KEEP[:-8]
    return t


def _attn_ref(qkv, table, n_img, H, W, ws, shift, nh, ld):
    """float64 window attention on QKV rows [n_img*H*W][3*ld] via the oracle's index lists; returns out [rows][ld]."""
    C = nh * 32
    q = torch.from_numpy(qkv).double().reshape(n_img, H * W, 3, ld)[..., :C].reshape(n_img, H * W, 3, nh, 32).requires_grad_(True)
    idx = so.window_tokens(H, W, ws, shift)
    nW, T = idx.shape
    g = q[:, idx.reshape(-1)].reshape(n_img, nW, T, 3, nh, 32)
    qq, kk, vv = (g[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))
    s = (qq * 32 ** -0.5) @ kk.transpose(-2, -1)
    s = s + torch.from_numpy(table).double()[so.relative_index(ws).reshape(-1)].reshape(T, T, nh).permute(2, 0, 1)
    if shift:
        reg = so.window_regions(H, W, ws, shift)
        s = s + torch.where(reg[:, :, None] != reg[:, None, :], -100.0, 0.0).double()[None, :, None]
    o = (torch.softmax(s, -1) @ vv).permute(0, 1, 3, 2, 4).reshape(n_img, nW * T, C)
    out = torch.zeros(n_img, H * W, C, dtype=torch.float64).index_add(1, idx.reshape(-1), o)
    return q, out


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,T,H", [(2, 2, 16), (1, 3, 56)])
def test_patch_gather(dt, B, T, H):
    """gdl_swin_patch_gather: frames [B, 3, T, H, W] f32 -> GEMM rows [B*T*(H/4)*(W/4)][64], columns (c, kh, kw) of the 4 x 4 patch
    (the flattened Conv2d weight's order, swin_transformer.py:463, 480), 16 zero columns.  bf16 runs the 16-byte-vector kernel."""
    x = rng.standard_normal((B, 3, T, H, H)).astype(np.float32)
    a = torch.full((B * T * (H // 4) ** 2, 64), float("nan"), device=DEV, dtype=_td(dt))
    xd = torch.from_numpy(x).to(DEV)
    L.call("gdl_swin_patch_gather", L.dtype_code(dt), L.ptr(xd), L.ptr(a), B, T, H, H, 4, L.cur_stream())
    g = x.transpose(0, 2, 1, 3, 4).reshape(B * T, 3, H // 4, 4, H // 4, 4).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 48)
    got = _np(a)
    np.testing.assert_array_equal(got[:, :48], _q(g, dt))
    assert np.all(got[:, 48:] == 0)


def _attn_ref_full(qkv, dout, table, n_img, H, W, ws, shift, nh, ld, chunk=16):
    """float64 window attention forward + backward over ALL images, `chunk` images at a time (the images are independent; the
    table gradient adds up): out [rows][C], dqkv [rows][3][C], dtable -- the oracle's index lists / region masks."""
    C = nh * 32
    tt = torch.from_numpy(table).double().requires_grad_(True)
    idx = so.window_tokens(H, W, ws, shift)
    nW, T = idx.shape
    ri = so.relative_index(ws).reshape(-1)
    mask = None
    if shift:
        reg = so.window_regions(H, W, ws, shift)
        mask = torch.where(reg[:, :, None] != reg[:, None, :], -100.0, 0.0).double()[None, :, None]
    outs, dqs = [], []
    L_ = H * W
    for i0 in range(0, n_img, chunk):
        n = min(chunk, n_img - i0)
        q = torch.from_numpy(qkv[i0 * L_:(i0 + n) * L_]).double().reshape(n, L_, 3, ld)[..., :C].reshape(n, L_, 3, nh, 32).requires_grad_(True)
        g = q[:, idx.reshape(-1)].reshape(n, nW, T, 3, nh, 32)
        qq, kk, vv = (g[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))
        s = (qq * 32 ** -0.5) @ kk.transpose(-2, -1) + tt[ri].reshape(T, T, nh).permute(2, 0, 1)
        if mask is not None:
            s = s + mask
        o = (torch.softmax(s, -1) @ vv).permute(0, 1, 3, 2, 4).reshape(n, nW * T, C)
        o2 = torch.zeros(n, L_, C, dtype=torch.float64).index_add(1, idx.reshape(-1), o)
        (o2 * torch.from_numpy(dout[i0 * L_:(i0 + n) * L_, :C]).double().reshape(n, L_, C)).sum().backward()
        outs.append(o2.detach().reshape(n * L_, C))
        dqs.append(q.grad.reshape(n * L_, 3, C))
    return torch.cat(outs).numpy(), torch.cat(dqs).numpy(), tt.grad.numpy()


@pytest.mark.parametrize("H,nh,ld,shift", [(56, 3, 128, 3), (28, 6, 192, 3), (14, 12, 384, 0), (7, 24, 768, 0)])
def test_window_attention_config5_size(H, nh, ld, shift):
    """The window attention at config 5's own launch sizes -- 192 frames, the four Swin-T stages: the launches whose block count
    exceeds one round of resident blocks take the larger window chunks -- BOTH the float32 kernels (csrc/swin.hip) and the bf16
    ones (csrc/swin_attn7.hip) against the float64 formulas of the oracle's index lists over all 192 images (VERDICT r4 weak #2:
    this test compared bf16 with the library's float32 kernels only): output rows, the QKV gradient, the table gradient."""
    n_img, ws = 192, 7
    rows, C = n_img * H * H, nh * 32
    g = torch.Generator(device=DEV).manual_seed(H)
    qkv = torch.zeros(rows, 3, ld, device=DEV)
    qkv[:, :, :C] = torch.randn(rows, 3, C, device=DEV, generator=g).bfloat16().float()
    qkv = qkv.reshape(rows, 3 * ld)
    dout = torch.zeros(rows, ld, device=DEV)
    dout[:, :C] = torch.randn(rows, C, device=DEV, generator=g).bfloat16().float()
    table = 0.5 * torch.randn((2 * ws - 1) ** 2, nh, device=DEV, generator=g)
    want_o, want_dq, want_dt = _attn_ref_full(qkv.cpu().numpy(), dout.cpu().numpy(), table.cpu().numpy(), n_img, H, H, ws, shift, nh, ld)
    sq, st_ = max(1.0, float(np.abs(want_dq).max())), max(1.0, float(np.abs(want_dt).max()))
    st = L.cur_stream()
    for dt in ("f32", "bf16"):
        dc, td = L.dtype_code(dt), _td(dt)
        q, d = qkv.to(td), dout.to(td)
        out = torch.full((rows, ld), float("nan"), device=DEV, dtype=td)
        dq = torch.full((rows, 3 * ld), float("nan"), device=DEV, dtype=td)
        dtab = torch.empty_like(table)
        wsb = torch.empty(max(L.load().gdl_swin_attn_bwd_workspace_bytes(n_img, H, H, ws, nh), 4), dtype=torch.uint8, device=DEV)
        L.call("gdl_swin_attn_fwd", dc, L.ptr(q), L.ptr(table), L.ptr(out), n_img, H, H, ws, shift, nh, ld, st)
        L.call("gdl_swin_attn_bwd", dc, L.ptr(q), L.ptr(table), L.ptr(d), L.ptr(dq), L.ptr(dtab), L.ptr(wsb), n_img, H, H, ws, shift, nh, ld, st)
        torch.cuda.synchronize()
        o, gq, gt = out.float().cpu().numpy(), dq.float().cpu().numpy().reshape(rows, 3, ld), dtab.cpu().numpy()
        assert not np.isnan(o).any() and not np.isnan(gq).any()
        eo, eq, et = float(np.abs(o[:, :C] - want_o).max()), float(np.abs(gq[..., :C] - want_dq).max()) / sq, float(np.abs(gt - want_dt).max()) / st_
        print(f"attention {H}x{H} x 192 {dt} vs float64: out {eo:.2e}, dqkv {eq:.2e}, dtable {et:.2e}")
        # (the table gradient is a sum over 192 images x windows: fp32 accumulation in fixed order, its bound scales with the sum)
        assert eo < _tol(dt, 3e-6, 3e-2) and eq < _tol(dt, 1e-5, 4e-2) and et < _tol(dt, 2e-4, 6e-2), (dt, eo, eq, et)
        assert np.all(o[:, C:] == 0) and np.all(gq[..., C:] == 0)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("H,ws,shift,nh", [(14, 7, 0, 3), (14, 7, 3, 3), (7, 7, 0, 6), (28, 7, 3, 3), (35, 7, 3, 3), (56, 7, 3, 3), (4, 2, 1, 2)])
def test_window_attention(dt, H, ws, shift, nh):
    dc = L.dtype_code(dt)
    n_img, W, ld = 3, H, 128 if nh * 32 <= 128 else 192
    C = nh * 32
    rows = n_img * H * W
    qkv = np.zeros((rows, 3, ld), np.float32)
    qkv[:, :, :C] = _q(rng.standard_normal((rows, 3, C)), dt)
    qkv = qkv.reshape(rows, 3 * ld)
    table = (0.5 * rng.standard_normal(((2 * ws - 1) ** 2, nh))).astype(np.float32)
    dout = np.zeros((rows, ld), np.float32)
    dout[:, :C] = _q(rng.standard_normal((rows, C)), dt)
    qd, td_, dd = _dev(qkv, dt), torch.from_numpy(table).to(DEV), _dev(dout, dt)
    out = torch.full((rows, ld), float("nan"), device=DEV, dtype=_td(dt))
    st = L.cur_stream()
    L.call("gdl_swin_attn_fwd", dc, L.ptr(qd), L.ptr(td_), L.ptr(out), n_img, H, W, ws, shift, nh, ld, st)
    qref, oref = _attn_ref(qkv, table, n_img, H, W, ws, shift, nh, ld)
    got = _np(out).reshape(n_img, H * W, ld)
    assert np.abs(got[..., :C] - oref.detach().numpy()).max() < _tol(dt, 3e-6, 3e-2) and np.all(got[..., C:] == 0)
    dq = torch.full((rows, 3 * ld), float("nan"), device=DEV, dtype=_td(dt))
    dtab = torch.empty_like(td_)
    ws_b = torch.empty(max(L.load().gdl_swin_attn_bwd_workspace_bytes(n_img, H, W, ws, nh), 4), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_attn_bwd", dc, L.ptr(qd), L.ptr(td_), L.ptr(dd), L.ptr(dq), L.ptr(dtab), L.ptr(ws_b), n_img, H, W, ws, shift, nh,
           ld, st)
    tt = torch.from_numpy(table).double().requires_grad_(True)
    # gradients of the reference (table included): redo the forward with the table as a leaf
    idx = so.window_tokens(H, W, ws, shift)
    nW, T = idx.shape
    g = qref[:, idx.reshape(-1)].reshape(n_img, nW, T, 3, nh, 32)
    qq, kk, vv = (g[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))
    s = (qq * 32 ** -0.5) @ kk.transpose(-2, -1) + tt[so.relative_index(ws).reshape(-1)].reshape(T, T, nh).permute(2, 0, 1)
    if shift:
        reg = so.window_regions(H, W, ws, shift)
        s = s + torch.where(reg[:, :, None] != reg[:, None, :], -100.0, 0.0).double()[None, :, None]
    o = (torch.softmax(s, -1) @ vv).permute(0, 1, 3, 2, 4).reshape(n_img, nW * T, C)
    o2 = torch.zeros(n_img, H * W, C, dtype=torch.float64).index_add(1, idx.reshape(-1), o)
    (o2 * torch.from_numpy(dout[:, :C].reshape(n_img, H * W, C)).double()).sum().backward()
    want_dq = qref.grad.numpy().reshape(rows, 3, C)
    gdq = _np(dq).reshape(rows, 3, ld)
    sc = max(1.0, np.abs(want_dq).max())
    assert np.abs(gdq[..., :C] - want_dq).max() < _tol(dt, 1e-5, 4e-2) * sc and np.all(gdq[..., C:] == 0)
    np.testing.assert_allclose(_np(dtab), tt.grad.numpy(), atol=_tol(dt, 2e-4, 6e-2) * max(1.0, np.abs(tt.grad.numpy()).max()))


@pytest.mark.parametrize("dt", DTS)
def test_merge_tokenmean_pack(dt):
    dc = L.dtype_code(dt)
    st = L.cur_stream()
    N, H, W, C, ld = 2, 6, 4, 96, 128
    x = np.zeros((N, H, W, ld), np.float32)
    x[..., :C] = _q(rng.standard_normal((N, H, W, C)), dt)
    xd = _dev(x.reshape(-1, ld), dt)
    cat = torch.empty((N * H * W // 4, 4 * C), device=DEV, dtype=_td(dt))
    L.call("gdl_swin_merge", dc, L.ptr(xd), L.ptr(cat), N, H, W, C, ld, 0, st)
    g = x[..., :C].reshape(N, H // 2, 2, W // 2, 2, C)
    want = np.concatenate([g[:, :, 0, :, 0], g[:, :, 1, :, 0], g[:, :, 0, :, 1], g[:, :, 1, :, 1]], -1).reshape(-1, 4 * C)
    np.testing.assert_array_equal(_np(cat), want)
    back = torch.full((N * H * W, ld), float("nan"), device=DEV, dtype=_td(dt))
    L.call("gdl_swin_merge", dc, L.ptr(cat), L.ptr(back), N, H, W, C, ld, 1, st)
    np.testing.assert_array_equal(_np(back), x.reshape(-1, ld))  # the adjoint of a permutation is its inverse
    feat = torch.empty((N, C), device=DEV)
    L.call("gdl_swin_token_mean", dc, L.ptr(xd), L.ptr(feat), N, H * W, C, ld, st)
    np.testing.assert_allclose(_np(feat), x[..., :C].astype(np.float64).reshape(N, -1, C).mean(1), atol=1e-6)
    dfe = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32)).to(DEV)
    dx = torch.full((N * H * W, ld), float("nan"), device=DEV, dtype=_td(dt))
    L.call("gdl_swin_token_mean_bwd", dc, L.ptr(dfe), L.ptr(dx), N, H * W, C, ld, st)
    wantdx = np.zeros((N, H * W, ld))
    wantdx[..., :C] = (_np(dfe) / (H * W))[:, None, :]
    assert np.abs(_np(dx).reshape(N, H * W, ld) - wantdx).max() < _tol(dt, 1e-7, 5e-3)
    # QKV-style packing: 3 segments of 96 rows -> pitch 128, columns 96 -> 128; transposed copy; and back
    w = rng.standard_normal((288, 96)).astype(np.float32)
    wd = torch.from_numpy(w).to(DEV)
    pk, pkT = torch.full((384, 128), float("nan"), device=DEV, dtype=_td(dt)), torch.full((128, 384), float("nan"), device=DEV, dtype=_td(dt))
    L.call("gdl_swin_pack_matrix", dc, L.ptr(wd), L.ptr(pk), L.ptr(pkT), 288, 96, 96, 128, 96, 128, st)
    wantp = np.zeros((3, 128, 128))
    wantp[:, :96, :96] = _q(w, dt).reshape(3, 96, 96)
    np.testing.assert_array_equal(_np(pk), wantp.reshape(384, 128))
    np.testing.assert_array_equal(_np(pkT), wantp.reshape(384, 128).T)
    if dt == "f32":
        rt = torch.empty((288, 96), device=DEV)
        L.call("gdl_swin_unpack_matrix", L.ptr(pk), L.ptr(rt), 288, 96, 96, 128, 96, 128, st)
        np.testing.assert_array_equal(_np(rt), w)


@pytest.mark.parametrize("dt", DTS)
def test_linear_with_bias_and_residual(dt):
    """gdl_conv_fwd_bias as an nn.Linear: y = x W^T + b + res on padded rows (1x1 convolution of the library)."""
    from gpu_util import gather_table

    dc = L.dtype_code(dt)
    M, K, N = 777, 128, 384
    x, w = _q(rng.standard_normal((M, K)), dt), _q(rng.standard_normal((N, K)) * 0.1, dt)
    b = rng.standard_normal(N).astype(np.float32)
    res = _q(rng.standard_normal((M, N)), dt)
    tab = gather_table(L.GATHER_FWD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    y = torch.empty((M, N), device=DEV, dtype=_td(dt))
    xd, wd, rd, bd = _dev(x, dt), _dev(w, dt), _dev(res, dt), torch.from_numpy(b).to(DEV)
    ge = torch.empty((M, N), device=DEV, dtype=_td(dt))
    L.call("gdl_conv_fwd_bias", dc, L.ptr(xd), L.ptr(wd), L.ptr(y), L.ptr(bd), L.ptr(rd), L.ptr(ge), L.ptr(tab), M, 1, 1, K, N, 1, 1, 1,
           0, L.cur_stream())
    want = x.astype(np.float64) @ w.astype(np.float64).T + b + res
    assert np.abs(_np(y) - want).max() < _tol(dt, 2e-5, 6e-2)
    import math

    yy = _np(y)
    assert np.abs(_np(ge) - 0.5 * yy * (1 + np.vectorize(math.erf)(yy / math.sqrt(2)))).max() < _tol(dt, 3e-6, 3e-2)


@pytest.mark.parametrize("M,K,N,opts", [
    (20008, 128, 384, ("bias",)),                 # qkv at stage 1; ragged last 16-row tile
    (16400, 128, 128, ("bias", "res")),           # proj: residual addend (prefetched one tile ahead)
    (18000, 128, 384, ("bias", "gelu")),          # fc1: two outputs
    (17000, 192, 576, ("bias",)),                 # stage 2: 96-column chunks
    (16384, 192, 768, ("bias", "gelu")),
    (16500, 192, 192, ()),                        # no epilogue options at all
])
def test_linear_stream_kernel(M, K, N, opts):
    """csrc/linear_stream.hip (round 4): plain GEMMs with K = 128 / 192 and >= 16 384 rows leave gdl_conv_fwd_bias / gdl_conv_fwd on
    the streaming kernel (weights in registers, a wave per 16-row tile, one memory wait per tile).  Against float64 (nn.Linear,
    /root/reference/models/swin_transformer.py:30-46, 98-101) -- and BIT-identical to the tile kernel, which the same call runs for
    the first 4 000 rows alone (below the streaming threshold): same accumulation order, same rounding points."""
    import math

    from gpu_util import gather_table

    dt = "bf16"
    dc = L.dtype_code(dt)
    st = L.cur_stream()
    x, w = _q(rng.standard_normal((M, K)), dt), _q(rng.standard_normal((N, K)) * 0.1, dt)
    b = rng.standard_normal(N).astype(np.float32)
    res = _q(rng.standard_normal((M, N)), dt)
    xd, wd, rd, bd = _dev(x, dt), _dev(w, dt), _dev(res, dt), torch.from_numpy(b).to(DEV)

    def run(rows):
        tab = gather_table(L.GATHER_FWD, dc, rows, 1, 1, K, N, 1, 1, 1, 0)
        y = torch.full((rows, N), float("nan"), device=DEV, dtype=_td(dt))
        ge = torch.full((rows, N), float("nan"), device=DEV, dtype=_td(dt))
        L.call("gdl_conv_fwd_bias", dc, L.ptr(xd), L.ptr(wd), L.ptr(y), L.ptr(bd) if "bias" in opts else None,
               L.ptr(rd) if "res" in opts else None, L.ptr(ge) if "gelu" in opts else None, L.ptr(tab), rows, 1, 1, K, N, 1, 1, 1, 0, st)
        torch.cuda.synchronize()
        return y, ge

    y, ge = run(M)
    want = x.astype(np.float64) @ w.astype(np.float64).T + (b if "bias" in opts else 0.0) + (res if "res" in opts else 0.0)
    assert not torch.isnan(y).any()
    assert np.abs(_np(y) - want).max() < 6e-2 * max(1.0, np.abs(want).max() / 4)
    if "gelu" in opts:
        yy = _np(y)
        assert not torch.isnan(ge).any()
        assert np.abs(_np(ge) - 0.5 * yy * (1 + np.vectorize(math.erf)(yy / math.sqrt(2)))).max() < 3e-2
    y0, ge0 = run(4000)  # the tile kernel (fewer than 16 384 rows)
    assert torch.equal(y[:4000].view(torch.int16), y0.view(torch.int16))
    if "gelu" in opts:
        assert torch.equal(ge[:4000].view(torch.int16), ge0.view(torch.int16))
    y2, _ = run(M)  # run to run
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))


@pytest.mark.parametrize("M,kreal,bias", [(16384 + 40, 128, False), (20000, 96, False), (20000, 96, True), (16384 + 71, 80, True)])
def test_linear_bwd_fused(M, kreal, bias):
    """csrc/linear_bwd.hip (round 4): dx = dy . W and dW = dy^T . x of a Linear (K = 128 -> N = 384: stage-1 qkv / Mlp.fc1,
    swin_transformer.py:26-42, 78-101) from ONE pass over dy, against float64 and against the two GEMMs it replaces (gdl_conv_dgrad +
    gdl_conv_wgrad; same bf16 inputs, fp32 accumulation in another order); ragged row counts (M not a multiple of the 32-row tile),
    rows beyond M untouched, run-to-run bit-identical.  kreal < 128: the input's padding columns (zeros) -- their tiles of dW are
    skipped and come back as zeros; bias: the column sums of dy (the Linear's bias gradient) from the first skipped tile."""
    from gpu_util import gather_table

    dt, K, N = "bf16", 128, 384
    dc = L.dtype_code(dt)
    st = L.cur_stream()
    assert L.load().gdl_linear_bwd_ok(dc, M, K, N) == 1 and L.load().gdl_linear_bwd_ok(dc, 1000, K, N) == 0
    dy, x = _q(rng.standard_normal((M, N)), dt), _q(rng.standard_normal((M, K)), dt)
    w = _q(rng.standard_normal((N, K)) * 0.1, dt)
    x[:, kreal:] = 0
    w[:, kreal:] = 0
    dyd, xd, wT = _dev(dy, dt), _dev(x, dt), _dev(np.ascontiguousarray(w.T), dt)
    nb = L.load().gdl_linear_bwd_workspace_bytes(M, K, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)

    def run():
        dx = torch.full((M + 8, K), float("nan"), device=DEV, dtype=_td(dt))
        dw = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        L.call("gdl_linear_bwd", dc, L.ptr(dyd), L.ptr(xd), L.ptr(wT), L.ptr(dx), L.ptr(dw), L.ptr(db) if bias else None, L.ptr(ws), nb, M, K,
               kreal, N, st)
        torch.cuda.synchronize()
        return dx, dw, db

    dx, dw, db = run()
    assert torch.isnan(dx[M:]).all() and not torch.isnan(dx[:M]).any() and not torch.isnan(dw).any()
    if kreal <= 96:
        assert (dw[:, 96:] == 0).all() and (dx[:M, kreal:] == 0).all()
    if bias:
        want_db = dy.astype(np.float64).sum(0)
        assert np.abs(db.cpu().numpy().astype(np.float64) - want_db).max() < 1e-4 * max(1.0, np.abs(want_db).max())
        cs = torch.empty(N, device=DEV)
        part = torch.empty(L.load().gdl_swin_partial_bytes(N), dtype=torch.uint8, device=DEV)
        L.call("gdl_swin_colsum", dc, L.ptr(dyd), None, L.ptr(cs), L.ptr(part), M, N, st)  # (the pass it replaces)
        torch.cuda.synchronize()
        assert np.abs(db.cpu().numpy() - cs.cpu().numpy()).max() < 1e-4 * max(1.0, np.abs(want_db).max())
        with pytest.raises(L.GdlError):  # no padding tile to carry it
            L.call("gdl_linear_bwd", dc, L.ptr(dyd), L.ptr(xd), L.ptr(wT), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nb, M, K, 128, N, st)
    else:
        assert torch.isnan(db).all()
    want_dx = dy.astype(np.float64) @ w.astype(np.float64)
    want_dw = dy.astype(np.float64).T @ x.astype(np.float64)
    assert np.abs(_np(dx[:M]) - want_dx).max() < 3e-2 * max(1.0, np.abs(want_dx).max() / 4)
    assert np.abs(dw.cpu().numpy().astype(np.float64) - want_dw).max() < 2e-3 * np.abs(want_dw).max()
    # the pair of GEMMs
    tab_d = gather_table(L.GATHER_DGRAD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    tab_f = gather_table(L.GATHER_FWD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    dx2 = torch.empty((M, K), device=DEV, dtype=_td(dt))
    dw2 = torch.empty((N, K), device=DEV)
    L.call("gdl_conv_dgrad", dc, L.ptr(dyd), L.ptr(wT), L.ptr(dx2), None, L.ptr(tab_d), M, 1, 1, K, N, 1, 1, 1, 0, st)
    wsb = L.load().gdl_conv_wgrad_workspace_bytes(dc, M, 1, 1, K, N, 1, 1, 1, 0)
    ws2 = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    L.call("gdl_conv_wgrad", dc, L.ptr(dyd), L.ptr(xd), L.ptr(dw2), L.ptr(tab_f), M, 1, 1, K, N, 1, 1, 1, 0, L.ptr(ws2), wsb, st)
    torch.cuda.synchronize()
    assert np.abs(_np(dx[:M]) - _np(dx2)).max() <= 2.0 ** -7 * max(1.0, np.abs(want_dx).max())  # (one bf16 ulp of the largest value)
    assert np.abs(dw.cpu().numpy() - dw2.cpu().numpy()).max() < 1e-4 * np.abs(want_dw).max()
    dx3, dw3, db3 = run()
    assert torch.equal(dx[:M].view(torch.int16), dx3[:M].view(torch.int16)) and torch.equal(dw, dw3)
    assert not bias or torch.equal(db, db3)


@pytest.mark.parametrize("M", [602112, (1 << 31) // 768 - 1])
def test_linear_bwd_largest_launches(M):
    """gdl_linear_bwd at config 5's own row count (192 frames x 56 x 56) and at the largest launch it accepts (dy just under
    2^31 bytes: its buffer descriptors are 32-bit), against the two GEMMs and the column-sum pass it replaces, all on the device."""
    from gpu_util import gather_table

    K, N, kreal = 128, 384, 96
    dc = L.dtype_code("bf16")
    st = L.cur_stream()
    lib = L.load()
    assert lib.gdl_linear_bwd_ok(dc, M, K, N) == 1 and lib.gdl_linear_bwd_ok(dc, (1 << 31) // 768 + 1, K, N) == 0
    g = torch.Generator(device=DEV).manual_seed(M % 1000)
    dy = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    x[:, kreal:] = 0
    wT = (torch.randn(K, N, device=DEV, generator=g) * 0.1).bfloat16()
    wT[kreal:] = 0
    nb = lib.gdl_linear_bwd_workspace_bytes(M, K, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dx, dw, db = torch.empty(M, K, device=DEV, dtype=torch.bfloat16), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    L.call("gdl_linear_bwd", dc, L.ptr(dy), L.ptr(x), L.ptr(wT), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nb, M, K, kreal, N, st)
    tab_d = gather_table(L.GATHER_DGRAD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    tab_f = gather_table(L.GATHER_FWD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    dx2, dw2, db2 = torch.empty_like(dx), torch.empty_like(dw), torch.empty_like(db)
    L.call("gdl_conv_dgrad", dc, L.ptr(dy), L.ptr(wT), L.ptr(dx2), None, L.ptr(tab_d), M, 1, 1, K, N, 1, 1, 1, 0, st)
    wsb = lib.gdl_conv_wgrad_workspace_bytes(dc, M, 1, 1, K, N, 1, 1, 1, 0)
    ws2 = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    L.call("gdl_conv_wgrad", dc, L.ptr(dy), L.ptr(x), L.ptr(dw2), L.ptr(tab_f), M, 1, 1, K, N, 1, 1, 1, 0, L.ptr(ws2), wsb, st)
    part = torch.empty(lib.gdl_swin_partial_bytes(N), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_colsum", dc, L.ptr(dy), None, L.ptr(db2), L.ptr(part), M, N, st)
    torch.cuda.synchronize()
    assert float((dx.float() - dx2.float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(dx2.float().abs().max()))  # (a bf16 ulp of the largest)
    assert torch.equal(dx[-1], dx[-1]) and not torch.isnan(dx[-40:].float()).any()  # (the last, partial tile)
    scale = float(dw2.abs().max())
    assert float((dw[:, :kreal] - dw2[:, :kreal]).abs().max()) < 2e-4 * scale and bool((dw[:, kreal:] == 0).all())
    assert float((db - db2).abs().max()) < 2e-4 * max(1.0, float(db2.abs().max()))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,K,N", [(777, 384, 128), (5000, 768, 192), (300, 3072, 768)])
def test_linear_dgrad_gelu_colsum(dt, M, K, N):
    """gdl_conv_dgrad_gelu: dx = (dy . W) * gelu'(u) in the GEMM's epilogue, and the column sums of dx (fc1's bias gradient)
    through the fixed-point accumulators + gdl_acc_to_float, against gdl_conv_dgrad followed by gdl_swin_colsum(g, u) (bit-identical
    dx) and float64 (Mlp backward, swin_transformer.py:32-47: fc2 is [N <- K], hidden width K)."""
    import math

    from gpu_util import gather_table

    dc = L.dtype_code(dt)
    st = L.cur_stream()
    dy, w = _q(rng.standard_normal((M, N)), dt), _q(rng.standard_normal((N, K)) * 0.1, dt)
    u = _q(rng.standard_normal((M, K)) * 1.5, dt)
    tab = gather_table(L.GATHER_DGRAD, dc, M, 1, 1, K, N, 1, 1, 1, 0)
    dyd, wT, ud = _dev(dy, dt), _dev(np.ascontiguousarray(w.T), dt), _dev(u, dt)  # w_crsk of a 1x1: [in = K][out = N]
    dx = torch.full((M, K), float("nan"), device=DEV, dtype=_td(dt))
    acc = torch.zeros((K, 2), dtype=torch.int64, device=DEV)
    scale = 2.0 ** (62 - 7 - max(1, (M - 1).bit_length()))
    L.call("gdl_conv_dgrad_gelu", dc, L.ptr(dyd), L.ptr(wT), L.ptr(dx), L.ptr(ud), L.ptr(acc), scale, L.ptr(tab), M, 1, 1, K, N, 1, 1,
           1, 0, st)
    db = torch.empty(K, device=DEV)
    L.call("gdl_acc_to_float", L.ptr(acc), K, 1.0 / scale, L.ptr(db), st)
    # the two-pass form
    ref = torch.full((M, K), float("nan"), device=DEV, dtype=_td(dt))
    L.call("gdl_conv_dgrad", dc, L.ptr(dyd), L.ptr(wT), L.ptr(ref), None, L.ptr(tab), M, 1, 1, K, N, 1, 1, 1, 0, st)
    db2 = torch.empty(K, device=DEV)
    part = torch.empty(L.load().gdl_swin_partial_bytes(K), dtype=torch.uint8, device=DEV)
    L.call("gdl_swin_colsum", dc, L.ptr(ref), L.ptr(ud), L.ptr(db2), L.ptr(part), M, K, st)
    assert torch.equal(dx, ref)
    np.testing.assert_allclose(_np(db), _np(dx).sum(0), atol=2e-5 * np.sqrt(M))  # (exact up to 2^-35 per tile and the tiles' fp32 sums)
    np.testing.assert_allclose(_np(db), _np(db2), atol=2e-4 * np.sqrt(M))
    uu = u.astype(np.float64)
    dgelu = 0.5 * (1 + np.vectorize(math.erf)(uu / math.sqrt(2))) + uu * np.exp(-0.5 * uu * uu) / math.sqrt(2 * math.pi)
    want = (dy.astype(np.float64) @ w.astype(np.float64)) * dgelu
    assert np.abs(_np(dx) - want).max() < _tol(dt, 3e-5, 6e-2) * max(1.0, np.abs(want).max())
    # run to run: integer accumulation is order-independent
    acc2 = torch.zeros_like(acc)
    L.call("gdl_conv_dgrad_gelu", dc, L.ptr(dyd), L.ptr(wT), L.ptr(dx), L.ptr(ud), L.ptr(acc2), scale, L.ptr(tab), M, 1, 1, K, N, 1, 1,
           1, 0, st)
    assert torch.equal(acc[:, 0], acc2[:, 0])


@pytest.mark.parametrize("B,n,dx_,dy_", [(5, 7, 512, 768), (64, 309, 512, 768), (3, 70, 100, 1100), (4, 33, 512, 1024)])
def test_head_concat_xy(B, n, dx_, dy_):
    """The concat DGL head at 512 + 768 features against the float64 formulas of fusion_modules.py:51-59 and their
    autograd with the DGL truncation flags; config 5's own size (64 samples, 309 classes: ten class chunks per sample), widths that
    are not multiples of 64 / beyond the register form (the per-class reload), and the one-launch unimodal path
    (gdl_head_uni_dfeat_w) bit for bit against forward + cross-entropy + backward."""
    x, y = rng.standard_normal((B, dx_)).astype(np.float32), rng.standard_normal((B, dy_)).astype(np.float32)
    W, b = (0.05 * rng.standard_normal((n, dx_ + dy_))).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    gx, gy, go = (rng.standard_normal((B, n)).astype(np.float32) for _ in range(3))
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    xd, yd, Wd, bd, gxd, gyd, god = map(t, (x, y, W, b, gx, gy, go))
    out, xo, yo = (torch.empty((B, n), device=DEV) for _ in range(3))
    st = L.cur_stream()
    L.call("gdl_head_concat_xy_fwd", L.ptr(xd), L.ptr(yd), L.ptr(Wd), L.ptr(bd), L.ptr(out), L.ptr(xo), L.ptr(yo), B, n, dx_, dy_, st)
    pa, pv = x.astype(np.float64) @ W[:, :dx_].T.astype(np.float64), y.astype(np.float64) @ W[:, dx_:].T.astype(np.float64)
    np.testing.assert_allclose(_np(out), pa + pv + b, atol=1e-4)
    np.testing.assert_allclose(_np(xo), pa + b, atol=1e-4)
    np.testing.assert_allclose(_np(yo), pv + b, atol=1e-4)
    dxd, dyd, dW, db = torch.empty_like(xd), torch.empty_like(yd), torch.empty_like(Wd), torch.empty_like(bd)
    L.call("gdl_head_concat_xy_bwd", L.ptr(xd), L.ptr(yd), L.ptr(Wd), L.ptr(gxd), L.ptr(gyd), L.ptr(god), 0, 0, L.ptr(dxd), L.ptr(dyd),
           L.ptr(dW), L.ptr(db), B, n, dx_, dy_, st)  # the DGL step: features see only the unimodal losses, fc_out only loss_f
    np.testing.assert_allclose(_np(dxd), gx.astype(np.float64) @ W[:, :dx_], atol=1e-4)
    np.testing.assert_allclose(_np(dyd), gy.astype(np.float64) @ W[:, dx_:], atol=1e-4)
    np.testing.assert_allclose(_np(dW), go.astype(np.float64).T @ np.concatenate([x, y], 1), atol=1e-4 * max(1, B // 8))
    np.testing.assert_allclose(_np(db), go.astype(np.float64).sum(0), atol=1e-5 * max(1, B // 8))
    if dy_ not in (512, 768, 1024) or dx_ != 512:
        with pytest.raises(L.GdlError):
            L.call("gdl_head_uni_dfeat_w", L.ptr(yd), Wd.data_ptr() + dx_ * 4, dx_ + dy_, L.ptr(bd), L.ptr(torch.zeros(B, dtype=torch.int64, device=DEV)),
                   2.0, L.ptr(dyd), B, n, dy_, st)
        return
    # one modality's path in one launch == head forward -> cross-entropy (x 2.0) -> head backward, same bits
    lab = torch.from_numpy(rng.integers(0, n, B).astype(np.int64)).to(DEV)
    loss = torch.zeros(3, device=DEV)
    ga, gv = torch.empty((B, n), device=DEV), torch.empty((B, n), device=DEV)
    L.call("gdl_softmax_ce3", L.ptr(out), L.ptr(xo), L.ptr(yo), L.ptr(lab), 1.0, 2.0, 2.0, L.ptr(loss), None, L.ptr(ga), L.ptr(gv), B, n, st)
    L.call("gdl_head_concat_xy_bwd", L.ptr(xd), L.ptr(yd), L.ptr(Wd), L.ptr(ga), L.ptr(gv), None, 0, 0, L.ptr(dxd), L.ptr(dyd), None, None,
           B, n, dx_, dy_, st)
    ux, uy = torch.full_like(xd, float("nan")), torch.full_like(yd, float("nan"))
    L.call("gdl_head_uni_dfeat_w", L.ptr(xd), L.ptr(Wd), dx_ + dy_, L.ptr(bd), L.ptr(lab), 2.0, L.ptr(ux), B, n, dx_, st)
    L.call("gdl_head_uni_dfeat_w", L.ptr(yd), Wd.data_ptr() + dx_ * 4, dx_ + dy_, L.ptr(bd), L.ptr(lab), 2.0, L.ptr(uy), B, n, dy_, st)
    torch.cuda.synchronize()
    assert torch.equal(ux.view(torch.int32), dxd.view(torch.int32)) and torch.equal(uy.view(torch.int32), dyd.view(torch.int32))
