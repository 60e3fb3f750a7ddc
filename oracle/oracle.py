"""CPU oracle for the DGL audio-visual training step.

TEST INFRASTRUCTURE ONLY -- see the header of gdl_oracle.c.  Only tests/,
`__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg import this.

Model logic (which operator is applied where) is restated here in Python on
numpy arrays, following the reference file:line cited at each function; the
arithmetic itself runs in `gdl_oracle.c` (plain C, fp32, OpenMP).

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were
captured from the imported reference (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgdl_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gdl_oracle.c")
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    # x86-64-v3 (AVX2+FMA): the .so is built in the build container and travels to the GPU box
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-fvisibility=hidden", src, "-o", _SO, "-lm"]
    subprocess.check_call(cmd)
    return _SO


_lib = None
_f = ctypes.POINTER(ctypes.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_softmax_ce.restype = ctypes.c_double
        _lib.orc_sumsq.restype = ctypes.c_double
        _lib.orc_abs_mean.restype = ctypes.c_double
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ primitives
def conv2d_fwd(x, w, stride, pad):
    x, w = _c(x), _c(w)
    N, C, H, W = x.shape
    K, C2, R, S = w.shape
    assert C == C2
    P, Q = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    y = np.empty((N, K, P, Q), np.float32)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(y), N, C, H, W, K, R, S, stride, pad)
    return y


def conv2d_bwd_data(dy, w, xshape, stride, pad):
    dy, w = _c(dy), _c(w)
    N, C, H, W = xshape
    K, _, R, S = w.shape
    dx = np.empty(xshape, np.float32)
    lib().orc_conv2d_bwd_data(_p(dy), _p(w), _p(dx), N, C, H, W, K, R, S, stride, pad)
    return dx


def conv2d_bwd_weight(dy, x, wshape, stride, pad):
    dy, x = _c(dy), _c(x)
    N, C, H, W = x.shape
    K, _, R, S = wshape
    dw = np.empty(wshape, np.float32)
    lib().orc_conv2d_bwd_weight(_p(dy), _p(x), _p(dw), N, C, H, W, K, R, S, stride, pad)
    return dw


def bn_fwd_train(x, gamma, beta, rm, rv, eps=1e-5, momentum=0.1):
    """rm / rv are updated IN PLACE (like the module buffers)."""
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    mean = np.empty(C, np.float32)
    invstd = np.empty(C, np.float32)
    lib().orc_bn_fwd_train(_p(x), _p(_c(gamma)), _p(_c(beta)), _p(y), _p(mean), _p(invstd), _p(rm), _p(rv), N, C, H * W,
                           ctypes.c_float(eps), ctypes.c_float(momentum))
    return y, mean, invstd


def bn_fwd_eval(x, gamma, beta, rm, rv, eps=1e-5):
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_bn_fwd_eval(_p(x), _p(_c(gamma)), _p(_c(beta)), _p(y), _p(_c(rm)), _p(_c(rv)), N, C, H * W,
                          ctypes.c_float(eps))
    return y


def bn_bwd(dy, x, gamma, mean, invstd):
    dy, x = _c(dy), _c(x)
    N, C, H, W = x.shape
    dx = np.empty_like(x)
    dg = np.empty(C, np.float32)
    db = np.empty(C, np.float32)
    lib().orc_bn_bwd(_p(dy), _p(x), _p(_c(gamma)), _p(mean), _p(invstd), _p(dx), _p(dg), _p(db), N, C, H * W)
    return dx, dg, db


def relu_(x):
    lib().orc_relu_fwd(_p(x), ctypes.c_size_t(x.size))
    return x


def relu_bwd(dy, y):
    dy, y = _c(dy), _c(y)
    dx = np.empty_like(dy)
    lib().orc_relu_bwd(_p(dy), _p(y), _p(dx), ctypes.c_size_t(dy.size))
    return dx


def maxpool_fwd(x):
    x = _c(x)
    N, C, H, W = x.shape
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = np.empty((N, C, P, Q), np.float32)
    idx = np.empty((N, C, P, Q), np.int32)
    lib().orc_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), N, C, H, W)
    return y, idx


def maxpool_bwd(dy, idx, xshape):
    dy = _c(dy)
    N, C, H, W = xshape
    dx = np.empty(xshape, np.float32)
    lib().orc_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), N, C, H, W)
    return dx


def avgpool_fwd(x, B, T):
    x = _c(x)
    BT, C, H, W = x.shape
    assert BT == B * T
    y = np.empty((B, C), np.float32)
    lib().orc_avgpool_fwd(_p(x), _p(y), B, T, C, H * W)
    return y


def avgpool_bwd(dy, xshape, B, T):
    dy = _c(dy)
    BT, C, H, W = xshape
    dx = np.empty(xshape, np.float32)
    lib().orc_avgpool_bwd(_p(dy), _p(dx), B, T, C, H * W)
    return dx


def linear_fwd(x, w, b):
    x, w = _c(x), _c(w)
    B, I = x.shape
    O = w.shape[0]
    y = np.empty((B, O), np.float32)
    lib().orc_linear_fwd(_p(x), _p(w), _p(_c(b)) if b is not None else None, _p(y), B, I, O)
    return y


def linear_bwd(dy, x, w, need_dx=True):
    dy, x, w = _c(dy), _c(x), _c(w)
    B, I = x.shape
    O = w.shape[0]
    dx = np.empty((B, I), np.float32) if need_dx else None
    dw = np.zeros((O, I), np.float32)
    db = np.zeros(O, np.float32)
    lib().orc_linear_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, I, O)
    return dx, dw, db


def softmax_ce(logits, labels, scale=1.0):
    logits = _c(logits)
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    B, n = logits.shape
    d = np.empty_like(logits)
    loss = lib().orc_softmax_ce(_p(logits), _p(labels), _p(d), B, n, ctypes.c_double(scale))
    return float(loss), d


def valid_counts(out, out_a, out_v, labels, n_classes):
    """Per-class counters of valid() (/root/reference/main_dgl.py:188-219), one batch: softmax, np.argmax per
    sample, num[label] += 1 and acc*[label] += 1 where the prediction equals the label."""
    num, acc, acc_a, acc_v = (np.zeros(n_classes, dtype=np.int64) for _ in range(4))

    def softmax(z):
        e = np.exp(z - z.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)

    pred, pa, pv = softmax(out), softmax(out_a), softmax(out_v)
    for i in range(out.shape[0]):
        lab = int(labels[i])
        num[lab] += 1
        if int(np.argmax(pred[i])) == lab:
            acc[lab] += 1
        if int(np.argmax(pv[i])) == lab:
            acc_v[lab] += 1
        if int(np.argmax(pa[i])) == lab:
            acc_a[lab] += 1
    return num, acc, acc_a, acc_v


def log_spectrogram(wave, n_fft, hop, pad_mode="constant"):
    """The datasets' audio feature (dataset/CramedDataset.py:62-66, KSDataset.py:144-149): clip to [-1, 1],
    librosa.stft(n_fft, hop_length=hop), log(|X| + 1e-7).  librosa is not vendored in the reference and is absent
    here; this restates its published algorithm (librosa.stft with its defaults: win_length = n_fft, periodic Hann
    window, center=True with `pad_mode` padding of n_fft//2 samples -- 'constant' since librosa 0.10, 'reflect' before)
    in float64.  wave: [B][L] float; returns float32 [B][n_fft//2+1][1 + L//hop].  Pinned against torch.stft
    (an independent implementation) by tests/test_oracle_golden.py::test_log_spectrogram_vs_torch."""
    w = np.clip(np.asarray(wave, dtype=np.float64), -1.0, 1.0)
    pad = n_fft // 2
    wp = np.pad(w, ((0, 0), (pad, pad)), mode=pad_mode)
    frames = 1 + w.shape[1] // hop
    idx = np.arange(frames)[:, None] * hop + np.arange(n_fft)[None, :]
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    X = np.fft.rfft(wp[:, idx] * hann, axis=-1)  # [B][frames][bins]
    return np.log(np.abs(X) + 1e-7).transpose(0, 2, 1).astype(np.float32)


def normalize_frames(frames_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """transforms.ToTensor() + transforms.Normalize(mean, std) (dataset/CramedDataset.py:77-81) on decoded uint8
    frames [N][H][W][3] -> float32 [N][3][H][W]; the same fp32 operations in the same order as torchvision."""
    x = np.asarray(frames_u8).astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def sumsq(g):
    g = _c(g)
    return float(lib().orc_sumsq(_p(g), ctypes.c_size_t(g.size)))


def abs_mean(g):
    g = _c(g)
    return float(lib().orc_abs_mean(_p(g), ctypes.c_size_t(g.size)))


def sgd_(p, g, buf, lr, mu, wd, first):
    lib().orc_sgd(_p(p), _p(_c(g)), _p(buf), ctypes.c_size_t(p.size), ctypes.c_float(lr), ctypes.c_float(mu),
                  ctypes.c_float(wd), int(first))


# ------------------------------------------------------------------ ResNet18 encoder
class _BN:
    def __init__(self, P, Bf, name):
        self.g, self.b = P[name + ".weight"], P[name + ".bias"]
        self.rm, self.rv = Bf[name + ".running_mean"], Bf[name + ".running_var"]
        self.Bf, self.name = Bf, name

    def fwd(self, x, train):
        if train:
            y, self.mean, self.invstd = bn_fwd_train(x, self.g, self.b, self.rm, self.rv)
            self.Bf[self.name + ".num_batches_tracked"] += 1
            self.x = x
            return y
        return bn_fwd_eval(x, self.g, self.b, self.rm, self.rv)

    def bwd(self, dy, G):
        dx, dg, db = bn_bwd(dy, self.x, self.g, self.mean, self.invstd)
        G[self.name + ".weight"] = dg
        G[self.name + ".bias"] = db
        return dx


class ResNet18:
    """`resnet18(modality, args)` of models/backbone.py:255-257: `ResNet(BasicBlock,[2,2,2,2])`
    (:75-156) whose forward (:158-201) returns the un-pooled layer4 map."""

    def __init__(self, params, buffers, prefix, modality):
        self.P, self.B, self.pre, self.modality = params, buffers, prefix, modality

    def forward(self, x, train=True):
        P, pre = self.P, self.pre
        self.train = train
        if self.modality == "visual":  # backbone.py:162-164
            B, C, T, H, W = x.shape
            x = np.ascontiguousarray(x.transpose(0, 2, 1, 3, 4)).reshape(B * T, C, H, W)
        self.x0 = x
        self.bn1 = _BN(P, self.B, pre + "bn1")
        y = conv2d_fwd(x, P[pre + "conv1.weight"], 2, 3)  # :96-101,166
        y = relu_(self.bn1.fwd(y, train))  # :171-172
        self.a0 = y
        y, self.pool_idx = maxpool_fwd(y)  # :173
        self.blocks = []
        inpl = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):  # :175-178
            for bi in range(2):
                stride = 2 if (bi == 0 and li > 1) else 1
                y = self._block_fwd(y, f"{pre}layer{li}.{bi}", stride, stride != 1 or inpl != planes, train)
                inpl = planes
        return y

    def _block_fwd(self, x, name, stride, has_ds, train):
        """BasicBlock.forward, backbone.py:52-68."""
        P = self.P
        c = {"name": name, "stride": stride, "has_ds": has_ds, "x": x}
        c["bn1"] = _BN(P, self.B, name + ".bn1")
        c["bn2"] = _BN(P, self.B, name + ".bn2")
        out = conv2d_fwd(x, P[name + ".conv1.weight"], stride, 1)
        out = relu_(c["bn1"].fwd(out, train))
        c["a1"] = out
        out = conv2d_fwd(out, P[name + ".conv2.weight"], 1, 1)
        out = c["bn2"].fwd(out, train)
        if has_ds:  # :141-145
            c["bnd"] = _BN(P, self.B, name + ".downsample.1")
            idn = c["bnd"].fwd(conv2d_fwd(x, P[name + ".downsample.0.weight"], stride, 0), train)
        else:
            idn = x
        out = relu_(out + idn)  # :65-66
        c["z"] = out
        self.blocks.append(c)
        return out

    def backward(self, dy):
        """Gradients of every encoder parameter for upstream gradient `dy` on the layer4 map.
        The input needs no gradient (main_dgl.py: inputs do not require grad)."""
        P, pre, G = self.P, self.pre, {}
        for c in reversed(self.blocks):
            name = c["name"]
            d = relu_bwd(dy, c["z"])
            d2 = c["bn2"].bwd(d, G)
            G[name + ".conv2.weight"] = conv2d_bwd_weight(d2, c["a1"], P[name + ".conv2.weight"].shape, 1, 1)
            d1 = conv2d_bwd_data(d2, P[name + ".conv2.weight"], c["a1"].shape, 1, 1)
            d1 = c["bn1"].bwd(relu_bwd(d1, c["a1"]), G)
            G[name + ".conv1.weight"] = conv2d_bwd_weight(d1, c["x"], P[name + ".conv1.weight"].shape, c["stride"], 1)
            dx = conv2d_bwd_data(d1, P[name + ".conv1.weight"], c["x"].shape, c["stride"], 1)
            if c["has_ds"]:
                dd = c["bnd"].bwd(d, G)
                G[name + ".downsample.0.weight"] = conv2d_bwd_weight(dd, c["x"], P[name + ".downsample.0.weight"].shape,
                                                                     c["stride"], 0)
                dx += conv2d_bwd_data(dd, P[name + ".downsample.0.weight"], c["x"].shape, c["stride"], 0)
            else:
                dx += d
            dy = dx
        d = maxpool_bwd(dy, self.pool_idx, self.a0.shape)
        d = self.bn1.bwd(relu_bwd(d, self.a0), G)
        G[pre + "conv1.weight"] = conv2d_bwd_weight(d, self.x0, P[pre + "conv1.weight"].shape, 2, 3)
        return G


# ------------------------------------------------------------------ fusion heads
def concat_dgl_fwd(x, y, W, b):
    """ConcatFusion_DGL.forward, fusion_modules.py:51-59 -> (x_out, y_out, output)."""
    out = linear_fwd(np.concatenate([x, y], 1), W, b)  # :53-56 (detached input)
    x_out = linear_fwd(np.concatenate([x, np.zeros_like(y)], 1), W, b)  # :57 (x and y may differ in width: 512 + C heads)
    y_out = linear_fwd(np.concatenate([np.zeros_like(x), y], 1), W, b)  # :58
    return x_out, y_out, out


def concat_dgl_bwd(x, y, W, g_x_out, g_y_out, g_out):
    """Autograd of the three Linear calls above for upstream gradients on
    (x_out, y_out, output); any of them may be None.  `output` sees a detached
    input, so it contributes to dW/db only."""
    nx = x.shape[1]
    dW = np.zeros_like(W)
    db = np.zeros(W.shape[0], np.float32)
    dx = np.zeros_like(x)
    dy = np.zeros_like(y)
    if g_x_out is not None:
        d, w_, b_ = linear_bwd(g_x_out, np.concatenate([x, np.zeros_like(y)], 1), W)
        dx += d[:, :nx]
        dW += w_
        db += b_
    if g_y_out is not None:
        d, w_, b_ = linear_bwd(g_y_out, np.concatenate([np.zeros_like(x), y], 1), W)
        dy += d[:, nx:]
        dW += w_
        db += b_
    if g_out is not None:
        _, w_, b_ = linear_bwd(g_out, np.concatenate([x, y], 1), W, need_dx=False)
        dW += w_
        db += b_
    return dx, dy, dW, db


def sum_dgl_fwd(x, y, Wx, bx, Wy, by):
    """SumFusion_DGL.forward, fusion_modules.py:22-30 -> (outx, outy, output)."""
    outx = linear_fwd(x, Wx, bx)  # :24
    outy = linear_fwd(y, Wy, by)  # :25
    out = linear_fwd(x, Wx, bx) + linear_fwd(y, Wy, by)  # :27-29 (detached inputs)
    return outx, outy, out


def sum_dgl_bwd(x, y, Wx, Wy, g_x_out, g_y_out, g_out):
    """Autograd of the four Linear calls above for upstream gradients on (outx, outy, output); any may be None.
    `output` sees detached inputs: it contributes to the weights / biases only."""
    dx, dy = np.zeros_like(x), np.zeros_like(y)
    dWx, dWy = np.zeros_like(Wx), np.zeros_like(Wy)
    dbx, dby = np.zeros(Wx.shape[0], np.float32), np.zeros(Wy.shape[0], np.float32)
    if g_x_out is not None:
        d, w, b = linear_bwd(g_x_out, x, Wx)
        dx += d
        dWx += w
        dbx += b
    if g_y_out is not None:
        d, w, b = linear_bwd(g_y_out, y, Wy)
        dy += d
        dWy += w
        dby += b
    if g_out is not None:
        _, w, b = linear_bwd(g_out, x, Wx)
        dWx += w
        dbx += b
        _, w, b = linear_bwd(g_out, y, Wy)
        dWy += w
        dby += b
    return dx, dy, dWx, dbx, dWy, dby


def _sigmoid(v):
    return (1.0 / (1.0 + np.exp(-v.astype(np.float64)))).astype(np.float32)


def gated_dgl_fwd(x, y, W1, b1, W2, b2, Wo, bo):
    """GatedFusion_DGL.forward with x_gate=True, fusion_modules.py:232-250 -> (out_x, out_y, output, hx, hy)."""
    hx = linear_fwd(x, W1, b1)  # :234
    hy = linear_fwd(y, W2, b2)  # :235
    out = linear_fwd(_sigmoid(hx) * hy, Wo, bo)  # :241-243 on detached hx, hy
    ox = linear_fwd(_sigmoid(hx) * hx, Wo, bo)  # :246-247
    oy = linear_fwd(_sigmoid(hy) * hy, Wo, bo)  # :248-249
    return ox, oy, out, hx, hy


def gated_dgl_bwd(x, y, hx, hy, W1, W2, Wo, g_x_out, g_y_out, g_out):
    """Autograd of the above for upstream gradients on (out_x, out_y, output); any may be None.  `output` uses
    detached hidden vectors: it reaches fc_out only.  Returns dx, dy, {fc_x, fc_y, fc_out gradients}."""
    B = x.shape[0]
    dx, dy = np.zeros_like(x), np.zeros_like(y)
    G = {"fc_x.weight": np.zeros_like(W1), "fc_x.bias": np.zeros(W1.shape[0], np.float32),
         "fc_y.weight": np.zeros_like(W2), "fc_y.bias": np.zeros(W2.shape[0], np.float32),
         "fc_out.weight": np.zeros_like(Wo), "fc_out.bias": np.zeros(Wo.shape[0], np.float32)}

    def uni(g, h, inp, W, kx):
        s = _sigmoid(h)
        dz, dWo, dbo = linear_bwd(g, (s * h).astype(np.float32), Wo)  # fc_out applied to swish(h)
        G["fc_out.weight"] += dWo
        G["fc_out.bias"] += dbo
        dh = (dz * (s * (1.0 + h * (1.0 - s)))).astype(np.float32)
        d, dW, db = linear_bwd(dh, inp, W)
        G[kx + ".weight"] += dW
        G[kx + ".bias"] += db
        return d

    if g_x_out is not None:
        dx += uni(g_x_out, hx, x, W1, "fc_x")
    if g_y_out is not None:
        dy += uni(g_y_out, hy, y, W2, "fc_y")
    if g_out is not None:
        _, dWo, dbo = linear_bwd(g_out, (_sigmoid(hx) * hy).astype(np.float32), Wo)
        G["fc_out.weight"] += dWo
        G["fc_out.bias"] += dbo
    return dx, dy, G


def film_dgl_fwd(x, y, Wfc, bfc, Wo, bo):
    """FiLM_DGL.forward, fusion_modules.py:140-178 -> (z_x, z_y, output, (hx, hf, hy)).  fc acts on the flattened
    outer product: fc(flatten(u (x) v))[k] = u^T W_k v + b_k with W_k = Wfc[k].reshape(512, 512)."""
    A = Wfc.reshape(512 * 512, 512)  # row (k, i), column j
    Tx = (A @ x.T).reshape(512, 512, -1)  # [k][i][b] = (W_k x_b)[i]
    Ty = (A @ y.T).reshape(512, 512, -1)
    hf = np.einsum("bi,kib->bk", x, Ty) + bfc  # :153-158, detached x, y
    hx = np.einsum("bi,kib->bk", x, Tx) + bfc  # :160-163
    hy = np.einsum("bi,kib->bk", y, Ty) + bfc  # :165-168
    hx, hf, hy = (h.astype(np.float32) for h in (hx, hf, hy))
    return linear_fwd(hx, Wo, bo), linear_fwd(hy, Wo, bo), linear_fwd(hf, Wo, bo), (hx, hf, hy)


def film_dgl_bwd(x, y, Wfc, Wo, hidden, g_x_out, g_y_out, g_out, want_fc=True):
    """Autograd of the above for upstream gradients on (z_x, z_y, output); any may be None.  `output` uses detached
    features: it reaches fc / fc_out only.  Returns dx, dy, {fc, fc_out gradients}."""
    hx, hf, hy = hidden
    W3 = Wfc.reshape(512, 512, 512)
    dx, dy = np.zeros_like(x), np.zeros_like(y)
    G = {"fc_out.weight": np.zeros_like(Wo), "fc_out.bias": np.zeros(Wo.shape[0], np.float32)}
    if want_fc:
        G["fc.weight"] = np.zeros_like(Wfc)
        G["fc.bias"] = np.zeros(512, np.float32)

    def through(g, h, u, v, reach_u, reach_v):
        dh, dWo, dbo = linear_bwd(g, h, Wo)
        G["fc_out.weight"] += dWo
        G["fc_out.bias"] += dbo
        du, dv = np.zeros_like(u), np.zeros_like(v)
        for b in range(u.shape[0]):
            Mb = np.tensordot(dh[b], W3, axes=(0, 0))  # sum_k dh[b,k] W_k
            if reach_u:
                du[b] = Mb @ v[b]
            if reach_v:
                dv[b] = Mb.T @ u[b]
        if want_fc:
            G["fc.weight"] += np.einsum("bk,bi,bj->kij", dh, u, v, optimize=True).reshape(512, -1).astype(np.float32)
            G["fc.bias"] += dh.sum(0)
        return du, dv

    if g_x_out is not None:
        du, dv = through(g_x_out, hx, x, x, True, True)
        dx += du + dv
    if g_y_out is not None:
        du, dv = through(g_y_out, hy, y, y, True, True)
        dy += du + dv
    if g_out is not None:
        through(g_out, hf, x, y, False, False)
    return dx, dy, G


def concat_fwd(x, y, W, b):
    """ConcatFusion.forward, fusion_modules.py:38-42 -> output."""
    return linear_fwd(np.concatenate([x, y], 1), W, b)


def concat_bwd(x, y, W, g_out):
    d, dW, db = linear_bwd(g_out, np.concatenate([x, y], 1), W)
    return d[:, :512].copy(), d[:, 512:].copy(), dW, db


# ------------------------------------------------------------------ full model + step
class AVModel:
    """AVClassifier_DGL (models/basic_model.py:10-86), full-modality path, with the
    concat heads.  `params` / `buffers` are ordered dicts of float32 arrays in the
    reference's state_dict naming; they are updated in place by `train_step`."""

    def __init__(self, params, buffers, mode="dgl"):
        self.P, self.B, self.mode = params, buffers, mode
        self.audio = ResNet18(params, buffers, "audio_net.", "audio")
        self.visual = ResNet18(params, buffers, "visual_net.", "visual")
        self.mom = {}
        self.steps = 0

    def forward(self, spec, image, train=True):
        """`model(spec.unsqueeze(1).float(), image.float())`, main_dgl.py:100 /
        basic_model.py:65-86.  Returns (out, a_out, v_out)."""
        audio = np.ascontiguousarray(spec[:, None].astype(np.float32))
        B, T = image.shape[0], image.shape[2]
        a = self.audio.forward(audio, train)
        v = self.visual.forward(np.ascontiguousarray(image, dtype=np.float32), train)
        self.a_map_shape, self.v_map_shape, self.BT = a.shape, v.shape, (B, T)
        self.fa = avgpool_fwd(a, B, 1)  # :78
        self.fv = avgpool_fwd(v, B, T)  # :73-79
        if "fusion_module.fc.weight" in self.P:  # FiLM_DGL
            P = self.P
            a_out, v_out, out, self.film_hidden = film_dgl_fwd(self.fa, self.fv, P["fusion_module.fc.weight"],
                                                               P["fusion_module.fc.bias"], P["fusion_module.fc_out.weight"],
                                                               P["fusion_module.fc_out.bias"])
            return out, a_out, v_out
        if "fusion_module.fc_x.weight" in self.P and "fusion_module.fc_out.weight" in self.P:  # GatedFusion_DGL
            P = self.P
            a_out, v_out, out, self.hx, self.hy = gated_dgl_fwd(
                self.fa, self.fv, P["fusion_module.fc_x.weight"], P["fusion_module.fc_x.bias"], P["fusion_module.fc_y.weight"],
                P["fusion_module.fc_y.bias"], P["fusion_module.fc_out.weight"], P["fusion_module.fc_out.bias"])
            return out, a_out, v_out
        if "fusion_module.fc_x.weight" in self.P:  # SumFusion_DGL (basic_model.py:29-30)
            P = self.P
            a_out, v_out, out = sum_dgl_fwd(self.fa, self.fv, P["fusion_module.fc_x.weight"], P["fusion_module.fc_x.bias"],
                                            P["fusion_module.fc_y.weight"], P["fusion_module.fc_y.bias"])
            return out, a_out, v_out
        W, b = self.P["fusion_module.fc_out.weight"], self.P["fusion_module.fc_out.bias"]
        if self.mode == "dgl":
            a_out, v_out, out = concat_dgl_fwd(self.fa, self.fv, W, b)
            return out, a_out, v_out
        return concat_fwd(self.fa, self.fv, W, b), None, None

    def train_step(self, spec, image, label, alpha, lr, momentum=0.9, wd=1e-4, max_norm=40.0):
        """Body of the reference step, main_dgl.py:97-154."""
        P = self.P
        B, T = image.shape[0], image.shape[2]
        out, out_a, out_v = self.forward(spec, image, True)
        r = {"out": out}
        G = {}
        film = "fusion_module.fc.weight" in P
        gated = "fusion_module.fc_x.weight" in P and "fusion_module.fc_out.weight" in P
        sum_head = "fusion_module.fc_x.weight" in P and not gated
        W = None if sum_head else P["fusion_module.fc_out.weight"]
        if film:
            Wfc = P["fusion_module.fc.weight"]
            loss_v, g_v = softmax_ce(out_v, label, alpha)
            loss_a, g_a = softmax_ce(out_a, label, alpha)
            loss_f, g_f = softmax_ce(out, label, 1.0)
            # phase 1 (:110): encoders get the unimodal gradients through fc_out and the quadratic forms u^T W_k u; the
            # head gradients it produces (incl. the 134 M of fc.weight) are dropped (:114-119) -- only their norm is kept
            dfa, dfv, Gu = film_dgl_bwd(self.fa, self.fv, Wfc, W, self.film_hidden, g_a, g_v, None)
            r["dropped_head_gradnorm"] = float(np.sqrt(sum(sumsq(v) for v in Gu.values())))
            del Gu
            # phase 2 (:122): loss_f on detached features reaches fc and fc_out
            _, _, Gf = film_dgl_bwd(self.fa, self.fv, Wfc, W, self.film_hidden, None, None, g_f)
            r.update(out_a=out_a, out_v=out_v, loss_a=loss_a, loss_v=loss_v)
            G["fusion_module.fc.weight"], G["fusion_module.fc.bias"] = Gf["fc.weight"], Gf["fc.bias"]
            dW, db = Gf["fc_out.weight"], Gf["fc_out.bias"]
        elif gated:
            W1, W2 = P["fusion_module.fc_x.weight"], P["fusion_module.fc_y.weight"]
            loss_v, g_v = softmax_ce(out_v, label, alpha)
            loss_a, g_a = softmax_ce(out_a, label, alpha)
            loss_f, g_f = softmax_ce(out, label, 1.0)
            # phase 1 (:110): encoders get the unimodal gradients through fc_out, swish and fc_x / fc_y; every
            # fusion_module.* gradient is then dropped (:114-119)
            dfa, dfv, Gu = gated_dgl_bwd(self.fa, self.fv, self.hx, self.hy, W1, W2, W, g_a, g_v, None)
            r["dropped_head_gradnorm"] = float(np.sqrt(sum(sumsq(v) for v in Gu.values())))
            # phase 2 (:122): loss_f reaches fc_out only -- fc_x / fc_y keep grad None and are never updated
            _, _, Gf = gated_dgl_bwd(self.fa, self.fv, self.hx, self.hy, W1, W2, W, None, None, g_f)
            r.update(out_a=out_a, out_v=out_v, loss_a=loss_a, loss_v=loss_v)
            dW, db = Gf["fc_out.weight"], Gf["fc_out.bias"]
        elif sum_head:
            Wx, Wy = P["fusion_module.fc_x.weight"], P["fusion_module.fc_y.weight"]
            loss_v, g_v = softmax_ce(out_v, label, alpha)
            loss_a, g_a = softmax_ce(out_a, label, alpha)
            loss_f, g_f = softmax_ce(out, label, 1.0)
            # phase 1 (:110): encoders get alpha*d(CE(outx)+CE(outy)); every fusion_module.* gradient is then dropped
            dfa, dfv, dWx_u, dbx_u, dWy_u, dby_u = sum_dgl_bwd(self.fa, self.fv, Wx, Wy, g_a, g_v, None)
            r["dropped_head_gradnorm"] = float(np.sqrt(sumsq(dWx_u) + sumsq(dbx_u) + sumsq(dWy_u) + sumsq(dby_u)))
            # phase 2 (:122): loss_f reaches fc_x / fc_y parameters only (detached features)
            _, _, dWx, dbx, dWy, dby = sum_dgl_bwd(self.fa, self.fv, Wx, Wy, None, None, g_f)
            r.update(out_a=out_a, out_v=out_v, loss_a=loss_a, loss_v=loss_v)
            G["fusion_module.fc_x.weight"], G["fusion_module.fc_x.bias"] = dWx, dbx
            G["fusion_module.fc_y.weight"], G["fusion_module.fc_y.bias"] = dWy, dby
        elif self.mode == "dgl":
            loss_v, g_v = softmax_ce(out_v, label, alpha)  # :102,108
            loss_a, g_a = softmax_ce(out_a, label, alpha)  # :103,108
            loss_f, g_f = softmax_ce(out, label, 1.0)  # :104
            # phase 1: loss_unimodal.backward(retain_graph=True)  (:110)
            dfa, dfv, dW_uni, db_uni = concat_dgl_bwd(self.fa, self.fv, W, g_a, g_v, None)
            r["dropped_head_gradnorm"] = float(np.sqrt(sumsq(dW_uni) + sumsq(db_uni)))
            # drop of the fusion-head grads (:114-119), then phase 2: loss_f.backward() (:122)
            _, _, dW, db = concat_dgl_bwd(self.fa, self.fv, W, None, None, g_f)
            r.update(out_a=out_a, out_v=out_v, loss_a=loss_a, loss_v=loss_v)
        else:
            loss_f, g_f = softmax_ce(out, label, 1.0)
            dfa, dfv, dW, db = concat_bwd(self.fa, self.fv, W, g_f)
        r["loss_f"] = loss_f
        if not sum_head:
            G["fusion_module.fc_out.weight"], G["fusion_module.fc_out.bias"] = dW, db
        G.update(self.audio.backward(avgpool_bwd(dfa, self.a_map_shape, B, 1)))
        G.update(self.visual.backward(avgpool_bwd(dfv, self.v_map_shape, B, T)))
        # clip_grad_norm_(model.parameters(), 40, 2)  (:129); fc_auxi has no grad
        total = float(np.sqrt(sum(sumsq(g) for g in G.values())))
        r["total_norm"] = total
        coef = min(1.0, max_norm / (total + 1e-6))
        for k in G:
            G[k] = (G[k] * np.float32(coef)).astype(np.float32)
        # logged sums (:132-143)
        r["audio_grad_sum"] = sum(abs_mean(G[k]) for k in P if k.startswith("audio_net."))
        r["visual_grad_sum"] = sum(abs_mean(G[k]) for k in P if k.startswith("visual_net."))
        r["grads"] = G
        # optimizer.step()  (:154)
        for k in P:
            if k not in G:
                continue  # grad is None -> SGD skips the parameter
            first = k not in self.mom
            if first:
                self.mom[k] = np.zeros_like(P[k])
            sgd_(P[k], G[k], self.mom[k], lr, momentum, wd, first)
        self.steps += 1
        return r
