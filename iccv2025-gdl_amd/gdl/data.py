"""Device-side stages of the reference's input pipeline over gdl_logspec / gdl_frames_normalize.

The reference computes both per sample on DataLoader workers (dataset/CramedDataset.py:58-95, KSDataset.py:136-190,
VGGSoundDataset.py:110-160): a log-magnitude librosa STFT of the clipped waveform and ToTensor + Normalize of the
decoded frames.  Here a whole batch is one kernel launch each; file decoding, resampling, tiling / cropping of the
waveform, image resizing and the random crops stay on the host, exactly where the reference has them.

Nothing is computed on the CPU: both functions need device tensors and the built library.
"""
import ctypes

import torch

from . import _lib as L

PAD_MODES = {"constant": 0, "reflect": 1}  # GDL_PAD_CONSTANT / GDL_PAD_REFLECT
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # the constants of every dataset of the reference
IMAGENET_STD = (0.229, 0.224, 0.225)

# (n_fft, hop_length) of the reference's datasets
STFT_CREMAD = (512, 353)   # CramedDataset.py:65 -> [257, 188] for 3 s at 22 050 Hz
STFT_KS = (256, 128)       # KSDataset.py:148, VGGSoundDataset.py:121 -> [129, 626] for 5 s at 16 kHz
STFT_AVE = (512, 256)      # AVEDataset.py:86, Audioset.py:149


def log_spectrogram(wave, n_fft=512, hop_length=353, pad_mode="constant", out=None):
    """`np.log(np.abs(librosa.stft(clip(wave, -1, 1), n_fft=n_fft, hop_length=hop_length)) + 1e-7)` for a batch.

    wave: float32 device tensor [B, n_samples] (or [n_samples]); returns float32 [B, n_fft//2+1, 1 + n_samples//hop]
    -- `spec` as main_dgl.py:108 receives it (it adds the channel axis itself).  pad_mode is librosa's: 'constant'
    (librosa >= 0.10, zeros) or 'reflect' (older releases); the reference does not pin a librosa version."""
    if pad_mode not in PAD_MODES:
        raise ValueError(f"gdl: pad_mode must be one of {sorted(PAD_MODES)}, not {pad_mode!r}")
    squeeze = wave.dim() == 1
    w = wave.reshape(1, -1) if squeeze else wave
    if w.dim() != 2 or w.dtype != torch.float32 or not w.is_cuda:
        raise ValueError("gdl: wave must be a float32 device tensor [B, n_samples]")
    w = w.contiguous()
    B, n = w.shape
    frames = L.load().gdl_logspec_frames(n, hop_length)
    if out is None:
        out = torch.empty(B, n_fft // 2 + 1, frames, dtype=torch.float32, device=w.device)
    elif tuple(out.shape) != (B, n_fft // 2 + 1, frames) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("gdl: out must be a contiguous float32 tensor [B, n_fft//2+1, frames]")
    L.call("gdl_logspec", L.ptr(w), B, n, n_fft, hop_length, PAD_MODES[pad_mode], L.ptr(out), L.cur_stream())
    return out[0] if squeeze else out


def normalize_frames(frames_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """transforms.ToTensor() + transforms.Normalize(mean, std) for a stack of decoded frames.

    frames_u8: uint8 device tensor [..., H, W, 3] (e.g. [B, T, 224, 224, 3]); returns float32 [..., 3, H, W]."""
    f = frames_u8
    if f.dtype != torch.uint8 or not f.is_cuda or f.dim() < 3 or f.shape[-1] != 3:
        raise ValueError("gdl: frames must be a uint8 device tensor [..., H, W, 3]")
    f = f.contiguous()
    lead, (H, W) = tuple(f.shape[:-3]), f.shape[-3:-1]
    n_img = 1
    for d in lead:
        n_img *= d
    if out is None:
        out = torch.empty(*lead, 3, H, W, dtype=torch.float32, device=f.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    L.call("gdl_frames_normalize", L.ptr(f), n_img, H, W, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
           L.ptr(out), L.cur_stream())
    return out
