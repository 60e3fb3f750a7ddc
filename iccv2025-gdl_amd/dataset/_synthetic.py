"""Synthetic stand-ins for the reference's dataset classes (import-resolving shims, SURVEY 8(b) "Who calls").

`main_dgl.py` imports five dataset classes at module top (/root/reference/main_dgl.py:12-18).  The real ones read
audio / video files and need librosa / torchvision / PIL -- data handling is OUT of scope here (last table).
These classes have the same names, constructor signature `(args, mode='train')` and `__getitem__` return order
`(spectrogram, images, label)` with the reference's tensor shapes, but produce deterministic random tensors: they let
the unmodified script start and train on synthetic data over the drop-in modules; they are NOT a data pipeline.
Length: GDL_SYNTH_LEN samples for training (default 512), a quarter of it for testing.
"""
import os

import torch
from torch.utils.data import Dataset

# dataset -> (spectrogram [F, T'], n_classes); frames are [3, T, 224, 224] with T = args.fps or 3
# (CramedDataset.py:62-66 n_fft 512 hop 353 on 3 s @ 22.05 kHz -> 257 x 188; KSDataset.py:139-149 / VGGSoundDataset.py:117-122:
#  n_fft 256 hop 128 on 5 s @ 16 kHz -> 129 x 626; class counts: basic_model.py:15-26)
SHAPES = {"CREMAD": ((257, 188), 6), "KineticSound": ((129, 626), 34), "VGGSound": ((129, 626), 309), "AVE": ((129, 626), 28),
          "kinect400": ((129, 626), 400)}


class SyntheticAV(Dataset):
    dataset = "CREMAD"

    def __init__(self, args=None, mode="train"):
        self.args, self.mode = args, mode
        (self.spec_hw, self.n_classes) = SHAPES[self.dataset]
        self.frames = int(getattr(args, "fps", 3) or 3) if args is not None else 3
        n = int(os.environ.get("GDL_SYNTH_LEN", "512"))
        self.n = n if mode == "train" else max(1, n // 4)
        self.seed = 0 if mode == "train" else 1_000_003

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed + int(idx))
        spectrogram = torch.randn(*self.spec_hw, generator=g)
        images = torch.randn(3, self.frames, 224, 224, generator=g)
        label = int(torch.randint(0, self.n_classes, (1,), generator=g).item())
        return spectrogram, images, label
