"""Import-resolving stand-in for /root/reference/dataset/VGGSoundDataset.py (synthetic tensors; see dataset/_synthetic.py)."""
from ._synthetic import SyntheticAV


class VGGSound(SyntheticAV):
    dataset = "VGGSound"
