"""Import-resolving stand-in for /root/reference/dataset/KSDataset.py (synthetic tensors; see dataset/_synthetic.py)."""
from ._synthetic import SyntheticAV


class KSDataset(SyntheticAV):
    dataset = "KineticSound"
