"""Import-resolving stand-in for /root/reference/dataset/AVEDataset.py (synthetic tensors; see dataset/_synthetic.py)."""
from ._synthetic import SyntheticAV


class AVEDataset(SyntheticAV):
    dataset = "AVE"
