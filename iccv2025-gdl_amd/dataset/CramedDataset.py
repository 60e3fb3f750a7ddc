"""Import-resolving stand-in for /root/reference/dataset/CramedDataset.py (synthetic tensors; see dataset/_synthetic.py)."""
from ._synthetic import SyntheticAV


class CramedDataset(SyntheticAV):
    dataset = "CREMAD"


class CramedDataset_swin(CramedDataset):
    """main_dgl.py:12 imports this name too (the Swin variant of the reference's loader)."""
