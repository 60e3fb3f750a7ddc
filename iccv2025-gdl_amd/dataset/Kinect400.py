"""Import-resolving stand-in for /root/reference/dataset/Kinect400.py (synthetic tensors; see dataset/_synthetic.py)."""
from ._synthetic import SyntheticAV


class Kinect400(SyntheticAV):
    dataset = "kinect400"
