import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "iccv2025-gdl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _native_libraries():
    """The HIP library and the oracle's C restatement are build products (git-ignored): build them when a fresh
    checkout runs the suite before `__graft_entry__.build()` did (hipcc cross-compiles gfx950 without a GPU).  An
    existing library is left alone -- on the GPU box the pre-built one travels with the snapshot."""
    import subprocess

    so = os.path.join(PKG, "csrc", "build", "libgdl_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"])
    from oracle import oracle as orc

    orc.build()
