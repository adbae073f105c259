import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_built():
    """The built libraries normally travel with the tree (they are git-ignored, not gpurun-ignored); a fresh checkout has
    none, so build what is missing (gcc + hipcc, the same recipe as __graft_entry__.build())."""
    import subprocess
    wanted = {
        os.path.join(ROOT, "cpecan_amd", "csrc"): [os.path.join(ROOT, "cpecan_amd", "libcpecan_hip.so"),
                                                   os.path.join(ROOT, "cpecan_amd", "cpecan_realign")],
        os.path.join(ROOT, "oracle"): [os.path.join(ROOT, "oracle", "liboracle.so")],
    }
    for makedir, products in wanted.items():
        if not all(os.path.exists(f) for f in products):
            subprocess.check_call(["make", "-C", makedir], stdout=subprocess.DEVNULL)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


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
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
