import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """The C-ABI library is a build product (git-ignored): compile it once if the checkout does not have it yet."""
    lib = os.path.join(ROOT, "competesmoe_amd", "lib", "libcsmoe_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


def pytest_configure(config):
    _ensure_built()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests never run on a box without a GPU, whatever -m says."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
