import os
import sys
import pathlib

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a device must fail loudly, not skip silently
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    markexpr = config.getoption("-m") or ""
    for item in items:
        if "gpu" in item.keywords and "gpu" not in markexpr.replace("not gpu", ""):
            item.add_marker(skip)
