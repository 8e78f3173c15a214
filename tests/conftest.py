import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference)")


def _have_gpu():
    """True when a HIP device is usable (device COUNT only: nothing is initialised on the GPU here)."""
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU skips the `gpu` tests instead of failing in them
    (the CPU suite is `-m "not gpu"`, the MI355X suite `-m gpu`)."""
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run on the MI355X box with -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
