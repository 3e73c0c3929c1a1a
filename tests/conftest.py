import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    try:
        import torch
        if not torch.cuda.is_available():
            return "needs an MI355X (torch.cuda.is_available() is False)"
    except Exception as e:                        # pragma: no cover
        return f"torch not importable: {e}"
    lib = os.path.join(ROOT, "audio-metrics_amd", "lib", "libaudio_metrics_hip.so")
    if not os.path.exists(lib):
        return f"{lib} has not been built"
    return None


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: tests marked gpu are skipped instead of failing / hanging."""
    if not any(item.get_closest_marker("gpu") for item in items):
        return
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if item.get_closest_marker("gpu"):
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    base = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(base, name + ".npz"), allow_pickle=False)
    return load
