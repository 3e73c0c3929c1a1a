import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:                             # pragma: no cover
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: tests marked gpu are skipped instead of failing / hanging.  On a box WITH a
    GPU nothing is skipped: a missing or stale HIP library must show up as a failure of the parity suite (the fixtures
    call _lib.load(), which raises), never as a green run of skipped tests."""
    gpu_items = [item for item in items if item.get_closest_marker("gpu")]
    if not gpu_items:
        return
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        return
    if _gpu_visible():
        lib = os.path.join(ROOT, "audio-metrics_amd", "lib", "libaudio_metrics_hip.so")
        if not os.path.exists(lib):
            raise pytest.UsageError(f"an MI355X is visible but {lib} has not been built: run `python __graft_entry__.py build` "
                                    "(the GPU parity tests are not skipped for a missing library)")
        import importlib.util
        spec = importlib.util.spec_from_file_location("am_build", os.path.join(ROOT, "audio-metrics_amd", "_build.py"))
        build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(build)
        if build.is_stale(build.LIB_PATH):
            raise pytest.UsageError(f"{lib} was built from other sources than this tree holds (content hash in its .stamp.json): "
                                    "run `python __graft_entry__.py build` - parity results of a stale library mean nothing")
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for item in gpu_items:
        item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    base = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(base, name + ".npz"), allow_pickle=False)
    return load
