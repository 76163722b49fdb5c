import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_present():
    """The .so files are git-ignored build artefacts: build them in-tree if a fresh checkout lacks them
    (hipcc cross-compiles without a GPU; the oracle library is plain gcc)."""
    import subprocess
    lib = os.path.join(ROOT, "steered_mixture_of_experts_amd", "libsmoe_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "steered_mixture_of_experts_amd", "csrc"), "-j4"])
    ora = os.path.join(ROOT, "oracle", "libsmoe_oracle.so")
    if not os.path.exists(ora):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield
