import os
import sys

import pytest

try:
    # torch ships its own HIP runtime; it has to be the first one loaded into the process (as in bench.py), or torch
    # finds "no HIP GPUs" once libhtmjoin_hip.so has pulled in the system one. Only the in-process multi-rank test
    # (test_gpu_parity.py) uses torch on the GPU; everything else goes through the C ABI alone.
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """The .so files and `main` are build products (git-ignored). Build them once if a checkout lacks them."""
    need = [os.path.join(ROOT, "htm-hashjoin_amd", "lib", "libhtmjoin_hip.so"),
            os.path.join(ROOT, "htm-hashjoin_amd", "bin", "main"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def _gpu_available():
    try:
        import htm_hashjoin_amd as hj
        return hj.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A -m gpu run on a box without a GPU must fail loudly, not skip silently;
    # an unmarked full run on the CPU box skips the gpu tests.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if not _gpu_available():
        skip = pytest.mark.skip(reason="no GPU in this container")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
