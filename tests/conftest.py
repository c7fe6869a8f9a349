import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


HOOKS_LIB = ROOT / "cachedembedding_amd" / "libce_hip_testhooks.so"


def hooks_build_loaded() -> bool:
    """is this process bound to the -DCE_TEST_HOOKS twin of the library (CE_LIBRARY, set by the wrapper test)?"""
    return os.environ.get("CE_LIBRARY", "").endswith(HOOKS_LIB.name)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "test_hooks: needs the fault / delay injection of libce_hip_testhooks.so; skipped "
                            "in a process on the product library and run by "
                            "tests/test_gpu_worker.py::test_hook_tests_on_the_test_hooks_build in a child process")


def pytest_collection_modifyitems(config, items):
    if hooks_build_loaded():
        return
    skip = pytest.mark.skip(reason="product library has no test hooks: run in a child process on libce_hip_testhooks.so "
                                   "by test_hook_tests_on_the_test_hooks_build")
    for item in items:
        if item.get_closest_marker("test_hooks"):
            item.add_marker(skip)


@pytest.fixture(scope="session")
def have_gpu():
    import torch
    return torch.cuda.is_available()
