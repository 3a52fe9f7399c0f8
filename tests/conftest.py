import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The HIP extension must exist (built in-tree); there is no fallback to test instead."""
    lib_path = os.path.join(ROOT, "video-subtitle-remover_amd", "lib", "libvsr_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__

        __graft_entry__.build()
    import vsr_amd  # noqa: F401
    from vsr_amd import _lib

    return _lib


@pytest.fixture(scope="session")
def gpu_device(built_lib):
    import torch

    if not torch.cuda.is_available() or built_lib.lib.vsr_device_count() <= 0:
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda", 0)
