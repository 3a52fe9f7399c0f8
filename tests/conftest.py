import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # The CPU oracle runs inside this process in most GPU tests.  On the GPU box (256 hardware threads) torch's default thread
    # count makes it pathologically slow on the 30x160 / 45x240 maps of this path -- bench.py's probe of the same oracle: 16 threads
    # 0.46 s, 64 threads 1.19 s, 256 threads 75 s for three frames -- so big hosts are capped (VSR_TEST_THREADS overrides).
    try:
        import torch

        want = int(os.environ.get("VSR_TEST_THREADS", "0"))
        if want > 0:
            torch.set_num_threads(want)
        elif (os.cpu_count() or 1) > 32:
            torch.set_num_threads(16)
    except ImportError:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "baseline_oracle: compares against a BASELINE-size CPU oracle run (tests/_baseline_oracle.py)")


def pytest_collection_finish(session):
    """Start the BASELINE-size oracle runs in background processes as soon as it is known that a selected test needs them."""
    wanted = set()
    for item in session.items:
        if item.get_closest_marker("baseline_oracle") is None:
            continue
        cs = getattr(item, "callspec", None)
        name = cs.params.get("name") if cs is not None else None
        wanted.add(name or {"test_det_batch_L47_vs_oracle": "det_1080p", "test_propainter_batch_L20_vs_oracle": "pp_1080p",
                            "test_det_portrait_vs_oracle": "det_portrait"}.get(item.originalname))
    wanted.discard(None)
    from tests._baseline_oracle import FIXTURE_JOBS

    wanted -= set(FIXTURE_JOBS)                      # their oracle runs are committed fixtures (about an hour of CPU each)
    if wanted and not session.config.option.collectonly:
        import torch

        if torch.cuda.is_available():
            from tests import _baseline_oracle

            _baseline_oracle.launch(sorted(wanted))


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
    import time

    import torch

    # a fresh box has been seen to report no device to the first process that asks (profiles/r02_cli_e2e.log: pytest found none,
    # the command after it did): ask again for a while before giving up -- there is still no CPU path to fall back to
    for attempt in range(30):
        if torch.cuda.is_available() and built_lib.lib.vsr_device_count() > 0:
            return torch.device("cuda", 0)
        time.sleep(1.0)
    pytest.fail("this test is marked gpu but no HIP device is visible")
