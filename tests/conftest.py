import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fgnn-artifacts_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _forced_scan_helping_path():
    """FGNN_SCAN_HELP_AFTER=<polls> (tests/test_coresidency_gpu.py re-runs the batch-driver parity tests with 0): the
    single-pass kernels' waits take the helping path for the whole test process.  Set here, by the tests -- the product
    (library and Python binding) reads no switch from the environment."""
    if os.environ.get("FGNN_SCAN_HELP_AFTER") is not None:
        from fgnn_hip import lib
        lib.load().fgnn_debug_set_scan_help_after(int(os.environ["FGNN_SCAN_HELP_AFTER"]))
    yield
