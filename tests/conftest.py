import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fgnn-artifacts_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--kernel-lib", default=None,
                     help="tools/ab_variants.sh only: run the suite against another build of the kernel library "
                          "('prof' = lib/libfgnn_hip_prof.so); the product binding reads nothing from the environment")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    which = config.getoption("--kernel-lib")
    if which:
        from fgnn_hip import lib
        lib.use_library(lib.PROF_LIB_PATH if which == "prof" else which)


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


def pytest_terminal_summary(terminalreporter):
    """every skipped test with its reason, whatever -r flags the run was given: the log of the first multi-GPU box must
    show at a glance whether the two-device hand-off tests RAN (they skip on a one-GPU box and say how many devices
    they found)"""
    skipped = terminalreporter.stats.get("skipped", [])
    if not skipped:
        return
    terminalreporter.write_line("skipped tests and why (%d):" % len(skipped))
    for rep in skipped:
        reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) and len(rep.longrepr) == 3 else str(rep.longrepr)
        terminalreporter.write_line("  %s -- %s" % (rep.nodeid, reason))
