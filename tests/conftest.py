import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: runs only with TECOGAN_SLOW=1 (sweeps, second variants of the multi-process tests)")
    config.addinivalue_line("markers", "experiments: a rejected variant; needs the experiments build of the library "
                                       "(csrc/build.sh --experiments, TECOGAN_LIB=.../libtecogan_hip_experiments.so) - skipped otherwise")


def pytest_collection_modifyitems(config, items):
    """tests of the rejected variants run only against the experiments library (never loaded by default); `slow` tests (sweeps and
    second variants of multi-process tests: minutes of box time, no parity row depends on them) only with TECOGAN_SLOW=1 - the
    default `-m gpu` run has to stay well inside the driver's 900-s budget (447 s and growing)"""
    if os.environ.get("TECOGAN_SLOW", "0") != "1":
        skip_slow = pytest.mark.skip(reason="slow: set TECOGAN_SLOW=1")
        for it in items:
            if it.get_closest_marker("slow"):
                it.add_marker(skip_slow)
    exp = [it for it in items if it.get_closest_marker("experiments")]
    if not exp:
        return
    lib = os.environ.get("TECOGAN_LIB", "")
    if "experiments" in os.path.basename(lib) and os.path.exists(lib):
        return
    skip = pytest.mark.skip(reason="needs the experiments build: csrc/build.sh --experiments and TECOGAN_LIB=<that library>")
    for it in exp:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
