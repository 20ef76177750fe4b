import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the measured kernel choices of the graphs the tests build stay out of the user's cache (io.DenominatorGraph.prepare)
import tempfile  # noqa: E402

os.environ.setdefault("TORCHAIN_TUNING_CACHE", os.path.join(tempfile.gettempdir(), "torchain_tuning_tests_%d.json" % os.getpid()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only): builds oracle/libchain_oracle.so on demand."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture
def kernel_family():
    """Puts graphs built inside the test on a kernel family they would not normally take, through the
    library's diagnostic switches (include/torchain_hip.h: tc_debug_set); everything is reset afterwards."""
    from torchain_amd._lib import check, lib
    keys = []

    def force(key, value=1):
        check(lib.tc_debug_set(key.encode(), int(value)), "tc_debug_set(%s)" % key)
        keys.append(key)

    yield force
    for key in keys:
        lib.tc_debug_set(key.encode(), 0)
