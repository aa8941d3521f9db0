import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The reference API's default summation mode is "any order" (include/pandaseq_amd_compat.h); the C clients of this suite compare
# `quality` bit for bit with the oracle, so they run with sums in the reference's order -- except the test that checks the default.
os.environ.setdefault("PAMD_COMPAT_SUMS", "exact")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle are built (cheap when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True
