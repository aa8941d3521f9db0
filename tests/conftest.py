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


def _libraries_are_fresh():
    """The built libraries travel to the GPU box, their object files do not: `make` there would compile every kernel again (a
    minute of the suite's time) to arrive at the same bytes.  True when each library is newer than everything it is made from."""
    import glob
    def newest(patterns):
        files = [f for p in patterns for f in glob.glob(os.path.join(ROOT, p))]
        return max(os.path.getmtime(f) for f in files) if files else float("inf")
    def built_after(lib, patterns):
        path = os.path.join(ROOT, lib)
        return os.path.exists(path) and os.path.getmtime(path) >= newest(patterns)
    product = ["pandaseq_amd/csrc/*", "include/*.h", "tools/isa_hazards.py"]
    return (built_after("pandaseq_amd/libpandaseq_amd.so", product) and built_after("pandaseq_amd/libpandaseq_amd_reflink.so", product)
            and built_after("oracle/liboracle.so", ["oracle/*.c", "oracle/*.h", "oracle/Makefile"]))


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle are built (cheap when up to date)."""
    import __graft_entry__ as g
    if _libraries_are_fresh() and not os.path.isdir("/root/reference"):     # (the GPU box: nothing to compile, only to load and check)
        import pandaseq_amd  # noqa: F401
        from pandaseq_amd import _lib, textio
        for sym in _lib.SYMBOLS + textio.SYMBOLS:
            getattr(_lib.lib(), sym)
        return True
    g.build()
    return True
