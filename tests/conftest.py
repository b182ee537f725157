import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# ZT_TEST_LIB=<path>: run the suite against another build of the library (tools/build_variant.sh: the chain modes and the
# cooperative prepass that were measured slower live in variant builds only) -- test infrastructure, not a product switch
if os.environ.get("ZT_TEST_LIB"):
    from zebra_amd import _capi as _capi_for_tests
    _p = os.environ["ZT_TEST_LIB"]
    _capi_for_tests.LIB_PATH = _p if os.path.isabs(_p) else os.path.join(ROOT, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/zebra_oracle.h)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.lib()
    return pyoracle
