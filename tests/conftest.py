"""pytest configuration: `gpu` marker + shared helpers.

`-m "not gpu"` covers the oracle against golden vectors / the reference build, host logic and the
C-ABI symbol check; `-m gpu` tests are the HIP-vs-oracle parity tests (they call through the C ABI).
"""
import os
import sys

import numpy as np
import pytest

try:  # torch first: it brings its own copy of the HIP runtime, and whichever copy is loaded first owns the GPU -- a test
    import torch  # noqa: F401  that loads libufomap_hip.so (linked against /opt/rocm) before torch would leave torch without devices
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def same_dump(a, b):
    """True when two tuples of numpy arrays are element-wise identical (bit-exact)."""
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.fixture(scope="session")
def port_available():
    import oracle
    assert oracle.build("port"), "oracle/libufo_oracle.so could not be built"
    return True


@pytest.fixture(scope="session")
def ref_available():
    import oracle
    oracle.build("reference")
    return oracle.available("reference")
