import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (checker).  Built on demand with plain gcc."""
    so = os.path.join(REPO, "oracle", "libf16_oracle.so")
    # always through make (incremental): a stale checker binary would silently compare the kernels with old code
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "libf16_oracle.so"])
    from oracle import mpc_oracle
    return mpc_oracle.COracle(so)


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly (no fallback) when it is missing."""
    from f16_mpc_oop_py_amd import lib
    return lib.load()
