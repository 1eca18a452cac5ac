import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def golden_gauss():
    """SparseGaussianRegression / SparseGaussianGLM vectors (tests/golden/make_fixtures.py::main_gaussian)"""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors_gaussian.npz"))


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """The oracle's C restatement is test infrastructure; build it on demand (gcc only)."""
    import subprocess
    # (always through make: a library older than pg_oracle.c -- one that travelled with a snapshot -- is rebuilt, an up-to-date one is left alone)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
