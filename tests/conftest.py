import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "geometry_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
