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

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            return self._c[name]
    return G()
