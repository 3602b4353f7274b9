import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    import torch
    torch.set_num_threads(_host_cores())        # GPU boxes expose 256 threads under a 16-CPU cgroup quota
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
