import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    # no more threads than CPUs this process may run on (an OpenMP runtime that sizes its team by the machine, on a box that
    # grants 16 of 256 hardware threads, turns every small parallel region into milliseconds)
    orc.set_num_threads(max(1, min(orc.num_threads(), len(os.sched_getaffinity(0)), 32)))
    return orc
