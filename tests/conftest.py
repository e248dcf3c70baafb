import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture()
def cpu_backend():
    """Install the oracle-backed CPU test double for host-logic tests (never used by product code)."""
    from _cpu_backend import CpuTestBackend

    from deepquantum_amd import backend

    be = CpuTestBackend()
    backend.set_test_backend(be)
    yield be
    backend.set_test_backend(None)
