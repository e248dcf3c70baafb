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


def pytest_report_header(config):
    """What the box shows: devices and the compute partition (an MI355X in CPX mode shows 8 devices: the sharded GPU
    tests then run over RCCL, one device per rank -- ``_helpers.pick_transport``)."""
    import shutil
    import subprocess

    lines = []
    try:
        import torch

        lines.append(f'devices visible to torch: {torch.cuda.device_count() if torch.cuda.is_available() else 0}')
    except Exception as e:  # noqa: BLE001
        lines.append(f'torch: {e}')
    smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if os.path.exists(smi) and os.path.exists('/dev/kfd'):
        try:
            out = subprocess.run([smi, '--showcomputepartition'], capture_output=True, text=True, timeout=20).stdout
            lines += ['rocm-smi --showcomputepartition: ' + ln.strip() for ln in out.splitlines() if 'artition' in ln][:9]
        except Exception as e:  # noqa: BLE001
            lines.append(f'rocm-smi: {e}')
    return lines
