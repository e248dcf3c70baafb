"""INTEGRATION.md option B, VERBATIM: the ``_dqhip.py`` stub a maintainer of the reference would add is cut out of the
document at test time, executed as a module against the in-tree ``libdqhip.so`` (``DQHIP_LIB``), and the 101 gate
cases of the real reference (``gate/{i}/in | matrix | out`` in tests/golden/golden.npz: every gate class x positions
x controls x inverse, with the exact matrices the reference used) go through its ``apply_gate`` -- the C ABI bound
with nothing of this package in between -- on the MI355X.  This is the boundary the reference would actually bind:
``evolve_state`` (qmath.py:485-506) and ``Gate.op_state_control`` (operation.py:203-219)."""

import os
import re
import types

import pytest
import torch

import deepquantum_amd as dq
from _helpers import gold, specs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_module():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    part = text[text.index('## B.'):]
    code = re.search(r'```python\n(# src/deepquantum/_dqhip\.py.*?)```', part, re.S).group(1)
    os.environ['DQHIP_LIB'] = os.path.join(ROOT, 'deepquantum_amd', 'libdqhip.so')
    mod = types.ModuleType('_dqhip_from_integration_md')
    exec(compile(code, 'INTEGRATION.md', 'exec'), mod.__dict__)
    return mod


def test_integration_stub_runs_the_reference_gate_cases():
    stub = _stub_module()
    dev = torch.device('cuda', 0)
    worst = 0.0
    for i, case in enumerate(specs.GATE_CASES):
        n = case['nqubit']
        gate = getattr(dq, case['cls'])(nqubit=n, **case['kwargs'])        # (for the wires / controls of the case only)
        psi = gold(f'gate/{i}/in').to(dev)                                    # (2**n, 1) complex128
        mat = gold(f'gate/{i}/matrix').to(dev)                                # the reference's own update_matrix()
        x = psi.reshape([1] + [2] * n)                                        # tensor_rep: what evolve_state is handed
        out = stub.apply_gate(x, mat, n, list(gate.wires), list(gate.controls))
        assert out.shape == x.shape
        err = (out.reshape(-1, 1).cpu() - gold(f'gate/{i}/out')).abs().max().item()
        worst = max(worst, err)
        assert err < 1e-10, (i, case['cls'], err)
        # complex64, batched, a per-sample matrix stack: the vmap case of circuit.py:232-240
        xb = torch.stack([psi.reshape(-1), psi.reshape(-1).flip(0)]).to(torch.complex64).reshape([2] + [2] * n)
        mb = torch.stack([mat, mat]).to(torch.complex64)
        outb = stub.apply_gate(xb, mb, n, list(gate.wires), list(gate.controls))
        assert (outb[0].reshape(-1, 1).to(torch.complex128).cpu() - gold(f'gate/{i}/out')).abs().max().item() < 1e-4
    print(f'INTEGRATION.md stub: {len(specs.GATE_CASES)} reference gate cases, max error {worst:.2e}')


def test_integration_stub_reports_errors():
    stub = _stub_module()
    x = torch.zeros([1] + [2] * 3, dtype=torch.complex64, device='cuda')
    x.reshape(-1)[0] = 1
    with pytest.raises(RuntimeError):
        stub.apply_gate(x, torch.eye(2, dtype=torch.complex64, device='cuda'), 3, [0], [0])      # target == control
