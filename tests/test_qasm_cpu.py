"""OpenQASM in and out against the reference's own texts and circuits (tests/golden/golden_qasm.json, made by
make_golden_qasm.py), plus the semantics the reference gets wrong (``inv @``)."""

import json
import os

import pytest
import torch

import deepquantum_amd as dq
from _helpers import specs

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'golden_qasm.json')))


def _export_circuit(name):
    c = specs.QASM_EXPORT[name]
    cir = specs.build(dq, c['nqubit'], c['spec'])
    if c['measure']:
        cir.measure(wires=c['measure'])
    return cir


@pytest.mark.parametrize('name', list(specs.QASM_EXPORT))
def test_written_texts_equal_the_references(name):
    cir = _export_circuit(name)
    assert cir.qasm() == GOLD['export'][name]['qasm2']
    assert dq.cir_to_qasm3(cir) == GOLD['export'][name]['qasm3']


def test_qasm2_rejects_what_qelib1_cannot_express():
    cir = dq.QubitCircuit(3)
    cir.rxx([0, 1], 0.3, controls=[2])
    with pytest.raises(ValueError, match='Too many control bits'):
        cir.qasm()
    cir = dq.QubitCircuit(2)
    cir.rxy([0, 1], 0.3)
    with pytest.raises(ValueError, match='NOT supported'):
        cir.qasm()


@pytest.mark.parametrize('name', list(GOLD['import']))
def test_programs_read_like_the_reference_reads_them(cpu_backend, name):
    g = GOLD['import'][name]
    cir = dq.qasm3_to_cir(g['program'])
    assert [type(op).__name__ for op in cir.operators] == g['gates']
    assert cir.wires_measure == g['wires_measure']
    ref = torch.tensor(g['state'], dtype=torch.float64)
    got = cir().reshape(-1)
    err = (torch.view_as_real(got.to(torch.complex128)) - ref).abs().max().item()
    assert err < 1e-5, err


def test_inverse_modifier_inverts(cpu_backend):
    """Where this reader deliberately differs from the reference's (which ignores ``inv @``): a block followed by
    its inverse is the identity, for built-in gates, nested ``def`` blocks, controls and powers."""
    prog = '''OPENQASM 3.0;
qubit[3] q;
def inner(a) x { rx(a) x; s x; t x; }
def outer(a, b) x, y {
  inner(a) x;
  cx x, y;
  u(a, b, 0.3) y;
  inv @ inner(b) y;
}
h q[0]; ry(0.7) q[1]; h q[2];
outer(0.4, 1.1) q[0], q[1];
inv @ outer(0.4, 1.1) q[0], q[1];
ctrl @ outer(0.2, 0.5) q[2], q[0], q[1];
inv @ ctrl @ outer(0.2, 0.5) q[2], q[0], q[1];
pow(3) @ inner(0.9) q[2];
pow(-3) @ inner(0.9) q[2];
pow(0.3) @ x q[1];
inv @ pow(0.3) @ x q[1];
u(0.1, 0.2, 0.3) q[0];
inv @ u(0.1, 0.2, 0.3) q[0];
'''
    got = dq.qasm3_to_cir(prog)().reshape(-1)
    base = dq.qasm3_to_cir('OPENQASM 3.0;\nqubit[3] q;\nh q[0]; ry(0.7) q[1]; h q[2];')().reshape(-1)
    assert (got - base).abs().max().item() < 1e-5
    # and the QASM3 this library writes reads back to the same circuit, inverted U3 included
    cir = dq.QubitCircuit(2)
    cir.u3(0, [0.3, 0.9, -0.4])
    cir.add(cir.operators[0].inverse())
    cir.rx(1, 0.5, controls=[0])
    again = dq.qasm3_to_cir(dq.cir_to_qasm3(cir))
    assert (cir().reshape(-1) - again().reshape(-1)).abs().max().item() < 1e-6


def test_reader_rejects_other_input():
    with pytest.raises(ValueError, match='Header missing'):
        dq.qasm3_to_cir('OPENQASM 2.0;\nqreg q[2];')
    with pytest.raises(ValueError, match='Qubit declaration'):
        dq.qasm3_to_cir('OPENQASM 3.0;\nh q[0];')
