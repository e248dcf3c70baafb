"""Circuit / gate specifications shared by the golden-vector generator (which drives the real reference)
and by the tests (which drive deepquantum_amd).  A spec is a list of (builder method, args, kwargs)
applied to a ``QubitCircuit``: both libraries expose the same builder API, which is the point."""

import math
import random

import torch


def random_spec(n, depth, seed):
    """SURVEY.md section 8(d) generator: per layer, per qubit: 1/3 H, 1/3 Rx(U(0, 2pi)), 1/3 CNOT."""
    rng = random.Random(seed)
    spec = []
    for _ in range(depth):
        for q in range(n):
            r = rng.random()
            if r < 1 / 3:
                spec.append(('h', [q], {}))
            elif r < 2 / 3:
                spec.append(('rx', [q, rng.uniform(0, 2 * math.pi)], {}))
            else:
                t = rng.randrange(n - 1)
                t += t >= q
                spec.append(('cnot', [q, t], {}))
    return spec


def build(dq, n, spec, **circuit_kwargs):
    cir = dq.QubitCircuit(n, **circuit_kwargs)
    for method, args, kwargs in spec:
        getattr(cir, method)(*args, **kwargs)
    return cir


def _unitary(k, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(2**k, 2**k, generator=g, dtype=torch.float64) + 1j * torch.randn(2**k, 2**k, generator=g, dtype=torch.float64)
    q, _ = torch.linalg.qr(a)
    return q.to(torch.complex64)


ZOO5 = [
    ('hlayer', [], {}),
    ('u3', [0, [0.3, 1.1, -0.4]], {}),
    ('p', [1, 0.7], {}),
    ('x', [2], {}), ('y', [3], {}), ('z', [4], {}),
    ('s', [0], {}), ('sdg', [1], {}), ('t', [2], {}), ('tdg', [3], {}),
    ('ry', [4, 0.9], {}), ('rz', [0, -1.3], {}), ('rx', [1, 2.2], {}),
    ('cx', [0, 3], {}), ('cy', [4, 1], {}), ('cz', [2, 0], {}), ('cnot', [3, 4], {}),
    ('swap', [[0, 4]], {}), ('iswap', [[1, 3]], {}),
    ('rxx', [[0, 2], 0.6], {}), ('ryy', [[3, 1], -0.8], {}), ('rzz', [[4, 2], 1.7], {}),
    ('rxy', [[2, 3], 0.45], {}), ('rbs', [[0, 1], 0.35], {}),
    ('toffoli', [0, 1, 2], {}), ('ccx', [4, 2, 3], {}), ('fredkin', [2, 0, 4], {}), ('cswap', [1, 3, 0], {}),
    ('crx', [0, 2, 0.5], {}), ('cry', [3, 1, 1.5], {}), ('crz', [4, 0, -0.6], {}),
    ('ch', [1, 4], {}), ('cs', [2, 3], {}), ('ct', [0, 1], {}), ('cp', [3, 0, 0.9], {}), ('cu', [2, 4, [0.2, 0.4, 0.6]], {}),
    ('crxx', [0, 1, 2, 0.3], {}), ('cryy', [4, 3, 2, 0.7], {}), ('crzz', [1, 0, 4, 1.1], {}), ('crxy', [2, 4, 3, 0.25], {}),
    ('rx', [2, 0.8], {'controls': [0, 4]}), ('h', [3], {'controls': [1, 2]}), ('p', [4, 0.3], {'controls': [0, 1, 2]}),
    ('swap', [[1, 2]], {'controls': [3, 4]}),
    ('rxlayer', [], {'inputs': [0.1, 0.2, 0.3, 0.4, 0.5]}),
    ('rylayer', [[0, 2, 4]], {'inputs': [1.0, 1.1, 1.2]}),
    ('rzlayer', [[1, 3]], {'inputs': [-0.5, 0.5]}),
    ('u3layer', [[0, 1]], {'inputs': [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]}),
    ('cnot_ring', [], {}), ('cnot_ring', [], {'minmax': [1, 4], 'step': 2, 'reverse': True}),
    ('xlayer', [[0, 3]], {}), ('ylayer', [[1]], {}), ('zlayer', [[2, 4]], {}),
    ('any', [_unitary(1, 1)], {'wires': [3]}),
    ('any', [_unitary(2, 2)], {'wires': [4, 1]}),
    ('any', [_unitary(3, 3)], {'wires': [2, 0, 3]}),
    ('cxlayer', [[[0, 1], [2, 3]]], {}),
]

BATCHED10 = (
    [('hlayer', [], {})]
    + [('rx', [q], {'encode': True}) for q in range(0, 10, 2)]
    + [('cnot', [q, (q + 3) % 10], {}) for q in range(10)]
    + [('ry', [q], {'encode': True}) for q in range(1, 10, 3)]
    + [('u3', [4], {'encode': True}), ('rzz', [[2, 7]], {'encode': True}), ('rz', [9], {'encode': True})]
    + [('cz', [0, 9], {}), ('toffoli', [1, 5, 8], {}), ('rx', [6, 0.77], {})]
    + [('crx', [3, 6], {'encode': True}), ('p', [0], {'encode': True})]
)
_NDATA10 = 5 + 3 + 3 + 1 + 1 + 1 + 1
_g = torch.Generator().manual_seed(21)
_DATA10 = (torch.rand(4, _NDATA10, generator=_g, dtype=torch.float64) * 6).tolist()

CIRCUITS = {
    'readme': {
        'nqubit': 2,
        'spec': [('h', [0], {}), ('cnot', [0, 1], {}), ('rx', [1, 0.2], {})],
        'observables': [([0], 'z')],
    },
    'zoo5': {'nqubit': 5, 'spec': ZOO5, 'observables': [([0], 'z'), ([1, 2], 'xy'), ([4, 3, 0], 'zyx')],
             'marginal': [0, 2], 'unitary': True},
    'batched10': {'nqubit': 10, 'spec': BATCHED10, 'data': _DATA10,
                  'observables': [([0], 'z'), ([3, 8], 'xx'), ([5], 'y')], 'marginal': [1, 4, 9]},
}
for _n in (4, 8, 12, 16):
    CIRCUITS[f'rand{_n}'] = {
        'nqubit': _n, 'spec': random_spec(_n, 20, 100 + _n),
        'observables': [([0], 'z'), ([1, 2], 'xy')], 'marginal': [0, 2], 'unitary': _n <= 4,
    }
CIRCUITS['rand14_seed1234'] = {'nqubit': 14, 'spec': random_spec(14, 40, 1234), 'observables': [([0], 'z')]}

# ---- F2: gate classes ------------------------------------------------------------------------------------
GATE_CASES = []
for cls in ('PauliX', 'PauliY', 'PauliZ', 'Hadamard', 'SGate', 'SDaggerGate', 'TGate', 'TDaggerGate'):
    for w in range(3):
        GATE_CASES.append({'cls': cls, 'nqubit': 3, 'kwargs': {'wires': [w]}})
    GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': [2], 'controls': [0]}})
    GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': [0], 'controls': [3, 1]}, 'inverse': True})
for cls, val in (('Rx', 0.37), ('Ry', -1.2), ('Rz', 2.5), ('PhaseShift', 0.81)):
    for w in range(3):
        GATE_CASES.append({'cls': cls, 'nqubit': 3, 'kwargs': {'wires': [w], 'inputs': val}})
    GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': [1], 'controls': [3], 'inputs': val}})
    GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': [3], 'controls': [0, 2], 'inputs': val}, 'inverse': True})
GATE_CASES.append({'cls': 'U3Gate', 'nqubit': 3, 'kwargs': {'wires': [1], 'inputs': [0.4, 1.3, -2.1]}})
GATE_CASES.append({'cls': 'U3Gate', 'nqubit': 3, 'kwargs': {'wires': [2], 'controls': [0], 'inputs': [1.4, 0.3, 0.1]}, 'inverse': True})
for plane in ('xy', 'yz', 'zx'):
    GATE_CASES.append({'cls': 'ProjectionJ', 'nqubit': 2, 'kwargs': {'wires': [1], 'inputs': 0.6, 'plane': plane}})
for cls in ('CNOT', 'Swap', 'ImaginarySwap'):
    for wires in ([0, 1], [2, 0], [1, 3]):
        GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': wires}})
GATE_CASES.append({'cls': 'Swap', 'nqubit': 4, 'kwargs': {'wires': [3, 0], 'controls': [1]}})
for cls, val in (('Rxx', 0.9), ('Ryy', -0.4), ('Rzz', 1.6), ('Rxy', 0.7), ('ReconfigurableBeamSplitter', 0.33)):
    for wires in ([0, 1], [3, 1]):
        GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': wires, 'inputs': val}})
    GATE_CASES.append({'cls': cls, 'nqubit': 4, 'kwargs': {'wires': [2, 0], 'controls': [3], 'inputs': val}, 'inverse': True})
for wires in ([0, 1, 2], [3, 0, 2], [2, 4, 1]):
    GATE_CASES.append({'cls': 'Toffoli', 'nqubit': 5, 'kwargs': {'wires': wires}})
    GATE_CASES.append({'cls': 'Fredkin', 'nqubit': 5, 'kwargs': {'wires': wires}})
GATE_CASES.append({'cls': 'UAnyGate', 'nqubit': 4, 'kwargs': {'unitary': _unitary(2, 9), 'wires': [3, 1]}})
GATE_CASES.append({'cls': 'UAnyGate', 'nqubit': 5, 'kwargs': {'unitary': _unitary(3, 10), 'minmax': [1, 3]}, 'inverse': True})
GATE_CASES.append({'cls': 'UAnyGate', 'nqubit': 6, 'kwargs': {'unitary': _unitary(5, 11), 'wires': [5, 0, 3, 1, 4]}})
GATE_CASES.append({'cls': 'UAnyGate', 'nqubit': 4, 'kwargs': {'unitary': _unitary(1, 12), 'wires': [2], 'controls': [0, 3]}})
# NB: a controlled UAnyGate is checked through forward only -- the reference's ArbitraryGate.get_unitary
# (gate.py:318-330) ignores `controls`, so its get_unitary disagrees with its own forward for such gates.
GATE_CASES.append({'cls': 'UAnyGate', 'nqubit': 5, 'kwargs': {'unitary': _unitary(2, 4), 'minmax': [1, 2], 'controls': [4]}})


# ---- F5: circuits with data inputs and trainable parameters ----------------------------------------------
def grad_circuit_a(dq, n):
    """Encoders (data) + trainable gates + controlled parametric gates + 2-qubit parametric gates."""
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.ry(1, encode=True)
    cir.rz(2, encode=True)
    cir.cnot_ring()
    cir.rxlayer()                      # trainable
    cir.crx(0, 3, encode=True)
    cir.rzz([1, 2], encode=True)
    cir.u3(3)                          # trainable
    cir.ryy([0, 2])                    # trainable
    cir.toffoli(0, 1, 3)
    cir.ry(2, controls=[0, 1])         # trainable, two controls
    cir.observable(0)
    cir.observable([1, 2], 'xy')
    cir.observable([3], 'x')
    return cir


def qaoa_circuit(dq, n):
    """Ring MaxCut QAOA layer in the style of the reference's examples/qaoa.py: data = (gamma, beta)."""
    cir = dq.QubitCircuit(n, reupload=True)
    cir.hlayer()
    edges = [(i, (i + 1) % n) for i in range(n)]
    for i, j in edges:
        cir.cnot(i, j)
        cir.rz(j, encode=True)
        cir.cnot(i, j)
    cir.rxlayer(encode=True)
    for i, j in edges:
        cir.observable([i, j], 'zz')
    return cir


# ---- F7: sharded semantics (the reference's own tests/test_circuit.py:45-139 circuit + scaled configs 4/5) -
def dist_test_circuit(dq, cls, n=4, observables=False):
    """The circuit of the reference's test_qubit_dist / test_qubit_expectation_and_differentiation_dist
    (tests/test_circuit.py:45-139), rebuilt through the public builder API: 4 qubits, re-uploaded data.
    With n > 4 the layers span all wires and the explicit gates stay on wires 0..3 (the global ones): the
    reference treats a controlled multi-qubit gate as a dense gate on controls + targets and needs them all
    to fit the local qubits (distributed.py:189), so n = 4 only runs un-sharded there."""
    cir = cls(n, reupload=True)
    cir.rxlayer(encode=True)
    cir.rylayer(encode=True)
    cir.rzlayer(encode=True)
    cir.u3layer(encode=True)
    cir.hlayer()
    cir.cnot_ring()
    cir.toffoli(0, 1, 2)
    cir.fredkin(2, 1, 0)
    cir.swap([2, 3])
    cir.rx(0, controls=[1, 2, 3], encode=True)
    cir.ry(1, controls=[0, 2, 3], encode=True)
    cir.rz(2, controls=[0, 1, 3], encode=True)
    cir.rxx([0, 1], controls=[2, 3], encode=True)
    cir.ryy([1, 2], controls=[0, 3], encode=True)
    cir.rzz([2, 3], controls=[0, 1], encode=True)
    cir.rxy([3, 0], controls=[1, 2], encode=True)
    if observables:
        cir.observable(0)
        cir.observable(1, 'x')
        cir.observable([2, 3], 'xy')
    return cir


def config4_circuit(dq, cls, n, depth=6, seed=1234):
    """BASELINE config 4 scaled down: the generator circuit plus a CNOT with a global control and one
    with a global target."""
    cir = cls(n)
    for method, args, kwargs in random_spec(n, depth, seed):
        getattr(cir, method)(*args, **kwargs)
    cir.cx(0, n - 1)
    cir.cx(n - 1, 0)
    cir.observable(0)
    return cir


def config5_circuit(dq, cls, n, depth=3, seed=1234):
    """BASELINE config 5 scaled down: generator circuit + one QAOA ring step with (gamma, beta) as data,
    one ZZ observable per ring edge (after examples/qaoa.py:31-44)."""
    cir = cls(n, reupload=True)
    for method, args, kwargs in random_spec(n, depth, seed):
        getattr(cir, method)(*args, **kwargs)
    cir.hlayer()
    edges = [(i, (i + 1) % n) for i in range(n)]
    for i, j in edges:
        cir.cnot(i, j)
        cir.rz(j, encode=True)
        cir.cnot(i, j)
    cir.rxlayer(encode=True)
    for i, j in edges:
        cir.observable([i, j], 'zz')
    return cir


DIST_CASES = {
    # name: (builder, kwargs, nqubit, data, world sizes)
    'dist4': dict(builder='dist_test_circuit', kwargs={'observables': True}, nqubit=4,
                  data=[0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0], worlds=(1,)),
    'dist7': dict(builder='dist_test_circuit', kwargs={'n': 7, 'observables': True}, nqubit=7,
                  data=[0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0], worlds=(1, 2, 4)),
    'config4_n8': dict(builder='config4_circuit', kwargs={'n': 8}, nqubit=8, data=None, worlds=(1, 4)),
    'config5_n9': dict(builder='config5_circuit', kwargs={'n': 9}, nqubit=9, data=[0.37, 1.21], worlds=(1, 8)),
}


# ---- Reset (gate.py:3027-3094): deterministic cases ------------------------------------------------------
_RESET_PREFIX = [('hlayer', [], {}), ('rx', [0, 0.7], {}), ('ry', [2, 1.3], {}), ('cnot', [0, 1], {}),
                 ('cnot', [2, 3], {}), ('rzz', [[1, 4], 0.9], {}), ('u3', [3, [0.4, 0.5, 0.6]], {})]
RESET_CASES = {
    'reset_ps0': _RESET_PREFIX + [('reset', [[1, 3]], {'postselect': 0}), ('h', [1], {}), ('cnot', [1, 2], {})],
    'reset_ps1': _RESET_PREFIX + [('reset', [2], {'postselect': 1}), ('ry', [2, 0.3], {})],
    # the postselected outcome has probability exactly 0 -> the other branch is taken (gate.py:3061-3063)
    'reset_ps1_empty': [('h', [0], {}), ('cnot', [0, 1], {}), ('reset', [2], {'postselect': 1}), ('h', [2], {})],
    'reset_all': _RESET_PREFIX + [('reset', [], {}), ('h', [0], {})],
    # sampled reset whose outcome is certain: wires 1, 3 are in |1>, |0>
    'reset_sampled_certain': [('h', [0], {}), ('x', [1], {}), ('cnot', [0, 2], {}),
                              ('reset', [[3, 1]], {'postselect': None}), ('h', [1], {})],
}

# Reset applied to given input states (pins the oracle's restatement directly): (wires, postselect, kind of input)
RESET_GATE_CASES = [([1], 0, 'random'), ([0, 3], 1, 'random'), ([2], 0, 'bit_set'), ([2], 1, 'bit_clear'),
                    ([0, 1, 2, 3], 0, 'random'), ([3, 1], 1, 'random')]


# ---- ansatz library (reference ansatz.py): name -> builder(dq) -> circuit ---------------
def _enc_then(dq, nq, parts):
    cir = parts[0]
    for p in parts[1:]:
        cir = cir + p
    return cir


def _hhl(dq):
    return dq.HHL(3, [[1.0, -1 / 3], [-1 / 3, 1.0]], t0=0.75)


def _qpe(dq):
    return dq.ansatz.QuantumPhaseEstimation(5, 3, _unitary(2, 21))


def _g3(dq):
    random.seed(7)
    return dq.RandomCircuitG3(6, 40)


def _cua(dq):
    nreg = 4
    nq = 2 * nreg + 2
    return dq.NumberEncoder(nq, 3, [0, nreg - 1]) + dq.ControlledUa(nq, 8, 15, [0, nreg - 1], list(range(nreg, 2 * nreg + 2)))


def _cmult(dq):
    nx, nb = 4, 5
    nq = nx + nb + 1
    return (dq.NumberEncoder(nq, 14, [0, nx - 1]) + dq.NumberEncoder(nq, 1, [nx, nq - 2])
            + dq.ControlledMultiplier(nq, 2, 15, [0, nq - 2], nx, [nq - 1]))


def _pma(dq):
    nq = 6
    mm = [0, nq - 2]
    qft = dq.QuantumFourierTransform(nq, mm, reverse=True)
    return dq.NumberEncoder(nq, 5, mm) + qft + dq.PhiModularAdder(nq, 1, 8, mm, [nq - 1]) + qft.inverse()


ANSATZ_CASES = {
    'qft5': lambda dq: dq.NumberEncoder(5, 11) + dq.QuantumFourierTransform(5),
    'qft5_reverse_barrier': lambda dq: dq.NumberEncoder(5, 19) + dq.QuantumFourierTransform(5, reverse=True, show_barrier=True),
    'qft_sub': lambda dq: dq.NumberEncoder(6, 5, [1, 4]) + dq.QuantumFourierTransform(6, [1, 4]),
    'iqft5': lambda dq: dq.NumberEncoder(5, 6) + dq.QuantumFourierTransform(5).inverse(),
    'qpe_single_exact': lambda dq: dq.QuantumPhaseEstimationSingleQubit(3, 1 / 8),
    'qpe_single_inexact': lambda dq: dq.QuantumPhaseEstimationSingleQubit(4, 0.3),
    'phi_adder': lambda dq: (dq.NumberEncoder(5, 1) + dq.QuantumFourierTransform(5, reverse=True) + dq.PhiAdder(5, 8)
                             + dq.QuantumFourierTransform(5, reverse=True).inverse()),
    'phi_adder_ctrl': lambda dq: (dq.NumberEncoder(6, 35) + dq.QuantumFourierTransform(6, [1, 5], reverse=True)
                                  + dq.PhiAdder(6, 13, [1, 5], controls=[0])),
    'phi_mod_adder': _pma,
    'cmult': _cmult,
    'cua': _cua,
    'shor15_special_a7': lambda dq: dq.ShorCircuitFor15(4, 7),
    'shor15_special_a11': lambda dq: dq.ShorCircuitFor15(3, 11),
    'shor15_general': lambda dq: dq.ShorCircuit(15, 3, 7),
    'hhl': _hhl,
    'qpe': _qpe,
    'g3': _g3,
}


# ---- density matrices + channels -------------------------------------------------------
DM_ZOO4 = [
    ('hlayer', [], {}), ('rx', [0, 0.7], {}), ('bit_flip', [0, 0.3], {}), ('cnot', [0, 1], {}),
    ('phase_flip', [1, 0.5], {}), ('ry', [2, 1.1], {}), ('depolarizing', [2, 0.4], {}), ('toffoli', [0, 1, 3], {}),
    ('pauli', [3, [0.3, 0.9, 0.5, 1.2]], {}), ('rzz', [[1, 3], 0.8], {}), ('amp_damp', [1, 0.6], {}),
    ('u3', [2, [0.2, 0.4, 0.6]], {}), ('phase_damp', [2, 0.45], {}), ('swap', [[0, 3]], {}),
    ('gen_amp_damp', [0, [0.8, 0.35]], {}), ('crx', [3, 1, 0.9], {}), ('cz', [2, 0], {}), ('s', [1], {}), ('t', [3], {}),
    ('rx', [2, 0.33], {'controls': [0, 1]}), ('fredkin', [1, 0, 2], {}), ('y', [0], {}), ('p', [1, 0.7], {}),
]
DM_CASES = {
    'dm_zoo4': dict(nqubit=4, spec=DM_ZOO4, init='zeros', observables=[([0], 'z'), ([1, 2], 'xy'), ([3], 'y'), ([0, 3], 'zx')]),
    'dm_ghz3': dict(nqubit=3, spec=[('ry', [0, 0.4], {}), ('amp_damp', [2, 0.5], {}), ('cnot', [2, 0], {})], init='ghz',
                    observables=[([0, 1, 2], 'xxx'), ([1], 'z')]),
    'dm_equal2': dict(nqubit=2, spec=[('depolarizing', [0, 0.7], {}), ('h', [1], {})], init='equal',
                      observables=[([0], 'x'), ([1], 'z')]),
}


# ---- OpenQASM --------------------------------------------------------------------------
QASM_EXPORT = {
    'zoo': dict(nqubit=5, measure=[0, 3], spec=[
        ('h', [0], {}), ('x', [1], {}), ('y', [2], {}), ('z', [3], {}), ('s', [4], {}), ('sdg', [0], {}), ('t', [1], {}),
        ('tdg', [2], {}), ('rx', [3, 0.3], {}), ('ry', [4, -1.2], {}), ('rz', [0, 2.5], {}), ('p', [1, 0.7], {}),
        ('u3', [2, [0.1, 0.2, 0.3]], {}), ('cnot', [0, 1], {}), ('cx', [2, 3], {}), ('cy', [4, 0], {}), ('cz', [1, 2], {}),
        ('ch', [3, 4], {}), ('cs', [0, 2], {}), ('cs', [1, 3], {}), ('crx', [0, 1, 0.4], {}), ('cry', [2, 4, 0.5], {}),
        ('crz', [3, 0, 0.6], {}), ('cp', [4, 1, 0.8], {}), ('cu', [0, 3, [0.4, 0.5, 0.6]], {}), ('swap', [[1, 4]], {}),
        ('swap', [[0, 2]], {'controls': [3]}), ('rxx', [[0, 1], 0.9], {}), ('ryy', [[2, 3], 1.0], {}),
        ('ryy', [[1, 4], 1.1], {}), ('rzz', [[0, 4], 1.2], {}), ('toffoli', [0, 1, 2], {}), ('fredkin', [3, 4, 0], {}),
        ('x', [4], {'controls': [0, 1]}), ('x', [3], {'controls': [0, 1, 2]}), ('x', [2], {'controls': [0, 1, 3, 4]}),
        ('sdg', [1], {'controls': [2]}), ('barrier', [[0, 1, 2]], {}), ('hlayer', [], {}), ('rxlayer', [[0, 2]], {'inputs': [0.1, 0.2]}),
    ]),
    'no_measure': dict(nqubit=2, measure=[], spec=[('h', [0], {}), ('cnot', [0, 1], {}), ('rz', [1, 0.25], {})]),
}
QASM3_PROGRAMS = {
    'defs_ctrl_pow': '''OPENQASM 3.0;
include "stdgates.inc";
qubit[4] q;
bit[2] c;
def bell a, b {
  h a;
  cx a, b;
}
def rot(theta, phi) a {
  rx(theta) a;
  rz(phi) a;
}
h q[0];
bell q[0], q[1];
rot(pi/3, 0.25) q[2];
rot(2*pi/5 - 0.1, -0.5) q[3];
ctrl @ rot(0.4, 0.9) q[0], q[2];
ctrl @ bell q[3], q[1], q[2];
ctrl @ ctrl @ x q[0], q[1], q[3];
pow(2) @ t q[1];
pow(3) @ rot(0.2, 0.1) q[0];
pow(0.5) @ x q[2];
ctrl @ pow(0.25) @ z q[1], q[3];
pow(-1) @ s q[3];
pow(-2) @ rx(0.3) q[0];
cz q[0], q[3];
ccx q[0], q[1], q[2];
cswap q[3], q[0], q[1];
swap q[1], q[2];
u(0.1, 0.2, 0.3) q[0];
p(0.5) q[1];
rxx(0.3) q[0], q[1];
ryy(0.2) q[1], q[2];
rzz(0.6) q[2], q[3];
ctrl @ ry(1.1) q[2], q[0];
y q[3];
sdg q[0];
tdg q[1];
barrier q[0], q[1];
c[0] = measure q[0];
c[1] = measure q[3];
''',
    'roundtrip': None,   # the QASM3 text the reference writes for QASM_EXPORT['zoo'], read back
}


# ---- gate classes the zoo does not reach: ProjectionJ planes, HamiltonianGate, LatentGate, CombinedSingleGate,
# Identity ----------------------------------------------------------------------------------------------------
def extra_gates_circuit(dq):
    g = torch.Generator().manual_seed(17)
    cir = dq.QubitCircuit(4)
    cir.hlayer()
    cir.j(0, 0.4, plane='xy')
    cir.j(1, -0.9, plane='yz')
    cir.j(2, 1.3, plane='zx', controls=[3])
    cir.hamiltonian([[0.7, 'x0'], [-0.4, 'z1y2'], [0.25, 'y3']], t=0.8)
    cir.hamiltonian([[1.0, 'z1z2'], [0.5, 'x2']], t=0.35, controls=[0])
    herm = torch.randn(4, 4, generator=g) + 1j * torch.randn(4, 4, generator=g)
    cir.hamiltonian((herm + herm.mH).to(torch.cfloat), t=0.2, wires=[3, 1])
    cir.latent(wires=[2], inputs=torch.randn(2, 2, generator=g))
    cir.latent(wires=[0, 3], inputs=torch.randn(4, 4, generator=g), controls=[1])
    cir.add(dq.gate.CombinedSingleGate([dq.gate.Rx(0.3), dq.gate.SGate(), dq.gate.Ry(-0.6)], nqubit=4, wires=[1]))
    cir.add(dq.gate.CombinedSingleGate([dq.gate.Hadamard(), dq.gate.PhaseShift(0.9)], nqubit=4, wires=[3], controls=[0, 2]))
    cir.add(dq.gate.Identity(nqubit=4, wires=[0, 1]))
    cir.observable([0, 1], 'xz')
    cir.observable(3, 'y')
    return cir


# ---- second order: the reference's own Hessian benchmark (examples/benchmarks/gradient_benchmark.py:147-163) and a
# circuit whose Hessian is taken with respect to nn.Parameters --------------------------------------------------
HESSIAN_CASES = [(4, 2), (4, 4), (6, 2), (6, 4), (8, 2), (8, 4)]


def hessian_benchmark_circuit(dq, n, layer):
    cir = dq.QubitCircuit(n)
    for _ in range(layer):
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        cir.rxlayer(encode=True)
        cir.rzlayer(encode=True)
        cir.rxlayer(encode=True)
    cir.observable(basis='x')
    return cir


def hessian_params(n, layer):
    g = torch.Generator().manual_seed(1000 * n + layer)
    return (torch.rand(3 * n * layer, generator=g, dtype=torch.float64) * 2 - 1) * 3.0


HESSIAN_PARAM_WEIGHTS = [0.8, -1.3, 0.5]


def hessian_param_circuit(dq, n):
    """Trainable one-qubit, controlled and two-qubit gates, two encoders, X / Z / Y-type observables."""
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.rylayer()                      # trainable
    cir.cnot_ring()
    cir.crz(1, 2)                      # trainable, controlled
    cir.rxx([0, 3])                    # trainable, two targets
    cir.u3(2)                          # trainable, three angles
    cir.ry(1, encode=True)
    cir.t(0)
    cir.rz(3, controls=[0, 2])         # trainable, two controls
    cir.observable(0)
    cir.observable([1, 2], 'xz')
    cir.observable([0, 3], 'zy')
    return cir
