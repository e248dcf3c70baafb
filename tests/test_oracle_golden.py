"""The CPU oracle (oracle/statevec_oracle.py) against the golden vectors captured from the real reference
(tests/golden/make_golden.py).  This is what pins the oracle: same torch, same op sequence -> the results
agree to the last few ulps (bit-exact in practice; BLAS thread count may reorder a 4-term sum)."""

import pytest
import torch

from _helpers import CDTYPE, gold, gold_extra, specs
from oracle import statevec_oracle as oracle


def to_oracle_spec(spec):
    out = []
    for m, args, _ in spec:
        if m == 'h':
            out.append(('h', args[0]))
        elif m == 'rx':
            out.append(('rx', args[0], args[1]))
        elif m == 'cnot':
            out.append(('cnot', args[0], args[1]))
        else:
            raise KeyError(m)
    return out


@pytest.mark.parametrize('name', ['readme', 'rand4', 'rand8', 'rand12', 'rand16', 'rand14_seed1234'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_oracle_circuits_match_reference(name, prec):
    c = specs.CIRCUITS[name]
    n = c['nqubit']
    state = oracle.run_spec(n, to_oracle_spec(c['spec']), dtype=CDTYPE[prec])
    ref = gold(f'{name}/{prec}/state').reshape(1, -1)
    tight = 1e-6 if prec == 'c64' else 1e-14
    assert (state - ref).abs().max().item() < tight
    for k, (wires, basis) in enumerate(c['observables']):
        ev = oracle.expectation_pauli(state, wires, basis)
        assert abs(ev.item() - gold(f'{name}/{prec}/expectation')[k].item()) < tight
    if 'marginal' in c:
        p = oracle.probabilities(state, c['marginal'])
        assert (p - gold(f'{name}/{prec}/marginal')).abs().max().item() < tight


def test_oracle_generator_is_the_survey_generator():
    # tests/golden/specs.random_spec and oracle.random_circuit_spec are the same seeded generator
    a = to_oracle_spec(specs.random_spec(9, 7, 1234))
    b = oracle.random_circuit_spec(9, 7, 1234)
    assert a == b
    import bench

    assert bench.random_circuit_spec(9, 7, 1234) == b      # bench.py states the generator itself


def test_oracle_gate_cases_match_reference():
    for i, case in enumerate(specs.GATE_CASES):
        n = case['nqubit']
        kw = case['kwargs']
        wires = kw.get('wires')
        if wires is None:
            mm = kw['minmax']
            wires = list(range(mm[0], mm[1] + 1))
        controls = kw.get('controls', [])
        controls = [controls] if isinstance(controls, int) else controls
        psi = gold(f'gate/{i}/in').reshape(1, -1)
        mat = gold(f'gate/{i}/matrix')
        out = oracle.apply_gate_wires(psi, mat, n, wires, controls)
        ref = gold(f'gate/{i}/out').reshape(1, -1)
        assert (out - ref).abs().max().item() < 1e-14, (i, case['cls'])


def test_oracle_matrices_match_reference():
    # float32-rounded constants even in complex128 (SURVEY summary item 3)
    idx = next(i for i, c in enumerate(specs.GATE_CASES) if c['cls'] == 'Hadamard')
    h = gold(f'gate/{idx}/matrix')
    assert h.dtype == torch.complex128
    assert torch.equal(h, oracle.fixed_matrix('h').to(torch.complex128))
    assert h[0, 0].real.item() == 0.7071067690849304
    idx = next(i for i, c in enumerate(specs.GATE_CASES) if c['cls'] == 'Rx' and not c.get('inverse') and 'controls' not in c['kwargs'])
    rx = gold(f'gate/{idx}/matrix')
    assert torch.equal(rx, oracle.rx_matrix(oracle.theta_tensor(0.37).to(torch.float64)).to(torch.complex128))


def test_oracle_large_pin_n24():
    """Config 2 (n=24, depth 20, complex128): squared norm, <Z0> and 1024 seeded amplitudes."""
    n = 24
    spec = oracle.random_circuit_spec(n, 20, 1234)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        state = oracle.run_spec(n, spec, dtype=torch.complex128)
    idx = gold('pin24/indices')
    assert (state[0, idx] - gold('pin24/amplitudes')).abs().max().item() < 1e-14
    assert abs((state.abs() ** 2).sum().item() - gold('pin24/norm2').item()) < 1e-12
    ev = oracle.expectation_pauli(state.contiguous(), [0], 'z')
    assert abs(ev.item() - gold('pin24/expectation_z0').item()) < 1e-12


def test_oracle_reset_matches_reference():
    for i, (wires, ps, _kind) in enumerate(specs.RESET_GATE_CASES):
        out = oracle.reset_state(gold_extra(f'resetgate/{i}/in'), 4, wires, ps)
        assert (out - gold_extra(f'resetgate/{i}/out')).abs().max().item() < 1e-6, (i, wires, ps)
