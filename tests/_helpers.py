"""Shared test helpers: golden fixtures, spec-driven circuit construction, tolerances."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import specs  # noqa: E402

_GOLDEN = None

# north-star tolerances: final amplitudes and expectations
TOL = {'c64': 1e-4, 'c128': 1e-10}
CDTYPE = {'c64': torch.complex64, 'c128': torch.complex128}


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = np.load(os.path.join(HERE, 'golden', 'golden.npz'))
    return _GOLDEN


def gold(key):
    return torch.from_numpy(golden()[key])


def build_circuit(dq, name, prec, device=None):
    c = specs.CIRCUITS[name]
    cir = specs.build(dq, c['nqubit'], c['spec'])
    for wires, basis in c.get('observables', []):
        cir.observable(wires, basis)
    if device is not None:
        cir.to(device)
    if prec == 'c128':
        cir.to(torch.double)
    data = None
    if 'data' in c:
        data = torch.tensor(c['data'], dtype=torch.float64 if prec == 'c128' else torch.float32, device=device)
    return cir, data, c


def check_circuit_against_golden(dq, name, prec, device=None, check_unitary=True):
    cir, data, c = build_circuit(dq, name, prec, device)
    tol = TOL[prec]
    with torch.no_grad():
        state = cir(data=data)
        ref = gold(f'{name}/{prec}/state')
        assert state.shape == ref.shape, (state.shape, ref.shape)
        assert state.dtype == CDTYPE[prec]
        err = (state.cpu() - ref).abs().max().item()
        assert err < tol, f'{name}/{prec}: amplitude error {err}'
        if c.get('observables'):
            ev = cir.expectation()
            ref_ev = gold(f'{name}/{prec}/expectation')
            assert ev.shape == ref_ev.shape, (ev.shape, ref_ev.shape)
            assert (ev.cpu() - ref_ev).abs().max().item() < tol
        if 'marginal' in c:
            res = cir.measure(shots=64, with_prob=True, wires=c['marginal'])
            res = res if isinstance(res, list) else [res]
            ref_m = gold(f'{name}/{prec}/marginal')
            for b, r in enumerate(res):
                for bits, (cnt, p) in r.items():
                    assert abs(float(p) - ref_m[b, int(bits, 2)].item()) < max(tol, 1e-6)
        if c.get('unitary') and check_unitary:
            u = cir.get_unitary()
            ref_u = gold(f'{name}/{prec}/unitary')
            assert (u.cpu() - ref_u).abs().max().item() < tol
    return err
