"""Generate the golden fixtures by running the REAL reference (TuringQ/deepquantum v4.5.0 mounted at
/root/reference) in the build container.  Only inputs and outputs are stored (tests/golden/*.npz,
*.json); no reference source travels.

The reference imports three packages that are not installed here and that the statevector path never
calls (qiskit -> only QubitCircuit.draw, bayes_opt -> optimizer.py, svgwrite -> photonic/draw.py).
Empty placeholder modules for them are created in a temporary directory so that ``import deepquantum``
succeeds; nothing on the path uses them.

usage: python tests/golden/make_golden.py        (about 3 minutes; the n=24 pin dominates)
"""

import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from specs import CIRCUITS, GATE_CASES, build, random_spec  # noqa: E402


def import_reference():
    stub = tempfile.mkdtemp(prefix='dq_ref_stubs_')
    for pkg, body in (('qiskit', 'class QuantumCircuit: pass\n'),
                      ('bayes_opt', 'class BayesianOptimization: pass\nclass UtilityFunction: pass\n'),
                      ('svgwrite', '')):
        os.makedirs(os.path.join(stub, pkg))
        with open(os.path.join(stub, pkg, '__init__.py'), 'w') as f:
            f.write(body)
    sys.path.insert(0, stub)
    sys.path.insert(0, '/root/reference/src')
    import deepquantum as dq

    assert dq.__version__ == '4.5.0', dq.__version__
    return dq


def to_np(t):
    return t.detach().cpu().resolve_conj().numpy()


def main():
    dq = import_reference()
    torch.manual_seed(0)
    out = {}
    meta = {'reference_version': dq.__version__, 'torch': torch.__version__, 'cases': {}}

    # ---- F1/F3/F4/F6: circuits (state, expectations, marginals, unitary) -------------------------
    for name, c in CIRCUITS.items():
        for prec in c.get('precisions', ('c64', 'c128')):
            cir = build(dq, c['nqubit'], c['spec'])
            for wires, basis in c.get('observables', []):
                cir.observable(wires, basis)
            if prec == 'c128':
                cir.to(torch.double)
            data = None
            if 'data' in c:
                data = torch.tensor(c['data'], dtype=torch.float64 if prec == 'c128' else torch.float32)
            with torch.no_grad():
                state = cir(data=data)
                key = f'{name}/{prec}'
                out[f'{key}/state'] = to_np(state)
                if c.get('observables'):
                    out[f'{key}/expectation'] = to_np(cir.expectation())
                if 'marginal' in c:
                    flat = state.reshape(-1, 2 ** c['nqubit'])
                    n = c['nqubit']
                    probs = []
                    for i in range(flat.shape[0]):
                        p = torch.abs(flat[i]) ** 2
                        w = sorted(c['marginal'])
                        pm = w + [j for j in range(n) if j not in w]
                        probs.append(p.reshape([2] * n).permute(pm).reshape(2 ** len(w), -1).sum(-1))
                    out[f'{key}/marginal'] = to_np(torch.stack(probs))
                if c.get('unitary'):
                    out[f'{key}/unitary'] = to_np(cir.get_unitary())
            meta['cases'][key] = {'nqubit': c['nqubit'], 'ngates': len(c['spec'])}
            print('circuit', key, tuple(state.shape))

    # ---- F2: every gate class, several positions, controls, inverse --------------------------------
    g = torch.Generator().manual_seed(77)
    for i, case in enumerate(GATE_CASES):
        n = case['nqubit']
        psi = torch.randn(2**n, 1, generator=g, dtype=torch.float64) + 1j * torch.randn(2**n, 1, generator=g, dtype=torch.float64)
        psi = (psi / psi.norm()).to(torch.complex128)
        cls = getattr(dq.gate, case['cls'])
        gate = cls(nqubit=n, **case['kwargs'])
        if case.get('inverse'):
            gate = gate.inverse()
        gate = gate.to(torch.double)
        with torch.no_grad():
            res = gate(psi)
        out[f'gate/{i}/in'] = to_np(psi)
        out[f'gate/{i}/out'] = to_np(res)
        out[f'gate/{i}/matrix'] = to_np(gate.update_matrix())
    meta['n_gate_cases'] = len(GATE_CASES)
    print('gate cases', len(GATE_CASES))

    # ---- F5: autograd (dense path) --------------------------------------------------------------------
    def grad_case(tag, nqubit, builder, data):
        cir = builder(dq, nqubit)
        data = data.clone().requires_grad_(True)
        cir(data)
        ev = cir.expectation()
        ev.sum().backward()
        out[f'grad/{tag}/data'] = to_np(data)
        out[f'grad/{tag}/expectation'] = to_np(ev)
        out[f'grad/{tag}/data_grad'] = to_np(data.grad)
        pg = [to_np(p.grad) for p in cir.parameters()]
        out[f'grad/{tag}/param_grads'] = np.concatenate([x.reshape(-1) for x in pg]) if pg else np.zeros(0)
        out[f'grad/{tag}/params'] = np.concatenate([to_np(p).reshape(-1) for p in cir.parameters()]) if pg else np.zeros(0)

    from specs import grad_circuit_a, qaoa_circuit

    torch.manual_seed(3)
    grad_case('mixed', 4, grad_circuit_a, torch.tensor([0.3, -0.7, 1.1, 0.5, 2.0]))
    torch.manual_seed(4)
    grad_case('qaoa', 7, qaoa_circuit, torch.tensor([0.4, 0.9]))

    # ---- F8: large-n pin, config 2 (n=24, depth 20, complex128) ---------------------------------------
    n, depth = 24, 20
    spec = random_spec(n, depth, 1234)
    cir = build(dq, n, spec)
    cir.observable(0)
    cir.to(torch.double)
    with torch.no_grad():
        state = cir().reshape(-1)
        ev = cir.expectation()
    idx = torch.randint(0, 2**n, (1024,), generator=torch.Generator().manual_seed(5))
    out['pin24/indices'] = idx.numpy()
    out['pin24/amplitudes'] = to_np(state[idx])
    out['pin24/norm2'] = np.array((state.abs() ** 2).sum().item())
    out['pin24/expectation_z0'] = to_np(ev)
    print('pin24 norm2', out['pin24/norm2'], 'Z0', out['pin24/expectation_z0'])

    np.savez_compressed(os.path.join(HERE, 'golden.npz'), **out)
    with open(os.path.join(HERE, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('wrote', os.path.join(HERE, 'golden.npz'), os.path.getsize(os.path.join(HERE, 'golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
