"""Second-order fixtures made by running the REAL reference here (same mechanism as make_golden.py): Hessians of the
reference's own benchmark circuit (examples/benchmarks/gradient_benchmark.py:147-163: CNOT chain + Rx / Rz / Rx encoder
layers, <X..X>), taken with ``torch.autograd.functional.hessian`` exactly as the benchmark does, plus a Hessian with
respect to ``nn.Parameter``s of a circuit with controlled and two-qubit trainable gates and several observables
(double ``torch.autograd.grad``).  Only inputs and outputs are stored.

usage: python tests/golden/make_golden_hessian.py       (about a minute)
"""

import os
import sys

import numpy as np
import torch
from torch.autograd.functional import hessian

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import specs  # noqa: E402
from make_golden import import_reference, to_np  # noqa: E402


def main():
    dq = import_reference()
    out = {}

    # ---- the reference's hessian_dq circuit --------------------------------------------------------
    for n, layer in specs.HESSIAN_CASES:
        for prec in ('c64', 'c128'):
            real = torch.float64 if prec == 'c128' else torch.float32

            def f(params):
                cir = specs.hessian_benchmark_circuit(dq, n, layer)
                if prec == 'c128':
                    cir.to(torch.double)
                cir(data=params)
                return cir.expectation()

            for tag, x in (('ones', torch.ones(3 * n * layer, dtype=real)),
                           ('rand', specs.hessian_params(n, layer).to(real))):
                key = f'hessian/{n}-{layer}/{prec}/{tag}'
                xg = x.clone().requires_grad_(True)
                val = f(xg)
                (g,) = torch.autograd.grad(val, xg)
                h = hessian(f, x)
                out[f'{key}/params'] = to_np(x)
                out[f'{key}/value'] = to_np(val)
                out[f'{key}/grad'] = to_np(g)
                out[f'{key}/hessian'] = to_np(h.reshape(x.numel(), x.numel()))
                print(key, 'max|H| =', float(h.abs().max()), 'max|H - H^T| =', float((h.reshape(x.numel(), -1) - h.reshape(x.numel(), -1).T).abs().max()))

    # ---- Hessian with respect to nn.Parameters (and one data entry), several observables ------------
    for prec in ('c64', 'c128'):
        torch.manual_seed(11)
        cir = specs.hessian_param_circuit(dq, 4)
        if prec == 'c128':
            cir.to(torch.double)
        data = torch.tensor([0.7, -0.4], dtype=torch.float64 if prec == 'c128' else torch.float32, requires_grad=True)
        cir(data=data)
        ev = cir.expectation()
        w = torch.tensor(specs.HESSIAN_PARAM_WEIGHTS, dtype=ev.dtype)
        loss = (ev.reshape(-1) * w).sum() + (ev.reshape(-1) ** 2).sum()
        leaves = [data] + list(cir.parameters())
        gs = torch.autograd.grad(loss, leaves, create_graph=True)
        gflat = torch.cat([g.reshape(-1) for g in gs])
        rows = []
        for i in range(gflat.numel()):
            r = torch.autograd.grad(gflat[i], leaves, retain_graph=True, allow_unused=True)
            rows.append(torch.cat([(torch.zeros_like(p) if x is None else x).reshape(-1) for x, p in zip(r, leaves)]))
        h = torch.stack(rows)
        key = f'hessian_params/{prec}'
        out[f'{key}/data'] = to_np(data)
        out[f'{key}/params'] = np.concatenate([to_np(p).reshape(-1) for p in cir.parameters()])
        out[f'{key}/expectation'] = to_np(ev)
        out[f'{key}/grad'] = to_np(gflat)
        out[f'{key}/hessian'] = to_np(h)
        print(key, h.shape, 'max|H| =', float(h.abs().max()), 'max|H - H^T| =', float((h - h.T).abs().max()))

    path = os.path.join(HERE, 'golden_hessian.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
