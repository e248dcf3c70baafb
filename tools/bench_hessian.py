"""Hessians beyond the launch-bound sizes: ``torch.autograd.functional.hessian`` of <Z0> of the benchmark generator's circuit
(H / CNOT fixed, every Rx angle a data input) by the tangent circuit (executor._SweepGrads: one forward and one fused reverse
sweep per row) and by the per-gate replay (two Python nodes and four launches per gate and row, one saved state per gate).
usage: python tools/bench_hessian.py [--n 16,20,24] [--depth 6] [--dtype c64]"""
import argparse
import os
import sys
import time

import torch
from torch.autograd.functional import hessian

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', default='16,20,24')
ap.add_argument('--depth', type=int, default=6)
ap.add_argument('--dtype', default='c64', choices=['c64', 'c128'])
ap.add_argument('--max-params', type=int, default=48)
args = ap.parse_args()
dev = torch.device('cuda', 0)
real = torch.float64 if args.dtype == 'c128' else torch.float32

for n in [int(v) for v in args.n.split(',')]:
    spec = random_circuit_spec(n, args.depth, 1234)
    nrx = sum(op[0] == 'rx' for op in spec)
    keep = min(nrx, args.max_params)               # the first `keep` Rx gates take data, the others keep their angles

    def build():
        cir = dq.QubitCircuit(n)
        seen = 0
        for op in spec:
            if op[0] == 'h':
                cir.h(op[1])
            elif op[0] == 'rx':
                seen += 1
                if seen <= keep:
                    cir.rx(op[1], encode=True)
                else:
                    cir.rx(op[1], op[2])
            else:
                cir.cnot(op[1], op[2])
        cir.observable(0)
        cir.to(dev)
        return cir.to(torch.double) if args.dtype == 'c128' else cir

    x = torch.rand(keep, device=dev, dtype=real, generator=None) * 6.28
    cir = build()

    def f(p):
        cir(data=p)
        return cir.expectation().sum()

    out = {}
    for mode in ('tangent', 'replay'):
        dq.executor.CONFIG['second_order'] = mode
        torch.cuda.reset_peak_memory_stats()
        h = hessian(f, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = hessian(f, x)
        torch.cuda.synchronize()
        out[mode] = (time.perf_counter() - t0, h, torch.cuda.max_memory_allocated() / 2**30)
    dq.executor.CONFIG['second_order'] = 'tangent'
    err = (out['tangent'][1] - out['replay'][1]).abs().max().item()
    print(f'n = {n:2d}, {len(spec)} gates, {keep} of {nrx} Rx angles differentiated, {args.dtype}: Hessian by the tangent circuit '
          f'{out["tangent"][0]:7.3f} s (peak {out["tangent"][2]:.2f} GiB), by the per-gate replay {out["replay"][0]:7.3f} s '
          f'(peak {out["replay"][2]:.2f} GiB); max difference {err:.1e}', flush=True)
