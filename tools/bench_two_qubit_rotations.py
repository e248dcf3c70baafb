"""Layers of two-qubit Pauli rotations (Rxx, Ryy, Rxy -- the entangling layers of the reference's ansatz library and of
its own test circuit, tests/test_circuit.py:87-139) between Rx layers: the X-shaped bodies of the pass kernel
(DQ_MODE_XCPLX: a complex 2x2 on (00, 11) and one on (01, 10)) against the general 4x4 bodies.
usage: python tools/bench_two_qubit_rotations.py [--n 26] [--layers 6] [--batch 1]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=26)
ap.add_argument('--layers', type=int, default=6)
ap.add_argument('--reps', type=int, default=5)
args = ap.parse_args()
n = args.n


def build():
    torch.manual_seed(0)
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    for layer in range(args.layers):
        cir.rxlayer()
        for q in range(layer % 2, n - 1, 2):
            (cir.rxx, cir.ryy, cir.rxy)[(q + layer) % 3]([q, q + 1])
    cir.observable(0)
    return cir.to('cuda')


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.reps * 1e3


rows = {}
for label, mode in (('general 4x4 bodies', 0), ('X-shaped bodies', 5)):
    for cls in (dq.Rxx, dq.Ryy, dq.Rxy):
        cls._kernel_mode2 = mode
    cir = build()

    def forward():
        with torch.no_grad():
            cir()
            return cir.expectation()

    def step():
        cir.zero_grad()
        cir()
        cir.expectation().sum().backward()

    f = timed(forward)
    passes = dq.executor.LAST_RUN['passes']
    s = timed(step)
    rows[label] = (f, passes, s, dict(dq.executor.LAST_SWEEP))
ngates = sum(1 for _ in build().operators)
print(f'n = {n}, {args.layers} layers of Rx + Rxx / Ryy / Rxy on neighbours, complex64')
for label, (f, passes, s, sweep) in rows.items():
    print(f'{label:22s}: forward + <Z0> {f:7.2f} ms ({passes} passes), training step {s:7.2f} ms (sweep: {sweep.get("passes")} passes)')
