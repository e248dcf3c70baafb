"""Launch-bound regime: circuits smaller than a tile (typical QML sizes) with a batch of encoded samples.
Compares the padded / batch-folded fused path with one launch per gate.
usage: python tools/bench_small.py [--n 8] [--depth 20] [--batch 64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=8)
ap.add_argument('--depth', type=int, default=20)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()


def build(trainable):
    cir = dq.QubitCircuit(args.n)
    for op in random_circuit_spec(args.n, args.depth, 1234):
        if op[0] == 'h':
            cir.h(op[1])
        elif op[0] == 'rx':
            cir.rx(op[1]) if trainable else cir.rx(op[1], encode=True)
        else:
            cir.cnot(op[1], op[2])
    cir.observable(0)
    return cir.to('cuda')


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.reps * 1e3


for label, thresh, sweep in (('fused (small-state path)', 6, True), ('fused, sweep: undo-then-reduce', 6, False),
                             ('one launch per gate', 10**9, True)):
    dq.executor.CONFIG['small_fuse_min_gates'] = thresh
    dq.executor.CONFIG['small_fused_sweep'] = sweep      # the reverse sweep on the zero-padded (psi, lambda) pair
    cir = build(trainable=False)
    data = torch.rand(args.batch, cir.ndata, device='cuda') * 6.28
    with torch.no_grad():
        fwd = timeit(lambda: (cir(data), cir.expectation()))
    cir2 = build(trainable=True)

    def step():
        cir2.zero_grad()
        cir2()
        cir2.expectation().sum().backward()

    tr = timeit(step)
    # the same two jobs replayed as HIP graphs
    with torch.no_grad():
        gf = dq.CapturedGraph(lambda: (cir(data), cir.expectation())[1])
    fwd_g = timeit(gf.replay)
    cir3 = build(trainable=True)

    def step3():
        cir3()
        loss = cir3.expectation().sum()
        loss.backward()
        return loss

    cir3.zero_grad(set_to_none=True)
    gt = dq.CapturedGraph(step3)
    tr_g = timeit(gt.replay)
    print(f'{label:30s} n={args.n} depth={args.depth} ({args.n * args.depth} gates) batch={args.batch}: '
          f'no-grad forward+<Z0> {fwd:7.2f} ms eager / {fwd_g:6.3f} ms HIP graph (per-sample angles); '
          f'training step (batch 1, every Rx trainable) {tr:7.2f} ms eager / {tr_g:6.3f} ms HIP graph')
