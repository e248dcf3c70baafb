"""BASELINE.json configs[2]: n = 24, depth 20, complex128, one sample, fixed angles — forward + <Z0>.
At this size a pass over the 256 MiB state takes ~0.15 ms, so host work per forward shows; --profile prints the
host-side breakdown; --cpu times the WHOLE circuit on the host cores with the oracle (the CPU restatement of the
reference's evolve_state path, oracle/statevec_oracle.py -- test infrastructure, here as the timed CPU baseline that
BASELINE.md section 2 quotes from the reference itself: 60.3 s on 8 cores) and checks the GPU state against it.
usage: python tools/bench_config2.py [--n 24] [--depth 20] [--reps 50] [--profile] [--cpu]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=24)
ap.add_argument('--depth', type=int, default=20)
ap.add_argument('--reps', type=int, default=50)
ap.add_argument('--profile', action='store_true')
ap.add_argument('--cpu', action='store_true', help='time the whole circuit on the host cores (the oracle) as well')
args = ap.parse_args()

gen = torch.Generator().manual_seed(7)
cir = dq.QubitCircuit(args.n)
spec = []
for op in random_circuit_spec(args.n, args.depth, 1234):
    if op[0] == 'h':
        cir.h(op[1])
        spec.append(op)
    elif op[0] == 'rx':
        theta = float(torch.rand((), generator=gen)) * 6.28
        cir.rx(op[1], inputs=theta)
        spec.append(('rx', op[1], theta))
    else:
        cir.cnot(op[1], op[2])
        spec.append(op)
cir.observable(0)
cir.to(torch.double).to('cuda')


def step():
    with torch.no_grad():
        cir()
        return cir.expectation()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.reps * 1e3
ngates = len(cir.operators)
print(f'n={args.n} depth={args.depth} complex128: {ms:.3f} ms per forward+expectation, '
      f'{ngates} gates, {ngates / ms * 1e3:.3e} gate-applies/s')
if args.profile:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.reps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)

if args.cpu:
    from oracle import statevec_oracle as oracle

    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = oracle.run_spec(args.n, spec, dtype=torch.complex128)
        ev_ref = oracle.expectation_pauli(ref, [0], 'z')
        cpu_s = time.perf_counter() - t0
        got = cir().reshape(1, -1).cpu()
        ev = step()
    err = (got - ref).abs().max().item()
    print(f'host cores ({torch.get_num_threads()} threads, oracle = the reference\'s permute / reshape / matmul path): {cpu_s:.2f} s per '
          f'forward+expectation = {ngates / cpu_s:.2f} gate-applies/s; GPU / CPU = {cpu_s * 1e3 / ms:.0f}x; max amplitude difference '
          f'{err:.2e}, <Z0> {ev.item():+.9f} vs {ev_ref.item():+.9f}')
    assert err < 1e-10
