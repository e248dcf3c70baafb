"""Density-matrix showcase: n qubits = 2n index bits on the same fused passes.  A noisy version of the
benchmark generator: every layer applies the random H / Rx / CNOT layer, then a depolarizing channel on
every qubit.  usage: python tools/bench_density.py [--n 14] [--depth 10] [--reps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=14)
ap.add_argument('--depth', type=int, default=10)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--p', type=float, default=0.05)
ap.add_argument('--no-zero-state', action='store_true', help="every pass moves the whole matrix (as from any state but 'zeros')")
ap.add_argument('--real-bodies', action='store_true', help='A/B: channel superoperators on the real 4x4 bodies of round 4 (not the X-shaped ones)')
args = ap.parse_args()
if args.no_zero_state:
    dq.executor.CONFIG['zero_state'] = False
if args.real_bodies:
    from deepquantum_amd import channel
    for cls in (channel.BitFlip, channel.Depolarizing, channel.AmplitudeDamping, channel.Pauli, channel.GeneralizedAmplitudeDamping):
        cls._kernel_mode = 1

n = args.n
cir = dq.QubitCircuit(n, den_mat=True)
layer = 0
for i, op in enumerate(random_circuit_spec(n, args.depth, 1234)):
    if op[0] == 'h':
        cir.h(op[1])
    elif op[0] == 'rx':
        cir.rx(op[1], op[2])
    else:
        cir.cnot(op[1], op[2])
    if (i + 1) % n == 0:
        for q in range(n):
            cir.depolarizing(q, args.p)
cir.observable(0)
cir.to('cuda')
with torch.no_grad():
    cir()
    cir.expectation()          # (first use loads torch's indexing kernels: keep it out of the timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        rho = cir()
        ev = cir.expectation()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
tr = rho.diagonal().sum().real.item()
purity = (rho.abs() ** 2).sum().item()
st = dq.executor.LAST_RUN
nops = len(cir.operators)
bytes_state = 2 * 8 * 4**n
print(f'n={n} (4^n = {4**n:.3e} entries, {8 * 4**n / 2**30:.1f} GiB) depth={args.depth}: {nops} gates+channels, '
      f'{st["passes"]} fused passes, {dt * 1e3:.1f} ms per forward+expectation, trace={tr:.6f} purity={purity:.4f} '
      f'<Z0>={ev.item():+.5f}; physical {st["passes"] * bytes_state / dt / 1e9:.0f} GB/s')
