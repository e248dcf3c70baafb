"""Randomised soak of the product path on the GPU, beyond what the test-suite runs every time: circuits over the whole
gate vocabulary under every scheduler configuration against the oracle (tests/_helpers.check_fuzz_against_oracle) and
fused reverse sweeps against per-gate autograd (check_fused_sweep_random), many seeds.
usage (GPU box): python tools/soak.py [first_seed] [count] [small | hvp | func]
``func``: the fused node under torch.func transforms (vmap over the circuit, jacrev, vmap(grad), vmap(jacrev)) against the
native batch and plain autograd, n = 5 .. 16, both precisions.
``hvp``: Hessian-vector products of random circuits by the tangent circuit (executor._SweepGrads) against the per-gate
replay, n = 3 .. 14, both precisions.
``small``: states below a tile only (n = 3 .. 11) -- the zero-padded forward and the reverse sweep on the zero-padded
(psi, lambda) pair (executor.CONFIG['small_fused_sweep']), both precisions every seed."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepquantum_amd as dq  # noqa: E402
from _helpers import check_fused_sweep_random, check_fuzz_against_oracle, check_hvp_random, check_transforms_random  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
small = len(sys.argv) > 3 and sys.argv[3] == 'small'
hvp = len(sys.argv) > 3 and sys.argv[3] == 'hvp'
func = len(sys.argv) > 3 and sys.argv[3] == 'func'
dev = torch.device('cuda', 0)
t0 = time.time()
for k in range(count if func else 0):
    seed = first + k
    n = 5 + seed % 12
    check_transforms_random(dq, device=dev, n=n, seed=seed, ngates=20 + 6 * (seed % 9))
    check_transforms_random(dq, device=dev, n=n, seed=seed, ngates=20 + 6 * (seed % 9), dtype=torch.float64, tol=1e-10)
    print(f'seed {seed}: n = {n} ok, fused nodes so far {dq.executor.LAST_RUN.get("fused_transform_nodes", 0)} ({time.time() - t0:.0f} s)', flush=True)
if func:
    print(f'{count} seeds from {first} (torch.func transforms over random circuits): all agree')
    sys.exit(0)
for k in range(count if hvp else 0):
    seed = first + k
    n = 3 + seed % 12
    check_hvp_random(dq, device=dev, n=n, batch=1 + seed % 3, seed=seed, ngates=15 + 5 * (seed % 8))
    check_hvp_random(dq, device=dev, n=n, batch=1 + seed % 3, seed=seed, ngates=15 + 5 * (seed % 8), tol=3e-4, dtype=torch.float32)
    print(f'seed {seed}: n = {n} ok ({time.time() - t0:.0f} s)', flush=True)
if hvp:
    print(f'{count} seeds from {first} (Hessian-vector products, tangent circuit vs per-gate replay): all agree')
    sys.exit(0)
for k in range(count if small else 0):
    seed = first + k
    n = 3 + seed % 9
    check_fused_sweep_random(dq, device=dev, n=n, batch=1 + seed % 3, seed=seed, ngates=20 + 10 * (seed % 7))
    check_fused_sweep_random(dq, device=dev, n=n, batch=1 + seed % 3, seed=seed, ngates=20 + 10 * (seed % 7), tol=1e-10,
                             dtype=torch.float64)
    if n >= 6:
        check_fuzz_against_oracle(dq, device=dev, n=n, seeds=(seed,), depth=5 + seed % 4, batch=1 + seed % 3, double=(seed % 2 == 1))
    print(f'seed {seed}: n = {n} ok ({time.time() - t0:.0f} s)', flush=True)
if small:
    print(f'{count} seeds from {first} (states below a tile): all agree')
    sys.exit(0)
for k in range(count):
    seed = first + k
    n = 13 + seed % 5
    check_fuzz_against_oracle(dq, device=dev, n=n, seeds=(seed,), depth=5 + seed % 4, batch=1 + seed % 3, double=(seed % 4 == 3))
    check_fused_sweep_random(dq, device=dev, n=12 + seed % 6, batch=1 + seed % 2, seed=seed, ngates=70 + 10 * (seed % 5))
    if seed % 3 == 0:        # complex128: the wave-tile kernel's float64 reductions, exact inverses / corrections, 1e-10
        check_fused_sweep_random(dq, device=dev, n=11 + seed % 7, batch=1 + seed % 2, seed=seed, ngates=60 + 10 * (seed % 5),
                                 tol=1e-10, dtype=torch.float64)
    print(f'seed {seed}: n = {n} ok ({time.time() - t0:.0f} s)', flush=True)
print(f'{count} seeds from {first}: all agree')
