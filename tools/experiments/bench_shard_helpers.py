"""Throughput of the shard helpers of the multi-GPU path (csrc/dq_dist.hip): permute_bits, pack, unpack_axpby."""
import os, sys, random
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from deepquantum_amd import backend

dev = torch.device('cuda', 0)
n, b = 28, 4
x = torch.randn(b, 1 << n, dtype=torch.complex64, device=dev)
out = torch.empty_like(x)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


nbytes = 2 * x.numel() * 8
rng = random.Random(1)
cases = {'identity': list(range(n)), 'swap top two': list(range(n - 2)) + [n - 1, n - 2],
         'k = 3 remap (bits 20..22 to the top)': [q for q in range(n) if q not in (20, 21, 22)] + [20, 21, 22],
         'low bit to the top': list(range(1, n)) + [0], 'random': rng.sample(range(n), n)}
for name, src in cases.items():
    ms = timeit(lambda: backend.permute_bits(x, src, out=out))
    print(f'permute_bits {name:40s} {ms:7.2f} ms  {nbytes / ms / 1e6:7.0f} GB/s')
ms = timeit(lambda: backend.pack(x, 1 << (n - 1), 1 << (n - 1)))
print(f'pack top bit (incl. its allocation)                    {ms:7.2f} ms  {1.0 * x.numel() * 8 / ms / 1e6:7.0f} GB/s (read half + write half)')
# ---- reductions (csrc/dq_reduce.hip): bytes = one read of the state(s) ---------------------------------------------------
x = x / x.norm(dim=-1, keepdim=True)
y = torch.randn_like(x)
one = x.numel() * 8
coefs = torch.randn(x.shape[0], n, dtype=torch.float64, device=x.device)
for name, fn, nbytes_ in (
        ('expect_pauli Z0', lambda: backend.expect_pauli(x, 0, 1 << (n - 1)), one),
        ('expect_pauli X0 Z3', lambda: backend.expect_pauli(x, 1 << (n - 1), 1 << (n - 4)), one),
        ('expect_pauli X(bit 0)', lambda: backend.expect_pauli(x, 1, 0), one),
        ('expect_z_multi (8 strings)', lambda: backend.expect_z_multi(x, [1 << q for q in range(8)]), one),
        ('expect_z_multi (28 ring ZZ strings)', lambda: backend.expect_z_multi(x, [(1 << q) | (1 << ((q + 1) % n)) for q in range(n)]), one),
        ('scale_z_signs (28 ring ZZ strings)', lambda: backend.scale_z_signs(x, [(1 << q) | (1 << ((q + 1) % n)) for q in range(n)], coefs), 2 * one),
        ('inner', lambda: backend.inner(x, y), 2 * one),
        ('probs', lambda: backend.probs(x), one + one // 2),
        ('marginal 5 low wires', lambda: backend.marginal(x, [0, 1, 2, 3, 4]), one),
        ('marginal 5 high wires', lambda: backend.marginal(x, [n - 1, n - 2, n - 3, n - 4, n - 5]), one),
        ('marginal 14 wires (bits 7..20)', lambda: backend.marginal(x, list(range(7, 21))), one),
        ('marginal 14 low wires', lambda: backend.marginal(x, list(range(14))), one),
        ('marginal bits 0, 9, 10, 11', lambda: backend.marginal(x, [0, 9, 10, 11]), one),
        ('marginal 12 spread wires', lambda: backend.marginal(x, list(range(1, 25, 2))), one),
        ('marginal all wires bit-reversed, batch 1', lambda: backend.marginal(x[:1], list(range(n))), one // x.shape[0] * 2),
        ('marginal all wires in order, batch 1', lambda: backend.marginal(x[:1], list(range(n - 1, -1, -1))), one // x.shape[0] * 2),
        ('marginal 27 of 28 wires, batch 1', lambda: backend.marginal(x[:1], [q for q in range(n - 1, -1, -1) if q != 13]), one // x.shape[0] * 3 // 2),
        ('gate_grad target 5', lambda: backend.gate_grad(x, y, [5], []), 2 * one),
        ('gate_grad target 27 ctrl 3', lambda: backend.gate_grad(x, y, [27], [3]), 2 * one)):
    try:
        ms = timeit(fn)
        print(f'{name:34s} {ms:7.2f} ms  {nbytes_ / ms / 1e6:7.0f} GB/s')
    except Exception as e:
        print(f'{name:34s} failed: {type(e).__name__} {str(e)[:80]}')
