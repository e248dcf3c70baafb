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
