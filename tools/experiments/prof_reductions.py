"""Kernel-trace target: the multi-string Z expectation / its backward / marginals at n = 28, batch 4, complex64
(rocprofv3 --kernel-trace --stats -- python tools/experiments/prof_reductions.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepquantum_amd import backend  # noqa: E402

n, b = 28, 4
x = torch.randn(b, 1 << n, dtype=torch.complex64, device='cuda')
x = x / x.norm(dim=-1, keepdim=True)
coefs = torch.randn(b, n, dtype=torch.float64, device='cuda')
ring = [(1 << q) | (1 << ((q + 1) % n)) for q in range(n)]
for _ in range(5):
    backend.expect_z_multi(x, [1 << q for q in range(8)])
    backend.expect_z_multi(x, ring)
    backend.scale_z_signs(x, ring, coefs)
    backend.expect_pauli(x, 0, 1 << (n - 1))
    backend.marginal(x, [0, 1, 2, 3, 4])
    backend.marginal(x, [0, 9, 10, 11])
    backend.marginal(x[:1], list(range(n)))
torch.cuda.synchronize()
