"""gate_grad (one launch per gate) vs gate_grad_multi (one read of both states per group) on big states."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepquantum_amd import backend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
x = torch.randn(1, 1 << n, dtype=torch.complex64, device='cuda')
y = torch.randn(1, 1 << n, dtype=torch.complex64, device='cuda')
def T(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for label, gates in (('8 low/mid targets', [(t, []) for t in (0, 2, 5, 7, 9, 11, 12, 13)]),
                     ('8 scattered high targets', [(t, []) for t in (1, 3, 8, 13, 17, 20, 22, n - 1)]),
                     ('1 target', [(n - 2, [])])):
    a = T(lambda: [backend.gate_grad(x, y, [t], c) for t, c in gates])
    b = T(lambda: backend.gate_grad_multi(x, y, gates))
    print(f'n={n} {label}: per-gate {a:.2f} ms, multi {b:.2f} ms ({2 * x.numel() * 8 / b / 1e6:.0f} GB/s)')
