"""The per-gate kernel (dq_apply_gate_*: autograd per-gate nodes, torch.vmap, channels with gradients, Reset) on a
2 GiB state: physical GB/s by target / control position.  usage (GPU box): python tools/bench_single_gate_kernel.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepquantum_amd import backend
dev = torch.device('cuda', 0)
n = 28
x = torch.zeros(1, 1 << n, dtype=torch.complex64, device=dev); x[0, 0] = 1
m = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2 ** 0.5).to(dev)
cn = torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat).to(dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for tb in (0, 1, 5, 14, 27):
    ms = t(lambda: backend.apply_gate(x, m, [tb], [], out=x))
    print(f'H on bit {tb}: {ms:.3f} ms = {2 * x.numel() * 8 / ms / 1e6:.0f} GB/s')
for c, tb in ((0, 5), (5, 0), (9, 20)):
    ms = t(lambda: backend.apply_gate(x, cn, [tb], [c], out=x))
    print(f'CNOT {c}->{tb}: {ms:.3f} ms = {x.numel() * 8 / ms / 1e6:.0f} GB/s (touched bytes)')
