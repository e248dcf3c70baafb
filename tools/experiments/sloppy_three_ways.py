import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepquantum_amd as dq
n = 12
DEV = os.environ.get('DEV', 'cuda')
if DEV == 'cpu':
    from _cpu_backend import CpuTestBackend
    dq.backend.set_test_backend(CpuTestBackend())
def build(dev):
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    cir = dq.QubitCircuit(n)
    cir.hlayer(); cir.rylayer()
    for k in range(10):
        a = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
        u = (torch.linalg.qr(a)[0] * (1 + 4e-5 * (1 if k % 2 else -1))).to(torch.complex64)
        cir.any(u, wires=[k % n], controls=[(k + 3) % n] if k % 3 == 0 else None)
        cir.cnot(k % n, (k + 1) % n)
        cir.rx((k + 2) % n)
    if os.environ.get('TWO', '1') == '1':
        a = torch.randn(4, 4, generator=g, dtype=torch.float64) + 1j * torch.randn(4, 4, generator=g, dtype=torch.float64)
        cir.any((torch.linalg.qr(a)[0] * (1 + 3e-5)).to(torch.complex64), wires=[1, n - 2])
    cir.rxlayer()
    cir.observable(0); cir.observable([1, n - 1], 'xz')
    return cir.to(dev)
res = {}
for mode, fused in (('per_gate', False), ('adjoint', False), ('adjoint', True)):
    dq.executor.CONFIG['grad_mode'] = mode; dq.executor.CONFIG['fused_sweep'] = fused
    cir = build(DEV)
    cir()
    loss = (cir.expectation() * torch.tensor([1.0, -0.7], device=DEV)).sum()
    loss.backward()
    res[mode, fused] = (loss.item(), torch.stack([p.grad.cpu().reshape(-1)[0] for p in cir.parameters()]))
a = res['per_gate', False]
for k, v in res.items():
    print(k, 'loss', v[0], 'max diff to per_gate', (v[1] - a[1]).abs().max().item(), 'argmax', (v[1] - a[1]).abs().argmax().item())
print({k: v[1][:6].tolist() for k, v in res.items()})
