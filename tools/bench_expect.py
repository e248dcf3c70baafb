"""Read-out cost of a cost Hamiltonian with one ZZ term per ring edge (QAOA / MaxCut style)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepquantum_amd as dq
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cir = dq.QubitCircuit(n)
cir.hlayer()
for q in range(n):
    cir.observable([q, (q + 1) % n], 'zz')
cir.to('cuda')
with torch.no_grad():
    cir.state = torch.randn(batch, 1 << n, 1, dtype=torch.complex64, device='cuda')
    def T(f, reps=3):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    a = T(lambda: cir.expectation())
    b = T(lambda: torch.stack([dq.qmath.expectation(cir.state, ob) for ob in cir.observables], dim=-1))
print(f'n={n} batch={batch}: {n} ZZ terms together {a:.2f} ms, one by one {b:.2f} ms')
