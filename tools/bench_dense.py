"""Dense gates on many wires and get_unitary: MFMA path (csrc/dq_dense.hip) against the round-1 VALU kernel.
usage (GPU box): python tools/bench_dense.py [--n 26]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq
from deepquantum_amd import _lib, backend

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=26)
args = ap.parse_args()
dev = torch.device('cuda', 0)
lib = _lib.load()
PEAK = {torch.complex64: 157.3, torch.complex128: 78.6}      # TFLOP/s, f32 / f64 MFMA = vector peak (MI355X_MICROARCH.md)


def unitary(k, dtype):
    g = torch.Generator().manual_seed(k)
    a = torch.randn(1 << k, 1 << k, generator=g, dtype=torch.float64) + 1j * torch.randn(1 << k, 1 << k, generator=g, dtype=torch.float64)
    return torch.linalg.qr(a)[0].to(dtype).to(dev)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('# one dense gate on k wires of an n-qubit state (targets = bits 5 .. 5+k-1), out of place')
print('# dtype  n  k   MFMA ms   TFLOP/s  frac-of-peak   GB/s(phys)   VALU ms   speed-up')
for dtype in (torch.complex64, torch.complex128):
    n = args.n if dtype == torch.complex64 else args.n - 1
    x = torch.zeros(1, 1 << n, dtype=dtype, device=dev); x[0, 0] = 1
    out = torch.empty_like(x)
    for k in (5, 6, 7, 8, 9, 10):
        m = unitary(k, dtype)
        t = list(range(5, 5 + k))
        lib.dq_set_dense_path(1)
        ms = timeit(lambda: backend.apply_gate(x, m, t, [], out=out))
        lib.dq_set_dense_path(0)
        ms_old = timeit(lambda: backend.apply_gate(x, m, t, [], out=out), reps=1)
        lib.dq_set_dense_path(1)
        flop = 8.0 * (1 << k) * (1 << n)
        gbs = 2 * x.numel() * x.element_size() / (ms * 1e-3) / 1e9
        print(f'{str(dtype)[-3:]:>5} {n:3d} {k:2d} {ms:9.3f} {flop / ms / 1e9:9.1f} {flop / ms / 1e9 / PEAK[dtype]:10.3f} {gbs:12.0f} {ms_old:9.2f} {ms_old / ms:9.1f}x')

print('# QubitCircuit.get_unitary(): n-qubit circuit with dense UAnyGate blocks (batch = 2^n identity columns)')
for n, k in ((12, 8), (12, 10), (13, 10)):
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    cir.any(unitary(k, torch.complex64).cpu(), wires=list(range(1, 1 + k)))
    cir.cnot_ring()
    cir.any(unitary(k, torch.complex64).cpu(), wires=list(range(n - k, n)))
    cir.to(dev)
    res = []
    for path in (1, 0):
        lib.dq_set_dense_path(path)
        with torch.no_grad():
            ms = timeit(lambda: cir.get_unitary(), reps=2 if path else 1)
        res.append(ms)
    lib.dq_set_dense_path(1)
    print(f'  n={n} two dense blocks k={k}: MFMA {res[0]:.2f} ms, VALU {res[1]:.2f} ms ({res[1] / res[0]:.1f}x)')
