"""Debug aid for the wave-tile kernel: tiny passes (one or two gates) on the GPU against the oracle, one line per case, so
that a wrong handler / trip / offset shows up by name.  python tools/debug_wave.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from deepquantum_amd import backend, fusion  # noqa: E402
from oracle import statevec_oracle as oracle  # noqa: E402
import _wave_emulator as emu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
x = (x / x.norm(dim=-1, keepdim=True)).to(torch.complex64)
H = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) * 2 ** -0.5
RX = torch.tensor([[0.8, -0.6j], [-0.6j, 0.8]], dtype=torch.complex128)
RY = torch.tensor([[0.8, -0.6], [0.6, 0.8]], dtype=torch.complex128)
a_ = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
GEN, _ = torch.linalg.qr(a_)
X = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)


def run(name, gates):
    ops, mats, off = [], [], 0
    for kind, t, c, m, mode in gates:
        t = (t,) if isinstance(t, int) else tuple(t)
        ops.append(fusion.PrimOp(kind, t, tuple(c), off, mode))
        mats.append(m.reshape(-1))
        off += m.numel()
    mats = torch.cat(mats).to(torch.complex64)
    geom = fusion.default_geometry(False)
    steps = fusion.schedule(ops, n, geom)
    ref = x
    for op in ops:
        d_ = 1 << op.k
        ref = oracle.apply_gate_bits(ref, mats[op.mat:op.mat + d_ * d_].reshape(d_, d_), list(op.targets), list(op.controls))
    km = fusion.kernel_matrices(steps, ops, mats)
    xd = x.to(dev)
    xe = x.numpy().copy()
    ids = []
    for st in steps:
        backend.apply_fused(xd, km.to(dev), 0, st.desc, out=xd)
        xe = emu.run_pass(st.desc, n, xe, km.numpy(), 0)
        kp = emu.descriptor(st.desc, n)
        ids += [kp.rec[i][0] for i in range(kp.nrec_bytes // 32)]
    torch.cuda.synchronize()
    err = (xd.cpu() - ref).abs().max().item()
    erre = (torch.from_numpy(xe) - ref).abs().max().item()
    print(f'{name:40s} gpu err {err:9.2e}  emu err {erre:9.2e}  ids {ids}  {"OK" if err < 1e-5 else "FAIL"}', flush=True)


D1 = torch.diag(torch.exp(1j * torch.tensor([0.3, 1.1], dtype=torch.float64)))
D2 = torch.diag(torch.exp(1j * torch.tensor([0.3, 1.1, -0.7, 2.0], dtype=torch.float64)))
if len(sys.argv) > 2 and sys.argv[2] == 'diag':
    for q in (0, 2, 5, 9, n - 1):
        run(f'DIAG1 q{q}', [('diag', q, (), D1, 0)])
        run(f'H q4; DIAG1 q{q}', [('gen', 4, (), H, 3), ('diag', q, (), D1, 0)])
    for c, q in ((1, 0), (0, 1), (n - 1, 5), (5, n - 1), (9, 10), (2, 3)):
        run(f'C-DIAG1 c{c} q{q}', [('diag', q, (c,), D1, 0)])
        run(f'H q4; C-DIAG1 c{c} q{q}', [('gen', 4, (), H, 3), ('diag', q, (c,), D1, 0)])
    for a_, b_ in ((0, 1), (1, 0), (5, 9), (9, 5), (0, n - 1), (n - 1, 0), (2, 3), (3, n - 1), (9, 10)):
        run(f'DIAG2 {a_},{b_}', [('diag', (a_, b_), (), D2, 0)])
        run(f'H q4; DIAG2 {a_},{b_}', [('gen', 4, (), H, 3), ('diag', (a_, b_), (), D2, 0)])
        run(f'H q4; C7-DIAG2 {a_},{b_}', [('gen', 4, (), H, 3), ('diag', (a_, b_), (7,), D2, 0)])
    sys.exit(0)
for q in range(n):
    run(f'H q{q}', [('gen', q, (), H, 3)])
for q in (0, 5, 11):
    run(f'RX q{q}', [('gen', q, (), RX, 2)])
    run(f'RY q{q}', [('gen', q, (), RY, 1)])
    run(f'GEN q{q}', [('gen', q, (), GEN, 0)])
    run(f'X q{q}', [('x', q, (), X, 0)])
for t, c in ((0, 1), (1, 0), (0, 11), (11, 0), (5, 6), (11, 10), (3, 8)):
    run(f'CNOT c{c} t{t}', [('x', t, (c,), X, 0)])
    run(f'CGEN c{c} t{t}', [('gen', t, (c,), GEN, 0)])
run('H q0, H q5', [('gen', 0, (), H, 3), ('gen', 5, (), H, 3)])
run('H q5, H q0', [('gen', 5, (), H, 3), ('gen', 0, (), H, 3)])
run('toffoli', [('x', 2, (7, 9), X, 0)])
