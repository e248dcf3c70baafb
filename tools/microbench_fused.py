"""Cost model of the fused pass kernel: time (HIP events) of synthetic passes that differ in one thing
only (number of H / Rx / CNOT gates, number of LDS round trips).  Run under rocprofv3 --pmc to get
instruction counts per launch; every case is launched REPS times back to back.

usage: python tools/microbench_fused.py [--n 28] [--batch 4] [--dtype c64] [--reps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepquantum_amd import backend, fusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=28)
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--dtype', default='c64')
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--m', type=int, default=None)
args = ap.parse_args()

is128 = args.dtype == 'c128'
dtype = torch.complex128 if is128 else torch.complex64
dev = torch.device('cuda', 0)
n = args.n
geom = fusion.default_geometry(is128, args.m)
geom.max_gates = 40

H = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5).to(dtype)
t = torch.tensor(0.7)
RX = torch.stack([torch.cos(t / 2) + 0j, -1j * torch.sin(t / 2), -1j * torch.sin(t / 2), torch.cos(t / 2) + 0j]).reshape(2, 2).to(dtype)
G = torch.linalg.qr(torch.randn(2, 2, dtype=torch.complex128))[0].to(dtype)
X = torch.tensor([[0, 1], [1, 0]], dtype=dtype)
mats = torch.cat([H.reshape(-1), RX.reshape(-1), G.reshape(-1), X.reshape(-1)]).to(dev)
OFF = {'h': 0, 'rx': 4, 'g': 8, 'x': 12, 'ry': 0}   # 'ry': the H matrix handled as a plain real matrix


def ops_seq(kind, count, targets, controls=()):
    out = []
    for i in range(count):
        tb = targets[i % len(targets)]
        k = 'x' if kind == 'x' else 'gen'
        out.append(fusion.PrimOp(k, (tb,), tuple(controls), OFF[kind], {'h': 3, 'rx': 2, 'ry': 1}.get(kind, 0)))
    return out


hi = n - 1
cases = []
for cnt in (1, 9, 17, 33):
    cases.append((f'H x{cnt} on one gathered bit', ops_seq('h', cnt, [hi])))
for cnt in (9, 33):
    cases.append((f'Rx x{cnt} on one gathered bit', ops_seq('rx', cnt, [hi])))
    cases.append((f'real 2x2 x{cnt} on one gathered bit', ops_seq('ry', cnt, [hi])))
    cases.append((f'general 2x2 x{cnt} on one gathered bit', ops_seq('g', cnt, [hi])))
for cnt in (9, 33):
    cases.append((f'CNOT x{cnt} target gathered, control outside tile', ops_seq('x', cnt, [hi], [hi - 8])))
    cases.append((f'CNOT x{cnt} target gathered, control register slot', ops_seq('x', cnt, [hi], [hi - 1])))
    cases.append((f'CNOT x{cnt} target gathered, control thread bit', ops_seq('x', cnt, [hi], [3])))
    cases.append((f'CH x{cnt} target gathered, control thread bit', ops_seq('h', cnt, [hi], [3])))
# LDS round trips: alternate between two slot sets
cases.append(('H x8 alternating gathered/low bit (8 rounds)', ops_seq('h', 8, [hi, 2])))
cases.append(('H x16 alternating gathered/low bit (16 rounds?)', ops_seq('h', 16, [hi, 2])))
cases.append(('H x12 on 12 different bits (3+ rounds)', ops_seq('h', 12, [hi, hi - 1, hi - 2, hi - 3, hi - 4, 0, 1, 2, 3, 4, 5, 6])))

# memory side: the same 7 gates on scattered gathered bits (256-byte runs at L = 5) vs adjacent ones
scat = [n - 1, n - 3, n - 6, n - 8, n - 11, n - 13, n - 15]
cases.append(('H x7 on 7 scattered gathered bits', ops_seq('h', 7, scat)))
cases.append(('H x28 on 7 scattered gathered bits', ops_seq('h', 28, scat)))
cases.append(('H x7 on 7 adjacent bits just above L', ops_seq('h', 7, [5, 6, 7, 8, 9, 10, 11])))
x = torch.zeros(args.batch, 1 << n, dtype=dtype, device=dev)
x[:, 0] = 1
state_bytes = 2 * x.numel() * x.element_size()
print(f'# n={n} batch={args.batch} {args.dtype} m={geom.m}: physical bytes per pass {state_bytes / 1e9:.2f} GB')
for name, ops in cases:
    steps = fusion.schedule(ops, n, geom)
    assert len(steps) == 1 and isinstance(steps[0], fusion.FusedStep), (name, len(steps))
    st = steps[0]
    km = fusion.kernel_matrices(steps, ops, mats)
    backend.apply_fused(x, km, 0, st.desc, out=x)  # warm
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        backend.apply_fused(x, km, 0, st.desc, out=x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    print(f'{name:58s} gates {len(ops):3d} rounds {st.nrounds:2d} lds_trips {st.ntranspose:2d}  {ms:7.3f} ms  {state_bytes / ms / 1e6:7.0f} GB/s')
