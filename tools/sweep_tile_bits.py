"""Memory-side landscape of the fused pass: time of a one-pass job (one H per gathered bit) as a function of
L (contiguous low bits) and of WHICH high bits are gathered.  usage: python tools/sweep_tile_bits.py [--batch 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepquantum_amd import backend, fusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=28)
ap.add_argument('--batch', type=int, default=4)
args = ap.parse_args()
n, dev = args.n, torch.device('cuda', 0)
H = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5).reshape(-1).to(dev)
x = torch.zeros(args.batch, 1 << n, dtype=torch.cfloat, device=dev)
x[:, 0] = 1


def run(L, bits):
    geom = fusion.default_geometry(False)
    geom.min_low = L
    geom.max_gates = 40
    ops = [fusion.PrimOp('gen', (b,), (), 0, 1) for b in bits]
    steps = fusion.schedule(ops, n, geom)
    assert len(steps) == 1, len(steps)
    st = steps[0]
    got = sorted(st.desc.high_pos[i] for i in range(st.desc.h))
    km = fusion.kernel_matrices(steps, ops, H)
    backend.apply_fused(x, km, 0, st.desc, out=x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        backend.apply_fused(x, km, 0, st.desc, out=x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f'L={st.desc.L} gathered={got}  trips={st.ntranspose}  {ms:6.3f} ms  {2 * x.numel() * 8 / ms / 1e6:6.0f} GB/s')


for L in (5, 6, 7, 8):
    h = 12 - L
    print(f'--- L = {L}: {h} gathered bits, contiguous run {8 << L} B')
    run(L, list(range(L, L + h)))                               # adjacent, just above the low bits
    run(L, list(range(n - h, n)))                               # the top bits
    run(L, list(range(14, 14 + h)))                             # a middle block
    run(L, [n - 1 - 2 * i for i in range(h)])                   # every other bit from the top
    run(L, [n - 1 - 3 * i for i in range(h)])                   # every third bit from the top
    run(L, [L + 1 + 2 * i for i in range(h)])                   # every other bit from the bottom
    run(L, list(range(L, L + h - 2)) + [n - 2, n - 1])          # low block + 2 top bits
    run(L, [10 + i for i in range(h - 1)] + [n - 1])
