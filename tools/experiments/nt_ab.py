"""A/B of the wave-tile kernel's skeleton on big states: a few synthetic passes (few gates, so the memory side is what is
timed) with contiguous and with permuted stores.  Run once per DQ_WAVE_NT value (the knob is read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from deepquantum_amd import backend, fusion  # noqa: E402

n, b = 28, 16
dev = torch.device('cuda', 0)
H = (torch.tensor([[1, 1], [1, -1]], dtype=torch.complex64) * 2 ** -0.5).reshape(-1)


def passes(qubits, permute):
    ops = [fusion.PrimOp('gen', (q,), (), 4 * i, 3) for i, q in enumerate(qubits)]
    geom = fusion.default_geometry(False)
    geom.permute_store = permute
    steps = fusion.schedule(ops, n, geom)
    mats = H.repeat(len(qubits))
    return steps, fusion.kernel_matrices(steps, ops, mats).to(dev)


a = torch.zeros(b, 1 << n, dtype=torch.complex64, device=dev)
a[:, 0] = 1
c = torch.empty_like(a)
for name, qubits, permute in (('1 gate, in place', [0], False), ('H on every qubit, in place', list(range(n)), False),
                              ('H on every qubit, permuted stores', list(range(n)), True),
                              ('H on every qubit twice, permuted stores', list(range(n)) + list(range(n - 1, -1, -1)), True)):
    steps, md = passes(qubits, permute)
    for rep in range(3):
        cur, nxt = a, c
        times = []
        for st in steps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if permute:
                backend.apply_fused(cur, md, 0, st.desc, out=nxt)
                cur, nxt = nxt, cur
            else:
                backend.apply_fused(cur, md, 0, st.desc, out=cur)
            e1.record()
            times.append((e0, e1))
        torch.cuda.synchronize()
    print(f'NT={os.environ.get("DQ_WAVE_NT", "default")} {name:42s}', ' '.join(f'{x.elapsed_time(y):6.2f}' for x, y in times), flush=True)
