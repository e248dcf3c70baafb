"""Pin of the HEADLINE workload (BASELINE config 3): the REAL reference (imported from
/root/reference in the build container, as in make_golden.py) runs QubitCircuit(28), seed-1234 H/Rx/CNOT
circuit of depth 40 (1120 gates), complex64, on batch element 0 of the benchmark (the generator's own
angles, fed through the encoder exactly as bench.py does).  Stored: 4096 amplitudes at seeded indices,
the squared norm, <Z_q> for every wire q from the reference's own ``expectation()``, and 5-wire marginals.
Only outputs are stored (tests/golden/pin28.npz).

usage: nohup python tests/golden/make_golden_pin28.py > /tmp/pin28.log 2>&1 &     (1-2 hours on 8 cores, ~10 GiB)

``--sample K`` (round 6; K in 1 .. 15): batch element K of the same timed workload instead -- its angles are row K of
bench.py's data matrix (``torch.rand(16, n_rx, generator=manual_seed(1234)) * 2 pi`` in float32, the reference's vmap
semantics, circuit.py:232-240: H / CNOT shared, Rx angles per sample) -> tests/golden/pin28_s{K}.npz.
"""

import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, to_np  # noqa: E402
from specs import random_spec  # noqa: E402


def main():
    n = int(os.environ.get('PIN_N', 28))
    depth = int(os.environ.get('PIN_DEPTH', 40))
    dq = import_reference()
    torch.set_num_threads(int(os.environ.get('PIN_THREADS', os.cpu_count() or 1)))
    spec = random_spec(n, depth, 1234)
    cir = dq.QubitCircuit(n)
    angles = []
    for method, args, _ in spec:
        if method == 'rx':
            cir.rx(args[0], encode=True)
            angles.append(args[1])
        else:
            getattr(cir, method)(*args)
    for q in range(n):
        cir.observable(q)
    c128 = os.environ.get('PIN_DTYPE', 'c64') == 'c128'      # the complex128 pin: PIN_N=26 PIN_DTYPE=c128
    if c128:
        cir.to(torch.double)
    data = torch.tensor(angles, dtype=torch.float64 if c128 else torch.float32)
    sample = int(sys.argv[sys.argv.index('--sample') + 1]) if '--sample' in sys.argv else 0
    if sample:
        assert 0 < sample < 16 and not c128
        g = torch.Generator().manual_seed(1234)                 # bench.build_circuit, shard 0
        rows = torch.rand(16, len(angles), generator=g, dtype=torch.float32) * 2 * math.pi
        data = rows[sample].clone()
    t0 = time.perf_counter()
    with torch.no_grad():
        state = cir(data)          # QubitCircuit.forward, circuit.py:180-263
        t1 = time.perf_counter()
        print(f'forward {t1 - t0:.1f} s', flush=True)
        ev = cir.expectation()
        print(f'expectation {time.perf_counter() - t1:.1f} s', flush=True)
        flat = state.reshape(-1)
        assert flat.dtype == (torch.complex128 if c128 else torch.complex64) and flat.numel() == 2**n
        idx = torch.randint(0, 2**n, (4096,), generator=torch.Generator().manual_seed(28))
        p = (flat.real.double() ** 2 + flat.imag.double() ** 2)
        out = {
            'nqubit': np.array(n), 'depth': np.array(depth), 'seed': np.array(1234), 'sample': np.array(sample),
            'angles_f32': data.numpy(),
            'indices': idx.numpy(),
            'amplitudes': to_np(flat[idx]),
            'norm2': np.array(p.sum().item()),
            'expectation_z': to_np(ev).reshape(-1),
        }
        # marginal over the five lowest and the five highest wires (reference layout: wire 0 = MSB)
        pt = p.reshape([2] * n)
        out['marginal_wires_0_4'] = pt.reshape(32, -1).sum(-1).numpy()
        out['marginal_wires_last5'] = pt.reshape(-1, 32).sum(0).numpy()
    name = 'pin28.npz' if (n, depth, c128) == (28, 40, False) else f'pin{n}_d{depth}{"_c128" if c128 else ""}.npz'
    if sample:
        name = name[:-4] + f'_s{sample}.npz'
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('norm2', out['norm2'], 'Z0', out['expectation_z'][0], 'wrote', name, flush=True)


if __name__ == '__main__':
    main()
