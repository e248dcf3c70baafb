#!/usr/bin/env python
"""Pins for the workloads no CPU can hold (n = 29 .. 34): made on ONE MI355X by the HIP path itself, on a route that
shares as little as possible with the sharded runs it will judge.

The HIP path is pinned to the real reference up to n = 28 (tests/golden/pin28.npz: 76 minutes of the reference).  Beyond
that the reference cannot run in the build container ("what cannot be oracle-checked"), so the pins of the
sharded workloads -- `bench.py --gpus N` weak series n = 29 / 30 / 31, `--strong` n = 32 / 33 / 34, `--config 4 | 5` with
the explicit cx(0, n-1), cx(n-1, 0) -- come from the one-GPU kernels in their plainest configuration:

* the state is one explicit (1, 2^n) buffer, every pass runs IN PLACE in the canonical qubit order (no permuted stores,
  no second buffer, no known-zero passes, no remaps, no exchange), one-qubit runs are NOT multiplied together;
* first the same route is run at n = 28 and compared with the reference's own pin (printed and stored): the route is
  validated where the reference can still be asked.

Stored per workload (tests/golden/pin_n{n}[_cx].npz, the layout of pin28.npz): 4096 amplitudes at seeded indices, the
squared norm, <Z_q> on every wire.  Run on the GPU box:   python tools/make_pins_large.py --out gpurun_out/pins
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def gate_list(dq, n, depth, seed, extra, device):
    """(primitives in program order, the generator's angles as float32) of the workload -- through the same builder as
    bench.py; the circuit object is only a container of gates (its sharded init state is lazy and never built)."""
    spec = bench.random_circuit_spec(n, depth, seed)
    full = spec + ([('cnot', 0, n - 1), ('cnot', n - 1, 0)] if extra else [])
    cir, data = bench.build_circuit(dq, n, full, None, torch.complex64, device, distributed=True)
    cir.encode(data)
    prims = [p for op in cir.operators for p in op.prims(decompose=True)]
    return prims, data, len(full)


def plain_run(dq, n, prims, device):
    """|0..0> -> final state, in place, canonical order, unmerged."""
    from deepquantum_amd import executor

    keep = dict(executor.CONFIG)
    executor.CONFIG.update(merge_min_amps=None, permute_store=False, zero_state=False)
    try:
        x = torch.zeros(1, 1 << n, dtype=torch.complex64, device=device)
        x[0, 0] = 1
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = executor.run(x, prims, inplace=True)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        assert out.data_ptr() == x.data_ptr()
        stats = {k: v for k, v in executor.LAST_RUN.items() if k != 'plan'}
    finally:
        executor.CONFIG.update(keep)
    return x, dt, stats


def take_pin(n, state, seed_idx):
    from deepquantum_amd import backend

    idx = torch.randint(0, 2**n, (4096,), generator=torch.Generator().manual_seed(seed_idx))
    amp = state[0, idx.to(state.device)].cpu().numpy()
    norm2 = float(backend.expect_pauli(state, 0, 0)[0])
    ez = np.array([float(backend.expect_pauli(state, 0, 1 << (n - 1 - q))[0]) for q in range(n)])
    return idx.numpy(), amp, norm2, ez


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'pins'))
    ap.add_argument('--sizes', type=int, nargs='*', default=[29, 30, 31, 32, 33, 34])
    ap.add_argument('--cx-sizes', type=int, nargs='*', default=[30, 31, 32, 33, 34])
    ap.add_argument('--depth', type=int, default=40)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--skip-validation', action='store_true')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import deepquantum_amd as dq

    device = torch.device('cuda', 0)
    log = {'made_by': 'tools/make_pins_large.py', 'route': 'one explicit state, in-place passes in canonical order, unmerged '
           'gates, no permuted stores / known-zero passes / remaps', 'pins': []}

    if not args.skip_validation:
        # the route against the REAL reference at the largest size it was run at
        ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'pin28.npz'))
        prims, data, _ = gate_list(dq, 28, 40, 1234, False, device)
        assert np.array_equal(data.cpu().numpy(), ref['angles_f32'])
        x, dt, stats = plain_run(dq, 28, prims, device)
        from deepquantum_amd import backend

        amp = x[0, torch.from_numpy(ref['indices']).to(device)].cpu().numpy()
        ez = np.array([float(backend.expect_pauli(x, 0, 1 << (27 - q))[0]) for q in range(28)])
        val = {'n': 28, 'max_amplitude_error': float(np.abs(amp - ref['amplitudes']).max()),
               'max_amplitude': float(np.abs(ref['amplitudes']).max()),
               'max_expectation_z_error': float(np.abs(ez - ref['expectation_z']).max()),
               'norm2': float(backend.expect_pauli(x, 0, 0)[0]), 'norm2_reference': float(ref['norm2']),
               'seconds': dt, 'passes': stats.get('passes')}
        assert val['max_amplitude_error'] < 1e-2 * val['max_amplitude'] and val['max_expectation_z_error'] < 1e-4
        log['validation_against_the_reference_pin_n28'] = val
        print('validation n=28 vs tests/golden/pin28.npz:', json.dumps(val), flush=True)
        del x, prims
        torch.cuda.empty_cache()

    jobs = [(n, False) for n in args.sizes] + [(n, True) for n in args.cx_sizes]
    for n, extra in jobs:
        prims, data, ngates = gate_list(dq, n, args.depth, args.seed, extra, device)
        x, dt, stats = plain_run(dq, n, prims, device)
        idx, amp, norm2, ez = take_pin(n, x, n)
        name = f'pin_n{n}{"_cx" if extra else ""}.npz'
        np.savez_compressed(os.path.join(args.out, name), nqubit=np.array(n), depth=np.array(args.depth),
                            seed=np.array(args.seed), extra_cx=np.array(extra), angles_f32=data.cpu().numpy(),
                            indices=idx, amplitudes=amp, norm2=np.array(norm2), expectation_z=ez.astype(np.float64))
        row = {'file': name, 'n': n, 'gates': ngates, 'extra_cx': extra, 'norm2': norm2, 'Z0': float(ez[0]),
               'max_amplitude': float(np.abs(amp).max()), 'seconds': dt, 'passes': stats.get('passes')}
        log['pins'].append(row)
        print(json.dumps(row), flush=True)
        del x, prims
        torch.cuda.empty_cache()
    json.dump(log, open(os.path.join(args.out, 'pins_log.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
