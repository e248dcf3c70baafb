#!/usr/bin/env python
"""Headline benchmark: gate-applies/s and HBM GB/s of the QubitCircuit statevector hot path.

Workloads (BASELINE.json configs; generator = seeded random H / Rx / CNOT, seed 1234):

* default, N = 1 -- config 3: QubitCircuit(28), depth 40 (1120 gates), complex64, batch = 16 with per-sample Rx
  angles (the ``torch.vmap`` case of the reference), |0...0> start, no_grad forward + <Z0>.
* default, N > 1 (one rank per GPU, torchrun) -- the SAME per-GPU state as N = 1 with the index bits sharded: the
  section-8d weak series n = 28 + log2(N) (29 / 30 / 31 qubits on 2 / 4 / 8 GPUs), batch 16: every rank holds a
  (16, 2^28) shard, global qubits are exchanged by the all-to-all qubit remap over RCCL / xGMI
  (deepquantum_amd/distributed.py).  ``--batch-shard`` measures independent replicas instead (every rank its own 16
  samples of the 28-qubit circuit, no collective in the data path).
* ``--strong`` -- the section-8d pair "31 qubits on one GPU vs 34 qubits on eight": n = 31 + log2(N), batch 1.
* ``--config 4`` / ``--config 5`` -- the sharded configs at full size (n = 30 / 31 qubits PER GPU + log2(N): 32 qubits
  on 4 GPUs, 34 on 8): generator circuit plus the explicit global-control / global-target ``cx``; config 5 adds the
  QAOA ring with the gradient of sum <Z_i Z_j> through the adjoint sweep.  They run on any N (n = 31 on one GPU).

One "step" = one forward pass of the whole circuit over the whole batch (+ the expectation), inputs resident in HBM.
``value`` = gate-applies of all ranks / wall time of the K timed steps (barrier + synchronize on both sides, max over
ranks).  Rank 0 prints ONE JSON line with two extra objects: ``roofline`` for the dominant kernel (the fused pass;
``frac`` is the PHYSICAL fraction of the HBM peak) and ``cpu_baseline`` (the oracle = restatement of the reference's
permute / reshape / matmul path on this host's cores, on a bounded sample of the same workload).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
XGMI_LINK_GBS = 153.0       # per link, 7 links per GPU (task statement); a k-qubit remap keeps 2^k - 1 of them busy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', type=int, default=3, choices=[3, 4, 5], help='BASELINE config (3 = headline)')
    ap.add_argument('--strong', action='store_true', help='n = 31 + log2(N), batch 1 (31 qubits x 1 GPU vs 34 x 8)')
    ap.add_argument('--batch-shard', action='store_true',
                    help='N > 1: independent replicas (every rank its own samples of the 28-qubit circuit)')
    ap.add_argument('--sharded-state', action='store_true', help='(the default for N > 1; kept for old command lines)')
    ap.add_argument('--nqubit', type=int, default=None, help='qubits PER GPU-sized shard (n = nqubit + log2 gpus)')
    ap.add_argument('--depth', type=int, default=40)
    ap.add_argument('--batch', type=int, default=None, help='samples (0 = un-batched 1-D data)')
    ap.add_argument('--dtype', choices=['c64', 'c128'], default='c64')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=25.0)
    ap.add_argument('--no-sweep', action='store_true', help='skip the single-gate sweep over all targets')
    ap.add_argument('--backend', default='nccl', help="process-group backend for N > 1 ('nccl' = RCCL; 'gloo' lets "
                    'several ranks share one GPU for a functional check)')
    # A/B knobs
    ap.add_argument('--min-low', type=int, default=None)
    ap.add_argument('--max-gates', type=int, default=None)
    ap.add_argument('--no-fuse', action='store_true')
    ap.add_argument('--max-far', type=int, default=None, help='scheduler: max gathered bits >= far-bit per pass')
    ap.add_argument('--far-bit', type=int, default=None)
    ap.add_argument('--plan-width', type=int, default=None,
                    help='pass planner beam width (0 = first-come tiles, 1 = greedy; default: library)')
    ap.add_argument('--plan-branch', type=int, default=None, help='pass planner: tiles tried per beam state')
    ap.add_argument('--plan-restarts', type=int, default=None, help='pass planner: beam searches with different seeds')
    ap.add_argument('--no-fused-expectation', action='store_true',
                    help='A/B: <Z0> from a separate read of the final state instead of the registers of the last pass')
    ap.add_argument('--no-fused-sweep', action='store_true',
                    help='A/B (config 5): gate-by-gate reverse sweep of the sharded adjoint instead of fused passes')
    ap.add_argument('--no-compare', action='store_true',
                    help='skip the extra runs (merging off, single-gate sweep): tools/profile.sh uses it so that the '
                         'profiled launches are the timed ones only')
    ap.add_argument('--no-permute-store', action='store_true',
                    help='A/B: in-place passes (every pass gathers its qubits where they canonically live)')
    ap.add_argument('--no-zero-state', action='store_true',
                    help='A/B: the first passes read, compute and write everything although the circuit starts from its '
                         'own |0..0> (executor.CONFIG["zero_state"])')
    ap.add_argument('--no-merge', action='store_true',
                    help='A/B: do not multiply runs of one-qubit gates on the same qubit into one matrix')
    ap.add_argument('--no-free-low', action='store_true', help='A/B: the contiguous low bits keep the same qubits in every pass')
    ap.add_argument('--overlap-groups', type=int, default=None, help='N > 1: sample groups of the overlapped remap')
    ap.add_argument('--virtual-bits', type=int, default=None,
                    help='N > 1, un-batched shards (--strong, --config 4 | 5): top local index bits treated as rank bits of '
                         'a virtual world, so that the rows of a shard overlap exchange and compute (default: chosen per '
                         'circuit by the library\'s dry-run model, distributed.choose_virtual_bits -- never worse than 0 by it)')
    ap.add_argument('--no-local-first-exchange', action='store_true',
                    help='N > 1, A/B: the first exchange behind reset() over the wire (round 5) instead of every rank '
                         'computing rank 0\'s first stretch itself')
    ap.add_argument('--rehearse-loopback', action='store_true',
                    help='with --rehearse-rank: the slices of a sliced exchange are copied send buffer -> receive buffer on the '
                         'exchange stream instead of being left out, so that the bytes of the hidden wire cross this GPU\'s HBM '
                         'beside the passes as they would with peers (measures the contention the model otherwise adds)')
    ap.add_argument('--no-defer-tail', action='store_true',
                    help='N > 1, A/B: run the under-filled last pass of a stretch instead of moving its gates behind the exchange')
    ap.add_argument('--slice-exchange', type=int, default=None,
                    help='N > 1, un-batched shards: bits the last pass in front of an exchange and the first pass behind it are '
                         'sliced by (0 = off; default: 3 under RCCL and in a rehearsal, distributed.CONFIG[\'slice_exchange\'])')
    ap.add_argument('--no-fold-permute', action='store_true',
                    help='N > 1, A/B: the re-labelling before an exchange as a pass of its own')
    ap.add_argument('--traffic-json', default=None, help='file with PMC-measured HBM bytes per launch')
    ap.add_argument('--rehearse-rank', type=int, default=None,
                    help='ONE process, one GPU: run rank R\'s schedule of the --gpus N job (shard, passes, re-labellings, '
                         'streams) with every exchange left out -- the compute half of the multi-GPU step, measured; prints '
                         'its own JSON line (amplitudes are meaningless, timing is data-independent)')
    ap.add_argument('--no-qaoa', action='store_true', help='config 5: the generator circuit only')
    ap.add_argument('--qaoa-nqubit', type=int, default=None,
                    help='config 5: qubits of the QAOA ring (default: the generator circuit\'s n).  The sharded adjoint holds '
                         'eight shard-sized buffers per rank: eight ranks SHARING one 288-GB GPU fit n = 31, not 33')
    ap.add_argument('--functional', action='store_true',
                    help='a functional run (the GPU tests of the full-size sharded configs): no separate set-up step -- the '
                         'first timed step plans and allocates -- so `value` is not a performance figure')
    ap.add_argument('--no-parity', action='store_true', help='skip the pin check after the timed steps')
    return ap.parse_args()


def random_circuit_spec(nqubit, depth, seed=1234):
    """The workload generator of SURVEY section 8(d), stated here so that the measured path imports nothing from
    ``oracle/`` (tests check that the oracle's and the fixtures' generators are this same one): per layer, per
    qubit q: 1/3 H(q), 1/3 Rx(q, U(0, 2 pi)), 1/3 CNOT(q, random other qubit)."""
    import random

    rng = random.Random(seed)
    spec = []
    for _ in range(depth):
        for q in range(nqubit):
            r = rng.random()
            if r < 1 / 3:
                spec.append(('h', q))
            elif r < 2 / 3:
                spec.append(('rx', q, rng.uniform(0, 2 * math.pi)))
            else:
                t = rng.randrange(nqubit - 1)
                t += t >= q
                spec.append(('cnot', q, t))
    return spec


def build_circuit(dq, n, spec, batch, dtype, device, distributed=False, shard=0):
    """The generator's circuit; Rx angles are encoder inputs so each batch sample has its own.  ``shard`` = which
    slice of a batch sharded over ranks this is (its samples get their own angles).  ``batch`` None: 1-D data (the
    reference's un-batched call)."""
    cir = dq.DistributedQubitCircuit(n) if distributed else dq.QubitCircuit(n)
    if distributed:
        # DROP-IN semantics in the timed step: the reference's forward returns the shards in its own qubit order
        # (distributed.py:57-202), so every step restores it; the lazy layout -- an extension: the qubits stay where the
        # last remap put them, <Z0> is taken from the shards as they lie -- is timed as well and reported NEXT to `value`
        # (`value_lazy_layout`), never instead of it
        cir.lazy_layout = False
    angles = []
    for op in spec:
        if op[0] == 'h':
            cir.h(op[1])
        elif op[0] == 'rx':
            cir.rx(op[1], encode=True)
            angles.append(op[2])
        else:
            cir.cnot(op[1], op[2])
    cir.observable(0)
    cir.to(device)
    if dtype == torch.complex128:
        cir.to(torch.double)
    real = torch.float64 if dtype == torch.complex128 else torch.float32
    if batch is None:
        return cir, torch.tensor(angles, dtype=real).to(device)
    g = torch.Generator().manual_seed(1234 + shard)
    data = torch.rand(batch, len(angles), generator=g, dtype=real) * 2 * math.pi
    if shard == 0:
        data[0] = torch.tensor(angles, dtype=real)  # sample 0 = the generator's own angles
    return cir, data.to(device)


def device_copy_bandwidth(device, nbytes=1 << 32, reps=5):
    """Read+write GB/s of torch's device-to-device copy of ``nbytes``.  NOT the yardstick (round 5: the fused pass ran at
    1.07 x this figure -- it measures the copy's implementation, not the memory); reported as `torch_copy_GBs` for the
    record.  The measured ceiling is the pass kernel's own skeleton: `single_gate_sweep(...)['skeleton']`."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    return 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def single_gate_sweep(dq, n, batch, dtype, device, reps=2, only_skeleton=False):
    """Physical read+write GB/s of ONE gate application over a resident (batch, 2^n) state -- the un-fused figure the
    north star's ">= 60 % of the HBM roofline" refers to -- for H on EVERY target bit and for CNOT pairs (near / far,
    control above / below the target).  Returns {'h': {bit: GB/s}, 'cnot': {'c->t': GB/s}}; a CNOT launch is charged
    the whole state (it reads and writes the tiles it visits; whether it may skip a tile depends on where the control
    sits), so its figure is a lower bound of the rate on the bytes it really touched."""
    from deepquantum_amd import backend, fusion

    is128 = dtype == torch.complex128
    h = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5).to(dtype).reshape(-1).to(device)
    x = torch.zeros(batch, 1 << n, dtype=dtype, device=device)
    x[:, 0] = 1
    nbytes = 2 * x.numel() * x.element_size()

    def time_ops(ops, mat):
        steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
        km = fusion.kernel_matrices(steps, ops, mat)
        assert len(steps) == 1
        backend.apply_fused(x, km, 0, steps[0].desc, out=x)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            backend.apply_fused(x, km, 0, steps[0].desc, out=x)
        e1.record()
        torch.cuda.synchronize(device)
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    out = {'h': {}, 'cnot': {}}
    # the kernel's SKELETON -- load the tile, one uncontrolled X on a register bit (64-bit moves, no arithmetic), store
    # in place -- is the measured ceiling of a pass on this box (an in-framework copy figure); best of 3
    skel = [fusion.PrimOp('x', (1,), (), 0, 0)]
    out['skeleton'] = max(time_ops(skel, h) for _ in range(3))
    if only_skeleton:
        return out
    for t in range(n):
        out['h'][t] = time_ops([fusion.PrimOp('gen', (t,), (), 0, 3)], h)
    pairs = [(0, 1), (1, 0), (0, n - 1), (n - 1, 0), (n // 2, n // 2 + 1), (n // 2 + 1, n // 2), (3, n - 2), (n - 2, 3),
             (n - 1, n - 2), (5, 17)]
    pairs = [(c, t) for c, t in dict.fromkeys(pairs) if 0 <= c < n and 0 <= t < n and c != t]
    for c, t in pairs:
        out['cnot'][f'{c}->{t}'] = time_ops([fusion.PrimOp('x', (t,), (c,), 0, 0)], h)
    return out


def _stats(values):
    v = sorted(values)
    return {'min': v[0], 'median': statistics.median(v), 'max': v[-1], 'n': len(v)}


def cpu_baseline(n, dtype, budget_s):
    """Oracle (port of the reference's evolve_state / op_state_control path) on this host's cores, batch element 0:
    the sample SURVEY 8(d) prescribes -- H and Rx on low / mid / high wires, near and far CNOTs in both directions --
    extrapolated to a gate rate (the full circuit is n * depth such gates)."""
    from oracle import statevec_oracle as oracle

    torch.set_num_threads(os.cpu_count() or 1)
    real = torch.float32 if dtype == torch.complex64 else torch.float64
    x = torch.zeros(1, 2**n, dtype=dtype)
    x[0, 0] = 1
    h = oracle.fixed_matrix('h').to(dtype)
    cnot = (torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]]) + 0j).to(dtype)
    rx = oracle.rx_matrix(oracle.theta_tensor(0.7).to(real)).to(dtype)
    lo, mid, hi = 0, n // 2, n - 1
    sample = [('h', lo), ('h', mid), ('h', hi), ('rx', lo + 1), ('rx', mid + 1), ('rx', hi - 1),
              ('cnot', lo, lo + 1), ('cnot', lo, hi), ('cnot', hi, lo), ('cnot', mid, mid + 1), ('cnot', hi, hi - 1)]
    per_gate, done, t0 = [], 0, time.perf_counter()
    with torch.no_grad():
        for op in sample:
            t1 = time.perf_counter()
            if op[0] == 'h':
                x = oracle.apply_gate_wires(x, h, n, [op[1]])
            elif op[0] == 'rx':
                x = oracle.apply_gate_wires(x, rx, n, [op[1]])
            else:
                x = oracle.apply_gate_wires(x, cnot, n, [op[1], op[2]])
            per_gate.append(time.perf_counter() - t1)
            done += 1
            if time.perf_counter() - t0 > budget_s and done >= 8:
                break
    dt = time.perf_counter() - t0
    names = ', '.join(f'{op[0]}{list(op[1:])}' for op in sample[:done])
    return {
        'value': done / dt,
        'unit': 'gate-applies/s',
        'cores': torch.get_num_threads(),
        'kind': 'port',
        'extrapolated': True,
        'seconds_per_gate': {'min': min(per_gate), 'max': max(per_gate)},
        'sample': f'{done} gates on wires low / mid / high of an n={n} state (wire 0 = MSB): {names}; batch element 0 '
                  f'only, {"c64" if dtype == torch.complex64 else "c128"}, {dt:.1f} s wall; the rate extrapolates to the '
                  f'n * depth gates of the circuit',
    }


def find_pin(n, depth, seed, dtype, extra_cx):
    """(npz, description) of the pin of this workload's sample 0, or (None, why not).  n = 28 without the cx pair: made by
    the REAL reference (tests/golden/make_golden_pin28.py).  n = 29 .. 34: made on one MI355X by the plain one-GPU route
    of the HIP path -- in place, canonical order, unmerged gates -- which that same script validates against the
    reference's n = 28 pin first (tools/make_pins_large.py, profiles/r05/pins_log.json)."""
    import numpy as np

    gold = os.environ.get('DQ_PIN_DIR') or os.path.join(ROOT, 'tests', 'golden')      # (the harness's own tests: a temp dir)
    if not (depth == 40 and seed == 1234 and dtype == torch.complex64):
        return None, 'no pin for this workload (pins are depth 40, seed 1234, complex64)'
    if n == 28 and not extra_cx:
        path, what = os.path.join(gold, 'pin28.npz'), 'tests/golden/pin28.npz (real reference, batch element 0)'
    else:
        name = f'pin_n{n}{"_cx" if extra_cx else ""}.npz'
        path, what = os.path.join(gold, name), (f'tests/golden/{name} (one-GPU HIP path, plain route validated against the '
                                                f'reference pin at n = 28: tools/make_pins_large.py)')
    if not os.path.exists(path):
        return None, f'no pin for n = {n}{" + cx pair" if extra_cx else ""}'
    return np.load(path), what


def pin_verdict(amp, ref_amp, norm2, ref_norm2, ez, ref_ez):
    """The parity criterion of every bench line.  Amplitudes are the strong check and it is RELATIVE: the largest
    deviation against the largest pinned amplitude, and the l2 deviation against the l2 norm of the pinned sample (an
    all-zero or scrambled result fails both however small the amplitudes are).  <Z_q> and the squared norm at the north
    star's absolute 1e-4."""
    import numpy as np

    amp_err = float(np.abs(amp - ref_amp).max())
    rel_max = amp_err / float(np.abs(ref_amp).max())
    rel_l2 = float(np.linalg.norm(amp - ref_amp) / np.linalg.norm(ref_amp))
    z_err = float(np.abs(np.asarray(ez) - np.asarray(ref_ez)).max())
    ok = rel_max < 1e-4 and rel_l2 < 1e-4 and amp_err < 1e-4 and z_err < 1e-4 and abs(norm2 - ref_norm2) < 1e-4
    return ok, {'amplitudes_checked': int(len(ref_amp)), 'max_amplitude_error': amp_err,
                'max_amplitude_error_relative_to_largest_amplitude': rel_max, 'l2_error_relative': rel_l2,
                'max_expectation_z_error': z_err, 'norm2': norm2, 'norm2_reference': ref_norm2,
                'tolerance': {'amplitudes_relative': 1e-4, 'expectation_z': 1e-4, 'norm2': 1e-4}}


def check_pin(cir, n, depth, seed, dtype, extra_cx=False):
    """Parity of the TIMED workload on one GPU: sample 0 against its pin (`find_pin`)."""
    from deepquantum_amd import backend

    pin, what = find_pin(n, depth, seed, dtype, extra_cx)
    if pin is None:
        return False, {'reason': what}
    state = cir.state.reshape(-1, 1 << n)[:1].contiguous()        # sample 0 = the generator's own angles
    idx = torch.from_numpy(pin['indices']).to(state.device)
    amp = state[0, idx].cpu().numpy()
    norm2 = float(backend.expect_pauli(state, 0, 0)[0])
    ez = [float(backend.expect_pauli(state, 0, 1 << (n - 1 - q))[0]) for q in range(n)]
    ok, rep = pin_verdict(amp, pin['amplitudes'], norm2, float(pin['norm2']), ez, pin['expectation_z'])
    rep['source'] = what
    # further samples of the timed batch pinned to the real reference (make_golden_pin28.py --sample K: row K of the same
    # data matrix through the reference's un-batched forward)
    import glob

    import numpy as np
    gold = os.environ.get('DQ_PIN_DIR') or os.path.join(ROOT, 'tests', 'golden')
    full = cir.state.reshape(-1, 1 << n)
    for path in sorted(glob.glob(os.path.join(gold, f'pin{n}_s*.npz'))) if not extra_cx else []:
        pk = np.load(path)
        k = int(pk['sample'])
        if k >= full.shape[0] or int(pk['depth']) != depth:
            continue
        row = full[k:k + 1].contiguous()
        ampk = row[0, torch.from_numpy(pk['indices']).to(row.device)].cpu().numpy()
        ezk = [float(backend.expect_pauli(row, 0, 1 << (n - 1 - q))[0]) for q in range(n)]
        okk, repk = pin_verdict(ampk, pk['amplitudes'], float(backend.expect_pauli(row, 0, 0)[0]), float(pk['norm2']),
                                ezk, pk['expectation_z'])
        repk['source'] = f'tests/golden/{os.path.basename(path)} (real reference, batch element {k})'
        rep.setdefault('more_samples', {})[str(k)] = repk
        ok = ok and okk
    return ok, rep


def check_pin_sharded(cir, n, depth, seed, dtype, extra_cx=False):
    """The same for the index-bit-sharded state (collective: every rank calls it).  The shards are read in the
    reference's layout (``state.amps``: rank r owns global indices [r 2^L, (r + 1) 2^L)); every rank compares the pinned
    amplitudes that fall into its shard, <Z_q> and the norm are summed over the ranks."""
    import numpy as np
    import torch.distributed as dist

    from deepquantum_amd import backend

    pin, what = find_pin(n, depth, seed, dtype, extra_cx)
    if pin is None:
        return False, {'reason': what}
    st = cir.state
    amps = st.amps                                  # canonical order (an exchange if the layout was lazy)
    L, rank = st.log_num_amps_per_node, st.rank
    view = amps.reshape(-1, 1 << L)[:1].contiguous()      # sample 0
    idx = pin['indices']
    mine = (idx >> L) == rank
    loc = torch.from_numpy(idx[mine] & ((1 << L) - 1)).to(view.device)
    got = np.zeros(len(idx), dtype=np.complex128)
    got[mine] = view[0, loc].cpu().numpy()
    vals = torch.zeros(n + 1, dtype=torch.float64, device=view.device)
    vals[n] = backend.expect_pauli(view, 0, 0)[0]
    for q in range(n):
        p_ = n - 1 - q
        if p_ < L:
            vals[q] = backend.expect_pauli(view, 0, 1 << p_)[0]
        else:
            vals[q] = vals[n] * (-1.0 if (rank >> (p_ - L)) & 1 else 1.0)
    buf = torch.from_numpy(np.concatenate([got.real, got.imag, [float(mine.sum())]])).to(view.device)
    if dist.is_initialized():
        dist.all_reduce(vals)
        dist.all_reduce(buf)
    buf = buf.cpu().numpy()
    k = len(idx)
    amp = buf[:k] + 1j * buf[k:2 * k]
    vals = vals.cpu().numpy()
    ok, rep = pin_verdict(amp, pin['amplitudes'], float(vals[n]), float(pin['norm2']), vals[:n], pin['expectation_z'])
    ok = ok and int(round(buf[2 * k])) == k
    rep['source'] = what
    rep['amplitudes_checked_per_rank_sum'] = int(round(buf[2 * k]))
    return ok, rep


def qaoa_ring(dq, n, device, distributed):
    """Config 5's second half (examples/qaoa.py:21-64 on a ring, one step): hlayer; per edge cnot . rz . cnot; rx layer;
    one <Z_i Z_j> observable per edge."""
    cir = dq.DistributedQubitCircuit(n) if distributed else dq.QubitCircuit(n)
    pairs = [(i, (i + 1) % n) for i in range(n)]
    cir.hlayer()
    for i, j in pairs:
        cir.cnot(i, j)
        cir.rz(j, encode=True)
        cir.cnot(i, j)
    for i in range(n):
        cir.rx(i, encode=True)
    for i, j in pairs:
        cir.observable([i, j])
    return cir.to(device), pairs


def self_launch(args) -> int:
    """``--gpus N`` (N > 1) outside a torchrun environment: start the N ranks here -- one process per GPU, the same
    command line -- instead of silently measuring one rank.  RCCL needs one GPU per rank (exit 2 if the box has
    fewer); with ``--backend gloo`` the ranks may share GPUs (device = LOCAL_RANK modulo the device count): the
    functional check of the sharded paths on a one-GPU box.  Returns the exit code of the job."""
    import socket
    import subprocess

    n = args.gpus
    have = torch.cuda.device_count()
    if args.backend == 'nccl' and have < n:
        print(f'bench.py: --gpus {n} over RCCL needs {n} GPUs, this box has {have}; nothing was measured '
              f'(--backend gloo lets the ranks share a GPU for a functional check)', file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable] + list(getattr(sys, 'orig_argv', [sys.executable] + sys.argv)[1:])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


class _Clock:
    """A HIP event on the current stream -- or, on a box without a GPU (the CPU tests of this harness; the product
    itself refuses CPU tensors), the host clock."""

    def __init__(self, device):
        self.ev = torch.cuda.Event(enable_timing=True) if device.type == 'cuda' else None
        self.t = None

    def record(self):
        if self.ev is not None:
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def elapsed_time(self, other) -> float:
        return self.ev.elapsed_time(other.ev) if self.ev is not None else (other.t - self.t) * 1e3


def rehearsal_line(dq, args, cir, n, per_gpu, nbatch, amp_bytes, ngates, elapsed, step_ms, remap_rows, events, setup_s, plan_s):
    """What `--rehearse-rank R` prints: the COMPUTE half of rank R's step of the `--gpus N` job, measured on this one GPU
    (every exchange left out), next to the modelled wire time of the exchanges it would have issued."""
    D = dq.distributed
    st = dict(D.LAST_RUN)
    W, R = args.gpus, args.rehearse_rank
    shard_bytes = (1 << per_gpu) * amp_bytes * nbatch
    kernel_ms = [ev[0].elapsed_time(ev[1]) for ev in events]
    kernel_bytes = [ev[3] for ev in events]
    per_step = len(kernel_ms) / args.steps
    vb = st.get('virtual_bits', 0)
    wire = []
    # one pass over the whole shard on this rank (read + write at the rate its fused launches reached)
    full_pass_ms = (2 * shard_bytes / (sum(kernel_bytes) / (sum(kernel_ms) * 1e-3)) * 1e3) if kernel_ms else None
    contention_ms = 0.0
    for row in remap_rows or []:
        k = row['qubits_exchanged']
        # a k-qubit remap: 2^k - 1 peers, one chunk of shard / 2^k to each over its own link, both directions at once
        t_ms = shard_bytes / (1 << k) / (XGMI_LINK_GBS * 1e9) * 1e3
        exposed = t_ms / (1 << vb) if vb else t_ms
        groups = max(1, row['samples'] // args.steps)
        if not vb and groups > 1:
            # a batched shard moves through a remap in `groups` sample groups on their own streams: a group's amplitudes are on
            # the links while the next group computes, and a group starts its next stretch as soon as ITS exchange is back --
            # what can stay exposed is the last group's share
            exposed = t_ms / groups
        sl, sf, nsl = row.get('slices_last') or 0, row.get('slices_first') or 0, row.get('slices') or 0
        if nsl and full_pass_ms:
            # a sliced exchange: the last pass in `sl` launches, the first pass behind it in `sf`; the wire of
            # slice j runs while slice j + 1 computes / while slice j - 1 is already computed on: what stays exposed is the
            # wire less the (1 - 1/S) of either pass it runs beside, and never less than one protocol slice's share
            hidden = full_pass_ms * ((1 - 1 / sl) if sl > 1 else 0.0) + full_pass_ms * ((1 - 1 / sf) if sf > 1 else 0.0)
            exposed = max(t_ms - hidden, t_ms / nsl)
            # ... and the hidden part is not free: its bytes cross this GPU's HBM (read to send, written on arrival) while
            # the passes want all of it
            if D.CONFIG['elide_exchange'] != 'loopback':     # (with --rehearse-loopback it is IN the measured compute)
                contention_ms += (t_ms - exposed) / t_ms * 2 * (1 - 0.5 ** k) * shard_bytes / (sum(kernel_bytes) / (sum(kernel_ms) * 1e-3)) * 1e3
        wire.append({'remap': row['remap'], 'qubits_exchanged': k, 'links': (1 << k) - 1,
                     'wire_ms_at_peak_link_rate': t_ms,
                     # with 2^v rows of the shard in flight one after the other only the first row's share is not hidden
                     'exposed_ms_model': exposed,
                     'protocol_slices': nsl or None, 'launches_of_the_last_pass': sl or None, 'launches_of_the_first_pass_behind': sf or None,
                     'local_passes_ms_median_per_row_group': row['local_passes_ms_median'], 'row_groups': row['samples'] // args.steps})
    ms = elapsed / args.steps * 1e3
    return {
        'rehearsal': f'rank {R} of {W}: its shard, schedule, passes, re-labellings and streams on ONE GPU, every exchange and '
                     f'all-reduce left out (amplitudes meaningless; kernel timing is data-independent)',
        'workload': f'QubitCircuit({n}) depth {args.depth} ({ngates} gates), {"c64" if amp_bytes == 8 else "c128"}, batch {nbatch}, '
                    f'{per_gpu} local qubits per rank' + (' (--strong)' if args.strong else f' (config {args.config})'),
        'rank': R, 'world': W, 'steps': args.steps, 'warmup': args.warmup,
        'compute_ms_per_step': ms,
        'compute_ms_per_step_hip_events_median': statistics.median(step_ms) if step_ms else None,
        'fused_launches_per_step': per_step,
        'fused_launch_ms_sum_per_step': sum(kernel_ms) / args.steps if kernel_ms else None,
        'fused_launch_GBs': (sum(kernel_bytes) / (sum(kernel_ms) * 1e-3) / 1e9) if kernel_ms else None,
        'virtual_rank_bits': vb,
        'virtual_bits_model': (D.virtual_bits_table([p_ for op_ in cir.operators for p_ in op_.prims(decompose=True)], n, per_gpu,
                                                    fresh=True, restore=False) if nbatch == 1 else None),
        'schedule': {k_: st[k_] for k_ in ('remaps', 'virtual_remaps', 'folded_permutes', 'permute_passes', 'local_flushes',
                                           'zero_shard_stretches', 'known_zero_stretches', 'local_first_exchanges',
                                           'sliced_remaps', 'slice_launches_last', 'slice_launches_first', 'zero_fills',
                                           'deferred_tails', 'deferred_gates')},
        'wire_model': {'peak_GBs_per_link': XGMI_LINK_GBS, 'remaps': wire,
                       'wire_ms_per_step_all_exposed': sum(w['wire_ms_at_peak_link_rate'] for w in wire),
                       'wire_ms_per_step_exposed_model': sum(w['exposed_ms_model'] for w in wire)},
        'modelled_step_ms': ms + sum(w['exposed_ms_model'] for w in wire),
        'full_pass_ms_on_this_shard': full_pass_ms,
        'hbm_contention_ms_of_the_hidden_wire_model': contention_ms,
        'modelled_step_ms_with_hbm_contention': ms + sum(w['exposed_ms_model'] for w in wire) + contention_ms,
        'slice_exchange_bits': D.slice_bits_wanted(cir.init_state) if getattr(cir, 'init_state', None) is not None else None,
        'loopback_copies_for_the_hidden_wire': D.CONFIG['elide_exchange'] == 'loopback',
        'plan_seconds': plan_s, 'first_step_seconds': setup_s,
    }


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and args.rehearse_rank is None:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.rehearse_rank is not None:
        world, rank, local_rank = 1, 0, 0
    elif world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    import deepquantum_amd as dq

    dtype = torch.complex64 if args.dtype == 'c64' else torch.complex128
    amp_bytes = 8 if dtype == torch.complex64 else 16
    multi = world > 1
    rehearse = args.rehearse_rank is not None
    distributed = (multi and not args.batch_shard) or rehearse      # index-bit-sharded state; otherwise the batch is sharded
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        if args.backend == 'gloo':
            local_rank %= torch.cuda.device_count()      # (ranks may share a GPU: functional check)
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    else:
        device = torch.device('cpu')
    if multi:
        dq.setup_distributed(args.backend)
    if rehearse:
        assert args.gpus > 1 and 0 <= args.rehearse_rank < args.gpus
        dq.DistributedQubitState.REHEARSE = (args.gpus, args.rehearse_rank)
        dq.distributed.CONFIG['elide_exchange'] = 'loopback' if args.rehearse_loopback else True
    nshards = args.gpus if rehearse else world        # ranks the index bits are sharded over

    # ---- which workload -------------------------------------------------------------------------------------
    per_gpu = {3: 28, 4: 30, 5: 31}[args.config]
    batch = 16 if args.config == 3 else None
    if args.strong:
        per_gpu, batch = 31, None
    if args.nqubit is not None:
        per_gpu = args.nqubit
    if args.batch is not None:
        batch = args.batch if args.batch > 0 else None
    n = per_gpu + (int(math.log2(nshards)) if distributed else 0)
    nbatch = batch or 1

    if args.min_low is not None:
        dq.executor.CONFIG['min_low_c64' if dtype == torch.complex64 else 'min_low_c128'] = args.min_low
    if args.max_gates is not None:
        dq.executor.CONFIG['max_gates'] = args.max_gates
    if args.no_fuse:
        dq.executor.CONFIG['fuse'] = False
    dq.executor.CONFIG['max_far'] = args.max_far
    dq.executor.CONFIG['far_bit'] = args.far_bit
    dq.executor.CONFIG['plan_width'] = args.plan_width
    dq.executor.CONFIG['plan_branch'] = args.plan_branch
    dq.executor.CONFIG['plan_restarts'] = args.plan_restarts
    if args.no_fused_expectation:
        dq.executor.CONFIG['fused_expectation'] = False
    if args.no_fused_sweep:
        dq.executor.CONFIG['fused_sweep'] = False
    if args.no_zero_state:
        dq.executor.CONFIG['zero_state'] = False
    if args.no_merge:
        dq.executor.CONFIG['merge_min_amps'] = None
    if args.no_permute_store:
        dq.executor.CONFIG['permute_store'] = False
    if args.no_free_low:
        dq.executor.CONFIG['free_low'] = False
    if args.overlap_groups is not None:
        dq.distributed.CONFIG['overlap_groups'] = args.overlap_groups
    if distributed and args.virtual_bits is not None:
        # an un-batched shard has no samples to overlap its exchanges with: its rows can take their place.  Default (None):
        # the library chooses v per circuit from its dry-run model (a rehearsal runs what the RCCL job would choose)
        dq.distributed.CONFIG['virtual_bits'] = args.virtual_bits
    if args.no_local_first_exchange:
        dq.distributed.CONFIG['first_exchange_local'] = False
    if args.slice_exchange is not None:
        dq.distributed.CONFIG['slice_exchange'] = args.slice_exchange
    if args.no_defer_tail:
        dq.distributed.CONFIG['defer_tail'] = 0
    if args.no_fold_permute:
        dq.distributed.CONFIG['fold_permute'] = False

    # What the FIRST step of a fresh process pays before anything is timed (round 5's driver run: 11.7 s on a fresh box for
    # a 0.2-s step): taken apart here, each piece forced and timed on its own -- the HIP runtime + torch's device set-up,
    # this library's code object (2 MB, loaded at the first launch of one of its kernels), the first touch of the two
    # state buffers (afterwards torch's caching allocator hands the same blocks to the step) -- and the planner below.
    cold = None
    if have_gpu and not args.functional and args.rehearse_rank is None and not multi:
        cold = {}
        t_c = time.perf_counter()
        torch.zeros(1, device=device)
        torch.cuda.synchronize(device)
        cold['hip_runtime_and_torch_device_init_s'] = time.perf_counter() - t_c
        t_c = time.perf_counter()
        probe = torch.zeros(1, 4, dtype=dtype, device=device)
        probe[0, 0] = 1
        dq.backend.apply_gate(probe, torch.eye(2, dtype=dtype, device=device).reshape(1, 2, 2), [0], [], out=probe)
        torch.cuda.synchronize(device)
        cold['library_load_and_first_kernel_s'] = time.perf_counter() - t_c
        t_c = time.perf_counter()
        touch = [torch.empty(nbatch << n, dtype=dtype, device=device).zero_() for _ in range(2)]
        torch.cuda.synchronize(device)
        cold['allocate_and_first_touch_of_two_state_buffers_s'] = time.perf_counter() - t_c
        cold['state_buffer_GiB_each'] = (nbatch << n) * amp_bytes / 2**30
        del touch, probe

    spec = random_circuit_spec(n, args.depth, args.seed)
    extra = []
    if args.config in (4, 5):       # global control / local target, and local control / GLOBAL target
        extra = [('cnot', 0, n - 1), ('cnot', n - 1, 0)]
    full_spec = spec + extra
    ngates = len(full_spec)
    cir, data = build_circuit(dq, n, full_spec, batch, dtype, device, distributed,
                              rank if multi and not distributed else 0)
    # algorithmic bytes per gate: 2 * 2^(n - nc) * sizeof(amp) per batch sample
    alg_bytes = sum(2 * (2 ** (n - (1 if op[0] == 'cnot' else 0))) * amp_bytes for op in full_spec) * nbatch

    if rehearse:
        cir.lazy_layout = True      # (the compute of the step proper: the restore is one or two more exchanges + one pass)
    prof = dq.executor.PROFILE

    def step():
        with torch.no_grad():
            cir(data)      # N > 1: (batch, 2^L) shards on every rank, one exchange schedule for the whole batch
            if rehearse:   # (<Z0> comes out of the last pass + one all-reduce of a few numbers: nothing to rehearse)
                return None
            return cir.expectation()

    def sync():
        if have_gpu:
            torch.cuda.synchronize(device)
        if multi:
            torch.distributed.barrier()
            if have_gpu:
                torch.cuda.synchronize(device)

    t_setup = time.perf_counter()
    if not args.functional:
        step()          # setup, not warm-up: pass plans (host work once per circuit structure), second state buffer
    sync()
    setup_s = time.perf_counter() - t_setup
    plan_s = dq.executor.PLAN_STATS['seconds']
    for _ in range(args.warmup):
        step()
    sync()
    prof['enabled'] = True
    prof['events'].clear()
    if distributed:            # per-exchange HIP events on the streams of the sample groups; collectives issued
        dq.distributed.TIMING['enabled'] = True
        dq.distributed.TIMING['remaps'].clear()
        for k_ in dq.communication.COMM_STATS:
            dq.communication.COMM_STATS[k_] = 0
    step_events = []
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        e0, e1 = _Clock(device), _Clock(device)
        e0.record()
        out = step()
        e1.record()
        step_events.append((e0, e1))
    sync()
    elapsed = time.perf_counter() - t0
    prof['enabled'] = False
    remap_rows, comm_stats = None, None
    if distributed:
        dq.distributed.TIMING['enabled'] = False
        comm_stats = {k_: v / args.steps for k_, v in dq.communication.COMM_STATS.items()}
        by_remap = {}
        for row in dq.distributed.remap_timings():
            by_remap.setdefault(row['remap'], []).append(row)
        dq.distributed.TIMING['remaps'].clear()
        remap_rows = []
        for r_ in sorted(by_remap):
            rows_ = by_remap[r_]
            med = lambda key: (sorted(x[key] for x in rows_ if key in x) or [None])[len([x for x in rows_ if key in x]) // 2]   # noqa: E731
            remap_rows.append({'remap': r_, 'qubits_exchanged': rows_[0]['k'], 'bytes_each_way_per_group': rows_[0]['bytes'],
                               'local_passes_ms_median': med('local_ms'), 'issue_to_wait_passed_ms_median': med('wire_ms'),
                               'samples': len(rows_), 'slices': rows_[0].get('slices'),
                               'slices_last': med('slices_last'), 'slices_first': med('slices_first')})
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    step_ms = [a.elapsed_time(b) for a, b in step_events]      # HIP events around every step (this rank)
    if rehearse:
        print(json.dumps(rehearsal_line(dq, args, cir, n, per_gpu, nbatch, amp_bytes, ngates, elapsed, step_ms, remap_rows,
                                        prof['events'], setup_s, plan_s)))
        return
    # ... and the same step with the lazy layout (no restore of the reference's shard order at the end of the forward: what
    # a training or benchmark step that only takes <Z..Z> from the state needs), a few steps, max over ranks
    lazy_ms = None
    if distributed and not rehearse and not args.functional:     # (a functional run checks the drop-in step only)
        cir.lazy_layout = True
        step()
        sync()
        t_l = time.perf_counter()
        nl = max(1, min(3, args.steps))
        for _ in range(nl):
            step()
        sync()
        lazy_ms = (time.perf_counter() - t_l) / nl * 1e3
        if multi:
            t = torch.tensor([lazy_ms], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            lazy_ms = t.item()
        cir.lazy_layout = False
        step()              # (the state the parity check reads below is the drop-in one)
        sync()

    # dominant kernel: the fused pass -- durations from HIP events recorded on the stream it is launched on, and the
    # bytes each launch physically moves (one read + one write of the rows it works on)
    kernel_ms = [ev[0].elapsed_time(ev[1]) for ev in prof['events']]
    kernel_bytes = [ev[3] for ev in prof['events']]
    launches = len(kernel_ms)
    shard_bytes = (2**n >> (int(math.log2(nshards)) if distributed else 0)) * amp_bytes * nbatch
    avg_ms = (sum(kernel_ms) / launches) if launches else float('nan')
    # the launches that move the whole state both ways (the first passes of a circuit started from |0..0> do not:
    # executor.CONFIG['zero_state'])
    full_ms = [t for t, b_ in zip(kernel_ms, kernel_bytes) if b_ == max(kernel_bytes)] if launches else []
    physical = (sum(kernel_bytes) / (sum(kernel_ms) * 1e-3) / 1e9) if launches else 0.0
    alg_per_launch = (alg_bytes / (nshards if distributed else 1) * args.steps / launches) if launches else 0.0
    effective = alg_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    # HBM bytes per launch from the PMC counters: measured by separate rocprofv3 --pmc runs of this same command
    # (tools/profile.sh; summaries committed under profiles/), NOT during this run -- reported with its source, and
    # only for the workload it was collected on
    traffic, traffic_src = None, None
    tj = args.traffic_json
    if tj is None and not distributed:   # (batch-sharded ranks run the single-GPU workload: same traffic)
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'traffic_n{n}_b{nbatch}_{args.dtype}.json')))
        tj = cands[-1] if cands else None
    if tj and os.path.exists(tj):
        traffic = json.load(open(tj)).get('hbm_bytes_per_launch')
        traffic_src = (os.path.relpath(tj, ROOT) + ': rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command by '
                       'the builder (tools/profile.sh), not measured during this run')
    stats = {k: v for k, v in dq.executor.LAST_RUN.items() if k != 'plan'}
    dstats = dict(dq.distributed.LAST_RUN) if distributed else None
    z0 = float(out.reshape(-1)[0]) if out is not None else None

    # parity of the timed workload: sample 0 against its pin -- one GPU: the real reference's (n = 28); sharded: every
    # rank checks the pinned amplitudes of ITS shard (a collective: all ranks take part), relative criterion
    parity_ok, parity = None, None
    if not args.no_parity:
        if distributed:
            parity_ok, parity = check_pin_sharded(cir, n, args.depth, args.seed, dtype, bool(extra))
        elif rank == 0 and not multi:
            parity_ok, parity = check_pin(cir, n, args.depth, args.seed, dtype, bool(extra))
    norm2 = None
    if distributed:
        from deepquantum_amd.distributed import inner_product_dist

        norm2 = inner_product_dist(cir.state, cir.state).real.reshape(-1)[0].item()

    extras = rank == 0 and not multi and not args.no_compare
    # the same circuit with every gate applied on its own (no products of one-qubit runs), for comparison: N = 1 only
    unmerged_ms = None
    if extras and not args.no_merge and dq.executor.CONFIG['merge_min_amps'] is not None:
        keep = dq.executor.CONFIG['merge_min_amps']
        dq.executor.CONFIG['merge_min_amps'] = None
        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        sync()
        unmerged_ms = (time.perf_counter() - t0) / 2 * 1e3
        dq.executor.CONFIG['merge_min_amps'] = keep
    # the same step with every pass moving the whole state (the first passes behind |0..0> read, compute and write what is
    # known to be zero, as the reference does): the A/B figure next to `ms_per_step`
    whole_state_ms = None
    if extras and not distributed and dq.executor.CONFIG['zero_state']:
        dq.executor.CONFIG['zero_state'] = False
        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        sync()
        whole_state_ms = (time.perf_counter() - t0) / 3 * 1e3
        dq.executor.CONFIG['zero_state'] = True
    copy_gbs = device_copy_bandwidth(device) if rank == 0 and have_gpu else None
    sweep = None
    if extras and not args.no_sweep and n >= 13:
        out = None
        cir.state = None
        if have_gpu:
            torch.cuda.empty_cache()
        sweep = single_gate_sweep(dq, n, nbatch, dtype, device) if have_gpu else None
    elif extras and n >= 13 and have_gpu:
        out = None
        cir.state = None
        torch.cuda.empty_cache()
        sweep = single_gate_sweep(dq, n, nbatch, dtype, device, only_skeleton=True)

    qaoa = None
    if args.config == 5 and not args.no_qaoa:   # QAOA ring, one step: gradient of sum <Z_i Z_j> w.r.t. (gamma, beta)
        cir.state = None
        cir.init_state = None          # (the generator circuit's shard and receive buffer: the ring needs the room)
        out = None
        if have_gpu:
            torch.cuda.empty_cache()
        nq = args.qaoa_nqubit or n
        qc, pairs = qaoa_ring(dq, nq, device, distributed)
        gamma = torch.tensor(0.1, device=device, requires_grad=True)
        beta = torch.tensor(1.0, device=device, requires_grad=True)
        params = torch.cat([(2 * gamma).repeat(len(pairs)), (2 * beta).repeat(nq)])
        sync()
        t0 = time.perf_counter()
        qc(params)
        cost = qc.expectation().sum()
        cost.backward()
        sync()
        # known answer (one QAOA layer on a triangle-free 2-regular graph, n >= 5): every edge has
        # <Z_i Z_j> = sin(4 beta) sin(4 gamma) / 2 with exp(-i gamma ZZ) per edge and exp(-i beta X) per qubit
        g_, b_ = 0.1, 1.0
        exact = {'cost': nq * 0.5 * math.sin(4 * b_) * math.sin(4 * g_),
                 'dcost_dgamma': nq * 2.0 * math.sin(4 * b_) * math.cos(4 * g_),
                 'dcost_dbeta': nq * 2.0 * math.cos(4 * b_) * math.sin(4 * g_)}
        qaoa = {'nqubit': nq, 'edges': len(pairs), 'gates': len(qc.operators), 'cost': float(cost.detach()), 'dcost_dgamma': float(gamma.grad),
                'dcost_dbeta': float(beta.grad), 'seconds_forward_backward': time.perf_counter() - t0,
                'fused_reverse_sweep': bool(distributed and dq.adjoint.LAST_SWEEP.get('fused')),
                'closed_form': exact}
        qaoa['max_relative_error_vs_closed_form'] = max(abs(qaoa[k_] - v_) / abs(v_) for k_, v_ in exact.items())
        qaoa['matches_closed_form'] = qaoa['max_relative_error_vs_closed_form'] < 1e-3
        if parity_ok is not None:
            parity_ok = bool(parity_ok) and qaoa['matches_closed_form']
        # what this run is against the STATED size of BASELINE config 5 -- QubitCircuit(34) on eight MI355X, 31 local qubits:
        # 16 + 16 GiB per rank for the forward, eight shard-sized buffers per rank for the sharded adjoint (128 GiB) -- which
        # needs one GPU per rank; ranks that share a GPU (the full-size test of the one-GPU box) stop at n = 33 / n = 31
        qaoa['stated_size'] = {'forward_nqubit': 34, 'qaoa_nqubit': 34, 'ranks': 8,
                               'this_run': {'forward_nqubit': n, 'qaoa_nqubit': nq, 'ranks': world},
                               'at_stated_size': bool(n == 34 and nq == 34 and world == 8),
                               'pin_for_the_stated_size': 'tests/golden/pin_n34_cx.npz (consumed by this harness when n = 34)'}

    if rank == 0:
        total_gate_applies = ngates * nbatch * args.steps * (world if multi and not distributed else 1)
        value = total_gate_applies / elapsed
        line = {
            'metric': 'gate-applies/sec, 28q random circuit depth 40 (HBM GB/s in roofline)',
            'value': value,
            'unit': 'gate-applies/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'c64' if dtype == torch.complex64 else 'c128',
            'data': 'synthetic',
            **({'functional_run_not_a_performance_figure': True} if args.functional else {}),
            'parity_checked': bool(parity_ok) if parity_ok is not None else False,
            'config': {
                'workload': f'QubitCircuit({n}) random H/Rx/CNOT depth {args.depth} ({ngates} gates, seed {args.seed}), '
                            f'{"complex64" if dtype == torch.complex64 else "complex128"}, batch={nbatch}'
                            + (' (per-sample Rx angles)' if batch else ' (un-batched)') + ', |0..0> start, no_grad forward + <Z0>'
                            + (f', index bits sharded over {world} GPUs ({per_gpu} local qubits each), shards restored to the '
                               f'reference\'s qubit order at the end of every forward (drop-in; lazy layout: value_lazy_layout)'
                               if distributed else '')
                            + (f'; {world} ranks x {nbatch} samples (global batch {world * nbatch})'
                               if multi and not distributed else '')
                            + ('; + cx(0, n-1), cx(n-1, 0)' if extra else ''),
                'baseline_config': args.config,
                'nqubit': n,
                'depth': args.depth,
                'batch': nbatch,
                'parallelism': (f'state-shard x{world}: index-bit partition, k-qubit all-to-all remap over RCCL, re-labelling '
                                f'folded into the last local pass, {dstats.get("groups")} sample groups overlapped'
                                if distributed else
                                f'batch-shard x{world} (independent samples, no collective in the data path)' if multi
                                else 'single GPU'),
                'lazy_layout': False,
                'ms_restore_canonical_layout': (elapsed / args.steps * 1e3 - lazy_ms) if lazy_ms is not None else None,
                'fused_passes_per_step': stats.get('passes') if not distributed else launches / args.steps,
                'lds_round_trips_per_step': stats.get('transposes') if not distributed else None,
                # passes that skip what is still known to be zero behind the circuit's own |0..0> (the first one touches one
                # tile per sample, the last of them is write-only); their launches count in `roofline` with the bytes
                # they really move
                'zero_state_passes_per_step': stats.get('zero_passes') if not distributed else None,
                # 2x2 matrices the kernel applies per sample after runs of one-qubit gates on the same qubit were
                # multiplied together (executor.merge_one_qubit_runs; `--no-merge` applies all `ngates` one by one);
                # `value` counts the circuit's gates, `unmerged_ms_per_step` times them one by one
                'kernel_gates_per_step': stats.get('gates') if not distributed else None,
                'unmerged_ms_per_step': unmerged_ms,
                # ... and with every pass moving the whole state (executor.CONFIG['zero_state'] off; merged gates)
                'ms_per_step_every_pass_moves_the_whole_state': whole_state_ms,
                # what a ONE-SHOT run pays on top of a steady-state step: the pass planner (host, once per circuit
                # structure, cached afterwards) and the whole first step including it and the allocations
                'plan_seconds': plan_s,
                'first_step_seconds': setup_s,
                # (the pieces before it, each forced and timed on its own in this process; `first_step_seconds` is what is
                # left: the planner -- `plan_seconds` -- matrix buffers, descriptors, the first launches)
                'cold_start_seconds': cold,
                'ms_per_step_hip_events_median': statistics.median(step_ms) if step_ms else None,
                'ms_per_step_hip_events_min': min(step_ms) if step_ms else None,
            },
            'roofline': {
                'bound': 'hbm',
                'kernel': 'dq::wave_pass_kernel',
                # PHYSICAL rate of the dominant kernel: bytes its launches read + wrote / their summed duration
                'achieved': physical,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': physical / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': traffic_src,
                'launches': launches,
                'avg_launch_ms': avg_ms,
                'full_launches': len(full_ms),
                'full_launch_avg_ms': (sum(full_ms) / len(full_ms)) if full_ms else None,
                'physical_bytes_per_launch': (sum(kernel_bytes) / launches) if launches else None,
                # SURVEY 8(d) accounting: every fused gate counted as its own read + write of the state.  NOT a
                # fraction of anything physical: effective / achieved = how many gate-passes one physical pass replaces
                'algorithmic_bytes_per_launch': alg_per_launch,
                'effective_GBs': effective,
                'fusion_factor': (effective / physical) if physical else None,
                'torch_copy_GBs': copy_gbs,
            },
        }
        if sweep is not None:
            # the measured ceiling: the pass kernel's own skeleton (one register-move record, in place) and the best
            # single-H pass of the same run; `frac` stays against the 8 TB/s of the microarchitecture guide
            ceil_ = max([sweep['skeleton']] + list(sweep['h'].values()))
            line['roofline']['measured_ceiling_GBs'] = ceil_
            line['roofline']['measured_ceiling_what'] = (
                'best of: the pass kernel with ONE uncontrolled-X record (register moves) in place over the same resident '
                'state' + ('' if not sweep['h'] else ', single-H passes on every target bit') + '; measured in this run')
            line['roofline']['skeleton_pass_GBs'] = sweep['skeleton']
            line['roofline']['frac_of_measured_ceiling'] = (physical / ceil_) if ceil_ else None
        if sweep is not None and sweep['h']:
            hs, cs = list(sweep['h'].values()), list(sweep['cnot'].values())
            line['roofline']['single_gate'] = {
                'what': 'one un-fused gate per pass over the same resident state, physical GB/s: H on every target bit; '
                        'CNOT pairs near / far, control above / below the target (charged the whole state)',
                'h_GBs': _stats(hs), 'h_frac_of_peak': {k: v / HBM_PEAK_GBS for k, v in _stats(hs).items() if k != 'n'},
                'h_slowest_bits': sorted(sweep['h'], key=sweep['h'].get)[:3],
                'cnot_GBs': _stats(cs), 'cnot_frac_of_peak': {k: v / HBM_PEAK_GBS for k, v in _stats(cs).items() if k != 'n'},
            }
        if parity is not None:
            line['parity'] = parity
        if z0 is not None:
            line['config']['expectation_Z0_sample0'] = z0
        if distributed:
            per_step = {k: dstats[k] for k in ('remaps', 'folded_permutes', 'permute_passes', 'pairwise_exchanges')}
            wire = dstats['wire_bytes']
            links = min(7, world - 1)
            assert nshards == world
            line['config']['exchange_per_step'] = per_step
            # the first local stretch behind reset(): rank 0 (this one) runs it with the known-zero masks, the other ranks --
            # all zeros -- not at all; --no-zero-state switches both off
            line['config']['first_stretch_uses_the_zero_state'] = bool(dq.executor.CONFIG['zero_state'])
            line['config']['zero_shard_stretches_rank0'] = dstats.get('zero_shard_stretches')
            # dry run of the exchange schedule (no data): steps that trade real rank bits / virtual ones, and how much
            # of the wire volume (in shards per rank) travels while other rows of the shard compute
            vb_ = dstats.get('virtual_bits', 0)
            prims_ = [p_ for op_ in cir.operators for p_ in op_.prims(decompose=True)]
            g_ = int(math.log2(world))
            line['config']['virtual_rank_bits'] = vb_
            if batch is None:
                line['config']['virtual_bits_model'] = dq.distributed.virtual_bits_table(
                    prims_, n, n - g_, fresh=True, restore=not cir.lazy_layout)
            line['config']['local_first_exchanges_per_step'] = dstats.get('local_first_exchanges')
            # the passes around an exchange in slices (un-batched shards): protocol bits, remaps that were sliced,
            # launches of the last pass in front of them / of the first pass behind them (this rank), memsets the step needed
            line['config']['slice_exchange'] = {
                'bits': (dq.distributed.slice_bits_wanted(cir.init_state)
                         if (batch is None and getattr(cir, 'init_state', None) is not None) else (0 if batch is not None else None)),
                'sliced_remaps_per_step': dstats.get('sliced_remaps'),
                'launches_of_the_last_passes': dstats.get('slice_launches_last'),
                'launches_of_the_first_passes_behind': dstats.get('slice_launches_first'),
                'zero_fills_per_step': dstats.get('zero_fills'),
                'deferred_tails_per_step': dstats.get('deferred_tails'), 'deferred_gates_per_step': dstats.get('deferred_gates')}
            line['config']['exchange_plan'] = {
                'with_virtual_bits': dq.distributed.count_exchange_steps(prims_, n, g_, virtual_bits=vb_, reorder=True),
                'without': dq.distributed.count_exchange_steps(prims_, n, g_, virtual_bits=0, reorder=True) if vb_ else None}
            # the extension: qubits left where the last remap put them (cir.lazy_layout = True), <Z0> from the shards as they lie
            line['ms_per_step_lazy_layout'] = lazy_ms
            line['value_lazy_layout'] = (total_gate_applies / args.steps / (lazy_ms * 1e-3)) if lazy_ms else None
            line['config']['collectives_per_step'] = comm_stats
            line['config']['remap_timings'] = {
                'what': 'per remap of a step (rank 0, medians over steps and sample groups, HIP events on the group\'s '
                        'stream): the local passes in front of the exchange; exchange issued -> the stream got past the '
                        'wait for it (with several groups in flight this includes what the stream did in between)',
                'remaps': remap_rows}
            line['config']['norm2_sample0'] = norm2
            line['xgmi'] = {
                'wire_bytes_per_rank_per_step': wire,
                'links_used': links,
                'peak_GBs_per_link': XGMI_LINK_GBS,
                # lower bound of the link rate: the whole step time is charged to the exchange (compute overlaps it)
                'achieved_GBs_per_rank_lower_bound': wire / (elapsed / args.steps) / 1e9,
                'frac_lower_bound': (wire / (elapsed / args.steps) / 1e9 / (links * XGMI_LINK_GBS)) if links else None,
                'shard_volumes_sent_per_step': wire / shard_bytes if shard_bytes else None,
                'global_qubits': int(math.log2(world)),
            }
        if qaoa is not None:
            line['config']['qaoa_ring'] = qaoa
        if not args.no_cpu_baseline and not multi:
            line['cpu_baseline'] = cpu_baseline(min(n, 28), dtype, args.cpu_seconds)
        print(json.dumps(line))
    if multi:
        torch.distributed.barrier()      # rank 0 measured a few extras; leave together
        dq.cleanup_distributed()


if __name__ == '__main__':
    main()
