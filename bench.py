#!/usr/bin/env python
"""Headline benchmark: gate-applies/s and HBM GB/s of the QubitCircuit statevector hot path.

Workload (BASELINE.json configs[2]): QubitCircuit(28), seeded random H / Rx / CNOT
circuit of depth 40 (1120 gates, seed 1234), complex64, batch = 16 with per-sample Rx angles (the
``torch.vmap`` case of the reference), initial state |0...0>, no_grad forward.  One "step" = one
forward pass of the whole circuit over the whole batch, inputs resident in HBM.

N > 1 (launched by torchrun, one rank per GPU): weak scaling.  The samples of a batch are independent
circuits, so the default shards THEM: every rank runs the same 28-qubit circuit on its own 16 samples (global
batch 16 N), no collective in the data path -- the process group only carries the barrier and the max over
ranks of the elapsed time.  ``--sharded-state`` instead measures the index-bit-sharded state (BASELINE configs
4/5 style): n = 28 + log2(N) qubits over the N ranks, the same 2^28 amplitudes x 16 samples per GPU, qubit
remapping by RCCL all-to-all (what a state that does not fit one GPU needs; link-bound on xGMI).

Prints ONE JSON line on rank 0 (see the contract in the task statement), with two extra objects:
``roofline`` for the dominant kernel (the fused pass) and ``cpu_baseline`` (the oracle = restatement of
the reference's permute/reshape/matmul path, timed on this host's cores on a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--nqubit', type=int, default=28, help='qubits per GPU-sized shard (n = nqubit + log2 gpus)')
    ap.add_argument('--depth', type=int, default=40)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--dtype', choices=['c64', 'c128'], default='c64')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--tile-bits', type=int, default=None, help='override fused tile size m')
    ap.add_argument('--min-low', type=int, default=None)
    ap.add_argument('--max-gates', type=int, default=None)
    ap.add_argument('--no-fuse', action='store_true')
    ap.add_argument('--backend', default='nccl', help="process-group backend for N > 1 ('nccl' = RCCL; 'gloo' lets "
                    'several ranks share one GPU for a functional check)')
    ap.add_argument('--sharded-state', action='store_true',
                    help='N > 1: index-bit-sharded state of 28 + log2(N) qubits instead of sharding the batch')
    ap.add_argument('--max-far', type=int, default=None, help='scheduler: max gathered bits >= far-bit per pass')
    ap.add_argument('--far-bit', type=int, default=None)
    ap.add_argument('--plan-width', type=int, default=None,
                    help='pass planner beam width (0 = first-come tiles, 1 = greedy; default: library)')
    ap.add_argument('--plan-branch', type=int, default=None, help='pass planner: tiles tried per beam state')
    ap.add_argument('--no-asm-loop', action='store_true', help='A/B: gate loop in C++ around the jump table')
    ap.add_argument('--no-compare', action='store_true',
                    help='skip the extra untimed-for-value runs with merging off (config.unmerged_ms_per_step): '
                         'tools/profile.sh uses it so that the profiled launches are the timed ones only')
    ap.add_argument('--no-permute-store', action='store_true',
                    help='A/B: in-place passes (every pass gathers its qubits where they canonically live)')
    ap.add_argument('--no-merge', action='store_true',
                    help='A/B: do not multiply runs of one-qubit gates on the same qubit into one matrix')
    ap.add_argument('--tiles-per-wg', type=int, default=None,
                    help='A/B: tiles a complex64 workgroup walks with next-tile prefetch (1 = off; default: library)')
    ap.add_argument('--tiles-per-wg', type=int, default=None,
                    help='A/B: tiles a complex64 workgroup walks with next-tile prefetch (1 = off; default: library)')
    ap.add_argument('--traffic-json', default=None, help='file with PMC-measured HBM bytes per launch')
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
    slice of a batch sharded over ranks this is (its samples get their own angles)."""
    cir = dq.DistributedQubitCircuit(n) if distributed else dq.QubitCircuit(n)
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
    g = torch.Generator().manual_seed(1234 + shard)
    data = torch.rand(batch, len(angles), generator=g, dtype=real) * 2 * math.pi
    if shard == 0:
        data[0] = torch.tensor(angles, dtype=real)  # sample 0 = the generator's own angles
    return cir, data.to(device)


def device_copy_bandwidth(device, nbytes=1 << 32, reps=5):
    """Read+write GB/s of a plain device-to-device copy of ``nbytes`` (the achievable-HBM yardstick)."""
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


def single_gate_bandwidth(dq, n, batch, dtype, device, reps=3):
    """Physical read+write GB/s of ONE gate application (H on the top qubit) over a resident (batch, 2^n) state:
    the un-fused figure the north star's ">= 60 % of the HBM roofline" refers to."""
    from deepquantum_amd import backend, fusion

    is128 = dtype == torch.complex128
    mat = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5).to(dtype).reshape(-1).to(device)
    ops = [fusion.PrimOp('gen', (n - 1,), (), 0, 3)]
    steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
    km = fusion.kernel_matrices(steps, ops, mat)
    x = torch.zeros(batch, 1 << n, dtype=dtype, device=device)
    x[:, 0] = 1
    backend.apply_fused(x, km, 0, steps[0].desc, out=x)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        backend.apply_fused(x, km, 0, steps[0].desc, out=x)
    e1.record()
    torch.cuda.synchronize(device)
    return 2 * x.numel() * x.element_size() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(n, spec, dtype, budget_s):
    """Oracle (port of the reference's evolve_state path) on this host: first gates of the same
    workload, batch element 0, until ``budget_s`` seconds are spent."""
    from oracle import statevec_oracle as oracle

    torch.set_num_threads(os.cpu_count() or 1)
    real = torch.float32 if dtype == torch.complex64 else torch.float64
    x = torch.zeros(1, 2**n, dtype=dtype)
    x[0, 0] = 1
    h = oracle.fixed_matrix('h').to(dtype)
    cnot = (torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]]) + 0j).to(dtype)
    done, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for op in spec:
            if op[0] == 'h':
                x = oracle.apply_gate_wires(x, h, n, [op[1]])
            elif op[0] == 'rx':
                x = oracle.apply_gate_wires(x, oracle.rx_matrix(oracle.theta_tensor(op[2]).to(real)).to(dtype), n, [op[1]])
            else:
                x = oracle.apply_gate_wires(x, cnot, n, [op[1], op[2]])
            done += 1
            if time.perf_counter() - t0 > budget_s or done >= 64:
                break
    x = x.contiguous()
    dt = time.perf_counter() - t0
    return {
        'value': done / dt,
        'unit': 'gate-applies/s',
        'cores': torch.get_num_threads(),
        'kind': 'port',
        'sample': f'first {done} gates of the same n={n} seed-1234 circuit, batch element 0 only, '
                  f'{"c64" if dtype == torch.complex64 else "c128"}, {dt:.1f} s wall',
    }


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    import deepquantum_amd as dq

    dtype = torch.complex64 if args.dtype == 'c64' else torch.complex128
    amp_bytes = 8 if dtype == torch.complex64 else 16
    multi = world > 1
    distributed = multi and args.sharded_state          # index-bit-sharded state; otherwise the batch is sharded
    if multi:
        dq.setup_distributed(args.backend)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    key = 'm_c64' if dtype == torch.complex64 else 'm_c128'
    if args.tile_bits is not None:
        dq.executor.CONFIG[key] = args.tile_bits
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
    if args.no_asm_loop:
        dq.executor.CONFIG['asm_loop'] = False
    if args.no_merge:
        dq.executor.CONFIG['merge_min_amps'] = None
    if args.no_permute_store:
        dq.executor.CONFIG['permute_store'] = False

    if args.tiles_per_wg is not None:
        from deepquantum_amd import _lib

        _lib.check(_lib.load().dq_fused_set_tiles_per_wg(args.tiles_per_wg), 'dq_fused_set_tiles_per_wg')
    if args.tiles_per_wg is not None:
        from deepquantum_amd import _lib

        _lib.check(_lib.load().dq_fused_set_tiles_per_wg(args.tiles_per_wg), 'dq_fused_set_tiles_per_wg')
    n = args.nqubit + (int(math.log2(world)) if distributed else 0)
    spec = random_circuit_spec(n, args.depth, args.seed)
    ngates = len(spec)
    cir, data = build_circuit(dq, n, spec, args.batch, dtype, device, distributed, rank if multi and not distributed else 0)
    # algorithmic bytes per gate: 2 * 2^(n - nc) * sizeof(amp) per batch sample
    alg_bytes = sum(2 * (2 ** (n - (1 if op[0] == 'cnot' else 0))) * amp_bytes for op in spec) * args.batch

    prof = dq.executor.PROFILE

    def step():
        with torch.no_grad():
            cir(data)      # N>1: (batch, 2^L) shards on every rank, one exchange schedule for the whole batch
            return cir.expectation()

    def sync():
        torch.cuda.synchronize(device)
        if multi:
            torch.distributed.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    prof['enabled'] = True
    prof['events'].clear()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof['enabled'] = False
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    # dominant kernel: the fused pass -- durations from HIP events recorded on the launch stream
    kernel_ms = [a.elapsed_time(b) for a, b, _ in prof['events']]
    launches = len(kernel_ms)
    alg_per_launch = (alg_bytes / (world if distributed else 1) * args.steps / launches) if launches else 0.0   # this rank's share
    avg_ms = (sum(kernel_ms) / launches) if launches else float('nan')
    achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc runs of this same command,
    # tools/profile.sh; committed under profiles/): only reported for the workload it was collected on.
    traffic = None
    tj = args.traffic_json
    if tj is None and not distributed:   # (batch-sharded ranks run the single-GPU workload: same traffic)
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'traffic_n{n}_b{args.batch}_{args.dtype}.json')))
        tj = cands[-1] if cands else None
    if tj and os.path.exists(tj):
        traffic = json.load(open(tj)).get('hbm_bytes_per_launch')
    stats = dict(dq.executor.LAST_RUN)
    z0 = float(out.reshape(-1)[0]) if out is not None else None
    # the same circuit with every gate applied on its own (no products of one-qubit runs), for comparison: N = 1 only
    unmerged_ms = None
    if not multi and not args.no_merge and not args.no_compare and dq.executor.CONFIG['merge_min_amps'] is not None:
        keep = dq.executor.CONFIG['merge_min_amps']
        dq.executor.CONFIG['merge_min_amps'] = None
        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(min(args.steps, 2)):
            step()
        sync()
        unmerged_ms = (time.perf_counter() - t0) / min(args.steps, 2) * 1e3
        dq.executor.CONFIG['merge_min_amps'] = keep
    copy_gbs = device_copy_bandwidth(device) if rank == 0 else None
    single_gbs = None
    if rank == 0 and not distributed and n >= 12:
        out = None
        cir.state = None
        torch.cuda.empty_cache()
        single_gbs = single_gate_bandwidth(dq, n, args.batch, dtype, device)

    if rank == 0:
        total_gate_applies = ngates * args.batch * args.steps * (world if multi and not distributed else 1)
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
            'config': {
                'workload': f'QubitCircuit({n}) random H/Rx/CNOT depth {args.depth} ({ngates} gates, seed {args.seed}), '
                            f'{"complex64" if dtype == torch.complex64 else "complex128"}, batch={args.batch} '
                            f'(per-sample Rx angles), |0..0> start, no_grad forward'
                            + (f', index-bit sharded over {world} GPUs' if distributed else '')
                            + (f'; {world} ranks x {args.batch} samples (global batch {world * args.batch})'
                               if multi and not distributed else ''),
                'nqubit': n,
                'depth': args.depth,
                'batch': args.batch,
                'parallelism': (f'state-shard x{world} (RCCL all-to-all qubit remap)' if distributed else
                                f'batch-shard x{world} (independent samples, no collective in the data path)' if multi
                                else 'single GPU'),
                'fused_passes_per_step': stats.get('passes'),
                'lds_round_trips_per_step': stats.get('transposes'),
                # 2x2 matrices the kernel applies per sample after runs of one-qubit gates on the same qubit were
                # multiplied together (executor.merge_one_qubit_runs; `--no-merge` applies all `ngates` one by one)
                'kernel_gates_per_step': stats.get('gates'),
                'unmerged_ms_per_step': unmerged_ms,
            },
            'roofline': {
                'bound': 'hbm',
                'kernel': 'dq::fused_pass_kernel',
                'achieved': achieved,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'launches': launches,
                'avg_launch_ms': avg_ms,
                'algorithmic_bytes_per_launch': alg_per_launch,
                'actual_state_bytes_per_launch': 2 * (2**n >> (int(math.log2(world)) if distributed else 0)) * amp_bytes * args.batch,
                'note': 'achieved counts every fused gate as its own read+write of the state (SURVEY 8d), so it '
                        'can exceed the HBM peak; actual_state_bytes_per_launch / avg_launch_ms is the physical rate',
            },
        }
        line['roofline']['physical_GBs'] = (line['roofline']['actual_state_bytes_per_launch'] / (avg_ms * 1e-3) / 1e9
                                            if launches else None)
        # also quote the physical rate against an in-framework device copy measured on this box
        line['roofline']['device_copy_GBs'] = copy_gbs
        if single_gbs is not None:
            line['roofline']['single_gate_GBs'] = single_gbs
            line['roofline']['single_gate_frac_of_peak'] = single_gbs / HBM_PEAK_GBS
        if launches and copy_gbs:
            line['roofline']['physical_frac_of_copy'] = line['roofline']['physical_GBs'] / copy_gbs
            line['roofline']['physical_frac_of_peak'] = line['roofline']['physical_GBs'] / HBM_PEAK_GBS
        if z0 is not None:
            line['config']['expectation_Z0_sample0'] = z0
        if not args.no_cpu_baseline and not multi:
            line['cpu_baseline'] = cpu_baseline(n, spec, dtype, args.cpu_seconds)
        print(json.dumps(line))
    if multi:
        torch.distributed.barrier()      # rank 0 measured a few extras; leave together
        dq.cleanup_distributed()


if __name__ == '__main__':
    main()
