"""Training step (forward + expectation + backward w.r.t. every Rx angle) of the benchmark generator circuit.
usage: python tools/bench_train.py [--n 24] [--depth 20] [--modes adjoint,per_gate] [--dtype c64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=24)
ap.add_argument('--depth', type=int, default=20)
ap.add_argument('--modes', default='adjoint,per_gate')
ap.add_argument('--dtype', default='c64')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--no-fused-sweep', action='store_true', help='A/B: the undo-then-reduce reverse sweep')
ap.add_argument('--graph', action='store_true', help='also: the whole step captured as ONE HIP graph (dq.CapturedGraph) and replayed')
args = ap.parse_args()

dq.executor.CONFIG['fused_sweep'] = not args.no_fused_sweep
if os.environ.get('DQ_REDUCED_GRAD') == '0':          # A/B: every reduction forms all of G
    dq.executor.CONFIG['reduced_grad_sums'] = False
if os.environ.get('DQ_MAX_GATES'):                    # A/B: gates + reductions a pass may hold (72 in rounds 3-5)
    dq.executor.CONFIG['max_gates'] = int(os.environ['DQ_MAX_GATES'])
if os.environ.get('DQ_TERMINAL_GRAD') == '0':         # A/B: two sums per X rotation (round 4) instead of one
    dq.executor.CONFIG['terminal_grad_sums'] = False
for mode in args.modes.split(','):
    dq.executor.CONFIG['grad_mode'] = mode
    cir = dq.QubitCircuit(args.n)
    nrx = 0
    for op in random_circuit_spec(args.n, args.depth, 1234):
        if op[0] == 'h':
            cir.h(op[1])
        elif op[0] == 'rx':
            cir.rx(op[1])
            nrx += 1
        else:
            cir.cnot(op[1], op[2])
    cir.observable(0)
    cir.to('cuda')
    if args.dtype == 'c128':
        cir.to(torch.double)

    def step():
        cir.zero_grad()
        cir()
        cir.expectation().sum().backward()

    for _ in range(3):              # plans, second buffers, the caching allocator's blocks: steady state from here on
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    lat = time.perf_counter() - t0      # one step with the device idle before and after: host work exposed
    t0 = time.perf_counter()
    for _ in range(args.reps):          # a training loop: steps back to back, the host runs ahead of the device
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    with torch.no_grad():
        for _ in range(2):          # its own plan (merged one-qubit runs) and its second state buffer (permuted
            cir()                   # stores) are made once, outside the timing
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cir()
        torch.cuda.synchronize()
        fwd = time.perf_counter() - t0
    sb = (8 if args.dtype == 'c64' else 16) * 2**args.n
    sweep = dict(dq.executor.LAST_SWEEP) if mode == 'adjoint' else {}
    print(f'{mode:9s} {sweep} n={args.n} depth={args.depth} ({args.n * args.depth} gates, {nrx} trainable) {args.dtype}: '
          f'step {dt * 1e3:8.1f} ms back to back, {lat * 1e3:6.1f} ms alone (no-grad forward {fwd * 1e3:6.1f} ms), peak {torch.cuda.max_memory_allocated() / sb:6.1f} states '
          f'= {torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB')
    if args.graph and mode == 'adjoint':
        # the same step as one HIP graph: no host work between the kernels at all (what the device alone needs)
        cir.zero_grad(set_to_none=True)

        def gstep():
            cir()
            cir.expectation().sum().backward()

        try:
            graph = dq.CapturedGraph(gstep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                graph.replay()
            torch.cuda.synchronize()
            print(f'          the step as ONE HIP graph: {(time.perf_counter() - t0) / args.reps * 1e3:8.1f} ms per replay')
            del graph
        except Exception as e:       # noqa: BLE001
            print(f'          the step as ONE HIP graph: failed ({type(e).__name__}: {e})')
    del cir
    torch.cuda.empty_cache()
