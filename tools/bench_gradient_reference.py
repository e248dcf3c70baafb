"""The one chart the reference publishes for ``QubitCircuit``: wall time of a gradient and of a
Hessian of its benchmark circuit -- per layer a CNOT chain and Rx / Rz / Rx encoder layers, <X..X> --
n-layers 4-2 .. 12-6 (examples/benchmarks/gradient_benchmark.py:127-163), measured here the way that script measures it
(build the circuit inside the timed function, ``exp.backward()`` / ``torch.autograd.functional.hessian``), on the HIP
path, eager and (gradient only) replayed as one HIP graph.  The reference's numbers are read off its log-scale bar
charts (hardware not stated): only their range is known, 0.02 .. 0.28 s per gradient and 0.25 .. 58 s per Hessian.

usage: python tools/bench_gradient_reference.py [--trials 5] [--no-hessian] [--max-hessian-params 150]
"""
import argparse
import os
import sys
import time

import torch
from torch.autograd.functional import hessian

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--trials', type=int, default=5)
ap.add_argument('--no-hessian', action='store_true')
ap.add_argument('--max-hessian-params', type=int, default=216)
ap.add_argument('--device', default='cuda')
ap.add_argument('--second-order', default='tangent', choices=['tangent', 'replay'], help='A/B: Hessian rows by the tangent circuit or by per-gate nodes')
ap.add_argument('--no-small-fused-sweep', action='store_true', help='A/B: reverse sweeps below a tile as undo-then-reduce')
args = ap.parse_args()
dev = torch.device(args.device)
dq.executor.CONFIG['small_fused_sweep'] = not args.no_small_fused_sweep
dq.executor.CONFIG['second_order'] = args.second_order


def circuit(n, layer):
    cir = dq.QubitCircuit(n)
    for _ in range(layer):
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        cir.rxlayer(encode=True)
        cir.rzlayer(encode=True)
        cir.rxlayer(encode=True)
    cir.observable(basis='x')
    return cir.to(dev)


def timed(fn, trials):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(trials):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts)


print('# reference (chart, hardware not stated): gradient 0.02 s (4-2) .. 0.28 s (12-6); Hessian 0.25 s (4-2) .. 58 s (12-6)')
print(f'# {"n-layers":>8s} {"params":>6s} | gradient eager (build + fwd + bwd) | same circuit kept | HIP graph | Hessian (functional.hessian) | Hessian (torch.func.jacrev(jacrev))')
for n in (4, 6, 8, 10, 12):
    for layer in (2, 4, 6):
        npar = 3 * n * layer
        params = torch.ones(npar, device=dev, requires_grad=True)

        def grad_as_reference():            # gradient_benchmark.py:127-144: a new circuit per call
            if params.grad is not None:
                params.grad.zero_()
            cir = circuit(n, layer)
            cir(data=params)
            cir.expectation().backward()
            return params.grad

        g_min, g_avg = timed(grad_as_reference, args.trials)
        kept = circuit(n, layer)

        def grad_kept():
            if params.grad is not None:
                params.grad.zero_()
            kept(data=params)
            kept.expectation().backward()
            return params.grad

        k_min, _ = timed(grad_kept, args.trials)
        # one HIP graph for forward + expectation + backward
        static = torch.ones(npar, device=dev, requires_grad=True)
        cg = circuit(n, layer)

        def step():
            cg(data=static)
            ev = cg.expectation()
            (g,) = torch.autograd.grad(ev.sum(), static)
            return g

        try:
            graph = dq.CapturedGraph(step)
            h_min, _ = timed(graph.replay, args.trials * 4)
            ok = torch.allclose(graph.replay(), grad_kept(), atol=1e-4)
            graph_txt = f'{h_min * 1e3:8.3f} ms{"" if ok else " (MISMATCH)"}'
        except Exception as e:               # noqa: BLE001
            graph_txt = f'failed: {type(e).__name__}'
        hs_txt = 'skipped'
        if not args.no_hessian and npar <= args.max_hessian_params:
            x = torch.ones(npar, device=dev)

            def f(p):
                cir = circuit(n, layer)
                cir(data=p)
                return cir.expectation()

            hs_min, _ = timed(lambda: hessian(f, x), max(1, args.trials // 2))
            hs_txt = f'{hs_min:8.3f} s'
            # all rows in ONE traversal: reverse over reverse with torch.func (the gates run as per-gate nodes with vmap rules)
            try:
                import torch.func as tf

                hv_min, _ = timed(lambda: tf.jacrev(tf.jacrev(f))(x), max(1, args.trials // 2))
                same = (tf.jacrev(tf.jacrev(f))(x) - hessian(f, x)).abs().max().item()
                hs_txt += f' | {hv_min:8.3f} s (max difference {same:.1e})'
            except Exception as e:               # noqa: BLE001
                hs_txt += f' | failed: {type(e).__name__}: {str(e)[:60]}'
        print(f'  {n:>2d}-{layer:<5d} {npar:>6d} | {g_min * 1e3:9.2f} ms (avg {g_avg * 1e3:8.2f}) | {k_min * 1e3:9.2f} ms | {graph_txt} | {hs_txt}',
              flush=True)
