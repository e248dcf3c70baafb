"""Large-size cross-check of every scheduler / kernel optimisation against the plain path: the benchmark generator at
n = 26..28, several seeds, run (a) with the defaults -- merged one-qubit runs, wide planner, permuted stores that also re-label the low bits, next-tile
prefetch, in-wave exchanges, deferred Hadamard / Rx factors -- and (b) with all of that off (one gate per record,
first-come tiles, in-place passes with fixed low bits, one tile per workgroup, LDS trips only); the two states must agree to complex64
round-off, and <Z0>, the norm and a checksum of checksums are printed.  usage (GPU box): python tools/crosscheck_large.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import deepquantum_amd as dq  # noqa: E402

dev = torch.device('cuda', 0)
PLAIN = {'merge_min_amps': None, 'permute_store': False, 'plan_width': 0, 'free_low': False}


def run(n, depth, seed, batch, dtype, plain):
    keep = dict(dq.executor.CONFIG)
    try:
        if plain:
            dq.executor.CONFIG.update(PLAIN)
        else:
            dq.executor.CONFIG['merge_min_amps'] = 1 << 20
            dq.executor.CONFIG['plan_big_amps'] = 1 << 20
        spec = bench.random_circuit_spec(n, depth, seed)
        cir, data = bench.build_circuit(dq, n, spec, batch, dtype, dev)
        with torch.no_grad():
            st = cir(data).reshape(batch, -1).clone()
            ev = cir.expectation().reshape(-1)[0].item()
        return st, ev, dict(dq.executor.LAST_RUN)
    finally:
        dq.executor.CONFIG.clear()
        dq.executor.CONFIG.update(keep)
        dq.executor._PLAN_CACHE.clear()


worst = 0.0
for n, depth, batch, dtype, seeds in ((26, 40, 4, torch.complex64, (7, 99, 2025)), (28, 40, 2, torch.complex64, (1234, 5)),
                                      (25, 30, 2, torch.complex128, (3,))):
    for seed in seeds:
        a, eva, sa = run(n, depth, seed, batch, dtype, plain=False)
        b, evb, sb = run(n, depth, seed, batch, dtype, plain=True)
        err = (a - b).abs().max().item()
        tol = 1e-4 if dtype == torch.complex64 else 1e-10
        worst = max(worst, err / tol)
        norm = (a.abs() ** 2).sum(-1)
        print(f'n={n} depth={depth} seed={seed} batch={batch} {str(dtype)[-3:]}: max |default - plain| = {err:.2e} (tol {tol:g}); '
              f'<Z0> {eva:+.6e} / {evb:+.6e}; norm {norm[0].item():.7f}; passes {sa["passes"]} (plain {sb["passes"]}), '
              f'kernel gates {sa["gates"]} (plain {sb["gates"]}), LDS trips {sa["transposes"]} (plain {sb["transposes"]})',
              flush=True)
        assert err < tol, 'optimised and plain paths disagree'
        del a, b
        torch.cuda.empty_cache()
print(f'all cases agree; worst error / tolerance = {worst:.3f}')
