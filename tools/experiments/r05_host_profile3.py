"""cProfile INSIDE a Hessian row (executor._SweepGrads.backward runs on the autograd engine's device thread): the reference's
chart circuit 12-6, torch.autograd.functional.hessian, circuit kept."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepquantum_amd as dq
from deepquantum_amd import executor
from torch.autograd.functional import hessian
dev = torch.device('cuda')

def circuit(n, layer):
    cir = dq.QubitCircuit(n)
    for _ in range(layer):
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        cir.rxlayer(encode=True); cir.rzlayer(encode=True); cir.rxlayer(encode=True)
    cir.observable(basis='x')
    return cir.to(dev)

pr = cProfile.Profile()
orig = executor._SweepGrads.backward
st = {'on': False, 't': 0.0, 'n': 0}
def wrapped(ctx, *cots):
    t0 = time.perf_counter()
    if st['on']: pr.enable()
    try:
        return orig(ctx, *cots)
    finally:
        if st['on']: pr.disable()
        st['t'] += time.perf_counter() - t0; st['n'] += 1
executor._SweepGrads.backward = staticmethod(wrapped)
n, layer = 12, 6
k = circuit(n, layer)
x = torch.ones(3 * n * layer, device=dev)
def f(p):
    k(data=p); return k.expectation()
hessian(f, x); torch.cuda.synchronize()
st.update(on=True, t=0.0, n=0)
t0 = time.perf_counter(); hessian(f, x); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f'hessian {tot:.2f} s; _SweepGrads.backward {st["t"]/st["n"]*1e3:.2f} ms per row ({st["n"]} rows, {st["t"]:.2f} s)')
for key in ('cumulative', 'tottime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    print('\n'.join(l[:165] for l in s.getvalue().splitlines()[:46]))
