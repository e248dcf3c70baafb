"""cProfile INSIDE the backward of the circuit node (it runs on the autograd engine's device thread, where a profile of
the main thread does not look): 12-6 chart circuit, circuit kept."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepquantum_amd as dq
from deepquantum_amd import executor
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
orig = executor._AdjointCircuit._first_order
state = {'on': False, 't': 0.0, 'n': 0}
def wrapped(*a, **k):
    t0 = time.perf_counter()
    if state['on']:
        pr.enable()
    try:
        return orig(*a, **k)
    finally:
        if state['on']:
            pr.disable()
        state['t'] += time.perf_counter() - t0; state['n'] += 1
executor._AdjointCircuit._first_order = staticmethod(wrapped)

n, layer = 12, 6
params = torch.ones(3 * n * layer, device=dev, requires_grad=True)
kept = circuit(n, layer)
def grad_kept():
    params.grad = None
    kept(data=params); kept.expectation().backward()
for _ in range(3): grad_kept()
torch.cuda.synchronize()
state.update(on=True, t=0.0, n=0)
t0 = time.perf_counter()
for _ in range(10): grad_kept()
torch.cuda.synchronize()
print(f'total {(time.perf_counter()-t0)/10*1e3:.2f} ms per gradient; _first_order {state["t"]/state["n"]*1e3:.2f} ms per call ({state["n"]} calls)')
for key in ('cumulative', 'tottime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
    print('\n'.join(l[:160] for l in s.getvalue().splitlines()[:44]))
