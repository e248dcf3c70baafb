"""Where the host time of a launch-bound gradient goes (the reference's chart circuit 12-6): cProfile of an eager
gradient with the circuit kept, with a new circuit per call, and of one Hessian (functional.hessian, 8-4)."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepquantum_amd as dq
dev = torch.device('cuda')

def circuit(n, layer):
    cir = dq.QubitCircuit(n)
    for _ in range(layer):
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        cir.rxlayer(encode=True); cir.rzlayer(encode=True); cir.rxlayer(encode=True)
    cir.observable(basis='x')
    return cir.to(dev)

def prof(fn, reps, title, top=28):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f'=== {title}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call')
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(top)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:top + 12]))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:30]))

n, layer = 12, 6
params = torch.ones(3 * n * layer, device=dev, requires_grad=True)
kept = circuit(n, layer)
def grad_kept():
    params.grad = None
    kept(data=params); kept.expectation().backward()
def grad_new():
    params.grad = None
    cir = circuit(n, layer); cir(data=params); cir.expectation().backward()
prof(grad_kept, 10, 'gradient, circuit kept (12-6)')
prof(grad_new, 10, 'gradient, new circuit per call (12-6)')
from torch.autograd.functional import hessian
x = torch.ones(3 * 8 * 4, device=dev)
k8 = circuit(8, 4)
def f(p):
    k8(data=p); return k8.expectation()
prof(lambda: hessian(f, x), 2, 'Hessian 8-4 (96 rows), circuit kept', top=34)
