"""Host time of the pieces of _AdjointCircuit.backward (the autograd engine runs it on its own thread: cProfile of the
main thread does not see it)."""
import os, sys, time, functools, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import deepquantum_amd as dq
from deepquantum_amd import executor, fusion, backend
from bench import random_circuit_spec

n, depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = collections.defaultdict(float)


def timed(mod, name, label=None):
    f = getattr(mod, name)

    @functools.wraps(f)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label or name] += time.perf_counter() - t0
    setattr(mod, name, w)


timed(executor._AdjointCircuit, '_sweep_fused')
timed(executor, '_run_nograd')
timed(executor, 'make_plan')
timed(executor, '_flat_mats')
timed(executor, '_inverse')
timed(backend, 'apply_fused')
timed(fusion, 'defer_rx')
bw = executor._AdjointCircuit.backward


def bw_timed(ctx, gy):
    t0 = time.perf_counter()
    try:
        return bw(ctx, gy)
    finally:
        T['backward total'] += time.perf_counter() - t0


executor._AdjointCircuit.backward = staticmethod(bw_timed)
cir = dq.QubitCircuit(n)
for op in random_circuit_spec(n, depth, 1234):
    if op[0] == 'h':
        cir.h(op[1])
    elif op[0] == 'rx':
        cir.rx(op[1])
    else:
        cir.cnot(op[1], op[2])
cir.observable(0)
cir.to('cuda')


def step():
    cir.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    cir()
    T['forward call'] += time.perf_counter() - t0
    t0 = time.perf_counter()
    loss = cir.expectation().sum()
    T['expectation call'] += time.perf_counter() - t0
    t0 = time.perf_counter()
    loss.backward()
    T['backward call (host)'] += time.perf_counter() - t0


for _ in range(3):
    step()
torch.cuda.synchronize()
T.clear()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'n={n} depth={depth}: step {tot / reps * 1e3:.1f} ms')
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f'  {k:24s} {v / reps * 1e3:8.2f} ms per step')
