"""Eager gradient of the reference's 12-6 chart circuit with the circuit kept: A/B of the round's last host-side changes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepquantum_amd as dq
n, layer = 12, 6
dev = torch.device('cuda')
cir = dq.QubitCircuit(n)
for _ in range(layer):
    for i in range(n - 1):
        cir.cnot(i, i + 1)
    cir.rxlayer(encode=True); cir.rzlayer(encode=True); cir.rxlayer(encode=True)
cir.observable(basis='x')
cir.to(dev)
params = torch.ones(3 * n * layer, device=dev, requires_grad=True)
def step():
    if params.grad is not None:
        params.grad.zero_()
    cir(data=params)
    cir.expectation().backward()
def timed(reps=30):
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3
import cProfile, pstats
for tag in ('plain', 'plain again'):
    if tag.startswith('after'):
        dq.executor._PLAN_CACHE.clear()
    lo, med = timed()
    print(f'{tag}: min {lo:.2f} ms, median {med:.2f} ms, sweep passes {dq.executor.LAST_SWEEP.get("passes")}, plans made {dq.executor.PLAN_STATS}', flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(48); st.sort_stats('tottime').print_stats(22)
