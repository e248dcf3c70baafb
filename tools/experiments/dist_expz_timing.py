"""Two ranks on one GPU (gloo): where a sharded step spends its time with / without the fused Z expectation values."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
import deepquantum_amd as dq

rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dq.setup_distributed('gloo')
n, depth, batch = 25, 10, 4
spec = bench.random_circuit_spec(n, depth, 1234)
for fused in (True, False, True, False):
    dq.executor.CONFIG['fused_expectation'] = fused
    cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex64, torch.device('cuda', 0), distributed=True)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); torch.distributed.barrier(); t0 = time.perf_counter()
        with torch.no_grad():
            cir(data)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.no_grad():
            ev = cir.expectation()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3), round((t2 - t1) * 1e3)))
    if rank == 0:
        print('fused', fused, 'forward / expectation ms per iteration:', ts, 'plan s', round(dq.executor.PLAN_STATS['seconds'], 2),
              dict(dq.distributed.LAST_RUN), flush=True)
dq.cleanup_distributed()
