"""Offline: what the kernel executes on the headline circuit after the one-qubit runs are merged (mode mix, per pass)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq
import bench
n, depth = int(os.environ.get('N', 28)), 40
spec = bench.random_circuit_spec(n, depth, 1234)
cir, data = bench.build_circuit(dq, n, spec, 2, torch.complex64, torch.device('cpu'))
cir.encode(data)
prims = cir.prims()
merged = dq.executor.merge_one_qubit_runs(prims)
def name(p):
    if p.kind == 'x': return 'x' + ('c' if p.controls else '')
    return {0: 'gen', 1: 'real', 2: 'rx', 3: 'had'}[p.mode]
print(len(prims), collections.Counter(name(p) for p in prims))
print(len(merged), collections.Counter(name(p) for p in merged))
plan = dq.executor.make_plan(merged, n, False, True)
steps = [s for s in plan.steps if isinstance(s, dq.fusion.FusedStep)]
cost = {'gen': 66, 'real': 34, 'rx': 34, 'had': 17, 'xc': 18, 'x': 12}
tot = 0
for i, s in enumerate(steps):
    c = collections.Counter(name(merged[oi]) for oi in s.ops)
    v = sum(cost[k] * m for k, m in c.items())
    tot += v
    print(f'pass {i:2d} m={s.desc.m} gates {len(s.ops):3d} rounds {s.nrounds} trips {s.ntranspose} swaps {s.nswaps} valu~{v} {dict(c)}')
print('trips', sum(s.ntranspose for s in steps)); print('LDS trips', sum(s.ntranspose for s in steps), 'in-wave exchange rounds', sum(s.nswaps for s in steps)); print('passes', len(steps), 'gate VALU per pass', tot / len(steps))
