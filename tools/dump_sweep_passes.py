"""Per-pass breakdown of the fused reverse sweep (benchmark generator, every Rx trainable): records, reductions,
LDS trips and the measured duration of every pass.  usage: python tools/dump_sweep_passes.py [n] [depth]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepquantum_amd as dq
from bench import random_circuit_spec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
    cir.zero_grad()
    cir()
    cir.expectation().sum().backward()

step()
torch.cuda.synchronize()
dq.executor.PROFILE['enabled'] = True
dq.executor.PROFILE['events'].clear()
step()
torch.cuda.synchronize()
dq.executor.PROFILE['enabled'] = False
plan = [p for p in dq.executor._PLAN_CACHE.values() if any(o.kind == 'grad' for o in p.prim_ops)][-1]
steps = [s for s in plan.steps if isinstance(s, dq.fusion.FusedStep)]
ev = dq.executor.PROFILE['events'][-len(steps):]
tot = 0.0
for i, (s, (a, b, ng, _nb)) in enumerate(zip(steps, ev)):
    ms = a.elapsed_time(b); tot += ms
    kinds = {}
    for oi in s.ops:
        op = plan.prim_ops[oi]
        k = 'grad' if op.kind == 'grad' else 'x' if op.kind == 'x' else ('h' if op.mode == 3 else 'rx' if op.mode == 2 else 'g')
        kinds[k] = kinds.get(k, 0) + 1
    print(f'pass {i:2d}: m={s.desc.m} records {len(s.ops):3d} {kinds} rounds {s.nrounds} trips {s.ntranspose}  {ms:6.2f} ms')
print('sweep total', round(tot, 1), 'ms over', len(steps), 'passes')
