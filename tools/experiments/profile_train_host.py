"""Host-side profile of one training step (forward + backward through the fused reverse sweep)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import deepquantum_amd as dq
from bench import random_circuit_spec

n, depth = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 20
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
    cir()
    cir.expectation().sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
