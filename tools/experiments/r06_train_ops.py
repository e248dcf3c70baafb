"""Round 6: which torch operators are the elementwise "glue" kernels of a training step (n = 28, depth 40)?  torch.profiler
over three steps, device time per operator / kernel, with the Python frames that launched the big ones."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepquantum_amd as dq
from bench import random_circuit_spec
n, depth = 28, 40
cir = dq.QubitCircuit(n)
for op in random_circuit_spec(n, depth, 1234):
    if op[0] == 'h': cir.h(op[1])
    elif op[0] == 'rx': cir.rx(op[1])
    else: cir.cnot(op[1], op[2])
cir.observable(0)
cir.to('cuda')
def step():
    cir.zero_grad(); cir(); cir.expectation().sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=50))
