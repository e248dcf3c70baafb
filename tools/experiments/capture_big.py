"""Does a whole training step capture as one HIP graph at state sizes beyond the launch-bound regime?  One size per
process (a failed capture may take the process down).  usage: python tools/experiments/capture_big.py N [fwd]"""
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import deepquantum_amd as dq  # noqa: E402
from bench import random_circuit_spec  # noqa: E402

faulthandler.enable()
n = int(sys.argv[1])
fwd_only = len(sys.argv) > 2
cir = dq.QubitCircuit(n)
depth = int(os.environ.get("DEPTH", 20))
for op in random_circuit_spec(n, depth, 1234):
    if op[0] == 'h':
        cir.h(op[1])
    elif op[0] == 'rx':
        cir.rx(op[1])
    else:
        cir.cnot(op[1], op[2])
cir.observable(0)
cir.to('cuda')
cir.zero_grad(set_to_none=True)


def step():
    if fwd_only:
        with torch.no_grad():
            cir()
            return cir.expectation()
    cir()
    cir.expectation().sum().backward()


if os.environ.get('EAGER'):          # as tools/bench_train.py does: eager steps and a no-grad forward first
    for _ in range(3):
        cir.zero_grad(set_to_none='A' in os.environ.get('VAR', ''))
        step()
    torch.cuda.synchronize()
    if 'B' in os.environ.get('VAR', ''):
        dq.executor._STEADY.clear()
        dq.executor._PLAN_CACHE.clear()
    if 'C' in os.environ.get('VAR', ''):
        torch.cuda.empty_cache()
    if 'D' in os.environ.get('VAR', ''):
        dq.backend._ws_cache.clear()
    if 'E' in os.environ.get('VAR', ''):
        import gc
        cir.state = None
        cir._expz = None
        gc.collect()
    if 'F' in os.environ.get('VAR', ''):
        import gc
        for m in cir.modules():
            for k in list(m.__dict__):
                if k.startswith('_') and k not in ('_parameters', '_buffers', '_modules', '_backward_hooks', '_forward_hooks', '_forward_pre_hooks', '_state_dict_hooks', '_load_state_dict_pre_hooks', '_non_persistent_buffers_set', '_backward_pre_hooks', '_forward_hooks_with_kwargs', '_forward_hooks_always_called', '_forward_pre_hooks_with_kwargs', '_state_dict_pre_hooks', '_load_state_dict_post_hooks', '_is_full_backward_hook', '_zero_mark') and isinstance(m.__dict__[k], (torch.Tensor, tuple, list, dict)) and k not in ('_complex_names',):
                    print('dropping', type(m).__name__, k)
                    m.__dict__[k] = None if not isinstance(m.__dict__[k], dict) else {}
        gc.collect()
    if os.environ.get('EAGER') == '2':
        with torch.no_grad():
            for _ in range(2):
                cir()
        torch.cuda.synchronize()
    cir.zero_grad(set_to_none=True)
print('n', n, 'fwd only' if fwd_only else 'training step', 'capturing ...', flush=True)
graph = dq.CapturedGraph(step)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
print('n', n, 'replay', (time.perf_counter() - t0) / 5 * 1e3, 'ms', flush=True)
