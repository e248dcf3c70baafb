"""Host cost of one dq_apply_fused call (validation + translation + launch) as Python sees it."""
import os, sys, time, ctypes as C
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from deepquantum_amd import backend, fusion, _lib
from test_wave_cpu import random_ops

n = 16
ops, mats = random_ops(n, 400, 3)
geom = fusion.default_geometry(False)
steps = fusion.schedule(ops, n, geom)
dev = torch.device('cuda', 0)
x = torch.zeros(1, 1 << n, dtype=torch.complex64, device=dev); x[0, 0] = 1
md = fusion.kernel_matrices(steps, ops, mats.to(torch.complex64)).to(dev)
for st in steps:
    backend.apply_fused(x, md, 0, st.desc, out=x)
torch.cuda.synchronize()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    for st in steps:
        backend.apply_fused(x, md, 0, st.desc, out=x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{len(steps)} passes, {sum(len(s.ops) for s in steps) / len(steps):.0f} gates per pass: host {1e6 * (t1 - t0) / reps / len(steps):.1f} us per call, '
      f'with the GPU {1e6 * (t2 - t0) / reps / len(steps):.1f} us')
lib = _lib.load()
fn = lib.dq_apply_fused_c64
p, o, m = x.data_ptr(), x.data_ptr(), md.data_ptr()
s = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
for _ in range(reps):
    for st in steps:
        fn(p, o, m, 0, n, 1, C.byref(st.desc), s)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'raw ctypes call: host {1e6 * (t1 - t0) / reps / len(steps):.1f} us per call')
