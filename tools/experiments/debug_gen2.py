import sys, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import deepquantum_amd as dq
from deepquantum_amd import backend, fusion
from _cpu_backend import CpuTestBackend
backend.set_test_backend(CpuTestBackend())
import _wave_emulator as emu
from test_fusion_cpu import run_reference
n=12
g = torch.Generator().manual_seed(1)
a = torch.randn(4,4,generator=g,dtype=torch.float64)+1j*torch.randn(4,4,generator=g,dtype=torch.float64)
m,_ = torch.linalg.qr(a); mats = m.reshape(-1).to(torch.complex64)
x = torch.randn(1, 1<<n, generator=g, dtype=torch.float64)+1j*torch.randn(1,1<<n,generator=g,dtype=torch.float64)
x = (x/x.norm()).to(torch.complex64)
for t in ((3,7),(7,3),(0,1),(1,0),(11,5)):
  for c in ((), (2,), (9,)):
    if set(c) & set(t): continue
    ops=[fusion.PrimOp('gen', t, c, 0)]
    geom=fusion.default_geometry(False); geom.plan_min_bits=12
    steps=fusion.schedule(ops,n,geom)
    km=fusion.kernel_matrices(steps,ops,mats)
    ref=run_reference(x,ops,mats)
    d=backend.apply_fused(x.clone(),km,0,steps[0].desc)
    e=emu.run_pass(steps[0].desc,n,x.numpy().copy(),km.numpy(),0)
    kp=emu.descriptor(steps[0].desc,n)
    print(t,c,'interp',(d-ref).abs().max().item(),'emu',np.abs(e-ref.numpy()).max(), [list(kp.rec[i])[:7] for i in range(kp.nrec_bytes//32)])
