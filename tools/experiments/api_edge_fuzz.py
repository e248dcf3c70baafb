import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import deepquantum_amd as dq
DEV = sys.argv[1] if len(sys.argv) > 1 else 'cuda'
if DEV == 'cpu':
    from _cpu_backend import CpuTestBackend
    dq.backend.set_test_backend(CpuTestBackend())
n = 13
def build(grad):
    torch.manual_seed(4)
    c = dq.QubitCircuit(n)
    c.hlayer(); c.rxlayer(encode=True); c.cnot_ring(); c.rylayer(); c.rzz([0, 5]); c.hlayer()
    c.observable(0); c.observable([1, 2], 'xy')
    return c.to(DEV)
g = torch.Generator().manual_seed(9)
B = 3
psi = torch.randn(B, 2**n, 1, generator=g) + 1j * torch.randn(B, 2**n, 1, generator=g)
psi = (psi / psi.norm(dim=1, keepdim=True)).to(torch.complex64).to(DEV)
data = torch.rand(B, n, generator=g).to(DEV)
variants = {
  'plain': (data, psi),
  'conj state': (data, psi.conj().resolve_conj().conj()),            # lazy conj bit, same values
  'strided state': (data, torch.stack([psi, psi], dim=-1)[..., 0]),    # stride 2 along the last dim
  'transposed data': (data.t().contiguous().t(), psi),
  'expanded state': (data, psi[:1].expand(B, -1, -1) ),
  'double data': (data.double(), psi),
}
ref = None
for name, (d, s) in variants.items():
    for mode in ('nograd', 'grad'):
        c = build(mode == 'grad')
        try:
            if mode == 'nograd':
                with torch.no_grad():
                    out = c(data=d, state=s); ev = c.expectation()
                val = (out.reshape(B, -1).cpu(), ev.cpu())
            else:
                d2 = d.clone().requires_grad_(True) if d.dtype == torch.float32 else d.clone().float().requires_grad_(True)
                out = c(data=d2, state=s); ev = c.expectation(); ev.sum().backward()
                val = (out.detach().reshape(B, -1).cpu(), ev.detach().cpu(), d2.grad.cpu())
        except Exception as e:
            print(f'{name:18s} {mode:6s} RAISED {type(e).__name__}: {str(e)[:100]}'); continue
        key = mode if name != 'expanded state' else mode + '_exp'
        if name in ('plain',):
            ref = ref or {}; ref[mode] = val
        if name == 'expanded state':
            c2 = build(False)
            continue_ref = None
        base = ref[mode]
        if name == 'expanded state':
            # reference: the same with a materialised copy
            c2 = build(mode == 'grad')
            s2 = s.contiguous()
            if mode == 'nograd':
                with torch.no_grad():
                    o2 = c2(data=d, state=s2); e2 = c2.expectation()
                base = (o2.reshape(B, -1).cpu(), e2.cpu())
            else:
                d3 = d.clone().requires_grad_(True)
                o2 = c2(data=d3, state=s2); e2 = c2.expectation(); e2.sum().backward()
                base = (o2.detach().reshape(B, -1).cpu(), e2.detach().cpu(), d3.grad.cpu())
        errs = [ (a - b).abs().max().item() for a, b in zip(val, base)]
        print(f'{name:18s} {mode:6s} max diffs {errs}')
