import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from deepquantum_amd import backend, fusion
n, dev = 28, torch.device('cuda', 0)
H = (torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5).reshape(-1).to(dev)
x = torch.zeros(4, 1 << n, dtype=torch.cfloat, device=dev); x[:, 0] = 1
def run(L, bits):
    geom = fusion.default_geometry(False); geom.min_low = L; geom.max_gates = 40
    ops = [fusion.PrimOp('gen', (b,), (), 0, 1) for b in bits]
    steps = fusion.schedule(ops, n, geom); assert len(steps) == 1
    st = steps[0]; km = fusion.kernel_matrices(steps, ops, H)
    backend.apply_fused(x, km, 0, st.desc, out=x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): backend.apply_fused(x, km, 0, st.desc, out=x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
print('single far bit, L=7, base [7,8,9,10]+b')
print(' '.join(f'{b}:{run(7,[7,8,9,10,b]):.2f}' for b in range(11, 28)))
print('single far bit, L=5, base [5..10]+b')
print(' '.join(f'{b}:{run(5,[5,6,7,8,9,10,b]):.2f}' for b in range(11, 28)))
print('pairs L=7 base [7,8,9]+(b,b+d)')
for d in (1, 2, 3, 4):
    print(d, ' '.join(f'{b}:{run(7,[7,8,9,b,b+d]):.2f}' for b in range(10, 28 - d)))
print('triples L=7 base [7,8]+(b,b+2,b+4)')
print(' '.join(f'{b}:{run(7,[7,8,b,b+2,b+4]):.2f}' for b in range(9, 24)))
print('L=5 all 7 spaced by 2 starting b')
print(' '.join(f'{b}:{run(5,[b+2*i for i in range(7)]):.2f}' for b in range(5, 16)))
print('L=5 all 7 spaced by 3 starting b')
print(' '.join(f'{b}:{run(5,[b+3*i for i in range(7)]):.2f}' for b in range(5, 10)))
print('L=5: low block [5..5+k) + top block')
print(' '.join(f'k{k}:{run(5, list(range(5,5+k)) + list(range(28-(7-k),28))):.2f}' for k in range(0, 8)))
