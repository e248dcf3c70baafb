"""torch.vmap over the circuit (the reference's own batching, circuit.py:232-240) and torch.func.jacrev around it: ONE node
with vmap rules (executor._FusedCircuit) against one node per gate (round 4; executor.CONFIG['fused_transforms'] = False)
and against the native batch.  usage: python tools/bench_vmap.py [n] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.func as tf
import deepquantum_amd as dq
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hess = len(sys.argv) > 3 and sys.argv[3] == 'hessian'
dev = torch.device('cuda', 0)
spec = bench.random_circuit_spec(n, 20, 1234)
cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex64, dev)


def timed(fn, reps=5):
    try:
        return _timed(fn, reps)
    except torch.OutOfMemoryError:          # (a node per gate keeps every intermediate state of every sample)
        torch.cuda.empty_cache()
        return float('nan')


def _timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def native():
    with torch.no_grad():
        return cir(data)


def vmapped():
    with torch.no_grad():
        return tf.vmap(cir._forward_helper, in_dims=(0, None))(data, cir.init_state.state)


def fvec(p):
    cir(data=p)
    return cir.expectation().reshape(-1)


for q in range(4):
    cir.observable(q + 1)
x = data[0].clone()
rows = {}
rows['native batch'] = timed(native)
for on in (True, False):
    dq.executor.CONFIG['fused_transforms'] = on
    tag = 'one node (fused passes)' if on else 'a node per gate (round 4)'
    rows[f'torch.vmap(circuit), {tag}'] = timed(vmapped)
    rows[f'torch.func.jacrev, 5 observables x {cir.ndata} angles, {tag}'] = timed(lambda: tf.jacrev(fvec)(x), reps=3)
    rows[f'torch.func.vmap(grad) over {batch} rows, {tag}'] = timed(lambda: tf.vmap(tf.grad(lambda p: fvec(p).sum()))(data), reps=3)
    rows[f'torch.func.jvp (one direction), {tag}'] = timed(lambda: tf.jvp(fvec, (x,), (torch.ones_like(x),)), reps=3)
    if hess:
        rows[f'torch.func.hessian (jacfwd over jacrev), {tag}'] = timed(lambda: tf.hessian(lambda p: fvec(p).sum())(x), reps=2)
        rows[f'torch.func.jacrev(jacrev): the {cir.ndata} x {cir.ndata} Hessian of sum <Z_q>, {tag}'] = timed(
            lambda: tf.jacrev(tf.jacrev(lambda p: fvec(p).sum()))(x), reps=2)
dq.executor.CONFIG['fused_transforms'] = True
print(f'n = {n}, depth 20 ({len(spec)} gates), complex64, batch {batch}')
for k, v in rows.items():
    print(f'{v:10.2f} ms  {k}' if v == v else f'   out of memory  {k}')
