"""A circuit started from its own |0..0> on the GPU (dq_apply_fused_zext_c64 / _c128 through the C ABI): the first passes
leave out what is still known to be zero.  Kernel level: every pass of a schedule with its mask, on buffers that hold NaN
wherever the kernel must not read and a sentinel wherever it must not write, against the oracle applying the gates one
by one (1e-4 complex64 / 1e-10 complex128).  Circuit level: on / off equality of states, expectation values and
gradients, and the bytes the bench accounts for."""

import numpy as np
import pytest
import torch

import deepquantum_amd as dq
from deepquantum_amd import backend, executor, fusion

from test_wave_cpu import random_ops, reference
from test_zero_state_cpu import _circuit, _written_dead

pytestmark = pytest.mark.gpu
TOL = {False: 1e-4, True: 1e-10}
PREC = pytest.mark.parametrize('is128', [False, True], ids=['c64', 'c128'])


def dev():
    return torch.device('cuda', 0)


@PREC
@pytest.mark.parametrize('n,ngates,seed,batch,shared', [(14, 120, 3, 2, False), (16, 240, 4, 3, True), (18, 300, 5, 1, False),
                                                        (20, 360, 6, 2, True)])
def test_passes_with_known_zero_bits(n, ngates, seed, batch, shared, is128):
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, ngates, seed)
    mats = mats.to(dtype)
    geom = fusion.default_geometry(is128)
    geom.permute_store = True
    geom.plan_min_bits = 12
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    masks = fusion.zero_state_masks(steps, n)
    assert masks is not None and sum(1 for k in masks if k) >= 2
    x0 = torch.zeros(1, 1 << n, dtype=dtype)
    x0[0, 0] = 1
    ref = reference(x0, ops, mats)
    md = fusion.kernel_matrices(steps, ops, mats).to(dev())
    idx = torch.arange(1 << n, dtype=torch.int64, device=dev())
    nan = complex(float('nan'), float('nan'))
    sentinel = complex(7.0, -7.0)
    cur = (x0 if shared else x0.expand(batch, -1).contiguous()).to(dev())
    for si, (st, kz) in enumerate(zip(steps, masks)):
        if kz:
            src = cur.clone()
            if si > 0 or not shared:
                src[:, (idx & kz) != 0] = nan          # (the very first input is a real state)
            out = torch.full((batch, 1 << n), sentinel, dtype=dtype, device=dev())
            backend.apply_fused(src, md, 0, st.desc, out=out, known_zero=kz)
            untouched = torch.from_numpy(_written_dead(st, kz, n, np.arange(1 << n, dtype=np.int64))).to(dev())
            assert bool((out[:, untouched] == sentinel).all()) and not bool((out[:, ~untouched] == sentinel).any())
            assert not bool(torch.isnan(out.real).any())
            cur = out
        else:
            assert not bool((cur == sentinel).any())       # every bit has been in a tile by now
            nxt = torch.empty_like(cur)
            backend.apply_fused(cur, md, 0, st.desc, out=nxt)
            cur = nxt
    err = (cur.cpu() - ref).abs().max().item()
    assert err < TOL[is128], err


def test_masks_the_library_refuses():
    ops, mats = random_ops(14, 60, 1)
    geom = fusion.default_geometry(False)
    geom.permute_store = True
    steps = fusion.schedule(ops, 14, geom)
    md = fusion.kernel_matrices(steps, ops, mats).to(dev())
    x = torch.zeros(1, 1 << 14, dtype=torch.complex64, device=dev())
    out = torch.empty_like(x)
    with pytest.raises(RuntimeError, match='known_zero'):
        backend.apply_fused(x, md, 0, steps[0].desc, out=out, known_zero=1)          # a contiguous low bit
    with pytest.raises(RuntimeError, match='known_zero'):
        backend.apply_fused(x, md, 0, steps[0].desc, out=out, known_zero=1 << 14)    # not a bit of the state


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128], ids=['c64', 'c128'])
@pytest.mark.parametrize('n,batch,depth', [(18, None, 10), (21, 4, 10), (24, 2, 12)])
def test_circuit_from_its_own_zero_state(dtype, n, batch, depth):
    import bench

    spec = bench.random_circuit_spec(n, depth, seed=21)
    cir, data = bench.build_circuit(dq, n, spec, batch, dtype, dev())
    old = dict(executor.CONFIG)
    try:
        with torch.no_grad():
            executor.CONFIG['zero_state'] = True
            a = cir(data=data).clone()
            zp, passes = executor.LAST_RUN['zero_passes'], executor.LAST_RUN['passes']
            ea = cir.expectation().clone()
            executor.CONFIG['zero_state'] = False
            b = cir(data=data).clone()
            assert executor.LAST_RUN['zero_passes'] == 0
            eb = cir.expectation().clone()
        assert zp >= 1 and passes >= zp
        assert not bool(torch.isnan(a.real).any())
        tol = 1e-12 if dtype == torch.complex128 else 1e-6
        assert (a - b).abs().max().item() < tol and (ea - eb).abs().max().item() < 10 * tol
        norm = (a.reshape(-1, 1 << n).abs() ** 2).sum(-1)
        assert (norm - 1).abs().max().item() < 2e-4      # (the reference's fixed matrices are float32-rounded in both precisions)
    finally:
        executor.CONFIG.update(old)


def test_gradients_from_the_zero_state_on_gpu():
    import bench

    n = 20
    spec = bench.random_circuit_spec(n, 8, seed=9)
    cir, data = bench.build_circuit(dq, n, spec, None, torch.complex64, dev())
    old = dict(executor.CONFIG)
    try:
        grads = []
        for on in (True, False):
            executor.CONFIG['zero_state'] = on
            p = data.clone().requires_grad_(True)
            cir(data=p)
            zp = executor.LAST_RUN['zero_passes']
            cir.expectation().sum().backward()
            grads.append(p.grad.clone())
            assert (zp >= 1) == on
        assert (grads[0] - grads[1]).abs().max().item() < 1e-5
    finally:
        executor.CONFIG.update(old)


def test_density_matrix_circuit_from_its_own_zero_state():
    n = 8                        # vec(rho): 16 index bits
    old = dict(executor.CONFIG)
    try:
        res = []
        for on in (True, False):
            executor.CONFIG['zero_state'] = on
            cir = dq.QubitCircuit(n, den_mat=True)
            cir.hlayer()
            for q in range(n - 1):
                cir.cnot(q, q + 1)
            cir.rxlayer(inputs=[0.3 + 0.1 * q for q in range(n)])
            cir.amp_damp(2, 0.2)
            cir.observable(0)
            cir.to(dev())
            with torch.no_grad():
                res.append((cir().clone(), cir.expectation().clone(), executor.LAST_RUN['zero_passes']))
        assert res[0][2] >= 1 and res[1][2] == 0
        assert (res[0][0] - res[1][0]).abs().max().item() < 1e-6 and (res[0][1] - res[1][1]).abs().max().item() < 1e-6
    finally:
        executor.CONFIG.update(old)


def test_bytes_accounted_for_a_zero_state_step():
    """executor.PROFILE (bench.py's roofline): a pass is charged what it moves, not the whole state."""
    import bench

    n, batch = 22, 4
    spec = bench.random_circuit_spec(n, 12, seed=2)
    cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex64, dev())
    prof = executor.PROFILE
    prof['enabled'], prof['events'] = True, []
    try:
        with torch.no_grad():
            cir(data=data)
        torch.cuda.synchronize()
        by = [ev[3] for ev in prof['events']]
        state = batch * (1 << n) * 8
        zp = executor.LAST_RUN['zero_passes']
        assert zp >= 2 and len(by) == executor.LAST_RUN['passes']
        assert by[0] < state // 256 and by[zp - 1] < 2 * state and all(b == 2 * state for b in by[zp:])
    finally:
        prof['enabled'], prof['events'] = False, []
