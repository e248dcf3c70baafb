"""Shared test helpers: golden fixtures, spec-driven circuit construction, tolerances."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import specs  # noqa: E402

_GOLDEN = None

# north-star tolerances: final amplitudes and expectations
TOL = {'c64': 1e-4, 'c128': 1e-10}
CDTYPE = {'c64': torch.complex64, 'c128': torch.complex128}


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = np.load(os.path.join(HERE, 'golden', 'golden.npz'))
    return _GOLDEN


def gold(key):
    return torch.from_numpy(golden()[key])


_GOLDEN_EXTRA = None


def gold_extra(key):
    global _GOLDEN_EXTRA
    if _GOLDEN_EXTRA is None:
        _GOLDEN_EXTRA = np.load(os.path.join(HERE, 'golden', 'golden_extra.npz'))
    return torch.from_numpy(_GOLDEN_EXTRA[key])


def check_reset_against_golden(dq, device=None):
    """Reset (gate.py:3027-3094): the reference's outputs for whole circuits, for the gate on given inputs,
    and for a batched circuit with gradients through two resets (tests/golden/make_golden_extra.py)."""
    for name, spec in specs.RESET_CASES.items():
        for prec in ('c64', 'c128'):          # the reference cannot run these in c128; same numbers expected
            cir = specs.build(dq, 5, spec)
            cir.observable(0)
            cir.observable([1, 2], 'xz')
            if device is not None:
                cir.to(device)
            if prec == 'c128':
                cir.to(torch.double)
            with torch.no_grad():
                st = cir().reshape(-1).cpu()
                ev = cir.expectation().cpu()
            assert (st - gold_extra(f'{name}/c64/state')).abs().max().item() < 1e-5, (name, prec)
            assert (ev - gold_extra(f'{name}/c64/expectation')).abs().max().item() < 1e-5, (name, prec)
    for i, (wires, ps, _kind) in enumerate(specs.RESET_GATE_CASES):
        psi = gold_extra(f'resetgate/{i}/in')
        gate = dq.Reset(nqubit=4, wires=wires, postselect=ps, tsr_mode=True)
        x = psi.reshape([2] + [2] * 4)
        out = gate(x.to(device) if device is not None else x).reshape(2, -1).cpu()
        assert (out - gold_extra(f'resetgate/{i}/out')).abs().max().item() < 1e-5, (i, wires, ps)
    cir = dq.QubitCircuit(4)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.ry(1, encode=True)
    cir.cnot(0, 2)
    cir.cnot(1, 3)
    cir.reset([2], postselect=0)
    cir.crx(0, 2, encode=True)
    cir.reset([3], postselect=1)
    cir.observable(0)
    cir.observable([1, 2], 'zx')
    data = gold_extra('reset_batched/data').clone()
    if device is not None:
        cir.to(device)
        data = data.to(device)
    data.requires_grad_(True)
    state = cir(data=data)
    ev = cir.expectation()
    ev.sum().backward()
    assert (state.reshape(3, -1).detach().cpu() - gold_extra('reset_batched/state')).abs().max().item() < 1e-5
    assert (ev.detach().cpu() - gold_extra('reset_batched/expectation')).abs().max().item() < 1e-5
    assert (data.grad.cpu() - gold_extra('reset_batched/grad')).abs().max().item() < 1e-4


def check_density_matrix_against_golden(dq, device=None, tol=2e-5):
    """Density-matrix circuits with every channel class, batched data with gradients, partial trace and a
    user-supplied rho (reference outputs: tests/golden/make_golden_extra.py)."""
    for name, c in specs.DM_CASES.items():
        cir = dq.QubitCircuit(c['nqubit'], init_state=c['init'], den_mat=True)
        for method, args, kwargs in c['spec']:
            getattr(cir, method)(*args, **kwargs)
        for wires, basis in c['observables']:
            cir.observable(wires, basis)
        if device is not None:
            cir.to(device)
        with torch.no_grad():
            rho = cir()
            ev = cir.expectation()
        ref = gold_extra(f'dm/{name}/rho')
        assert rho.shape == ref.shape, (rho.shape, ref.shape)
        assert (rho.cpu() - ref).abs().max().item() < tol, name
        assert (ev.cpu() - gold_extra(f'dm/{name}/expectation')).abs().max().item() < tol, name
        probs = cir.measure(shots=200, with_prob=True)
        diag = ref.diagonal().real
        for key, (_cnt, p) in probs.items():
            assert abs(float(p) - float(diag[int(key, 2)])) < tol
    cir = dq.QubitCircuit(3, den_mat=True)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.bit_flip(0, encode=True)
    cir.cnot(0, 1)
    cir.amp_damp(1)
    cir.ry(2, encode=True)
    cir.depolarizing(2, 0.3)
    cir.crz(1, 2, encode=True)
    cir.observable(0)
    cir.observable([1, 2], 'zx')
    prm = list(cir.parameters())
    assert len(prm) == 1
    with torch.no_grad():
        prm[0].copy_(gold_extra('dm/batched/theta'))
    data = gold_extra('dm/batched/data').clone()
    if device is not None:
        cir.to(device)
        data = data.to(device)
    data.requires_grad_(True)
    rho = cir(data=data)
    ev = cir.expectation()
    ev.sum().backward()
    assert (rho.detach().cpu() - gold_extra('dm/batched/rho')).abs().max().item() < tol
    assert (ev.detach().cpu() - gold_extra('dm/batched/expectation')).abs().max().item() < tol
    assert (data.grad.cpu() - gold_extra('dm/batched/data_grad')).abs().max().item() < 10 * tol
    assert (list(cir.parameters())[0].grad.cpu() - gold_extra('dm/batched/theta_grad')).abs().max().item() < 10 * tol
    rho0 = gold_extra('dm/user/rho0')
    assert (dq.qmath.partial_trace(rho0, 3, [0, 2]) - gold_extra('dm/user/ptrace_02')).abs().max().item() < 1e-6
    assert (dq.qmath.partial_trace(rho0, 3, [1]) - gold_extra('dm/user/ptrace_1')).abs().max().item() < 1e-6
    cir = dq.QubitCircuit(3, init_state=rho0, den_mat=True)
    cir.h(0)
    cir.cnot(0, 2)
    cir.phase_damp(1, 0.6)
    if device is not None:
        cir.to(device)
    with torch.no_grad():
        assert (cir().cpu() - gold_extra('dm/user/rho')).abs().max().item() < tol


def build_circuit(dq, name, prec, device=None):
    c = specs.CIRCUITS[name]
    cir = specs.build(dq, c['nqubit'], c['spec'])
    for wires, basis in c.get('observables', []):
        cir.observable(wires, basis)
    if device is not None:
        cir.to(device)
    if prec == 'c128':
        cir.to(torch.double)
    data = None
    if 'data' in c:
        data = torch.tensor(c['data'], dtype=torch.float64 if prec == 'c128' else torch.float32, device=device)
    return cir, data, c


def check_circuit_against_golden(dq, name, prec, device=None, check_unitary=True):
    cir, data, c = build_circuit(dq, name, prec, device)
    tol = TOL[prec]
    with torch.no_grad():
        state = cir(data=data)
        ref = gold(f'{name}/{prec}/state')
        assert state.shape == ref.shape, (state.shape, ref.shape)
        assert state.dtype == CDTYPE[prec]
        err = (state.cpu() - ref).abs().max().item()
        assert err < tol, f'{name}/{prec}: amplitude error {err}'
        if c.get('observables'):
            ev = cir.expectation()
            ref_ev = gold(f'{name}/{prec}/expectation')
            assert ev.shape == ref_ev.shape, (ev.shape, ref_ev.shape)
            assert (ev.cpu() - ref_ev).abs().max().item() < tol
        if 'marginal' in c:
            res = cir.measure(shots=64, with_prob=True, wires=c['marginal'])
            res = res if isinstance(res, list) else [res]
            ref_m = gold(f'{name}/{prec}/marginal')
            for b, r in enumerate(res):
                for bits, (cnt, p) in r.items():
                    assert abs(float(p) - ref_m[b, int(bits, 2)].item()) < max(tol, 1e-6)
        if c.get('unitary') and check_unitary:
            u = cir.get_unitary()
            ref_u = gold(f'{name}/{prec}/unitary')
            assert (u.cpu() - ref_u).abs().max().item() < tol
    return err


def check_adjoint_grad_mode(dq, device=None, dtype=torch.float64, n=6, tol=1e-10):
    """The default autograd mode (one node per circuit: fused forward, reverse sweep with recomputation) against
    the one-node-per-gate mode, on a circuit with fixed float32-rounded gates (exact inverses matter), diagonal,
    controlled, two-qubit trainable gates, batched data and a state that itself requires grad."""
    def build():
        torch.manual_seed(3)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rxlayer(encode=True)
        cir.cnot_ring()
        cir.rylayer()                         # trainable
        cir.rz(0, encode=True)
        cir.p(1)                              # trainable diagonal
        cir.crx(0, 2, encode=True)
        cir.rzz([1, 3])                       # trainable two-qubit diagonal
        cir.rxx([2, 4])                       # trainable two-qubit dense
        cir.toffoli(0, 1, 5)
        cir.u3(3, controls=[0, 5])            # trainable, two controls
        cir.s(2)
        cir.t(4)
        cir.swap([1, 5])
        cir.hlayer()
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        cir.observable([3, 5], 'zz')
        if device is not None:
            cir.to(device)
        if dtype == torch.float64:
            cir.to(torch.double)
        return cir

    results = {}
    for mode in ('per_gate', 'adjoint'):
        dq.executor.CONFIG['grad_mode'] = mode
        try:
            cir = build()
            g = torch.Generator().manual_seed(8)
            data = torch.rand(3, cir.ndata, generator=g, dtype=dtype)
            psi0 = torch.randn(3, 2**n, 1, generator=g, dtype=dtype) + 1j * torch.randn(3, 2**n, 1, generator=g, dtype=dtype)
            psi0 = psi0 / psi0.norm(dim=1, keepdim=True)
            if device is not None:
                data, psi0 = data.to(device), psi0.to(device)
            data.requires_grad_(True)
            psi0.requires_grad_(True)
            cir(data=data, state=psi0)
            loss = (cir.expectation() * torch.tensor([1.0, -0.5, 0.25], dtype=dtype, device=data.device)).sum()
            loss.backward()
            results[mode] = (loss.detach().cpu(), data.grad.cpu(), psi0.grad.cpu(),
                             [p.grad.cpu() for p in cir.parameters()])
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
    a, b = results['per_gate'], results['adjoint']
    assert abs(a[0] - b[0]).item() < tol
    assert (a[1] - b[1]).abs().max().item() < tol, (a[1] - b[1]).abs().max()
    assert (a[2] - b[2]).abs().max().item() < tol, (a[2] - b[2]).abs().max()
    assert len(a[3]) == len(b[3]) and len(a[3]) > 0
    for x, y in zip(a[3], b[3], strict=True):
        assert (x - y).abs().max().item() < tol, (x - y).abs().max()


def check_fused_sweep(dq, device=None, n=12, batch=2, tol=3e-5, dtype=torch.float32):
    # (reduced DQ_FG_GRAD rows leave the other components zero -- checked inside the sweep while this runs; the
    # check reads the accumulator on the host, so it must not stay on for the HIP-graph tests that follow)
    dq.executor.CONFIG['check_grad_rows'] = True
    try:
        _check_fused_sweep(dq, device, n, batch, tol, dtype)
        assert dq.executor.LAST_SWEEP.get('checked_rows', 0) > 0
    finally:
        dq.executor.CONFIG['check_grad_rows'] = False


def _check_fused_sweep(dq, device=None, n=12, batch=2, tol=3e-5, dtype=torch.float32):
    """Circuits whose trainable gates have one or two targets run their reverse sweep as fused passes over psi and the
    cotangent interleaved along an extra index bit, the reductions folded into the passes (DQ_FG_GRAD):
    against per-gate autograd and against the undo-then-reduce sweep, with controlled / diagonal / general trainable
    gates, fixed gates of every kind, batched encoded data and an initial state that requires grad."""
    def build():
        torch.manual_seed(5)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rxlayer(encode=True)
        cir.cnot_ring()
        cir.rylayer()                         # trainable, real
        cir.rzlayer()                         # trainable, diagonal
        cir.cnot_ring(reverse=True)
        cir.u3layer()                         # trainable, general
        cir.rz(0, encode=True)
        cir.p(1)
        cir.crx(0, 2, encode=True)            # trainable with a control
        cir.u3(3, controls=[0, n - 1])        # two controls
        cir.toffoli(0, 1, n - 2)
        cir.rxlayer()
        cir.s(2)
        cir.t(4)
        cir.swap([1, n - 1])
        cir.rxx([2, n - 2])                   # trainable gates on TWO targets: four one-target reduction records each
        cir.ryy([0, 5], controls=[n - 1])
        cir.rzz([3, 4])                       # ... diagonal: two records
        cir.rxy([1, 6])
        cir.rxx([4, n - 3], encode=True)      # ... with one angle per sample
        cir.cry(n - 1, 0)
        cir.hlayer()
        cir.rylayer(encode=True)
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        cir.observable([3, n - 1], 'zz')
        if device is not None:
            cir.to(device)
        if dtype == torch.float64:
            cir.to(torch.double)
        return cir

    results = {}
    for mode, fused in (('per_gate', False), ('adjoint', False), ('adjoint', True)):
        dq.executor.CONFIG['grad_mode'] = mode
        dq.executor.CONFIG['fused_sweep'] = fused
        try:
            cir = build()
            g = torch.Generator().manual_seed(8)
            data = torch.rand(batch, cir.ndata, generator=g, dtype=dtype)
            psi0 = (torch.randn(batch, 2**n, 1, generator=g, dtype=dtype)
                    + 1j * torch.randn(batch, 2**n, 1, generator=g, dtype=dtype))
            psi0 = psi0 / psi0.norm(dim=1, keepdim=True)
            if device is not None:
                data, psi0 = data.to(device), psi0.to(device)
            data.requires_grad_(True)
            psi0.requires_grad_(True)
            cir(data=data, state=psi0)
            loss = (cir.expectation() * torch.tensor([1.0, -0.5, 0.25], device=data.device, dtype=dtype)).sum()
            loss.backward()
            if mode == 'adjoint':
                assert dq.executor.LAST_SWEEP['fused'] == fused
                assert not fused or (dq.executor.LAST_SWEEP['reductions'] > 5 * n and dq.executor.LAST_SWEEP['passes'] >= 1)
                # a backward that records no graph: the Rx / CRx rotations reduce ONE sum each (DQ_FG_GRAD variant 4)
                assert not fused or (4 in dq.executor.LAST_SWEEP['variants'] and 2 not in dq.executor.LAST_SWEEP['variants'])
            results[mode, fused] = (loss.detach().cpu(), data.grad.cpu(), psi0.grad.cpu(),
                                    [p.grad.cpu() for p in cir.parameters()])
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
            dq.executor.CONFIG['fused_sweep'] = True
    a = results['per_gate', False]
    for key in (('adjoint', False), ('adjoint', True)):
        b = results[key]
        assert abs(a[0] - b[0]).item() < tol
        assert (a[1] - b[1]).abs().max().item() < tol, (key, (a[1] - b[1]).abs().max())
        assert (a[2] - b[2]).abs().max().item() < tol, (key, (a[2] - b[2]).abs().max())
        assert len(a[3]) == len(b[3]) and len(a[3]) > 0
        for x, y in zip(a[3], b[3], strict=True):
            assert (x - y).abs().max().item() < tol, (key, (x - y).abs().max())


def check_grad_records(n, device, is128=False):
    """DQ_FG_GRAD records (dq_apply_fused_grad_c64 / _c128) against numpy: psi and lambda interleaved along index bit 0, a
    reduction sum lambda (x) conj(psi) for every target bit, with controls that land on register slots, on thread
    bits and outside the tile, in between gates that leave deferred factors in the registers (Hadamards, Rx); the
    accumulator is added to, rows not named stay untouched, a plain pass refuses the records."""
    import random

    import numpy as np
    import pytest
    from test_fusion_cpu import run_reference

    from deepquantum_amd import backend, fusion

    rng = random.Random(100 + n)
    b = 2
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(b, 2**n, generator=gen, dtype=torch.float64) + 1j * torch.randn(b, 2**n, generator=gen, dtype=torch.float64)
    cdt = torch.complex128 if is128 else torch.complex64
    x = (x / x.norm(dim=1, keepdim=True)).to(cdt)
    ops, mats_l, want = [], [], []
    cur = x.clone()
    off = 0
    h = torch.tensor([[1, 1], [1, -1]], dtype=cdt) / 2 ** 0.5
    for step in range(3 * (n - 1)):
        t = 1 + step % (n - 1)
        th = rng.uniform(0.3, 2.8)
        mat = h if step % 3 == 0 else torch.tensor([[np.cos(th / 2), -1j * np.sin(th / 2)], [-1j * np.sin(th / 2), np.cos(th / 2)]],
                                                  dtype=cdt)
        ops.append(fusion.PrimOp('gen', (t,), (), off, 3 if step % 3 == 0 else 2))
        mats_l.append(mat.reshape(-1))
        off += 4
        cur = run_reference(cur, [fusion.PrimOp('gen', (t,), (), 0, 0)], mat.reshape(-1))
        q = 1 + rng.randrange(n - 1)
        ctrl = tuple(rng.sample([c for c in range(1, n) if c != q], rng.choice([0, 0, 1, 2])))
        row = len(want) + 1                      # row 0 stays untouched
        ops.append(fusion.PrimOp('grad', (q, 0), ctrl, 0, row))
        v = cur.numpy().astype(np.complex128).reshape(b, -1)
        idx = np.arange(1 << n)
        ok = np.ones(1 << n, dtype=bool)
        for c in ctrl:
            ok &= ((idx >> c) & 1) == 1
        g = np.zeros((b, 2, 2), dtype=np.complex128)
        for a_ in range(2):
            for b_ in range(2):
                la = v[:, ok & ((idx & 1) == 1) & (((idx >> q) & 1) == a_)]
                ps = v[:, ok & ((idx & 1) == 0) & (((idx >> q) & 1) == b_)]
                g[:, a_, b_] = (la * ps.conj()).sum(-1)
        want.append(g)
    mats = torch.cat(mats_l)
    geom = fusion.default_geometry(is128)
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    xd, md = x.to(device), fusion.kernel_matrices(steps, ops, mats).to(device)
    acc = torch.full((b, len(want) + 1, 8), 0.5, dtype=torch.float64, device=device)
    with pytest.raises((RuntimeError, AssertionError)):
        backend.apply_fused(xd.clone(), md, 0, steps[0].desc)          # reduction records outside a reverse-sweep pass
    for st in steps:
        backend.apply_fused(xd, md, 0, st.desc, out=xd, grads=acc)
    assert (xd.cpu() - cur).abs().max().item() < (1e-10 if is128 else 1e-4)
    got = torch.view_as_complex((acc - 0.5).reshape(b, -1, 4, 2)).reshape(b, -1, 2, 2).cpu().numpy()
    assert np.abs(got[:, 0]).max() == 0.0
    for r_, g in enumerate(want):
        assert np.abs(got[:, r_ + 1] - g).max() < (1e-12 if is128 else 2e-5) * max(1.0, np.abs(g).max()), (r_, ops[2 * r_ + 1])


def check_fused_sweep_random(dq, device=None, n=13, batch=2, seed=0, ngates=90, tol=5e-5, dtype=torch.float32,
                             expect_fused=True):
    """Fuzz of the fused reverse sweep: a random sequence from the whole gate menu -- fixed, trainable and encoded
    (batched) gates, controls of every arity, diagonal and two-qubit gates in between -- differentiated by the fused
    sweep and by per-gate autograd."""
    import random

    def build():
        rng = random.Random(seed)
        torch.manual_seed(seed)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        for _ in range(ngates):
            menu = ['h', 'x', 'y', 'z', 's', 't', 'rx', 'ry', 'rz', 'p', 'u3', 'cnot', 'cz', 'crx', 'cry', 'crz',
                    'toffoli', 'rzz_enc', 'rx_enc', 'ry_ctrl', 'u3_ctrl2', 'cp']
            menu += ['swap', 'rxx_enc', 'fredkin', 'rxx_train', 'ryy_train_ctrl', 'rzz_train']
            kind = rng.choice(menu)
            w = rng.sample(range(n), 3)
            if kind in ('h', 'x', 'y', 'z', 's', 't'):
                getattr(cir, kind)(w[0], controls=[w[1]] if rng.random() < 0.2 else None)
            elif kind in ('rx', 'ry', 'rz', 'p', 'u3'):
                getattr(cir, kind)(w[0])
            elif kind in ('cnot', 'cz', 'crx', 'cry', 'crz', 'cp'):
                getattr(cir, kind)(w[0], w[1])
            elif kind == 'toffoli':
                cir.toffoli(w[0], w[1], w[2])
            elif kind == 'fredkin':
                cir.fredkin(w[0], w[1], w[2])
            elif kind == 'swap':
                cir.swap([w[0], w[1]])
            elif kind == 'rxx_train':
                cir.rxx([w[0], w[1]])
            elif kind == 'ryy_train_ctrl':
                cir.ryy([w[0], w[1]], controls=[w[2]])
            elif kind == 'rzz_train':
                cir.rzz([w[0], w[1]])
            elif kind == 'rxx_enc':
                cir.rxx([w[0], w[1]], inputs=rng.uniform(0.0, 6.0))       # fixed angle: two-target gates stay untrainable
            elif kind == 'rzz_enc':
                cir.rzz([w[0], w[1]], inputs=rng.uniform(0.0, 6.0))
            elif kind == 'rx_enc':
                cir.rx(w[0], encode=True)
            elif kind == 'ry_ctrl':
                cir.ry(w[0], controls=[w[1]])
            else:
                cir.u3(w[0], controls=[w[1], w[2]])
        cir.observable(0)
        cir.observable([1, n - 1], 'xz')
        if device is not None:
            cir.to(device)
        if dtype == torch.float64:
            cir.to(torch.double)
        return cir

    results = {}
    for mode in ('per_gate', 'adjoint'):
        dq.executor.CONFIG['grad_mode'] = mode
        try:
            cir = build()
            g = torch.Generator().manual_seed(100 + seed)
            data = torch.rand(batch, max(cir.ndata, 1), generator=g, dtype=dtype) * 6.0
            if device is not None:
                data = data.to(device)
            data.requires_grad_(True)
            cir(data=data if cir.ndata else None)
            loss = (cir.expectation() * torch.tensor([1.0, -0.7], device=data.device, dtype=dtype)).sum()
            loss.backward()
            if mode == 'adjoint':
                assert dq.executor.LAST_SWEEP['fused'] == expect_fused and dq.executor.LAST_SWEEP['reductions'] > 0
            results[mode] = (loss.detach().cpu(), data.grad.cpu() if cir.ndata else None,
                             [p.grad.cpu() for p in cir.parameters()])
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
    a, b = results['per_gate'], results['adjoint']
    assert abs(a[0] - b[0]).item() < tol
    if a[1] is not None:
        assert (a[1] - b[1]).abs().max().item() < tol, (a[1] - b[1]).abs().max()
    assert len(a[2]) == len(b[2]) and len(a[2]) > 0
    for x, y in zip(a[2], b[2], strict=True):
        assert (x - y).abs().max().item() < tol, (x - y).abs().max()


def check_edge_cases(dq, device=None):
    """Degenerate and extreme inputs of the circuit driver: no gates, one qubit, a batch of one, only diagonal
    gates, many controls, gates on the first and last wire, repeated forward calls on the same object."""
    def dev_(cir):
        return cir.to(device) if device is not None else cir

    # empty circuit: the initial state comes back, in the reference's shapes
    cir = dev_(dq.QubitCircuit(3))
    out = cir()
    assert out.shape == (8, 1) and out[0, 0] == 1 and out.abs().sum() == 1
    cir = dev_(dq.QubitCircuit(3, init_state='equal'))
    assert (cir().abs() - 8 ** -0.5).abs().max().item() < 1e-6
    # one qubit
    cir = dq.QubitCircuit(1)
    cir.h(0)
    cir.rz(0, 0.5)
    cir.observable(0, 'x')
    dev_(cir)
    st = cir().reshape(-1)
    assert abs(st[0].abs().item() - 0.7071068) < 1e-6 and abs(cir.expectation().item() - 0.8775826) < 1e-5
    # batch of one through 2-D data keeps the batch dimension
    cir = dq.QubitCircuit(2)
    cir.rx(0, encode=True)
    cir.cnot(0, 1)
    dev_(cir)
    data = torch.tensor([[1.2]])
    data = data.to(device) if device is not None else data
    assert cir(data).shape == (1, 4, 1)
    assert cir(data[0]).shape == (4, 1)
    # only diagonal gates, and many controls (9 controls on a 10-qubit register)
    n = 10
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    cir.z(9, controls=list(range(9)))
    cir.p(0, 0.3, controls=list(range(1, 10)))
    cir.x(4, controls=[0, 1, 2, 3, 5, 6, 7, 8, 9])
    cir.rzz([0, 9], 0.7)
    cir.observable(list(range(n)), 'z' * n)
    dev_(cir)
    st = cir().reshape(-1).cpu()
    amp = 0.7071067690849304 ** n
    phase_rzz = torch.exp(torch.tensor(-0.35j))
    # all-ones amplitude: sign from CZ, phase from CP, swapped with |1111011111> (equal amplitude) by the CX
    assert abs(st[0] - amp * phase_rzz) < 1e-6
    assert abs(abs(st[-1]) - amp) < 1e-6 and abs((st.abs() ** 2).sum().item() - (2 * 0.7071067690849304**2) ** n) < 1e-5
    # the same circuit object can be called again and again (plan cache, lazy matrices)
    a = cir().clone()
    b = cir()
    assert torch.equal(a, b)


def check_many_z_observables(dq, device=None, dtype=torch.float32, n=9):
    """A ring of ZZ terms plus single Z, mixed with X / Y strings (the QAOA read-out, examples/qaoa.py:31-44): the
    Z-type strings are evaluated together in one read of the state; values and gradients must equal the
    observable-by-observable evaluation."""
    def build():
        torch.manual_seed(2)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rylayer(encode=True)
        cir.cnot_ring()
        cir.rxlayer()
        for q in range(n):
            cir.observable([q, (q + 1) % n], 'zz')
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        cir.observable([3, 4, 5], 'zzz')
        cir.observable(2, 'x')
        if device is not None:
            cir.to(device)
        if dtype == torch.float64:
            cir.to(torch.double)
        return cir

    g = torch.Generator().manual_seed(4)
    data = torch.rand(3, n, generator=g, dtype=dtype)
    data = data.to(device) if device is not None else data
    tol = 1e-10 if dtype == torch.float64 else 2e-5
    res = {}
    for together in (True, False):
        cir = build()
        d = data.clone().requires_grad_(True)
        cir(d)
        if together:
            ev = cir.expectation()
        else:
            ev = torch.stack([dq.qmath.expectation(cir.state, ob) for ob in cir.observables], dim=-1)
        w = torch.linspace(0.5, 1.5, ev.shape[-1], dtype=dtype, device=ev.device)
        (ev * w).sum().backward()
        res[together] = (ev.detach().cpu(), d.grad.cpu(), [p.grad.cpu() for p in cir.parameters()])
    a, b = res[True], res[False]
    assert a[0].shape == (3, n + 4)
    assert (a[0] - b[0]).abs().max().item() < tol
    assert (a[1] - b[1]).abs().max().item() < 10 * tol
    for x, y in zip(a[2], b[2], strict=True):
        assert (x - y).abs().max().item() < 10 * tol


def check_readout_against_golden(dq, device=None, tol=2e-5):
    """Conditional (measurement-controlled) gates, post-selection, deferred measurement, amplitudes / probabilities,
    measurement probabilities, custom initial states: against the reference's outputs (make_golden_extra.py)."""
    def build(init_state='zeros'):
        cir = dq.QubitCircuit(4, init_state=init_state)
        cir.h(0)
        cir.ry(1, encode=True)
        cir.cnot(0, 2)
        cir.x(3, controls=[0], condition=True)
        cir.rz(2, controls=[1], condition=True, encode=True)
        cir.h(2)
        return cir.to(device) if device is not None else cir

    def close(a, key, t=tol):
        ref = gold_extra(key)
        a = a.detach().cpu()
        assert a.shape == ref.shape, (key, a.shape, ref.shape)
        assert (a - ref).abs().max().item() < t, key

    data = gold_extra('readout/data')
    data = data.to(device) if device is not None else data
    with torch.no_grad():
        cir = build()
        close(cir(data), 'readout/state')
        assert sorted(cir.wires_condition) == gold_extra('readout/wires_condition').tolist()
        for bits in ('00', '01', '10', '11'):
            close(cir.post_select(bits), f'readout/post_select_{bits}')
        close(cir.get_amplitude('0110'), 'readout/amp_0110')
        close(cir.get_prob('1011'), 'readout/prob_1011')
        states, keys, probs = cir.defer_measure(with_prob=True)        # sampled: consistent with post-selection
        for i, key in enumerate(keys):
            assert (states[i].cpu() - gold_extra(f'readout/post_select_{key}')[i]).abs().max().item() < tol
            assert 0 < float(probs[i]) <= 1 + 1e-6
        cir1 = build()
        close(cir1(data[1]), 'readout/single_state')
        close(cir1.post_select('10'), 'readout/single_post_select_10')
        close(cir1.get_prob('01', wires=[1, 3]), 'readout/single_prob_wires')
        close(cir1.get_amplitude('1001'), 'readout/single_amp_1001')
        for wires, tag in ((None, 'measure'), ([0, 2], 'measure02')):
            res = cir1.measure(shots=20000, with_prob=True, wires=wires)
            want = dict(zip(gold_extra(f'readout/{tag}_keys').tolist(), gold_extra(f'readout/{tag}_probs').tolist()))
            for key, (count, prob) in res.items():
                assert abs(float(prob) - want[int(key, 2)]) < tol
                assert abs(count / 20000 - float(prob)) < 0.02
        close(build(init_state=gold_extra('readout/init_vec'))(data[0]), 'readout/state_from_vec')
        batch = gold_extra('readout/init_batch')
        st = dq.QubitState(4, batch).state
        st = st.to(device) if device is not None else st
        close(build()(data[:2], state=st), 'readout/state_from_batch')
        close(dq.amplitude_encoding(torch.arange(1.0, 11.0), 4), 'readout/amplitude_encoding', 1e-6)


def check_extra_gates_against_golden(dq, device=None, tol=2e-5):
    """ProjectionJ in its three planes, HamiltonianGate (Pauli-string list and matrix forms, controlled), LatentGate,
    CombinedSingleGate, Identity: state, expectations and the circuit unitary against the reference's."""
    cir = specs.extra_gates_circuit(dq)
    if device is not None:
        cir.to(device)
    with torch.no_grad():
        st = cir().reshape(-1).cpu()
        ev = cir.expectation().cpu()
        u = cir.get_unitary().cpu()
    assert (st - gold_extra('extra_gates/state')).abs().max().item() < tol
    assert (ev - gold_extra('extra_gates/expectation')).abs().max().item() < tol
    # the reference's get_unitary ignores the controls of ArbitraryGate subclasses (gate.py:318-330: its own unitary
    # disagrees with its own forward by 0.81 here); ours must be consistent with the forward pass and unitary
    assert (u[:, 0] - st).abs().max().item() < tol
    assert (u @ u.mH - torch.eye(16)).abs().max().item() < 10 * tol
    ref_u = gold_extra('extra_gates/unitary')
    assert (ref_u[:, 0] - gold_extra('extra_gates/state')).abs().max().item() > 0.5     # (documents the divergence)


def check_fuzz_against_oracle(dq, device=None, n=13, seeds=(0, 1, 2), depth=6, batch=2, double=False):
    """Random circuits over the whole gate vocabulary (fixed, parametric with per-sample data, controlled, two- and
    three-qubit) through the product path under every scheduler / merging configuration, against the oracle applying
    the same gate matrices one by one.  Exercises the planner, the one-qubit-run products, the handler dispatch and
    the per-gate fallbacks on structures the golden circuits do not contain."""
    import random

    from oracle import statevec_oracle as oracle
    one = ['h', 'x', 'y', 'z', 's', 't', 'sdg', 'tdg']
    par = ['rx', 'ry', 'rz', 'p']
    for seed in seeds:
        rng = random.Random(seed)
        cir = dq.QubitCircuit(n)
        for _ in range(depth):
            for q in range(n):
                r = rng.random()
                others = [w for w in range(n) if w != q]
                if r < 0.3:
                    getattr(cir, rng.choice(one))(q)
                elif r < 0.5:
                    getattr(cir, rng.choice(par))(q, encode=True)
                elif r < 0.6:
                    getattr(cir, rng.choice(par))(q, inputs=rng.uniform(0, 6.28))
                elif r < 0.75:
                    cir.cnot(q, rng.choice(others))
                elif r < 0.8:
                    cir.cz(q, rng.choice(others))
                elif r < 0.85:
                    cir.swap([q, rng.choice(others)])
                elif r < 0.9:
                    c1, c2 = rng.sample(others, 2)
                    cir.toffoli(c1, c2, q)
                elif r < 0.95:
                    cir.rxx([q, rng.choice(others)], inputs=rng.uniform(0, 6.28))
                else:
                    cir.rx(q, inputs=rng.uniform(0, 6.28), controls=[rng.choice(others)])
        if double:
            cir.to(torch.double)
        if device is not None:
            cir.to(device)
        gen = torch.Generator().manual_seed(100 + seed)
        data = torch.rand(batch, cir.ndata, generator=gen, dtype=torch.double if double else torch.float) * 6.28
        data = data.to(device) if device is not None else data
        keep = dict(dq.executor.CONFIG)
        outs = []
        try:
            for cfg in ({'merge_min_amps': None, 'plan_width': 0}, {'merge_min_amps': 0, 'plan_width': 1},
                        {'merge_min_amps': 0, 'plan_width': 4},
                        {'merge_min_amps': 0, 'permute_store': True, 'permute_min_bits': 0},
                        {'merge_min_amps': None, 'plan_width': 0, 'permute_store': True, 'permute_min_bits': 0},
                        {'merge_min_amps': 0, 'permute_store': True, 'permute_min_bits': 0, 'free_low': False},
                        {'merge_min_amps': 0, 'permute_store': True, 'permute_min_bits': 0, 'free_low': 'force',
                         'plan_width': 4},
                        {'merge_min_amps': None, 'permute_store': True, 'permute_min_bits': 0, 'free_low': 'force',
                         'plan_width': 2}):
                dq.executor.CONFIG.update(keep)
                dq.executor.CONFIG.update(cfg)
                dq.executor._PLAN_CACHE.clear()
                with torch.no_grad():
                    outs.append(cir(data).detach().cpu().reshape(batch, -1).clone())
            # the reference: the same matrices (encoded with the last sample by forward()) applied by the oracle
            ref = []
            for b in range(batch):
                cir.encode(data[b])
                x = torch.zeros(1, 1 << n, dtype=outs[0].dtype)
                x[0, 0] = 1
                for op in cir.operators:
                    m = op.update_matrix().detach().cpu().to(outs[0].dtype)
                    x = oracle.apply_gate_wires(x, m, n, list(op.wires), list(op.controls))
                ref.append(x)
            ref = torch.cat(ref)
        finally:
            dq.executor.CONFIG.clear()
            dq.executor.CONFIG.update(keep)
            dq.executor._PLAN_CACHE.clear()
        for k, o in enumerate(outs):
            err = (o - ref).abs().max().item()
            assert err < (1e-12 if double else 5e-6), (seed, k, err)


def check_fused_sweep_with_sloppy_user_matrices(dq, device=None, n=12, tol=2e-5):
    """UAnyGate accepts matrices that are unitary to 1e-4 (reference gate.py:2745-2788).  The fused reverse sweep undoes
    those with the exact inverse and corrects the cotangent by U^dagger U -- with the adjoint standing in for the inverse
    (fine for matrices that are unitary by construction) ten such gates leave psi 1e-3 off."""
    import math

    def build():
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(12)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rylayer()
        for k in range(10):
            a = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
            u = (torch.linalg.qr(a)[0] * (1 + 4e-5 * (1 if k % 2 else -1))).to(torch.complex64)
            cir.any(u, wires=[k % n], controls=[(k + 3) % n] if k % 3 == 0 else None)
            cir.cnot(k % n, (k + 1) % n)
            cir.rx((k + 2) % n)
        a = torch.randn(4, 4, generator=g, dtype=torch.float64) + 1j * torch.randn(4, 4, generator=g, dtype=torch.float64)
        cir.any((torch.linalg.qr(a)[0] * (1 + 3e-5)).to(torch.complex64), wires=[1, n - 2])
        cir.rxlayer()
        cir.observable(0)
        cir.observable([1, n - 1], 'xz')
        assert sum(1 for op in cir.operators if getattr(op, '_exact_unitary', True) is False) == 11
        if device is not None:
            cir.to(device)
        return cir

    results = {}
    for mode in ('per_gate', 'adjoint'):
        dq.executor.CONFIG['grad_mode'] = mode
        try:
            cir = build()
            cir()
            loss = (cir.expectation() * torch.tensor([1.0, -0.7], device=cir.state.device)).sum()
            loss.backward()
            if mode == 'adjoint':
                assert dq.executor.LAST_SWEEP['fused']
            results[mode] = [p.grad.cpu() for p in cir.parameters()]
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
    worst = max((x - y).abs().max().item() for x, y in zip(results['per_gate'], results['adjoint'], strict=True))
    assert worst < tol, worst
    assert math.isfinite(worst)


def check_module_dtype_and_device(dq, device=None):
    """The reference's tests/test_module.py:7-42, the QubitCircuit part: a circuit inside an nn.Module follows
    .double() (every buffer float64 / complex128) and .to(device), and still runs afterwards."""
    from torch import nn

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.cir_qubit = dq.QubitCircuit(1)
            self.cir_qubit.h(0)
            self.cir_qubit.hamiltonian([[1, 'x0'], [1, 'y0'], [1, 'z0']], encode=True)
            self.cir_qubit.observable(0)
            self.cir_two = dq.QubitCircuit(3)
            self.cir_two.hlayer()
            self.cir_two.rxlayer(encode=True)
            self.cir_two.cnot_ring()
            self.cir_two.observable(1)

    model = Model()
    model.double()
    assert len(list(model.buffers())) > 0
    for buffer in model.buffers():
        assert buffer.dtype in (torch.double, torch.cdouble)
    if device is not None:
        model.to(device)
        for buffer in model.buffers():
            assert buffer.device.type == torch.device(device).type
    data = torch.tensor([0.3, 0.2, 0.1], dtype=torch.double, device=device)
    state = model.cir_two(data=data)
    assert state.dtype == torch.cdouble and (device is None or state.device.type == torch.device(device).type)
    ev = model.cir_two.expectation()
    # H, Rx(t), CNOT ring: <Z1> by hand from the oracle-free closed form is not needed -- float32 and float64 agree
    model.float()
    for buffer in model.buffers():
        assert buffer.dtype in (torch.float, torch.cfloat)
    state32 = model.cir_two(data=data.float())
    assert state32.dtype == torch.cfloat
    assert (state32.to(torch.cdouble) - state).abs().max().item() < 1e-6
    assert (model.cir_two.expectation().double() - ev).abs().max().item() < 1e-6


# made with the real reference (tests/golden/make_golden.py:import_reference): QubitCircuit(10), per wire h / rx / ry / rz
# (encode=True), cnot_ring, data = torch.randn(4, 30, generator=manual_seed(3)); get_amplitude('0101010101')
GET_AMPLITUDE_REF = [(2.5445e-05, 4.693e-05), (-0.001651335, 0.003223858), (0.001179231, 0.000622286),
                     (0.012580904, 0.012054744)]


def check_get_amplitude(dq, device=None):
    """The reference's tests/test_get_amplitude.py:6-34 (its dense half; the MPS half is out of scope): batched data,
    one amplitude per sample, against the values the reference returns for the same seed."""
    n = 10
    data = torch.randn(4, 3 * n, generator=torch.Generator().manual_seed(3))
    cir = dq.QubitCircuit(nqubit=n)
    for i in range(n):
        cir.h(i)
        cir.rx(i, encode=True)
        cir.ry(i, encode=True)
        cir.rz(i, encode=True)
    cir.cnot_ring()
    if device is not None:
        cir.to(device)
        data = data.to(device)
    cir(data=data)
    amp = cir.get_amplitude('0101010101').reshape(-1).cpu()
    ref = torch.tensor([complex(*v) for v in GET_AMPLITUDE_REF], dtype=amp.dtype)
    assert amp.shape == (4,)
    assert (amp - ref).abs().max().item() < 1e-6


_GOLDEN_HESSIAN = None


def gold_hessian(key):
    global _GOLDEN_HESSIAN
    if _GOLDEN_HESSIAN is None:
        _GOLDEN_HESSIAN = np.load(os.path.join(HERE, 'golden', 'golden_hessian.npz'))
    return torch.from_numpy(_GOLDEN_HESSIAN[key])


def check_hessian_benchmark_against_golden(dq, n, layer, prec, tag, device=None):
    """The reference's own Hessian benchmark (examples/benchmarks/gradient_benchmark.py:147-163), taken the way it takes
    it -- ``torch.autograd.functional.hessian`` over a function that builds the circuit -- in whatever
    ``executor.CONFIG['grad_mode']`` is current, against the real reference's value, gradient and Hessian."""
    from torch.autograd.functional import hessian

    key = f'hessian/{n}-{layer}/{prec}/{tag}'
    x = gold_hessian(f'{key}/params')
    if device is not None:
        x = x.to(device)

    def f(params):
        cir = specs.hessian_benchmark_circuit(dq, n, layer)
        if device is not None:
            cir.to(device)
        if prec == 'c128':
            cir.to(torch.double)
        cir(data=params)
        return cir.expectation()

    xg = x.clone().requires_grad_(True)
    val = f(xg)
    (g,) = torch.autograd.grad(val, xg)
    h = hessian(f, x).reshape(x.numel(), x.numel())
    tol = TOL[prec]
    assert (val.detach().cpu() - gold_hessian(f'{key}/value')).abs().max().item() < tol, key
    assert (g.cpu() - gold_hessian(f'{key}/grad')).abs().max().item() < tol, key
    err = (h.cpu() - gold_hessian(f'{key}/hessian')).abs().max().item()
    assert err < tol, (key, err)
    return err


def check_hessian_params_against_golden(dq, prec, device=None):
    """Hessian with respect to nn.Parameters and data of a circuit with controlled / two-target trainable gates and
    three observables, by double ``torch.autograd.grad`` (tests/golden/make_golden_hessian.py)."""
    key = f'hessian_params/{prec}'
    cir = specs.hessian_param_circuit(dq, 4)
    if device is not None:
        cir.to(device)
    if prec == 'c128':
        cir.to(torch.double)
    params = gold_hessian(f'{key}/params')
    off = 0
    with torch.no_grad():
        for p in cir.parameters():
            p.copy_(params[off : off + p.numel()].reshape(p.shape))
            off += p.numel()
    data = gold_hessian(f'{key}/data').clone()
    if device is not None:
        data = data.to(device)
    data.requires_grad_(True)
    cir(data=data)
    ev = cir.expectation()
    tol = TOL[prec]
    assert (ev.detach().cpu() - gold_hessian(f'{key}/expectation')).abs().max().item() < tol
    w = torch.tensor(specs.HESSIAN_PARAM_WEIGHTS, dtype=ev.dtype, device=ev.device)
    loss = (ev.reshape(-1) * w).sum() + (ev.reshape(-1) ** 2).sum()
    leaves = [data] + list(cir.parameters())
    gs = torch.autograd.grad(loss, leaves, create_graph=True)
    gflat = torch.cat([g.reshape(-1) for g in gs])
    assert (gflat.detach().cpu() - gold_hessian(f'{key}/grad')).abs().max().item() < tol
    rows = []
    for i in range(gflat.numel()):
        r = torch.autograd.grad(gflat[i], leaves, retain_graph=True, allow_unused=True)
        rows.append(torch.cat([(torch.zeros_like(p) if x is None else x).reshape(-1) for x, p in zip(r, leaves)]))
    h = torch.stack(rows)
    err = (h.cpu() - gold_hessian(f'{key}/hessian')).abs().max().item()
    assert err < tol, (key, err)
    return err


def check_hvp_random(dq, device=None, n=6, batch=1, seed=0, ngates=40, tol=1e-9, dtype=torch.float64):
    """Fuzz of the second-order routes: Hessian-vector products of a random circuit over the whole gate menu -- fixed,
    trainable and encoded gates, controls of every arity, diagonal and two-target trainable gates -- with respect to its
    parameters AND its data, by the tangent circuit (executor._SweepGrads) and by the per-gate replay."""
    import random

    def build():
        rng = random.Random(seed)
        torch.manual_seed(seed)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        for _ in range(ngates):
            kind = rng.choice(['h', 'x', 'y', 'z', 's', 't', 'rx', 'ry', 'rz', 'p', 'u3', 'cnot', 'cz', 'crx', 'cry', 'crz',
                               'toffoli', 'rzz_enc', 'rx_enc', 'ry_ctrl', 'u3_ctrl2', 'cp', 'swap', 'rxx_enc', 'fredkin',
                               'rxx_train', 'ryy_train_ctrl', 'rzz_train', 'u3_enc', 'rzz_data'])
            w = rng.sample(range(n), 3)
            if kind in ('h', 'x', 'y', 'z', 's', 't'):
                getattr(cir, kind)(w[0], controls=[w[1]] if rng.random() < 0.2 else None)
            elif kind in ('rx', 'ry', 'rz', 'p', 'u3'):
                getattr(cir, kind)(w[0])
            elif kind in ('cnot', 'cz', 'crx', 'cry', 'crz', 'cp'):
                getattr(cir, kind)(w[0], w[1])
            elif kind == 'toffoli':
                cir.toffoli(w[0], w[1], w[2])
            elif kind == 'fredkin':
                cir.fredkin(w[0], w[1], w[2])
            elif kind == 'swap':
                cir.swap([w[0], w[1]])
            elif kind == 'rxx_train':
                cir.rxx([w[0], w[1]])
            elif kind == 'ryy_train_ctrl':
                cir.ryy([w[0], w[1]], controls=[w[2]])
            elif kind == 'rzz_train':
                cir.rzz([w[0], w[1]])
            elif kind == 'rxx_enc':
                cir.rxx([w[0], w[1]], inputs=rng.uniform(0.0, 6.0))
            elif kind == 'rzz_enc':
                cir.rzz([w[0], w[1]], inputs=rng.uniform(0.0, 6.0))
            elif kind == 'rx_enc':
                cir.rx(w[0], encode=True)
            elif kind == 'u3_enc':
                cir.u3(w[0], encode=True)
            elif kind == 'rzz_data':
                cir.rzz([w[0], w[1]], encode=True)
            elif kind == 'ry_ctrl':
                cir.ry(w[0], controls=[w[1]])
            else:
                cir.u3(w[0], controls=[w[1], w[2]])
        cir.observable(0)
        cir.observable([1, n - 1], 'xz')
        if device is not None:
            cir.to(device)
        if dtype == torch.float64:
            cir.to(torch.double)
        return cir

    results = {}
    for mode in ('tangent', 'replay'):
        dq.executor.CONFIG['second_order'] = mode
        try:
            rows = dq.executor.GRAPH_BACKWARDS['tangent_rows']
            cir = build()
            g = torch.Generator().manual_seed(200 + seed)
            data = torch.rand(batch, max(cir.ndata, 1), generator=g, dtype=dtype) * 6.0
            if device is not None:
                data = data.to(device)
            data.requires_grad_(True)
            leaves = [data] + list(cir.parameters()) if cir.ndata else list(cir.parameters())
            cir(data=data if cir.ndata else None)
            loss = (cir.expectation() * torch.tensor([1.0, -0.7], device=data.device, dtype=dtype)).sum()
            first = torch.autograd.grad(loss, leaves, create_graph=True, allow_unused=True)
            vs = [torch.randn(f.shape, generator=g, dtype=dtype).to(f.device) for f in first if f is not None]
            dot = sum((f * v).sum() for f, v in zip([f for f in first if f is not None], vs, strict=True))
            second = torch.autograd.grad(dot, leaves, allow_unused=True)
            assert (dq.executor.GRAPH_BACKWARDS['tangent_rows'] > rows) == (mode == 'tangent'), mode
            results[mode] = [None if s_ is None else s_.detach().cpu() for s_ in second]
        finally:
            dq.executor.CONFIG['second_order'] = 'tangent'
    scale = max(1.0, max(r.abs().max().item() for r in results['replay'] if r is not None))
    for a, b in zip(results['tangent'], results['replay'], strict=True):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max().item() < tol * scale, (seed, (a - b).abs().max().item(), scale)


def check_transforms_random(dq, device=None, n=10, seed=0, ngates=40, dtype=torch.float32, tol=5e-5):
    """Fuzz of the fused node under ``torch.func`` transforms (executor._FusedCircuit / _FusedSweep): a random circuit over
    the gate menu whose parametric gates are all data-encoded -- ``vmap`` over the circuit against the native batch,
    ``jacrev`` against ``torch.autograd.functional.jacobian``, ``vmap(grad)`` against a loop, ``vmap(jacrev)``."""
    import random

    import torch.func as tf

    rng = random.Random(seed)
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    for _ in range(ngates):
        kind = rng.choice(['h', 'x', 's', 't', 'cnot', 'cz', 'toffoli', 'swap', 'rx', 'ry', 'rz', 'u3', 'crx', 'cry', 'p', 'cp',
                           'rxx', 'ryy', 'rzz', 'rx2c'])
        w = rng.sample(range(n), 3)
        if kind in ('h', 'x', 's', 't'):
            getattr(cir, kind)(w[0], controls=[w[1]] if rng.random() < 0.2 else None)
        elif kind in ('cnot', 'cz'):
            getattr(cir, kind)(w[0], w[1])
        elif kind == 'toffoli':
            cir.toffoli(w[0], w[1], w[2])
        elif kind == 'swap':
            cir.swap([w[0], w[1]])
        elif kind in ('rx', 'ry', 'rz', 'u3', 'p'):
            getattr(cir, kind)(w[0], encode=True)
        elif kind in ('crx', 'cry', 'cp'):
            getattr(cir, kind)(w[0], w[1], encode=True)
        elif kind in ('rxx', 'ryy', 'rzz'):
            getattr(cir, kind)([w[0], w[1]], encode=True)
        else:
            cir.rx(w[0], controls=[w[1], w[2]], encode=True)
    cir.observable(0)
    cir.observable([1, n - 1], 'zz')
    cir.observable([2, 3], 'xy')
    if device is not None:
        cir.to(device)
    real = torch.float32
    if dtype == torch.float64:
        cir.to(torch.double)
        real = torch.float64
    if cir.ndata == 0:
        return
    g = torch.Generator().manual_seed(seed)
    data = (torch.rand(3, cir.ndata, generator=g, dtype=real) * 6.28).to(device)
    with torch.no_grad():
        native = cir(data).clone()
        vm = tf.vmap(cir._forward_helper, in_dims=(0, None))(data, cir.init_state.state)
    assert (vm.reshape(native.shape) - native).abs().max().item() < tol, ('vmap', seed)

    def fvec(p):
        cir(data=p)
        return cir.expectation().reshape(-1)

    x = data[0].clone()
    jac = torch.autograd.functional.jacobian(fvec, x)
    assert (tf.jacrev(fvec)(x) - jac).abs().max().item() < tol, ('jacrev', seed)
    want = torch.stack([torch.autograd.functional.jacobian(lambda p: fvec(p).sum(), r) for r in data])
    assert (tf.vmap(tf.grad(lambda p: fvec(p).sum()))(data) - want).abs().max().item() < tol, ('vmap(grad)', seed)
    wantj = torch.stack([torch.autograd.functional.jacobian(fvec, r) for r in data])
    assert (tf.vmap(tf.jacrev(fvec))(data) - wantj).abs().max().item() < tol, ('vmap(jacrev)', seed)
    if cir.ndata <= 48:       # two reverse levels: the tangent circuit inside the fused nodes (executor._FusedSweep.backward)
        fs = lambda p: (fvec(p) * torch.arange(1, 4, dtype=real, device=p.device)).sum()       # noqa: E731
        hes = torch.autograd.functional.hessian(fs, x)
        assert (tf.jacrev(tf.jacrev(fs))(x) - hes).abs().max().item() < 20 * tol, ('jacrev(jacrev)', seed)
        assert (tf.hessian(fs)(x) - hes).abs().max().item() < 40 * tol, ('hessian = jacfwd(jacrev)', seed)
        assert (tf.jacfwd(fvec)(x) - jac).abs().max().item() < 20 * tol, ('jacfwd', seed)


def pick_transport(world, device_count=None):
    """Transport of a sharded GPU test with ``world`` ranks: ``('nccl', one device per rank)`` -- RCCL over xGMI, the
    reference's own transport (communication.py:9-35, README.md:223-255) -- as soon as the box shows at least ``world``
    devices (a multi-GPU lease, or an MI355X in CPX mode: 8 devices), else ``('gloo', every rank on device 0)`` with
    host-staged exchanges.  Returns (backend, [device index of rank r])."""
    if device_count is None:
        device_count = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if world > 1 and device_count >= world and os.environ.get('DQ_TEST_TRANSPORT', '') != 'gloo':
        return 'nccl', list(range(world))
    return 'gloo', [0] * world
