"""A circuit started from its own |0..0> (the reference's default, circuit.py:49): index bits no pass has had in its tile
yet are known to be |0>, and the first passes neither read, compute nor write where one of them is 1
(fusion.zero_state_masks, include/dq_hip.h dq_apply_fused_zext_*).  Here without a GPU: the masks of a schedule, the
kernel's records under the emulator (poisoned input, sentinel output), the executor on the CPU double (which poisons
what the kernel would leave untouched), and who may claim that a state is |0..0>."""

import pickle

import numpy as np
import pytest
import torch

import deepquantum_amd as dq
from deepquantum_amd import _lib, backend, executor, fusion

import _wave_emulator as emu
from test_wave_cpu import random_ops, reference


def _schedule(n, ngates, seed, is128, permute=True):
    ops, mats = random_ops(n, ngates, seed)
    geom = fusion.default_geometry(is128)
    geom.permute_store = permute
    geom.plan_min_bits = 12
    steps = fusion.schedule(ops, n, geom)
    return ops, mats.to(torch.complex128 if is128 else torch.complex64), steps


@pytest.mark.parametrize('is128', [False, True])
def test_masks_of_a_schedule(cpu_backend, is128):
    n = 17
    ops, _mats, steps = _schedule(n, 260, 11, is128)
    assert all(isinstance(s, fusion.FusedStep) for s in steps) and len(steps) >= 3
    masks = fusion.zero_state_masks(steps, n)
    assert masks is not None and len(masks) == len(steps)
    L = steps[0].desc.L
    assert masks[0] == ((1 << n) - 1) & ~((1 << L) - 1)          # nothing but the contiguous low bits can be non-zero
    live = [n - bin(k).count('1') for k in masks]
    assert all(a <= b for a, b in zip(live, live[1:])) and masks[-1] == 0
    m = steps[0].desc.m
    assert live[1] <= m                                           # after one pass: its tile
    # a gate that runs on its own reads the whole buffer: no masks while bits are left
    assert fusion.zero_state_masks([fusion.SingleStep(0)] + list(steps), n) is None
    # a schedule that never brings a bit into a tile would leave uninitialised memory in the result
    assert fusion.zero_state_masks(list(steps[:1]), n) is None


@pytest.mark.parametrize('prec', ['c64', 'c128'])
@pytest.mark.parametrize('n,ngates,seed', [(14, 120, 3), (16, 240, 4), (18, 300, 5)])
def test_records_with_known_zero_bits_under_the_emulator(cpu_backend, n, ngates, seed, prec):
    """The passes of a schedule on |0..0>, each with its mask: the emulator executes the library's descriptor (tile count,
    the loads that are left out) on an input that is NaN wherever a known-zero bit is 1, into an output full of a
    sentinel; what it writes must equal the ordinary pass on the real state, and the sentinel must survive exactly
    where a known-zero bit outside the tile is 1 (at its write position)."""
    is128 = prec == 'c128'
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats, steps = _schedule(n, ngates, seed, is128)
    masks = fusion.zero_state_masks(steps, n)
    assert masks is not None and sum(1 for k in masks if k) >= 2
    km = fusion.kernel_matrices(steps, ops, mats)
    x = torch.zeros(1, 1 << n, dtype=dtype)
    x[0, 0] = 1
    ref = reference(x, ops, mats)
    idx = np.arange(1 << n, dtype=np.int64)
    cur = x.numpy().copy()                      # what is in memory: poisoned where nothing was written
    full = x.clone()                            # the real state, pass by pass (the descriptor interpreter, no masks)
    sentinel = complex(7.0, -7.0)
    tol = 1e-12 if is128 else 2e-6
    for st, kz in zip(steps, masks):
        nxt = torch.empty_like(full)
        backend.apply_fused(full, km, 0, st.desc, out=nxt)
        full = nxt
        if kz:
            assert np.all(full.numpy()[0][_written_dead(st, kz, n, idx)] == 0)       # what is skipped IS zero
            poisoned = cur.copy()
            poisoned[0][(idx & kz) != 0] = complex(float('nan'), float('nan'))
            out = np.full_like(cur, sentinel)
            emu.run_pass(st.desc, n, poisoned, km.numpy(), 0, known_zero=kz, out=out)
            untouched = _written_dead(st, kz, n, idx)
            assert np.all(out[0][untouched] == sentinel) and not np.any(out[0][~untouched] == sentinel)
            assert np.abs(out[0][~untouched] - full.numpy()[0][~untouched]).max() < tol
            cur = out
        else:
            assert not np.any(cur == sentinel) and not np.any(np.isnan(cur))          # every bit has been in a tile
            cur = emu.run_pass(st.desc, n, cur, km.numpy(), 0)
            assert np.abs(cur - full.numpy()).max() < tol
    assert np.abs(cur - ref.numpy()).max() < 20 * tol


def _written_dead(st, kz, n, idx):
    """Indices (write side of the pass) where a known-zero bit OUTSIDE the tile is 1: the kernel leaves them alone."""
    d = st.desc
    tile = set(range(d.L)) | {d.high_pos[i] for i in range(d.h)}
    blk = [p for p in range(d.L, n) if p not in tile]
    wmask = 0
    for j, p in enumerate(blk):
        if (kz >> p) & 1:
            wmask |= 1 << d.store_blk_pos[j]
    return (idx & wmask) != 0


def test_library_refuses_masks_it_cannot_honour(cpu_backend):
    ops, _mats, steps = _schedule(14, 60, 1, False)
    lib = _lib.load()
    import ctypes as C

    d = steps[0].desc
    assert lib.dq_wave_descriptor(C.byref(d), 14, 0, None, 0) > 0
    assert lib.dq_wave_descriptor(C.byref(d), 14, ((1 << 14) - 1) & ~((1 << d.L) - 1), None, 0) > 0
    kp = emu.descriptor(d, 14, ((1 << 14) - 1) & ~((1 << d.L) - 1))
    assert kp.zext & 63 == 0                     # one tile: every bit of the tile number is known to be zero


def _circuit(n, depth, seed, batch, dtype):
    import bench

    spec = bench.random_circuit_spec(n, depth, seed=seed)
    return bench.build_circuit(dq, n, spec, batch, dtype, 'cpu')


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
@pytest.mark.parametrize('n,batch', [(15, None), (16, 3)])
def test_circuit_from_its_own_zero_state_skips_what_is_zero(cpu_backend, dtype, n, batch):
    """On/off equality through the executor.  The CPU double writes NaN wherever the kernel would leave the output
    untouched, so a mask that claims too little or too much shows."""
    cir, data = _circuit(n, 8, 5, batch, dtype)
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        with torch.no_grad():
            executor.CONFIG['zero_state'] = True
            a = cir(data=data).clone()
            assert executor.LAST_RUN['zero_passes'] >= 2 and executor.LAST_RUN['passes'] > executor.LAST_RUN['zero_passes']
            ea = cir.expectation().clone()
            executor.CONFIG['zero_state'] = False
            b = cir(data=data).clone()
            assert executor.LAST_RUN['zero_passes'] == 0
            eb = cir.expectation().clone()
        assert torch.equal(a, b) and torch.equal(ea, eb)
        # a state the caller supplies is nobody's |0..0>, even if it is one
        executor.CONFIG['zero_state'] = True
        psi0 = torch.zeros(1 << n, 1, dtype=dtype)
        psi0[0] = 1
        with torch.no_grad():
            c = cir(data=data, state=psi0)
        assert executor.LAST_RUN['zero_passes'] == 0 and torch.equal(c.reshape(a.shape), a)
    finally:
        executor.CONFIG.update(old)


def test_gradients_from_the_zero_state(cpu_backend):
    """The adjoint node's forward takes the same route (its reverse sweep works on full states)."""
    n = 14
    cir, data = _circuit(n, 6, 9, None, torch.complex64)
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
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
        assert torch.allclose(grads[0], grads[1], atol=1e-6)
    finally:
        executor.CONFIG.update(old)


def test_who_may_say_a_state_is_zero():
    qs = dq.QubitState(5)
    assert qs.is_zero_state()
    qs.to(torch.double)
    assert qs.is_zero_state() and qs.state.dtype == torch.complex128
    clone = pickle.loads(pickle.dumps(qs))
    assert not clone.is_zero_state()                   # (nobody vouches for what was restored)
    qs.state[3] = 0.5                                  # written into: not the constructor's |0..0> any more
    assert not qs.is_zero_state()
    assert not dq.QubitState(5, state='equal').is_zero_state()
    assert not dq.QubitState(2, state=[1, 0, 0, 0]).is_zero_state()
    qs2 = dq.QubitState(4)
    qs2.state = torch.zeros(16, 1, dtype=torch.cfloat)        # replaced
    assert not qs2.is_zero_state()
    qs4 = dq.QubitState(4)
    qs4.state.data = torch.ones(16, 1, dtype=torch.cfloat)   # (no version bump, but another storage)
    assert not qs4.is_zero_state()
    with torch.inference_mode():
        qs3 = dq.QubitState(3)
    assert not qs3.is_zero_state()                     # (an inference tensor has no version counter to watch)
    rho = dq.QubitState(3, den_mat=True)
    assert rho.is_zero_state() and rho.state.shape == (8, 8)
    cir = dq.QubitCircuit(4)
    assert cir.init_state.is_zero_state()
    cir.to(torch.double)
    assert cir.init_state.is_zero_state()
    # writes the version counter does not see: the claim follows who HOLDS the buffer, and comes back only
    # after a look at the memory
    qs5 = dq.QubitState(6)
    held = qs5.state
    assert not qs5.is_zero_state()                     # somebody outside holds the tensor: they may write any time
    del held
    assert qs5.is_zero_state()                         # nobody does any more, and the memory still is |0..0>
    qs5.state.data.zero_()
    qs5.state.data[1 << 4] = 1                         # (.data: no version bump, the temporaries are gone afterwards)
    assert not qs5.is_zero_state()
    qs5.state.data.zero_()
    qs5.state.data[0] = 1
    assert qs5.is_zero_state()                         # verified again
    alias = qs5.state.numpy()
    assert not qs5.is_zero_state()
    alias[5] = 1
    del alias
    assert not qs5.is_zero_state()
    sd = qs5.state_dict()                              # (a state dict holds aliases of the buffers)
    qs6 = dq.QubitState(6)
    _ = list(qs6.buffers())
    assert not qs6.is_zero_state()
    del _
    assert qs6.is_zero_state()
    qs6.invalidate()
    assert qs6.is_zero_state()                         # (memory untouched: verified and taken up again)
    del sd


def test_a_buffer_tampered_with_behind_the_version_counter_is_not_taken_for_zero(cpu_backend):
    """The advisor's reproduction: ``init_state.state.data`` rewritten to another basis state -- the forward must give
    what the run without the known-zero passes gives."""
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        res = []
        for on in (True, False):
            executor.CONFIG['zero_state'] = on
            n = 14
            cir = dq.QubitCircuit(n)
            for q in range(n):
                cir.h(q)
                cir.rx(q, inputs=0.1 * (q + 1))
            for q in range(n - 1):
                cir.cnot(q, q + 1)
            cir.observable(1)
            cir.init_state.state.data.zero_()
            cir.init_state.state.data[1 << 12] = 1
            with torch.no_grad():
                res.append(cir().clone())
            assert executor.LAST_RUN.get('zero_passes', 0) == 0
        assert torch.allclose(res[0], res[1], atol=1e-6)
        assert abs(res[0].abs().pow(2).sum().item() - 1) < 1e-4
    finally:
        executor.CONFIG.update(old)


def test_circuits_the_masks_do_not_apply_to_fall_back(cpu_backend):
    """A qubit no gate ever touches (bits would be left at the end), and a gate that runs on its own while bits are left
    (it reads the whole buffer): the step runs as if nobody knew about |0..0>, same results."""
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        for kind in ('untouched', 'single'):
            res = []
            for on in (True, False):
                executor.CONFIG['zero_state'] = on
                n = 14
                cir = dq.QubitCircuit(n)
                if kind == 'single':
                    g = torch.Generator().manual_seed(1)
                    u, _ = torch.linalg.qr(torch.randn(8, 8, generator=g, dtype=torch.complex64))
                    cir.any(u, [0, 5, 9])                       # three targets: not a record of the pass kernel
                for q in range(n - 1 if kind == 'untouched' else n):
                    cir.h(q)
                    cir.rx(q, inputs=0.1 * (q + 1))
                for q in range(n - 2):
                    cir.cnot(q, q + 1)
                for q in range(n - 1 if kind == 'untouched' else n):
                    cir.ry(q, inputs=0.2 * (q + 1))
                cir.observable(1)
                with torch.no_grad():
                    out = cir().clone()
                    res.append((out, cir.expectation().clone(), executor.LAST_RUN['zero_passes']))
            assert res[0][2] == 0 and res[1][2] == 0, kind
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
            assert not torch.isnan(res[0][0].real).any()
    finally:
        executor.CONFIG.update(old)
