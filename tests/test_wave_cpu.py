"""The wave-tile kernel without a GPU: the library's translation of a pass (rounds -> records: physical slots, trips,
pair masks, byte offsets; csrc/dq_wave.hip) executed by the emulator of tests/_wave_emulator.py -- which moves data with
the very instruction lists of the generator that writes the kernel -- against the descriptor interpreter of
tests/_cpu_backend.py and against the oracle applying the gates one by one."""

import random

import numpy as np
import pytest
import torch

from deepquantum_amd import _lib, backend, fusion
from oracle import statevec_oracle as oracle

import _wave_emulator as emu


def random_ops(n, ngates, seed, modes=True):
    """One-target gates with 0..3 controls anywhere; 'gen' matrices general / real / Rx-like / Hadamard-like."""
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    ops, mats, off = [], [], 0
    for _ in range(ngates):
        kind = rng.choice(['gen', 'gen', 'x', 'diag', 'diag2'])
        nc = rng.choice([0, 0, 0, 1, 1, 2, 3])
        k = 2 if kind == 'diag2' else 1
        bits = rng.sample(range(n), k + nc)
        mode = 0
        if kind in ('diag', 'diag2'):
            d = 1 << k
            m = torch.diag(torch.exp(1j * torch.rand(d, generator=g, dtype=torch.float64) * 6.28))
            ops.append(fusion.PrimOp('diag', tuple(bits[:k]), tuple(bits[k:]), off, 0))
            mats.append(m.reshape(-1))
            off += d * d
            continue
        if kind == 'x':
            m = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
        else:
            mode = rng.choice([0, 1, 2, 3]) if modes else 0
            th = float(torch.rand(1, generator=g, dtype=torch.float64)) * 6.28
            c, s_ = np.cos(th / 2), np.sin(th / 2)
            if mode == 1:
                m = torch.tensor([[c, -s_], [s_, c]], dtype=torch.complex128)
            elif mode == 2:
                m = torch.tensor([[c, -1j * s_], [-1j * s_, c]], dtype=torch.complex128)
            elif mode == 3:
                m = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) * (2 ** -0.5)
            else:
                a = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
                m, _ = torch.linalg.qr(a)
        ops.append(fusion.PrimOp(kind, (bits[0],), tuple(bits[1:]), off, mode))
        mats.append(m.reshape(-1))
        off += 4
    return ops, torch.cat(mats).to(torch.complex64)


def reference(state, ops, mats):
    x = state
    for op in ops:
        d = 1 << op.k
        x = oracle.apply_gate_bits(x, mats[op.mat:op.mat + d * d].reshape(d, d), list(op.targets), list(op.controls))
    return x


@pytest.mark.parametrize('prec', ['c64', 'c128'])
@pytest.mark.parametrize('n,ngates,seed,permute', [(12, 40, 0, False), (13, 120, 1, False), (14, 200, 2, False),
                                                   (15, 300, 3, True), (16, 260, 4, True), (14, 150, 5, True)])
def test_translated_passes_match_the_descriptor_semantics(cpu_backend, n, ngates, seed, permute, prec):
    is128 = prec == 'c128'
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, ngates, seed)
    mats = mats.to(dtype)
    geom = fusion.default_geometry(is128)
    assert geom.m == (11 if is128 else 12)
    geom.permute_store = permute
    geom.plan_min_bits = 12          # (plan the tiles as for big states, so that free low bits are exercised)
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    km = fusion.kernel_matrices(steps, ops, mats)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(dtype)
    ref = reference(x, ops, mats)
    cur_d, cur_e = x.clone(), x.numpy().copy()
    trips = 0
    tol = 1e-12 if is128 else 2e-6
    for st in steps:
        nxt = torch.empty_like(cur_d)
        backend.apply_fused(cur_d, km, 0, st.desc, out=nxt)           # the descriptor interpreter
        cur_d = nxt
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0)        # the library's records, emulated
        kp = emu.descriptor(st.desc, n)
        ids = [kp.rec[i][0] for i in range(kp.nrec_bytes // 32)]
        trips += sum(emu.gen(is128).ID_TRIP0 <= i < emu.gen(is128).ID_SWAP for i in ids)
        assert np.abs(cur_e - cur_d.numpy()).max() < tol
    assert (cur_d - ref).abs().max().item() < 10 * tol
    assert np.abs(cur_e - ref.numpy()).max() < 10 * tol
    assert trips > 0 or n <= 12


def test_trips_cover_every_slot_mask_and_stay_inside_the_buffer():
    """Every generated trip variant (k <= 4 outgoing slots, any mask) transposes a sub-tile correctly when driven the way
    the translator drives it: checked on the identity pass 'load layout -> a layout with other slots -> store layout'."""
    g = emu.gen()
    assert len(g.TRIP_MASKS) == 56 and g.MAXK == 4
    for mask in g.TRIP_MASKS:
        k = bin(mask).count('1')
        lines = [ln for ln in g.trip(k, mask) if ln.startswith('ds_')]
        assert len(lines) == 128
        a, S = 5 - k, 64 + (1 << (5 - k))
        imm_w = sorted({int(ln.split('offset:')[1]) for ln in lines if ln.startswith('ds_write')})
        imm_r = sorted({int(ln.split('offset:')[1]) for ln in lines if ln.startswith('ds_read')})
        assert imm_w == [8 * S * x for x in range(1 << k)] and imm_r == [8 * (y << a) for y in range(1 << k)]
        assert 8 * (S * ((1 << k) - 1) + 63) + 8 <= 8448


def test_everything_the_scheduler_fuses_runs_on_the_wave_tile(cpu_backend):
    """One pass kernel: one-target dense gates and X, diagonal gates on one or two targets, dense gates on two targets,
    in both precisions (complex128 two-target dense gates since round 4)."""
    ops = [fusion.PrimOp('gen', (3,), (), 0, 0), fusion.PrimOp('gen', (5, 7), (), 4, 0)]
    for is128 in (False, True):
        assert fusion.wave_supports(ops, is128)
        geom = fusion.default_geometry(is128)
        steps = fusion.schedule(ops, 13, geom)
        lib = _lib.load()
        import ctypes as C
        assert lib.dq_wave_descriptor(C.byref(steps[0].desc), 13, 0, None, 0) > 0
    assert fusion.wave_supports([fusion.PrimOp('diag', (5, 2), (1,), 0, 0)])


@pytest.mark.parametrize('n,seed,is128', [(12, 0, False), (13, 1, False), (15, 2, False), (11, 3, True), (13, 4, True)])
def test_reduction_records_of_the_reverse_sweep_translate(cpu_backend, n, seed, is128):
    """DQ_FG_GRAD records through the library's translation (psi / lambda slot brought to physical slot 0, group masks
    of the register controls, accumulator rows): emulator against the descriptor interpreter, states and sums."""
    rng = random.Random(seed)
    base_ops, base_mats = random_ops(n - 1, 60, seed)
    ops, mats, off, rows = [], [], 0, 0
    for op in base_ops:                       # index bit 0 tells psi from lambda: every gate moves up by one bit
        d = 1 << op.k
        if op.kind == 'gen' and rng.random() < 0.5:
            ctrl = tuple(c + 1 for c in rng.sample([c for c in range(n - 1) if c not in op.targets], rng.choice([0, 0, 1, 2])))
            ops.append(fusion.PrimOp('grad', (op.targets[0] + 1, 0), ctrl, 0, rows))
            rows += 1
        ops.append(fusion.PrimOp(op.kind, tuple(t + 1 for t in op.targets), tuple(c + 1 for c in op.controls), off, op.mode))
        mats.append(base_mats[op.mat:op.mat + d * d])
        off += d * d
    cdt = torch.complex128 if is128 else torch.complex64
    mats = torch.cat(mats).to(cdt)
    geom = fusion.default_geometry(is128)
    geom.plan_min_bits = 11
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    km = fusion.kernel_matrices(steps, ops, mats)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(cdt)
    acc_d = torch.zeros(2, rows, 8, dtype=torch.float64)
    acc_e = np.zeros((2, rows, 8))
    cur_d, cur_e = x.clone(), x.numpy().copy()
    for st in steps:
        backend.apply_fused(cur_d, km, 0, st.desc, out=cur_d, grads=acc_d)
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0, grads=acc_e)
    assert np.abs(cur_e - cur_d.numpy()).max() < (1e-13 if is128 else 2e-6)
    assert np.abs(acc_e - acc_d.numpy()).max() < (1e-12 if is128 else 2e-5) * max(1.0, float(acc_d.abs().max()))
    assert float(acc_d.abs().max()) > 0


def long_sweep_ops(n, ngates, seed):
    """A reverse sweep's gate list with a reduction in front of every dense one-target gate (bit 0 tells psi from lambda),
    scheduled with the record cap of a sweep (executor.CONFIG['sweep_max_gates']): passes of more than 112 records."""
    rng = random.Random(seed)
    base_ops, base_mats = random_ops(n - 1, ngates, seed)
    ops, mats, off, rows = [], [], 0, 0
    for op in base_ops:
        d = 1 << op.k
        if op.kind == 'gen':
            ctrl = tuple(c + 1 for c in rng.sample([c for c in range(n - 1) if c not in op.targets], rng.choice([0, 0, 0, 1])))
            variant = {2: rng.choice([2, 4]), 1: 1}.get(op.mode, 0)
            ops.append(fusion.PrimOp('grad', (op.targets[0] + 1, 0), ctrl, 0, rows | (variant << fusion.GRAD_VARIANT_SHIFT)))
            rows += 1
        ops.append(fusion.PrimOp(op.kind, tuple(t + 1 for t in op.targets), tuple(c + 1 for c in op.controls), off, op.mode))
        mats.append(base_mats[op.mat:op.mat + d * d])
        off += d * d
    return ops, torch.cat(mats), rows


def long_sweep_steps(ops, n, is128):
    geom = fusion.default_geometry(is128)
    geom.plan_min_bits = 11
    geom.max_gates = 104
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    return steps


@pytest.mark.parametrize('n,seed,is128,ngates', [(13, 0, False, 260), (14, 1, False, 260), (13, 2, True, 400), (14, 3, True, 400)])
def test_sweep_passes_with_more_records_than_the_kernel_arguments_hold(cpu_backend, n, seed, is128, ngates):
    """ABI 24: a pass of a reverse sweep may hold up to 104 gates + reductions; with its layout changes that is more than
    the 112 records of the kernel-argument segment, and the kernel reads them from device memory (dq_wave_records /
    dq_apply_fused_grad_ext_*).  Here: the records hook against the descriptor hook, the emulator (which executes the
    records) against the descriptor interpreter (which does not know about records), states and sums."""
    import ctypes as C

    ops, mats, rows = long_sweep_ops(n, ngates, seed)
    cdt = torch.complex128 if is128 else torch.complex64
    mats = mats.to(cdt)
    steps = long_sweep_steps(ops, n, is128)
    lib = _lib.load()
    sizes = []
    for st in steps:
        nb = lib.dq_wave_records(C.byref(st.desc), n, None, 0)
        assert nb > 0 and nb % 32 == 0
        buf = (C.c_uint8 * nb)()
        assert lib.dq_wave_records(C.byref(st.desc), n, buf, nb) == nb
        kp = emu.descriptor(st.desc, n)
        assert kp.nrec_bytes == nb and bytes(buf) == bytes(kp)[emu.WaveKernPass.rec.offset:emu.WaveKernPass.rec.offset + nb]
        sizes.append(nb // 32)
    assert max(sizes) > backend.KERNARG_RECORDS, sizes
    km = fusion.kernel_matrices(steps, ops, mats)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(cdt)
    acc_d = torch.zeros(2, rows, 8, dtype=torch.float64)
    acc_e = np.zeros((2, rows, 8))
    cur_d, cur_e = x.clone(), x.numpy().copy()
    for st in steps:
        backend.apply_fused(cur_d, km, 0, st.desc, out=cur_d, grads=acc_d)
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0, grads=acc_e)
    assert np.abs(cur_e - cur_d.numpy()).max() < (1e-13 if is128 else 2e-6)
    assert np.abs(acc_e - acc_d.numpy()).max() < (1e-12 if is128 else 2e-5) * max(1.0, float(acc_d.abs().max()))
    assert float(acc_d.abs().max()) > 0


@pytest.mark.parametrize('n,seed,is128', [(12, 0, False), (14, 1, False), (15, 2, False), (11, 3, True), (13, 4, True)])
def test_z_string_expectations_from_the_registers(cpu_backend, n, seed, is128):
    """DQ_FG_EXPZ records: <Z..Z> of several strings reduced inside the last pass, descriptor interpreter and emulator
    against numpy on the final state."""
    rng = random.Random(seed)
    ops, mats = random_ops(n, 70, seed)
    ops = list(ops)
    masks = [1 << (n - 1), 1, (1 << (n - 1)) | 1, sum(1 << q for q in rng.sample(range(n), 4)), (1 << n) - 1]
    for r, zm in enumerate(masks):
        ops.append(fusion.PrimOp('expz', (), tuple(q for q in range(n) if (zm >> q) & 1), 0, r, 0, tuple(range(n))))
    cdt = torch.complex128 if is128 else torch.complex64
    mats = mats.to(cdt)
    geom = fusion.default_geometry(is128)
    geom.plan_min_bits = 11
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    km = fusion.kernel_matrices(steps, ops, mats)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(cdt)
    acc_d = torch.zeros(2, len(masks), 8, dtype=torch.float64)
    acc_e = np.zeros((2, len(masks), 8))
    cur_d, cur_e = x.clone(), x.numpy().copy()
    for st in steps:
        backend.apply_fused(cur_d, km, 0, st.desc, out=cur_d, grads=acc_d)
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0, grads=acc_e)
    ref = reference(x, [op for op in ops if op.kind != 'expz'], mats)
    assert (cur_d - ref).abs().max().item() < (1e-10 if is128 else 1e-4)
    p = (ref.real.double() ** 2 + ref.imag.double() ** 2).numpy()
    idx = np.arange(1 << n)
    tol = 1e-12 if is128 else 2e-6
    for r, zm in enumerate(masks):
        sign = 1.0 - 2.0 * (np.array([bin(int(i) & zm).count('1') & 1 for i in idx]))
        want = (p * sign[None, :]).sum(-1)
        assert np.abs(acc_d[:, r, 0].numpy() - want).max() < tol, (r, acc_d[:, r, 0], want)
        assert np.abs(acc_e[:, r, 0] - want).max() < tol, (r, acc_e[:, r, 0], want)
    assert float(acc_d[:, :, 1:].abs().max()) == 0.0


@pytest.mark.parametrize('n,seed,is128', [(12, 0, False), (13, 1, False), (15, 2, False), (11, 3, True), (12, 4, True), (14, 5, True)])
def test_two_target_dense_gates_on_the_wave_tile_kernel(cpu_backend, n, seed, is128):
    """Dense 4x4 gates on two register slots (any target order, register / thread / outside controls) among the other
    records, both precisions -- the library's translation executed by the emulator, the descriptor interpreter and the
    oracle."""
    from test_fusion_cpu import random_ops as mixed_ops, run_reference

    ops, mats = mixed_ops(n, 60, seed, kinds=('gen', 'x', 'diag', 'gen2', 'gen2', 'gen2real', 'gen2x', 'gen2x', 'gen2xc', 'gen2xc', 'diag2'))
    cd = torch.complex128 if is128 else torch.complex64
    mats = mats.to(cd)
    assert fusion.wave_supports(ops, is128) and any(op.kind == 'gen' and op.k == 2 for op in ops)
    geom = fusion.default_geometry(is128)
    geom.plan_min_bits = geom.m
    steps = fusion.schedule(ops, n, geom)
    assert all(isinstance(s, fusion.FusedStep) and s.desc.slots == geom.slots for s in steps)
    km = fusion.kernel_matrices(steps, ops, mats)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(cd)
    ref = run_reference(x, ops, mats)
    cur_d, cur_e = x.clone(), x.numpy().copy()
    ngen2 = nreal = nx = nxc = 0
    gm = emu.gen(is128)
    xid = getattr(gm, 'ID_GEN2X', None)      # (complex128 has no X-shaped bodies: such a matrix takes the real ones)
    for st in steps:
        backend.apply_fused(cur_d, km, 0, st.desc, out=cur_d)
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0)
        kp = emu.descriptor(st.desc, n)
        ids = [kp.rec[i][0] for i in range(kp.nrec_bytes // 32)]
        ngen2 += sum(gm.ID_GEN2 <= i_ < gm.ID_GEN2R for i_ in ids)
        nreal += sum(gm.ID_GEN2R <= i_ < gm.ID_GEN2R + len(gm.SWAP_PAIRS) for i_ in ids)
        nx += 0 if xid is None else sum(xid <= i_ < xid + len(gm.SWAP_PAIRS) for i_ in ids)
        nxc += 0 if xid is None else sum(gm.ID_GEN2XC <= i_ < gm.ID_GEN2XC + len(gm.SWAP_PAIRS) for i_ in ids)
    count = lambda modes: sum(op.kind == 'gen' and op.k == 2 and op.mode in modes for op in ops)      # noqa: E731
    assert ngen2 == (count((0,)) if xid is not None else count((0, 5)))
    assert nxc == (count((5,)) if xid is not None else 0) and count((5,)) > 0
    assert nreal == (count((1,)) if xid is not None else count((1, 4))) > 0
    assert nx == (count((4,)) if xid is not None else 0) and count((4,)) > 0
    tol = 1e-12 if is128 else 1e-5
    assert (cur_d - ref).abs().max().item() < tol
    assert np.abs(cur_e - ref.numpy()).max() < tol


def test_the_cpu_double_runs_a_pass_in_slices():
    """The test double's reading of dq_apply_fused_slice_* (tests/_cpu_backend.py): the slices of a pass by index bits outside
    its tile are the pass -- what the sharded state's sliced passes rely on in the CPU suite."""
    from _cpu_backend import CpuTestBackend

    be = CpuTestBackend()
    n = 15
    ops, mats = random_ops(n, 150, 21)
    geom = fusion.default_geometry(False)
    geom.permute_store = True
    geom.plan_min_bits = 11
    steps = fusion.schedule(ops, n, geom)
    g = torch.Generator().manual_seed(5)
    cur = torch.randn(2, 1 << n, generator=g, dtype=torch.complex64)
    km = fusion.kernel_matrices(steps, ops, mats.to(torch.complex64))
    for st in steps:
        whole = torch.empty_like(cur)
        be.apply_fused(cur, km, 0, st.desc, whole)
        tile = set(range(st.desc.L)) | {st.desc.high_pos[i] for i in range(st.desc.h)}
        bits = [p for p in range(n - 1, -1, -1) if p not in tile][:2]
        mask = sum(1 << p for p in bits)
        parts = torch.full_like(cur, float('nan'))
        for v in range(4):
            be.apply_fused(cur, km, 0, st.desc, parts, slice_bits=(mask, sum(((v >> i) & 1) << bits[i] for i in range(2))))
        assert torch.equal(torch.view_as_real(parts), torch.view_as_real(whole))
        cur = whole
