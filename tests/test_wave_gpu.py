"""The wave-tile kernels (csrc/dq_wave.hip, the default geometry of both precisions) on the GPU, through the C ABI, against
the oracle applying the same gates one by one: 1e-4 (complex64) / 1e-10 (complex128) on amplitudes, the north star's bars
(measured ~1e-6 / ~1e-15)."""

import pytest
import torch

from deepquantum_amd import backend, fusion

from test_wave_cpu import long_sweep_ops, long_sweep_steps, random_ops, reference

pytestmark = pytest.mark.gpu
TOL = {False: 1e-4, True: 1e-10}
PREC = pytest.mark.parametrize('is128', [False, True], ids=['c64', 'c128'])


def dev():
    return torch.device('cuda', 0)


def cdtype(is128):
    return torch.complex128 if is128 else torch.complex64


def rand_state(b, n, seed, is128=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(b, 1 << n, generator=g, dtype=torch.float64)
    return (x / x.norm(dim=-1, keepdim=True)).to(cdtype(is128))


def wave_steps(ops, n, permute=False, is128=False):
    geom = fusion.default_geometry(is128)
    geom.permute_store = permute
    geom.plan_min_bits = 11
    steps = fusion.schedule(ops, n, geom)
    want = (5, 11) if is128 else (6, 12)
    assert all(isinstance(s, fusion.FusedStep) and (s.desc.slots, s.desc.m) == want for s in steps)
    return steps


@PREC
@pytest.mark.parametrize('n,ngates,seed', [(12, 60, 0), (13, 120, 1), (14, 200, 2), (15, 300, 3), (17, 300, 4), (19, 400, 5)])
def test_wave_passes_match_oracle_in_place(n, ngates, seed, is128):
    ops, mats = random_ops(n, ngates, seed)
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, is128=is128)
    x = rand_state(2, n, 10 + seed, is128)
    ref = reference(x, ops, mats)
    xd, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    for st in steps:
        backend.apply_fused(xd, md, 0, st.desc, out=xd)
    err = (xd.cpu() - ref).abs().max().item()
    assert err < TOL[is128], err


@PREC
@pytest.mark.parametrize('n,ngates,seed', [(14, 150, 5), (16, 260, 4), (18, 300, 6), (20, 400, 7)])
def test_wave_passes_with_permuted_stores(n, ngates, seed, is128):
    """Out-of-place passes that re-label index bits on the way out (every tile after the first is contiguous on the
    read side, the low bits move too): the store layout is reached by a trip, a lane permutation or a slot swap."""
    ops, mats = random_ops(n, ngates, seed)
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, permute=True, is128=is128)
    assert any(s.permutes for s in steps)
    x = rand_state(2, n, 20 + seed, is128)
    ref = reference(x, ops, mats)
    cur, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, md, 0, st.desc, out=nxt)
        cur = nxt
    err = (cur.cpu() - ref).abs().max().item()
    assert err < TOL[is128], err


@PREC
def test_wave_kernel_equals_its_cpu_emulation(is128):
    """Pass by pass against tests/_wave_emulator.py (the library's own records executed on the CPU)."""
    import _wave_emulator as emu

    n = 14
    ops, mats = random_ops(n, 200, 8)
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, permute=True, is128=is128)
    x = rand_state(2, n, 3, is128)
    km = fusion.kernel_matrices(steps, ops, mats)
    cur, md = x.to(dev()), km.to(dev())
    cur_e = x.numpy().copy()
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, md, 0, st.desc, out=nxt)
        cur = nxt
        cur_e = emu.run_pass(st.desc, n, cur_e, km.numpy(), 0)
        assert (cur.cpu() - torch.from_numpy(cur_e)).abs().max().item() < (1e-13 if is128 else 2e-6)


@PREC
@pytest.mark.parametrize('n,b', [(15, 3), (16, 16), (13, 4), (12, 5)])
def test_wave_batched_matrices_and_one_shared_input_state(n, b, is128):
    """Per-sample matrices (the vmap case of circuit.py:232-240) and the first pass of a batched circuit reading ONE
    input state (dq_apply_fused_bcast_c64)."""
    ops, mats0 = random_ops(n, 60, 21)
    g = torch.Generator().manual_seed(5)
    mats = mats0.unsqueeze(0).repeat(b, 1)
    for i, op in enumerate(ops):                 # per-sample Rx-like angles on the mode-2 gates, shared elsewhere
        if op.kind == 'gen' and op.mode == 2:
            th = torch.rand(b, generator=g, dtype=torch.float64) * 6.28
            c, s_ = torch.cos(th / 2), torch.sin(th / 2)
            m = torch.stack([c + 0j, -1j * s_, -1j * s_, c + 0j], dim=1).to(torch.complex64)
            mats[:, op.mat:op.mat + 4] = m
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, is128=is128)
    x1 = rand_state(1, n, 31, is128)
    ref = torch.cat([reference(x1, ops, mats[i]) for i in range(b)])
    km = fusion.kernel_matrices(steps, ops, mats).to(dev())
    out = torch.empty(b, 1 << n, dtype=cdtype(is128), device=dev())
    src = x1.to(dev())
    for k, st in enumerate(steps):
        if k == 0:
            backend.apply_fused(src, km.reshape(-1), km.shape[1], st.desc, out=out)     # one state in, b states out
        else:
            backend.apply_fused(out, km.reshape(-1), km.shape[1], st.desc, out=out)
    err = (out.cpu() - ref).abs().max().item()
    assert err < TOL[is128], err
    assert torch.equal(src.cpu(), x1)


@PREC
@pytest.mark.parametrize('n,seed', [(12, 0), (14, 1), (17, 2), (21, 3)])
def test_z_string_expectations_from_the_registers_on_gpu(n, seed, is128):
    """DQ_FG_EXPZ records on the kernel: <Z..Z> of several strings out of the last pass against float64 sums of the oracle's
    final state (complex64: float32 partial sums per workgroup, 1e-6)."""
    import numpy as np

    ops, mats = random_ops(n, 100, seed)
    ops = list(ops)
    masks = [1 << (n - 1), 1, (1 << (n - 1)) | 1, 0b1011 << (n // 2), (1 << n) - 1]
    for r, zm in enumerate(masks):
        ops.append(fusion.PrimOp('expz', (), tuple(q for q in range(n) if (zm >> q) & 1), 0, r, 0, tuple(range(n))))
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, permute=True, is128=is128)
    x = rand_state(2, n, 40 + seed, is128)
    ref = reference(x, [op for op in ops if op.kind != 'expz'], mats)
    cur, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    acc = torch.zeros(2, len(masks), 8, dtype=torch.float64, device=dev())
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, md, 0, st.desc, out=nxt, grads=acc)
        cur = nxt
    assert (cur.cpu() - ref).abs().max().item() < TOL[is128]
    p = (ref.real.double() ** 2 + ref.imag.double() ** 2)
    idx = torch.arange(1 << n)
    for r, zm in enumerate(masks):
        par = torch.zeros(1 << n, dtype=torch.long)
        for q in range(n):
            if (zm >> q) & 1:
                par ^= (idx >> q) & 1
        want = (p * (1.0 - 2.0 * par)[None, :]).sum(-1)
        assert (acc[:, r, 0].cpu() - want).abs().max().item() < (1e-12 if is128 else 1e-6), (r, acc[:, r, 0], want)
    assert float(acc[:, :, 1:].abs().max()) == 0.0


@PREC
@pytest.mark.parametrize('n,ngates,seed,permute', [(12, 60, 0, False), (14, 150, 1, True), (17, 250, 2, True), (20, 300, 3, True)])
def test_two_target_dense_gates_on_the_wave_tile_kernel_on_gpu(n, ngates, seed, permute, is128):
    """4x4 gates on two register slots (DQ_FG_GEN2 on the wave-tile geometries; complex128: the matrix passes through the
    scalar registers two rows at a time) with controls of every kind, among one-target, X and diagonal gates, against the
    oracle."""
    from test_fusion_cpu import random_ops as mixed_ops, run_reference

    ops, mats = mixed_ops(n, ngates, seed, kinds=('gen', 'x', 'diag', 'gen2', 'gen2', 'gen2real', 'gen2x', 'gen2x', 'gen2xc', 'gen2xc', 'diag2'))
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, permute=permute, is128=is128)
    x = rand_state(2, n, 50 + seed, is128)
    ref = run_reference(x, ops, mats)
    cur, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    for st in steps:
        nxt = torch.empty_like(cur) if permute else cur
        backend.apply_fused(cur, md, 0, st.desc, out=nxt)
        cur = nxt
    assert (cur.cpu() - ref).abs().max().item() < TOL[is128]


@PREC
@pytest.mark.parametrize('n,seed', [(13, 0), (14, 4), (15, 5)])
def test_sweep_passes_with_their_records_in_device_memory_on_gpu(n, seed, is128):
    """ABI 24 on the GPU: passes of a reverse sweep with more than 112 records (dq_apply_fused_grad_ext_*: the kernel reads
    them from device memory) against the same passes on the descriptor interpreter -- states and the reductions' sums."""
    from _cpu_backend import CpuTestBackend

    ops, mats, rows = long_sweep_ops(n, 400, seed)
    mats = mats.to(cdtype(is128))
    steps = long_sweep_steps(ops, n, is128)
    km = fusion.kernel_matrices(steps, ops, mats)
    x = rand_state(2, n, seed, is128)
    cur, acc = x.to(dev()), torch.zeros(2, rows, 8, dtype=torch.float64, device=dev())
    md = km.to(dev())
    ext = 0
    for st in steps:
        backend.apply_fused(cur, md, 0, st.desc, out=cur, grads=acc)
        ext += any(t is not None for t in st.desc.__dict__.get('_dev_records', {}).values())
    assert ext > 0, 'no pass took the device-memory records'
    double = CpuTestBackend()
    ref, racc = x.clone(), torch.zeros(2, rows, 8, dtype=torch.float64)
    for st in steps:
        double.apply_fused(ref, km, 0, st.desc, ref, grads=racc)
    assert (cur.cpu() - ref).abs().max().item() < (1e-12 if is128 else 3e-6)
    scale = max(1.0, float(racc.abs().max()))
    assert (acc.cpu() - racc).abs().max().item() < (1e-11 if is128 else 3e-5) * scale
    assert float(racc.abs().max()) > 0


@PREC
@pytest.mark.parametrize('b,n', [(1, 1), (1, 12), (3, 13), (2, 24)])
def test_interleave_and_back_on_gpu(b, n, is128):
    """dq_interleave_* / dq_deinterleave_*: psi and lambda side by side along a new index bit 0 (the pair state of a fused
    reverse sweep) -- bit-exact against torch.stack."""
    x, y = rand_state(b, n, 1, is128).to(dev()), rand_state(b, n, 2, is128).to(dev())
    pair = backend.interleave(x, y)
    assert torch.equal(pair, torch.stack([x, y], dim=-1).reshape(b, -1))
    assert torch.equal(backend.deinterleave(pair, 0), x) and torch.equal(backend.deinterleave(pair, 1), y)


def slice_choices(desc, n, nbits):
    """``nbits`` index bits (read side) outside the tile of ``desc`` -- the highest ones -- as a mask."""
    tile = set(range(desc.L)) | {desc.high_pos[i] for i in range(desc.h)}
    free = [p for p in range(n - 1, -1, -1) if p not in tile]
    return sum(1 << p for p in free[:nbits]), free[:nbits]


@PREC
@pytest.mark.parametrize('n,ngates,seed,nbits', [(15, 200, 11, 1), (17, 300, 12, 2), (19, 300, 13, 2)])
def test_a_pass_in_slices_is_the_pass(n, ngates, seed, nbits, is128):
    """dq_apply_fused_slice_* (ABI 25): the 2^b launches that hold b index bits outside the tile at every value are the
    whole pass, bit for bit -- out-of-place passes with permuted stores (the held bits land elsewhere on the write side),
    gates controlled by a held bit included -- and a single slice writes nothing outside its own tiles."""
    ops, mats = random_ops(n, ngates, seed)
    mats = mats.to(cdtype(is128))
    steps = wave_steps(ops, n, permute=True, is128=is128)
    x = rand_state(2, n, 30 + seed, is128)
    cur, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    controlled = 0
    for st in steps:
        whole = torch.empty_like(cur)
        backend.apply_fused(cur, md, 0, st.desc, out=whole)
        mask, bits = slice_choices(st.desc, n, nbits)
        controlled += sum(1 for gi in range(st.desc.rounds[st.desc.nrounds - 1].gate_end)
                          if st.desc.gates[gi].out_cmask & mask)
        parts = torch.full_like(cur, float('nan'))
        for v in range(1 << nbits):
            value = sum(((v >> i) & 1) << bits[i] for i in range(nbits))
            before = parts.clone()
            backend.apply_fused(cur, md, 0, st.desc, out=parts, slice_bits=(mask, value))
            changed = (torch.view_as_real(parts) != torch.view_as_real(before)) & ~(torch.isnan(torch.view_as_real(parts)) & torch.isnan(torch.view_as_real(before)))
            assert int(changed.any(dim=-1).sum()) <= parts.numel() >> nbits            # (only its own tiles)
        assert torch.equal(torch.view_as_real(parts), torch.view_as_real(whole))
        cur = whole
    assert controlled > 0, 'no gate of any pass was controlled by a held bit: the case the kernel change is for'
    with pytest.raises(RuntimeError):           # a held bit inside the tile is refused
        backend.apply_fused(x.to(dev()), md, 0, steps[0].desc, out=torch.empty_like(cur), slice_bits=(1 << steps[0].desc.high_pos[0], 0))
