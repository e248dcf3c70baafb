"""Parity of the HIP kernels (through the C ABI / ctypes) against the CPU oracle on seeded inputs.

Tolerances are the north-star ones: 1e-10 for complex128, 1e-4 for complex64 (absolute, on amplitudes of
a normalised state and on expectation values); actual errors are ~1e-15 / ~1e-6."""

import random

import pytest
import torch

from deepquantum_amd import _lib, backend, fusion
from oracle import statevec_oracle as oracle
from test_fusion_cpu import random_ops, run_reference

pytestmark = pytest.mark.gpu

TOL = {torch.complex64: 1e-4, torch.complex128: 1e-10}


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda', 0)


def rand_state(b, n, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(b, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(b, 1 << n, generator=g, dtype=torch.float64)
    return (s / s.norm(dim=-1, keepdim=True)).to(dtype)


def rand_unitary(k, dtype, seed, batch=None):
    g = torch.Generator().manual_seed(seed)
    d = 1 << k
    shape = (d, d) if batch is None else (batch, d, d)
    a = torch.randn(*shape, generator=g, dtype=torch.float64) + 1j * torch.randn(*shape, generator=g, dtype=torch.float64)
    q, _ = torch.linalg.qr(a)
    return q.to(dtype)


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
@pytest.mark.parametrize('k,nc', [(1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (3, 0), (3, 2), (4, 0), (5, 0), (6, 1)])
def test_apply_gate_all_positions(dtype, k, nc):
    n, b = 9, 3
    rng = random.Random(1000 * k + nc)
    x = rand_state(b, n, dtype, 7)
    for trial in range(6):
        bits = rng.sample(range(n), k + nc)
        batched = trial % 2 == 1
        m = rand_unitary(k, dtype, trial, batch=b if batched else None)
        ref = oracle.apply_gate_bits(x, m, bits[:k], bits[k:])
        got = backend.apply_gate(x.to(dev()), m.to(dev()), bits[:k], bits[k:]).cpu()
        assert (got - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
@pytest.mark.parametrize('k,nc,n', [(5, 0, 7), (5, 2, 11), (6, 0, 9), (6, 1, 12), (7, 0, 12), (8, 1, 12), (9, 0, 11), (10, 0, 12),
                                    (10, 2, 13), (7, 0, 7)])
def test_dense_gates_on_the_matrix_cores(dtype, k, nc, n):
    """Dense 2^k x 2^k blocks, k = 5..10 (csrc/dq_dense.hip: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64): random
    target / control positions (low bits and the top bit included, targets in any order), one matrix for the batch and
    one per sample, fewer columns than a workgroup tile, against the oracle's permute / reshape / matmul -- and the
    VALU kernel it replaces gives the same answer."""
    b = 3
    rng = random.Random(77 * k + nc + n)
    x = rand_state(b, n, dtype, 11)
    for trial in range(4):
        bits = rng.sample(range(n), k + nc)
        if trial == 0:
            bits = list(range(k)) + bits[k:] if nc == 0 else bits          # the k lowest bits as targets
        if trial == 1 and nc == 0:
            bits = list(range(n - 1, n - 1 - k, -1))                       # the k highest, descending
        m = rand_unitary(k, dtype, 100 + trial, batch=b if trial % 2 else None)
        ref = oracle.apply_gate_bits(x, m, bits[:k], bits[k:])
        got = backend.apply_gate(x.to(dev()), m.to(dev()), bits[:k], bits[k:]).cpu()
        assert (got - ref).abs().max().item() < TOL[dtype], (k, nc, n, trial)
    try:
        _lib.check(_lib.load().dq_set_dense_path(0), 'dq_set_dense_path')
        old = backend.apply_gate(x.to(dev()), m.to(dev()), bits[:k], bits[k:]).cpu()
    finally:
        _lib.check(_lib.load().dq_set_dense_path(1), 'dq_set_dense_path')
    assert (old - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_apply_gate_in_place_and_edges(dtype):
    n = 6
    x = rand_state(2, n, dtype, 3)
    m = rand_unitary(1, dtype, 5)
    for t in (0, n - 1):
        xd = x.to(dev())
        ref = oracle.apply_gate_bits(x, m, [t], [])
        backend.apply_gate(xd, m.to(dev()), [t], [], out=xd)
        assert (xd.cpu() - ref).abs().max().item() < TOL[dtype]
    # one-qubit state
    x1 = rand_state(1, 1, dtype, 9)
    got = backend.apply_gate(x1.to(dev()), m.to(dev()), [0], []).cpu()
    assert (got - oracle.apply_gate_bits(x1, m, [0], [])).abs().max().item() < TOL[dtype]


def test_apply_gate_rejects_bad_arguments():
    x = rand_state(1, 4, torch.complex64, 1).to(dev())
    m = rand_unitary(1, torch.complex64, 1).to(dev())
    with pytest.raises(RuntimeError):
        backend.apply_gate(x, m, [4], [])          # bit out of range
    with pytest.raises(RuntimeError):
        backend.apply_gate(x, m, [1], [1])         # target == control
    with pytest.raises(RuntimeError):
        backend.apply_gate(x.cpu(), m.cpu(), [1], [])  # CPU tensors: no fallback


@pytest.mark.parametrize('is128,n', [(False, 12), (False, 15), (False, 14), (True, 11), (True, 14), (True, 13)])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_fused_passes_match_oracle(is128, n, seed):
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, 80, seed, kinds=('gen', 'x', 'diag', 'gen2', 'diag2'))
    mats = mats.to(dtype)
    steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    x = rand_state(2, n, dtype, 50 + seed)
    ref = run_reference(x, ops, mats)
    xd, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    for st in steps:
        backend.apply_fused(xd, md, 0, st.desc, out=xd)
    err = (xd.cpu() - ref).abs().max().item()
    assert err < TOL[dtype], err


@pytest.mark.parametrize('is128', [False, True])
def test_fused_batched_matrices_and_out_of_place(is128):
    dtype = torch.complex128 if is128 else torch.complex64
    n, b = 13, 3
    ops, mats0 = random_ops(n, 40, 11, kinds=('gen', 'x', 'diag', 'gen2'))
    per_sample = []
    for i in range(b):
        g = torch.Generator().manual_seed(i)
        mi = mats0.clone()
        # same structure, different unitaries per sample: scale every non-permutation matrix by a phase
        for op in ops:
            if op.kind != 'x':
                d2 = (1 << op.k) ** 2
                mi[op.mat : op.mat + d2] *= torch.exp(1j * torch.rand(1, generator=g, dtype=torch.float64) * 3)
        per_sample.append(mi)
    mats = torch.stack(per_sample).to(dtype)
    steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
    x = rand_state(b, n, dtype, 5)
    ref = torch.cat([run_reference(x[i : i + 1], ops, mats[i]) for i in range(b)])
    xd, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev()).contiguous()
    cur = xd
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, md, md.shape[1], st.desc, out=nxt)
        cur = nxt
    assert (cur.cpu() - ref).abs().max().item() < TOL[dtype]
    assert torch.equal(xd.cpu(), x)  # input untouched


@pytest.mark.parametrize('is128,n,b', [(False, 15, 3), (False, 16, 16), (True, 14, 5), (False, 13, 4), (True, 11, 2)])
def test_fused_pass_with_one_shared_input_state(is128, n, b):
    """dq_apply_fused_bcast_*: every sample reads the same input state (the first pass of a batched circuit),
    with its own matrices; workgroups of one tile are remapped to neighbours on one XCD."""
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats0 = random_ops(n, 30, 21, kinds=('gen', 'x', 'diag', 'gen2'))
    per_sample = []
    for i in range(b):
        g = torch.Generator().manual_seed(100 + i)
        mi = mats0.clone()
        for op in ops:
            if op.kind != 'x':
                d2 = (1 << op.k) ** 2
                mi[op.mat : op.mat + d2] *= torch.exp(1j * torch.rand(1, generator=g, dtype=torch.float64) * 3)
        per_sample.append(mi)
    mats = torch.stack(per_sample).to(dtype)
    steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
    assert all(isinstance(s_, fusion.FusedStep) for s_ in steps)
    x = rand_state(1, n, dtype, 77)
    ref = torch.cat([run_reference(x, ops, mats[i]) for i in range(b)])
    xd = x.to(dev())
    md = fusion.kernel_matrices(steps, ops, mats).to(dev()).contiguous()
    out = torch.empty(b, 1 << n, dtype=dtype, device=dev())
    backend.apply_fused(xd, md, md.shape[1], steps[0].desc, out=out)      # broadcast read
    for st in steps[1:]:
        backend.apply_fused(out, md, md.shape[1], st.desc, out=out)
    assert (out.cpu() - ref).abs().max().item() < TOL[dtype]
    assert torch.equal(xd.cpu(), x)


@pytest.mark.parametrize('n,seed,is128', [(18, 5, False), (17, 1, False), (16, 2, False), (20, 7, False), (15, 3, True), (16, 4, True)])
def test_stores_that_relabel_the_low_bits_on_gpu(n, seed, is128):
    """Passes that write other qubits to the contiguous low bits than they read there (store_low_pos, the explicit store
    layout store_rb / store_tb): schedules in which every pass picks all its tile qubits, against the oracle."""
    from test_fusion_cpu import free_low_schedule

    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, 300, seed, kinds=('gen', 'x', 'diag'))
    mats = mats.to(dtype)
    steps = free_low_schedule(ops, n, is128)
    assert steps is not None
    assert sum([s.desc.store_low_pos[i] for i in range(s.desc.L)] != list(range(s.desc.L)) for s in steps) >= 2
    x = rand_state(2, n, dtype, 60 + seed)
    ref = run_reference(x, ops, mats)
    cur, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    with pytest.raises(RuntimeError):
        backend.apply_fused(cur, md, 0, steps[0].desc, out=cur)        # a re-labelling pass cannot run in place
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, md, 0, st.desc, out=nxt)
        cur = nxt
    err = (cur.cpu() - ref).abs().max().item()
    assert err < TOL[dtype], err


@pytest.mark.parametrize('n', [15, 13, 12])
def test_reductions_inside_fused_passes_match_numpy(n):
    from _helpers import check_grad_records

    check_grad_records(n, dev())


@pytest.mark.parametrize('n', [15, 13, 12, 11])
def test_reductions_inside_fused_passes_match_numpy_c128(n):
    """dq_apply_fused_grad_c128 (wave-tile kernel, float64 sums all the way): 1e-12 relative."""
    from _helpers import check_grad_records

    check_grad_records(n, dev(), is128=True)


@pytest.mark.parametrize('dtype,n', [(torch.complex64, 15), (torch.complex64, 11), (torch.complex128, 13),
                                     (torch.complex128, 10), (torch.complex64, 9)])
def test_gate_grad_multi_equals_gate_by_gate(dtype, n):
    """Several single-target gate gradients from one read of both states (dq_gate_grad_multi_*) against one
    dq_gate_grad_* launch per gate: targets below / above the contiguous run, repeated targets, controls inside
    and outside the tile, more gates and more distinct high targets than one launch holds, a batch."""
    b = 3
    x, gy = rand_state(b, n, dtype, 31).to(dev()), rand_state(b, n, dtype, 32).to(dev())
    rng = random.Random(n)
    gates = []
    for t in list(range(n)) + [0, n - 1, n // 2]:
        others = [q for q in range(n) if q != t]
        nc = rng.choice([0, 0, 1, 2])
        gates.append((t, rng.sample(others, nc)))
    got = backend.gate_grad_multi(x, gy, gates)
    assert got.shape == (b, len(gates), 2, 2)
    tol = 1e-4 if dtype == torch.complex64 else 1e-12
    for k, (t, c) in enumerate(gates):
        ref = backend.gate_grad(x, gy, [t], c)
        assert (got[:, k] - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item()), (k, t, c)


def test_batches_wider_than_a_grid_dimension_go_in_slices(monkeypatch):
    """The batch is a grid dimension (<= 65535); get_unitary of a 16-qubit circuit has 65536 columns.  The slicing
    in backend.apply_gate / apply_fused is exercised here with a small limit."""
    monkeypatch.setattr(backend, 'MAX_BATCH', 3)
    dtype, n, b = torch.complex64, 13, 8
    ops, mats0 = random_ops(n, 25, 5, kinds=('gen', 'x', 'diag'))
    mats = torch.stack([mats0 * torch.exp(torch.tensor(0.1j * i)) for i in range(b)]).to(dtype)
    for op in ops:      # X-type matrices are not read by the kernels: keep the per-sample phase off them
        if op.kind == 'x':
            mats[:, op.mat:op.mat + 4] = mats0[op.mat:op.mat + 4].to(dtype)
    steps = fusion.schedule(ops, n, fusion.default_geometry(False))
    x = rand_state(b, n, dtype, 3)
    ref = torch.cat([run_reference(x[i:i + 1], ops, mats[i]) for i in range(b)])
    xd = x.to(dev())
    md = fusion.kernel_matrices(steps, ops, mats).to(dev()).contiguous()
    for st in steps:
        backend.apply_fused(xd, md, md.shape[1], st.desc, out=xd)
    assert (xd.cpu() - ref).abs().max().item() < TOL[dtype]
    # single-gate kernel with per-sample matrices, and the shared-input fused entry
    m = torch.stack([rand_unitary(1, dtype, 20 + i) for i in range(b)]).to(dev())
    got = backend.apply_gate(x.to(dev()), m, [5], [2]).cpu()
    want = torch.cat([oracle.apply_gate_bits(x[i:i + 1], m[i].cpu(), [5], [2]) for i in range(b)])
    assert (got - want).abs().max().item() < TOL[dtype]
    out = torch.empty(b, 1 << n, dtype=dtype, device=dev())
    backend.apply_fused(x[:1].to(dev()), md, md.shape[1], steps[0].desc, out=out)
    one = torch.cat([run_reference(x[:1], [ops[i] for i in steps[0].ops], mats[j]) for j in range(b)])
    assert (out.cpu() - one).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize('n', [13, 9, 16])
@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_many_z_strings_in_one_read(dtype, n):
    """dq_expect_zmulti_* / dq_scale_zsigns_*: 40 random Z-type strings (two launches of <= 32) against the
    single-string kernel and against the diagonal operator applied amplitude by amplitude (n = 9: the grid stride is
    more than a quarter of the state, so the kernels take their element-by-element sign path)."""
    b = 3
    x = rand_state(b, n, dtype, 41)
    xd = x.to(dev())
    rng = random.Random(9)
    masks = [rng.randrange(1, 1 << n) for _ in range(40)]
    got = backend.expect_z_multi(xd, masks)
    assert got.shape == (b, 40)
    for k, z in enumerate(masks):
        assert (got[:, k] - backend.expect_pauli(xd, 0, z)).abs().max().item() < 1e-12
    coef = torch.randn(b, 40, dtype=torch.float64)
    out = backend.scale_z_signs(xd, masks, coef.to(dev())).cpu()
    i = torch.arange(1 << n)
    w = torch.zeros(b, 1 << n, dtype=torch.float64)
    for k, z in enumerate(masks):
        par = torch.zeros_like(i)
        for p in range(n):
            if (z >> p) & 1:
                par ^= (i >> p) & 1
        w += coef[:, k:k + 1] * (1 - 2 * par)
    ref = x.to(torch.complex128) * w
    assert (out.to(torch.complex128) - ref).abs().max().item() < (1e-5 if dtype == torch.complex64 else 1e-12)


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_reductions(dtype):
    n, b = 10, 3
    x = rand_state(b, n, dtype, 21)
    xd = x.to(dev())
    rng = random.Random(5)
    for _ in range(8):
        k = rng.randint(1, 4)
        wires = rng.sample(range(n), k)
        basis = ''.join(rng.choice('xyz') for _ in range(k))
        xm = zm = 0
        for w, p in zip(wires, basis):
            bit = 1 << (n - 1 - w)
            xm |= bit if p in 'xy' else 0
            zm |= bit if p in 'zy' else 0
        ref = oracle.expectation_pauli(x, wires, basis).to(torch.float64)
        got = backend.expect_pauli(xd, xm, zm).cpu()
        assert (got - ref).abs().max().item() < TOL[dtype]
    y = rand_state(b, n, dtype, 22)
    ref = (x.conj() * y).sum(-1).to(torch.complex128)
    assert (backend.inner(xd, y.to(dev())).cpu() - ref).abs().max().item() < TOL[dtype]
    assert (backend.probs(xd).cpu() - torch.abs(x) ** 2).abs().max().item() < TOL[dtype]
    for wires in ([0], [3, 1], [9, 0, 4], list(range(n))[:7]):
        ref = oracle.probabilities(x, wires).to(torch.float64)
        got = backend.marginal(xd, [n - 1 - w for w in sorted(wires)]).cpu()
        assert (got - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_marginals_over_random_wire_sets(dtype):
    """dq_marginal_*: every size 1 .. n of measured set, in random outcome-bit order, for states smaller and larger than a
    workgroup's 4096-amplitude chunk, against permute / reshape / sum of |psi|^2 in float64 (reference qmath.py:624-626).
    The sets cover measured bits inside the chunk's contiguous part, among the bits a thread holds itself, and outside."""
    rng = random.Random(77)
    for n, b in ((1, 2), (2, 3), (5, 2), (9, 3), (12, 2), (13, 2), (17, 2)):
        x = rand_state(b, n, dtype, 60 + n)
        xd = x.to(dev())
        p = (x.to(torch.complex128).abs() ** 2).reshape([b] + [2] * n)
        sets = [rng.sample(range(n), rng.randint(1, n)) for _ in range(6)] + [list(range(min(n, 5))), [n - 1], [0]]
        if n >= 13:
            sets += [[0, 9, 10, 11], [8, 9, 10, 11], [12, 0], list(range(7, n)), list(range(12))]
        for bits in sets:
            got = backend.marginal(xd, bits).cpu()
            axes = [1 + (n - 1 - q) for q in bits]
            ref = p.permute([0] + axes + [a for a in range(1, n + 1) if a not in axes]).reshape(b, 1 << len(bits), -1).sum(-1)
            assert got.shape == ref.shape
            assert (got - ref).abs().max().item() < (1e-9 if dtype == torch.complex64 else 1e-13), (n, bits)


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
@pytest.mark.parametrize('k,nc', [(1, 0), (1, 2), (2, 0), (2, 1)])
def test_gate_grad(dtype, k, nc):
    from _cpu_backend import CpuTestBackend

    n, b = 8, 2
    x, gy = rand_state(b, n, dtype, 1), rand_state(b, n, dtype, 2)
    bits = random.Random(k * 10 + nc).sample(range(n), k + nc)
    ref = CpuTestBackend().gate_grad(x, gy, bits[:k], bits[k:])
    got = backend.gate_grad(x.to(dev()), gy.to(dev()), bits[:k], bits[k:]).cpu()
    assert (got - ref).abs().max().item() < TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
def test_pack_unpack(dtype):
    from _cpu_backend import CpuTestBackend

    cpu = CpuTestBackend()
    nl, b = 9, 2
    x = rand_state(b, nl, dtype, 4)
    for mask, value in ((0b100, 0b100), (0b100010, 0b000010), (0, 0), (0b1, 0)):
        ref = cpu.pack(x, mask, value)
        got = backend.pack(x.to(dev()), mask, value)
        assert torch.equal(got.cpu(), ref)
        y = rand_state(b, nl, dtype, 6)[:, : ref.shape[1]].contiguous()
        coef = torch.tensor([[0.3 + 0.1j, -0.2 + 0.5j]], dtype=dtype)
        a1, a2 = x.clone(), x.clone().to(dev())
        cpu.unpack_axpby(a1, ref, y, coef, mask, value)
        backend.unpack_axpby(a2, got, y.to(dev()), coef.to(dev()), mask, value)
        assert (a2.cpu() - a1).abs().max().item() < TOL[dtype]
        a1, a2 = x.clone(), x.clone().to(dev())
        cpu.unpack_axpby(a1, y, None, None, mask, value)
        backend.unpack_axpby(a2, y.to(dev()), None, None, mask, value)
        assert torch.equal(a2.cpu(), a1)


def handler_ops(n, ngates, seed):
    """2x2 gates of every promised structure (general, real, Rx-like, Hadamard) mostly uncontrolled, X with up to one
    control: the bodies specialised by matrix structure and the deferred factors."""
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    ops, mats, off = [], [], 0
    for _ in range(ngates):
        mode = rng.choice([0, 1, 2, 3, 'x', 'x'])
        nc = rng.choice([0, 1]) if mode == 'x' else int(rng.random() < 0.1)   # (a slot-controlled 2x2 has no handler)
        bits = rng.sample(range(n), 1 + nc)
        th = float(torch.rand(1, generator=g, dtype=torch.float64)) * 6.28
        c, s_ = torch.cos(torch.tensor(th / 2, dtype=torch.float64)), torch.sin(torch.tensor(th / 2, dtype=torch.float64))
        if mode == 'x':
            m = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
        elif mode == 0:
            a = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
            m, _ = torch.linalg.qr(a)
        elif mode == 1:
            m = torch.stack([torch.stack([c, -s_]), torch.stack([s_, c])]).to(torch.complex128)
        elif mode == 2:
            m = torch.stack([torch.stack([c + 0j, -1j * s_]), torch.stack([-1j * s_, c + 0j])])
        else:
            m = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) * float(torch.tensor(0.5, dtype=torch.float32).sqrt())
        ops.append(fusion.PrimOp('x' if mode == 'x' else 'gen', (bits[0],), tuple(bits[1:]), off,
                                 0 if mode == 'x' else (1 if (mode == 3 and nc) else mode)))
        mats.append(m.reshape(-1))
        off += 4
    return ops, torch.cat(mats)


@pytest.mark.parametrize('is128,n', [(False, 13), (False, 16), (True, 12), (True, 15)])
@pytest.mark.parametrize('seed', [0, 1])
def test_structured_gate_bodies_match_oracle(is128, n, seed):
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = handler_ops(n, 150, seed)
    mats = mats.to(dtype)
    steps = fusion.schedule(ops, n, fusion.default_geometry(is128))
    x = rand_state(2, n, dtype, 70 + seed)
    ref = run_reference(x, ops, mats)
    xd, md = x.to(dev()), fusion.kernel_matrices(steps, ops, mats).to(dev())
    for st in steps:
        backend.apply_fused(xd, md, 0, st.desc, out=xd)
    err = (xd.cpu() - ref).abs().max().item()
    assert err < TOL[dtype], err


@pytest.mark.parametrize('dtype', [torch.complex64, torch.complex128])
@pytest.mark.parametrize('n,batch', [(5, 2), (11, 3), (12, 1), (13, 2), (17, 3), (20, 1)])
def test_permute_bits_against_index_arithmetic(dtype, n, batch):
    """dq_permute_bits (the relayout before an all-to-all, the canonical order afterwards): out[i] = in[sigma(i)], every kernel
    variant -- per-element below 2^12, tiled with 16-byte pairs when bit 0 stays (complex64) and without, through LDS tiles
    when the low destination bits come from high source bits."""
    import random

    rng = random.Random(n)
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(batch, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(batch, 1 << n, generator=g, dtype=torch.float64)).to(dtype)
    idx = torch.arange(1 << n)
    perms = [list(range(n)), rng.sample(range(n), n), [0] + [1 + q for q in rng.sample(range(n - 1), n - 1)],
             list(range(1, n)) + [0], [q for q in range(n) if q not in (n - 3, n - 2)] + [n - 3, n - 2],
             list(range(n))[::-1], [n - 1] + list(range(n - 1)), [1, 0] + list(range(2, n))]
    perms += [rng.sample(range(n), n) for _ in range(5 if n <= 17 else 1)]
    for src_of_dst in perms:
        sidx = torch.zeros(1 << n, dtype=torch.long)
        for p, sp in enumerate(src_of_dst):
            sidx |= ((idx >> p) & 1) << sp
        out = torch.empty_like(x, device=dev())
        backend.permute_bits(x.to(dev()), src_of_dst, out=out)
        assert torch.equal(out.cpu(), x[:, sidx]), src_of_dst


def test_reductions_and_relayout_at_full_size():
    """Size-independent properties at the headline's size (n = 28, complex64, one sample = 2 GiB), where no CPU reference
    finishes in seconds: marginals are consistent with each other and with the norm whatever bits are measured, the
    matrix-core sums over Z strings equal the one-string kernel, scale_z_signs is its own inverse for a single string,
    and permute_bits followed by the inverse permutation is the identity bit for bit (both kernel variants)."""
    import random

    n = 28
    g = torch.Generator(device=dev()).manual_seed(28)
    x = torch.randn(1, 1 << n, 2, generator=g, device=dev(), dtype=torch.float32)
    x = torch.view_as_complex(x / x.norm()).contiguous()
    rng = random.Random(28)
    norm2 = backend.probs(x).double().sum().item()
    assert abs(norm2 - 1.0) < 1e-5
    for _ in range(4):
        big = rng.sample(range(n), rng.randint(6, 14))
        small = big[: rng.randint(1, 5)]
        pb = backend.marginal(x, big)                                   # (1, 2^len(big)), bits[0] = MSB
        ps = backend.marginal(x, small)
        assert abs(pb.sum().item() - norm2) < 1e-9 and abs(ps.sum().item() - norm2) < 1e-9
        folded = pb.reshape(1, 1 << len(small), -1).sum(-1)             # the leading bits of `big` are `small`
        assert (folded - ps).abs().max().item() < 1e-10, (big, small)
    masks = [rng.randrange(1, 1 << n) for _ in range(20)] + [1, 1 << (n - 1), (1 << n) - 1]
    many = backend.expect_z_multi(x, masks)
    for k, z in enumerate(masks):
        assert abs(many[0, k].item() - backend.expect_pauli(x, 0, z)[0].item()) < 1e-10, hex(z)
    one = torch.ones(1, 1, dtype=torch.float64, device=dev())
    y = backend.scale_z_signs(backend.scale_z_signs(x, [masks[0]], one), [masks[0]], one)
    assert torch.equal(y, x)
    for perm in (rng.sample(range(n), n), [1, 0] + list(range(2, n)), list(range(1, n)) + [0]):
        inv = [0] * n
        for p, sp in enumerate(perm):
            inv[sp] = p
        a = torch.empty_like(x)
        b = torch.empty_like(x)
        backend.permute_bits(x, perm, out=a)
        backend.permute_bits(a, inv, out=b)
        assert torch.equal(b, x), perm


@pytest.mark.parametrize('batch', [1, 5])
def test_deferred_rx_blocks_bit_for_bit(batch):
    """dq_defer_rx_c64 (one launch) against the same rewrite in tensor operations (fusion.defer_rx): every block
    { f, i t, -, flag } of include/dq_hip.h's DQ_MODE_RX, both forms (|a| >= |b| and |a| < |b|), the rest of the buffer
    untouched -- bit for bit."""
    g = torch.Generator().manual_seed(11)
    ngates, total = 300, 4 * 300 + 64
    theta = torch.rand(batch, ngates, generator=g) * 12.566
    theta[:, :4] = torch.tensor([0.0, torch.pi, torch.pi / 2, 3 * torch.pi / 2])       # a = +-1, b = 0 / a ~ 0 / |a| ~ |b|
    flat = torch.randn(batch, total, 2, generator=g).view(torch.float32)
    flat = torch.view_as_complex(flat.reshape(batch, total, 2).contiguous())
    perm = torch.randperm(ngates + 10, generator=g)[:ngates] * 4                       # blocks in any order, with gaps
    c, s = torch.cos(theta / 2), torch.sin(theta / 2)
    for k in range(ngates):
        o = int(perm[k])
        flat[:, o] = torch.complex(c[:, k], torch.zeros(batch))
        flat[:, o + 1] = torch.complex(torch.zeros(batch), -s[:, k])
        flat[:, o + 2] = flat[:, o + 1]
        flat[:, o + 3] = flat[:, o]
    flat = flat.to(dev())
    index = perm.to(torch.long).to(dev())
    want = fusion.defer_rx(flat.clone(), index)
    got = backend.defer_rx(flat.clone(), index)
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(want))
    assert not torch.equal(torch.view_as_real(got), torch.view_as_real(flat))
    flags = got[:, index + 3].real
    assert 0 < flags.sum().item() < flags.numel()                                      # both forms occurred
