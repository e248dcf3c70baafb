"""Host-side scheduler: legality of the reordering and of the emitted descriptors (no GPU).

The descriptors are executed by the independent interpreter in tests/_cpu_backend.py and compared with
the oracle applying the same gates one by one in program order."""

import random

import pytest
import torch

import deepquantum_amd as dq
from deepquantum_amd import _lib, backend, fusion
from oracle import statevec_oracle as oracle


def random_ops(n, ngates, seed, kinds=('gen', 'x', 'diag', 'gen2', 'diag2', 'big')):
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    ops, mats, off = [], [], 0
    for _ in range(ngates):
        kind = rng.choice(kinds)
        k = {'gen': 1, 'x': 1, 'diag': 1, 'gen2': 2, 'gen2real': 2, 'gen2x': 2, 'gen2xc': 2, 'diag2': 2, 'big': 3}[kind]
        nc = rng.choice([0, 0, 0, 1, 1, 2])
        bits = rng.sample(range(n), k + nc)
        d = 1 << k
        if kind == 'x':
            m = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
        elif kind in ('diag', 'diag2'):
            m = torch.diag(torch.exp(1j * torch.rand(d, generator=g, dtype=torch.float64) * 6.28))
        elif kind == 'gen2real':      # a real, sparse, non-unitary 4x4 promised real (mode 1): a channel's superoperator
            m = torch.randn(d, d, generator=g, dtype=torch.float64) * (torch.rand(d, d, generator=g) < 0.5)
            m = (m / max(float(m.abs().sum(dim=1).max()), 1e-3)).to(torch.complex128)
        elif kind == 'gen2x':         # real and X-shaped (mode 4, DQ_MODE_XREAL): a 2x2 block on (00, 11), one on (01, 10)
            keep = torch.tensor([[(i ^ j) in (0, 3) for j in range(4)] for i in range(4)])
            m = torch.randn(d, d, generator=g, dtype=torch.float64) * keep
            m = (m / max(float(m.abs().sum(dim=1).max()), 1e-3)).to(torch.complex128)
        elif kind == 'gen2xc':        # X-shaped with complex entries (mode 5, DQ_MODE_XCPLX): a unitary 2x2 on each block
            m = torch.zeros(4, 4, dtype=torch.complex128)
            for blk in ((0, 3), (1, 2)):
                a = torch.randn(2, 2, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 2, generator=g, dtype=torch.float64)
                q2, _ = torch.linalg.qr(a)
                for i_ in range(2):
                    for j_ in range(2):
                        m[blk[i_], blk[j_]] = q2[i_, j_]
        else:
            a = torch.randn(d, d, generator=g, dtype=torch.float64) + 1j * torch.randn(d, d, generator=g, dtype=torch.float64)
            m, _ = torch.linalg.qr(a)
        pk = {'gen': 'gen', 'x': 'x', 'diag': 'diag', 'gen2': 'gen', 'gen2real': 'gen', 'gen2x': 'gen', 'gen2xc': 'gen', 'diag2': 'diag', 'big': 'gen'}[kind]
        ops.append(fusion.PrimOp(pk, tuple(bits[:k]), tuple(bits[k:]), off, {'gen2real': 1, 'gen2x': 4, 'gen2xc': 5}.get(kind, 0)))
        mats.append(m.reshape(-1))
        off += d * d
    return ops, torch.cat(mats)


def run_reference(state, ops, mats):
    x = state
    for op in ops:
        d = 1 << op.k
        m = mats[op.mat : op.mat + d * d].reshape(d, d)
        x = oracle.apply_gate_bits(x, m, list(op.targets), list(op.controls))
    return x


def run_steps(state, ops, mats, steps):
    x = state.clone()
    km = fusion.kernel_matrices(steps, ops, mats)      # matrices of a pass back to back, in gate order
    for st in steps:
        if isinstance(st, fusion.FusedStep):
            backend.apply_fused(x, km, 0, st.desc, out=x)
        else:
            op = ops[st.op]
            d = 1 << op.k
            x = backend.apply_gate(x, mats[op.mat : op.mat + d * d].reshape(d, d), op.targets, op.controls)
    return x


@pytest.mark.parametrize('is128,n', [(False, 12), (False, 14), (True, 13), (False, 15), (True, 11)])
@pytest.mark.parametrize('seed', [0, 1])
def test_schedule_matches_program_order(cpu_backend, is128, n, seed):
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, 60, seed)
    mats = mats.to(dtype)
    geom = fusion.default_geometry(is128)
    steps = fusion.schedule(ops, n, geom)
    executed = sorted(i for st in steps for i in (st.ops if isinstance(st, fusion.FusedStep) else [st.op]))
    assert executed == list(range(len(ops)))
    assert any(isinstance(st, fusion.FusedStep) for st in steps)
    g = torch.Generator().manual_seed(100 + seed)
    state = torch.randn(2, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 1 << n, generator=g, dtype=torch.float64)
    state = (state / state.norm(dim=-1, keepdim=True)).to(dtype)
    ref = run_reference(state, ops, mats)
    got = run_steps(state, ops, mats, steps)
    tol = 1e-12 if is128 else 2e-5
    assert torch.allclose(got, ref, atol=tol, rtol=0), (got - ref).abs().max()


def test_benchmark_circuit_fuses_deeply():
    n, depth = 28, 40
    spec = oracle.random_circuit_spec(n, depth)
    ops = []
    for i, op in enumerate(spec):
        if op[0] in ('h', 'rx'):
            ops.append(fusion.PrimOp('gen', (n - 1 - op[1],), (), 4 * i))
        else:
            ops.append(fusion.PrimOp('x', (n - 1 - op[2],), (n - 1 - op[1],), 4 * i))
    steps = fusion.schedule(ops, n, fusion.default_geometry(False))
    assert all(isinstance(s, fusion.FusedStep) for s in steps)
    assert sum(len(s.ops) for s in steps) == n * depth
    assert len(steps) < n * depth / 8  # >= 8 gates per HBM pass on the headline workload


def test_small_state_is_not_fused():
    ops = [fusion.PrimOp('gen', (0,), (), 0)]
    steps = fusion.schedule(ops, 5, fusion.default_geometry(False))
    assert len(steps) == 1 and isinstance(steps[0], fusion.SingleStep)


def test_descriptor_struct_sizes_match_header():
    """(The full comparison with the compiled library is tests/test_abi_cpu.py::test_struct_layout_matches_header.)"""
    import ctypes

    assert ctypes.sizeof(_lib.DqFusedGate) == 32
    assert ctypes.sizeof(_lib.DqFusedRound) == _lib.FUSED_MAX_SLOTS + _lib.FUSED_MAX_TBITS + 3
    # (a HOST struct, handed over by pointer; what travels in the 4 KiB kernel-argument segment is the kernel-side descriptor
    # of csrc/dq_wave.hip, and from ABI 24 on the records of a long pass lie in device memory)
    assert ctypes.sizeof(_lib.DqFusedPass) <= 8192 and _lib.FUSED_MAX_GATES == 128


def test_x_type_gates_commute_in_the_dag():
    """Gates that act on a shared qubit as functions of X (X, Rx-like matrices, CNOT targets) commute, like gates that
    act as functions of Z (controls, diagonal gates): the DAG must not order them, and must order everything else."""
    P = fusion.PrimOp
    rx = lambda q: P('gen', (q,), (), 0, 2)           # noqa: E731
    h = lambda q: P('gen', (q,), (), 0, 3)            # noqa: E731
    cx = lambda c, t: P('x', (t,), (c,), 0, 0)        # noqa: E731
    ops = [rx(0), cx(1, 0), rx(0), cx(2, 0), h(0), rx(0), cx(0, 1), P('diag', (0,), (), 0, 0), h(1)]
    dag = fusion._Dag(ops, 3)
    succ = {i: set(s) for i, s in enumerate(dag.succ)}
    assert dag.ready == [0, 1, 2, 3]                  # the four X-type gates on qubit 0 are mutually free
    assert all(4 in succ[i] for i in range(4))        # ... and all precede the Hadamard
    assert succ[4] == {5}                             # H -> Rx (a new X group of one)
    assert succ[5] == {6, 7}                          # Rx -> the two gates diagonal in qubit 0
    assert 8 in succ[6] and 8 not in succ[7]          # CNOT(0 -> 1) acts as X on qubit 1: before H(1)
    strict = fusion._Dag(ops, 3, x_commute=False)
    assert strict.ready == [0]


def test_planned_tiles_need_fewer_passes_and_keep_the_result(cpu_backend):
    """The dry-run planner (fusion._plan_tiles) against first-come tiles on a random H / Rx / CNOT circuit: fewer
    passes, same state."""
    import random
    n, rng = 20, random.Random(5)
    spec = []
    for _ in range(10):
        for q in range(n):
            r = rng.random()
            if r < 1 / 3:
                spec.append(('h', q))
            elif r < 2 / 3:
                spec.append(('rx', q, rng.uniform(0, 6.28)))
            else:
                t = rng.randrange(n - 1)
                spec.append(('cnot', q, t + (t >= q)))

    def run(width):
        # (a cap of 40 gates per pass: with the default 72 this small circuit is bounded by the cap, not by the tiles,
        # and every rule needs the same passes)
        dq.executor.CONFIG['plan_width'], dq.executor.CONFIG['max_gates'] = width, 40
        dq.executor._PLAN_CACHE.clear()
        try:
            cir = dq.QubitCircuit(n)
            for op in spec:
                getattr(cir, op[0])(*op[1:]) if op[0] != 'rx' else cir.rx(op[1], inputs=op[2])
            with torch.no_grad():
                out = cir().clone()
            return out, dq.executor.LAST_RUN['passes']
        finally:
            dq.executor.CONFIG['plan_width'], dq.executor.CONFIG['max_gates'] = None, None
            dq.executor._PLAN_CACHE.clear()

    s0, p0 = run(0)
    s1, p1 = run(1)
    s4, p4 = run(4)
    assert p4 <= p1 <= p0 and p4 < p0, (p0, p1, p4)
    assert torch.allclose(s0, s1, atol=1e-5) and torch.allclose(s0, s4, atol=1e-5)


def test_one_qubit_runs_are_merged_and_the_state_is_unchanged(cpu_backend):
    """executor.merge_one_qubit_runs: which gates merge (by structure and commutation), and that a circuit gives the
    same state with and without merging (per-sample angles included)."""
    from deepquantum_amd.executor import Prim, merge_one_qubit_runs
    h = torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2 ** 0.5
    x = torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat)

    def rx(t):
        t = torch.as_tensor(t, dtype=torch.float)
        c, s_ = torch.cos(t / 2) + 0j, -1j * torch.sin(t / 2)
        return torch.stack([torch.stack([c, s_], -1), torch.stack([s_, c], -1)], -2)

    P = lambda m, q, mode: Prim('gen', m, (q,), (), mode)           # noqa: E731
    cx = lambda c, t: Prim('x', x, (t,), (c,), 0)                   # noqa: E731
    prims = [P(h, 0, 3), P(h, 0, 3),                                 # H H = 2 s^2 I: a SCALAR, which rides on the next factor
             P(rx(0.3), 1, 2), cx(2, 1), P(rx([0.1, 0.2]), 1, 2),    # Rx . CX-target . Rx (batched) -> one Rx-like
             P(h, 2, 3), P(rx(0.5), 2, 2),                           # H Rx         -> stays two gates
             P(rx(0.4), 0, 2), cx(0, 1), P(rx(0.6), 0, 2)]           # (H H) Rx -> one Rx-like; control; Rx stays (Z-type in between)
    out = merge_one_qubit_runs(prims)
    kinds = [(p.kind, p.targets, p.mode, tuple(p.matrix.shape)) for p in out]
    assert kinds == [('gen', (0,), 2, (2, 2)),          # H H Rx(0.4): the scalar times an Rx-like matrix is Rx-like
                     ('gen', (1,), 2, (2, 2, 2)), ('x', (1,), 0, (2, 2)),
                     ('gen', (2,), 3, (2, 2)), ('gen', (2,), 2, (2, 2)),        # (a general matrix would cost the kernel more)
                     ('x', (1,), 0, (2, 2)), ('gen', (0,), 2, (2, 2))]
    assert torch.allclose(out[0].matrix, rx(0.4) @ h @ h, atol=1e-6)
    assert torch.allclose(out[1].matrix, rx([0.1, 0.2]) @ rx(0.3), atol=1e-6)
    assert bool((out[1].matrix[..., 0, 0].imag == 0).all()) and bool((out[1].matrix[..., 0, 1].real == 0).all())
    # an even number of Hadamard-like factors with nothing to merge into on ITS qubit: no gate at all -- the scalar is
    # multiplied into a carrier, the nearest following one-qubit gate (else the nearest one before); an odd number stays
    # Hadamard-like; a circuit that has no other one-qubit gate keeps the pair as a real matrix
    two = merge_one_qubit_runs([P(h, 0, 3), P(h, 0, 3), cx(0, 1), P(rx(0.3), 2, 2), P(h, 1, 3)])
    assert [(p.kind, p.targets, p.mode) for p in two] == [('x', (1,), 0), ('gen', (2,), 2), ('gen', (1,), 3)]
    assert torch.allclose(two[1].matrix, (h @ h)[0, 0] * rx(0.3), atol=1e-7) and torch.equal(two[2].matrix, h)
    last = merge_one_qubit_runs([P(rx(0.3), 2, 2), cx(2, 0), P(h, 0, 3), P(h, 0, 3), P(h, 0, 3), P(h, 0, 3)])
    assert [(p.kind, p.targets, p.mode) for p in last] == [('gen', (2,), 2), ('x', (0,), 0)]
    assert torch.allclose(last[0].matrix, (h @ h @ h @ h)[0, 0] * rx(0.3), atol=1e-7)
    three = merge_one_qubit_runs([P(h, 0, 3), P(h, 0, 3), P(h, 0, 3)])
    assert [(p.kind, p.targets, p.mode) for p in three] == [('gen', (0,), 3)] and torch.allclose(three[0].matrix, h @ h @ h, atol=1e-7)
    assert (three[0].matrix[0, 0] == three[0].matrix[0, 1]) and (three[0].matrix[1, 0] == -three[0].matrix[1, 1])    # exactly s [[1, 1], [1, -1]]
    alone = merge_one_qubit_runs([P(h, 0, 3), P(h, 0, 3), cx(0, 1)])
    assert [(p.kind, p.targets, p.mode) for p in alone] == [('gen', (0,), 1), ('x', (1,), 0)]
    hh = h @ h
    assert hh[0, 1] == 0 and hh[1, 0] == 0 and hh[0, 0] == hh[1, 1]       # the pair IS a scalar, to the last bit

    import random
    n, rng = 13, random.Random(11)
    cir = dq.QubitCircuit(n)
    for _ in range(8):
        for q in range(n):
            r = rng.random()
            if r < 0.4:
                cir.h(q)
            elif r < 0.75:
                cir.rx(q, encode=True)
            elif r < 0.85:
                cir.ry(q, inputs=rng.uniform(0, 6))
            else:
                t = rng.randrange(n - 1)
                cir.cnot(q, t + (t >= q))
    data = torch.rand(3, cir.ndata) * 6.28
    keep = dq.executor.CONFIG['merge_min_amps']
    try:
        with torch.no_grad():
            dq.executor.CONFIG['merge_min_amps'] = None
            ref = cir(data).clone()
            plain = dq.executor.LAST_RUN['gates']
            dq.executor.CONFIG['merge_min_amps'] = 0
            got = cir(data).clone()
            merged = dq.executor.LAST_RUN['gates']
    finally:
        dq.executor.CONFIG['merge_min_amps'] = keep
    assert merged < plain
    assert torch.allclose(got, ref, atol=2e-6)


def test_permuted_stores_make_every_later_tile_contiguous(cpu_backend):
    """fusion._place_writes: with permuted stores every pass after the first gathers exactly the bits right above
    the contiguous run, the write positions of every pass are a permutation of [L, n), the composition of all passes
    is the identity (the state comes back in canonical order) -- and the result equals the in-place schedule's."""
    n = 18
    ops, mats = random_ops(n, 200, 5, kinds=('gen', 'x', 'diag', 'gen2'))
    mats = mats.to(torch.complex64)
    geom = fusion.default_geometry(False)
    geom.permute_store = True
    geom.free_low = False      # (the low bits stay put here; the test below moves them too)
    steps = fusion.schedule(ops, n, geom)
    fused = [s for s in steps if isinstance(s, fusion.FusedStep)]
    assert len(fused) >= 3 and any(s.permutes for s in fused)
    where = list(range(n))                      # where[p] = which canonical bit currently lives at physical bit p
    for k, st in enumerate(fused):
        d, L, h = st.desc, st.desc.L, st.desc.h
        read = [d.high_sorted[i] for i in range(h)]
        if k > 0 and fused[k - 1].permutes:
            assert read == list(range(L, L + h)), 'tile not contiguous after a permuting pass'
        blk = [p for p in range(L, n) if p not in read]
        wr = {read[i]: d.store_high_pos[i] for i in range(h)}
        wr.update({p: d.store_blk_pos[j] for j, p in enumerate(blk)})
        assert sorted(wr.values()) == list(range(L, n))
        assert st.permutes == any(a != b for a, b in wr.items())
        new = list(where)
        for src, dst in wr.items():
            new[dst] = where[src]
        where = new
    assert where == list(range(n)), 'the passes do not compose to the canonical order'
    x = torch.randn(2, 1 << n, dtype=torch.complex64)
    km = fusion.kernel_matrices(steps, ops, mats)
    cur = x.clone()
    for st in steps:
        if isinstance(st, fusion.FusedStep):
            nxt = torch.empty_like(cur)
            backend.apply_fused(cur, km, 0, st.desc, out=nxt)
            cur = nxt
        else:
            op = ops[st.op]
            d2 = 1 << op.k
            cur = backend.apply_gate(cur, mats[op.mat : op.mat + d2 * d2].reshape(d2, d2), op.targets, op.controls)
    ref = run_reference(x, ops, mats)
    assert (cur - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('n', [13, 12])
def test_reduction_records_of_the_reverse_sweep(cpu_backend, n):
    """DQ_FG_GRAD records through the descriptor interpreter (the GPU suite runs the same check on the kernel)."""
    from _helpers import check_grad_records

    check_grad_records(n, torch.device('cpu'))


@pytest.mark.parametrize('n', [12, 11])
def test_reduction_records_of_the_reverse_sweep_c128(cpu_backend, n):
    from _helpers import check_grad_records

    check_grad_records(n, torch.device('cpu'), is128=True)


def free_low_schedule(ops, n, is128, width=4):
    geom = fusion.default_geometry(is128)
    geom.permute_store = geom.free_low = True
    return fusion._schedule(ops, n, geom, width, None, free_low=True)


@pytest.mark.parametrize('n,seed,is128', [(18, 5, False), (17, 1, False), (16, 2, False), (15, 3, True), (16, 4, True)])
def test_stores_that_relabel_the_low_bits(cpu_backend, n, seed, is128):
    """Schedules in which every pass picks ALL its tile qubits (fusion._schedule(free_low=True)): a pass writes the
    qubits its successor wants on the contiguous low bits there (store_low_pos), which needs them in its own tile;
    the write positions of a pass are a permutation of [0, n), complex64 passes write the tile bit of store slot 0 to
    index bit 0, the passes compose to the canonical order, and the state equals the reference's."""
    dtype = torch.complex128 if is128 else torch.complex64
    ops, mats = random_ops(n, 300, seed, kinds=('gen', 'x', 'diag'))
    mats = mats.to(dtype)
    steps = free_low_schedule(ops, n, is128)
    assert steps is not None and all(isinstance(s, fusion.FusedStep) for s in steps)
    moved = 0
    where = list(range(n))                      # where[p] = which canonical bit currently lives at physical bit p
    for k, st in enumerate(steps):
        d, L, h = st.desc, st.desc.L, st.desc.h
        read = list(range(L)) + [d.high_sorted[i] for i in range(h)]
        if k > 0:
            assert read == list(range(L + h)), 'tile not contiguous after a permuting pass'
        blk = [p for p in range(L, n) if p not in read]
        wr = {p: d.store_low_pos[p] for p in range(L)}
        wr.update({read[L + i]: d.store_high_pos[i] for i in range(h)})
        wr.update({p: d.store_blk_pos[j] for j, p in enumerate(blk)})
        assert sorted(wr.values()) == list(range(n))
        low_src = sorted(p for p, w in wr.items() if w < L)
        assert all(p in read for p in low_src), 'a low write position is fed from outside the tile'
        moved += [wr[p] for p in range(L)] != list(range(L))
        if not is128:
            tl = d.store_rb[0]
            assert wr[read[tl]] == 0
        new = list(where)
        for src, dst in wr.items():
            new[dst] = where[src]
        where = new
    assert where == list(range(n)), 'the passes do not compose to the canonical order'
    assert moved >= 2
    x = torch.randn(2, 1 << n, dtype=dtype)
    km = fusion.kernel_matrices(steps, ops, mats)
    cur = x.clone()
    for st in steps:
        nxt = torch.empty_like(cur)
        backend.apply_fused(cur, km, 0, st.desc, out=nxt)
        cur = nxt
    ref = run_reference(x, ops, mats)
    assert (cur - ref).abs().max().item() < (1e-10 if is128 else 2e-5)


def test_free_low_schedules_win_on_deep_layered_circuits():
    """The benchmark generator at n = 26: picking all tile qubits per pass needs fewer passes than sharing four fixed low
    qubits, and fusion.schedule takes the better of the two; circuits with a gate that must run on its own, or whose
    last tile has no room for qubits 0 .. L-1, keep the fixed low bits (None from _schedule)."""
    import bench

    n = 26
    ops, off = [], 0
    for op in bench.random_circuit_spec(n, 40, 1234):
        if op[0] == 'cnot':
            ops.append(fusion.PrimOp('x', (n - 1 - op[2],), (n - 1 - op[1],), off, 0))
        else:
            ops.append(fusion.PrimOp('gen', (n - 1 - op[1],), (), off, 3 if op[0] == 'h' else 2))
        off += 4
    counts = {}
    for fl in (False, True):
        geom = fusion.default_geometry(False)
        geom.permute_store = True
        geom.free_low = fl
        geom.plan_width, geom.plan_branch, geom.plan_restarts = 4, 3, 1
        steps = fusion.schedule(ops, n, geom)
        counts[fl] = len(steps)
        relabelled = any([s.desc.store_low_pos[i] for i in range(s.desc.L)] != list(range(s.desc.L)) for s in steps)
        assert relabelled == fl
    assert counts[True] < counts[False], counts
    big = ops + [fusion.PrimOp('gen', (5, 9, 13), (), off, 0)]          # a three-qubit gate runs on its own
    geom = fusion.default_geometry(False)
    geom.permute_store = True
    assert fusion._schedule(big, n, geom, 4, None, free_low=True) is None
    assert any(isinstance(s, fusion.SingleStep) for s in fusion.schedule(big, n, geom))
