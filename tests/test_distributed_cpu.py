"""Index-bit-sharded execution with real inter-process exchange (gloo, world_size 2 and 4, CPU): every
rank's shard must equal the corresponding slice of the dense result, expectation values and adjoint
gradients must match the dense autograd path.  Kernels are the CPU test double; the exchange logic,
rank predicates, pack/unpack plans and collective matching are the product code under test."""

import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, ret):
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.join(HERE, 'golden'))
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        torch.set_num_threads(1)
        import deepquantum_amd as dq
        from _cpu_backend import CpuTestBackend

        dq.backend.set_test_backend(CpuTestBackend())
        dq.DistributedQubitState.POISON_LAZY = True      # a lazy reset() leaves NaN where it does not clear (round 6)
        dq.setup_distributed('gloo')
        globals()['_case_' + case](dq, rank, world)
        dq.cleanup_distributed()
        ret[rank] = 'ok'
    except Exception:  # noqa: BLE001
        ret[rank] = traceback.format_exc()


def _run(case, world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == 'ok', f'rank {r}: {ret.get(r)}'


def _apply_spec(cir, spec):
    for method, args, kwargs in spec:
        getattr(cir, method)(*args, **kwargs)


def _shard_check(dq, rank, world, n, spec, double=True, tol=1e-10):
    dense = dq.QubitCircuit(n)
    _apply_spec(dense, spec)
    shard = dq.DistributedQubitCircuit(n)
    _apply_spec(shard, spec)
    if double:
        dense.to(torch.double)
        shard.to(torch.double)
    with torch.no_grad():
        ref = dense().reshape(-1)
        st = shard()
    per = 2**n // world
    err = (st.amps - ref[rank * per : (rank + 1) * per]).abs().max().item()
    assert err < tol, f'rank {rank}: shard error {err}'
    return dense, shard


GLOBAL_HEAVY = [
    ('hlayer', [], {}),
    ('rx', [0, 0.3], {}), ('ry', [1, 0.7], {}), ('rz', [0, 1.1], {}), ('p', [1, 0.4], {}),       # global targets
    ('cnot', [0, 4], {}), ('cnot', [4, 0], {}), ('cnot', [0, 1], {}), ('cx', [3, 1], {}),          # global ctrl / target
    ('toffoli', [0, 1, 4], {}), ('toffoli', [4, 3, 0], {}), ('ccx', [1, 4, 0], {}),
    ('swap', [[0, 4]], {}), ('swap', [[0, 1]], {}), ('swap', [[3, 4]], {}), ('fredkin', [2, 0, 3], {}),
    ('rxx', [[0, 3], 0.5], {}), ('ryy', [[1, 0], 0.8], {}), ('rzz', [[0, 1], 1.2], {}), ('rzz', [[4, 0], 0.6], {}),
    ('rxy', [[4, 1], 0.9], {}), ('cz', [0, 1], {}), ('cz', [0, 4], {}), ('cp', [4, 0, 0.3], {}),
    ('crx', [2, 0, 0.6], {}), ('crx', [0, 2, 0.9], {}), ('cry', [1, 0, 0.2], {}), ('crz', [3, 1, 1.4], {}),
    ('rx', [0, 0.8], {'controls': [2, 4]}), ('h', [1], {'controls': [0, 3]}), ('u3', [0, [0.3, 0.5, 0.7]], {}),
    ('crxx', [0, 1, 2, 0.3], {}), ('crzz', [4, 0, 1, 1.1], {}), ('iswap', [[1, 2]], {}),
    ('s', [0], {}), ('t', [1], {}), ('y', [0], {}), ('x', [1], {}), ('z', [0], {}),
]


def _both_modes(dq, fn):
    from deepquantum_amd import distributed as D

    out = {}
    for mode in ('pairwise', 'remap'):
        D.CONFIG['mode'] = mode
        fn()
        out[mode] = dict(D.LAST_RUN)
    D.CONFIG['mode'] = 'remap'
    return out


def _case_gates_w2(dq, rank, world):
    st = _both_modes(dq, lambda: _shard_check(dq, rank, world, 5, GLOBAL_HEAVY))
    assert st['pairwise']['remaps'] == 0 and st['pairwise']['pairwise_exchanges'] > 0
    assert st['remap']['remaps'] > 0 and st['remap']['pairwise_exchanges'] == 0


def _case_gates_w4(dq, rank, world):
    _both_modes(dq, lambda: _shard_check(dq, rank, world, 5, GLOBAL_HEAVY))
    _both_modes(dq, lambda: _shard_check(dq, rank, world, 6, [(m, a, k) for m, a, k in GLOBAL_HEAVY]
                                         + [('cnot_ring', [], {}), ('hlayer', [], {})]))


def _case_random_remap_w4(dq, rank, world):
    """The benchmark generator at n = 10 over 4 ranks: the all-to-all remap needs far fewer exchange steps
    than one pairwise exchange per global-qubit gate, and both give the dense result."""
    import specs

    spec = specs.random_spec(10, 12, 4242)
    st = _both_modes(dq, lambda: _shard_check(dq, rank, world, 10, spec, double=True, tol=1e-10))
    assert 0 < st['remap']['remaps'] < st['pairwise']['pairwise_exchanges'] / 2, st


def _case_remap_w8(dq, rank, world):
    import specs

    _shard_check(dq, rank, world, 9, specs.random_spec(9, 10, 77) + [('toffoli', [0, 1, 2], {}), ('rzz', [[0, 8], 0.3], {}),
                                                                    ('rxx', [[1, 2], 0.4], {}), ('swap', [[0, 2]], {})])


def _case_fused_local_w2(dq, rank, world):
    import specs

    # 14 qubits over 2 ranks: 13 local qubits >= the c64 tile (12 bits) -> local stretches are fused passes
    dense, shard = _shard_check(dq, rank, world, 14, specs.random_spec(14, 12, 99), double=False, tol=1e-5)
    assert dq.executor.LAST_RUN['passes'] > 0


def _case_expectation_grad_w4(dq, rank, world):
    n = 5

    def make(cls):
        cir = cls(n)
        cir.hlayer()
        cir.rx(0, encode=True)      # global target
        cir.ry(3, encode=True)
        cir.rz(1, encode=True)      # diagonal on a global qubit
        cir.cnot(0, 2)
        cir.cnot(4, 1)
        cir.crx(1, 4, encode=True)  # global control
        cir.crx(3, 0, encode=True)  # global target, local control
        cir.rzz([0, 3], encode=True)
        cir.ryy([2, 4], encode=True)
        cir.toffoli(0, 1, 3)
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        cir.observable([3, 4], 'zz')
        return cir

    data = torch.tensor([0.3, 1.1, -0.4, 0.8, 0.5, 1.7, 0.9])
    d1 = data.clone().requires_grad_(True)
    dense = make(dq.QubitCircuit)
    dense(d1)
    ev1 = dense.expectation()
    ev1.sum().backward()
    d2 = data.clone().requires_grad_(True)
    shard = make(dq.DistributedQubitCircuit)
    shard(d2)
    ev2 = shard.expectation()
    assert (ev1.detach() - ev2.detach()).abs().max().item() < 1e-5
    ev2.sum().backward()
    assert (d1.grad - d2.grad).abs().max().item() < 1e-4, (d1.grad, d2.grad)


def _fused_sweep_case(dq, rank, world, n, double, device=None):
    """The reverse sweep of the sharded adjoint as fused passes on the (psi, lambda) pair (adjoint._sweep_fused_sharded):
    trainable / encoded one-target gates on local and on global qubits, with local and global controls, fixed gates of
    every kind in between, trainable gates on two targets; against the dense circuit's autograd and against the
    gate-by-gate sweep of the reference (adjoint.py:42-83)."""
    from deepquantum_amd import adjoint, executor

    def make(cls):
        torch.manual_seed(3)
        cir = cls(n)
        cir.hlayer()
        cir.rxlayer(encode=True)
        cir.cnot_ring()
        cir.rylayer()                              # trainable
        cir.rz(0, encode=True)                     # diagonal, global target
        cir.crx(1, n - 1, encode=True)             # global control
        cir.crx(n - 2, 0, encode=True)             # global target, local control
        cir.u3(1, controls=[0, n - 1])             # trainable, general, a global and a local control
        cir.toffoli(0, 1, n - 3)
        cir.swap([0, n - 2])                       # fixed two-target gate
        # trainable / encoded gates on TWO targets, as in the reference's own distributed test circuit
        # (tests/test_circuit.py:87-139: rxx / ryy / rzz / rxy with controls): four one-target reduction records each
        cir.rxx([0, 1], controls=[2, n - 1], encode=True)      # global targets, a global and a local control
        cir.ryy([1, n - 2])                                     # trainable; a global and a local target
        cir.rzz([n - 3, n - 1])                                 # trainable, diagonal: two records
        cir.rxy([n - 1, 0], controls=[1], encode=True)
        cir.p(2)
        cir.cnot_ring(reverse=True)
        cir.rxlayer()
        cir.hlayer()
        cir.observable(0)
        cir.observable([1, n - 1], 'xy')
        if double:
            cir.to(torch.double)
        if device is not None:
            cir.to(device)
        return cir

    dt = torch.double if double else torch.float
    # complex128: the fixed gates are unitary to float32 rounding only (the reference's matrices after .to(double)) and
    # the sweep undoes with gate.inverse() like the reference's (adjoint.py:56-61): every Hadamard leaves psi a factor
    # 1 + 6e-8 off, in the reference's sweep as in this one
    tol = 2e-6 if double else 2e-4
    res = {}
    for which in ('dense', 'fused', 'gate_by_gate', 'per_observable'):
        executor.CONFIG['fused_sweep'] = which != 'gate_by_gate'
        executor.CONFIG['joint_adjoint'] = which != 'per_observable'     # (one sweep per observable: the reference's way)
        try:
            cir = make(dq.QubitCircuit if which == 'dense' else dq.DistributedQubitCircuit)
            data = torch.rand(cir.ndata, generator=torch.Generator().manual_seed(4), dtype=dt)
            if device is not None:
                data = data.to(device)
            data.requires_grad_(True)
            cir(data)
            ev = cir.expectation()
            (ev * torch.tensor([1.0, -0.5], dtype=dt, device=ev.device)).sum().backward()
            if which != 'dense':
                assert adjoint.LAST_SWEEP['fused'] == (which != 'gate_by_gate'), adjoint.LAST_SWEEP
                if which == 'fused':
                    assert adjoint.LAST_SWEEP['rows'] >= 2 * n and (world == 1 or adjoint.LAST_SWEEP['remaps'] >= 1)
            res[which] = (ev.detach().cpu(), data.grad.cpu(), [p.grad.cpu() for p in cir.parameters()])
        finally:
            executor.CONFIG['fused_sweep'] = True
            executor.CONFIG['joint_adjoint'] = True
    a = res['dense']
    for which in ('fused', 'gate_by_gate', 'per_observable'):
        b = res[which]
        assert (a[0] - b[0]).abs().max().item() < tol, (which, a[0], b[0])
        assert (a[1] - b[1]).abs().max().item() < tol, (which, (a[1] - b[1]).abs().max())
        assert len(a[2]) == len(b[2]) > n
        for x, y in zip(a[2], b[2], strict=True):
            assert (x - y).abs().max().item() < tol, (which, (x - y).abs().max())


def _case_fused_sweep_w2(dq, rank, world):
    _fused_sweep_case(dq, rank, world, 12, False)          # 11 local qubits: the pair is one 12-bit tile per rank
    _fused_sweep_case(dq, rank, world, 12, True)           # complex128: two 11-bit tiles


def _case_fused_sweep_w4(dq, rank, world):
    _fused_sweep_case(dq, rank, world, 13, False)


def _case_batched_w4(dq, rank, world):
    """Extension over the reference: a batch of encoded samples on the sharded state -- (B, 2^L) shards,
    per-sample matrices through global targets / controls, both exchange modes."""
    import specs
    from deepquantum_amd.distributed import inner_product_dist

    n, B = 10, 3
    def add_obs(c):
        c.observable(0)                     # Z on a global qubit
        c.observable([1, 5], 'zx')          # global Z, local X
        c.observable([0, 9], 'yz')          # Y on a global qubit: exchange path
        c.observable([4, 7], 'xy')          # local only

    dense = specs.build(dq, n, specs.BATCHED10)
    add_obs(dense)
    data = torch.rand(B, dense.ndata, generator=torch.Generator().manual_seed(5)) * 6.28
    with torch.no_grad():
        ref = dense(data).reshape(B, -1)
        ref_ev = dense.expectation()
    per = 2**n // world

    def run():
        shard = dq.DistributedQubitCircuit(n)
        _apply_spec(shard, specs.BATCHED10)
        add_obs(shard)
        st = shard(data)
        assert st.amps.shape == (B, per)
        ev = shard.expectation()
        assert ev.shape == ref_ev.shape and (ev - ref_ev).abs().max().item() < 1e-5, (ev, ref_ev)
        err = (st.amps - ref[:, rank * per : (rank + 1) * per]).abs().max().item()
        assert err < 1e-5, f'rank {rank}: batched shard error {err}'
        nrm = inner_product_dist(st, st)
        assert nrm.shape == (B,) and (nrm - 1).abs().max().item() < 1e-5
        # the same circuit object falls back to the reference's un-batched shard for 1-D data
        st1 = shard(data[1])
        assert st1.amps.shape == (per,)
        assert (st1.amps - ref[1, rank * per : (rank + 1) * per]).abs().max().item() < 1e-5
        with torch.no_grad():
            assert (shard.expectation() - ref_ev[1]).abs().max().item() < 1e-5

    _both_modes(dq, run)


def _virtual_bits_check(dq, rank, world, n, double):
    """CONFIG['virtual_bits']: an un-batched shard as 2^v rows that move through the remaps like the samples of a batch
    (rows = ranks of a virtual world; a remap that touches a virtual bit exchanges chunks between rows as well).  The
    same shards and expectation values as without them, for gates with targets, controls and diagonal factors on real
    rank bits, on virtual bits and on row-local bits."""
    import specs
    from deepquantum_amd import distributed as D

    g = world.bit_length() - 1
    spec = specs.random_spec(n, 6, 99) + [
        ('rz', [0, 0.37], {}), ('rz', [g, -0.9], {}), ('rz', [g + 1, 1.3], {}),              # diagonal on real / virtual bits
        ('cnot', [g, n - 1], {}), ('cnot', [g + 1, 0], {}), ('cnot', [n - 1, g], {}),         # controls / targets on virtual bits
        ('rzz', [[0, g], 0.6], {}), ('rzz', [[g, g + 1], -0.4], {}), ('crz', [g + 1, n - 4, 0.8], {}),
        ('toffoli', [0, g, n - 2], {}), ('rxx', [[g, n - 1], 0.5], {}), ('swap', [[g + 1, n - 3]], {}),
    ] + specs.random_spec(n, 3, 7)
    dense = dq.QubitCircuit(n)
    _apply_spec(dense, spec)
    dense.observable(0)
    dense.observable([g, n - 1], 'zz')
    dense.observable([g + 1, 2], 'xz')
    if double:
        dense.to(torch.double)
    with torch.no_grad():
        ref = dense().reshape(-1)
        ref_ev = dense.expectation()
    per = 2**n // world
    tol = 1e-10 if double else 2e-5
    stats = {}
    try:
        for vb in (0, 1, 2):
            D.CONFIG['virtual_bits'] = vb
            for lazy in (False, True):
                shard = dq.DistributedQubitCircuit(n)
                _apply_spec(shard, spec)
                shard.observable(0)
                shard.observable([g, n - 1], 'zz')
                shard.observable([g + 1, 2], 'xz')
                shard.lazy_layout = lazy
                if double:
                    shard.to(torch.double)
                with torch.no_grad():
                    st = shard()
                    stats[(vb, lazy)] = dict(D.LAST_RUN)
                    ev = shard.expectation()
                    amps = st.amps
                assert (ev - ref_ev).abs().max().item() < tol, (vb, lazy, ev, ref_ev)
                err = (amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
                assert err < tol, f'rank {rank}: virtual_bits {vb} lazy {lazy}: shard error {err}'
    finally:
        D.CONFIG['virtual_bits'] = 0
    for vb in (1, 2):
        st_ = stats[(vb, True)]
        assert st_['virtual_bits'] == vb and st_['virtual_remaps'] > 0, st_
        if st_['remaps'] > st_['virtual_remaps']:         # a remap of real rank bits only: the rows move in groups
            assert st_['groups'] == min(4, 1 << vb), st_
    assert stats[(0, True)]['virtual_bits'] == 0 and stats[(0, True)]['virtual_remaps'] == 0
    # behind reset() every rank but the first holds zeros until the first exchange of REAL rank bits: it runs none of the
    # stretches before it, however many re-labellings of virtual bits come first
    # (v = 0: the first exchange takes no wire since round 6 -- every rank computes rank 0's first stretch itself)
    for key, st_ in stats.items():
        if st_['remaps'] > st_['virtual_remaps'] and not st_['local_first_exchanges']:
            assert (st_['zero_shard_stretches'] >= 1) == (rank != 0), (key, rank, st_)
        if key[0] == 0 and st_['remaps'] > 0:
            assert st_['local_first_exchanges'] == 1, (key, rank, st_)


def _sliced_exchange_check(dq, rank, world, n, double, device='cpu'):
    """CONFIG['slice_exchange'] (round 6): the last pass in front of a remap and the first pass behind it in slices by two
    index bits below the chunk bits, the exchange slice by slice through a third buffer -- against the dense circuit and
    against the unsliced run, lazy and canonical layout; some remap must really have been cut (more launches than remaps)."""
    import bench
    from deepquantum_amd import distributed as D
    from deepquantum_amd import executor

    dtype = torch.complex128 if double else torch.complex64
    spec = bench.random_circuit_spec(n, 24 if device == 'cpu' else 40, seed=77)      # (several exchanges behind the first, local one)
    dense, data = bench.build_circuit(dq, n, spec, None, dtype, device)
    with torch.no_grad():
        ref = dense(data).reshape(-1)
        ref_ev = dense.expectation()
    per = (1 << n) // world
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 11
    tol = 1e-10 if double else 2e-5
    try:
        got = {}
        for nb in (2, 1, 0):
            D.CONFIG['slice_exchange'] = nb
            for lazy in (True, False):
                cir, _ = bench.build_circuit(dq, n, spec, None, dtype, device, distributed=True)
                cir.lazy_layout = lazy
                with torch.no_grad():
                    for _rep in range(2):           # (the second forward starts from whatever the first left in all three buffers)
                        st = cir(data)
                        stats = dict(D.LAST_RUN)
                        ev = cir.expectation()
                amps = st.amps.reshape(-1).clone()
                err = (amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
                assert err < tol, f'rank {rank}: slice_exchange {nb} lazy {lazy}: shard error {err}; {stats}'
                assert (ev.reshape(-1) - ref_ev.reshape(-1)).abs().max().item() < 10 * tol
                got[(nb, lazy)] = (amps, stats)
                if nb:
                    assert stats['sliced_remaps'] >= 1, stats
                else:
                    assert stats['sliced_remaps'] == 0
        for key, (amps, _) in got.items():
            assert (amps - got[(0, key[1])][0]).abs().max().item() < tol
        # (... and passes really were cut, on the send side or behind an exchange)
        assert any(max(st_['slice_launches_last'], st_['slice_launches_first']) >= 2 for (nb_, _), (_, st_) in got.items() if nb_), got
        if os.environ.get('DQ_TEST_VERBOSE') and rank == 0:
            print({k_: {s_: v_[1][s_] for s_ in ('remaps', 'sliced_remaps', 'slice_launches_last', 'slice_launches_first', 'wire_bytes')} for k_, v_ in got.items()}, flush=True)
    finally:
        executor.CONFIG.update(old)
        D.CONFIG['slice_exchange'] = None


def _deferred_tail_check(dq, rank, world, n, double):
    """CONFIG['defer_tail'] (round 6): an under-filled last pass of a stretch is not run, its gates move behind the exchange.
    Every rank must defer the SAME gates (decided on the stretch as the rank with all rank bits set sees it): the counts are
    compared across the ranks, the shards with the dense circuit and with the run that defers nothing -- with and without the
    exchanges in slices."""
    import bench
    import torch.distributed as dist
    from deepquantum_amd import distributed as D
    from deepquantum_amd import executor

    dtype = torch.complex128 if double else torch.complex64
    per = (1 << n) // world
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 11
    tol = 1e-10 if double else 3e-5
    deferred = 0
    try:
        for ci, (seed, depth) in enumerate(((1505, 22), (3, 30), (11, 30), (29, 40), (31, 22), (47, 40), (53, 30), (61, 40))):
            if ci >= 2 and deferred:        # (at least two circuits, and on until some stretch had its tail deferred)
                break
            spec = bench.random_circuit_spec(n, depth, seed=seed)
            dense, data = bench.build_circuit(dq, n, spec, None, dtype, 'cpu')
            with torch.no_grad():
                ref = dense(data).reshape(-1)
            got = {}
            for cap, nb in ((0, 0), (40, 0), (40, 2)):
                D.CONFIG['defer_tail'], D.CONFIG['slice_exchange'] = cap, nb
                cir, _ = bench.build_circuit(dq, n, spec, None, dtype, 'cpu', distributed=True)
                cir.lazy_layout = False
                with torch.no_grad():
                    st = cir(data)
                    stats = dict(D.LAST_RUN)
                amps = st.amps.reshape(-1).clone()
                err = (amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
                assert err < tol, f'rank {rank} seed {seed} defer_tail {cap} slices {nb}: shard error {err}; {stats}'
                got[(cap, nb)] = amps
                counts = torch.tensor([stats['deferred_tails'], stats['deferred_gates']], dtype=torch.int64)
                every = [torch.zeros_like(counts) for _ in range(world)]
                dist.all_gather(every, counts)
                assert all(torch.equal(c, counts) for c in every), f'the ranks deferred different gates: {every}'
                assert (cap > 0) or stats['deferred_tails'] == 0
                if cap:
                    deferred += stats['deferred_tails']
            for amps in got.values():
                assert (amps - got[(0, 0)]).abs().max().item() < tol
        assert deferred > 0, 'no stretch of any circuit had its tail deferred'
    finally:
        executor.CONFIG.update(old)
        D.CONFIG['defer_tail'], D.CONFIG['slice_exchange'] = 12, None


def _case_deferred_tail_w2(dq, rank, world):
    _deferred_tail_check(dq, rank, world, 17, double=True)


def _case_deferred_tail_w4(dq, rank, world):
    _deferred_tail_check(dq, rank, world, 18, double=False)


def _case_sliced_exchange_w2(dq, rank, world):
    _sliced_exchange_check(dq, rank, world, 16, double=True)
    _sliced_exchange_check(dq, rank, world, 17, double=False)


def _case_sliced_exchange_w4(dq, rank, world):
    _sliced_exchange_check(dq, rank, world, 18, double=False)


def _case_virtual_bits_w2(dq, rank, world):
    _virtual_bits_check(dq, rank, world, 14, double=True)       # complex128: rows of 2^11 amplitudes = one tile


def _case_virtual_bits_w4(dq, rank, world):
    _virtual_bits_check(dq, rank, world, 16, double=False)      # complex64: rows of 2^12


def _case_grouped_exchange_w4(dq, rank, world):
    """One coalesced exchange per group of samples (communication.exchange_chunks) against one collective per sample:
    the same shards, a fraction of the collectives; and the per-remap timing records stay consistent."""
    import specs
    from deepquantum_amd import communication as comm
    from deepquantum_amd import distributed as D

    n, B = 10, 6
    data = torch.rand(B, specs.build(dq, n, specs.BATCHED10).ndata, generator=torch.Generator().manual_seed(5)) * 6.28
    per = 2**n // world
    res, calls = {}, {}
    for grouped in (True, False):
        comm.COMM_CONFIG['grouped_exchange'] = grouped
        for k_ in comm.COMM_STATS:
            comm.COMM_STATS[k_] = 0
        shard = dq.DistributedQubitCircuit(n)
        _apply_spec(shard, specs.BATCHED10)
        shard.observable(0)
        with torch.no_grad():
            st = shard(data)
            res[grouped] = (st.amps.clone(), shard.expectation().clone())
        calls[grouped] = (comm.COMM_STATS['collectives'], D.LAST_RUN['remaps'], D.LAST_RUN['groups'])
    comm.COMM_CONFIG['grouped_exchange'] = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    dense = specs.build(dq, n, specs.BATCHED10)
    with torch.no_grad():
        ref = dense(data).reshape(B, -1)
    assert (res[True][0] - ref[:, rank * per:(rank + 1) * per]).abs().max().item() < 1e-5
    (g_calls, remaps, groups), (s_calls, _, _) = calls[True], calls[False]
    assert remaps > 0 and s_calls >= B * remaps - B, (calls,)     # (per sample: B collectives per remap, + the restore)
    assert g_calls <= groups * (remaps + 2), (calls,)              # (grouped: one per sample group and remap)
    assert g_calls * 2 <= s_calls, (calls,)


def _zero_state_check(dq, rank, world, n, batch, dtype=torch.complex64, device=None, depth=8):
    """The first local stretch behind ``reset()``: rank 0 holds |0..0> and its first passes skip what is still known to be
    zero (dq_apply_fused_zext_*), every other rank holds zeros and runs no pass at all.  On / off equality of the shards
    and of <Z0>, against the dense circuit too."""
    import bench
    from deepquantum_amd import distributed as D
    from deepquantum_amd import executor

    spec = bench.random_circuit_spec(n, depth, seed=31)
    dense, data = bench.build_circuit(dq, n, spec, batch, dtype, device or 'cpu')
    with torch.no_grad():
        ref = dense(data).reshape(-1, 1 << n)
        ref_ev = dense.expectation()
    per = (1 << n) // world
    calls = {'zext': 0}
    be = dq.backend.get_test_backend()
    if be is not None and device is None:
        inner = be.apply_fused

        def counting(*a, **kw):
            calls['zext'] += bool(kw.get('known_zero'))
            return inner(*a, **kw)

        be.apply_fused = counting
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        out = {}
        # (zero-state masks on / off; with them: the first exchange without the wire -- every rank computes rank 0's first
        # stretch and keeps its chunk, round 6 -- and over the wire as before)
        for on, local_first in ((True, True), (True, False), (False, True)):
            executor.CONFIG['zero_state'] = on
            D.CONFIG['first_exchange_local'] = local_first
            calls['zext'] = 0
            cir, _ = bench.build_circuit(dq, n, spec, batch, dtype, device or 'cpu', distributed=True)
            cir.lazy_layout = False
            with torch.no_grad():
                st = cir(data)
                ev = cir.expectation()
            stats = dict(D.LAST_RUN)
            amps = st.amps.reshape(-1, per).clone()
            out[(on, local_first)] = (amps, ev.clone())
            assert stats['remaps'] >= 1, stats
            if os.environ.get('DQ_TEST_VERBOSE'):
                print(f'rank {rank} n {n} zero_state {on} local_first {local_first}: {stats}', flush=True)
            if on and local_first:
                # no rank sat the first stretch out (unless its exchange group does not hold rank 0: k < log2 W), nothing of
                # the first exchange went over the wire, and EVERY rank ran masked passes
                assert stats['local_first_exchanges'] == 1, (rank, stats)
                assert stats['zero_shard_stretches'] in ((0,) if world == 2 else (0, 1)), (rank, stats)
                assert stats['known_zero_stretches'] >= 1, (rank, stats)
                if be is not None and device is None and stats['zero_shard_stretches'] == 0:
                    assert calls['zext'] >= 1, (rank, calls)
            elif on:
                assert stats['local_first_exchanges'] == 0
                assert stats['zero_shard_stretches'] == (1 if rank else 0), (rank, stats)
                # ... and behind the first exchange EVERY rank knows that the qubits that came from the rank bits are still
                # |0>: the stretch after it starts with their mask
                assert stats['known_zero_stretches'] == 1, (rank, stats)
                # (rank 0: its first stretch runs with the masks; behind the exchange the masks apply when the schedule of
                # that stretch can honour them -- with the free first placement the qubits on the rank bits are the ones the
                # circuit needs LAST, and a short circuit may leave one of them untouched to the end)
                if be is not None and device is None and rank == 0:
                    assert calls['zext'] >= 1, (rank, calls)
            else:
                assert stats['zero_shard_stretches'] == 0 and stats['known_zero_stretches'] == 0 and calls['zext'] == 0
            tol = 1e-10 if dtype == torch.complex128 else 2e-5
            err = (amps - ref[:, rank * per:(rank + 1) * per].to(amps.device)).abs().max().item()
            assert err < tol, f'rank {rank}, zero_state {on}: shard error {err}'
            assert (ev.reshape(-1) - ref_ev.reshape(-1).to(ev.device)).abs().max().item() < 10 * tol
        for key in ((True, True), (True, False)):
            assert torch.equal(out[key][0], out[(False, True)][0]) or (out[key][0] - out[(False, True)][0]).abs().max().item() < 1e-6
    finally:
        executor.CONFIG.update(old)
        D.CONFIG['first_exchange_local'] = True
        if be is not None and device is None:
            be.apply_fused = inner


def _case_initial_placement_w4(dq, rank, world):
    """CONFIG['initial_placement']: behind reset() the first qubit placement is free (|0..0> is the same vector under any
    permutation of the qubits) -- same shards, same <Z0> as with the reference's start, canonical and lazy layout, one
    exchange less on the benchmark generator's circuit."""
    import bench
    from deepquantum_amd import distributed as D
    from deepquantum_amd import executor

    n, depth = 16, 12
    spec = bench.random_circuit_spec(n, depth, seed=5)
    dense, data = bench.build_circuit(dq, n, spec, None, torch.complex64, 'cpu')
    with torch.no_grad():
        ref = dense(data).reshape(-1)
        ref_ev = dense.expectation()
    per = (1 << n) // world
    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        remaps = {}
        for on in (True, False):
            D.CONFIG['initial_placement'] = on
            for lazy in (False, True):
                cir, _ = bench.build_circuit(dq, n, spec, None, torch.complex64, 'cpu', distributed=True)
                cir.lazy_layout = lazy
                with torch.no_grad():
                    st = cir(data)
                    ev = cir.expectation()
                remaps[on] = D.LAST_RUN['remaps']
                err = (st.amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
                assert err < 2e-5, (rank, on, lazy, err)
                assert (ev.reshape(-1) - ref_ev.reshape(-1)).abs().max().item() < 1e-4
        assert remaps[True] <= remaps[False], remaps
    finally:
        D.CONFIG['initial_placement'] = True
        executor.CONFIG.update(old)


def _case_zero_state_w2(dq, rank, world):
    _zero_state_check(dq, rank, world, 16, 3)
    _zero_state_check(dq, rank, world, 15, None, dtype=torch.complex128)


def _case_zero_state_w4(dq, rank, world):
    _zero_state_check(dq, rank, world, 16, 2)


def _golden_dist_check(dq, rank, world, names, device=None, tol=2e-5):
    """Shards, expectation values and adjoint gradients against what the REAL reference produced under the
    same number of gloo ranks (tests/golden/golden_dist.npz, made by make_golden_dist.py)."""
    import numpy as np
    import specs

    gold = np.load(os.path.join(HERE, 'golden', 'golden_dist.npz'))
    for name in names:
        case = specs.DIST_CASES[name]
        ref_world = world if world in case['worlds'] else 1     # the reference could not shard this one
        shards = torch.from_numpy(gold[f'{name}/W{ref_world}/shards']).reshape(-1)
        per = shards.numel() // world
        cir = getattr(specs, case['builder'])(dq, dq.DistributedQubitCircuit, **case['kwargs'])
        data = None
        if case['data'] is not None:
            data = torch.tensor(case['data'], dtype=torch.float, requires_grad=True)
        if device is not None:
            cir.to(device)
            data = data.detach().to(device).requires_grad_(True) if data is not None else None
        st = cir(data=data)
        err = (st.amps.detach().cpu() - shards[rank * per:(rank + 1) * per]).abs().max().item()
        assert err < tol, f'{name} W={world} rank {rank}: shard error {err}'
        ev = cir.expectation()
        want = torch.from_numpy(gold[f'{name}/W{ref_world}/expectation'])
        assert (ev.detach().cpu() - want).abs().max().item() < tol, (name, ev, want)
        if data is not None:
            ev.sum().backward()
            gwant = torch.from_numpy(gold[f'{name}/W{ref_world}/grad'])
            assert (data.grad.cpu() - gwant).abs().max().item() < 10 * tol, (name, data.grad, gwant)


def _case_golden_w2(dq, rank, world):
    _both_modes(dq, lambda: _golden_dist_check(dq, rank, world, ['dist4', 'dist7']))


def _case_golden_w4(dq, rank, world):
    _both_modes(dq, lambda: _golden_dist_check(dq, rank, world, ['dist4', 'dist7', 'config4_n8']))


def _case_golden_w8(dq, rank, world):
    _both_modes(dq, lambda: _golden_dist_check(dq, rank, world, ['config5_n9']))


def _case_folded_permute_w2(dq, rank, world):
    """The remap's re-labelling of the local qubits rides on the last fused pass before the exchange (permuted store
    into the receive buffer), the batch moves in groups of samples: same amplitudes as the dense circuit, as a pass of
    its own (fold off), and with one group."""
    import specs
    from deepquantum_amd import distributed as D

    n, B = 15, 4                                    # 14 local qubits: the shards take the 13-bit tile
    spec = specs.random_spec(n, 6, 321)
    spec = [(m_, [a[0]], {'encode': True}) if m_ == 'rx' else (m_, a, k) for m_, a, k in spec]
    dense = specs.build(dq, n, spec)
    obs = [(0, 'z'), ([1, n - 1], 'zz'), ([2, 5, n - 2], 'zzz'), ([n - 1, 3], 'xz')]
    for w, basis in obs:
        dense.observable(w, basis)
    data = torch.rand(B, dense.ndata, generator=torch.Generator().manual_seed(8)) * 6.28
    with torch.no_grad():
        ref = dense(data).reshape(B, -1)
        ref_ev = dense.expectation()
    per = 2**n // world
    lazy = 0
    keep = dq.executor.CONFIG['permute_min_bits']
    dq.executor.CONFIG['permute_min_bits'] = 12
    try:
        remaps = {}
        for fold, groups, reorder in ((True, 4, False), (False, 4, False), (True, 1, False), (True, 2, False),
                                      (True, 4, True), (False, 2, True)):
            D.CONFIG['fold_permute'], D.CONFIG['overlap_groups'], D.CONFIG['reorder'] = fold, groups, reorder
            shard = dq.DistributedQubitCircuit(n)
            shard.lazy_layout = reorder          # (off: the reference's behaviour, canonical shards out of forward)
            _apply_spec(shard, spec)
            for w, basis in obs:
                shard.observable(w, basis)
            st = shard(data)
            stats = dict(D.LAST_RUN)
            assert shard.lazy_layout or D._is_canonical(st)
            # the circuit leaves the qubits where its last remap put them; Z-type expectation values are taken from
            # the shard as it lies (no exchange), an X factor on a rank bit or a look at the amplitudes restores the
            # reference's order first
            moved = not D._is_canonical(st)
            lazy += moved
            with torch.no_grad():
                ev3 = torch.stack([D.expect_pauli_dist(st, ob) for ob in list(shard.observables)[:3]], dim=-1)
            assert D._is_canonical(st) == (not moved)
            # (complex64: the dense circuit's Z values come out of its last pass in float64, the shards' from the float32
            # reductions of the test double -- up to 7e-6 apart on <Z0> = 0)
            assert (ev3 - ref_ev[:, :3]).abs().max().item() < 3e-5
            # (the three Z strings were reduced by the forward's last local pass, DQ_FG_EXPZ, signs of the rank bits and an
            # all-reduce included: `expectation` takes them from the cache, the 'xz' string from the shards)
            assert D.cached_expect_z(st) is not None and len(D.cached_expect_z(st)['masks']) == 3
            with torch.no_grad():
                ev = shard.expectation()
            assert (ev - ref_ev).abs().max().item() < 3e-5
            dq.executor.CONFIG['fused_expectation'] = False
            try:
                st2 = shard(data)
                assert D.cached_expect_z(st2) is None
                with torch.no_grad():
                    assert (shard.expectation() - ev).abs().max().item() < 3e-5
            finally:
                dq.executor.CONFIG['fused_expectation'] = True
            st = shard(data)
            err = (st.amps - ref[:, rank * per : (rank + 1) * per]).abs().max().item()
            assert D._is_canonical(st)
            assert err < 1e-5, f'rank {rank} fold={fold} groups={groups} reorder={reorder}: {err}'
            D.LAST_RUN.update(stats)
            assert D.LAST_RUN['remaps'] > 0 and D.LAST_RUN['groups'] in (groups, 1)
            remaps[reorder] = D.LAST_RUN['remaps']
            if not reorder:         # (in program order this circuit's remaps need a re-labelling of local qubits)
                if fold:
                    assert D.LAST_RUN['folded_permutes'] > 0, D.LAST_RUN
                else:
                    assert D.LAST_RUN['folded_permutes'] == 0 and D.LAST_RUN['permute_passes'] > 0, D.LAST_RUN
            # (round 6: the first exchange behind reset() takes no wire -- `first_exchange_local` -- and may be the only one)
            assert D.LAST_RUN['wire_bytes'] > 0 or D.LAST_RUN['local_first_exchanges'] > 0
        assert remaps[True] < remaps[False], remaps      # gates re-ordered along the commutation DAG: fewer exchanges
        assert lazy > 0, 'no run ended in a non-canonical qubit order: the lazy restore was not exercised'
    finally:
        dq.executor.CONFIG['permute_min_bits'] = keep
        D.CONFIG['fold_permute'], D.CONFIG['overlap_groups'], D.CONFIG['reorder'] = True, 4, True


def _case_measure_w2(dq, rank, world):
    cir = dq.DistributedQubitCircuit(4)
    cir.h(0)
    cir.cnot(0, 1)
    cir.cnot(1, 3)
    cir()
    res = cir.measure(shots=300, with_prob=True)
    if rank == 0:
        assert set(res) <= {'0000', '1101'} and sum(v[0] for v in res.values()) == 300
        assert all(abs(v[1] - 0.5) < 1e-6 for v in res.values())
    else:
        assert res == {}
    res = cir.measure(shots=100, wires=[0, 3])
    if rank == 0:
        assert set(res) <= {'00', '11'}


def _case_sampled_expectation_w4(dq, rank, world):
    """expectation(shots=...) on the sharded state (reference circuit.py:1739-1758): basis change on a copy of the shards,
    measure_dist on the observable's wires, parity average on rank 0 (empty tensors elsewhere) -- against the exact
    values, for Z, X, Y and a mixed two-wire string, on global and local wires."""
    torch.manual_seed(7 + rank * 0)
    n = 6
    cir = dq.DistributedQubitCircuit(n)
    cir.hlayer()
    cir.rx(0, 0.7)
    cir.ry(1, -0.4)
    cir.cnot(0, 5)
    cir.rz(5, 1.3)
    cir.cnot(5, 2)
    cir.observable(0)
    cir.observable(5, 'x')
    cir.observable(1, 'y')
    cir.observable([0, 4], 'zx')
    with torch.no_grad():
        cir()
        exact = cir.expectation()
        est = cir.expectation(shots=20000)
    if rank == 0:
        assert est.shape == exact.shape
        assert (est - exact).abs().max().item() < 0.03, (est, exact)
    else:
        assert est.numel() == 0
    assert cir.shots == 20000


@pytest.mark.parametrize('case,world', [('gates_w2', 2), ('gates_w4', 4), ('fused_local_w2', 2), ('sampled_expectation_w4', 4),
                                        ('random_remap_w4', 4), ('remap_w8', 8),
                                        ('expectation_grad_w4', 4), ('measure_w2', 2), ('batched_w4', 4), ('folded_permute_w2', 2),
                                        ('golden_w2', 2), ('golden_w4', 4), ('golden_w8', 8),
                                        ('fused_sweep_w2', 2), ('fused_sweep_w4', 4), ('grouped_exchange_w4', 4), ('virtual_bits_w2', 2), ('virtual_bits_w4', 4), ('initial_placement_w4', 4),
                                        ('zero_state_w2', 2), ('zero_state_w4', 4), ('sliced_exchange_w2', 2), ('sliced_exchange_w4', 4), ('deferred_tail_w2', 2), ('deferred_tail_w4', 4)])
def test_sharded_circuit(case, world):
    _run(case, world)


def test_reference_dist_tests_world_of_one(cpu_backend):
    """The reference's own tests/test_circuit.py:45-139 run un-sharded (no process group): same here,
    against the states / expectations / gradients the reference produced."""
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    _golden_dist_check(dq_mod(), 0, 1, ['dist4', 'dist7', 'config4_n8', 'config5_n9'])


def dq_mod():
    import deepquantum_amd as dq

    return dq


def test_single_process_world_of_one(cpu_backend):
    """Without a process group the sharded classes degrade to one shard (as in the reference's own
    tests, tests/test_circuit.py:45-139)."""
    import deepquantum_amd as dq

    dense = dq.QubitCircuit(4)
    shard = dq.DistributedQubitCircuit(4)
    for cir in (dense, shard):
        _apply_spec(cir, [('hlayer', [], {}), ('cnot_ring', [], {}), ('toffoli', [0, 1, 2], {}), ('rzz', [[0, 3], 0.4], {}),
                          ('swap', [[1, 2]], {}), ('rx', [2, 0.3], {'controls': [0]})])
    assert (shard().amps - dense().reshape(-1)).abs().max().item() < 1e-6


def test_lazily_built_shard_is_usable_by_everyone(cpu_backend, monkeypatch):
    """A shard bigger than LAZY_AMPS is not allocated by the constructor (2^31 amplitudes must not pass through host
    memory); whoever touches it first -- a gate routine, ``cir(state=s)``, ``state_dict`` -- finds |0...0> (the
    reference always constructs a usable shard, state.py:342-383)."""
    import deepquantum_amd as dq
    from deepquantum_amd.state import DistributedQubitState

    monkeypatch.setattr(DistributedQubitState, 'LAZY_AMPS', 4)
    s = DistributedQubitState(5)
    assert tuple(s._buffers['amps'].shape) == (0,)                 # not built yet
    assert tuple(s.amps.shape) == (32,) and s.amps[0] == 1 and s.amps.abs().sum() == 1
    assert tuple(s.buffer.shape) == (32,)
    s2 = DistributedQubitState(5)
    assert tuple(s2.state_dict()['amps'].shape) == (32,)
    cir = dq.DistributedQubitCircuit(5)
    cir.h(0)
    cir.cnot(0, 4)
    out = cir(state=DistributedQubitState(5))
    dense = dq.QubitCircuit(5)
    dense.h(0)
    dense.cnot(0, 4)
    assert (out.amps - dense().reshape(-1)).abs().max().item() < 1e-6
    s3 = DistributedQubitState(5, batch=3)
    assert tuple(s3.amps.shape) == (3, 32) and torch.all(s3.amps[:, 0] == 1)


def test_gates_reordered_along_the_commutation_dag_need_fewer_exchanges():
    """distributed._order_for_remaps: a permutation of the gate list that keeps every pair of non-commuting gates in
    order (so the circuit is the same operator: checked on a random state with the oracle), and that cuts the
    exchange steps and the bytes on the wire of the benchmark circuit several times over."""
    import bench
    from deepquantum_amd import distributed as D
    from deepquantum_amd.executor import Prim
    from oracle import statevec_oracle as oracle

    def prims_of(n, depth, seed):
        gen = torch.Generator().manual_seed(seed)
        h = torch.tensor([[1, 1], [1, -1]], dtype=torch.complex128) / 2**0.5
        x = torch.tensor([[0, 1], [1, 0]], dtype=torch.complex128)
        out = []
        for op in bench.random_circuit_spec(n, depth, seed):
            if op[0] == 'cnot':
                out.append(Prim('x', x, (n - 1 - op[2],), (n - 1 - op[1],), 0))
            elif op[0] == 'h':
                out.append(Prim('gen', h, (n - 1 - op[1],), (), 3))
            else:
                th = torch.rand((), generator=gen, dtype=torch.float64) * 6.0
                c, s_ = torch.cos(th / 2), torch.sin(th / 2)
                out.append(Prim('gen', torch.stack([c + 0j, -1j * s_, -1j * s_, c + 0j]).reshape(2, 2), (n - 1 - op[1],), (), 2))
        return out

    n, g = 9, 2
    prims = prims_of(n, 12, 5)
    for k in (7, 19, 33):           # a few diagonal gates (they run anywhere) and a two-target gate
        prims.insert(k, Prim('diag', torch.diag(torch.tensor([1, 1j], dtype=torch.complex128)), (k % n,), ((k + 3) % n,), 0))
    order = D._order_for_remaps(prims, list(range(n)), n, n - g)
    assert len(order) == len(prims) and {id(p) for p in order} == {id(p) for p in prims}
    assert [id(p) for p in order] != [id(p) for p in prims]
    x0 = torch.randn(1, 1 << n, dtype=torch.complex128, generator=torch.Generator().manual_seed(1))
    a, b = x0.clone(), x0.clone()
    for p in prims:
        a = oracle.apply_gate_bits(a, p.matrix, list(p.targets), list(p.controls))
    for p in order:
        b = oracle.apply_gate_bits(b, p.matrix, list(p.targets), list(p.controls))
    assert (a - b).abs().max().item() < 1e-12
    gained = False
    for world, base in ((2, 20), (8, 20), (2, 28)):      # (the last one: the weak series' n = 29 on two ranks, 4 -> 3 exchanges)
        gg = world.bit_length() - 1
        nn = base + gg
        big = prims_of(nn, 40, 1234)
        plain = D.count_exchange_steps(big, nn, gg)
        better = D.count_exchange_steps(D._order_for_remaps(big, list(range(nn)), nn, nn - gg), nn, gg)
        assert better['remap_steps'] * 2 <= plain['remap_steps'], (plain, better)
        assert better['remap_volume'] * 2 <= plain['remap_volume'], (plain, better)
        # ... and with the free first placement behind reset() (`initial_placement`) one exchange fewer still
        # (never worse than the reference's start: that start is one of the candidates)
        placed = D.count_exchange_steps(big, nn, gg, reorder=True, placement=True)
        assert (placed['remap_steps'], placed['remap_volume']) <= (better['remap_steps'], better['remap_volume']), (better, placed)
        gained = gained or placed['remap_steps'] < better['remap_steps']
    assert gained


def test_exchange_watchdog_names_a_stalled_exchange(capfd):
    """An exchange that does not complete is reported on stderr with what it is, without blocking anybody
    (communication._watch); one that completes is forgotten silently."""
    import time

    from deepquantum_amd import communication as comm

    class Work:
        def __init__(self, done):
            self.done = done

        def is_completed(self):
            return self.done

        def wait(self):
            pass

    old = comm.COMM_CONFIG['watchdog_seconds']
    comm.COMM_CONFIG['watchdog_seconds'] = 0.2
    try:
        stuck = comm.Exchange([Work(False)], 'shard exchange of remap 3 (peers [1, 2], 4096 bytes each way)')
        fine = comm.Exchange([Work(True)], 'shard exchange of remap 4 (peers [3], 64 bytes each way)')
        comm._watch(stuck)
        comm._watch(fine)
        time.sleep(1.0)
        err = capfd.readouterr().err
        assert 'watchdog' in err and 'remap 3' in err and 'peers [1, 2]' in err and 'remap 4' not in err
        stuck.works[0].done = True              # it completes after all: no further reports
        time.sleep(0.5)
        capfd.readouterr()
        time.sleep(0.6)
        assert 'remap 3' not in capfd.readouterr().err
    finally:
        comm.COMM_CONFIG['watchdog_seconds'] = old
