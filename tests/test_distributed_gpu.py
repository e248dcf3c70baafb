"""The sharded path with the REAL kernels: two (and four) ranks share the one GPU of the test box and
exchange through gloo (host staged), so the pack / unpack / permute-bits kernels, the per-rank predicates
and the remap planner run end to end on HIP.  RCCL needs one GPU per rank: here it runs at world size 1
(``test_rccl_process_group_of_one_on_the_gpu``: every call, dtype view and stream hand-off of the N > 1 path), between
ranks by ``bench.py --gpus N`` only."""

import os
import socket
import sys
import time
import traceback

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, ret, backend='gloo', local=0):
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.join(HERE, 'golden'))
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY='0')
        import deepquantum_amd as dq

        torch.cuda.set_device(local)
        dq.setup_distributed(backend)
        globals()['_case_' + case](dq, rank, world)
        dq.cleanup_distributed()
        ret[rank] = 'ok'
    except Exception:  # noqa: BLE001
        ret[rank] = traceback.format_exc()


def _spawn_one(rank, world, port, case, ret, backend, devices):
    _worker(rank, world, port, case, ret, backend, devices[rank])


def _run(case, world, backend=None):
    """``backend`` None: RCCL with one device per rank when the box shows >= ``world`` devices, else gloo with every rank
    on device 0 (``_helpers.pick_transport``; the selector has a CPU test)."""
    sys.path.insert(0, HERE)
    from _helpers import pick_transport

    devices = [0] * world
    if backend is None:
        backend, devices = pick_transport(world)
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_spawn_one, args=(world, port, case, ret, backend, devices), nprocs=world, join=False)
    deadline = time.monotonic() + float(os.environ.get('DQ_TEST_RANKS_TIMEOUT', 600))
    while not ctx.join(timeout=5):
        if time.monotonic() > deadline:                 # a stalled exchange must not hang the suite (or the box)
            for pr in ctx.processes:
                pr.kill()
            pytest.fail(f'{case}: {world} ranks over {backend} did not finish in time; finished: {dict(ret)}')
    for r in range(world):
        assert ret.get(r) == 'ok', f'rank {r}: {ret.get(r)}'


def _build(dq, cls, n, spec, obs=True):
    cir = cls(n)
    for method, args, kwargs in spec:
        getattr(cir, method)(*args, **kwargs)
    if obs:
        cir.observable(0)
        cir.observable([1, n - 1], 'xz')
    return cir.to('cuda')


def _case_random_c64(dq, rank, world):
    import specs
    from deepquantum_amd import distributed as D

    n = 18
    spec = specs.random_spec(n, 12, 2024)
    dense = _build(dq, dq.QubitCircuit, n, spec)
    per = 2**n // world
    with torch.no_grad():
        ref = dense().reshape(-1)
        ref_ev = dense.expectation()
    for mode in ('pairwise', 'remap'):
        D.CONFIG['mode'] = mode
        shard = _build(dq, dq.DistributedQubitCircuit, n, spec)
        with torch.no_grad():
            st = shard()
            ev = shard.expectation()
        err = (st.amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
        assert err < 1e-5, f'{mode} rank {rank}: {err}'
        assert (ev - ref_ev).abs().max().item() < 1e-5
        assert dq.executor.LAST_RUN['passes'] > 0          # local stretches ran as fused passes
    D.CONFIG['mode'] = 'remap'


def _case_virtual_bits(dq, rank, world):
    """CONFIG['virtual_bits'] with the real kernels: an un-batched shard as 2^v rows (rows = ranks of a virtual world:
    real exchanges per row on the group streams, virtual trades as re-labellings folded into whole-shard passes) gives
    the shards and expectation values of v = 0, in both precisions, with and without the lazy layout."""
    import specs
    from deepquantum_amd import distributed as D

    g = world.bit_length() - 1
    for n, double in ((20, False), (19, True)):
        spec = specs.random_spec(n, 10, 31) + [('rz', [g, 0.4], {}), ('cnot', [g, n - 1], {}), ('cnot', [n - 2, g + 1], {}),
                                                ('rzz', [[0, g + 1], 0.7], {}), ('toffoli', [0, g, n - 3], {}),
                                                ('rxx', [[g, n - 1], 0.5], {})] + specs.random_spec(n, 4, 5)
        dense = _build(dq, dq.QubitCircuit, n, spec)
        if double:
            dense.to(torch.double)
        per = 2**n // world
        with torch.no_grad():
            ref = dense().reshape(-1)
            ref_ev = dense.expectation()
        tol = 1e-10 if double else 2e-5
        try:
            for vb in (1, 2, 0):
                D.CONFIG['virtual_bits'] = vb
                for lazy in (True, False):
                    shard = _build(dq, dq.DistributedQubitCircuit, n, spec)
                    shard.lazy_layout = lazy
                    if double:
                        shard.to(torch.double)
                    with torch.no_grad():
                        st = shard()
                        stats = dict(D.LAST_RUN)
                        ev = shard.expectation()
                        err = (st.amps - ref[rank * per:(rank + 1) * per]).abs().max().item()
                    assert (ev - ref_ev).abs().max().item() < tol, (vb, lazy, ev, ref_ev)
                    assert err < tol, f'rank {rank}: virtual_bits {vb} lazy {lazy} double {double}: {err}'
                    assert stats['virtual_bits'] == vb and (vb == 0 or stats['virtual_remaps'] > 0), stats
        finally:
            D.CONFIG['virtual_bits'] = 0


def _case_sliced_exchange(dq, rank, world):
    """CONFIG['slice_exchange'] with the real kernels (dq_apply_fused_slice_*): the passes around a remap in slices, the
    exchange slice by slice through the third buffer (tests/test_distributed_cpu.py, _sliced_exchange_check)."""
    from test_distributed_cpu import _sliced_exchange_check

    dev = torch.device('cuda', torch.cuda.current_device())
    _sliced_exchange_check(dq, rank, world, 19 + (world > 2), False, device=dev)
    _sliced_exchange_check(dq, rank, world, 18 + (world > 2), True, device=dev)


def _case_zero_state(dq, rank, world):
    """The first local stretch behind ``reset()`` with the real kernels: rank 0's first passes skip what is still known
    to be zero, the other ranks -- all zeros -- run no pass at all (tests/test_distributed_cpu.py, _zero_state_check)."""
    from test_distributed_cpu import _zero_state_check

    dev = torch.device('cuda', torch.cuda.current_device())
    _zero_state_check(dq, rank, world, 21, 3, device=dev, depth=10)
    _zero_state_check(dq, rank, world, 20, None, dtype=torch.complex128, device=dev, depth=10)


def _case_batched_c128(dq, rank, world):
    """Batched shards with per-sample matrices, double precision, golden-style tolerance 1e-10."""
    import specs

    n, B = 14, 3
    spec = specs.random_spec(n, 6, 31)
    spec = [(m, [a[0]], {'encode': True}) if m == 'rx' else (m, a, k) for m, a, k in spec]
    spec += [('rzz', [[0, n - 1]], {'encode': True}), ('crx', [n - 1, 0], {'encode': True}),
             ('toffoli', [0, 1, n - 2], {}), ('u3', [1], {'encode': True}), ('swap', [[0, 5]], {})]
    dense = _build(dq, dq.QubitCircuit, n, spec).to(torch.double)
    shard = _build(dq, dq.DistributedQubitCircuit, n, spec).to(torch.double)
    data = (torch.rand(B, dense.ndata, generator=torch.Generator().manual_seed(9), dtype=torch.double) * 6.28).cuda()
    per = 2**n // world
    with torch.no_grad():
        ref = dense(data).reshape(B, -1)
        st = shard(data)
        assert st.amps.shape == (B, per) and st.amps.dtype == torch.complex128
        assert (st.amps - ref[:, rank * per:(rank + 1) * per]).abs().max().item() < 1e-10
        assert (shard.expectation() - dense.expectation()).abs().max().item() < 1e-10


def _case_adjoint_grad(dq, rank, world):
    n = 12

    def make(cls):
        cir = cls(n)
        cir.hlayer()
        cir.rx(0, encode=True)
        cir.ry(5, encode=True)
        cir.rz(1, encode=True)
        cir.cnot(0, 7)
        cir.crx(1, 9, encode=True)
        cir.crx(6, 0, encode=True)
        cir.rzz([0, 8], encode=True)
        cir.ryy([1, 11], encode=True)
        cir.toffoli(0, 1, 3)
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        return cir.to('cuda')

    data = torch.tensor([0.3, 1.1, -0.4, 0.8, 0.5, 1.7, 0.9], device='cuda')
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


def _case_fused_sweep(dq, rank, world):
    """The sharded adjoint's reverse sweep as fused passes on the (psi, lambda) pair with the real kernels
    (dq_apply_fused_grad_c64 / _c128 between remaps): against the dense circuit and the gate-by-gate sweep."""
    from test_distributed_cpu import _fused_sweep_case

    _fused_sweep_case(dq, rank, world, 13 + (world > 2), False, device='cuda')
    _fused_sweep_case(dq, rank, world, 13 + (world > 2), True, device='cuda')
    _fused_sweep_case(dq, rank, world, 17, False, device='cuda')           # several tiles per shard, permuted stores


def _case_golden(dq, rank, world):
    """Reference-made fixtures (real DistributedQubitCircuit under gloo) vs. the HIP kernels."""
    from deepquantum_amd import distributed as D
    from test_distributed_cpu import _golden_dist_check

    names = {2: ['dist4', 'dist7'], 4: ['dist4', 'dist7', 'config4_n8'], 8: ['config5_n9']}[world]
    for mode in ('pairwise', 'remap'):
        D.CONFIG['mode'] = mode
        _golden_dist_check(dq, rank, world, names, device='cuda')
    D.CONFIG['mode'] = 'remap'


def _case_folded_permute(dq, rank, world):
    """Remap with the real kernels at a size where the shards run permuted stores (>= 2^20 amplitudes): the last fused
    pass before every exchange writes the re-labelled shard into the receive buffer; groups of samples on their own
    streams.  Against the dense circuit, fold on / off, 1 / 2 / 4 groups."""
    import specs
    from deepquantum_amd import distributed as D

    n, B = 21 + (world.bit_length() - 1), 4
    spec = specs.random_spec(n, 6, 99)
    spec = [(m_, [a[0]], {'encode': True}) if m_ == 'rx' else (m_, a, k) for m_, a, k in spec]
    dense = _build(dq, dq.QubitCircuit, n, spec)
    data = (torch.rand(B, dense.ndata, generator=torch.Generator().manual_seed(8)) * 6.28).cuda()
    per = 2**n // world
    with torch.no_grad():
        ref = dense(data).reshape(B, -1)[:, rank * per:(rank + 1) * per].clone()
        ref_ev = dense.expectation()
    del dense
    torch.cuda.empty_cache()
    try:
        remaps = {}
        for fold, groups, reorder in ((True, 4, False), (False, 2, False), (True, 1, False), (True, 4, True), (False, 2, True)):
            D.CONFIG['fold_permute'], D.CONFIG['overlap_groups'], D.CONFIG['reorder'] = fold, groups, reorder
            shard = _build(dq, dq.DistributedQubitCircuit, n, spec)
            shard.lazy_layout = reorder      # the canonical order comes back when somebody reads st.amps
            with torch.no_grad():
                st = shard(data)
                stats = dict(D.LAST_RUN)
                ev = shard.expectation()
            assert (st.amps - ref).abs().max().item() < 1e-5, (fold, groups, reorder)
            assert (ev - ref_ev).abs().max().item() < 1e-5
            assert stats['remaps'] > 0
            remaps[reorder] = stats['remaps']
            if not reorder:         # (in program order this circuit's remaps re-label local qubits)
                assert (stats['folded_permutes'] > 0) == fold, stats
            else:
                assert stats['folded_permutes'] == 0 or fold, stats
        assert remaps[True] <= remaps[False], remaps     # gates re-ordered along the commutation DAG: no more exchanges
    finally:
        D.CONFIG['fold_permute'], D.CONFIG['overlap_groups'], D.CONFIG['reorder'] = True, 4, True


def _case_measure(dq, rank, world):
    """measure_dist on the HIP kernels (reference: distributed.py:205-285): the probabilities it reports are the
    dense circuit's marginals, measured global wires select the rank's slot, the dict lives on rank 0 only."""
    import specs
    from deepquantum_amd import distributed as D

    n = 12
    spec = specs.random_spec(n, 5, 77)
    dense = _build(dq, dq.QubitCircuit, n, spec, obs=False)
    shard = _build(dq, dq.DistributedQubitCircuit, n, spec, obs=False)
    with torch.no_grad():
        psi = dense().reshape(-1)
        st = shard()
    p = (psi.abs() ** 2).double().reshape([2] * n).cpu()
    for wires in (None, [0, 3], [1], [2, 5, n - 1], list(range(2, n)), [0, 1]):
        res = D.measure_dist(st, shots=4000, with_prob=True, wires=wires)
        if rank != 0:
            assert res == {}
            continue
        w = sorted(wires) if wires is not None else list(range(n))
        marg = p.permute(w + [q for q in range(n) if q not in w]).reshape(2 ** len(w), -1).sum(-1)
        assert sum(v[0] for v in res.values()) == 4000
        for bits, (_count, prob) in res.items():
            assert len(bits) == len(w)
            assert abs(prob - marg[int(bits, 2)].item()) < 1e-6, (wires, bits, prob, marg[int(bits, 2)].item())
        if len(w) <= 2:   # every outcome with noticeable weight shows up, with about the right frequency
            for o in range(2 ** len(w)):
                if marg[o] > 0.05:
                    got = res.get(format(o, f'0{len(w)}b'), (0, 0))[0] / 4000
                    assert abs(got - marg[o].item()) < 0.05
    # the circuit-level entry point (DistributedQubitCircuit.measure) takes the same path
    res = shard.measure(shots=64, wires=[0, n - 1])
    assert (rank == 0) == bool(res)


def _case_rccl(dq, rank, world):
    """Every RCCL call, dtype view and stream hand-off that ``bench.py --gpus N`` hits first, on the one GPU of the test
    box (process group 'nccl' = RCCL, world size 1; reference: communication.py:9-35, 58-91): asynchronous
    all_to_all_single on a side stream with complex shards viewed as reals, work.wait(), joining the streams, the
    pairwise exchange helper, all_reduce of the small results, and a sharded circuit -- forward, expectation, adjoint
    backward -- in both exchange modes.  It cannot test a remap between ranks; it does test that nothing on that path
    raises, hangs or mis-orders streams."""
    import torch.distributed as dist

    import specs
    from deepquantum_amd import communication as C
    from deepquantum_amd import distributed as D

    assert dist.get_backend() == 'nccl' and world == 1
    dev = torch.device('cuda', torch.cuda.current_device())
    for dtype in (torch.complex64, torch.complex128):
        g = torch.Generator().manual_seed(3)
        send = torch.randn(1 << 16, generator=g, dtype=torch.float64).to(dtype).to(dev) * (1 + 2j)
        recv = torch.zeros_like(send)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            sr, rr = torch.view_as_real(send).reshape(-1), torch.view_as_real(recv).reshape(-1)
            w = C.all_to_all_flat(rr, sr, [sr.numel()], async_op=True)
            assert w is not None
            w.wait()                                    # orders the collective before the side stream's next kernel
            doubled = recv * 2
        torch.cuda.current_stream(dev).wait_stream(side)
        assert torch.equal(recv, send) and torch.equal(doubled, send * 2)
        recv.zero_()
        C.comm_exchange_arrays(send, recv, None)        # the pairwise helper: nobody to talk to at world size 1
        torch.cuda.synchronize()
        assert not recv.any()
        # the small collectives of the reductions (inner products, marginals): complex scalars travel as reals
        z = torch.tensor([1.5 - 2j], dtype=dtype, device=dev)
        zr = torch.view_as_real(z)
        dist.all_reduce(zr)
        assert torch.equal(z.cpu(), torch.tensor([1.5 - 2j], dtype=dtype))
    n = 12
    spec = specs.random_spec(n, 5, 31)
    dense = _build(dq, dq.QubitCircuit, n, spec)
    for mode in ('remap', 'pairwise'):
        D.CONFIG['mode'] = mode
        try:
            shard = _build(dq, dq.DistributedQubitCircuit, n, spec)
            with torch.no_grad():
                ref = dense().reshape(-1)
                ref_ev = dense.expectation()
                st = shard()
                ev = shard.expectation()
            assert (st.amps - ref).abs().max().item() < 1e-5, mode
            assert (ev.reshape(-1) - ref_ev.reshape(-1)).abs().max().item() < 1e-5, mode
            # batched shards: sample groups on their own streams, each with its own (self) exchange in flight
            data = torch.rand(4, max(1, shard.ndata), device=dev) if shard.ndata else None
            if data is not None:
                with torch.no_grad():
                    sb = shard(data)
                    rb = dense(data)
                assert (sb.amps - rb.reshape(4, -1)).abs().max().item() < 1e-5
        finally:
            D.CONFIG['mode'] = 'remap'
    # adjoint-mode gradient through the sharded state (reference: adjoint.py:19-83, circuit.py:1706-1738)
    qa = dq.DistributedQubitCircuit(8)
    qd = dq.QubitCircuit(8)
    th_a = torch.tensor([0.3, 1.1, -0.4], device=dev, requires_grad=True)
    th_d = th_a.detach().clone().requires_grad_(True)
    for cir, th in ((qa, th_a), (qd, th_d)):
        cir.hlayer()
        cir.rx(0, encode=True)
        cir.cnot(0, 7)
        cir.ry(7, encode=True)
        cir.cnot(7, 3)
        cir.rz(3, encode=True)
        cir.observable([0, 7])
        cir.observable(3, 'x')
        cir.to(dev)
        cir(th)
        cir.expectation().sum().backward()
    assert (th_a.grad - th_d.grad).abs().max().item() < 1e-5


def test_rccl_process_group_of_one_on_the_gpu():
    _run('rccl', 1, backend='nccl')


def test_reference_dist_tests_on_gpu_world_of_one():
    import deepquantum_amd as dq
    from test_distributed_cpu import _golden_dist_check

    sys.path.insert(0, os.path.join(HERE, 'golden'))
    _golden_dist_check(dq, 0, 1, ['dist4', 'dist7', 'config4_n8', 'config5_n9'], device='cuda')


@pytest.mark.parametrize('case,world', [('golden', 2), ('golden', 4), ('golden', 8), ('random_c64', 2), ('random_c64', 4), ('batched_c128', 4), ('adjoint_grad', 2),
                                        ('measure', 2), ('measure', 4), ('folded_permute', 2), ('folded_permute', 4),
                                        ('fused_sweep', 2), ('fused_sweep', 4), ('virtual_bits', 2), ('virtual_bits', 4),
                                        ('zero_state', 2), ('zero_state', 4), ('sliced_exchange', 2), ('sliced_exchange', 4)])
def test_sharded_on_gpu(case, world):
    _run(case, world)
