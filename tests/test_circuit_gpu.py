"""End-to-end parity on the MI355X: the QubitCircuit API running on the HIP kernels against the golden
vectors of the real reference (tolerances of the north star: 1e-10 complex128, 1e-4 complex64), plus
size-independent properties at the benchmark's full size where no oracle can run."""

import math

import pytest
import torch

import deepquantum_amd as dq
from _helpers import TOL, check_circuit_against_golden, gold, specs

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


@pytest.mark.parametrize('name', ['readme', 'zoo5', 'rand4', 'rand8', 'rand12', 'rand16', 'rand14_seed1234', 'batched10'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_circuits_match_reference(name, prec):
    err = check_circuit_against_golden(dq, name, prec, device=dev())
    n = specs.CIRCUITS[name]['nqubit']
    m = 12 if prec == 'c64' else 11
    if n >= m:
        assert dq.executor.LAST_RUN['passes'] > 0, 'fused HIP path did not run'
    print(f'{name}/{prec}: max amplitude error {err:.3e}')


def test_gate_classes_match_reference():
    for i, case in enumerate(specs.GATE_CASES):
        gate = getattr(dq, case['cls'])(nqubit=case['nqubit'], **case['kwargs'])
        if case.get('inverse'):
            gate = gate.inverse()
        gate = gate.to(dev()).to(torch.double)
        with torch.no_grad():
            out = gate(gold(f'gate/{i}/in').to(dev()))
        assert (out.cpu() - gold(f'gate/{i}/out')).abs().max().item() < 1e-10, (i, case['cls'])


def test_config2_pin_n24_complex128():
    """BASELINE config 2: QubitCircuit(24), depth 20, complex128 -- 1024 seeded amplitudes, squared norm and
    <Z0> of the real reference (note the norm is 0.9999954..., not 1: float32-rounded constants)."""
    n = 24
    cir = specs.build(dq, n, specs.random_spec(n, 20, 1234))
    cir.observable(0)
    cir.to(dev()).to(torch.double)
    with torch.no_grad():
        state = cir().reshape(-1)
        ev = cir.expectation()
    idx = gold('pin24/indices').to(dev())
    assert (state[idx].cpu() - gold('pin24/amplitudes')).abs().max().item() < 1e-10
    assert abs((state.abs() ** 2).sum().item() - gold('pin24/norm2').item()) < 1e-10
    assert abs(ev.item() - gold('pin24/expectation_z0').item()) < 1e-10
    assert dq.executor.LAST_RUN['passes'] > 0


def _grad_check(builder, tag, nqubit):
    cir = builder(dq, nqubit).to(dev())
    params = gold(f'grad/{tag}/params')
    off = 0
    with torch.no_grad():
        for p in cir.parameters():
            p.copy_(params[off : off + p.numel()].reshape(p.shape))
            off += p.numel()
    data = gold(f'grad/{tag}/data').to(dev()).requires_grad_(True)
    cir(data)
    ev = cir.expectation()
    assert (ev.cpu() - gold(f'grad/{tag}/expectation')).abs().max().item() < 1e-4
    ev.sum().backward()
    assert (data.grad.cpu() - gold(f'grad/{tag}/data_grad')).abs().max().item() < 1e-4
    if params.numel():
        pg = torch.cat([p.grad.reshape(-1) for p in cir.parameters()]).cpu()
        assert (pg - gold(f'grad/{tag}/param_grads')).abs().max().item() < 1e-4


def test_autograd_through_hip_kernels():
    _grad_check(specs.grad_circuit_a, 'mixed', 4)
    _grad_check(specs.qaoa_circuit, 'qaoa', 7)


def test_autograd_matches_between_fused_and_eager_at_n13():
    """Gradient of a 13-qubit circuit (eager autograd kernels) vs finite differences of the fused forward."""
    n = 13
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.cnot_ring()
    cir.ry(7, encode=True)
    cir.rzz([2, 11], encode=True)
    cir.observable([0, 5], 'zx')
    cir.to(dev()).to(torch.double)
    data = torch.tensor([0.3, 0.9, -0.5], dtype=torch.float64, device=dev(), requires_grad=True)
    cir(data)
    ev = cir.expectation().sum()
    ev.backward()
    eps = 1e-5
    for k in range(3):
        d = data.detach().clone()
        with torch.no_grad():
            d[k] += eps
            cir(d)
            up = cir.expectation().sum().item()
            d[k] -= 2 * eps
            cir(d)
            dn = cir.expectation().sum().item()
        assert abs((up - dn) / (2 * eps) - data.grad[k].item()) < 1e-6


@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_measure_marginals_and_sampling(prec):
    n = 14
    cir = dq.QubitCircuit(n)
    cir.h(0)
    for q in range(1, n):
        cir.cnot(0, q)
    cir.to(dev())
    if prec == 'c128':
        cir.to(torch.double)
    with torch.no_grad():
        cir()
    res = cir.measure(shots=500)
    assert set(res) <= {'0' * n, '1' * n} and sum(res.values()) == 500
    assert 150 < res.get('0' * n, 0) < 350
    res = cir.measure(shots=100, wires=[0, 7, 13], with_prob=True)
    assert set(res) <= {'000', '111'}
    assert all(abs(float(v[1]) - 0.5) < 1e-5 for v in res.values())


@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_marginals_over_more_than_twelve_wires(prec):
    """dq_marginal_* with 13 .. n measured bits (LDS histograms; reference qmath.py:624-626) against permute / reshape / sum of
    |psi|^2 in float64, for wire sets that do and do not contain the low index bits, batched."""
    from deepquantum_amd import backend

    n, b = 18, 3
    dtype = torch.complex64 if prec == 'c64' else torch.complex128
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, 1 << n, generator=g, dtype=torch.float64) + 1j * torch.randn(b, 1 << n, generator=g, dtype=torch.float64)
    x = (x / x.norm(dim=-1, keepdim=True)).to(dtype)
    p = (x.to(torch.complex128).abs() ** 2).reshape([b] + [2] * n)
    for bits in ([17, 3, 5, 0, 1, 2, 9, 16, 4, 12, 13, 7, 8], list(range(17, 3, -1)), list(range(13)),
                 [2 * i % n if i < 9 else 2 * (i - 9) + 1 for i in range(16)], list(range(n))[::-1]):
        assert len(set(bits)) == len(bits) > 12
        got = backend.marginal(x.to(dev()), bits).cpu()
        axes = [1 + (n - 1 - q) for q in bits]                 # index bit q = tensor axis n - q (axis 0 = batch)
        ref = p.permute([0] + axes + [a for a in range(1, n + 1) if a not in axes]).reshape(b, 1 << len(bits), -1).sum(-1)
        assert (got - ref).abs().max().item() < (1e-9 if prec == 'c64' else 1e-13), bits
    # and through the circuit-level entry point
    cir = dq.QubitCircuit(16)
    cir.hlayer(list(range(0, 16, 2)))
    cir.to(dev())
    with torch.no_grad():
        cir()
    res = cir.measure(shots=64, wires=list(range(14)), with_prob=True)
    assert all(len(k) == 14 and k[1::2] == '0' * 7 and abs(float(v[1]) - 2.0 ** -7) < 1e-6 for k, v in res.items())


def test_sampling_a_state_of_more_than_one_block():
    """measure() on 2^26 outcomes: block_sample's multi-block branch (reference qmath.py:543-565: blocks of 2^24
    probabilities, a block first, then an outcome inside it)."""
    n = 26
    cir = dq.QubitCircuit(n)
    cir.h(0)                       # wire 0 = index bit 25: the two halves of the outcome range live in different blocks
    cir.h(n - 1)
    cir.x(3)
    cir.to(dev())
    with torch.no_grad():
        cir()
    res = cir.measure(shots=400)
    want = {a + '00' + '1' + '0' * (n - 5) + c for a in '01' for c in '01'}
    assert set(res) <= want and sum(res.values()) == 400
    assert all(40 < res.get(k, 0) < 170 for k in want), res
    res = cir.measure(shots=200, wires=list(range(13)), with_prob=True)          # 13 wires
    assert set(res) <= {'0001' + '0' * 9, '1001' + '0' * 9} and all(abs(float(v[1]) - 0.5) < 1e-6 for v in res.values())


# ---- properties at the benchmark's full size (BASELINE config 3: n=28, complex64) -------------------------
def test_full_size_ghz_and_norm_n28():
    n = 28
    cir = dq.QubitCircuit(n)
    cir.h(0)
    for q in range(1, n):
        cir.cnot(q - 1, q)
    cir.observable([0, n - 1], 'zz')
    cir.observable(0)
    cir.to(dev())
    with torch.no_grad():
        state = cir().reshape(-1)
        ev = cir.expectation()
    s = 0.7071067690849304  # the float32-rounded 1/sqrt(2) of the reference's Hadamard
    assert abs(state[0].item() - s) < 1e-6 and abs(state[-1].item() - s) < 1e-6
    assert abs((state.abs() ** 2).sum().item() - 2 * s * s) < 1e-5
    assert abs(ev[0].item() - 2 * s * s) < 1e-5 and abs(ev[1].item()) < 1e-6
    assert int((state != 0).sum().item()) == 2


def test_full_size_random_circuit_times_inverse_n28_batched():
    """U^-1 U |0> = |0> on the headline workload shape (n=28, depth 6, batch 2 with per-sample angles)."""
    n, depth, b = 28, 6, 2
    spec = specs.random_spec(n, depth, 4321)
    cir = dq.QubitCircuit(n)
    nrx = 0
    for m, args, _ in spec:
        if m == 'rx':
            cir.rx(args[0], encode=True)
            nrx += 1
        else:
            getattr(cir, m)(*args)
    full = cir + cir.inverse(encode=True)
    full.to(dev())
    g = torch.Generator().manual_seed(1)
    data = (torch.rand(b, nrx, generator=g) * 2 * math.pi)
    both = torch.cat([data, data.flip(-1)], dim=-1).to(dev())  # the inverse consumes the angles in reverse order
    with torch.no_grad():
        out = full(both).reshape(b, -1)
    assert dq.executor.LAST_RUN['passes'] > 0
    assert (out[:, 0].abs() - 1).abs().max().item() < 1e-3      # float32-rounded H: |amp| drifts ~1e-7 per gate
    assert out[:, 1:].abs().max().item() < 1e-3


def test_full_size_hlayer_uniform_n28():
    n = 28
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    for q in range(n):
        cir.observable(q, 'x')
    cir.to(dev())
    with torch.no_grad():
        state = cir().reshape(-1)
        ev = cir.expectation()
    amp = 0.7071067690849304**n
    assert (state.real - amp).abs().max().item() < 1e-9 and state.imag.abs().max().item() == 0
    assert (ev - (2 * 0.7071067690849304**2) ** n).abs().max().item() < 1e-4


def test_torch_vmap_over_the_circuit_on_gpu():
    cir = dq.QubitCircuit(6)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.cnot(0, 4)
    cir.ry(3, encode=True)
    cir.crx(1, 5, encode=True)
    cir.rzz([0, 3], encode=True)
    cir.observable([0, 3], 'zx')
    cir.to(dev())
    data = torch.rand(7, 4, generator=torch.Generator().manual_seed(0)).to(dev())
    with torch.no_grad():
        native = cir(data)
        ev = cir.expectation()
        vm = torch.vmap(cir._forward_helper, in_dims=(0, None))(data, cir.init_state.state)
    assert (vm - native).abs().max().item() < 1e-6
    assert ev.shape == (7, 1)


def test_reset_matches_reference_on_gpu():
    from _helpers import check_reset_against_golden

    check_reset_against_golden(dq, device=dev())


def test_ansatz_library_on_gpu():
    """QFT / QPE / Beauregard arithmetic / Shor / HHL / QCNN built with the reference's
    constructors: amplitudes equal to the reference's, known answers of its tests/test_ansatz.py (Shor with
    8 counting qubits = 18 qubits, ~6000 multi-controlled phase gates through the fused passes)."""
    from _ansatz_checks import check_known_answers, check_qcnn, check_states

    check_states(dq, device=dev())
    check_qcnn(dq, device=dev())
    check_known_answers(dq, device=dev(), shor_ncount=8)


def test_density_matrix_path_on_gpu():
    from _helpers import check_density_matrix_against_golden

    check_density_matrix_against_golden(dq, device=dev())


def test_adjoint_grad_mode_on_gpu():
    from _helpers import check_adjoint_grad_mode

    check_adjoint_grad_mode(dq, device=dev(), dtype=torch.float64, tol=1e-10)
    check_adjoint_grad_mode(dq, device=dev(), dtype=torch.float32, tol=2e-5)
    # 13 qubits: the undo stretches between trainable gates run as fused passes
    check_adjoint_grad_mode(dq, device=dev(), dtype=torch.float64, n=13, tol=1e-10)


def test_fused_reverse_sweep_on_gpu():
    """complex64: the reverse sweep as fused passes with the reductions inside (dq_apply_fused_grad_c64) against
    per-gate autograd and the undo-then-reduce sweep: 12-bit and 13-bit tiles, one sample and a batch."""
    from _helpers import check_fused_sweep

    check_fused_sweep(dq, device=dev(), n=11, batch=2)
    check_fused_sweep(dq, device=dev(), n=12, batch=1)
    check_fused_sweep(dq, device=dev(), n=16, batch=3)
    check_fused_sweep(dq, device=dev(), n=20, batch=2, tol=6e-5)


def test_fused_reverse_sweep_c128_on_gpu():
    """complex128: the same sweep on the wave-tile kernel (dq_apply_fused_grad_c128, float64 sums) at the north star's
    1e-10 against per-gate autograd and the undo-then-reduce sweep."""
    from _helpers import check_fused_sweep

    check_fused_sweep(dq, device=dev(), n=10, batch=2, tol=1e-10, dtype=torch.float64)
    check_fused_sweep(dq, device=dev(), n=12, batch=1, tol=1e-10, dtype=torch.float64)
    check_fused_sweep(dq, device=dev(), n=16, batch=3, tol=1e-10, dtype=torch.float64)
    check_fused_sweep(dq, device=dev(), n=20, batch=2, tol=1e-10, dtype=torch.float64)


@pytest.mark.gpu
def test_fused_reverse_sweep_of_states_smaller_than_a_tile():
    """n + 1 < m: the (psi, lambda) pair is zero-padded to one tile and swept in fused passes with the reductions inside
    (executor.CONFIG['small_fused_sweep']) instead of a pass per layer and a reduction launch per trainable gate: against
    per-gate autograd and against the undo-then-reduce sweep, both precisions, with and without a batch."""
    from _helpers import check_fused_sweep, check_fused_sweep_random

    check_fused_sweep(dq, device=dev(), n=8, batch=2)
    check_fused_sweep(dq, device=dev(), n=10, batch=1)
    check_fused_sweep(dq, device=dev(), n=8, batch=3, tol=1e-10, dtype=torch.float64)
    check_fused_sweep(dq, device=dev(), n=9, batch=1, tol=1e-10, dtype=torch.float64)
    for seed, n in enumerate((3, 4, 6, 9)):
        check_fused_sweep_random(dq, device=dev(), n=n, batch=1 + seed % 2, seed=seed, ngates=40)
        check_fused_sweep_random(dq, device=dev(), n=n, batch=1 + seed % 2, seed=seed, ngates=40, tol=1e-10, dtype=torch.float64)
    dq.executor.CONFIG['small_fused_sweep'] = False
    try:
        check_fused_sweep_random(dq, device=dev(), n=6, batch=2, seed=9, ngates=40, expect_fused=False)
    finally:
        dq.executor.CONFIG['small_fused_sweep'] = True


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_fused_reverse_sweep_on_random_circuits_on_gpu(seed):
    from _helpers import check_fused_sweep_random

    check_fused_sweep_random(dq, device=dev(), n=13 + seed % 3, batch=1 + seed % 2, seed=seed)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_fused_reverse_sweep_c128_on_random_circuits_on_gpu(seed):
    from _helpers import check_fused_sweep_random

    check_fused_sweep_random(dq, device=dev(), n=11 + 2 * seed, batch=1 + seed % 2, seed=seed, tol=1e-10, dtype=torch.float64)


def test_adjoint_backward_memory_is_independent_of_depth():
    """Training step at n = 24 (128 MiB per state), 480 gates: stock per-gate autograd would hold one state per
    gate (~60 GiB); the adjoint node peaks at a handful of states."""
    n, depth = 24, 20
    cir = dq.QubitCircuit(n)
    from oracle.statevec_oracle import random_circuit_spec

    nrx = 0
    for op in random_circuit_spec(n, depth, 1234):
        if op[0] == 'h':
            cir.h(op[1])
        elif op[0] == 'rx':
            cir.rx(op[1])            # trainable
            nrx += 1
        else:
            cir.cnot(op[1], op[2])
    cir.observable(0)
    cir.to(dev())
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    cir()
    loss = cir.expectation().sum()
    loss.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    state_bytes = 8 * 2**n
    grads = [p.grad for p in cir.parameters()]
    assert len(grads) == nrx and all(g is not None and torch.isfinite(g).all() for g in grads)
    assert peak < 8 * state_bytes, f'peak {peak / state_bytes:.1f} states'
    print(f'n={n}: {nrx} trainable angles, peak memory {peak / state_bytes:.1f} states')


def _qml_circuit(n, trainable, seed=1234):
    import bench

    cir = dq.QubitCircuit(n)
    for op in bench.random_circuit_spec(n, 12, seed):
        if op[0] == 'h':
            cir.h(op[1])
        elif op[0] == 'rx':
            cir.rx(op[1]) if trainable else cir.rx(op[1], encode=True)
        else:
            cir.cnot(op[1], op[2])
    cir.observable(0)
    cir.observable([1, 2], 'zx')
    return cir.to(dev())


@pytest.mark.parametrize('n', [7, 12])
def test_hip_graph_capture_of_forward_and_training_step(n):
    """Launch-bound sizes: the whole evaluation (and the whole forward + backward) replays as one HIP graph; the
    backward is the fused reverse sweep (dq_apply_fused_grad_c64) inside the captured graph -- at n = 7 on the (psi, lambda)
    pair zero-padded to one tile."""
    cir = _qml_circuit(n, trainable=False)
    data = torch.zeros(16, cir.ndata, device=dev())
    with torch.no_grad():
        graph = dq.CapturedGraph(lambda: (cir(data), cir.expectation())[1])
        for seed in (1, 2):
            batch = torch.rand(16, cir.ndata, generator=torch.Generator().manual_seed(seed)).to(dev()) * 6.28
            data.copy_(batch)
            got = graph.replay().clone()
            ref_cir = _qml_circuit(n, trainable=False)
            ref_cir(batch)
            assert (got - ref_cir.expectation()).abs().max().item() < 1e-5

    for mode in ('adjoint', 'per_gate'):
        dq.executor.CONFIG['grad_mode'] = mode
        try:
            torch.manual_seed(5)
            train = _qml_circuit(n, trainable=True)
            torch.manual_seed(5)
            eager = _qml_circuit(n, trainable=True)
            for p, q in zip(train.parameters(), eager.parameters(), strict=True):
                assert torch.equal(p, q)

            def step():
                train()
                loss = train.expectation().sum()
                loss.backward()
                return loss

            train.zero_grad(set_to_none=True)
            graph = dq.CapturedGraph(step)
            assert mode != 'adjoint' or dq.executor.LAST_SWEEP['fused']      # (n = 7: on the zero-padded pair)
            for it in range(2):
                for p in train.parameters():
                    p.grad.zero_()
                loss = graph.replay()
                eager.zero_grad(set_to_none=True)
                eager()
                ref = eager.expectation().sum()
                ref.backward()
                assert abs(loss.item() - ref.item()) < 1e-5
                for p, q in zip(train.parameters(), eager.parameters(), strict=True):
                    assert (p.grad - q.grad).abs().max().item() < 1e-4, mode
                with torch.no_grad():       # an SGD step on both, then replay again
                    for p, q in zip(train.parameters(), eager.parameters(), strict=True):
                        p.sub_(0.1 * p.grad)
                        q.sub_(0.1 * q.grad)
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'


@pytest.mark.parametrize('n', [12, 22])
def test_training_step_captured_after_eager_steps_on_the_default_stream(n):
    """What a user does: train eagerly for a while (default stream), then capture the step.  The gates used to keep the last
    matrices -- with their autograd graphs -- in their primitive caches, so the parameters' AccumulateGrad nodes (made on
    the default stream) lived on into the capture and dragged the legacy stream into it: hipStreamEndCapture died.
    n = 22: the state is far beyond the launch-bound sizes (permuted stores, second buffer from the graph's pool)."""
    torch.manual_seed(3)
    train = _qml_circuit(n, trainable=True)
    torch.manual_seed(3)
    eager = _qml_circuit(n, trainable=True)

    def step():
        train()
        loss = train.expectation().sum()
        loss.backward()
        return loss

    for _ in range(3):                      # eager steps on the default stream, grads zeroed in place
        train.zero_grad()
        step()
    with torch.no_grad():
        train()                             # and a no-grad forward, as a validation loop would do
    torch.cuda.synchronize()
    for g in train.modules():               # nothing of those steps' graphs is kept by the gates
        c = g.__dict__.get('_prims_cache')
        assert c is None or not c[0].requires_grad
    train.zero_grad(set_to_none=True)
    graph = dq.CapturedGraph(step)
    for p in train.parameters():
        p.grad.zero_()
    loss = graph.replay()
    eager()
    ref = eager.expectation().sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    for p, q in zip(train.parameters(), eager.parameters(), strict=True):
        assert (p.grad - q.grad).abs().max().item() < 1e-4


def test_captured_training_step_keeps_the_device_records_of_its_long_sweep_passes_alive():
    """ABI 24 + HIP graphs: the reverse sweep of the reference's chart circuit (10 qubits x 6 layers: 180 reductions + 234
    gates) has passes of more than 112 records, whose records live in device memory, copied there on the first (eager) run.
    A captured graph has their addresses baked in: they must survive the plan that made them."""
    import gc

    n, layers = 10, 6
    cir = dq.QubitCircuit(n)
    for _ in range(layers):
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        cir.rxlayer(encode=True)
        cir.rzlayer(encode=True)
        cir.rxlayer(encode=True)
    cir.observable(basis='x')
    cir.to(dev())
    params = torch.linspace(0.1, 2.9, 3 * n * layers, device=dev()).requires_grad_(True)

    def step():
        if params.grad is not None:
            params.grad.zero_()
        cir(data=params)
        cir.expectation().backward()
        return params.grad

    # eager steps on the default stream first, as a user would: the encoders let go of the previous call's angles before a
    # new graph is made, so that no AccumulateGrad node of the leaf lives on into the capture
    # (test_training_step_captured_after_eager_steps_on_the_default_stream has the same story for trainable gates)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    graph = dq.CapturedGraph(step)
    assert dq.executor.LAST_SWEEP['fused']
    # (the sweep is the last fused run of the step: its long passes keep their records in device memory, and the capture has
    # put exactly those tensors on the list that outlives the plans)
    held = [t for st in dq.executor.LAST_RUN['plan'].steps if hasattr(st, 'desc')
            for t in st.desc.__dict__.get('_dev_records', {}).values() if t is not None]
    assert held, 'no pass of the captured sweep kept its records in device memory'
    # (the pins of a capture belong to ITS CapturedGraph and die with it; nothing the capture allocated itself --
    # the matrix buffer of this trainable circuit -- is pinned)
    assert all(id(t) in graph._pins for t in held)
    assert all(getattr(t, '_dq_owned', False) for t in graph._pins.values())
    del held
    first = graph.replay().clone()
    dq.executor._PLAN_CACHE.clear()         # the plans (and with them the descriptors' own references) go away ...
    dq.executor._STEADY.clear()
    gc.collect()
    junk = [torch.full((1 << 20,), 7, dtype=torch.uint8, device=dev()) for _ in range(64)]      # ... and freed memory is reused
    del junk
    for _ in range(2):
        got = graph.replay()
        assert (got - first).abs().max().item() < 1e-6      # (sums of atomics: equal to rounding)
    got = got.clone()
    params.grad = None
    ref = step()                            # eager, after the graph: the same numbers
    assert (got - ref).abs().max().item() < 1e-5 and float(ref.abs().max()) > 1e-3


@pytest.mark.parametrize('n', [8, 14])
def test_captured_inference_survives_the_eviction_of_its_plans_and_caches(n):
    """A no-grad evaluation captured as a HIP graph reads what the steady-state caches own -- the flat matrix buffer of the
    fixed gates, the offsets of the deferred Rx blocks: dropped caches and reused memory must not reach the replay."""
    import gc

    torch.manual_seed(2)
    cir = dq.QubitCircuit(n)
    cir.hlayer()
    for _ in range(4):
        cir.rxlayer()
        cir.cnot_ring()
        cir.rylayer(encode=True)
    cir.observable(0)
    cir.observable([1, 2], 'zz')
    cir.to(dev())
    data = torch.rand(3, cir.ndata, device=dev())

    def evaluate():
        with torch.no_grad():
            cir(data)
            return cir.expectation()

    graph = dq.CapturedGraph(evaluate)
    first = graph.replay().clone()
    dq.executor._PLAN_CACHE.clear()
    dq.executor._STEADY.clear()
    gc.collect()
    junk = [torch.full((1 << 18,), 5, dtype=torch.uint8, device=dev()) for _ in range(256)]
    junk += [torch.full((1 << 10,), 5, dtype=torch.uint8, device=dev()) for _ in range(4096)]
    del junk
    for _ in range(2):       # (the reductions add with atomics: equal to rounding, not bit for bit)
        assert (graph.replay() - first).abs().max().item() < 1e-5
    assert (evaluate() - first).abs().max().item() < 1e-5 and float(first.abs().max()) > 1e-3


def test_edge_cases_on_gpu():
    from _helpers import check_edge_cases

    check_edge_cases(dq, device=dev())


@pytest.mark.parametrize('n,dtype', [(32, torch.float32), (31, torch.float64)])
def test_largest_single_gpu_states(n, dtype):
    """Maximum sizes (no oracle can hold these): a 32-qubit complex64 state is 32 GiB, a 31-qubit
    complex128 one too.  Known answers: GHZ, then a layer of parametrised gates and its inverse on top."""
    cir = dq.QubitCircuit(n)
    cir.h(0)
    for q in range(1, n):
        cir.cnot(q - 1, q)
    layer = dq.QubitCircuit(n)
    for q in range(0, n, 3):
        layer.rx(q, 0.1 + 0.05 * q)
        layer.ry((q + 1) % n, 0.2 + 0.03 * q)
    layer.cz(0, n - 1)
    layer.rzz([1, n - 2], 0.4)
    full = cir + layer + layer.inverse()
    full.observable([0, n - 1], 'zz')
    full.observable(n // 2, 'x')
    full.to(dev())
    if dtype == torch.float64:
        full.to(torch.double)
    with torch.no_grad():
        state = full().reshape(-1)
        ev = full.expectation()
    assert state.numel() == 2**n and dq.executor.LAST_RUN['passes'] > 0
    s = 0.7071067690849304
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert abs(state[0].item() - s) < tol and abs(state[-1].item() - s) < tol
    assert abs((state.abs() ** 2).sum().item() - 2 * s * s) < 10 * tol
    assert abs(ev[0].item() - 2 * s * s) < 10 * tol and abs(ev[1].item()) < 10 * tol
    del state
    torch.cuda.empty_cache()


def test_many_z_observables_in_one_pass_on_gpu():
    from _helpers import check_many_z_observables

    check_many_z_observables(dq, device=dev(), dtype=torch.float64)
    check_many_z_observables(dq, device=dev(), dtype=torch.float32)
    check_many_z_observables(dq, device=dev(), dtype=torch.float32, n=14)     # fused passes + 16 Z-type strings


def test_readout_functions_match_reference_on_gpu():
    from _helpers import check_readout_against_golden

    check_readout_against_golden(dq, device=dev())


def test_remaining_gate_classes_match_reference_on_gpu():
    from _helpers import check_extra_gates_against_golden

    check_extra_gates_against_golden(dq, device=dev())


def test_bench_prints_one_contract_line():
    """`python bench.py` (small workload) prints exactly one JSON line carrying the driver's contract keys plus the
    `roofline` and `cpu_baseline` objects."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--nqubit', '16', '--batch', '4', '--depth', '6',
                          '--steps', '2', '--warmup', '1', '--cpu-seconds', '1'],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'gate-applies/s' and d['dtype'] == 'c64' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 16 * 6 * 4 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and r['launches'] > 0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 'traffic' in r
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'gate-applies/s' and c['sample']


def test_random_circuits_under_every_scheduler_configuration_on_gpu():
    from _helpers import check_fuzz_against_oracle
    check_fuzz_against_oracle(dq, device=dev(), n=13, seeds=(0, 1, 2, 3), depth=6)
    check_fuzz_against_oracle(dq, device=dev(), n=16, seeds=(4, 5), depth=5, batch=3)
    check_fuzz_against_oracle(dq, device=dev(), n=12, seeds=(6, 7, 8), depth=6, double=True)
    check_fuzz_against_oracle(dq, device=dev(), n=15, seeds=(9,), depth=5, double=True)


def test_qasm_programs_run_on_the_hip_path():
    """an imported OpenQASM 3 program executes on the HIP kernels and reaches the state the reference
    computed for it (tests/golden/golden_qasm.json), and a circuit written out as QASM 3 and read back runs to the
    same state as the original."""
    import json
    import os

    gold_q = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_qasm.json')))
    for name, g in gold_q['import'].items():
        cir = dq.qasm3_to_cir(g['program']).to(dev())
        with torch.no_grad():
            got = cir().reshape(-1)
        assert got.is_cuda
        ref = torch.tensor(g['state'], dtype=torch.float64)
        err = (torch.view_as_real(got.cpu().to(torch.complex128)) - ref).abs().max().item()
        assert err < 1e-5, (name, err)
    # a 14-qubit circuit big enough for the fused passes: write -> read -> run
    cir = specs.build(dq, 14, specs.random_spec(14, 5, 11))
    cir.u3(3, [0.3, 0.9, -0.4])
    cir.rzz([2, 9], 0.7)
    cir.toffoli(0, 5, 13)
    again = dq.qasm3_to_cir(dq.cir_to_qasm3(cir)).to(dev())
    cir.to(dev())
    with torch.no_grad():
        a, b = cir().reshape(-1), again().reshape(-1)
    assert dq.executor.LAST_RUN['passes'] > 0
    assert (a - b).abs().max().item() < 1e-5


def test_config3_pin_n28_complex64_batch16():
    """BASELINE config 3 -- the TIMED workload of bench.py, exactly as it runs there (batch 16 with per-sample angles,
    one-qubit runs merged, permuted stores, next-tile prefetch) -- against what the REAL reference computed for batch
    element 0 (tests/golden/pin28.npz <- make_golden_pin28.py, 76 minutes of the reference on 8 cores): 4096 seeded
    amplitudes, the squared norm, <Z_q> on every wire and two 5-wire marginals, at the north star's 1e-4."""
    import os
    import sys

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from deepquantum_amd import backend

    free, _total = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip('needs 80 GiB of free device memory')
    pin = np.load(os.path.join(root, 'tests', 'golden', 'pin28.npz'))
    n, depth, batch = 28, 40, 16
    spec = bench.random_circuit_spec(n, depth, 1234)
    cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex64, dev())
    assert np.array_equal(data[0].cpu().numpy(), pin['angles_f32'])      # sample 0 = the reference's own angles
    with torch.no_grad():
        cir(data)
        ev = cir.expectation()
    assert dq.executor.LAST_RUN['passes'] > 0 and dq.executor.LAST_RUN['gates'] < len(spec)   # merged, fused
    state = cir.state.reshape(batch, 1 << n)[:1].contiguous()
    idx = torch.from_numpy(pin['indices']).to(state.device)
    got = state[0, idx].cpu().numpy()
    assert np.abs(got - pin['amplitudes']).max() < 1e-4
    # the criteria that bite at n = 28 (amplitudes ~ 1.6e-4: the absolute 1e-4 is vacuous there) are RELATIVE: the largest
    # deviation against the largest pinned amplitude and l2 against l2, both at 1e-4 (measured: 5.4e-6 / 5.5e-6)
    assert np.abs(got - pin['amplitudes']).max() < 1e-4 * np.abs(pin['amplitudes']).max()
    assert np.linalg.norm(got - pin['amplitudes']) < 1e-4 * np.linalg.norm(pin['amplitudes'])
    assert abs(float(backend.expect_pauli(state, 0, 0)[0]) - float(pin['norm2'])) < 1e-4
    ez = np.array([float(backend.expect_pauli(state, 0, 1 << (n - 1 - q))[0]) for q in range(n)])
    assert np.abs(ez - pin['expectation_z']).max() < 1e-4
    assert abs(float(ev.reshape(batch, -1)[0, 0]) - float(pin['expectation_z'][0])) < 1e-4
    p0 = backend.marginal(state, [n - 1 - w for w in range(5)]).reshape(-1).cpu().numpy()
    p1 = backend.marginal(state, [n - 1 - w for w in range(n - 5, n)]).reshape(-1).cpu().numpy()
    assert np.abs(p0 - pin['marginal_wires_0_4']).max() < 1e-4
    assert np.abs(p1 - pin['marginal_wires_last5']).max() < 1e-4
    # the other samples carry other angles: they differ from sample 0 but stay normalised the same way
    full = cir.state.reshape(batch, 1 << n)
    norms = backend.expect_pauli(full, 0, 0).cpu().numpy()
    assert np.abs(norms - 1).max() < 1e-4
    evb = ev.reshape(batch, -1)[:, 0].cpu().numpy().copy()
    # a second sample of the timed batch against the REAL reference (make_golden_pin28.py --sample 7: row 7 of the same
    # data matrix through the reference's un-batched forward; vmap semantics circuit.py:232-240)
    for name in sorted(os.listdir(os.path.join(root, 'tests', 'golden'))):
        if not (name.startswith('pin28_s') and name.endswith('.npz')):
            continue
        pk = np.load(os.path.join(root, 'tests', 'golden', name))
        k = int(pk['sample'])
        assert np.array_equal(data[k].cpu().numpy(), pk['angles_f32'])
        gk = full[k, torch.from_numpy(pk['indices']).to(full.device)].cpu().numpy()
        assert np.abs(gk - pk['amplitudes']).max() < 1e-4 * np.abs(pk['amplitudes']).max(), name
        assert np.linalg.norm(gk - pk['amplitudes']) < 1e-4 * np.linalg.norm(pk['amplitudes']), name
        assert abs(float(norms[k]) - float(pk['norm2'])) < 1e-4
        assert abs(float(evb[k]) - float(pk['expectation_z'][0])) < 1e-4
    # samples 1 .. 15 against UN-BATCHED runs of the same angles through the same circuit (shared matrices, no per-sample
    # stride): the whole 2^28 amplitudes of each, relative to the largest one and in l2
    single, _ = bench.build_circuit(dq, n, spec, None, torch.complex64, dev())
    for k in range(1, batch):
        with torch.no_grad():
            single(data[k])
            e1 = single.expectation()
        one = single.state.reshape(-1)
        diff = (one - full[k])
        scale = one.abs().max().item()
        assert diff.abs().max().item() < 1e-4 * scale, k
        assert torch.linalg.vector_norm(diff).item() < 1e-4 * torch.linalg.vector_norm(one).item(), k
        assert abs(float(e1.reshape(-1)[0]) - float(evb[k])) < 1e-5, k
        del diff, one


def test_pin_n26_complex128_batch4():
    """The benchmark circuit in complex128 at the size the wave-tile kernel's big-state paths start (n = 26, depth 40, 1040
    gates; batch 4 = 4 GiB: merged one-qubit runs, permuted stores, streaming accesses, <Z0> out of the last pass) against
    what the REAL reference computed in double precision for batch element 0 (tests/golden/pin26_d40_c128.npz <-
    PIN_N=26 PIN_DTYPE=c128 make_golden_pin28.py, 17 minutes of the reference): 4096 seeded amplitudes, the squared
    norm, <Z_q> on every wire and two 5-wire marginals, at the north star's 1e-10."""
    import os
    import sys

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from deepquantum_amd import backend

    pin = np.load(os.path.join(root, 'tests', 'golden', 'pin26_d40_c128.npz'))
    n, depth, batch = 26, 40, 4
    assert int(pin['nqubit']) == n and int(pin['depth']) == depth
    spec = bench.random_circuit_spec(n, depth, 1234)
    cir, data = bench.build_circuit(dq, n, spec, batch, torch.complex128, dev())
    assert data.dtype == torch.float64 and np.array_equal(data[0].cpu().numpy(), pin['angles_f32'])
    with torch.no_grad():
        cir(data)
        ev = cir.expectation()
    assert dq.executor.LAST_RUN['passes'] > 0 and dq.executor.LAST_RUN['gates'] < len(spec) + 1   # merged, fused
    assert cir._expz is not None
    state = cir.state.reshape(batch, 1 << n)[:1].contiguous()
    idx = torch.from_numpy(pin['indices']).to(state.device)
    assert np.abs(state[0, idx].cpu().numpy() - pin['amplitudes']).max() < 1e-10
    assert abs(float(backend.expect_pauli(state, 0, 0)[0]) - float(pin['norm2'])) < 1e-10
    ez = np.array([float(backend.expect_pauli(state, 0, 1 << (n - 1 - q))[0]) for q in range(n)])
    assert np.abs(ez - pin['expectation_z']).max() < 1e-10
    assert abs(float(ev.reshape(batch, -1)[0, 0]) - float(pin['expectation_z'][0])) < 1e-10        # (from the last pass)
    p0 = backend.marginal(state, [n - 1 - w for w in range(5)]).reshape(-1).cpu().numpy()
    p1 = backend.marginal(state, [n - 1 - w for w in range(n - 5, n)]).reshape(-1).cpu().numpy()
    assert np.abs(p0 - pin['marginal_wires_0_4']).max() < 1e-10
    assert np.abs(p1 - pin['marginal_wires_last5']).max() < 1e-10


@pytest.mark.parametrize('n,dt,batch,tol', [(14, torch.float32, 4, 3e-6), (20, torch.float32, None, 3e-6), (22, torch.float32, 3, 3e-6),
                                            (13, torch.float64, 2, 1e-12), (21, torch.float64, None, 1e-12)])
def test_z_expectations_come_out_of_the_last_pass_on_gpu(n, dt, batch, tol):
    """`expectation()` after a no-grad forward: Z-type observables from the registers of the last pass (DQ_FG_EXPZ), the
    others from the state; equal to the separate reductions."""
    def build():
        torch.manual_seed(1)
        c = dq.QubitCircuit(n)
        c.hlayer(); c.rxlayer(encode=True); c.cnot_ring(); c.rylayer(); c.rz(0, encode=True); c.cnot_ring(reverse=True); c.hlayer()
        c.observable(0); c.observable([1, n - 1], 'zz'); c.observable([2, 3], 'xy'); c.observable(list(range(n)), 'z' * n)
        c.to(dev())
        if dt == torch.float64:
            c.to(torch.double)
        return c
    data = torch.rand((batch, n + 1) if batch else (n + 1,), dtype=dt, device=dev())
    res = {}
    for fused in (True, False):
        dq.executor.CONFIG['fused_expectation'] = fused
        try:
            c = build()
            with torch.no_grad():
                c(data)
                assert (c._expz is not None) == fused
                res[fused] = c.expectation()
        finally:
            dq.executor.CONFIG['fused_expectation'] = True
    assert res[True].shape == res[False].shape and (res[True] - res[False]).abs().max().item() < tol, (res[True], res[False])


def test_fused_reverse_sweep_with_user_matrices_that_are_unitary_to_1e_4_only_on_gpu():
    from _helpers import check_fused_sweep_with_sloppy_user_matrices

    check_fused_sweep_with_sloppy_user_matrices(dq, device=dev(), n=12)
    check_fused_sweep_with_sloppy_user_matrices(dq, device=dev(), n=16)


def test_user_matrices_with_foreign_strides_and_pending_conjugation():
    """A user matrix may be column-major (torch.linalg.qr returns such) and its .mH -- the backward of the per-gate
    path, an inverted UAnyGate -- is then "contiguous" with the conjugation still pending: the kernels take raw pointers,
    so it has to be resolved first (it was not: silently wrong gradients in 'per_gate' mode)."""
    n = 7
    g = torch.Generator().manual_seed(3)
    a = torch.randn(4, 4, generator=g, dtype=torch.float64) + 1j * torch.randn(4, 4, generator=g, dtype=torch.float64)
    q = torch.linalg.qr(a)[0].to(torch.complex64)
    u_col = q.t().contiguous().t()                 # same values, column-major strides
    assert not u_col.is_contiguous() and torch.equal(u_col, q)
    res = {}
    for name, u in (('row-major', q.contiguous()), ('column-major', u_col)):
        for mode in ('per_gate', 'adjoint'):
            dq.executor.CONFIG['grad_mode'] = mode
            try:
                torch.manual_seed(2)
                cir = dq.QubitCircuit(n)
                cir.hlayer(); cir.rylayer()
                cir.any(u, wires=[1, 4], controls=[6])
                cir.any(torch.tensor([[0.6, 0.8], [-0.8, 0.6]], dtype=torch.complex64), wires=[2])
                cir.add(dq.UAnyGate(u, nqubit=n, wires=[0, 3]).inverse())
                cir.rxlayer()
                cir.observable(0); cir.observable([1, 2], 'xz')
                cir.to(dev())
                cir()
                cir.expectation().sum().backward()
                res[name, mode] = torch.stack([p.grad.cpu().reshape(-1)[0] for p in cir.parameters()])
            finally:
                dq.executor.CONFIG['grad_mode'] = 'adjoint'
    ref = res['row-major', 'adjoint']
    for key, v in res.items():
        assert (v - ref).abs().max().item() < 2e-5, (key, (v - ref).abs().max())
    with pytest.raises(ValueError, match='pending conjugation'):
        from deepquantum_amd import backend
        backend._ptr(u_col.mH)


@pytest.mark.gpu
def test_circuit_inside_a_module_follows_dtype_and_device_on_gpu():
    from _helpers import check_module_dtype_and_device

    check_module_dtype_and_device(dq, device=dev())


@pytest.mark.gpu
def test_get_amplitude_matches_reference_on_gpu():
    from _helpers import check_get_amplitude

    check_get_amplitude(dq, device=dev())


# ---- second order (Hessians were silently wrong in the default grad mode) ------------------------------
@pytest.mark.parametrize('prec', ['c64', 'c128'])
@pytest.mark.parametrize('case', specs.HESSIAN_CASES)
def test_hessian_of_the_reference_benchmark_circuit_on_gpu(case, prec):
    """``torch.autograd.functional.hessian`` over the reference's hessian_dq circuit
    (examples/benchmarks/gradient_benchmark.py:147-163) in the DEFAULT grad mode, through the HIP kernels, against the
    real reference's Hessians: 1e-4 (complex64) / 1e-10 (complex128)."""
    from _helpers import check_hessian_benchmark_against_golden

    assert dq.executor.CONFIG['grad_mode'] == 'adjoint'
    n, layer = case
    before = dq.executor.GRAPH_BACKWARDS['count']
    for tag in ('ones', 'rand'):
        err = check_hessian_benchmark_against_golden(dq, n, layer, prec, tag, device=dev())
        print(f'hessian {n}-{layer} {prec} {tag}: max error {err:.2e}')
    assert dq.executor.GRAPH_BACKWARDS['count'] > before


@pytest.mark.parametrize('mode', ['adjoint', 'per_gate'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_hessian_with_respect_to_parameters_on_gpu(mode, prec):
    from _helpers import check_hessian_benchmark_against_golden, check_hessian_params_against_golden

    old = dq.executor.CONFIG['grad_mode']
    dq.executor.CONFIG['grad_mode'] = mode
    try:
        check_hessian_params_against_golden(dq, prec, device=dev())
        check_hessian_benchmark_against_golden(dq, 4, 4, prec, 'rand', device=dev())
    finally:
        dq.executor.CONFIG['grad_mode'] = old


def test_gradgradcheck_of_the_autograd_nodes_on_gpu():
    """Numerical first and second derivatives of the nodes themselves on the kernels (complex128)."""
    from torch.autograd import gradcheck, gradgradcheck

    from deepquantum_amd import ops

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, dtype=torch.complex128, generator=g).to(dev()).requires_grad_(True)
    y = torch.randn(2, 32, dtype=torch.complex128, generator=g).to(dev()).requires_grad_(True)
    m1 = torch.randn(1, 2, 2, dtype=torch.complex128, generator=g).to(dev()).requires_grad_(True)
    m2 = torch.randn(2, 4, 4, dtype=torch.complex128, generator=g).to(dev()).requires_grad_(True)
    w = torch.randn(2, 3, dtype=torch.float64, generator=g).to(dev()).requires_grad_(True)
    zs = (0b00101, 0b01000, 0b10011)
    for controls in ((), (0,), (0, 3)):
        assert gradgradcheck(lambda a, b: ops.apply_gate(a, b, (2,), controls), (x, m1))
        assert gradgradcheck(lambda a, b: ops.apply_gate(a, b, (4, 1), controls), (x, m2))
        assert gradcheck(lambda a, b: ops.gate_grad(a, b, (2,), controls), (x, y))
        assert gradgradcheck(lambda a, b: ops.gate_grad(a, b, (4, 1), controls), (x, y))
    assert gradgradcheck(lambda a: ops.expect_pauli(a, 0b00110, 0b00011), (x,))
    assert gradgradcheck(lambda a: ops.marginal(a, (3, 0)), (x,))
    assert gradgradcheck(lambda a: ops.expect_z_multi(a, zs), (x,))
    assert gradcheck(lambda a, b: ops.scale_z_signs(a, zs, b), (x, w))
    assert gradgradcheck(lambda a, b: ops.scale_z_signs(a, zs, b), (x, w))


def test_hessian_of_a_density_matrix_circuit_with_channels_on_gpu():
    from test_hessian_cpu import check_noisy_hessian

    check_noisy_hessian(dq, device=dev())


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_hessian_by_the_tangent_circuit_on_gpu(prec):
    """executor._SweepGrads on the kernels: Hessian rows as sweeps of the tangent circuit against the per-gate replay."""
    from test_hessian_cpu import check_hessian_by_the_tangent_circuit

    check_hessian_by_the_tangent_circuit(dq, prec, device=dev())


@pytest.mark.gpu
def test_third_order_through_the_sweep_node_on_gpu():
    from test_hessian_cpu import check_third_order_through_the_sweep_node

    check_third_order_through_the_sweep_node(dq, device=dev())


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_hessian_vector_products_of_random_circuits_on_gpu(seed):
    """Second order on the kernels over the whole gate menu: the tangent circuit against the per-gate replay, states below
    and above a tile, both precisions."""
    from _helpers import check_hvp_random

    n = (4, 7, 10, 12, 13, 14)[seed]
    check_hvp_random(dq, device=dev(), n=n, batch=1 + seed % 3, seed=seed, ngates=30 + 5 * seed)
    check_hvp_random(dq, device=dev(), n=n, batch=1 + seed % 3, seed=seed, ngates=30 + 5 * seed, tol=3e-4, dtype=torch.float32)


@pytest.mark.gpu
def test_torch_func_transforms_over_a_circuit_on_gpu():
    from test_api_cpu import check_torch_func_transforms

    check_torch_func_transforms(dq, device=dev())


@pytest.mark.gpu
@pytest.mark.parametrize('n', [13, 20])
def test_fused_node_under_torch_func_transforms_on_gpu(n):
    """``torch.vmap(circuit)`` -- the reference's batching (circuit.py:232-240) -- ``grad``, ``jacrev`` and ``vmap(grad)``
    around a circuit keep their fused passes on the GPU (executor._FusedCircuit: LAST_RUN['passes'] > 0 is asserted
    inside), at a size where a launch per gate is an order of magnitude off (n = 20)."""
    from test_api_cpu import check_fused_node_under_transforms

    check_fused_node_under_transforms(dq, device=dev(), n=n)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [10, 11, 12, 13, 14, 15])
def test_torch_func_transforms_over_random_circuits_on_gpu(seed):
    """Fuzz of the fused node under torch.func transforms over the gate menu (controls of every arity, two-target encoded
    gates), below and above a tile, both precisions."""
    from _helpers import check_transforms_random

    n = (5, 9, 12, 14, 16, 18)[seed - 10]
    check_transforms_random(dq, device=dev(), n=n, seed=seed, ngates=30 + 6 * (seed - 10))
    check_transforms_random(dq, device=dev(), n=n, seed=seed, ngates=30 + 6 * (seed - 10), dtype=torch.float64, tol=1e-10)
