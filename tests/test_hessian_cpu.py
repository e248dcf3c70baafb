"""Second-order autograd against the real reference's Hessians (tests/golden/golden_hessian.npz), host logic on the CPU
test double -- which, like the HIP entry points, returns tensors WITHOUT an autograd graph (``backend._NoGraph``), so a
backward that is not built from differentiable nodes shows up here as a wrong Hessian, as it would on the GPU.
tests/test_circuit_gpu.py repeats the checks through the kernels."""

import pytest
import torch

import deepquantum_amd as dq
from _helpers import check_hessian_benchmark_against_golden, check_hessian_params_against_golden


@pytest.mark.parametrize('mode', ['adjoint', 'per_gate'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
@pytest.mark.parametrize('case', [(4, 2, 'ones'), (4, 4, 'rand'), (6, 2, 'rand')])
def test_hessian_of_the_reference_benchmark_circuit(cpu_backend, mode, prec, case):
    old = dq.executor.CONFIG['grad_mode']
    dq.executor.CONFIG['grad_mode'] = mode
    try:
        n, layer, tag = case
        before = dq.executor.GRAPH_BACKWARDS['count']
        check_hessian_benchmark_against_golden(dq, n, layer, prec, tag)
        if mode == 'adjoint':
            assert dq.executor.GRAPH_BACKWARDS['count'] > before, 'create_graph=True must take the differentiable route'
    finally:
        dq.executor.CONFIG['grad_mode'] = old


@pytest.mark.parametrize('mode', ['adjoint', 'per_gate'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_hessian_with_respect_to_parameters(cpu_backend, mode, prec):
    old = dq.executor.CONFIG['grad_mode']
    dq.executor.CONFIG['grad_mode'] = mode
    try:
        check_hessian_params_against_golden(dq, prec)
    finally:
        dq.executor.CONFIG['grad_mode'] = old


def test_first_order_backward_still_takes_the_sweep(cpu_backend):
    cir = dq.QubitCircuit(4)
    cir.hlayer()
    cir.rxlayer()
    cir.cnot_ring()
    cir.observable(0)
    cir()
    cir.expectation().sum().backward()
    assert not dq.executor.LAST_SWEEP['with_graph']


@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_one_sum_per_x_rotation_only_when_nothing_differentiates_the_cotangent_again(cpu_backend, prec):
    """DQ_FG_GRAD variant 4 (ONE real sum for a unitary a I + i b X): the trace part of sum lambda (x) conj(psi) cancels against
    the tangent of a rotation, so a backward that records no graph leaves it out.  Under ``create_graph=True`` the VALUE of
    the matrix cotangent is differentiated again (d2M/dtheta2 is radial, not tangent): the two-sum variant stays, and the
    Hessian agrees with the per-gate nodes."""
    n = 12
    dt = torch.float64 if prec == 'c128' else torch.float32
    tol = 1e-11 if prec == 'c128' else 2e-5

    def build():
        torch.manual_seed(11)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rxlayer(encode=True)
        cir.cnot_ring()
        cir.rxlayer()
        cir.crx(0, 5)
        cir.rx(3, controls=[1, 7])
        cir.cnot_ring(reverse=True)
        cir.rxlayer(encode=True)
        cir.observable(0)
        cir.observable([2, 5], 'zx')
        if prec == 'c128':
            cir.to(torch.double)
        return cir

    data = torch.rand(n * 2, generator=torch.Generator().manual_seed(1), dtype=dt)
    out = {}
    for key, (mode, terminal_sums) in {'per_gate': ('per_gate', True), 'two': ('adjoint', False), 'one': ('adjoint', True)}.items():
        dq.executor.CONFIG['grad_mode'] = mode
        dq.executor.CONFIG['terminal_grad_sums'] = terminal_sums
        try:
            cir = build()
            x = data.clone().requires_grad_(True)
            cir(data=x)
            cir.expectation().sum().backward()
            if mode == 'adjoint':
                assert dq.executor.LAST_SWEEP['fused']
                assert dq.executor.LAST_SWEEP['variants'] == ((4,) if terminal_sums else (2,))
            out[key] = torch.cat([x.grad] + [p.grad.reshape(-1) for p in cir.parameters()])
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
            dq.executor.CONFIG['terminal_grad_sums'] = True
    assert (out['one'] - out['per_gate']).abs().max().item() < tol
    assert (out['two'] - out['one']).abs().max().item() < tol
    # with a graph: the full sums, and a Hessian-vector product equal to the per-gate nodes'
    hv = {}
    for mode in ('per_gate', 'adjoint'):
        dq.executor.CONFIG['grad_mode'] = mode
        try:
            cir = build()
            x = data.clone().requires_grad_(True)
            cir(data=x)
            (g,) = torch.autograd.grad(cir.expectation().sum(), x, create_graph=True)
            if mode == 'adjoint':
                assert 4 not in dq.executor.LAST_SWEEP.get('variants', ())
            (hv[mode],) = torch.autograd.grad((g * torch.linspace(-1, 1, g.numel(), dtype=dt)).sum(), x)
        finally:
            dq.executor.CONFIG['grad_mode'] = 'adjoint'
    assert (hv['adjoint'] - hv['per_gate']).abs().max().item() < 20 * tol


def test_the_test_double_builds_no_graph(cpu_backend):
    """The double is no more capable than the kernels: raw backend calls return graph-less tensors."""
    x = torch.randn(1, 16, dtype=torch.complex128, requires_grad=True)
    gy = torch.randn(1, 16, dtype=torch.complex128, requires_grad=True)
    m = torch.eye(2, dtype=torch.complex128).unsqueeze(0).requires_grad_(True)
    assert not dq.backend.gate_grad(x, gy, [1], [0]).requires_grad
    assert not dq.backend.apply_gate(x, m, [1], []).requires_grad
    assert not dq.backend.expect_pauli(x, 1, 2).requires_grad
    assert not dq.backend.scale_z_signs(x, [3], torch.ones(1, 1, dtype=torch.float64, requires_grad=True)).requires_grad


@pytest.mark.parametrize('controls', [(), (0,), (0, 3)])
def test_gradgradcheck_of_the_gate_nodes(cpu_backend, controls):
    """torch.autograd.gradcheck / gradgradcheck on the autograd nodes themselves (complex128, numerical Jacobians)."""
    from torch.autograd import gradcheck, gradgradcheck

    from deepquantum_amd import ops

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, dtype=torch.complex128, generator=g, requires_grad=True)
    m1 = torch.randn(1, 2, 2, dtype=torch.complex128, generator=g, requires_grad=True)
    m2 = torch.randn(2, 4, 4, dtype=torch.complex128, generator=g, requires_grad=True)
    assert gradcheck(lambda a, b: ops.apply_gate(a, b, (2,), controls), (x, m1))
    assert gradgradcheck(lambda a, b: ops.apply_gate(a, b, (2,), controls), (x, m1))
    assert gradgradcheck(lambda a, b: ops.apply_gate(a, b, (4, 1), controls), (x, m2))
    y = torch.randn(2, 32, dtype=torch.complex128, generator=g, requires_grad=True)
    assert gradcheck(lambda a, b: ops.gate_grad(a, b, (2,), controls), (x, y))
    assert gradgradcheck(lambda a, b: ops.gate_grad(a, b, (4, 1), controls), (x, y))


def test_gradgradcheck_of_the_reductions(cpu_backend):
    from torch.autograd import gradcheck, gradgradcheck

    from deepquantum_amd import ops

    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 16, dtype=torch.complex128, generator=g, requires_grad=True)
    assert gradcheck(lambda a: ops.expect_pauli(a, 0b0110, 0b0011), (x,))
    assert gradgradcheck(lambda a: ops.expect_pauli(a, 0b0110, 0b0011), (x,))
    assert gradgradcheck(lambda a: ops.marginal(a, (3, 0)), (x,))
    assert gradcheck(lambda a: ops.expect_z_multi(a, (0b0101, 0b1000, 0b0011)), (x,))
    assert gradgradcheck(lambda a: ops.expect_z_multi(a, (0b0101, 0b1000, 0b0011)), (x,))
    w = torch.randn(2, 3, dtype=torch.float64, generator=g, requires_grad=True)
    assert gradcheck(lambda a, b: ops.scale_z_signs(a, (0b0101, 0b1000, 0b0011), b), (x, w))
    assert gradgradcheck(lambda a, b: ops.scale_z_signs(a, (0b0101, 0b1000, 0b0011), b), (x, w))


def _noisy_loss(dq, params, device=None):
    """A density-matrix circuit with encoders, a trainable-strength channel and a Reset-free read-out: everything on the
    per-gate nodes (channels are not reversible)."""
    cir = dq.QubitCircuit(3, den_mat=True)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.cnot(0, 1)
    cir.ry(1, encode=True)
    cir.rzz([1, 2], encode=True)
    cir.amp_damp(1, 0.3)
    cir.depolarizing(0, 0.1)
    cir.rx(2, encode=True)
    cir.observable(0)
    cir.observable([1, 2], 'xz')
    cir.to(torch.double)
    if device is not None:
        cir.to(device)
    cir(data=params)
    ev = cir.expectation().reshape(-1)
    return ev[0] + 0.5 * ev[1] ** 2


def check_noisy_hessian(dq, device=None):
    from torch.autograd.functional import hessian

    x = torch.tensor([0.3, -0.8, 1.1, 0.5], dtype=torch.float64, device=device)
    h = hessian(lambda p: _noisy_loss(dq, p, device), x)
    eps = 1e-4
    num = torch.zeros(4, 4, dtype=torch.float64)
    for i in range(4):
        for sgn in (1, -1):
            xi = x.clone()
            xi[i] += sgn * eps
            xi.requires_grad_(True)
            (g,) = torch.autograd.grad(_noisy_loss(dq, xi, device), xi)
            num[i] += sgn * g.detach().cpu() / (2 * eps)
    assert (h.cpu() - num).abs().max().item() < 1e-6, (h, num)
    assert (h - h.T).abs().max().item() < 1e-10


def test_hessian_of_a_density_matrix_circuit_with_channels(cpu_backend):
    """Second derivatives through the non-reversible path (channels keep per-gate nodes): against central differences of
    the first-order gradient."""
    check_noisy_hessian(dq)


def _hessian_both_routes(build, x, tol):
    res = {}
    for mode in ('tangent', 'replay'):
        dq.executor.CONFIG['second_order'] = mode
        try:
            before, rows = dq.executor.GRAPH_BACKWARDS['count'], dq.executor.GRAPH_BACKWARDS['tangent_rows']
            res[mode] = torch.autograd.functional.hessian(build, x)
            assert dq.executor.GRAPH_BACKWARDS['count'] > before
            assert (dq.executor.GRAPH_BACKWARDS['tangent_rows'] - rows >= x.numel()) == (mode == 'tangent')
        finally:
            dq.executor.CONFIG['second_order'] = 'tangent'
    err = (res['tangent'] - res['replay']).abs().max().item()
    assert err < tol, err
    assert res['replay'].abs().max().item() > 1e-3
    return res['tangent']


def check_hessian_by_the_tangent_circuit(dq, prec, device=None):
    """``executor._SweepGrads``: a row of a Hessian as one forward and one reverse sweep of the tangent circuit (blocks
    [[U, 0], [C, U]] on one more qubit) against the per-gate formulation -- general, diagonal and controlled trainable
    gates on one and two targets, fixed gates of every kind in between, data and ``nn.Parameter``s."""
    dt = torch.float64 if prec == 'c128' else torch.float32
    n = 5

    def circuit():
        torch.manual_seed(3)
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        cir.rxlayer(encode=True)
        cir.cnot_ring()
        cir.u3(1, encode=True)                 # general 2x2
        cir.rz(2, encode=True)                 # diagonal
        cir.p(0, encode=True)
        cir.crx(0, 3, encode=True)             # with a control
        cir.toffoli(0, 1, 4)
        cir.rzz([1, 3], encode=True)           # diagonal on two targets
        cir.rxx([0, 2], encode=True)           # dense on two targets
        cir.s(2)
        cir.swap([1, 4])
        cir.ry(4, encode=True)
        cir.ry(3, controls=[0, 1], encode=True)
        cir.rylayer()                          # nn.Parameters (constants here)
        cir.observable(0)
        cir.observable([1, 2], 'xy')
        cir = cir.to(torch.double) if prec == 'c128' else cir
        return cir if device is None else cir.to(device)

    def f(p):
        cir = circuit()
        cir(data=p)
        return (cir.expectation() * torch.tensor([1.0, -0.5], dtype=dt, device=device)).sum()

    x = (torch.rand(circuit().ndata, dtype=dt, generator=torch.Generator().manual_seed(5)) * 3.0).to(device)
    h = _hessian_both_routes(f, x, 1e-10 if prec == 'c128' else 2e-5)
    assert (h - h.T).abs().max().item() < (1e-10 if prec == 'c128' else 2e-5)

    # a batch of data, every angle a function of two numbers
    cir = circuit()
    data = (torch.rand(2, cir.ndata, dtype=dt, generator=torch.Generator().manual_seed(6)) * 3.0).to(device)

    def loss_of(theta):
        cir2 = circuit()
        # the trainable layer's angles come from ``theta`` through the data path of a second circuit with the same gates
        cir2(data=data * theta.sum())
        return cir2.expectation().sum()

    theta = torch.tensor([0.7, -0.3], dtype=dt, device=device)
    _hessian_both_routes(loss_of, theta, 1e-9 if prec == 'c128' else 1e-3)


@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_hessian_by_the_tangent_circuit_equals_the_per_gate_replay(cpu_backend, prec):
    check_hessian_by_the_tangent_circuit(dq, prec)


def check_third_order_through_the_sweep_node(dq, device=None):
    """Under create_graph=True the sweep node's own backward differentiates the per-gate formulation: third derivatives."""
    def f(t):
        cir = dq.QubitCircuit(3)
        cir.hlayer()
        cir.rx(0, encode=True)
        cir.cnot(0, 1)
        cir.ry(1, encode=True)
        cir.cnot(1, 2)
        cir.rx(2, encode=True)
        cir.rz(0, encode=True)
        cir.observable([0, 2], 'zx')
        cir.to(torch.double)
        if device is not None:
            cir.to(device)
        cir(data=t.expand(4) * torch.tensor([1.0, 2.0, 0.5, 1.5], dtype=torch.double, device=device))
        return cir.expectation().sum()

    t = torch.tensor([0.4], dtype=torch.double, device=device, requires_grad=True)
    d1, = torch.autograd.grad(f(t), t, create_graph=True)
    d2, = torch.autograd.grad(d1.sum(), t, create_graph=True)
    d3, = torch.autograd.grad(d2.sum(), t)
    eps = 1e-4

    def second(v):
        tv = torch.tensor([v], dtype=torch.double, device=device, requires_grad=True)
        a, = torch.autograd.grad(f(tv), tv, create_graph=True)
        b, = torch.autograd.grad(a.sum(), tv)
        return b.item()

    num = (second(0.4 + eps) - second(0.4 - eps)) / (2 * eps)
    assert abs(d3.item() - num) < 1e-6, (d3.item(), num)


def test_third_order_through_the_sweep_node(cpu_backend):
    check_third_order_through_the_sweep_node(dq)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_hessian_vector_products_of_random_circuits_by_both_routes(cpu_backend, seed):
    from _helpers import check_hvp_random

    n = 4 + seed
    check_hvp_random(dq, n=n, batch=1 + seed % 3, seed=seed, ngates=30)
    check_hvp_random(dq, n=n, batch=1 + seed % 3, seed=seed, ngates=30, tol=2e-4, dtype=torch.float32)


def test_second_order_with_an_initial_state_that_requires_grad(cpu_backend):
    """The cotangent of the sweep node's first output (d loss / d initial state) seeds the alpha half of the tangent
    circuit: Hessian-vector products with respect to the state's real and imaginary parts, the data and the parameters, by
    both routes."""
    n = 4
    res = {}
    for mode in ('tangent', 'replay'):
        dq.executor.CONFIG['second_order'] = mode
        try:
            torch.manual_seed(1)
            cir = dq.QubitCircuit(n)
            cir.hlayer()
            cir.rxlayer(encode=True)
            cir.cnot_ring()
            cir.u3(1, encode=True)
            cir.crx(0, 2, encode=True)
            cir.rzz([1, 3], encode=True)
            cir.rylayer()
            cir.observable(0)
            cir.observable([1, 2], 'xy')
            cir.to(torch.double)
            g = torch.Generator().manual_seed(2)
            a = torch.randn(2, 2**n, generator=g, dtype=torch.double).requires_grad_()
            b = torch.randn(2, 2**n, generator=g, dtype=torch.double).requires_grad_()
            data = (torch.rand(2, cir.ndata, generator=g, dtype=torch.double) * 3).requires_grad_()
            psi = torch.complex(a, b)
            psi = psi / psi.norm(dim=-1, keepdim=True)
            cir(data=data, state=psi.unsqueeze(-1))
            loss = (cir.expectation() * torch.tensor([1.0, -0.5], dtype=torch.double)).sum()
            leaves = [a, b, data] + list(cir.parameters())
            first = torch.autograd.grad(loss, leaves, create_graph=True)
            vs = [torch.randn(f.shape, generator=g, dtype=torch.double) for f in first]
            rows = dq.executor.GRAPH_BACKWARDS['tangent_rows']
            res[mode] = torch.autograd.grad(sum((f * v).sum() for f, v in zip(first, vs, strict=True)), leaves)
            assert (dq.executor.GRAPH_BACKWARDS['tangent_rows'] > rows) == (mode == 'tangent')
        finally:
            dq.executor.CONFIG['second_order'] = 'tangent'
    for x, y in zip(res['tangent'], res['replay'], strict=True):
        assert (x - y).abs().max().item() < 1e-12
    assert res['replay'][0].abs().max().item() > 1e-2


def test_second_order_with_an_input_state_the_node_did_not_keep(cpu_backend):
    """A circuit node pins its INPUT state only where that is free (shared, small, differentiated:
    executor._keep_input); the second-order routes of its backward otherwise recompute it from the output with the
    exact inverses (executor._input_of): same Hessian either way."""
    from deepquantum_amd import executor

    old = dict(executor.CONFIG)
    executor.CONFIG['permute_min_bits'] = 12
    try:
        n = 13
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        for i in range(n):
            cir.rx(i, encode=True)
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        for i in range(4):
            cir.ry(i, encode=True)
        cir.observable(0)
        cir.observable([1, 3], 'zz')
        cir.to(torch.double)
        g = torch.Generator().manual_seed(3)
        st = torch.randn(2, 1 << n, 1, dtype=torch.complex128, generator=g)
        st = st / st.norm(dim=1, keepdim=True)
        x = torch.rand(cir.ndata, dtype=torch.float64, generator=g)

        def f(p):
            cir(data=p, state=st)
            return cir.expectation().sum()

        res = {}
        for kb in (64 << 20, 0):
            executor.CONFIG['keep_input_bytes'] = kb
            res[kb] = torch.autograd.functional.hessian(f, x)
        assert res[0].abs().max().item() > 1e-3
        assert (res[64 << 20] - res[0]).abs().max().item() < 1e-12
    finally:
        executor.CONFIG.update(old)
