"""Host-side API (gate library, layers, circuit driver, batching, autograd glue, dtype plumbing) against the
golden vectors of the real reference, with the kernels replaced by the oracle-backed CPU test double.
These tests validate everything above the C ABI without a GPU; tests/test_circuit_gpu.py repeats them
through the HIP kernels."""

import pytest
import torch

import deepquantum_amd as dq
from _helpers import (CDTYPE, TOL, check_circuit_against_golden, check_get_amplitude, check_module_dtype_and_device, gold,
                      specs)


@pytest.mark.parametrize('name', ['readme', 'zoo5', 'rand4', 'rand8', 'rand12', 'batched10'])
@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_circuits_match_reference(cpu_backend, name, prec):
    check_circuit_against_golden(dq, name, prec)


@pytest.mark.parametrize('prec', ['c64', 'c128'])
def test_fused_path_used_for_large_states(cpu_backend, prec):
    check_circuit_against_golden(dq, 'rand14_seed1234', prec)
    assert cpu_backend.fused_calls > 0 and dq.executor.LAST_RUN['passes'] > 0
    assert dq.executor.LAST_RUN['passes'] < dq.executor.LAST_RUN['gates'] / 8


def test_gate_classes_match_reference(cpu_backend):
    for i, case in enumerate(specs.GATE_CASES):
        cls = getattr(dq, case['cls'])
        gate = cls(nqubit=case['nqubit'], **case['kwargs'])
        if case.get('inverse'):
            gate = gate.inverse()
        gate = gate.to(torch.double)
        m = gate.update_matrix()
        assert (m - gold(f'gate/{i}/matrix')).abs().max().item() == 0.0, (i, case['cls'])  # bit-identical matrices
        with torch.no_grad():
            out = gate(gold(f'gate/{i}/in'))
        ref = gold(f'gate/{i}/out')
        assert out.shape == ref.shape
        assert (out - ref).abs().max().item() < 1e-12, (i, case['cls'])


def _grad_check(builder, tag, nqubit):
    cir = builder(dq, nqubit)
    params = gold(f'grad/{tag}/params')
    off = 0
    with torch.no_grad():
        for p in cir.parameters():
            p.copy_(params[off : off + p.numel()].reshape(p.shape))
            off += p.numel()
    assert off == params.numel()
    data = gold(f'grad/{tag}/data').clone().requires_grad_(True)
    cir(data)
    ev = cir.expectation()
    assert (ev - gold(f'grad/{tag}/expectation')).abs().max().item() < 1e-5
    ev.sum().backward()
    assert (data.grad - gold(f'grad/{tag}/data_grad')).abs().max().item() < 1e-4
    if params.numel():
        pg = torch.cat([p.grad.reshape(-1) for p in cir.parameters()])
        assert (pg - gold(f'grad/{tag}/param_grads')).abs().max().item() < 1e-4


def test_autograd_mixed_circuit(cpu_backend):
    _grad_check(specs.grad_circuit_a, 'mixed', 4)


def test_autograd_qaoa(cpu_backend):
    _grad_check(specs.qaoa_circuit, 'qaoa', 7)


def test_forward_shapes_follow_reference(cpu_backend):
    cir = dq.QubitCircuit(3)
    cir.h(0)
    cir.rx(1, encode=True)
    cir.cnot(0, 2)
    cir.observable(0)
    cir.observable(1, 'x')
    assert cir(torch.tensor([0.3])).shape == (8, 1)
    assert cir.expectation().shape == (2,)
    assert cir(torch.tensor([[0.3], [0.4]])).shape == (2, 8, 1)
    assert cir.expectation().shape == (2, 2)
    batch_state = torch.zeros(2, 8, 1, dtype=torch.cfloat)
    batch_state[:, 0] = 1
    assert cir(torch.tensor([0.3]), state=batch_state).shape == (2, 8, 1)
    assert cir(torch.tensor([[0.3], [0.4]]), state=batch_state).shape == (2, 8, 1)
    # after a batched call the encoders hold the last sample again (reference: circuit.py:241)
    assert cir.encoders[0].theta.numel() == 1


def test_gate_forward_representations(cpu_backend):
    g = dq.Hadamard(nqubit=2, wires=[1])
    x = torch.tensor([1, 0, 0, 0], dtype=torch.cfloat)
    assert g(x).shape == (4, 1)
    assert g(x.reshape(1, 4, 1).repeat(3, 1, 1)).shape == (3, 4, 1)
    g.tsr_mode = True
    assert g(x.reshape(1, 2, 2)).shape == (1, 2, 2)


def test_to_double_promotes_complex_buffers(cpu_backend):
    cir = dq.QubitCircuit(2)
    cir.h(0)
    cir.rx(1, 0.3)
    cir.cnot(0, 1)
    cir.to(torch.double)
    assert cir.init_state.state.dtype == torch.complex128
    assert cir.operators[0].matrix.dtype == torch.complex128
    assert cir.operators[1].theta.dtype == torch.float64
    assert cir().dtype == torch.complex128
    # H keeps its float32-rounded entries (parity with the reference, SURVEY summary item 3)
    assert cir.operators[0].matrix[0, 0].real.item() == 0.7071067690849304


def test_inverse_circuit_returns_initial_state(cpu_backend):
    c = specs.CIRCUITS['rand8']
    cir = specs.build(dq, 8, c['spec'] + [('u3', [3, [0.3, 0.2, 0.1]], {}), ('rzz', [[1, 6], 0.4], {}), ('s', [2], {}), ('t', [5], {})])
    cir.to(torch.double)
    full = cir + cir.inverse()
    out = full()
    ref = torch.zeros(256, 1, dtype=torch.complex128)
    ref[0] = 1
    assert (out - ref).abs().max().item() < 1e-5  # float32-rounded constants are not exactly unitary


def test_measure_and_amplitudes(cpu_backend):
    cir = dq.QubitCircuit(3)
    cir.h(0)
    cir.cnot(0, 1)
    cir.cnot(1, 2)
    cir()
    res = cir.measure(shots=200)
    assert set(res) <= {'000', '111'} and sum(res.values()) == 200
    assert abs(cir.get_prob('111').item() - 0.5) < 1e-6
    assert abs(cir.get_prob('1', wires=[2]).item() - 0.5) < 1e-6
    assert abs(cir.get_amplitude('000').abs().item() - 0.5**0.5) < 1e-6
    res = cir.measure(shots=50, wires=[0, 2], with_prob=True)
    assert set(res) <= {'00', '11'}


def test_expectation_with_shots(cpu_backend):
    cir = dq.QubitCircuit(2)
    cir.x(0)
    cir.observable(0)
    cir.observable(1, 'z')
    cir()
    ev = cir.expectation(shots=100)
    assert torch.allclose(ev, torch.tensor([-1.0, 1.0]))


def test_cpu_tensors_without_test_backend_fail_loudly():
    cir = dq.QubitCircuit(2)
    cir.h(0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        cir()


def test_out_of_scope_paths_raise():
    with pytest.raises(NotImplementedError):
        dq.QubitCircuit(2, mps=True)
    with pytest.raises(AssertionError):
        dq.QubitCircuit(2).bit_flip(0)          # channels need den_mat=True, as in the reference
    with pytest.raises(NotImplementedError):
        dq.QubitCircuit(2).pattern()            # MBQC transpilation: out of scope


def test_reupload_encoding(cpu_backend):
    cir = dq.QubitCircuit(2, reupload=True)
    for _ in range(3):
        cir.rx(0, encode=True)
        cir.ry(1, encode=True)
    data = torch.tensor([0.3, 0.5, 0.7, 0.9])
    cir(data)
    got = [g.theta.item() for g in cir.encoders]
    assert got == pytest.approx([0.3, 0.5, 0.7, 0.9, 0.3, 0.5])


def test_torch_vmap_over_the_circuit_matches_native_batching(cpu_backend):
    """The reference batches with torch.vmap over _forward_helper (circuit.py:232-240); the per-gate op has a
    vmap rule, so the same call works here and equals the native batched path."""
    cir = dq.QubitCircuit(4)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.cnot(0, 2)
    cir.ry(3, encode=True)
    cir.crx(1, 2, encode=True)
    cir.rzz([0, 3], encode=True)
    g = torch.Generator().manual_seed(0)
    data = torch.rand(5, 4, generator=g)
    native = cir(data)
    vm = torch.vmap(cir._forward_helper, in_dims=(0, None))(data, cir.init_state.state)
    assert vm.shape == native.shape
    assert (vm - native).abs().max().item() < 1e-6
    cir.encode(data[-1])
    # mapped initial states too
    states = torch.stack([cir.init_state.state] * 5)
    vm2 = torch.vmap(cir._forward_helper)(data, states)
    assert (vm2 - native).abs().max().item() < 1e-6


def test_reset_matches_reference(cpu_backend):
    from _helpers import check_reset_against_golden

    check_reset_against_golden(dq)


def test_reset_sampled_is_a_valid_collapse(cpu_backend):
    """postselect=None samples the outcome: whatever it is, the reset wires end in |0>, the state is
    normalised, and the remaining wires carry one of the two conditional states."""
    torch.manual_seed(3)
    cir = dq.QubitCircuit(3)
    cir.h(0)
    cir.cnot(0, 1)
    cir.ry(2, 0.4)
    cir.reset([1], postselect=None)
    seen = set()
    for _ in range(20):
        st = cir().reshape(2, 2, 2)
        assert abs((st.abs() ** 2).sum().item() - 1) < 1e-5
        assert st[:, 1, :].abs().max().item() == 0
        seen.add(int(st[1, 0, :].abs().max().item() > 0.5))     # wire 0 collapsed with wire 1
    assert seen == {0, 1}


def test_ansatz_states_match_reference(cpu_backend):
    from _ansatz_checks import check_qcnn, check_states

    check_states(dq)
    check_qcnn(dq)


def test_ansatz_known_answers(cpu_backend):
    from _ansatz_checks import check_known_answers

    check_known_answers(dq, shor_ncount=3)      # 18-qubit Shor (ncount = 8) runs in the GPU suite


def test_density_matrix_path_matches_reference(cpu_backend):
    from _helpers import check_density_matrix_against_golden

    check_density_matrix_against_golden(dq)


def test_adjoint_grad_mode_matches_per_gate_autograd(cpu_backend):
    from _helpers import check_adjoint_grad_mode

    check_adjoint_grad_mode(dq, dtype=torch.float64, tol=1e-10)
    check_adjoint_grad_mode(dq, dtype=torch.float32, tol=2e-5)


def test_edge_cases(cpu_backend):
    from _helpers import check_edge_cases

    check_edge_cases(dq)


def test_many_z_observables_in_one_pass(cpu_backend):
    from _helpers import check_many_z_observables

    check_many_z_observables(dq, dtype=torch.float64)
    check_many_z_observables(dq, dtype=torch.float32)


def test_readout_functions_match_reference(cpu_backend):
    from _helpers import check_readout_against_golden

    check_readout_against_golden(dq)


def test_remaining_gate_classes_match_reference(cpu_backend):
    from _helpers import check_extra_gates_against_golden

    check_extra_gates_against_golden(dq)


def test_fixed_angle_matrices_are_reused_and_invalidated(cpu_backend):
    """A gate with a fixed angle evaluates its matrix once; an in-place change of the angle, a new angle, the
    inverse flag and a parameter that needs a graph all force a fresh evaluation."""
    g = dq.Rx(0.3, nqubit=2, wires=0)
    g.update_matrix()
    m0 = g.update_matrix()
    assert g.update_matrix() is m0                       # reused
    with torch.no_grad():
        g.theta.mul_(2.0)                                # in-place edit bumps the version counter
    m1 = g.update_matrix()
    assert m1 is not m0 and torch.allclose(m1, g.get_matrix(torch.tensor(0.6)))
    g.init_para(0.9)
    assert torch.allclose(g.update_matrix(), g.get_matrix(torch.tensor(0.9)))
    g.inv_mode = True
    assert torch.allclose(g.update_matrix(), g.get_matrix(torch.tensor(-0.9)))
    t = dq.U3Gate([0.1, 0.2, 0.3], nqubit=1, requires_grad=True)
    a = t.update_matrix()
    assert a.requires_grad and t.update_matrix() is not a  # trainable: a graph per evaluation
    with torch.no_grad():
        b = t.update_matrix()
        assert not b.requires_grad and t.update_matrix() is b and torch.equal(a.detach(), b)
    # a circuit of fixed gates gives the same state on every call, also after an angle edit
    cir = dq.QubitCircuit(3)
    cir.rxlayer(inputs=[0.1, 0.2, 0.3]); cir.cnot_ring(); cir.rylayer(inputs=[0.4, 0.5, 0.6])
    s0 = cir().clone()
    assert torch.equal(cir(), s0)
    with torch.no_grad():
        cir.operators[1].theta.add_(0.5)
    ref = dq.QubitCircuit(3)
    ref.rxlayer(inputs=[0.1, 0.7, 0.3]); ref.cnot_ring(); ref.rylayer(inputs=[0.4, 0.5, 0.6])
    assert torch.allclose(cir(), ref(), atol=1e-6)


def test_random_circuits_under_every_scheduler_configuration(cpu_backend):
    from _helpers import check_fuzz_against_oracle
    check_fuzz_against_oracle(dq, n=13, seeds=(0, 1), depth=5)
    check_fuzz_against_oracle(dq, n=12, seeds=(2,), depth=5, double=True)
    check_fuzz_against_oracle(dq, n=15, seeds=(3,), depth=8, batch=1)     # several passes: permuted stores matter


def test_combined_gate_inverse_leaves_the_original_alone():
    """inverse() builds a new gate (reference gate.py:1883-1900); the original's factors and matrix stay."""
    g = dq.gate.CombinedSingleGate([dq.gate.Rx(inputs=0.3), dq.gate.Hadamard(), dq.gate.Rz(inputs=0.7)], nqubit=2, wires=[1])
    before = g.update_matrix().clone()
    factors = list(g.gates)
    inv = g.inverse()
    assert list(g.gates) == factors and inv.gates is not g.gates
    assert (g.update_matrix() - before).abs().max().item() == 0.0
    assert (inv.update_matrix() @ before - torch.eye(2)).abs().max().item() < 1e-6
    cir = dq.QubitCircuit(2)
    cir.add(g)
    cir.inverse()
    assert (g.update_matrix() - before).abs().max().item() == 0.0


def test_uany_kernel_kind_follows_the_matrix():
    """A diagonal user matrix takes the diagonal kernels only while it IS diagonal (state_dict reload, in-place writes)."""
    u = dq.gate.UAnyGate(torch.diag(torch.tensor([1, 1j], dtype=torch.cfloat)), nqubit=2, wires=[0])
    assert u._kernel_kind == 'diag' and u.prims()[0].kind == 'diag'
    x = dq.gate.UAnyGate(torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat), nqubit=2, wires=[0])
    u.load_state_dict(x.state_dict())
    assert u._kernel_kind == 'gen' and u.prims()[0].kind == 'gen'
    u.matrix = torch.diag(torch.tensor([1, -1], dtype=torch.cfloat))
    assert u._kernel_kind == 'diag'


def test_fused_reverse_sweep_matches_per_gate_autograd(cpu_backend):
    from _helpers import check_fused_sweep

    check_fused_sweep(dq, n=11, batch=2)
    check_fused_sweep(dq, n=10, batch=2, tol=1e-10, dtype=torch.float64)      # complex128: wave-tile geometry


def test_fused_reverse_sweep_of_states_smaller_than_a_tile(cpu_backend):
    """The (psi, lambda) pair of a state below a tile is zero-padded to one and swept fused (reference sweep:
    adjoint.py:42-83; executor.CONFIG['small_fused_sweep'])."""
    from _helpers import check_fused_sweep, check_fused_sweep_random

    check_fused_sweep(dq, n=8, batch=2)
    check_fused_sweep(dq, n=9, batch=1, tol=1e-10, dtype=torch.float64)
    for seed, n in enumerate((3, 4, 6, 9)):
        check_fused_sweep_random(dq, n=n, batch=1 + seed % 2, seed=seed, ngates=40)
        check_fused_sweep_random(dq, n=n, batch=1 + seed % 2, seed=seed, ngates=40, tol=1e-10, dtype=torch.float64)
    dq.executor.CONFIG['small_fused_sweep'] = False
    try:
        check_fused_sweep_random(dq, n=6, batch=2, seed=9, ngates=40, expect_fused=False)
    finally:
        dq.executor.CONFIG['small_fused_sweep'] = True


@pytest.mark.parametrize('seed', [0, 1])
def test_fused_reverse_sweep_on_random_circuits(cpu_backend, seed):
    from _helpers import check_fused_sweep_random

    check_fused_sweep_random(dq, n=11, batch=2, seed=seed, ngates=60)


def test_z_expectations_come_out_of_the_last_pass(cpu_backend):
    """No-grad forwards take <Z..Z> of the Z-type observables from the registers of the last pass (DQ_FG_EXPZ): same values
    as the separate reduction, for one sample and a batch, both precisions; other observables still read the state."""
    from deepquantum_amd import executor

    for n, dt, tol in ((12, torch.float32, 3e-6), (13, torch.float64, 1e-12)):
        for batch in (None, 3):
            def build():
                torch.manual_seed(1)
                c = dq.QubitCircuit(n)
                c.hlayer(); c.rxlayer(encode=True); c.cnot_ring(); c.rylayer(); c.rz(0, encode=True); c.hlayer()
                c.observable(0); c.observable([1, n - 1], 'zz'); c.observable([2, 3], 'xy'); c.observable(list(range(n)), 'z' * n)
                if dt == torch.float64:
                    c.to(torch.double)
                return c
            data = torch.rand((batch, n + 1) if batch else (n + 1,), dtype=dt)
            res = {}
            for fused in (True, False):
                executor.CONFIG['fused_expectation'] = fused
                try:
                    c = build()
                    with torch.no_grad():
                        c(data)
                        assert (c._expz is not None) == fused
                        res[fused] = c.expectation()
                        if fused:           # the values belong to the state they were taken from
                            c.state = c.state.clone()
                            assert (c.expectation() - res[True]).abs().max().item() < tol
                finally:
                    executor.CONFIG['fused_expectation'] = True
            assert res[True].shape == res[False].shape and (res[True] - res[False]).abs().max().item() < tol
    c = dq.QubitCircuit(12)
    c.hlayer(); c.rxlayer(); c.observable(0)
    c()                                     # under grad mode: the differentiable reduction
    assert c._expz is None and c.expectation().requires_grad


def test_fused_reverse_sweep_with_user_matrices_that_are_unitary_to_1e_4_only(cpu_backend):
    from _helpers import check_fused_sweep_with_sloppy_user_matrices

    check_fused_sweep_with_sloppy_user_matrices(dq, n=12)


def test_circuit_inside_a_module_follows_dtype_and_device(cpu_backend):
    check_module_dtype_and_device(dq)


def test_get_amplitude_matches_reference(cpu_backend):
    check_get_amplitude(dq)


def test_steady_state_cache_follows_parameter_updates(cpu_backend):
    """executor._steady: merged runs / plan / matrix buffer are reused only while the forward sees the same primitive
    objects at the same versions -- an in-place parameter update, a reloaded state_dict, an appended gate and a changed
    batch all give what a cold run gives."""
    from deepquantum_amd import executor

    def build(angles):
        cir = dq.QubitCircuit(12)
        cir.hlayer()
        for q in range(12):
            cir.rx(q, inputs=angles[q])
            cir.rz(q, inputs=angles[q] / 2)
        cir.cnot_ring()
        for q in range(12):
            cir.ry(q, inputs=angles[(q + 3) % 12])
        cir.observable(0)
        cir.observable([1, 2], 'zz')
        return cir

    def run(cir, cold=False):
        keep = dict(executor.CONFIG)
        executor.CONFIG.update({'merge_min_amps': 0, 'steady_cache': not cold})
        try:
            with torch.no_grad():
                return cir().clone(), cir.expectation().clone()
        finally:
            executor.CONFIG.clear()
            executor.CONFIG.update(keep)

    a0 = [0.1 * (q + 1) for q in range(12)]
    cir = build(a0)
    s1, e1 = run(cir)
    s2, e2 = run(cir)                                           # the warm run
    assert torch.equal(s1, s2) and torch.equal(e1, e2)
    assert any(e.get('merged') is not None for e in executor._STEADY.values())
    sc, ec = run(cir, cold=True)
    assert torch.equal(s1, sc) and torch.equal(e1, ec)
    # an optimiser-style in-place update of one angle
    gate = [op for op in cir.operators if hasattr(op, 'theta')][5]
    with torch.no_grad():
        gate.theta.add_(0.37)
    s3, e3 = run(cir)
    sc, ec = run(cir, cold=True)
    assert not torch.equal(s3, s1) and torch.equal(s3, sc) and torch.equal(e3, ec)
    # a reloaded state_dict
    other = build([0.05 * (q + 2) for q in range(12)])
    ref_s, ref_e = run(other, cold=True)
    cir.load_state_dict(other.state_dict())
    s4, e4 = run(cir)
    assert torch.equal(s4, ref_s) and torch.equal(e4, ref_e)
    # one more gate
    cir.x(3)
    other.x(3)
    s5, _ = run(cir)
    ref_s, _ = run(other, cold=True)
    assert torch.equal(s5, ref_s)


# ---- round-3 ADVICE items -----------------------------------------------------------------------------------------
def test_forward_under_inference_mode(cpu_backend):
    """Matrices computed under ``torch.inference_mode()`` track no version counter: the caches keyed on
    ``tensor._version`` step aside instead of raising (operation.py / gate.py)."""
    cir = dq.QubitCircuit(3)
    cir.h(0)
    cir.rx(1)
    cir.cnot(0, 2)
    cir.crx(0, 1)
    cir.rzz([1, 2])
    cir.observable(0)
    cir.observable([1, 2], 'xz')
    with torch.no_grad():
        ref = cir.expectation() if cir() is not None else None
    with torch.inference_mode():
        cir.prims() if hasattr(cir, 'prims') else None
        cir()
        got = cir.expectation()
        cir()                                   # (twice: a cache filled under inference mode must not be served)
        assert torch.equal(got, cir.expectation())
    assert torch.allclose(ref, got, atol=1e-6)
    with torch.inference_mode():                # a model BUILT under inference mode: its parameters track no version
        c2 = dq.QubitCircuit(3)
        c2.h(0)
        c2.rx(1)
        c2.ry(2, 0.3)
        c2.crx(0, 1, 0.2)
        c2.observable(0)
        for _ in range(2):
            c2()
            c2.expectation()
    # the steady-state cache (>= 16 primitives) under inference mode
    c3 = dq.QubitCircuit(4)
    for _ in range(5):
        c3.rxlayer()
        c3.cnot_ring()
    c3.observable(0)
    with torch.inference_mode():
        c3()
        a = c3.expectation()
        c3()
        assert torch.equal(a, c3.expectation())


def test_circuit_pickles_after_a_no_grad_forward(cpu_backend):
    """The cache of expectation values taken by the last pass holds a weak reference: it must not travel with
    ``pickle`` / ``torch.save`` / ``copy`` (circuit.py)."""
    import copy
    import io
    import pickle

    n = 12
    cir = specs.build(dq, n, specs.random_spec(n, 3, 5))
    cir.observable(0)
    cir.observable([1, 2], 'zz')
    with torch.no_grad():
        cir()
        ev = cir.expectation()
    cir._expz = cir._expz or {'state': None}    # (the double may not have filled it: the field must be dropped either way)
    import weakref
    cir._expz['state'] = weakref.ref(cir.state)
    clone = pickle.loads(pickle.dumps(cir))
    assert clone._expz is None
    buf = io.BytesIO()
    torch.save(cir, buf)
    assert copy.copy(cir)._expz is None and copy.deepcopy(cir)._expz is None
    with torch.no_grad():
        clone()
        assert torch.allclose(clone.expectation(), ev, atol=1e-6)


def test_sharded_state_keeps_a_shard_of_another_shape(cpu_backend):
    """Only the empty LAZY marker triggers the lazy build of a shard; a shard of an equivalent but different shape
    is left alone (state.py)."""
    st = dq.DistributedQubitState(4)
    assert st.amps.shape == (16,) and st.amps[0] == 1
    col = torch.arange(16, dtype=torch.float32).to(torch.cfloat).reshape(16, 1)
    st.amps = col
    assert st.amps.shape == (16, 1) and torch.equal(st.amps, col)
    old = dq.DistributedQubitState.LAZY_AMPS
    dq.DistributedQubitState.LAZY_AMPS = 4
    try:
        lazy = dq.DistributedQubitState(4)
        assert lazy._buffers['amps'].numel() == 0
        assert lazy.amps.shape == (16,) and lazy.amps[0] == 1 and lazy.buffer.shape == (16,)
    finally:
        dq.DistributedQubitState.LAZY_AMPS = old


def test_caches_do_not_keep_autograd_graphs_alive(cpu_backend):
    """A matrix that carries an autograd graph is a new object every forward; a cache entry for it can never hit and would
    only keep the graph -- and the parameters' AccumulateGrad nodes, with the stream they were made on -- alive until the
    next forward has already built its graph on the same nodes (a training step captured into a HIP graph after eager
    steps on the default stream then drags the default stream into the capture: tests/test_circuit_gpu.py)."""
    from deepquantum_amd import executor

    cir = dq.QubitCircuit(13)
    for q in range(13):
        cir.h(q)
        cir.rx(q)
        cir.cnot(q, (q + 1) % 13)
        cir.ry(q)
    cir.crx(0, 5)
    cir.observable(0)
    cir()
    loss = cir.expectation().sum()
    loss.backward()
    for g in cir.modules():
        c = g.__dict__.get('_prims_cache')
        assert c is None or not c[0].requires_grad, type(g).__name__
    for e in executor._STEADY.values():
        assert not any(p.matrix is not None and p.matrix.requires_grad for p in e['prims'])
    with torch.no_grad():                   # ... while fixed matrices are still served from the caches
        cir()
        a = [id(p) for p in cir.prims()]
        cir()
        assert a == [id(p) for p in cir.prims()]


@pytest.mark.parametrize('batch', [None, 3])
def test_gradient_with_respect_to_encoded_data_by_finite_differences(cpu_backend, batch):
    """``encode`` hands the columns of differentiated data to the gates from one unbind (one autograd node, one kernel in
    the backward) instead of a slice per layer and gate: gates of one and of three parameters, single encoder gates,
    with and without data re-uploading (reference: circuit.py:265-293), against central differences in float64."""
    for reupload, width in ((False, 4 * 3 + 4 + 1 + 3), (True, 5)):
        cir = dq.QubitCircuit(4, reupload=reupload)
        cir.hlayer()
        cir.u3layer(encode=True)
        cir.cnot_ring()
        cir.rxlayer(encode=True)
        cir.rzz([0, 2], encode=True)
        cir.u3(1, encode=True)
        cir.rylayer()
        cir.observable(0)
        cir.observable([1, 3], 'xz')
        cir.to(torch.double)
        g = torch.Generator().manual_seed(3)
        shape = (width,) if batch is None else (batch, width)
        data = torch.rand(shape, generator=g, dtype=torch.double) * 3.0

        def f(x):
            cir(data=x)
            return (cir.expectation() * torch.tensor([1.0, -0.5], dtype=torch.double)).sum()

        x = data.clone().requires_grad_(True)
        f(x).backward()
        num = torch.zeros_like(data)
        flat = data.reshape(-1)
        eps = 1e-6
        with torch.no_grad():
            for i in range(flat.numel()):
                up, dn = flat.clone(), flat.clone()
                up[i] += eps
                dn[i] -= eps
                num.reshape(-1)[i] = (f(up.reshape(shape)) - f(dn.reshape(shape))) / (2 * eps)
        assert (x.grad - num).abs().max().item() < 1e-7, (reupload, (x.grad - num).abs().max())


def check_torch_func_transforms(dq, device=None):
    """``torch.func`` transforms over a circuit -- the reference composes with them because it is made of tensor
    operations (qmath.py:485-506); here the gates of a call that sees functorch wrappers run as per-gate nodes
    (``setup_context`` style, ``vmap`` rules): grad, jacrev, reverse-over-reverse Hessians (all rows in ONE traversal),
    vmap(grad) over data rows -- against plain autograd; circuits kept and built inside the function."""
    import torch.func as tf

    n, layer = 4, 2

    def circuit():
        torch.manual_seed(2)
        cir = dq.QubitCircuit(n)
        for _ in range(layer):
            for i in range(n - 1):
                cir.cnot(i, i + 1)
            cir.rxlayer(encode=True)
            cir.rzlayer(encode=True)
            cir.u3(1, controls=[0], encode=True)        # (no nn.Parameter: .to() inside a transform may not touch one)
            cir.rxx([0, 2], encode=True)
        cir.observable(basis='x')
        cir.observable([1, 3], 'zy')
        cir.to(torch.double)
        return cir if device is None else cir.to(device)

    kept = circuit()

    def f_kept(p):
        kept(data=p)
        return kept.expectation().sum()

    def f_new(p):
        cir = circuit()
        cir(data=p)
        return cir.expectation().sum()

    x = torch.rand(kept.ndata, dtype=torch.float64, generator=torch.Generator().manual_seed(4)).to(device)
    jac = torch.autograd.functional.jacobian(f_kept, x)
    hes = torch.autograd.functional.hessian(f_kept, x)
    for f in (f_kept, f_new):
        assert (tf.grad(f)(x) - jac).abs().max().item() < 1e-10
        assert (tf.jacrev(f)(x) - jac).abs().max().item() < 1e-10
        assert (tf.jacrev(tf.jacrev(f))(x) - hes).abs().max().item() < 1e-9
    rows = torch.stack([x, 0.5 * x, x + 0.1])
    want = torch.stack([torch.autograd.functional.jacobian(f_kept, r) for r in rows])
    assert (tf.vmap(tf.grad(f_kept))(rows) - want).abs().max().item() < 1e-10
    # forward mode: every node has a jvp rule (the tangent of a gate application is two gate applications) -- jvp, jacfwd,
    # torch.func.hessian (forward over reverse), reverse over forward; Z-type strings (the multi-string reduction and its
    # cotangent node) as well as general Pauli strings
    v = torch.randn(x.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5)).to(device)
    assert abs(tf.jvp(f_kept, (x,), (v,))[1].item() - (jac * v).sum().item()) < 1e-10
    assert (tf.jacfwd(f_kept)(x) - jac).abs().max().item() < 1e-10
    assert (tf.hessian(f_kept)(x) - hes).abs().max().item() < 1e-9
    assert (tf.jacrev(tf.jacfwd(f_kept))(x) - hes).abs().max().item() < 1e-9
    zcir = circuit()
    zcir.observables = torch.nn.ModuleList()
    zcir.observable(0)
    zcir.observable([1, 2], 'zz')
    zcir.observable(3)

    def f_z(p):
        zcir(data=p)
        return (zcir.expectation() * torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=device)).sum()

    zjac = torch.autograd.functional.jacobian(f_z, x)
    zhes = torch.autograd.functional.hessian(f_z, x)
    assert (tf.jacfwd(f_z)(x) - zjac).abs().max().item() < 1e-10
    assert (tf.jacrev(f_z)(x) - zjac).abs().max().item() < 1e-10
    assert (tf.hessian(f_z)(x) - zhes).abs().max().item() < 1e-9
    assert (tf.jacrev(tf.jacrev(f_z))(x) - zhes).abs().max().item() < 1e-9
    # plain forward-mode AD (torch.autograd.forward_ad): a dual tensor is routed like a wrapper, its tangent survives
    import torch.autograd.forward_ad as fwad

    with fwad.dual_level():
        tangent = fwad.unpack_dual(f_z(fwad.make_dual(x, v))).tangent
    assert abs(tangent.item() - (zjac * v).sum().item()) < 1e-10
    # a JVP is linear in its tangent, in complex64 too: the rules that polarise a quadratic reduction (Z strings,
    # marginals) scale the tangent to the state's size first -- at scale 1 a tangent of 1e-6 came back per cent off
    zc64 = circuit().to(torch.float)
    zc64.observables = torch.nn.ModuleList()
    zc64.observable(0)
    zc64.observable([1, 2], 'zz')

    def f_z32(p):
        zc64(data=p)
        return zc64.expectation().sum()

    def f_m32(p):      # the differentiable marginal of a Reset-free circuit: through ops.marginal
        from deepquantum_amd import ops as dq_ops

        out = zc64(data=p)
        return dq_ops.marginal(out.reshape(1, -1), (0, 2))[0, 1]

    x32, v32 = x.float(), v.float()
    for f32 in (f_z32, f_m32):
        full = tf.jvp(f32, (x32,), (v32,))[1].item()
        for eps in (1e-3, 1e-6):
            small = tf.jvp(f32, (x32,), (eps * v32,))[1].item()
            assert abs(small / eps - full) < 2e-5 * max(1.0, abs(full)), (f32.__name__, eps, small / eps, full)
    # forward over forward through autograd.Function nodes is wrong in this PyTorch (any Function): refused by name
    with pytest.raises(RuntimeError, match='nested forward-mode'):
        tf.jacfwd(tf.jacfwd(f_kept))(x)
    # a batch of cotangents at the circuit node itself is refused by name, not by a cryptic batching-rule error
    with pytest.raises(RuntimeError, match='torch.func.jacrev'):
        torch.autograd.functional.jacobian(f_kept, x, vectorize=True)


def check_fused_node_under_transforms(dq, device=None, n=13):
    """``torch.vmap`` over the circuit (the reference's batching, circuit.py:232-240), ``grad`` / ``jacrev`` /
    ``vmap(grad)`` around it: ONE node with vmap rules (executor._FusedCircuit / _FusedSweep) -- fused passes and a fused
    reverse sweep, not a launch per gate -- against the native batch and plain autograd; a vmapped forward differentiated
    by plain autograd afterwards; second order still takes the per-gate nodes."""
    import torch.func as tf

    from deepquantum_amd import executor

    def circuit():
        cir = dq.QubitCircuit(n)
        cir.hlayer()
        for i in range(n):
            cir.rx(i, encode=True)
        for i in range(n - 1):
            cir.cnot(i, i + 1)
        for i in range(n):
            cir.ry(i, encode=True)
        cir.crx(1, 5, encode=True)
        cir.rzz([0, 3], encode=True)
        cir.observable(0)
        cir.observable([1, 3], 'zz')
        cir.observable(2, 'x')
        return cir if device is None else cir.to(device)

    cir = circuit()
    data = torch.rand(5, cir.ndata, generator=torch.Generator().manual_seed(0)).to(device)
    count = lambda: executor.LAST_RUN.get('fused_transform_nodes', 0)      # noqa: E731
    with torch.no_grad():
        native = cir(data).clone()
        c0 = count()
        vm = tf.vmap(cir._forward_helper, in_dims=(0, None))(data, cir.init_state.state)
    assert (vm.reshape(native.shape) - native).abs().max().item() < 1e-6
    assert count() == c0 + 1 and executor.LAST_RUN['passes'] > 0        # one node, fused passes

    def f(p):
        cir(data=p)
        return cir.expectation().sum()

    def fvec(p):
        cir(data=p)
        return cir.expectation().reshape(-1)

    x = data[0].clone()
    jac = torch.autograd.functional.jacobian(fvec, x)
    c0 = count()
    assert (tf.grad(f)(x) - jac.sum(0)).abs().max().item() < 1e-5
    assert count() == c0 + 1
    executor.LAST_SWEEP.update(fused=False, reductions=0)
    assert (tf.jacrev(fvec)(x) - jac).abs().max().item() < 1e-5
    assert count() == c0 + 2 and executor.LAST_SWEEP['fused'] and executor.LAST_SWEEP['reductions'] > 0
    rows = torch.stack([x, 0.5 * x, x + 0.1])
    want = torch.stack([torch.autograd.functional.jacobian(f, r) for r in rows])
    assert (tf.vmap(tf.grad(f))(rows) - want).abs().max().item() < 1e-5
    # (two vmap levels in the sweep: the data rows outside, the basis cotangents of jacrev inside)
    wantj = torch.stack([torch.autograd.functional.jacobian(fvec, r) for r in rows])
    assert (tf.vmap(tf.jacrev(fvec))(rows) - wantj).abs().max().item() < 1e-5
    # vmap over the forward, plain autograd afterwards (the reference's training step: vmap inside forward, backward outside)
    w = torch.arange(1 << n, dtype=torch.float32, device=device).reshape(1, -1, 1) / (1 << n)
    d2 = data.clone().requires_grad_(True)
    (tf.vmap(cir._forward_helper, in_dims=(0, None))(d2, cir.init_state.state).abs() ** 2 * w).sum().backward()
    d3 = data.clone().requires_grad_(True)
    (cir(d3).abs() ** 2 * w).sum().backward()
    assert (d2.grad - d3.grad).abs().max().item() < 1e-5
    # two reverse levels: still ONE node -- the second level runs the tangent circuit (executor._FusedSweep.backward), all
    # rows of the Hessian as samples of one forward and one sweep; forward mode (torch.func.hessian = jacfwd(jacrev)): the
    # per-gate nodes, as before
    hes = torch.autograd.functional.hessian(f, x)
    c0, r0 = count(), executor.GRAPH_BACKWARDS['tangent_rows']
    assert (tf.jacrev(tf.jacrev(f))(x) - hes).abs().max().item() < 1e-4
    assert count() == c0 + 1 and executor.GRAPH_BACKWARDS['tangent_rows'] == r0 + 1
    v = torch.randn(x.shape, generator=torch.Generator().manual_seed(9)).to(device)
    hv = tf.grad(lambda p: (tf.grad(f)(p) * v).sum())(x)             # a Hessian-vector product: grad of grad
    assert (hv - hes @ v).abs().max().item() < 1e-4
    # forward mode: the node's jvp rule is the tangent circuit's forward, the sweep node's jvp the same second-order routine
    # (executor._second_order) -- torch.func.jvp, jacfwd, hessian (= jacfwd(jacrev)), jacrev(jacfwd): one node each
    for name, fn in (('hessian', lambda: tf.hessian(f)(x)), ('jacrev(jacfwd)', lambda: tf.jacrev(tf.jacfwd(f))(x))):
        c0 = count()
        assert (fn() - hes).abs().max().item() < 1e-4, name
        assert count() == c0 + 1, name
    c0 = count()
    assert (tf.jacfwd(fvec)(x) - jac).abs().max().item() < 1e-4 and count() == c0 + 1
    assert (tf.jvp(fvec, (x,), (v,))[1] - jac @ v).abs().max().item() < 1e-4 and count() == c0 + 2
    import torch.autograd.forward_ad as fwad

    with fwad.dual_level():          # plain forward_ad (no torch.func.jvp around it): the per-gate nodes, as before
        c0 = count()
        tangent = fwad.unpack_dual(fvec(fwad.make_dual(x, v))).tangent
    assert (tangent - jac @ v).abs().max().item() < 1e-4 and count() == c0
    with pytest.raises(RuntimeError, match='nested forward-mode'):
        tf.jacfwd(tf.jacfwd(f))(x)
    # A/B switch
    executor.CONFIG['fused_transforms'] = False
    try:
        c0 = count()
        assert (tf.jacrev(fvec)(x) - jac).abs().max().item() < 1e-5 and count() == c0
    finally:
        executor.CONFIG['fused_transforms'] = True


def test_torch_func_transforms_over_a_circuit(cpu_backend):
    check_torch_func_transforms(dq)


def test_fused_node_under_torch_func_transforms(cpu_backend):
    from deepquantum_amd import executor

    old = executor.CONFIG['permute_min_bits']
    executor.CONFIG['permute_min_bits'] = 12
    try:
        check_fused_node_under_transforms(dq)
    finally:
        executor.CONFIG['permute_min_bits'] = old


def test_functorch_probes_are_guarded(monkeypatch):
    """The probes into functorch's interpreter stack (ops.transform_stack) are private API: checked on the releases
    ops.FUNCTORCH_PROBES_CHECKED_ON names; where they fail, callers get None and take the conservative route."""
    import torch.func as tf

    from deepquantum_amd import ops

    if any(torch.__version__.startswith(v) for v in ops.FUNCTORCH_PROBES_CHECKED_ON):
        seen = []

        def f(x):
            seen.append(ops.transform_stack())
            return (x * x).sum()

        tf.vmap(tf.grad(f))(torch.ones(2, 3))
        tf.jacfwd(f)(torch.ones(3))
        assert seen[0] == ['Vmap', 'Grad'] and 'Jvp' in seen[1] and ops.transform_stack() == []
    import torch._functorch.pyfunctorch as pf

    monkeypatch.delattr(pf, 'retrieve_all_functorch_interpreters')
    with pytest.warns(RuntimeWarning, match='retrieve_all_functorch_interpreters'):
        assert ops.transform_stack() is None
    ops._single_forward_level()          # (no refusal without the probe: it goes ahead)
    from deepquantum_amd import _functorch as fx

    assert fx.no_transforms() is False                      # ("unknown" is not "none": the conservative route)


def test_every_functorch_probe_goes_through_the_one_guarded_helper(monkeypatch):
    """VERDICT r5 (weak 10): no module but ``_functorch.py`` touches ``torch._C._functorch`` or functorch's interpreter
    stack, and with the private probes gone the helper still tells a wrapper from a plain tensor (public fallback) and says
    so in a warning instead of silently dropping the fused batching."""
    import glob
    import os
    import re

    import torch.func as tf

    import deepquantum_amd
    from deepquantum_amd import _functorch as fx

    pkg = os.path.dirname(deepquantum_amd.__file__)
    for path in glob.glob(os.path.join(pkg, '*.py')):
        if os.path.basename(path) == '_functorch.py':
            continue
        code = '\n'.join(ln.split('#', 1)[0] for ln in open(path).read().splitlines())
        assert not re.search(r'torch\._C\._functorch|retrieve_all_functorch_interpreters', code), path
    seen = []

    def f(x):
        seen.append((fx.is_wrapped_tensor(x), fx.is_batched(x), fx.is_legacy_batched(x)))
        return x.sum()

    tf.vmap(f)(torch.ones(2, 3))
    tf.grad(f)(torch.ones(3))
    assert seen == [(True, True, False), (True, False, False)]
    assert not fx.is_wrapped_tensor(torch.ones(2)) and fx.no_transforms()
    monkeypatch.setattr(fx, '_C', None)
    monkeypatch.setattr(fx, '_WARNED', set())
    seen.clear()
    with pytest.warns(RuntimeWarning, match='is_functorch_wrapped_tensor'):
        tf.vmap(f)(torch.ones(2, 3))
    assert seen == [(True, True, False)]                     # (data_ptr() refuses on a wrapper: the public fallback)
    assert not fx.is_wrapped_tensor(torch.ones(2))


def test_a_circuit_moves_its_small_buffers_in_bulk():
    """``cir.to(device)``: the tiny buffers of all gates travel as one copy per dtype (utils.BulkMove) -- same shapes, dtypes
    and values, every buffer a view of the moved block; dtype-only conversions and parameters go the ordinary way."""
    def build():
        torch.manual_seed(1)
        cir = dq.QubitCircuit(6)
        for _ in range(3):
            for i in range(5):
                cir.cnot(i, i + 1)
            cir.rxlayer(encode=True)
            cir.rzlayer()                       # trainable: nn.Parameters, never part of the bulk move
            cir.u3layer(encode=True)
        cir.observable(0)
        return cir

    cir = build()
    before = {k: (v.shape, v.dtype) for k, v in cir.state_dict().items()}
    cir.to('meta')
    after = {k: (v.shape, v.dtype, v.device.type) for k, v in cir.state_dict().items()}
    assert {k: v[:2] for k, v in after.items()} == before and all(v[2] == 'meta' for v in after.values())
    views = [op.matrix for op in cir.operators if type(op).__name__ == 'CNOT']
    assert all(m._base is not None for m in views)                         # slices of ONE moved block
    assert all(p._base is None for p in cir.parameters())
    cir2 = build().to(torch.double)                                          # (no device move: nothing batched)
    assert cir2.operators[0].matrix.dtype == torch.complex128 and cir2.operators[0].matrix._base is None
    ref = build()
    assert all(torch.equal(a, b) for a, b in zip(cir2.to(torch.float).state_dict().values(), ref.state_dict().values()))


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_torch_func_transforms_over_random_circuits(cpu_backend, seed):
    from _helpers import check_transforms_random
    from deepquantum_amd import executor

    old = executor.CONFIG['permute_min_bits']
    executor.CONFIG['permute_min_bits'] = 12
    try:
        n = (5, 8, 11, 13)[seed]
        check_transforms_random(dq, n=n, seed=seed, ngates=24 + 6 * seed)
        check_transforms_random(dq, n=n, seed=seed, ngates=24 + 6 * seed, dtype=torch.float64, tol=1e-10)
    finally:
        executor.CONFIG['permute_min_bits'] = old


def test_a_new_forward_lets_go_of_the_previous_autograd_graph_first(cpu_backend):
    """The encoders hold views of the caller's data; while one of the previous call's views lives, the leaf's AccumulateGrad
    node does too and the next graph reuses it -- created on whatever stream the FIRST step ran on.  After eager steps on the
    default stream, capturing the step in a HIP graph on a side stream then dies in hipStreamEndCapture
    (tests/test_circuit_gpu.py::test_captured_training_step_...).  Here, without a GPU: every step gets a fresh node."""
    cir = dq.QubitCircuit(4)
    for _ in range(2):
        cir.cnot_ring()
        cir.rxlayer(encode=True)
        cir.u3(1, encode=True)
        cir.rzlayer(encode=True)
    cir.observable(basis='x')
    params = torch.linspace(0.1, 2.9, cir.ndata).requires_grad_(True)

    def accumulator():
        return params.view_as(params).grad_fn.next_functions[0][0]

    for it in range(3):
        if params.grad is not None:
            params.grad.zero_()
        cir(data=params)
        cir.expectation().backward()
        node = accumulator()
        assert 'seen' not in node.metadata, f'step {it} reused the AccumulateGrad node of the step before'
        node.metadata['seen'] = True
        del node
    assert params.grad.abs().max().item() > 0


def test_structure_promises_of_two_wire_gates_and_channels_hold(cpu_backend):
    """DQ_MODE_XREAL / DQ_MODE_XCPLX are promises of a CLASS (the kernels never look at values): every gate and channel that
    makes one has a matrix that is non-zero only on the blocks (00, 11) and (01, 10) -- and real where it says so -- for any
    angle, inverted, batched, and as the row / column pair of a density matrix."""
    from deepquantum_amd import channel

    keep = torch.tensor([[(i ^ j) in (0, 3) for j in range(4)] for i in range(4)])
    g = torch.Generator().manual_seed(0)
    seen = 0
    for cls in (dq.Rxx, dq.Ryy, dq.Rxy, dq.ReconfigurableBeamSplitter):
        assert cls._kernel_mode2 in (4, 5)
        for gate in (cls(float(torch.rand(1, generator=g) * 6), nqubit=3, wires=[0, 2]),
                     cls(float(torch.rand(1, generator=g) * 6), nqubit=3, wires=[2, 1], controls=[0]).inverse()):
            for prims in (gate.prims(), gate.dm_prims()):
                for p in prims:
                    assert p.kind == 'gen' and p.mode == cls._kernel_mode2 and len(p.targets) == 2
                    m = p.matrix.reshape(-1, 4, 4)
                    assert float(m[:, ~keep].abs().max()) == 0.0, cls.__name__
                    assert cls._kernel_mode2 == 5 or float(m.imag.abs().max()) == 0.0, cls.__name__
                    seen += 1
    for cls, inp in ((channel.BitFlip, 0.3), (channel.Depolarizing, 0.4), (channel.AmplitudeDamping, 0.5),
                     (channel.Pauli, [0.3, 0.5, 0.7, 0.9]), (channel.GeneralizedAmplitudeDamping, [0.3, 0.6])):
        (p,) = cls(inp, nqubit=2, wires=[1]).dm_prims()
        m = p.matrix.reshape(4, 4)
        assert p.mode == 4 and float(m[~keep].abs().max()) == 0.0 and float(m.imag.abs().max()) == 0.0, cls.__name__
        seen += 1
    assert seen == 4 * 2 * 3 + 5
