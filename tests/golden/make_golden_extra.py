"""Further golden fixtures made by running the REAL reference here (same mechanism as make_golden.py, kept in
a separate file so the large n = 24 pin need not be regenerated): Reset cases (batched encoders included).

usage: python tests/golden/make_golden_extra.py
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import specs  # noqa: E402
from make_golden import import_reference, to_np  # noqa: E402


def main():
    dq = import_reference()
    out = {}
    for name, spec in specs.RESET_CASES.items():
        for prec in ('c64',):   # Reset.to(torch.double) raises in the reference (operation.py:166), like Barrier
            cir = specs.build(dq, 5, spec)
            cir.observable(0)
            cir.observable([1, 2], 'xz')
            state = cir()
            out[f'{name}/{prec}/state'] = to_np(state.reshape(-1))
            out[f'{name}/{prec}/expectation'] = to_np(cir.expectation())
    # the Reset gate on given inputs (batch of 2, 4 qubits, tensor representation in and out)
    g = torch.Generator().manual_seed(77)
    for i, (wires, ps, kind) in enumerate(specs.RESET_GATE_CASES):
        psi = torch.randn(2, 16, generator=g) + 1j * torch.randn(2, 16, generator=g)
        if kind != 'random':       # zero the branch with wire 2 = 0 (bit_set) or = 1 (bit_clear)
            keep = 1 if kind == 'bit_set' else 0
            idx = torch.arange(16)
            psi[:, ((idx >> 1) & 1) != keep] = 0
        psi = (psi / psi.norm(dim=-1, keepdim=True)).to(torch.cfloat)
        gate = dq.gate.Reset(nqubit=4, wires=wires, postselect=ps, tsr_mode=True)
        out[f'resetgate/{i}/in'] = to_np(psi)
        out[f'resetgate/{i}/out'] = to_np(gate(psi.reshape([2] + [2] * 4)).reshape(2, -1))
    # batched data through a reset, gradient w.r.t. the data
    cir = dq.QubitCircuit(4)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.ry(1, encode=True)
    cir.cnot(0, 2)
    cir.cnot(1, 3)
    cir.reset([2], postselect=0)
    cir.crx(0, 2, encode=True)
    cir.reset([3], postselect=1)
    cir.observable(0)
    cir.observable([1, 2], 'zx')
    data = torch.tensor([[0.3, 1.2, 0.5], [2.0, -0.7, 1.1], [0.0, 0.4, 3.0]], requires_grad=True)
    state = cir(data=data)
    ev = cir.expectation()
    ev.sum().backward()
    out['reset_batched/data'] = to_np(data)
    out['reset_batched/state'] = to_np(state.reshape(3, -1))
    out['reset_batched/expectation'] = to_np(ev)
    out['reset_batched/grad'] = to_np(data.grad)
    # ansatz library: final states (and the parameters of the QCNN, which are drawn at random)
    for name, builder in specs.ANSATZ_CASES.items():
        cir = builder(dq)
        out[f'ansatz/{name}/state'] = to_np(cir().reshape(-1))
        out[f'ansatz/{name}/ngate'] = np.array(len(cir.operators))
    torch.manual_seed(11)
    qcnn = dq.QuantumConvolutionalNeuralNetwork(8, 2)
    for i, prm in enumerate(qcnn.parameters()):
        out[f'ansatz/qcnn/param{i}'] = to_np(prm)
    out['ansatz/qcnn/nparam'] = np.array(len(list(qcnn.parameters())))
    out['ansatz/qcnn/state'] = to_np(qcnn().reshape(-1))
    # read-out functions: conditional gates, post-selection, amplitudes / probabilities, custom
    # initial states given as amplitude vectors (single and batched)
    def readout_circuit(lib, init_state='zeros'):
        cir = lib.QubitCircuit(4, init_state=init_state)
        cir.h(0)
        cir.ry(1, encode=True)
        cir.cnot(0, 2)
        cir.x(3, controls=[0], condition=True)        # measurement-conditioned X on wire 3
        cir.rz(2, controls=[1], condition=True, encode=True)
        cir.h(2)
        return cir

    cir = readout_circuit(dq)
    data = torch.tensor([[0.7, 0.3], [2.1, -0.9], [1.3, 1.7]])
    state = cir(data)
    out['readout/data'] = to_np(data)
    out['readout/state'] = to_np(state)
    out['readout/wires_condition'] = np.array(sorted(cir.wires_condition))
    for bits in ('00', '01', '10', '11'):
        out[f'readout/post_select_{bits}'] = to_np(cir.post_select(bits))
    out['readout/amp_0110'] = to_np(cir.get_amplitude('0110'))
    out['readout/prob_1011'] = to_np(cir.get_prob('1011'))
    cir1 = readout_circuit(dq)
    st1 = cir1(data[1])
    out['readout/single_state'] = to_np(st1)
    out['readout/single_post_select_10'] = to_np(cir1.post_select('10'))
    out['readout/single_prob_wires'] = to_np(cir1.get_prob('01', wires=[1, 3]))
    out['readout/single_amp_1001'] = to_np(cir1.get_amplitude('1001'))
    res = cir1.measure(shots=20000, with_prob=True)
    keys = sorted(res)
    out['readout/measure_keys'] = np.array([int(k, 2) for k in keys])
    out['readout/measure_probs'] = np.array([float(res[k][1]) for k in keys])
    res2 = cir1.measure(shots=20000, with_prob=True, wires=[0, 2])
    keys2 = sorted(res2)
    out['readout/measure02_keys'] = np.array([int(k, 2) for k in keys2])
    out['readout/measure02_probs'] = np.array([float(res2[k][1]) for k in keys2])
    g = torch.Generator().manual_seed(12)
    vec = torch.randn(16, generator=g) + 1j * torch.randn(16, generator=g)
    batch = torch.randn(2, 16, generator=g) + 1j * torch.randn(2, 16, generator=g)
    out['readout/init_vec'] = to_np(vec)
    out['readout/init_batch'] = to_np(batch)
    out['readout/state_from_vec'] = to_np(readout_circuit(dq, init_state=vec)(data[0]))
    cirb = readout_circuit(dq)
    out['readout/state_from_batch'] = to_np(cirb(data[:2], state=dq.QubitState(4, batch).state))
    out['readout/amplitude_encoding'] = to_np(dq.amplitude_encoding(torch.arange(1.0, 11.0), 4))

    cir = specs.extra_gates_circuit(dq)
    out['extra_gates/state'] = to_np(cir().reshape(-1))
    out['extra_gates/expectation'] = to_np(cir.expectation())
    out['extra_gates/unitary'] = to_np(cir.get_unitary())

    # density matrices and channels
    for name, c in specs.DM_CASES.items():
        cir = dq.QubitCircuit(c['nqubit'], init_state=c['init'], den_mat=True)
        for method, args, kwargs in c['spec']:
            getattr(cir, method)(*args, **kwargs)
        for wires, basis in c['observables']:
            cir.observable(wires, basis)
        rho = cir()
        out[f'dm/{name}/rho'] = to_np(rho)
        out[f'dm/{name}/expectation'] = to_np(cir.expectation())
    # batched data + a trainable channel, gradients w.r.t. the data and the channel parameter
    torch.manual_seed(5)
    cir = dq.QubitCircuit(3, den_mat=True)
    cir.hlayer()
    cir.rx(0, encode=True)
    cir.bit_flip(0, encode=True)
    cir.cnot(0, 1)
    cir.amp_damp(1)                      # trainable
    cir.ry(2, encode=True)
    cir.depolarizing(2, 0.3)
    cir.crz(1, 2, encode=True)
    cir.observable(0)
    cir.observable([1, 2], 'zx')
    data = torch.tensor([[0.3, 0.5, 1.2, -0.4], [1.0, 0.2, 0.1, 0.9]], requires_grad=True)
    rho = cir(data=data)
    ev = cir.expectation()
    ev.sum().backward()
    prm = [p for p in cir.parameters()]
    out['dm/batched/data'] = to_np(data)
    out['dm/batched/theta'] = to_np(prm[0])
    out['dm/batched/rho'] = to_np(rho)
    out['dm/batched/expectation'] = to_np(ev)
    out['dm/batched/data_grad'] = to_np(data.grad)
    out['dm/batched/theta_grad'] = to_np(prm[0].grad)
    # partial trace and a user-supplied density matrix as the initial state
    g = torch.Generator().manual_seed(3)
    a = torch.randn(8, 8, generator=g) + 1j * torch.randn(8, 8, generator=g)
    rho0 = a @ a.mH
    rho0 = (rho0 / rho0.diagonal().sum()).to(torch.cfloat)
    out['dm/user/rho0'] = to_np(rho0)
    out['dm/user/ptrace_02'] = to_np(dq.qmath.partial_trace(rho0, 3, [0, 2]))
    out['dm/user/ptrace_1'] = to_np(dq.qmath.partial_trace(rho0, 3, [1]))
    cir = dq.QubitCircuit(3, init_state=rho0, den_mat=True)
    cir.h(0)
    cir.cnot(0, 2)
    cir.phase_damp(1, 0.6)
    out['dm/user/rho'] = to_np(cir())
    np.savez_compressed(os.path.join(HERE, 'golden_extra.npz'), **out)
    print('wrote golden_extra.npz', os.path.getsize(os.path.join(HERE, 'golden_extra.npz')), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
    main()
