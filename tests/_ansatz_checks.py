"""Checks of the ansatz library shared by the CPU (test double) and GPU (HIP) test files: amplitude parity
with the reference's final states and the known answers of the reference's tests/test_ansatz.py."""

import math
from fractions import Fraction

import torch

from _helpers import gold_extra, specs


def _run(cir, device):
    if device is not None:
        cir.to(device)
    with torch.no_grad():
        return cir()


def check_states(dq, device=None, names=None, tol=2e-5):
    for name, builder in specs.ANSATZ_CASES.items():
        if names is not None and name not in names:
            continue
        cir = builder(dq)
        assert len(cir.operators) == int(gold_extra(f'ansatz/{name}/ngate')), name
        st = _run(cir, device).reshape(-1).cpu()
        err = (st - gold_extra(f'ansatz/{name}/state')).abs().max().item()
        assert err < tol, f'{name}: amplitude error {err}'


def check_qcnn(dq, device=None):
    torch.manual_seed(0)
    qcnn = dq.QuantumConvolutionalNeuralNetwork(8, 2)
    params = list(qcnn.parameters())
    assert len(params) == int(gold_extra('ansatz/qcnn/nparam'))
    with torch.no_grad():
        for i, prm in enumerate(params):
            prm.copy_(gold_extra(f'ansatz/qcnn/param{i}'))
    for op in qcnn.operators:       # the copies of a shared gate cache their matrix
        if hasattr(op, '_invalidate'):
            op._invalidate()
    st = _run(qcnn, device).reshape(-1).cpu()
    assert (st - gold_extra('ansatz/qcnn/state')).abs().max().item() < 2e-5
    # the parameters are shared inside a layer: far fewer parameters than gates, and all of them get gradients
    qcnn.observable(0)
    qcnn()
    qcnn.expectation().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in qcnn.parameters())


def _argmax(res):
    return max(res, key=res.get)


def check_known_answers(dq, device=None, shor_ncount=8):
    # reference tests/test_ansatz.py:7-15
    t, phase = 3, 1 / 8
    qpe = dq.QuantumPhaseEstimationSingleQubit(t, phase)
    _run(qpe, device)
    assert int(_argmax(qpe.measure(wires=list(range(t)))), 2) / 2**t == phase
    # :18-30
    enc, qft, add = dq.NumberEncoder(5, 1), dq.QuantumFourierTransform(5, reverse=True), dq.PhiAdder(5, 8)
    cir = enc + qft + add + qft.inverse()
    _run(cir, device)
    assert int(_argmax(cir.measure()), 2) == 9
    # :33-49
    n1, n2, mod = 5, 1, 8
    nq = len(bin(mod))
    mm = [0, nq - 2]
    qft = dq.QuantumFourierTransform(nq, mm, reverse=True)
    cir = dq.NumberEncoder(nq, n1, mm) + qft + dq.PhiModularAdder(nq, n2, mod, mm, [nq - 1]) + qft.inverse()
    _run(cir, device)
    assert int(_argmax(cir.measure(wires=list(range(mm[0], mm[1] + 1)))), 2) == (n1 + n2) % mod
    # :52-71
    n1, n2, n3, mod = 1, 2, 14, 15
    nx, nb = len(bin(n3)) - 2, len(bin(mod)) - 1
    nq = nx + nb + 1
    cir = (dq.NumberEncoder(nq, n3, [0, nx - 1]) + dq.NumberEncoder(nq, n1, [nx, nq - 2])
           + dq.ControlledMultiplier(nq, n2, mod, [0, nq - 2], nx, [nq - 1]))
    _run(cir, device)
    assert int(_argmax(cir.measure(wires=list(range(nx, nq - 1)))), 2) == (n1 + n2 * n3) % mod
    # :74-93
    mod, a, x = 15, 8, 3
    nreg = len(bin(mod)) - 2
    nq = 2 * nreg + 2
    anc = list(range(nreg, 2 * nreg + 2))
    cir = dq.NumberEncoder(nq, x, [0, nreg - 1]) + dq.ControlledUa(nq, a, mod, [0, nreg - 1], anc)
    _run(cir, device)
    assert int(_argmax(cir.measure(wires=list(range(nreg)))), 2) == (a * x) % mod
    assert int(_argmax(cir.measure(wires=anc)), 2) == 0
    # :96-151  Shor: the order of 7 mod 15 is 4, so the measured phases are the multiples of 1/4 -- exactly
    # for the hand-made circuit; the general circuit feeds unreduced multiples 2^k a to the modular adders
    # (as the reference does, ansatz.py:132-133) and leaks ~25 % onto other outcomes, the four peaks remain
    for cls, args, exact in ((dq.ShorCircuitFor15, (shor_ncount, 7), True), (dq.ShorCircuit, (15, shor_ncount, 7), False)):
        cir = cls(*args)
        _run(cir, device)
        res = cir.measure(wires=list(range(shor_ncount)), shots=2000)
        peaks = sorted(res, key=res.get, reverse=True)[:4]
        assert sorted(int(k, 2) / 2**shor_ncount for k in peaks) == [0, 0.25, 0.5, 0.75], (cls.__name__, res)
        if exact:
            assert len(res) == 4, res
        factors = set()
        for key in peaks:
            ph = int(key, 2) / 2**shor_ncount
            r = Fraction(ph).limit_denominator(15).denominator
            if ph != 0:
                for g in (math.gcd(7 ** (r // 2) - 1, 15), math.gcd(7 ** (r // 2) + 1, 15)):
                    if g not in (1, 15):
                        factors.add(g)
        assert factors == {3, 5}, (cls.__name__, res)
