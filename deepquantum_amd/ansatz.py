"""Ready-made circuits on top of the builder API: QFT, phase estimation, the
Draper/Beauregard arithmetic blocks (arXiv:quant-ph/0205095), Shor, HHL, a QCNN and a random Clifford+T
circuit.  Same constructors and gate sequences as the reference's ``ansatz.py`` (classes at :14-896), so the
final states agree with it amplitude by amplitude (tests/golden/make_golden_extra.py); everything here is
host-side circuit construction, the gates run on the fused HIP passes like any other circuit.

Organisation differs from the reference: the arithmetic blocks are written once as *emitters* -- functions
that append a block (or its inverse) to any circuit -- and the public classes are thin shells around them,
instead of nesting circuit objects and inverting them gate by gate.
"""

from __future__ import annotations

import math
import random
from typing import Any

import torch

from .circuit import QubitCircuit
from .gate import Rxx, Ryy, Rzz, U3Gate


def int_to_bitstring(x: int, n: int, debug: bool = False) -> str:
    """``x`` as ``n`` binary digits, MSB first; an overflowing ``x`` keeps its low ``n`` bits
    (reference: qmath.py:41-54)."""
    assert isinstance(x, int) and isinstance(n, int)
    if x >= 2**n and debug:
        print(f'Quantum register ({n}) overflowed for {x}.')
    return format(x % 2**n if x >= 2**n else x, f'0{n}b')


def is_unitary(matrix: torch.Tensor, rtol: float = 1e-5, atol: float = 1e-4) -> bool:
    """U U^dagger = 1 within tolerance (reference: qmath.py:97-114)."""
    if matrix.shape[-1] != matrix.shape[-2]:
        return False
    eye = torch.eye(matrix.shape[0], dtype=matrix.dtype, device=matrix.device)
    return torch.allclose(matrix @ matrix.mH, eye, rtol=rtol, atol=atol)


# ---- emitters -------------------------------------------------------------------------------------------
# A block is first written down as a list of elementary steps; the inverse of a block is the reversed list
# with every angle negated (H, X, CNOT, SWAP are involutions).  ``_replay`` appends the steps to a circuit.
def _replay(cir: QubitCircuit, steps: list[tuple], inverse: bool = False) -> None:
    for step in (reversed(steps) if inverse else steps):
        kind = step[0]
        if kind == 'h':
            cir.h(step[1])
        elif kind == 'x':
            cir.x(step[1], controls=step[2] or None)
        elif kind == 'cnot':
            cir.cnot(step[1], step[2])
        elif kind == 'swap':
            cir.swap([step[1], step[2]], controls=step[3] or None)
        elif kind == 'cp':
            cir.cp(step[1], step[2], -step[3] if inverse else step[3])
        elif kind == 'p':
            cir.p(wires=step[1], inputs=-step[2] if inverse else step[2], controls=step[3] or None)
        elif kind == 'barrier':
            cir.barrier(step[1])
        else:  # pragma: no cover
            raise KeyError(kind)


def _qft_steps(wires: list[int], swaps: bool, barrier: bool = False) -> list[tuple]:
    """Textbook QFT on ``wires`` (MSB first): H then controlled phases pi/2, pi/4, ... from the less
    significant wires; ``swaps`` appends the bit reversal (reference: ansatz.py:600-618)."""
    steps: list[tuple] = []
    m = len(wires)
    for i, w in enumerate(wires):
        steps.append(('h', w))
        for dist in range(1, m - i):
            steps.append(('cp', wires[i + dist], w, math.pi / 2**dist))
        if barrier:
            steps.append(('barrier', list(wires)))
    if swaps:
        steps.extend(('swap', wires[i], wires[m - 1 - i], []) for i in range(m // 2))
    return steps


def _phi_add_steps(wires: list[int], number: int, controls: list[int], debug: bool = False) -> list[tuple]:
    """Draper adder in Fourier space: wire i gets the phase 2 pi * number / 2^(m - i) restricted to the bits
    it can see (Fig. 2/3 of the paper; reference: ansatz.py:388-396)."""
    bits = int_to_bitstring(number, len(wires), debug=debug)
    steps = []
    for i, w in enumerate(wires):
        phi = 0
        for k, b in enumerate(bits[i:]):
            if b == '1':
                phi += math.pi / 2**k
        if phi != 0:
            steps.append(('p', w, phi, list(controls)))
    return steps


def _phi_mod_add_steps(wires: list[int], number: int, mod: int, anc: int, controls: list[int],
                       debug: bool = False) -> list[tuple]:
    """Beauregard's modular adder Phi(b) -> Phi(a + b mod N) (Fig. 5; reference: ansatz.py:452-488): add a,
    subtract N, copy the sign into the ancilla, add N back if it was negative, then uncompute the ancilla by
    comparing with a."""
    msb = wires[0]
    add_a = _phi_add_steps(wires, number, controls, debug)
    add_n = _phi_add_steps(wires, mod, [], debug)
    add_n_if_anc = _phi_add_steps(wires, mod, [anc], debug)
    qft = _qft_steps(wires, swaps=False)

    def inv(steps):
        return [_negated(s) for s in reversed(steps)]

    return (add_a + inv(add_n) + inv(qft) + [('cnot', msb, anc)] + qft + add_n_if_anc + inv(add_a) + inv(qft)
            + [('x', msb, []), ('cnot', msb, anc), ('x', msb, [])] + qft + add_a)


def _negated(step: tuple) -> tuple:
    if step[0] == 'cp':
        return (step[0], step[1], step[2], -step[3])
    if step[0] == 'p':
        return (step[0], step[1], -step[2], step[3])
    return step


def _cmult_steps(xwires: list[int], bwires: list[int], a: int, mod: int, anc: int, controls: list[int],
                 debug: bool = False, name: str = 'ControlledMultiplier') -> list[tuple]:
    """|x>|b> -> |x>|b + a x mod N> (Fig. 6; reference: ansatz.py:125-147): one modular addition of
    2^k a per bit of x, inside a QFT / inverse QFT of the b register.  The numbers 2^k a are used unreduced,
    as in the reference."""
    qft = _qft_steps(bwires, swaps=False)
    steps = list(qft)
    for k, xw in enumerate(reversed(xwires)):
        if debug and 2**k * a >= 2 * mod:
            print(f'The number 2^{k}*{a} in {name} may be too large, unless the control qubit {xw} is 0.')
        steps += _phi_mod_add_steps(bwires, 2**k * a, mod, anc, list(controls) + [xw], debug)
    steps += [_negated(s) for s in reversed(qft)]
    return steps


def _cua_steps(xwires: list[int], ancilla: list[int], a: int, mod: int, controls: list[int],
               debug: bool = False) -> list[tuple]:
    """|x> -> |a x mod N> with a clean ancilla register (Fig. 7; reference: ansatz.py:203-233): multiply into
    the ancilla, swap, un-multiply with the modular inverse."""
    bwires, last = ancilla[:-1], ancilla[-1]
    forward = _cmult_steps(xwires, bwires, a, mod, last, controls, debug)
    swaps = [('swap', xw, ancilla[i + 1], list(controls)) for i, xw in enumerate(xwires)]
    backward = _cmult_steps(xwires, bwires, pow(a, -1, mod), mod, last, controls, debug)
    return forward + swaps + [_negated(s) for s in reversed(backward)]


# ---- public classes -------------------------------------------------------------------------------------
class Ansatz(QubitCircuit):
    """Base class: a circuit that knows the wires it acts on, its ancillas and its controls
    (reference: ansatz.py:14-66)."""

    def __init__(self, nqubit: int, wires=None, minmax=None, ancilla=None, controls=None, init_state: Any = 'zeros',
                 name: str | None = None, den_mat: bool = False, reupload: bool = False, mps: bool = False,
                 chi: int | None = None) -> None:
        super().__init__(nqubit=nqubit, init_state=init_state, name=name, den_mat=den_mat, reupload=reupload,
                         mps=mps, chi=chi)
        if wires is None:
            if minmax is None:
                minmax = [0, nqubit - 1]
            self._check_minmax(minmax)
            wires = list(range(minmax[0], minmax[1] + 1))
        wires = self._convert_indices(wires)
        ancilla = self._convert_indices([] if ancilla is None else ancilla)
        controls = self._convert_indices([] if controls is None else controls)
        assert not (set(wires) & set(ancilla)) and not (set(wires) & set(controls)), 'Use repeated wires'
        self.wires = sorted(wires)
        self.minmax = [self.wires[0], self.wires[-1]]
        self.ancilla = ancilla
        self.controls = controls


class NumberEncoder(Ansatz):
    """X gates that write ``number`` (MSB on the first wire) into the register (reference: ansatz.py:311-347)."""

    def __init__(self, nqubit: int, number: int, minmax=None, den_mat=False, mps=False, chi=None) -> None:
        super().__init__(nqubit=nqubit, minmax=minmax, name='NumberEncoder', den_mat=den_mat, mps=mps, chi=chi)
        for wire, bit in zip(self.wires, int_to_bitstring(number, len(self.wires)), strict=True):
            if bit == '1':
                self.x(wire)


class QuantumFourierTransform(Ansatz):
    """QFT on ``minmax``; ``reverse=True`` leaves out the final bit-reversal swaps, i.e. the phases come
    out as x/2^n, ..., x/2 (reference: ansatz.py:565-618)."""

    def __init__(self, nqubit: int, minmax=None, reverse: bool = False, init_state: Any = 'zeros', den_mat=False,
                 mps=False, chi=None, show_barrier: bool = False) -> None:
        super().__init__(nqubit=nqubit, minmax=minmax, init_state=init_state, name='QuantumFourierTransform',
                         den_mat=den_mat, mps=mps, chi=chi)
        self.reverse = reverse
        _replay(self, _qft_steps(self.wires, swaps=not reverse, barrier=show_barrier))


class PhiAdder(Ansatz):
    """Phi(b) -> Phi(a + b) (reference: ansatz.py:350-396)."""

    def __init__(self, nqubit: int, number: int, minmax=None, controls=None, den_mat=False, mps=False, chi=None,
                 debug: bool = False) -> None:
        super().__init__(nqubit=nqubit, minmax=minmax, controls=controls, name='PhiAdder', den_mat=den_mat,
                         mps=mps, chi=chi)
        _replay(self, _phi_add_steps(self.wires, number, self.controls, debug))


class PhiModularAdder(Ansatz):
    """Phi(b) -> Phi(a + b mod N) with one ancilla (reference: ansatz.py:399-488)."""

    def __init__(self, nqubit: int, number: int, mod: int, minmax=None, ancilla=None, controls=None, den_mat=False,
                 mps=False, chi=None, debug: bool = False) -> None:
        if minmax is None:
            minmax = [0, nqubit - 2]
        if ancilla is None:
            ancilla = [minmax[1] + 1]
        super().__init__(nqubit=nqubit, minmax=minmax, ancilla=ancilla, controls=controls, name='PhiModularAdder',
                         den_mat=den_mat, mps=mps, chi=chi)
        if debug and number >= 2 * mod:
            print(f'The number {number} in {self.name} is too large.')
        _replay(self, _phi_mod_add_steps(self.wires, number, mod, self.ancilla[0], self.controls, debug))


class ControlledMultiplier(Ansatz):
    """|x>|b> -> |x>|b + a x mod N>: the first ``nqubitx`` wires hold x, the rest b
    (reference: ansatz.py:69-147)."""

    def __init__(self, nqubit: int, a: int, mod: int, minmax=None, nqubitx=None, ancilla=None, controls=None,
                 den_mat=False, mps=False, chi=None, debug: bool = False) -> None:
        assert isinstance(a, int) and isinstance(mod, int)
        if minmax is None:
            minmax = [0, nqubit - 2]
        if nqubitx is None:
            nqubitx = mod.bit_length()
        if ancilla is None:
            ancilla = [minmax[1] + 1]
        super().__init__(nqubit=nqubit, minmax=minmax, ancilla=ancilla, controls=controls,
                         name='ControlledMultiplier', den_mat=den_mat, mps=mps, chi=chi)
        # b needs one qubit more than N so that the intermediate sums cannot overflow
        assert len(self.wires) >= nqubitx + mod.bit_length() + 1, 'Quantum register is not enough.'
        _replay(self, _cmult_steps(self.wires[:nqubitx], self.wires[nqubitx:], a, mod, self.ancilla[0],
                                   self.controls, debug, self.name))


class ControlledUa(Ansatz):
    """|x> -> |a x mod N> for gcd(a, N) = 1; needs len(bin(N)) ancillas (reference: ansatz.py:150-233)."""

    def __init__(self, nqubit: int, a: int, mod: int, minmax=None, ancilla=None, controls=None, den_mat=False,
                 mps=False, chi=None, debug: bool = False) -> None:
        nregister = mod.bit_length()
        nancilla = nregister + 2
        if minmax is None:
            minmax = [0, nregister - 1]
        if ancilla is None:
            ancilla = list(range(minmax[1] + 1, minmax[1] + 1 + nancilla))
        super().__init__(nqubit=nqubit, minmax=minmax, ancilla=ancilla, controls=controls, name='ControlledUa',
                         den_mat=den_mat, mps=mps, chi=chi)
        assert len(self.wires) == nregister and len(self.ancilla) == nancilla
        # the multiplier sees x followed by the first nancilla - 1 ancillas as one contiguous register
        assert self.ancilla[:-1] == list(range(self.minmax[1] + 1, self.minmax[1] + nancilla)), \
            'the ancillas must follow the register'
        _replay(self, _cua_steps(self.wires, self.ancilla, a, mod, self.controls, debug))


class QuantumPhaseEstimation(Ansatz):
    """Phase estimation of an arbitrary unitary: ``ncount`` counting wires followed by the register
    (reference: ansatz.py:621-684)."""

    def __init__(self, nqubit: int, ncount: int, unitary: Any, minmax=None, den_mat=False, mps=False, chi=None,
                 show_barrier: bool = False) -> None:
        if not isinstance(unitary, torch.Tensor):
            unitary = torch.tensor(unitary, dtype=torch.cfloat)
        assert is_unitary(unitary)
        nreg = int(math.log2(len(unitary)))
        if minmax is None:
            minmax = [0, ncount + nreg - 1]
        assert minmax[1] - minmax[0] == ncount + nreg - 1
        self.unitary = unitary
        super().__init__(nqubit=nqubit, minmax=minmax, name='QuantumPhaseEstimation', den_mat=den_mat, mps=mps,
                         chi=chi)
        counting = self.wires[:ncount]
        register = self.wires[ncount:]
        self.hlayer(counting)
        if show_barrier:
            self.barrier()
        for i, wire in enumerate(counting):
            self.any(unitary=torch.linalg.matrix_power(unitary, 2 ** (ncount - 1 - i)), wires=register, controls=wire)
        if show_barrier:
            self.barrier()
        _replay(self, _qft_steps(counting, swaps=True, barrier=show_barrier), inverse=True)


class QuantumPhaseEstimationSingleQubit(Ansatz):
    """Phase estimation of diag(1, e^{2 pi i phase}) with ``t`` counting qubits (reference: ansatz.py:687-720)."""

    def __init__(self, t: int, phase: Any, den_mat=False, mps=False, chi=None) -> None:
        self.phase = phase
        super().__init__(nqubit=t + 1, name='QuantumPhaseEstimationSingleQubit', den_mat=den_mat, mps=mps, chi=chi)
        self.hlayer(list(range(t)))
        self.x(t)
        for i in range(t):
            self.cp(i, t, torch.pi * phase * (2 ** (t - i)))
        _replay(self, _qft_steps(list(range(t)), swaps=True), inverse=True)


class HHL(Ansatz):
    """HHL for a Hermitian ``mat``: wire 0 is the rotation ancilla, then ``ncount`` counting wires, then
    the register holding |b> (reference: ansatz.py:236-308)."""

    def __init__(self, ncount: int, mat: Any, t0: float = 1, den_mat=False, mps=False, chi=None,
                 show_barrier: bool = False) -> None:
        if not isinstance(mat, torch.Tensor):
            mat = torch.tensor(mat)
        t0 *= 2 * torch.pi
        unitary = torch.linalg.matrix_exp(1j * mat * t0 / 2**ncount)
        assert is_unitary(unitary)
        nqubit = 1 + ncount + int(math.log2(len(unitary)))
        self.unitary = unitary
        super().__init__(nqubit=nqubit, name='HHL', den_mat=den_mat, mps=mps, chi=chi)
        qpe = QuantumPhaseEstimation(nqubit=nqubit, ncount=ncount, unitary=unitary, minmax=[1, nqubit - 1],
                                     den_mat=den_mat, mps=mps, chi=chi, show_barrier=show_barrier)
        self.add(qpe)
        if show_barrier:
            self.barrier()
        counting = list(range(1, ncount + 1))
        for value in range(2**ncount):
            # eigenvalue register == value (LSB on wire 1): flip the zeros, rotate, flip back
            zeros = [1 + j for j in range(ncount) if not (value >> j) & 1]
            for w in zeros:
                self.x(w)
            self.ry(0, inputs=2 * torch.pi * value / 2**ncount, controls=counting)
            for w in zeros:
                self.x(w)
            if show_barrier:
                self.barrier()
        self.add(qpe.inverse())
        if show_barrier:
            self.barrier()


class QuantumConvolutionalNeuralNetwork(Ansatz):
    """QCNN with parameter sharing inside every convolution / pooling layer (reference: ansatz.py:491-562)."""

    def __init__(self, nqubit: int, nlayer: int, minmax=None, init_state: Any = 'zeros', den_mat=False,
                 requires_grad: bool = True, mps=False, chi=None) -> None:
        super().__init__(nqubit=nqubit, minmax=minmax, init_state=init_state,
                         name='QuantumConvolutionalNeuralNetwork', den_mat=den_mat, mps=mps, chi=chi)
        wires = self.wires
        self.requires_grad = requires_grad
        even, odd = self._u3(), self._u3()
        for i, wire in enumerate(wires[1::2]):
            self.add(even, wires=wires[2 * i])
            self.add(odd, wires=wire)
        for _ in range(nlayer):
            self.conv(wires)
            self.pool(wires)
            wires = wires[::2]
        self.latent(wires=wires)

    def _u3(self) -> U3Gate:
        return U3Gate(nqubit=self.nqubit, den_mat=self.den_mat, requires_grad=self.requires_grad)

    def conv(self, wires: list[int]) -> None:
        """Two staggered rows of Rxx Ryy Rzz + local U3 on neighbouring pairs; one parameter set per layer."""
        kw = dict(nqubit=self.nqubit, den_mat=self.den_mat, requires_grad=self.requires_grad)
        pair_gates = [Rxx(**kw), Ryy(**kw), Rzz(**kw)]
        left, right = self._u3(), self._u3()
        for start in (1, 2):
            for i, wire in enumerate(wires[start::2]):
                partner = wires[2 * i + start - 1]
                for g in pair_gates:
                    self.add(g, wires=[partner, wire])
                self.add(left, wires=partner)
                self.add(right, wires=wire)

    def pool(self, wires: list[int]) -> None:
        cu = self._u3()
        for i, wire in enumerate(wires[1::2]):
            self.add(cu, wires=wires[2 * i], controls=wire)


class RandomCircuitG3(Ansatz):
    """``ngate`` gates drawn from {CNOT, H, T} with Python's ``random`` module, so ``random.seed`` reproduces
    the reference's circuit draw by draw (reference: ansatz.py:723-771)."""

    def __init__(self, nqubit: int, ngate: int, wires=None, minmax=None, init_state: Any = 'zeros', den_mat=False,
                 mps=False, chi=None) -> None:
        super().__init__(nqubit=nqubit, wires=wires, minmax=minmax, init_state=init_state, name='RandomCircuitG3',
                         den_mat=den_mat, mps=mps, chi=chi)
        self.ngate = ngate
        self.gate_set = ['CNOT', 'H', 'T']
        for _ in range(ngate):
            gate = random.sample(self.gate_set, 1)[0]
            where = random.sample(self.wires, 2 if gate == 'CNOT' else 1)
            if gate == 'CNOT':
                self.cnot(where[0], where[1])
            elif gate == 'H':
                self.h(where)
            else:
                self.t(where)


class ShorCircuit(Ansatz):
    """Order finding for ``a`` modulo the odd number ``mod``: ncount counting wires, the register (initialised
    to 1), then len(bin(mod)) ancillas (reference: ansatz.py:774-837)."""

    def __init__(self, mod: int, ncount: int, a: int, den_mat=False, mps=False, chi=None, debug: bool = False) -> None:
        nreg = mod.bit_length()
        nqubit = ncount + 2 * nreg + 2
        super().__init__(nqubit=nqubit, name='ShorCircuit', den_mat=den_mat, mps=mps, chi=chi)
        register = list(range(ncount, ncount + nreg))
        ancilla = list(range(ncount + nreg, nqubit))
        self.hlayer(list(range(ncount)))
        self.x(register[-1])
        power = a % mod
        for control in range(ncount - 1, -1, -1):       # the last counting wire controls a^1, the one before a^2, ...
            _replay(self, _cua_steps(register, ancilla, power, mod, [control], debug))
            power = power * power % mod
        _replay(self, _qft_steps(list(range(ncount)), swaps=True), inverse=True)


class ShorCircuitFor15(Ansatz):
    """Order finding modulo 15 with hand-made multiplication circuits (reference: ansatz.py:840-896)."""

    def __init__(self, ncount: int, a: int, den_mat=False, mps=False, chi=None) -> None:
        self.ncount = ncount
        super().__init__(nqubit=ncount + 4, name='ShorCircuitFor15', den_mat=den_mat, mps=mps, chi=chi)
        self.hlayer(list(range(ncount)))
        self.x(ncount + 3)
        for n, control in enumerate(range(ncount - 1, -1, -1)):
            self.cua(a, 2**n, control)
        _replay(self, _qft_steps(list(range(ncount)), swaps=True), inverse=True)

    def cua(self, a: int, power: int, controls) -> None:
        """Controlled multiplication by a^power mod 15: multiplication by 2, 4, 8 is a rotation of the four
        register bits, the others follow by complementing (x -> 15 - x)."""
        assert a in [2, 4, 7, 8, 11, 13]
        r = [self.ncount + q for q in range(4)]
        rotation = {2: [(2, 3), (1, 2), (0, 1)], 13: [(2, 3), (1, 2), (0, 1)],
                    7: [(0, 1), (1, 2), (2, 3)], 8: [(0, 1), (1, 2), (2, 3)],
                    4: [(1, 3), (0, 2)], 11: [(1, 3), (0, 2)]}[a]
        for _ in range(power):
            for i, j in rotation:
                self.swap([r[i], r[j]], controls)
            if a in (7, 11, 13):
                for q in r:
                    self.x(q, controls)
