"""Layers of gates and the Pauli-string ``Observable``, API-compatible with the reference's layer.py
(``SingleLayer`` :14-48, ``ParametricSingleLayer`` :51-96, ``DoubleLayer`` :99-124, ``Observable``
:127-165, ``U3Layer``..``RzLayer`` :168-409, ``CnotLayer`` :412-443, ``CnotRing`` :446-483)."""

from __future__ import annotations

from copy import deepcopy
from typing import Any

import torch
from torch import nn

from .gate import CNOT, Hadamard, PauliX, PauliY, PauliZ, Rx, Ry, Rz, U3Gate
from .operation import Layer
from .qmath import multi_kron


class SingleLayer(Layer):
    """One single-qubit gate per listed wire."""

    def __init__(self, name=None, nqubit=1, wires=None, den_mat=False, tsr_mode=False):
        if wires is None:
            wires = [[i] for i in range(nqubit)]
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        assert all(len(w) == 1 for w in self.wires)

    def get_unitary(self) -> torch.Tensor:
        assert len(self.gates) > 0, 'There is no quantum gate'
        first = self.gates[0].update_matrix()
        lst = [torch.eye(2, dtype=first.dtype, device=first.device)] * self.nqubit
        for gate in self.gates:
            lst[gate.wires[0]] = gate.update_matrix()
        return multi_kron(lst)


class ParametricSingleLayer(SingleLayer):
    def __init__(self, name=None, nqubit=1, wires=None, den_mat=False, tsr_mode=False, requires_grad=True):
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        self.requires_grad = requires_grad

    def _fill(self, gate_cls, inputs: Any, per_gate: int) -> None:
        for i, wire in enumerate(self.wires):
            if inputs is None:
                val = None
            elif per_gate == 1:
                val = inputs[..., i] if isinstance(inputs, torch.Tensor) else inputs[i]
            else:
                val = inputs[..., per_gate * i : per_gate * (i + 1)] if isinstance(inputs, torch.Tensor) \
                    else inputs[per_gate * i : per_gate * (i + 1)]
            gate = gate_cls(inputs=val, nqubit=self.nqubit, wires=wire, den_mat=self.den_mat, tsr_mode=True,
                            requires_grad=self.requires_grad)
            self.gates.append(gate)
            self.npara += gate.npara

    def inverse(self) -> 'ParametricSingleLayer':
        layer = deepcopy(self)
        gates = nn.Sequential()
        for gate in self.gates[::-1]:
            gates.append(gate.inverse())
        layer.gates = gates
        layer.wires = self.wires[::-1]
        return layer


class DoubleLayer(Layer):
    def __init__(self, name=None, nqubit=2, wires=None, den_mat=False, tsr_mode=False):
        if wires is None:
            wires = [[i, i + 1] for i in range(0, nqubit - 1, 2)]
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        assert all(len(w) == 2 for w in self.wires)


class Observable(SingleLayer):
    """A Pauli string: ``basis[i]`` in {x, y, z} on ``wires[i]``."""

    def __init__(self, nqubit=1, wires=None, basis='z', den_mat=False, tsr_mode=False):
        super().__init__(name='Observable', nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        basis = basis.lower()
        self.basis = basis * len(self.wires) if len(basis) == 1 else basis
        assert len(self.wires) == len(self.basis), 'The number of wires is not equal to the number of bases'
        table = {'x': PauliX, 'y': PauliY, 'z': PauliZ}
        for wire, b in zip(self.wires, self.basis, strict=True):
            if b not in table:
                raise ValueError('Use illegal measurement basis')
            self.gates.append(table[b](nqubit=nqubit, wires=wire, den_mat=den_mat, tsr_mode=True))

    def pauli_masks(self) -> tuple[int, int]:
        """(xmask, zmask) over amplitude-index bits; a Y sets both."""
        xmask = zmask = 0
        for wire, b in zip(self.wires, self.basis, strict=True):
            bit = 1 << (self.nqubit - 1 - wire[0])
            if b in 'xy':
                xmask |= bit
            if b in 'zy':
                zmask |= bit
        return xmask, zmask


def _fixed_layer(cls_name: str, gate_cls, doc: str):
    def __init__(self, nqubit=1, wires=None, den_mat=False, tsr_mode=False):
        SingleLayer.__init__(self, name=cls_name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        for wire in self.wires:
            self.gates.append(gate_cls(nqubit=nqubit, wires=wire, den_mat=den_mat, tsr_mode=True))

    return type(cls_name, (SingleLayer,), {'__init__': __init__, '__doc__': doc})


XLayer = _fixed_layer('XLayer', PauliX, 'A layer of Pauli-X gates (reference: layer.py:204-226).')
YLayer = _fixed_layer('YLayer', PauliY, 'A layer of Pauli-Y gates (reference: layer.py:229-251).')
ZLayer = _fixed_layer('ZLayer', PauliZ, 'A layer of Pauli-Z gates (reference: layer.py:254-276).')
HLayer = _fixed_layer('HLayer', Hadamard, 'A layer of Hadamard gates (reference: layer.py:279-301).')


def _param_layer(cls_name: str, gate_cls, per_gate: int, doc: str):
    def __init__(self, nqubit=1, wires=None, inputs=None, den_mat=False, tsr_mode=False, requires_grad=True):
        ParametricSingleLayer.__init__(self, name=cls_name, nqubit=nqubit, wires=wires, den_mat=den_mat,
                                       tsr_mode=tsr_mode, requires_grad=requires_grad)
        self._fill(gate_cls, inputs, per_gate)

    return type(cls_name, (ParametricSingleLayer,), {'__init__': __init__, '__doc__': doc})


U3Layer = _param_layer('U3Layer', U3Gate, 3, 'A layer of U3 gates (reference: layer.py:168-201).')
RxLayer = _param_layer('RxLayer', Rx, 1, 'A layer of Rx gates (reference: layer.py:304-337).')
RyLayer = _param_layer('RyLayer', Ry, 1, 'A layer of Ry gates (reference: layer.py:340-373).')
RzLayer = _param_layer('RzLayer', Rz, 1, 'A layer of Rz gates (reference: layer.py:376-409).')


class CnotLayer(DoubleLayer):
    """A layer of CNOT gates, ``wires[i] = [control, target]``."""

    def __init__(self, nqubit=2, wires=None, name='CnotLayer', den_mat=False, tsr_mode=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        for wire in self.wires:
            self.gates.append(CNOT(nqubit=nqubit, wires=wire, den_mat=den_mat, tsr_mode=True))

    def inverse(self) -> 'CnotLayer':
        return CnotLayer(nqubit=self.nqubit, wires=list(reversed(self.wires)), name=self.name, den_mat=self.den_mat,
                         tsr_mode=self.tsr_mode)


class CnotRing(CnotLayer):
    """CNOTs chained cyclically over ``minmax`` with stride ``step``."""

    def __init__(self, nqubit=2, minmax=None, step=1, reverse=False, den_mat=False, tsr_mode=False):
        if minmax is None:
            minmax = [0, nqubit - 1]
        self.nqubit = nqubit
        self._check_minmax(minmax)
        assert minmax[0] < minmax[1]
        self.minmax, self.step, self.reverse = minmax, step, reverse
        lo, span = minmax[0], minmax[1] - minmax[0] + 1
        if reverse:
            wires = [[lo + i, lo + (i - step) % span] for i in range(span - 1, -1, -1)]
        else:
            wires = [[lo + i, lo + (i + step) % span] for i in range(span)]
        super().__init__(nqubit=nqubit, wires=wires, name='CnotRing', den_mat=den_mat, tsr_mode=tsr_mode)
