"""Single-qubit noise channels for the density-matrix path (reference:
channel.py:16-383).  Each class only defines its Kraus operators as a function of ``theta`` (probabilities
are sin^2 theta, so any real input is valid); :class:`~deepquantum_amd.operation.Channel` turns them into one
4x4 superoperator on the (row bit, column bit) pair of the wire, which the gate kernels apply like any other
two-qubit matrix.  ``theta`` may carry a leading batch dimension (encoded data): the Kraus stack is then
(K, B, 2, 2) and the superoperator (B, 4, 4).
"""

from __future__ import annotations

from typing import Any

import torch

from .operation import Channel

_I = torch.tensor([[1, 0], [0, 1]], dtype=torch.cfloat)
_X = torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat)
_Y = torch.tensor([[0, -1j], [1j, 0]])
_Z = torch.tensor([[1, 0], [0, -1]], dtype=torch.cfloat)


def _col(weight: torch.Tensor) -> torch.Tensor:
    """A weight per sample as something that multiplies a 2x2 matrix: (1,) for one sample (the reference's
    shape), (B, 1, 1) for a batch."""
    weight = weight.reshape(-1)
    return weight if weight.numel() == 1 else weight.reshape(-1, 1, 1)


def _mat2(a, b, c, d) -> torch.Tensor:
    """[[a, b], [c, d]] from (broadcastable) real entries -> (..., 2, 2) complex."""
    a, b, c, d = torch.broadcast_tensors(a, b, c, d)
    return torch.stack([torch.stack([a, b], dim=-1), torch.stack([c, d], dim=-1)], dim=-2) + 0j


class _OneParameter(Channel):
    _name = None

    def __init__(self, inputs: Any = None, nqubit: int = 1, wires: int | list[int] | None = None,
                 tsr_mode: bool = False, requires_grad: bool = False) -> None:
        super().__init__(inputs=inputs, name=self._name, nqubit=nqubit, wires=wires, tsr_mode=tsr_mode,
                         requires_grad=requires_grad)

    def _prob(self, theta: Any) -> torch.Tensor:
        return torch.sin(self.inputs_to_tensor(theta).reshape(-1)) ** 2


class BitFlip(_OneParameter):
    r""":math:`\rho \to (1-p)\rho + p X\rho X` (reference: channel.py:16-55)."""
    _name = 'BitFlip'
    _kernel_mode = 4      # I, X (Y, Z): diagonal or anti-diagonal Kraus operators -> a real X-shaped superoperator

    def get_matrix(self, theta: Any) -> torch.Tensor:
        p = self._prob(theta)
        return torch.stack([_col(torch.sqrt(1 - p)) * _I.to(p.device), _col(torch.sqrt(p)) * _X.to(p.device)])


class PhaseFlip(_OneParameter):
    r""":math:`\rho \to (1-p)\rho + p Z\rho Z` (reference: channel.py:58-97)."""
    _name = 'PhaseFlip'
    _kernel_kind = 'diag'

    def get_matrix(self, theta: Any) -> torch.Tensor:
        p = self._prob(theta)
        return torch.stack([_col(torch.sqrt(1 - p)) * _I.to(p.device), _col(torch.sqrt(p)) * _Z.to(p.device)])


class Depolarizing(_OneParameter):
    r""":math:`\rho \to (1-p)\rho + \tfrac p3 (X\rho X + Y\rho Y + Z\rho Z)` (reference: channel.py:100-149)."""
    _name = 'Depolarizing'
    _kernel_mode = 4      # I, X (Y, Z): diagonal or anti-diagonal Kraus operators -> a real X-shaped superoperator

    def get_matrix(self, theta: Any) -> torch.Tensor:
        p = self._prob(theta)
        third = _col(torch.sqrt(p / 3))
        return torch.stack([_col(torch.sqrt(1 - p)) * _I.to(p.device), third * _X.to(p.device),
                            third * _Y.to(p.device), third * _Z.to(p.device)])


class AmplitudeDamping(_OneParameter):
    r""":math:`K_0 = \mathrm{diag}(1, \sqrt{1-p})`, :math:`K_1 = \sqrt p\,|0\rangle\langle 1|`
    (reference: channel.py:215-263)."""
    _name = 'AmplitudeDamping'
    _kernel_mode = 4      # diag(1, .) and |0><1|: diagonal / anti-diagonal

    def get_matrix(self, theta: Any) -> torch.Tensor:
        p = self._prob(theta)
        zero, one = torch.zeros_like(p), torch.ones_like(p)
        k = torch.stack([_mat2(one, zero, zero, torch.sqrt(1 - p)), _mat2(zero, torch.sqrt(p), zero, zero)])
        return k.squeeze(1) if p.numel() == 1 else k


class PhaseDamping(_OneParameter):
    r""":math:`K_0 = \mathrm{diag}(1, \sqrt{1-p})`, :math:`K_1 = \mathrm{diag}(0, \sqrt p)`
    (reference: channel.py:266-314)."""
    _name = 'PhaseDamping'
    _kernel_kind = 'diag'

    def get_matrix(self, theta: Any) -> torch.Tensor:
        p = self._prob(theta)
        zero, one = torch.zeros_like(p), torch.ones_like(p)
        k = torch.stack([_mat2(one, zero, zero, torch.sqrt(1 - p)), _mat2(zero, zero, zero, torch.sqrt(p))])
        return k.squeeze(1) if p.numel() == 1 else k


class Pauli(Channel):
    r""":math:`\rho \to p_i\rho + p_x X\rho X + p_y Y\rho Y + p_z Z\rho Z` with the four probabilities
    :math:`\sin^2\theta_j` normalised to one (reference: channel.py:152-212)."""
    _kernel_mode = 4

    def __init__(self, inputs: Any = None, nqubit: int = 1, wires: int | list[int] | None = None,
                 tsr_mode: bool = False, requires_grad: bool = False) -> None:
        super().__init__(inputs=inputs, name='Pauli', nqubit=nqubit, wires=wires, tsr_mode=tsr_mode,
                         requires_grad=requires_grad)
        self.npara = 4

    @property
    def prob(self) -> torch.Tensor:
        prob = torch.sin(self.theta) ** 2
        return prob / prob.sum(-1, keepdim=True)

    def inputs_to_tensor(self, inputs: Any = None) -> torch.Tensor:
        if inputs is None:
            inputs = torch.rand(4) * torch.pi
        elif not isinstance(inputs, torch.Tensor):
            inputs = torch.tensor(inputs, dtype=torch.float).reshape(-1)[:4]
        return inputs

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = self.inputs_to_tensor(theta)
        prob = torch.sin(theta.reshape(-1, 4)) ** 2
        prob = prob / prob.sum(-1, keepdim=True)
        w = [_col(torch.sqrt(prob[:, j])) for j in range(4)]
        dev = prob.device
        return torch.stack([w[0] * _I.to(dev), w[1] * _X.to(dev), w[2] * _Y.to(dev), w[3] * _Z.to(dev)])

    def extra_repr(self) -> str:
        p = self.prob.reshape(-1, 4)[0]
        return f'wires={self.wires}, px={p[1].item()}, py={p[2].item()}, pz={p[3].item()}'


class GeneralizedAmplitudeDamping(Channel):
    r"""Amplitude damping towards a thermal state: the first parameter gives the probability :math:`p` of the
    zero-temperature branch, the second the damping rate :math:`\gamma` (reference: channel.py:317-383)."""
    _kernel_mode = 4      # four Kraus operators, each diagonal or anti-diagonal

    def __init__(self, inputs: Any = None, nqubit: int = 1, wires: int | list[int] | None = None,
                 tsr_mode: bool = False, requires_grad: bool = False) -> None:
        super().__init__(inputs=inputs, name='GeneralizedAmplitudeDamping', nqubit=nqubit, wires=wires,
                         tsr_mode=tsr_mode, requires_grad=requires_grad)
        self.npara = 2

    def inputs_to_tensor(self, inputs: Any = None) -> torch.Tensor:
        if inputs is None:
            inputs = torch.rand(2) * torch.pi
        elif not isinstance(inputs, torch.Tensor):
            inputs = torch.tensor(inputs, dtype=torch.float).reshape(-1)[:2]
        return inputs

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = self.inputs_to_tensor(theta)
        prob = torch.sin(theta.reshape(-1, 2)) ** 2
        p, g = prob[:, 0], prob[:, 1]
        zero, one = torch.zeros_like(p), torch.ones_like(p)
        sp, sq = torch.sqrt(p).reshape(-1, 1, 1), torch.sqrt(1 - p).reshape(-1, 1, 1)
        k = torch.stack([sp * _mat2(one, zero, zero, torch.sqrt(1 - g)), sp * _mat2(zero, torch.sqrt(g), zero, zero),
                         sq * _mat2(torch.sqrt(1 - g), zero, zero, one), sq * _mat2(zero, zero, torch.sqrt(g), zero)])
        return k.squeeze(1) if prob.shape[0] == 1 else k

    def extra_repr(self) -> str:
        p = self.prob.reshape(-1, 2)[0]
        return f'wires={self.wires}, probability={p[0].item()}, rate={p[1].item()}'
