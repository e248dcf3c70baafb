"""Base classes of the gate/layer hierarchy, API-compatible with the reference's operation.py
(``Operation`` :16-108, ``Gate`` :110-409, ``Layer`` :412-522) for the statevector path.

What differs from the reference is everything below ``Gate.forward``: instead of permute / reshape /
matmul / cat on a (batch, 2, ..., 2) tensor, a gate describes itself as kernel-level primitives
(:meth:`Gate.prims`) and the executor hands them to the HIP kernels.

Density matrices (``den_mat=True``; reference ``evolve_den_mat`` qmath.py:509-540, ``op_den_mat_control``
operation.py:233-263, ``Channel`` :525-600) ride the same kernels: rho of n qubits is a 2n-"qubit" vector
whose high n index bits are the row and low n bits the column, a gate is U on the row bits and conj(U) on the
column bits (:meth:`Gate.dm_prims`), a channel is the 4x4 superoperator sum_k K_k (x) conj(K_k) on one
(row bit, column bit) pair.  MPS is not on this path and raises ``NotImplementedError``.
"""

from __future__ import annotations

from copy import copy
from dataclasses import replace
from typing import Any

import torch
from torch import nn

from . import executor
from .executor import Prim
from .utils import complex_apply


def tensor_version(t: torch.Tensor) -> int | None:
    """``t._version``, or None for a tensor that does not track one (a matrix computed under
    ``torch.inference_mode()``): such a tensor is never served from a cache."""
    if torch.is_inference(t):
        return None
    return t._version


_PLAIN_ATTRIBUTE_TYPES = frozenset((int, bool, float, str, list, tuple, dict, type(None)))


class Operation(nn.Module):
    r"""A quantum operation on ``nqubit`` qubits acting on ``wires``.

    ``tsr_mode=True`` means inputs/outputs are (batch, 2, ..., 2) tensors; otherwise (2**n, 1) /
    (batch, 2**n, 1) column vectors (same contract as the reference)."""

    def __init__(
        self,
        name: str | None = None,
        nqubit: int = 1,
        wires: int | list[int] | None = None,
        den_mat: bool = False,
        tsr_mode: bool = False,
    ) -> None:
        super().__init__()
        self.name = name
        self.nqubit = nqubit
        self.wires = wires
        self.den_mat = den_mat
        self.tsr_mode = tsr_mode
        self.npara = 0

    def __setattr__(self, name: str, value: Any) -> None:
        # A gate is a module with a dozen plain attributes, and building circuits is part of what the reference's own
        # benchmark times (examples/benchmarks/gradient_benchmark.py:127-144): numbers, strings, lists and None skip
        # nn.Module's search for parameters / buffers / submodules (6 us per assignment) unless the name is one of those.
        if type(value) in _PLAIN_ATTRIBUTE_TYPES:
            d = self.__dict__
            if name in d or not ('_parameters' in d and (name in d['_parameters'] or name in d['_buffers']
                                                         or name in d['_modules'])):
                object.__setattr__(self, name, value)
                return
        nn.Module.__setattr__(self, name, value)

    # ---- representations ----------------------------------------------------------------------------
    def tensor_rep(self, x: torch.Tensor) -> torch.Tensor:
        if self.den_mat:
            assert x.shape[-1] == x.shape[-2] == 2**self.nqubit
            return x.reshape([-1] + [2] * 2 * self.nqubit)
        if x.ndim == 1:
            assert x.shape[-1] == 2**self.nqubit
        else:
            assert x.shape[-1] == 2**self.nqubit or x.shape[-2] == 2**self.nqubit
        return x.reshape([-1] + [2] * self.nqubit)

    def vector_rep(self, x: torch.Tensor) -> torch.Tensor:
        return x.reshape(-1, 2**self.nqubit, 1)

    def matrix_rep(self, x: torch.Tensor) -> torch.Tensor:
        return x.reshape(-1, 2**self.nqubit, 2**self.nqubit)

    def get_unitary(self) -> torch.Tensor:
        raise NotImplementedError

    def init_para(self) -> None:
        pass

    def set_nqubit(self, nqubit: int) -> None:
        self.nqubit = nqubit

    def set_wires(self, wires: int | list[int]) -> None:
        self.wires = self._convert_indices(wires)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.tsr_mode:
            return self.tensor_rep(x)
        return self.matrix_rep(x) if self.den_mat else self.vector_rep(x)

    def _convert_indices(self, indices: int | list[int]) -> list[int]:
        if isinstance(indices, int):
            indices = [indices]
        assert isinstance(indices, list), 'Invalid input type'
        assert all(isinstance(i, int) for i in indices), 'Invalid input type'
        if indices:
            assert min(indices) > -1 and max(indices) < self.nqubit, 'Invalid input'
        assert len(set(indices)) == len(indices), 'Invalid input'
        return indices

    def _check_minmax(self, minmax: list[int]) -> None:
        assert isinstance(minmax, list) and len(minmax) == 2
        assert all(isinstance(i, int) for i in minmax)
        assert -1 < minmax[0] <= minmax[1] < self.nqubit

    def _flat_state(self, x: torch.Tensor) -> torch.Tensor:
        """Any accepted state representation -> (batch, 2**n), or (batch, 4**n) for density matrices."""
        x = self.tensor_rep(x)
        return x.reshape(x.shape[0], -1)


class Gate(Operation):
    r"""Base class of gates: a 2^k x 2^k ``matrix`` on ``wires``, optionally conditioned on ``controls``."""

    _qasm_new_gate = ['c3x', 'c4x']
    #: how the kernels may treat the matrix: 'gen' dense, 'diag' diagonal, 'x' bit-flip permutation
    _kernel_kind = 'gen'
    #: structure of a 2x2 matrix known from the class: 0 general, 1 all real, 2 real diag + imaginary off-diag
    _kernel_mode = 0
    #: structure of a 4x4 matrix known from the class (gates on two wires): 0 general, 4 / 5 X-shaped real / complex
    _kernel_mode2 = 0

    def __init__(
        self,
        name: str | None = None,
        nqubit: int = 1,
        wires: int | list[int] | None = None,
        controls: int | list[int] | None = None,
        condition: bool = False,
        den_mat: bool = False,
        tsr_mode: bool = False,
    ) -> None:
        self.nqubit = nqubit
        wires = self._convert_indices([0] if wires is None else wires)
        controls = self._convert_indices([] if controls is None else controls)
        assert not set(wires) & set(controls), 'Use repeated wires'
        if condition:
            assert len(controls) > 0
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        self.controls = controls
        self.condition = condition
        self.nodes = self.wires
        self.nancilla = 1

    def _apply(self, fn: Any, *args, **kwargs) -> 'Gate':
        # .to(torch.double) must promote the complex64 matrix buffer to complex128 (utils.complex_apply)
        if self.npara > 0:
            return super()._apply(fn, *args, **kwargs)
        held = {}
        if 'matrix' in self._buffers and self._buffers['matrix'] is not None:
            held['matrix'] = self._buffers.pop('matrix')
        super()._apply(fn, *args, **kwargs)
        for key, value in complex_apply(fn, held).items():
            self.register_buffer(key, value)
        return self

    def set_controls(self, controls: int | list[int]) -> None:
        self.controls = self._convert_indices(controls)

    def get_matrix(self, inputs: Any) -> torch.Tensor:
        return self.matrix

    def update_matrix(self) -> torch.Tensor:
        return self.matrix

    def _real_wrapper(self, x: Any) -> torch.Tensor:
        return torch.view_as_real(self.get_matrix(x))

    def get_derivative(self, inputs: Any) -> torch.Tensor:
        return torch.zeros_like(self.matrix)

    # ---- kernel-level description -------------------------------------------------------------------
    def _bits(self, wires: list[int]) -> tuple[int, ...]:
        return tuple(self.nqubit - 1 - w for w in wires)

    def prims(self, decompose: bool = True) -> list[Prim]:
        """The gate as kernel primitives.  ``decompose=True`` may split permutation gates into
        CNOT-like bit flips (exactly equal results, cheaper in the fused kernel)."""
        m = self.update_matrix()
        # the same matrix object (at the same version: no in-place write since) on the same wires as last time: the same
        # (immutable) primitives -- a gate with fixed
        # angles costs a forward one dictionary look-up instead of a Prim and two tuples
        d = self.__dict__
        c = d.get('_prims_cache')
        ver = tensor_version(m)
        if (c is not None and ver is not None and c[0] is m and c[5] == ver and c[1] == self.wires
                and c[2] == self.controls and c[3] == self.nqubit):
            return c[4]
        # (what does not depend on the matrix -- bit positions, mode -- once per gate: a trainable gate gets a new matrix, and
        # with it a new primitive, every forward)
        st = d.get('_prim_struct')
        if st is None or st[0] != self.wires or st[1] != self.controls or st[2] != self.nqubit or st[3] != self._kernel_kind:
            mode = self._kernel_mode if len(self.wires) == 1 else (self._kernel_mode2 if len(self.wires) == 2 else 0)
            st = d['_prim_struct'] = (list(self.wires), list(self.controls), self.nqubit, self._kernel_kind,
                                      self._bits(self.wires), self._bits(self.controls), mode)
        out = [Prim(st[3], m, st[4], st[5], st[6], exact=d.get('_exact_unitary', True))]
        # (not a matrix that carries an autograd graph: it is a new object every forward, and keeping it would keep its graph
        # -- with the AccumulateGrad nodes of the parameters and the stream they were made on -- alive until the next forward,
        # by which time the next graph has already picked the same nodes up: a training step captured into a HIP graph after
        # eager steps on the default stream then drags the default stream into the capture)
        d['_prims_cache'] = None if m.requires_grad else (m, list(self.wires), list(self.controls), self.nqubit, out, ver)
        return out

    def dm_prims(self, decompose: bool = True) -> list[Prim]:
        """The gate acting on a vectorised density matrix (row bits n..2n-1, column bits 0..n-1):
        every statevector primitive once on the row bits and once, complex-conjugated, on the column bits."""
        return lift_to_density_matrix(self.prims(decompose), self.nqubit)

    # ---- forward ------------------------------------------------------------------------------------
    def op_den_mat(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, 2, ..., 2) with 2n qubit axes -> same (reference: operation.py:221-263)."""
        shape = x.shape
        out = executor.run(x.reshape(shape[0], -1), self.dm_prims(decompose=False))
        return out.reshape(shape)

    def op_state(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, 2, ..., 2) -> same; replaces op_state_base / op_state_control of the reference."""
        shape = x.shape
        out = executor.run(x.reshape(shape[0], -1), self.prims(decompose=False))
        return out.reshape(shape)

    def op_dist_state(self, x):
        from .distributed import dist_gate

        return dist_gate(x, self)

    def forward(self, x):
        from .state import DistributedQubitState

        if isinstance(x, DistributedQubitState):
            return self.op_dist_state(x)
        if self.den_mat:
            if self.tsr_mode:
                assert x.ndim == 2 * self.nqubit + 1
                return self.op_den_mat(x)
            x = self.op_den_mat(self.tensor_rep(x))
            return self.matrix_rep(x).squeeze(0)
        if self.tsr_mode:
            assert x.ndim == self.nqubit + 1
            return self.op_state(x)
        x = self.op_state(self.tensor_rep(x))
        return self.vector_rep(x).squeeze(0)

    def inverse(self) -> 'Gate':
        return self

    def qpd(self, label: int | None = None) -> 'Gate':
        return self

    def get_unitary(self) -> torch.Tensor:
        """Global 2^n x 2^n matrix: the gate applied to the columns of the identity (the trick
        ArbitraryGate.get_unitary already uses in the reference, gate.py:326-330)."""
        matrix = self.update_matrix()
        dim = 2**self.nqubit
        eye = torch.eye(dim, dtype=matrix.dtype, device=matrix.device)
        with torch.no_grad():
            cols = executor.run(eye, self.prims(decompose=False))
        return cols.T.contiguous()

    def extra_repr(self) -> str:
        s = f'wires={self.wires}'
        return s if self.controls == [] else s + f', controls={self.controls}'


def lift_to_density_matrix(prims: list[Prim], nqubit: int) -> list[Prim]:
    """Statevector primitives on n qubits -> primitives on the 2n index bits of vec(rho):
    rho -> U rho U^dagger is U on the row bits (bit + n) and conj(U) on the column bits."""
    out: list[Prim] = []
    for p in prims:
        # (``replace`` keeps every other field: unitary / exact / order travel with the primitive)
        out.append(replace(p, targets=tuple(t + nqubit for t in p.targets), controls=tuple(c + nqubit for c in p.controls),
                           order=tuple(o + nqubit for o in p.order)))
        conj = p.matrix if p.kind == 'x' else p.matrix.conj().resolve_conj()
        out.append(replace(p, matrix=conj))
    return out


class Channel(Operation):
    r"""Base class of single-qubit noise channels :math:`\rho \to \sum_k K_k \rho K_k^\dagger` with the
    Kraus operators a function of ``theta`` (error probability :math:`\sin^2\theta`); density matrices only
    (reference: operation.py:525-600).  On the kernels a channel is ONE non-unitary two-"qubit" gate: the
    superoperator :math:`\sum_k K_k \otimes \bar K_k` on the (row bit, column bit) pair of its wire."""

    def __init__(self, inputs: Any = None, name: str | None = None, nqubit: int = 1,
                 wires: int | list[int] | None = None, tsr_mode: bool = False, requires_grad: bool = False) -> None:
        self.nqubit = nqubit
        wires = self._convert_indices([0] if wires is None else wires)
        super().__init__(name=name, nqubit=nqubit, wires=wires, den_mat=True, tsr_mode=tsr_mode)
        self.npara = 1
        self.requires_grad = requires_grad
        self.init_para(inputs)

    @property
    def prob(self) -> torch.Tensor:
        """The error probability."""
        return torch.sin(self.theta) ** 2

    def inputs_to_tensor(self, inputs: Any = None) -> torch.Tensor:
        while isinstance(inputs, list):
            inputs = inputs[0]
        if inputs is None:
            inputs = torch.rand(1)[0] * torch.pi
        elif not isinstance(inputs, torch.Tensor):
            inputs = torch.tensor(inputs, dtype=torch.float)
        return inputs

    def get_matrix(self, theta: Any) -> torch.Tensor:
        """The Kraus operators, stacked: (K, 2, 2)."""
        raise NotImplementedError

    def update_matrix(self) -> torch.Tensor:
        matrix = self.get_matrix(self.theta)
        self.matrix = matrix.detach()
        return matrix

    def init_para(self, inputs: Any = None) -> None:
        theta = self.inputs_to_tensor(inputs)
        if 'theta' in self._parameters:
            del self._parameters['theta']
        if 'theta' in self._buffers:
            del self._buffers['theta']
        if self.requires_grad:
            self.theta = nn.Parameter(theta)
        else:
            self.register_buffer('theta', theta)
        self.update_matrix()

    def superoperator(self) -> torch.Tensor:
        """sum_k K_k (x) conj(K_k): 4x4, index = (row bit, column bit)."""
        theta = self.theta
        ver = tensor_version(theta)
        key = (id(theta), ver, theta.dtype, theta.device)
        cached = self.__dict__.get('_sup_cache')
        if ver is not None and cached is not None and cached[0] == key and not theta.requires_grad:
            return cached[1]                                # fixed noise strength: built once, not per forward
        kraus = self.update_matrix()                       # (K, 2, 2) or (K, B, 2, 2)
        sup = torch.einsum('k...ab,k...cd->...acbd', kraus, kraus.conj())
        sup = sup.reshape(*sup.shape[:-4], 4, 4)
        if not theta.requires_grad:
            self.__dict__['_sup_cache'] = (key, sup)
        return sup

    #: 'gen': dense real superoperator (the kernels skip its exact zeros); 'diag' for channels whose Kraus
    #: operators are all diagonal (phase flip, phase damping): a diagonal two-"qubit" gate, no tile constraint
    _kernel_kind = 'gen'
    #: structure of the 4x4 superoperator promised by the class (include/dq_hip.h, DqFusedMode): 1 real; 4 real and X-shaped
    #: -- every Kraus operator diagonal or anti-diagonal, so that K (x) conj(K) keeps or flips BOTH bits of the (row, column)
    #: pair: a 2x2 block on (00, 11) and one on (01, 10).  The classes of channel.py say 4; a subclass with other Kraus
    #: operators inherits 1
    _kernel_mode = 1

    def dm_prims(self, decompose: bool = True) -> list[Prim]:
        bit = self.nqubit - 1 - self.wires[0]
        # every channel of channel.py has a REAL superoperator (K (x) conj(K) of Pauli / damping operators)
        return [Prim(self._kernel_kind, self.superoperator(), (bit + self.nqubit, bit), (), self._kernel_mode, unitary=False)]

    def prims(self, decompose: bool = True) -> list[Prim]:
        raise NotImplementedError('a channel acts on density matrices only')

    def op_den_mat(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        out = executor.run(x.reshape(shape[0], -1), self.dm_prims())
        return out.reshape(shape)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.tsr_mode:
            x = self.tensor_rep(x)
        assert x.ndim == 2 * self.nqubit + 1
        x = self.op_den_mat(x)
        return x if self.tsr_mode else self.matrix_rep(x).squeeze(0)

    def extra_repr(self) -> str:
        return f'wires={self.wires}, probability={self.prob.item()}'


class Layer(Operation):
    r"""A group of gates added to a circuit at once; ``wires`` is a list of wire lists."""

    def __init__(
        self,
        name: str | None = None,
        nqubit: int = 1,
        wires: int | list[int] | list[list[int]] | None = None,
        den_mat: bool = False,
        tsr_mode: bool = False,
    ) -> None:
        super().__init__(name=name, nqubit=nqubit, wires=None, den_mat=den_mat, tsr_mode=tsr_mode)
        self.wires = self._convert_indices([[0]] if wires is None else wires)
        self.gates = nn.Sequential()
        self.nodes = copy(self.wires)

    def get_unitary(self) -> torch.Tensor:
        u = None
        for gate in self.gates:
            u = gate.get_unitary() if u is None else gate.get_unitary() @ u
        return u

    def init_para(self, inputs: Any = None) -> None:
        count = 0
        for gate in self.gates:
            if inputs is None:
                gate.init_para()
            else:
                gate.init_para(inputs[..., count : count + gate.npara])
            count += gate.npara

    def update_npara(self) -> None:
        self.npara = sum(gate.npara for gate in self.gates)

    def set_nqubit(self, nqubit: int) -> None:
        self.nqubit = nqubit
        for gate in self.gates:
            gate.nqubit = nqubit

    def set_wires(self, wires: int | list[int] | list[list[int]]) -> None:
        self.wires = self._convert_indices(wires)
        for i, gate in enumerate(self.gates):
            gate.wires = self.wires[i]

    def prims(self, decompose: bool = True) -> list[Prim]:
        out: list[Prim] = []
        for gate in self.gates:
            out.extend(gate.prims(decompose))
        return out

    def dm_prims(self, decompose: bool = True) -> list[Prim]:
        return lift_to_density_matrix(self.prims(decompose), self.nqubit)

    def forward(self, x):
        from .state import DistributedQubitState

        if isinstance(x, DistributedQubitState):
            return self.gates(x)
        flat = self._flat_state(x)
        if self.den_mat:
            out = executor.run(flat, self.dm_prims())
            if self.tsr_mode:
                return out.reshape([-1] + [2] * (2 * self.nqubit))
            return self.matrix_rep(out).squeeze(0)
        out = executor.run(flat, self.prims())
        if self.tsr_mode:
            return out.reshape([-1] + [2] * self.nqubit)
        return self.vector_rep(out).squeeze(0)

    def inverse(self) -> 'Layer':
        return self

    def _convert_indices(self, indices: int | list) -> list[list[int]]:
        if isinstance(indices, int):
            indices = [[indices]]
        assert isinstance(indices, list), 'Invalid input type'
        if all(isinstance(i, int) for i in indices):
            indices = [[i] for i in indices]
        assert all(isinstance(i, list) for i in indices), 'Invalid input type'
        for idx in indices:
            assert all(isinstance(i, int) for i in idx), 'Invalid input type'
            assert min(idx) > -1 and max(idx) < self.nqubit, 'Invalid input'
            assert len(set(idx)) == len(idx), 'Invalid input'
        return indices
