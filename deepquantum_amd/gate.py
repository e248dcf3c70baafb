"""Gate library, API-compatible with the reference's gate.py for the statevector path.

Matrices are built with exactly the arithmetic of the reference (file:line cited per class) because
parity to 1e-10 in complex128 requires bit-identical matrices: fixed gates are float32-rounded
complex64 buffers that ``.to(torch.double)`` merely widens (SURVEY summary item 3), parametric gates
evaluate cos/sin/exp in the precision of their parameter.  A parameter may carry a leading batch
dimension (one value per data sample): the matrix builders then return (B, D, D), which replaces the
``torch.vmap`` over the whole circuit that the reference uses for batched data (circuit.py:232-240).
"""

from __future__ import annotations

from copy import copy
from typing import Any

import torch
from torch import nn
from torch.autograd.functional import jacobian

from .executor import Prim
from .operation import Gate, tensor_version
from .qmath import multi_kron  # noqa: F401  (re-exported for API parity)


def _is_batched(t: torch.Tensor) -> bool:
    return t.ndim >= 1 and t.numel() > 1


def _prep(theta: torch.Tensor) -> torch.Tensor:
    """Parameter -> shape () (single) or (B,) (one value per sample)."""
    return theta.reshape(-1) if _is_batched(theta) else theta.reshape(())


def _mat(entries: list[torch.Tensor], d: int) -> torch.Tensor:
    """Row-major entries, each () or (B,), -> (d, d) or (B, d, d)."""
    m = torch.stack(entries, dim=-1)
    return m.reshape(*m.shape[:-1], d, d)


def _diag(entries: list[torch.Tensor]) -> torch.Tensor:
    return torch.stack(entries, dim=-1).diag_embed()


def _as_param_tensor(x: Any) -> torch.Tensor:
    if isinstance(x, (torch.Tensor, nn.Parameter)):
        return x
    return torch.tensor(x, dtype=torch.float)


# ======================================================================================================
# structural base classes
# ======================================================================================================
class SingleGate(Gate):
    """Single-qubit gate (reference: gate.py:20-85)."""

    def __init__(self, name=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)
        assert len(self.wires) == 1
        self.nancilla = 2


class DoubleGate(Gate):
    """Two-qubit gate (reference: gate.py:88-185)."""

    def __init__(self, name=None, nqubit=2, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        if wires is None:
            wires = [0, 1]
        assert len(wires) == 2
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)


class DoubleControlGate(DoubleGate):
    """Two-qubit gate of the form |0><0| (x) I + |1><1| (x) U with wires[0] the control
    (reference: gate.py:188-226).  The kernels see it as U on wires[1] controlled by wires[0]."""

    def __init__(self, name=None, nqubit=2, wires=None, den_mat=False, tsr_mode=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=None, condition=False,
                         den_mat=den_mat, tsr_mode=tsr_mode)

    def prims(self, decompose: bool = True) -> list[Prim]:
        full = self.update_matrix()
        d = self.__dict__
        c = d.get('_prims_cache')          # as in Gate.prims: same matrix object, version and wires -> same primitives
        ver = tensor_version(full)
        if c is not None and ver is not None and c[0] is full and c[3] == ver and c[1] == self.wires:
            return c[2]
        m = full[..., 2:4, 2:4]
        out = [Prim(self._sub_kind, m, self._bits([self.wires[1]]), self._bits([self.wires[0]]))]
        # (a matrix that carries an autograd graph is a new object every forward: keeping it would only keep that graph --
        # and the AccumulateGrad nodes of its parameters, with the stream they were made on -- alive until the next one)
        d['_prims_cache'] = None if full.requires_grad else (full, list(self.wires), out, ver)
        return out

    _sub_kind = 'gen'


class TripleGate(Gate):
    """Three-qubit gate (reference: gate.py:229-265)."""

    def __init__(self, name=None, nqubit=3, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        if wires is None:
            wires = [0, 1, 2]
        assert len(wires) == 3
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)


class ArbitraryGate(Gate):
    """Gate on an arbitrary list of wires or a ``minmax`` range (reference: gate.py:268-338)."""

    def __init__(self, name=None, nqubit=1, wires=None, minmax=None, controls=None, condition=False,
                 den_mat=False, tsr_mode=False):
        self.nqubit = nqubit
        if wires is None:
            if minmax is None:
                minmax = [0, nqubit - 1]
            self._check_minmax(minmax)
            wires = list(range(minmax[0], minmax[1] + 1))
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)
        self.minmax = [min(self.wires), max(self.wires)]
        self.local = all(b - a == 1 for a, b in zip(self.wires[:-1], self.wires[1:]))
        self.inv_mode = False

    def inverse(self) -> 'ArbitraryGate':
        gate = copy(self)
        gate.inv_mode = not self.inv_mode
        if isinstance(self.name, str):
            gate.name = self.name[:-7] if self.name.endswith('_dagger') else self.name + '_dagger'
        return gate


class _Parametric:
    """Shared machinery of parametric gates: parameter storage, inverse, derivative."""

    _param_names = ('theta',)

    def _setup_params(self, inputs: Any, requires_grad: bool) -> None:
        self.npara = len(self._param_names)
        self.requires_grad = requires_grad
        self.inv_mode = False
        self.init_para(inputs)

    def inputs_to_tensor(self, inputs: Any = None) -> torch.Tensor:
        """Reference: gate.py:368-376 -- nested lists unwrap to their first element, ``None`` draws
        U(0, 4 pi), Python numbers become float32 tensors."""
        while isinstance(inputs, list):
            inputs = inputs[0]
        if inputs is None:
            # (= torch.rand(1)[0] * 4 * torch.pi to the last bit, same draw from the generator; two tiny ops instead of four)
            return torch.rand(()).mul_(4 * torch.pi)
        return _as_param_tensor(inputs)

    # ``matrix`` is evaluated lazily: the reference recomputes it inside every ``init_para`` /
    # ``update_matrix`` call (gate.py:395-415), i.e. a handful of tiny device kernels per gate per forward.
    # Here ``init_para`` only stores the parameter; the circuit driver evaluates all gates of one class in
    # a single vectorised call (``QubitCircuit._precompute_matrices``), and ``gate.matrix`` computes on
    # demand when somebody reads it.
    @property
    def matrix(self) -> torch.Tensor:
        m = self.__dict__.get('_matrix_cache')
        if m is None:
            self.update_matrix()
            m = self.__dict__['_matrix_cache']
        return m

    @matrix.setter
    def matrix(self, value: torch.Tensor) -> None:
        self.__dict__['_matrix_cache'] = value

    def _invalidate(self) -> None:
        self.__dict__['_matrix_cache'] = None
        self.__dict__['_precomputed'] = None
        self.__dict__['_matrix_key'] = None

    def _param_key(self) -> tuple:
        """Identity and version of every parameter tensor (+ the inverse flag): what the cached matrix belongs to."""
        return tuple((t, tensor_version(t)) for t in (getattr(self, n) for n in self._param_names)) + (self.inv_mode,)

    def _stamp(self) -> None:
        self.__dict__['_matrix_key'] = self._param_key()

    def _fixed_matrix(self) -> torch.Tensor | None:
        """The cached matrix, if it is still the matrix of the current parameters and no autograd graph has to be
        built: a gate with fixed angles is evaluated once, not once per forward."""
        m, key = self.__dict__.get('_matrix_cache'), self.__dict__.get('_matrix_key')
        if m is None or key is None or key[-1] != self.inv_mode:
            return None
        grad = torch.is_grad_enabled()
        for (t, version), name in zip(key[:-1], self._param_names):
            cur = getattr(self, name)
            if cur is not t or version is None or tensor_version(cur) != version or (grad and cur.requires_grad):
                return None
        return m

    def init_para(self, inputs: Any = None) -> None:
        theta = self.inputs_to_tensor(inputs)
        if self.requires_grad:
            self.theta = nn.Parameter(theta)
        else:
            self.register_buffer('theta', theta)
        self._invalidate()

    def update_matrix(self) -> torch.Tensor:
        pre = self.__dict__.get('_precomputed')
        if pre is not None:
            return pre
        fixed = self._fixed_matrix()
        if fixed is not None:
            return fixed
        theta = -self.theta if self.inv_mode else self.theta
        matrix = self.get_matrix(theta)
        self.matrix = matrix.detach()
        self._stamp()
        return matrix

    def _apply(self, fn: Any, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def get_derivative(self, theta: Any) -> torch.Tensor:
        """dU/dtheta by differentiating the matrix builder (reference: gate.py:402-406)."""
        theta = self.inputs_to_tensor(theta).squeeze()
        du = jacobian(self._real_wrapper, theta)
        return du[..., 0] + du[..., 1] * 1j

    def inverse(self):
        gate = copy(self)
        gate.inv_mode = not self.inv_mode
        gate._invalidate()
        return gate

    def extra_repr(self) -> str:
        theta = -self.theta if self.inv_mode else self.theta
        val = theta.item() if theta.numel() == 1 else f'<batch of {theta.numel()}>'
        s = f'wires={self.wires}, theta={val}'
        return s if self.controls == [] else s + f', controls={self.controls}'


class ParametricSingleGate(_Parametric, SingleGate):
    """Single-qubit gate with one parameter (reference: gate.py:341-429)."""

    def __init__(self, name=None, inputs=None, nqubit=1, wires=None, controls=None, condition=False,
                 den_mat=False, tsr_mode=False, requires_grad=False):
        SingleGate.__init__(self, name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                            den_mat=den_mat, tsr_mode=tsr_mode)
        self._setup_params(inputs, requires_grad)


class ParametricDoubleGate(_Parametric, DoubleGate):
    """Two-qubit gate with one parameter (reference: gate.py:432-520)."""

    def __init__(self, name=None, inputs=None, nqubit=2, wires=None, controls=None, condition=False,
                 den_mat=False, tsr_mode=False, requires_grad=False):
        DoubleGate.__init__(self, name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                            den_mat=den_mat, tsr_mode=tsr_mode)
        self._setup_params(inputs, requires_grad)


# ======================================================================================================
# fixed single-qubit gates
# ======================================================================================================
def _fixed_single(cls_name: str, gate_name: str, build, kind: str = 'gen', inverse_name: str | None = None,
                  nancilla: int | None = None, doc: str = '', mode: int = 0):
    def __init__(self, nqubit=1, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        SingleGate.__init__(self, name=gate_name, nqubit=nqubit, wires=wires, controls=controls,
                            condition=condition, den_mat=den_mat, tsr_mode=tsr_mode)
        self.register_buffer('matrix', build())
        if nancilla is not None:
            self.nancilla = nancilla

    def inverse(self):
        if inverse_name is None:
            return self
        other = globals()[inverse_name]
        return other(nqubit=self.nqubit, wires=self.wires, controls=self.controls, condition=self.condition,
                     den_mat=self.den_mat, tsr_mode=self.tsr_mode).to(self.matrix.device, self.matrix.real.dtype)

    return type(cls_name, (SingleGate,), {'__init__': __init__, 'inverse': inverse, '_kernel_kind': kind,
                                          '_kernel_mode': mode, '__doc__': doc})


PauliX = _fixed_single('PauliX', 'PauliX', lambda: torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat), kind='x',
                       doc='Pauli-X (reference: gate.py:800-872, matrix :841).')
PauliY = _fixed_single('PauliY', 'PauliY', lambda: torch.tensor([[0, -1j], [1j, 0]]), nancilla=4,
                       doc='Pauli-Y (reference: gate.py:875-951, matrix :916).')
PauliZ = _fixed_single('PauliZ', 'PauliZ', lambda: torch.tensor([[1, 0], [0, -1]], dtype=torch.cfloat), kind='diag',
                       doc='Pauli-Z (reference: gate.py:954-1024, matrix :995).')
Hadamard = _fixed_single('Hadamard', 'Hadamard',
                         lambda: torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5, nancilla=1, mode=3,
                         doc='Hadamard; note the float32-rounded 1/sqrt(2) (reference: gate.py:1027-1099, :1069).')
SGate = _fixed_single('SGate', 'SGate', lambda: torch.tensor([[1, 0], [0, 1j]]), kind='diag',
                      inverse_name='SDaggerGate', doc='S (reference: gate.py:1102-1188, matrix :1143).')
SDaggerGate = _fixed_single('SDaggerGate', 'SDaggerGate', lambda: torch.tensor([[1, 0], [0, -1j]]), kind='diag',
                            inverse_name='SGate', doc='S^dagger (reference: gate.py:1191-1258, matrix :1233).')
TGate = _fixed_single('TGate', 'TGate', lambda: torch.tensor([[1, 0], [0, (1 + 1j) / 2**0.5]]), kind='diag',
                      inverse_name='TDaggerGate', doc='T (reference: gate.py:1261-1322, matrix :1303).')
TDaggerGate = _fixed_single('TDaggerGate', 'TDaggerGate', lambda: torch.tensor([[1, 0], [0, (1 - 1j) / 2**0.5]]),
                            kind='diag', inverse_name='TGate', doc='T^dagger (reference: gate.py:1325-1388, :1367).')


class Identity(Gate):
    """Identity on any number of wires (reference: gate.py:756-797)."""

    _kernel_kind = 'diag'

    def __init__(self, nqubit=1, wires=None, den_mat=False, tsr_mode=False):
        super().__init__(name='Identity', nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        self.register_buffer('matrix', torch.eye(2**self.nqubit, dtype=torch.cfloat))

    def prims(self, decompose: bool = True) -> list[Prim]:
        return []

    def get_unitary(self) -> torch.Tensor:
        return self.matrix

    def forward(self, x: Any) -> Any:
        return x


# ======================================================================================================
# parametric single-qubit gates
# ======================================================================================================
class U3Gate(ParametricSingleGate):
    r"""U3(theta, phi, lambda) (reference: gate.py:523-674, matrix :594-602)."""

    _param_names = ('theta', 'phi', 'lambd')

    def __init__(self, inputs=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        super().__init__(name='U3Gate', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)
        self.npara = 3

    def inputs_to_tensor(self, inputs: Any = None):
        if inputs is None:
            theta = torch.rand(()).mul_(torch.pi)          # (the reference's torch.rand(1)[0] * .., bit for bit)
            phi = torch.rand(()).mul_(2 * torch.pi)
            lambd = torch.rand(()).mul_(2 * torch.pi)
        elif isinstance(inputs, torch.Tensor) and inputs.ndim == 2:  # (batch, 3): one triple per sample
            theta, phi, lambd = inputs.unbind(-1)
        else:
            theta, phi, lambd = inputs[0], inputs[1], inputs[2]
        return _as_param_tensor(theta), _as_param_tensor(phi), _as_param_tensor(lambd)

    def get_matrix(self, theta: Any, phi: Any, lambd: Any) -> torch.Tensor:
        theta, phi, lambd = self.inputs_to_tensor([theta, phi, lambd])
        theta, phi, lambd = _prep(theta), _prep(phi), _prep(lambd)
        cos_t = torch.cos(theta / 2) + 0j
        sin_t = torch.sin(theta / 2) + 0j
        e_il = torch.exp(1j * lambd)
        e_ip = torch.exp(1j * phi)
        e_ipl = torch.exp(1j * (phi + lambd))
        return _mat([cos_t, -e_il * sin_t, e_ip * sin_t, e_ipl * cos_t], 2)

    def update_matrix(self) -> torch.Tensor:
        pre = self.__dict__.get('_precomputed')
        if pre is not None:
            return pre
        fixed = self._fixed_matrix()
        if fixed is not None:
            return fixed
        if self.inv_mode:
            theta, phi, lambd = -self.theta, -self.lambd, -self.phi
        else:
            theta, phi, lambd = self.theta, self.phi, self.lambd
        matrix = self.get_matrix(theta, phi, lambd)
        self.matrix = matrix.detach()
        self._stamp()
        return matrix

    def _real_wrapper(self, x: torch.Tensor) -> torch.Tensor:
        return torch.view_as_real(self.get_matrix(x[0], x[1], x[2]))

    def get_derivative(self, inputs: Any) -> torch.Tensor:
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.tensor(inputs, dtype=torch.float)
        du = jacobian(self._real_wrapper, inputs.reshape(self.npara)).permute(3, 0, 1, 2)
        return du[..., 0] + du[..., 1] * 1j

    def init_para(self, inputs: Any = None) -> None:
        theta, phi, lambd = self.inputs_to_tensor(inputs)
        for name, val in (('theta', theta), ('phi', phi), ('lambd', lambd)):
            if self.requires_grad:
                setattr(self, name, nn.Parameter(val))
            else:
                self.register_buffer(name, val)
        self._invalidate()

    def extra_repr(self) -> str:
        s = f'wires={self.wires}, npara=3'
        return s if self.controls == [] else s + f', controls={self.controls}'


class PhaseShift(ParametricSingleGate):
    r"""diag(1, e^{i theta}) (reference: gate.py:677-753, matrix :737-742)."""

    _kernel_kind = 'diag'

    def __init__(self, inputs=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        super().__init__(name='PhaseShift', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = _prep(self.inputs_to_tensor(theta))
        e_it = torch.exp(1j * theta)
        return _diag([torch.ones_like(e_it), e_it])


class Rx(ParametricSingleGate):
    r"""exp(-i theta X / 2) (reference: gate.py:1389-1480, matrix :1443-1448)."""

    _kernel_mode = 2  # cos + 0j on the diagonal, (0 -/+ i sin) off it: exact zeros by construction

    def __init__(self, inputs=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        super().__init__(name='Rx', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = _prep(self.inputs_to_tensor(theta))
        cos = torch.cos(theta / 2) + 0j
        isin = torch.sin(theta / 2) * 1j
        return _mat([cos, -isin, -isin, cos], 2)


class Ry(ParametricSingleGate):
    r"""exp(-i theta Y / 2) (reference: gate.py:1483-1579, matrix :1538-1543)."""

    _kernel_mode = 1  # real entries + 0j

    def __init__(self, inputs=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        super().__init__(name='Ry', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = _prep(self.inputs_to_tensor(theta))
        cos = torch.cos(theta / 2)
        sin = torch.sin(theta / 2)
        return _mat([cos, -sin, sin, cos], 2) + 0j


class Rz(ParametricSingleGate):
    r"""exp(-i theta Z / 2) (reference: gate.py:1582-1671, matrix :1634-1639)."""

    _kernel_kind = 'diag'

    def __init__(self, inputs=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        super().__init__(name='Rz', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = _prep(self.inputs_to_tensor(theta))
        return _diag([torch.exp(-1j * theta / 2), torch.exp(1j * theta / 2)])


class ProjectionJ(ParametricSingleGate):
    r"""Measurement-plane rotation J(theta) used by MBQC transpilation (reference: gate.py:1674-1787,
    matrix :1751-1767)."""

    def __init__(self, inputs=None, nqubit=1, wires=None, plane='xy', controls=None, condition=False,
                 den_mat=False, tsr_mode=False, requires_grad=False):
        self.plane = plane.lower()
        super().__init__(name='ProjectionJ', inputs=inputs, nqubit=nqubit, wires=wires, controls=controls,
                         condition=condition, den_mat=den_mat, tsr_mode=tsr_mode, requires_grad=requires_grad)

    def get_matrix(self, theta: Any) -> torch.Tensor:
        theta = _prep(self.inputs_to_tensor(theta))
        if self.plane in ('xy', 'yx'):
            one = torch.ones_like(theta) + 0j
            e_m = torch.exp(-1j * theta)
            return _mat([one, e_m, one, -e_m], 2) / 2**0.5
        if self.plane in ('yz', 'zy'):
            cps = torch.cos(theta / 2) + torch.sin(theta / 2) + 0j
            cms = torch.cos(theta / 2) - torch.sin(theta / 2) + 0j
            return _mat([cps, -1j * cms, cms, 1j * cps], 2) / 2**0.5
        if self.plane in ('zx', 'xz'):
            cos = torch.cos(theta / 2)
            sin = torch.sin(theta / 2)
            return _mat([cos, sin, sin, -cos], 2) + 0j
        raise ValueError(f'Unsupported measurement plane: {self.plane}')

    def extra_repr(self) -> str:
        return super().extra_repr() + f', plane={self.plane}'


class CombinedSingleGate(SingleGate):
    """Product of single-qubit gates on one wire, applied as one 2x2 matrix
    (reference: gate.py:1790-1903, matrix :1837-1848)."""

    def __init__(self, gates, name=None, nqubit=1, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)
        for g in gates:
            g.nqubit, g.wires, g.controls = self.nqubit, self.wires, self.controls
            g.condition, g.den_mat, g.tsr_mode = self.condition, self.den_mat, self.tsr_mode
        self.gates = nn.ModuleList(gates)
        self.update_npara()
        self.update_matrix()

    def get_matrix(self) -> torch.Tensor:
        matrix = None
        for g in self.gates:
            matrix = g.update_matrix() if matrix is None else g.update_matrix() @ matrix
        return matrix

    def update_matrix(self) -> torch.Tensor:
        matrix = self.get_matrix()
        self.matrix = matrix.detach()
        return matrix

    def get_derivative(self, inputs: Any) -> torch.Tensor:
        """d(matrix)/d(each parameter): product rule over the factors (reference: gate.py:1850-1863)."""
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.tensor(inputs, dtype=torch.float)
        inputs = inputs.reshape(self.npara)
        mats = [g.update_matrix() for g in self.gates]
        out, count = [], 0
        for i, g in enumerate(self.gates):
            if g.npara == 0:
                continue
            du = g.get_derivative(inputs[count : count + g.npara])
            du = du.unsqueeze(0) if du.ndim == 2 else du
            for d in du:
                acc = None
                for j, m in enumerate(mats):
                    term = d if j == i else m
                    acc = term if acc is None else term @ acc
                out.append(acc)
            count += g.npara
        return torch.stack(out)

    def update_npara(self) -> None:
        self.npara = sum(g.npara for g in self.gates)

    def add(self, gate: SingleGate) -> None:
        gate.nqubit, gate.wires, gate.controls = self.nqubit, self.wires, self.controls
        gate.condition, gate.den_mat, gate.tsr_mode = self.condition, self.den_mat, self.tsr_mode
        self.gates.append(gate)
        self.matrix = gate.matrix @ self.matrix
        self.npara += gate.npara

    def init_para(self, inputs: Any = None) -> None:
        count = 0
        for g in self.gates:
            if g.npara == 0:
                continue
            g.init_para(None if inputs is None else inputs[..., count : count + g.npara])
            count += g.npara
        self.update_matrix()

    def inverse(self) -> 'CombinedSingleGate':
        # a NEW gate, as the reference builds one (gate.py:1883-1900): a shallow copy would share ``_modules`` with
        # ``self``, so assigning ``inv.gates`` would also replace the original's factors
        name = self.name + '_dagger' if isinstance(self.name, str) else self.name
        return CombinedSingleGate(gates=[g.inverse() for g in reversed(self.gates)], name=name, nqubit=self.nqubit,
                                  wires=self.wires, controls=self.controls, condition=self.condition,
                                  den_mat=self.den_mat, tsr_mode=self.tsr_mode)


# ======================================================================================================
# two-qubit gates
# ======================================================================================================
class CNOT(DoubleControlGate):
    """CNOT with wires = [control, target] (reference: gate.py:1906-1960, matrix :1934)."""

    _sub_kind = 'x'

    def __init__(self, nqubit=2, wires=None, den_mat=False, tsr_mode=False):
        super().__init__(name='CNOT', nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        self.register_buffer('matrix', torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]]) + 0j)
        self.nancilla = 2


def _swap_prims(gate: Gate, a: int, b: int, extra_controls: list[int]) -> list[Prim]:
    """SWAP(a, b) = CNOT(a,b) CNOT(b,a) CNOT(a,b): three exact bit-flip permutations."""
    x = gate.matrix.new_tensor([[0, 1], [1, 0]])
    ca, cb = gate._bits([a])[0], gate._bits([b])[0]
    ec = gate._bits(extra_controls)
    return [Prim('x', x, (cb,), (ca,) + ec), Prim('x', x, (ca,), (cb,) + ec), Prim('x', x, (cb,), (ca,) + ec)]


class Swap(DoubleGate):
    """SWAP (reference: gate.py:1963-2023, matrix :2006)."""

    def __init__(self, nqubit=2, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        super().__init__(name='Swap', nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)
        self.register_buffer('matrix', torch.tensor([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) + 0j)

    def prims(self, decompose: bool = True) -> list[Prim]:
        if decompose:
            return _swap_prims(self, self.wires[0], self.wires[1], self.controls)
        return super().prims(decompose)


class ImaginarySwap(DoubleGate):
    """iSWAP (reference: gate.py:2026-2082, matrix :2069)."""

    def __init__(self, nqubit=2, wires=None, controls=None, condition=False, den_mat=False, tsr_mode=False):
        super().__init__(name='ImaginarySwap', nqubit=nqubit, wires=wires, controls=controls, condition=condition,
                         den_mat=den_mat, tsr_mode=tsr_mode)
        self.register_buffer('matrix', torch.tensor([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]]))


def _param_double(cls_name: str, gate_name: str, builder, kind: str = 'gen', doc: str = '', mode2: int = 0):
    def __init__(self, inputs=None, nqubit=2, wires=None, controls=None, condition=False, den_mat=False,
                 tsr_mode=False, requires_grad=False):
        ParametricDoubleGate.__init__(self, name=gate_name, inputs=inputs, nqubit=nqubit, wires=wires,
                                      controls=controls, condition=condition, den_mat=den_mat, tsr_mode=tsr_mode,
                                      requires_grad=requires_grad)

    def get_matrix(self, theta):
        return builder(_prep(self.inputs_to_tensor(theta)))

    return type(cls_name, (ParametricDoubleGate,),
                {'__init__': __init__, 'get_matrix': get_matrix, '_kernel_kind': kind, '_kernel_mode2': mode2, '__doc__': doc})


def _antidiag(entries: list[torch.Tensor]) -> torch.Tensor:
    return torch.stack(entries, dim=-1).diag_embed().flip(-1)


def _rxx(theta):
    cos = torch.cos(theta / 2)
    isin = torch.sin(theta / 2) * 1j
    return _diag([cos, cos, cos, cos]) + _antidiag([-isin, -isin, -isin, -isin])


def _ryy(theta):
    cos = torch.cos(theta / 2)
    isin = torch.sin(theta / 2) * 1j
    return _diag([cos, cos, cos, cos]) + _antidiag([isin, -isin, -isin, isin])


def _rzz(theta):
    e_m = torch.exp(-1j * theta / 2)
    e_p = torch.exp(1j * theta / 2)
    return _diag([e_m, e_p, e_p, e_m])


def _embed_middle(block: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """block_diag(1, block(2x2), 1) with optional leading batch dim."""
    out = torch.zeros(*block.shape[:-2], 4, 4, dtype=block.dtype, device=block.device)
    out[..., 0, 0] = 1
    out[..., 3, 3] = 1
    out[..., 1:3, 1:3] = block
    return out


def _rxy(theta):
    cos = torch.cos(theta / 2) + 0j
    isin = torch.sin(theta / 2) * 1j
    return _embed_middle(_mat([cos, -isin, -isin, cos], 2), theta)


def _rbs(theta):
    cos = torch.cos(theta)
    sin = torch.sin(theta)
    return _embed_middle(_mat([cos, sin, -sin, cos], 2) + 0j, theta)


# (mode2: the structure of the 4x4 matrix the class promises -- include/dq_hip.h, DQ_MODE_XCPLX = 5: non-zero only on the
# blocks (00, 11) and (01, 10); DQ_MODE_XREAL = 4: and real)
Rxx = _param_double('Rxx', 'Rxx', _rxx, mode2=5, doc='exp(-i theta XX / 2) (reference: gate.py:2085-2155, matrix :2139-2146).')
Ryy = _param_double('Ryy', 'Ryy', _ryy, mode2=5, doc='exp(-i theta YY / 2) (reference: gate.py:2158-2238, matrix :2212-2219).')
Rzz = _param_double('Rzz', 'Rzz', _rzz, kind='diag',
                    doc='exp(-i theta ZZ / 2) (reference: gate.py:2241-2309, matrix :2295-2300).')
Rxy = _param_double('Rxy', 'Rxy', _rxy, mode2=5, doc='exp(-i theta (XX+YY) / 4) (reference: gate.py:2312-2390, :2366-2373).')
ReconfigurableBeamSplitter = _param_double(
    'ReconfigurableBeamSplitter', 'ReconfigurableBeamSplitter', _rbs, mode2=4,
    doc='RBS gate (reference: gate.py:2393-2479, matrix :2455-2462).')


# ======================================================================================================
# three-qubit gates
# ======================================================================================================
class Toffoli(TripleGate):
    """CCX with wires = [control1, control2, target] (reference: gate.py:2482-2649, matrix :2521-2536)."""

    def __init__(self, nqubit=3, wires=None, den_mat=False, tsr_mode=False):
        super().__init__(name='Toffoli', nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        m = torch.eye(8) + 0j
        m[6:8, 6:8] = torch.tensor([[0, 1], [1, 0]]) + 0j
        self.register_buffer('matrix', m)

    def prims(self, decompose: bool = True) -> list[Prim]:
        return [Prim('x', self.matrix[6:8, 6:8], self._bits([self.wires[2]]), self._bits(self.wires[:2]))]


class Fredkin(TripleGate):
    """CSWAP with wires = [control, target1, target2] (reference: gate.py:2652-2742, matrix :2691-2706)."""

    def __init__(self, nqubit=3, wires=None, den_mat=False, tsr_mode=False):
        super().__init__(name='Fredkin', nqubit=nqubit, wires=wires, den_mat=den_mat, tsr_mode=tsr_mode)
        m = torch.eye(8) + 0j
        m[4:8, 4:8] = torch.tensor([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) + 0j
        self.register_buffer('matrix', m)

    def prims(self, decompose: bool = True) -> list[Prim]:
        if decompose:
            return _swap_prims(self, self.wires[1], self.wires[2], [self.wires[0]])
        return [Prim('gen', self.matrix[4:8, 4:8], self._bits(self.wires[1:]), self._bits([self.wires[0]]))]


# ======================================================================================================
# arbitrary gates
# ======================================================================================================
class UAnyGate(ArbitraryGate):
    """User-supplied unitary on ``wires`` / ``minmax`` (reference: gate.py:2745-2788)."""

    def __init__(self, unitary, nqubit=1, wires=None, minmax=None, controls=None, name='UAnyGate', den_mat=False,
                 tsr_mode=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, minmax=minmax, controls=controls, den_mat=den_mat,
                         tsr_mode=tsr_mode)
        if not isinstance(unitary, torch.Tensor):
            unitary = torch.tensor(unitary, dtype=torch.cfloat).reshape(-1, 2 ** len(self.wires))
        assert unitary.dtype in (torch.cfloat, torch.cdouble)
        assert unitary.shape[-1] == unitary.shape[-2] == 2 ** len(self.wires)
        eye = torch.eye(unitary.shape[-1], dtype=unitary.dtype, device=unitary.device)
        assert torch.allclose(unitary @ unitary.mH, eye, rtol=1e-5, atol=1e-4), 'Please check the unitary matrix'
        # the reference accepts 1e-4; the adjoint method's reverse sweep must then tell U^-1 from U^dagger (executor)
        self._exact_unitary = bool(torch.allclose(unitary @ unitary.mH, eye, rtol=0.0, atol=2e-6))
        self.register_buffer('matrix', unitary)
        self._kind_cache: tuple | None = None

    @property
    def _kernel_kind(self) -> str:
        """'diag' when the CURRENT matrix is exactly diagonal (the diagonal kernels ignore off-diagonal entries),
        re-derived whenever the buffer is replaced or written in place (``load_state_dict``, ``.to``): the test
        reads the matrix values, so its result is cached on the buffer's identity and version counter."""
        m = self.matrix
        ver = tensor_version(m)
        key = (m.data_ptr(), ver, m.device, m.dtype)
        if ver is None or self._kind_cache is None or self._kind_cache[0] != key:
            diag_only = m.ndim == 2 and len(self.wires) <= 2 and bool(
                (m == torch.diag_embed(m.diagonal(dim1=-2, dim2=-1))).all())
            self._kind_cache = (key, 'diag' if diag_only else 'gen')
        return self._kind_cache[1]

    def update_matrix(self) -> torch.Tensor:
        return self.matrix.mH if self.inv_mode else self.matrix


class LatentGate(ArbitraryGate):
    """Unitary obtained from the polar factor (u @ vh of the SVD) of a trainable latent matrix
    (reference: gate.py:2791-2864, matrix :2836-2840)."""

    def __init__(self, inputs=None, nqubit=1, wires=None, minmax=None, controls=None, name='LatentGate',
                 den_mat=False, tsr_mode=False, requires_grad=False):
        super().__init__(name=name, nqubit=nqubit, wires=wires, minmax=minmax, controls=controls, den_mat=den_mat,
                         tsr_mode=tsr_mode)
        self.requires_grad = requires_grad
        self.init_para(inputs)

    def inputs_to_tensor(self, inputs: Any = None) -> torch.Tensor:
        d = 2 ** len(self.wires)
        if inputs is None:
            return torch.randn(d, d)
        if not isinstance(inputs, (torch.Tensor, nn.Parameter)):
            inputs = torch.tensor(inputs, dtype=torch.float)
        return inputs.reshape(d, d)

    def get_matrix(self, inputs: Any) -> torch.Tensor:
        latent = self.inputs_to_tensor(inputs) + 0j
        u, _, vh = torch.linalg.svd(latent)
        return u @ vh

    def update_matrix(self) -> torch.Tensor:
        latent = self.latent.mH if self.inv_mode else self.latent
        matrix = self.get_matrix(latent)
        self.matrix = matrix.detach()
        return matrix

    def get_derivative(self, latent: Any) -> torch.Tensor:
        latent = self.inputs_to_tensor(latent)
        du = jacobian(self._real_wrapper, latent).permute(3, 4, 0, 1, 2)
        return du[..., 0] + du[..., 1] * 1j

    def init_para(self, inputs: Any = None) -> None:
        latent = self.inputs_to_tensor(inputs)
        if self.requires_grad:
            self.latent = nn.Parameter(latent)
        else:
            self.register_buffer('latent', latent)
        self.update_matrix()
        self.npara = self.latent.numel()


class HamiltonianGate(ArbitraryGate):
    """exp(-i H t).  ``hamiltonian`` is a list such as ``[[0.5, 'x0y1'], [-1, 'z3y1']]`` (coefficient,
    Pauli letters each followed by its wire) or a dense tensor on ``wires`` / ``minmax``
    (reference: gate.py:2867-3024, matrix :2990-2994)."""

    def __init__(self, hamiltonian, t=None, nqubit=1, wires=None, minmax=None, controls=None, name='HamiltonianGate',
                 den_mat=False, tsr_mode=False, requires_grad=False):
        self.nqubit = nqubit
        self.ham_lst = None
        if isinstance(hamiltonian, list):
            self.ham_lst = self._terms(hamiltonian)
            wires = None
            minmax = self.get_minmax(hamiltonian)
        super().__init__(name=name, nqubit=nqubit, wires=wires, minmax=minmax, controls=controls, den_mat=den_mat,
                         tsr_mode=tsr_mode)
        self.npara = 1
        self.requires_grad = requires_grad
        self.register_buffer('ham_tsr', self._ham_matrix(hamiltonian))
        self.init_para([None, t])

    def _apply(self, fn: Any, *args, **kwargs):
        from .utils import complex_apply

        held = {'ham_tsr': self._buffers.pop('ham_tsr')}
        nn.Module._apply(self, fn, *args, **kwargs)
        for key, value in complex_apply(fn, held).items():
            self.register_buffer(key, value)
        return self

    @staticmethod
    def _terms(ham: list) -> list[tuple[float, list[tuple[str, int]]]]:
        import re

        if len(ham) == 2 and isinstance(ham[1], str):
            ham = [ham]
        out = []
        for pair in ham:
            assert isinstance(pair, list) and isinstance(pair[1], str), 'Invalid input type'
            factors = [(p.lower(), int(w)) for p, w in re.findall(r'([xyzXYZ])(\d+)', pair[1])]
            out.append((pair[0], factors))
        return out

    def get_minmax(self, hamiltonian: list) -> list[int]:
        wires = [w for _, fs in self._terms(hamiltonian) for _, w in fs]
        return [min(wires), max(wires)]

    def _ham_matrix(self, ham: Any) -> torch.Tensor:
        if not isinstance(ham, list):
            ham = ham if isinstance(ham, torch.Tensor) else torch.tensor(ham, dtype=torch.cfloat)
            return ham if ham.is_complex() else ham + 0j
        paulis = {
            'x': torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat),
            'y': torch.tensor([[0, -1j], [1j, 0]]),
            'z': torch.tensor([[1, 0], [0, -1]], dtype=torch.cfloat),
        }
        lo, hi = self.minmax
        total = None
        for coef, factors in self._terms(ham):
            lst = [torch.eye(2, dtype=torch.cfloat)] * (hi - lo + 1)
            for p, w in factors:
                lst[w - lo] = paulis[p]
            term = multi_kron(lst) * coef
            total = term if total is None else total + term
        return total

    def inputs_to_tensor(self, inputs: Any = None):
        if inputs is None:
            return self.ham_tsr, torch.rand(1)[0]
        if isinstance(inputs, (list, tuple)) and len(inputs) == 2 and not isinstance(inputs[0], (int, float)):
            ham, t = inputs
        else:
            ham, t = None, inputs
        ham_tsr = self.ham_tsr if ham is None else self._ham_matrix(ham).to(self.ham_tsr)
        while isinstance(t, list):
            t = t[0]
        if t is None:
            t = torch.rand(1)[0]
        return ham_tsr, _as_param_tensor(t)

    def get_matrix(self, hamiltonian: Any, t: Any) -> torch.Tensor:
        ham, t = self.inputs_to_tensor([hamiltonian, t])
        t = _prep(t)
        if _is_batched(t):
            return torch.linalg.matrix_exp(-1j * ham * t.reshape(-1, 1, 1))
        return torch.linalg.matrix_exp(-1j * ham * t)

    def update_matrix(self) -> torch.Tensor:
        t = -self.t if self.inv_mode else self.t
        matrix = self.get_matrix(self.ham_tsr, t)
        assert matrix.shape[-1] == matrix.shape[-2] == 2 ** len(self.wires)
        self.matrix = matrix.detach()
        return matrix

    def _real_wrapper(self, x: Any) -> torch.Tensor:
        return torch.view_as_real(self.get_matrix(self.ham_tsr, x))

    def get_derivative(self, t: Any) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.tensor(t, dtype=torch.float)
        du = jacobian(self._real_wrapper, t.squeeze())
        return du[..., 0] + du[..., 1] * 1j

    def init_para(self, inputs: Any = None) -> None:
        ham, t = self.inputs_to_tensor(inputs)
        self.register_buffer('ham_tsr', ham)
        if self.requires_grad:
            self.t = nn.Parameter(t)
        else:
            self.register_buffer('t', t)
        self.update_matrix()


class Reset(Gate):
    r"""Reset qubits to :math:`|0\rangle` (reference: gate.py:3027-3094).

    ``postselect`` in (0, 1): project every wire on that outcome (or on the other one if its probability
    is exactly zero), renormalise and move the amplitude to :math:`|0\rangle` -- per wire one marginal
    reduction and one (non-unitary, per-sample) 2x2 gate launch.  ``postselect=None``: sample the joint
    outcome of the wires, project, renormalise, relabel to :math:`|0..0\rangle`.  The matrices depend on the
    state, so a circuit containing a ``Reset`` runs as separate fused stretches around it."""

    _state_dependent = True

    def __init__(self, nqubit=1, wires=None, postselect=0, tsr_mode=False):
        if wires is None:
            wires = list(range(nqubit))
        super().__init__(name='Reset', nqubit=nqubit, wires=wires, tsr_mode=tsr_mode)
        self.postselect = postselect

    def prims(self, decompose: bool = True) -> list[Prim]:
        raise NotImplementedError('Reset depends on the state: it has no fixed matrix / unitary')

    def apply_flat(self, x: torch.Tensor) -> torch.Tensor:
        """(B, 2**n) -> (B, 2**n)."""
        from . import ops

        n = self.nqubit
        if len(self.wires) == n:
            out = torch.zeros_like(x)
            out[:, 0] = 1
            return out
        if self.postselect in (0, 1):
            ps = self.postselect
            for wire in self.wires:
                bit = n - 1 - wire
                probs = ops.marginal(x, [bit])                                  # (B, 2)
                mask = 1 - torch.sign(probs[:, ps])
                norm = torch.sqrt(probs[:, ps] + mask)
                keep, other = (1 - mask) / norm, mask / norm
                zero = torch.zeros_like(keep)
                row0 = torch.stack([keep, other] if ps == 0 else [other, keep], dim=-1)
                mats = torch.stack([row0, torch.stack([zero, zero], dim=-1)], dim=-2) + 0j   # (B, 2, 2)
                x = ops.apply_gate(x, mats, [bit], [])
            return x
        assert self.postselect is None, 'postselect must be 0, 1 or None'
        wires = sorted(self.wires)
        bits = [n - 1 - w for w in wires]
        probs = ops.marginal(x, bits)                                          # (B, 2**k)
        sample = torch.multinomial(probs.detach().clamp_min(0), 1)             # (B, 1)
        amp = torch.gather(probs, 1, sample) ** -0.5                           # (B, 1)
        d = probs.shape[-1]
        mats = torch.zeros(x.shape[0], d, d, dtype=probs.dtype, device=x.device)
        mats = mats.index_put((torch.arange(x.shape[0], device=x.device), torch.zeros_like(sample[:, 0]),
                               sample[:, 0]), amp[:, 0]) + 0j
        return ops.apply_gate(x, mats, bits, [])

    def op_state(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        return self.apply_flat(x.reshape(shape[0], -1)).reshape(shape)

    def op_dist_state(self, x):
        raise NotImplementedError('Reset on a sharded state is not supported (nor is it in the reference)')

    def get_unitary(self) -> torch.Tensor:
        raise NotImplementedError('Reset is not unitary')


class Barrier(Gate):
    """No-op separator (reference: gate.py:3097-3126)."""

    def __init__(self, nqubit=1, wires=None, name='Barrier'):
        if wires is None:
            wires = list(range(nqubit))
        super().__init__(name=name, nqubit=nqubit, wires=wires)

    def prims(self, decompose: bool = True) -> list[Prim]:
        return []

    def forward(self, x: Any) -> Any:
        return x

    def get_unitary(self) -> torch.Tensor:
        return torch.eye(2**self.nqubit, dtype=torch.cfloat)
