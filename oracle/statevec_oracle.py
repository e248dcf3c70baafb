"""CPU oracle for the QubitCircuit statevector hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the ``deepquantum_amd`` package never does.

It restates, in plain PyTorch CPU ops, the algorithm of the reference TuringQ/deepquantum v4.5.0
(paths relative to ``/root/reference/src/deepquantum``).  The arithmetic of the reference lives in
PyTorch itself (``torch>=2.4.0`` in its pyproject.toml; torch 2.10.0+rocm7.0 CPU ops in this image),
so the restatement runs the very same ATen kernels.

Parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference in the build container,
runs it on seeded circuits and stores inputs/outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this oracle against every stored vector, bit for bit where the
op sequence is identical.
"""

from __future__ import annotations

import math
import random
from typing import Sequence

import torch


# --------------------------------------------------------------------------------------------------
# qmath.py:84-94  inverse_permutation
def inverse_permutation(permute_shape: Sequence[int]) -> list[int]:
    inv = [0] * len(permute_shape)
    for i, p in enumerate(permute_shape):
        inv[p] = i
    return inv


# qmath.py:485-506  evolve_state: permute -> reshape(2^k, -1) -> matrix @ state -> reshape -> permute^-1
def evolve_state(state: torch.Tensor, matrix: torch.Tensor, nqudit: int, wires: Sequence[int]) -> torch.Tensor:
    """``state``: (batch, 2, ..., 2) with wire i on axis i+1 (operation.py:45-55)."""
    nt = len(wires)
    axes = [w + 1 for w in wires]
    pm = [i for i in range(nqudit + 1) if i not in axes]
    pm = axes + pm
    x = state.permute(pm).reshape(2**nt, -1)
    x = (matrix @ x).reshape([2] * nt + [-1] + [2] * (nqudit - nt))
    return x.permute(inverse_permutation(pm))


# operation.py:203-219  Gate.op_state_control: only the controls = 1...1 slice is multiplied
def op_state_control(
    x: torch.Tensor, matrix: torch.Tensor, nqubit: int, wires: Sequence[int], controls: Sequence[int]
) -> torch.Tensor:
    nt, nc = len(wires), len(controls)
    w = [i + 1 for i in wires]
    c = [i + 1 for i in controls]
    pm = [i for i in range(nqubit + 1) if i not in w and i not in c]
    pm = w + pm + c
    x = x.permute(pm).reshape(2**nt, -1, 2**nc)
    x = torch.cat([x[:, :, :-1], (matrix @ x[:, :, -1]).unsqueeze(-1)], dim=-1)
    x = x.reshape([2] * nt + [-1] + [2] * (nqubit - nt - nc) + [2] * nc)
    return x.permute(inverse_permutation(pm))


def apply_gate_wires(
    state: torch.Tensor, matrix: torch.Tensor, nqubit: int, wires: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """Gate.op_state (operation.py:191-197) on a (batch, 2**n) state, returns (batch, 2**n).

    A batched matrix (B, D, D) is the torch.vmap case of circuit.py:232-240: sample b sees matrix b.
    """
    b = state.shape[0]
    if matrix.ndim == 3 and matrix.shape[0] > 1:
        outs = [apply_gate_wires(state[i : i + 1], matrix[i], nqubit, wires, controls) for i in range(b)]
        return torch.cat(outs, dim=0)
    if matrix.ndim == 3:
        matrix = matrix[0]
    x = state.reshape([b] + [2] * nqubit)
    if len(controls) == 0:
        x = evolve_state(x, matrix, nqubit, wires)
    else:
        x = op_state_control(x, matrix, nqubit, wires, controls)
    return x.reshape(b, -1)


def apply_gate_bits(
    state: torch.Tensor, matrix: torch.Tensor, targets: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """Same with bit positions (LSB = 0): wire = n - 1 - bit (gate.py:79, operation.py:271)."""
    n = state.shape[-1].bit_length() - 1
    return apply_gate_wires(state, matrix, n, [n - 1 - t for t in targets], [n - 1 - c for c in controls])


# --------------------------------------------------------------------------------------------------
# Gate matrices exactly as the reference builds them (float32-rounded constants, SURVEY item 3).
def fixed_matrix(name: str) -> torch.Tensor:
    """gate.py:841 (x), :916 (y), :995 (z), :1069 (h), :1143 (s), :1233 (sdg), :1303 (t), :1367 (tdg)."""
    if name == 'x':
        return torch.tensor([[0, 1], [1, 0]], dtype=torch.cfloat)
    if name == 'y':
        return torch.tensor([[0, -1j], [1j, 0]])
    if name == 'z':
        return torch.tensor([[1, 0], [0, -1]], dtype=torch.cfloat)
    if name == 'h':
        return torch.tensor([[1, 1], [1, -1]], dtype=torch.cfloat) / 2**0.5
    if name == 's':
        return torch.tensor([[1, 0], [0, 1j]])
    if name == 'sdg':
        return torch.tensor([[1, 0], [0, -1j]])
    if name == 't':
        return torch.tensor([[1, 0], [0, (1 + 1j) / 2**0.5]])
    if name == 'tdg':
        return torch.tensor([[1, 0], [0, (1 - 1j) / 2**0.5]])
    raise KeyError(name)


def rx_matrix(theta: torch.Tensor) -> torch.Tensor:
    """gate.py:1443-1448."""
    cos = torch.cos(theta / 2) + 0j
    isin = torch.sin(theta / 2) * 1j
    return torch.stack([cos, -isin, -isin, cos]).reshape(2, 2)


def ry_matrix(theta: torch.Tensor) -> torch.Tensor:
    """gate.py:1538-1543."""
    cos = torch.cos(theta / 2)
    sin = torch.sin(theta / 2)
    return torch.stack([cos, -sin, sin, cos]).reshape(2, 2) + 0j


def rz_matrix(theta: torch.Tensor) -> torch.Tensor:
    """gate.py:1634-1639."""
    e_m_it = torch.exp(-1j * theta / 2)
    e_it = torch.exp(1j * theta / 2)
    return torch.stack([e_m_it, e_it]).reshape(-1).diag_embed().reshape(2, 2)


def theta_tensor(value: float) -> torch.Tensor:
    """ParametricSingleGate.inputs_to_tensor (gate.py:368-376): Python floats become float32."""
    return torch.tensor(value, dtype=torch.float)


# --------------------------------------------------------------------------------------------------
# qmath.py:830-860 expectation with layer.py:127-165 Observable (a layer of Pauli gates)
def expectation_pauli(state: torch.Tensor, wires: Sequence[int], basis: str) -> torch.Tensor:
    """``state``: (batch, 2**n).  Returns real (batch,) in the state's real dtype:
    ``state.mH @ observable(state)`` then ``.real``."""
    b = state.shape[0]
    n = state.shape[-1].bit_length() - 1
    x = state
    for w, p in zip(wires, basis, strict=True):
        mat = fixed_matrix(p).to(state.dtype)
        x = apply_gate_wires(x, mat, n, [w])
    col = state.reshape(b, -1, 1)
    return (col.mH @ x.reshape(b, -1, 1)).squeeze(-1).squeeze(-1).real


# qmath.py:624-626: probabilities and marginal by permute + sum
def probabilities(state: torch.Tensor, wires: Sequence[int] | None = None) -> torch.Tensor:
    b = state.shape[0]
    n = state.shape[-1].bit_length() - 1
    probs = torch.abs(state) ** 2
    if wires is None:
        return probs
    wires = sorted(wires)
    pm = [i for i in range(n) if i not in wires]
    pm = wires + pm
    out = []
    for i in range(b):
        p = probs[i].reshape([2] * n).permute(pm).reshape([2] * len(wires) + [-1]).sum(-1).reshape(-1)
        out.append(p)
    return torch.stack(out)


# --------------------------------------------------------------------------------------------------
# the seeded random H / Rx / CNOT generator of the benchmark configs.
def reset_state(state: torch.Tensor, nqubit: int, wires: Sequence[int], postselect: int = 0) -> torch.Tensor:
    """Reset.op_state for postselect in (0, 1) (gate.py:3047-3066) on a (B, 2**n) state: per wire, keep
    the postselected branch (the other one if its probability is exactly 0), renormalise, relabel to |0>."""
    b = state.shape[0]
    if len(wires) == nqubit:
        out = torch.zeros_like(state)
        out[:, 0] = 1
        return out
    x = state.reshape([b] + [2] * nqubit)
    for wire in wires:
        pm_shape = list(range(1, nqubit + 1))
        pm_shape.remove(wire + 1)
        pm_shape = [wire + 1] + pm_shape + [0]
        x = x.permute(pm_shape)
        probs = (x.abs() ** 2).sum(list(range(1, nqubit)))
        mask = 1 - torch.sign(probs[postselect])
        norm = torch.sqrt(probs[postselect] + mask)
        state0 = ((1 - mask) * x[postselect] + mask * x[1 - postselect]) / norm
        x = torch.stack([state0, torch.zeros_like(state0)])
        x = x.permute(inverse_permutation(pm_shape))
    return x.reshape(b, -1)


def random_circuit_spec(nqubit: int, depth: int, seed: int = 1234) -> list[tuple]:
    """Returns [('h', q), ('rx', q, theta), ('cnot', q, t), ...] -- n * depth gates."""
    rng = random.Random(seed)
    ops: list[tuple] = []
    for _ in range(depth):
        for q in range(nqubit):
            r = rng.random()
            if r < 1 / 3:
                ops.append(('h', q))
            elif r < 2 / 3:
                ops.append(('rx', q, rng.uniform(0, 2 * math.pi)))
            else:
                t = rng.randrange(nqubit - 1)
                t += t >= q
                ops.append(('cnot', q, t))
    return ops


def run_spec(
    nqubit: int,
    spec: Sequence[tuple],
    dtype: torch.dtype = torch.complex64,
    data: torch.Tensor | None = None,
    state: torch.Tensor | None = None,
) -> torch.Tensor:
    """Run a gate list the way QubitCircuit.forward does (circuit.py:244-263).  ``data`` (B, n_rx)
    replaces the Rx angles per batch sample (the encode/vmap case); returns (B, 2**n)."""
    real = torch.float32 if dtype == torch.complex64 else torch.float64
    b = 1 if data is None else data.shape[0]
    if state is None:
        state = torch.zeros(b, 2**nqubit, dtype=dtype)
        state[:, 0] = 1
    x = state
    h = fixed_matrix('h').to(dtype)
    cnot = (torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]]) + 0j).to(dtype)
    irx = 0
    for op in spec:
        if op[0] == 'h':
            x = apply_gate_wires(x, h, nqubit, [op[1]])
        elif op[0] == 'rx':
            if data is None:
                # reference: theta held as float32 buffer, up-cast by .to(double) (gate.py:392)
                m = rx_matrix(theta_tensor(op[2]).to(real))
            else:
                m = torch.stack([rx_matrix(data[i, irx].to(real)) for i in range(b)])
            irx += 1
            x = apply_gate_wires(x, m.to(dtype), nqubit, [op[1]])
        elif op[0] == 'cnot':
            # reference applies the 4x4 CNOT matrix on wires [control, target] through evolve_state
            # (gate.py:1934, operation.py:199-201) -- not through the controlled path
            x = apply_gate_wires(x, cnot, nqubit, [op[1], op[2]])
        else:
            raise KeyError(op[0])
    return x
