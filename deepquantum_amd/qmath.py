"""Math entry points of the statevector path, API-compatible with the reference's qmath.py for the
functions the QubitCircuit hot path uses.  All heavy lifting is delegated to the HIP kernels through
``ops`` / ``backend``; only shape bookkeeping happens here."""

from __future__ import annotations

from collections import Counter
from typing import TYPE_CHECKING, Any

import torch
from torch import nn

from . import backend, ops

if TYPE_CHECKING:
    from .layer import Observable


def is_power_of_two(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def inverse_permutation(permute_shape: list[int]) -> list[int]:
    """``inv`` with ``inv[permute_shape[i]] = i`` (reference: qmath.py:84-94)."""
    inv = [0] * len(permute_shape)
    for i, p in enumerate(permute_shape):
        inv[p] = i
    return inv


def amplitude_encoding(data: Any, nqubit: int) -> torch.Tensor:
    """Normalised state(s) of ``nqubit`` qubits from raw amplitudes, zero padded or truncated
    (reference: qmath.py:456-482).  Returns (batch, 2**n, 1)."""
    if not isinstance(data, (torch.Tensor, nn.Parameter)):
        data = torch.tensor(data)
    single = data.ndim == 1 or (data.ndim == 2 and data.shape[-1] == 1)
    batch = 1 if single else data.shape[0]
    data = data.reshape(batch, -1)
    dim = 2**nqubit
    state = torch.zeros(batch, dim, dtype=data.dtype, device=data.device) + 0j
    data = nn.functional.normalize(data[:, :dim], p=2, dim=-1)
    width = min(dim, data.shape[1])
    state[:, :width] = data[:, :width]
    return state.unsqueeze(-1)


def evolve_state(state: torch.Tensor, matrix: torch.Tensor, nqudit: int, wires: list[int], qudit: int = 2) -> torch.Tensor:
    """``(U on wires) psi`` for a tensor-form state (batch, 2, ..., 2); returns the same shape.

    Drop-in for the reference seam qmath.evolve_state (qmath.py:485-506); the permute / reshape /
    matmul chain there becomes one gate kernel.  Only qubits (qudit = 2) are on this path."""
    if qudit != 2:
        raise NotImplementedError('deepquantum_amd accelerates the qubit path only (qudit == 2)')
    shape = state.shape
    flat = state.reshape(shape[0], -1)
    bits = [nqudit - 1 - w for w in wires]
    return ops.apply_gate(flat, matrix, bits, []).reshape(shape)


def multi_kron(lst: list[torch.Tensor]) -> torch.Tensor:
    """Kronecker product of a list, balanced tree order (reference: qmath.py:390-405)."""
    if len(lst) == 1:
        return lst[0].contiguous()
    mid = len(lst) // 2
    return torch.kron(multi_kron(lst[:mid]), multi_kron(lst[mid:])).contiguous()


def slice_state_vector(state: torch.Tensor, nqubit: int, wires: list[int], bits: str, normalize: bool = True) -> torch.Tensor:
    """Project ``wires`` onto ``bits`` and return the remaining (batch, 2**(n-len)) amplitudes
    (reference: qmath.py:365-387)."""
    if len(bits) == 1:
        bits = bits * len(wires)
    assert len(wires) == len(bits)
    mask = value = 0
    for w, b in zip(wires, bits, strict=True):
        assert b in '01'
        mask |= 1 << (nqubit - 1 - w)
        value |= int(b) << (nqubit - 1 - w)
    flat = state.reshape(-1, 2**nqubit)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    out = backend.pack(flat, mask, value) if not flat.requires_grad else _slice_autograd(flat, nqubit, wires, bits)
    if normalize:
        out = nn.functional.normalize(out, p=2, dim=-1)
    return out


def _slice_autograd(flat: torch.Tensor, nqubit: int, wires: list[int], bits: str) -> torch.Tensor:
    x = flat.reshape([-1] + [2] * nqubit)
    axes = [w + 1 for w in wires]
    pm = axes + [i for i in range(nqubit + 1) if i not in axes]
    x = x.permute(pm)
    for b in bits:
        x = x[int(b)]
    return x.reshape(flat.shape[0], -1)


def block_sample(probs: torch.Tensor, shots: int = 1024, block_size: int = 2**24) -> list:
    """Two-level multinomial sampling over blocks of ``block_size`` outcomes so that
    ``torch.multinomial`` never sees more than 2**24 categories (reference: qmath.py:543-565)."""
    nblocks = -(-len(probs) // block_size)
    if nblocks == 1:
        return torch.multinomial(probs, shots, replacement=True).cpu().numpy().tolist()
    pad = nblocks * block_size - len(probs)
    padded = torch.cat([probs, probs.new_zeros(pad)]) if pad else probs
    block_p = padded.reshape(nblocks, block_size).sum(1)
    picks = Counter(torch.multinomial(block_p, shots, replacement=True).cpu().numpy().tolist())
    samples: list = []
    for blk, cnt in picks.items():
        lo = blk * block_size
        hi = min(lo + block_size, len(probs))
        sub = torch.multinomial(probs[lo:hi], cnt, replacement=True) + lo
        samples.extend(sub.cpu().numpy().tolist())
    return samples


def is_density_matrix(rho: torch.Tensor) -> bool:
    """Hermitian, unit trace, positive semi-definite; 2-D or batched 3-D (reference: qmath.py:117-149).
    The spectrum is computed on the host (Hermitian eigensolver) with a small tolerance for round-off."""
    if not isinstance(rho, torch.Tensor) or rho.ndim not in (2, 3):
        return False
    if not is_power_of_two(rho.shape[-2]) or not is_power_of_two(rho.shape[-1]) or rho.shape[-1] != rho.shape[-2]:
        return False
    if rho.ndim == 2:
        rho = rho.unsqueeze(0)
    rho = rho.detach()
    if not torch.allclose(rho, rho.mH):
        return False
    trace = rho.diagonal(dim1=-2, dim2=-1).sum(-1)
    if not torch.allclose(trace, torch.ones_like(trace)):
        return False
    eig = torch.linalg.eigvalsh(rho.cpu())
    return bool((eig >= -1e-6).all())


def partial_trace(rho: torch.Tensor, nqudit: int, trace_lst: list[int], qudit: int = 2) -> torch.Tensor:
    """Trace out the qudits in ``trace_lst`` of (batch, d^n, d^n) density matrices
    (reference: qmath.py:408-436)."""
    if rho.ndim == 2:
        rho = rho.unsqueeze(0)
    assert rho.ndim == 3 and rho.shape[1] == rho.shape[2] == qudit**nqudit
    b = rho.shape[0]
    keep = [i for i in range(nqudit) if i not in trace_lst]
    letters = list(range(1, 2 * nqudit + 1))            # einsum index ids: rows 1..n, columns n+1..2n
    for i in trace_lst:
        letters[nqudit + i] = letters[i]                # repeated index = traced
    out = [letters[i] for i in keep] + [letters[nqudit + i] for i in keep]
    red = torch.einsum(rho.reshape([b] + [qudit] * 2 * nqudit), [0] + letters, [0] + out)
    d = qudit ** len(keep)
    return red.reshape(b, d, d).squeeze(0)


def evolve_den_mat(state: torch.Tensor, matrix: torch.Tensor, nqudit: int, wires: list[int], qudit: int = 2) -> torch.Tensor:
    """rho -> U rho U^dagger for U on ``wires`` of a (batch, 2, ..., 2) tensor with 2n qubit axes
    (reference: qmath.py:509-540): the gate kernel on the row bits, its conjugate on the column bits."""
    if qudit != 2:
        raise NotImplementedError('deepquantum_amd: qudits with d != 2 belong to the photonic path')
    from . import executor
    from .operation import lift_to_density_matrix

    shape = state.shape
    prim = executor.Prim('gen', matrix, tuple(nqudit - 1 - w for w in wires), ())
    out = executor.run(state.reshape(shape[0], -1), lift_to_density_matrix([prim], nqudit))
    return out.reshape(shape)


def _parity(x: torch.Tensor) -> torch.Tensor:
    for shift in (32, 16, 8, 4, 2, 1):
        x = x ^ (x >> shift)
    return x & 1


def measure(
    state: torch.Tensor,
    shots: int = 1024,
    with_prob: bool = False,
    wires: int | list[int] | None = None,
    den_mat: bool = False,
    block_size: int = 2**24,
) -> dict | list[dict]:
    """Sample bit strings from |psi|^2 (reference: qmath.py:568-638).  Probabilities and marginals are
    computed by the HIP reduction kernels; sampling stays ``torch.multinomial`` on the device."""
    if den_mat:
        assert is_density_matrix(state), 'Please input density matrices'
        state = state.diagonal(dim1=-2, dim2=-1)
    single = state.ndim == 1 or (state.ndim == 2 and state.shape[-1] == 1)
    batch = 1 if single else state.shape[0]
    flat = state.reshape(batch, -1)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    dim = flat.shape[-1]
    assert is_power_of_two(dim), 'The length of the quantum state is not in the form of 2^n'
    n = dim.bit_length() - 1
    if wires is not None:
        if isinstance(wires, int):
            wires = [wires]
        wires = sorted(wires)
    nbits = len(wires) if wires else n
    with torch.no_grad():
        if den_mat:                                   # the diagonal of rho already holds the probabilities
            all_probs = torch.abs(flat)
            if wires is not None and len(wires) != n:
                axes = [w + 1 for w in wires]
                pm = [0] + axes + [i for i in range(1, n + 1) if i not in axes]
                all_probs = all_probs.reshape([batch] + [2] * n).permute(pm).reshape(batch, 2 ** len(wires), -1).sum(-1)
        elif wires is None or len(wires) == n:
            all_probs = backend.probs(flat)
        else:           # one read of the state whatever the number of wires (dq_marginal_*)
            all_probs = backend.marginal(flat, [n - 1 - w for w in wires]).to(flat.real.dtype)
    results = []
    for i in range(batch):
        probs = all_probs[i]
        counts = Counter(block_sample(probs, shots, block_size))
        res = {bin(k)[2:].zfill(nbits): v for k, v in counts.items()}
        if with_prob:
            for k in res:
                res[k] = res[k], probs[int(k, 2)]
        results.append(res)
    return results[0] if batch == 1 else results


def expectation(state: torch.Tensor, observable: 'Observable', den_mat: bool = False, chi: int | None = None) -> torch.Tensor:
    """``Re <psi| O |psi>`` for a Pauli-string observable (reference: qmath.py:830-860), computed in a
    single pass over the state by the Pauli-expectation kernel."""
    if isinstance(state, list):
        raise NotImplementedError('matrix product states are outside the accelerated path')
    if den_mat:
        # Tr(P rho) = sum_j P[j ^ x, j] rho[j, j ^ x]: only 2^n of the 4^n entries of rho are needed
        # (the reference multiplies the full 2^n x 2^n observable into rho, qmath.py:855)
        single = state.ndim == 2
        rho = state.reshape(1 if single else state.shape[0], -1)
        xmask, zmask = observable.pauli_masks()
        dim = int(round(rho.shape[-1] ** 0.5))
        j = torch.arange(dim, device=rho.device)
        entries = rho[:, j * dim + (j ^ xmask)]
        sign = (1 - 2 * _parity(j & zmask)).to(rho.real.dtype)
        val = (entries * sign).sum(-1) * (1j) ** bin(xmask & zmask).count('1')
        out = val.real
        return out.squeeze(0) if single else out
    single = state.ndim == 2
    flat = state.reshape(1 if single else state.shape[0], -1)
    xmask, zmask = observable.pauli_masks()
    out = ops.expect_pauli(flat, xmask, zmask)
    return out.squeeze(0) if single else out


def sample2expval(sample: dict) -> torch.Tensor:
    """Parity expectation from measurement counts (reference: qmath.py:863-871)."""
    total = sum(sample.values())
    acc = sum(cnt * (-1) ** (bits.count('1') % 2) for bits, cnt in sample.items())
    return torch.tensor([acc / total])
