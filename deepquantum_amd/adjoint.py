"""Adjoint differentiation of ``Re <psi|O|psi>`` on the sharded state: memory stays at three state
vectors regardless of depth (reference: adjoint.py:19-83, after arXiv:2009.02823).

Per gate of the reverse sweep: ``phi <- U^dagger phi``; for every trainable parameter
``grad = 2 Re <lambda| dU/dtheta |phi>``; ``lambda <- U^dagger lambda``.  When the gate's targets are
local the bracket is evaluated without materialising ``dU phi``: with
``G[i, j] = sum_groups lambda_i conj(phi_j)`` over the controlled amplitude groups (one pass of the
gate-gradient kernel over both states), ``<lambda| dU |phi> = sum_ij dU[i, j] conj(G[i, j])``.  A gate
with a global target falls back to building ``mu = dU phi`` through the exchange path.
"""

from __future__ import annotations

from copy import deepcopy

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import backend
from .distributed import (
    _many_target_global, _one_target_global, _rank_controls_ok, dist_apply_prims, inner_product_dist,
)
from .executor import Prim
from .gate import CombinedSingleGate
from .state import DistributedQubitState


def _flatten_gates(operators) -> list:
    gates = []
    for op in operators:
        gates.extend(op.gates if isinstance(op, CombinedSingleGate) else [op])
    return gates


def _gate_parameters(gate) -> torch.Tensor:
    tensors = list(gate.parameters()) if gate.requires_grad else list(gate.buffers())
    # buffers() of a parametric gate are exactly its parameters (the matrix is a plain attribute)
    return torch.stack([t.reshape(()) if t.numel() == 1 else t for t in tensors]).squeeze(0)


def _bracket(lam: DistributedQubitState, phi: DistributedQubitState, gate, dmat: torch.Tensor) -> torch.Tensor:
    """<lambda| (dmat on the gate's wires, zero outside the controlled subspace) |phi>, all-reduced."""
    n, L = gate.nqubit, phi.log_num_amps_per_node
    targets = [n - 1 - w for w in gate.wires]
    controls = [n - 1 - w for w in gate.controls]
    if all(t < L for t in targets):
        if _rank_controls_ok(phi, controls):
            lc = [c for c in controls if c < L]
            g = backend.gate_grad(phi.amps.view(1, -1), lam.amps.view(1, -1), targets, lc)[0]
            val = (dmat.to(g.dtype) * g.conj()).sum()
        else:
            val = torch.zeros((), dtype=torch.complex128, device=phi.amps.device)
    else:
        # a target lives on a global qubit: build mu = dU phi through the exchange path
        mu = deepcopy(phi)
        prim = Prim('gen', dmat.to(mu.amps.dtype), tuple(targets), tuple(controls))
        if len(targets) == 1:
            _one_target_global(mu, prim, derivative=True)
        else:
            _many_target_global(mu, prim)
            # dU acts as ZERO (not identity) outside the controlled subspace
            if not _rank_controls_ok(mu, controls):
                mu.amps.zero_()
            else:
                mask = 0
                for c in controls:
                    if c < L:
                        mask |= 1 << c
                if mask:
                    keep = backend.pack(mu.amps.view(1, -1), mask, mask)
                    mu.amps.zero_()
                    backend.unpack_axpby(mu.amps.view(1, -1), keep, None, None, mask, mask)
        val = backend.inner(lam.amps.view(1, -1), mu.amps.view(1, -1))[0]
    if phi.world_size > 1:
        buf = torch.view_as_real(val.clone())
        dist.all_reduce(buf, dist.ReduceOp.SUM)
        val = torch.view_as_complex(buf)
    return val


def _sweep_fused_sharded(ctx, grad_out, params):
    """The reverse sweep as ONE gate list on the (psi, lambda) pair: the two shards interleaved along an extra lowest
    index bit form the shard of an (n + 1)-qubit sharded state, every gate's inverse acts on both halves at once, and in
    front of every trainable gate a 'grad' primitive reduces G' = sum lambda (x) conj(psi) on the gate's target inside the
    fused pass that holds the qubit (DQ_FG_GRAD, include/dq_hip.h; executor._AdjointCircuit._sweep_fused on one GPU).
    The list goes through the same machinery as a forward circuit -- commutation-DAG order, remaps that bring a target
    from the rank bits (both halves move together), local stretches as fused passes -- so the sweep costs a few passes
    and a handful of exchanges instead of two kernel launches and up to two exchanges per gate.  The partial sums of the
    ranks are all-reduced once at the end;  <lambda| dU |phi> = sum dU[i, j] conj(G[i, j]),  G = G' U^-dagger
    (phi = U^-1 psi).  Returns the gradients (list, forward parameter order) or None: not applicable -- a trainable
    gate on two or more targets, batched shards, shards smaller than a tile."""
    from . import distributed as D
    from . import executor, fusion

    phi, lam = ctx.state_phi, ctx.state_lambda
    if not (executor.CONFIG['fused_sweep'] and executor.CONFIG['fuse']) or phi.batch is not None:
        return None
    is128 = phi.amps.dtype == torch.complex128
    geom = executor._geometry(is128)
    if phi.log_num_amps_per_node + 1 < geom.m:
        return None
    gates = list(reversed(_flatten_gates(ctx.operators)))
    prims: list[Prim] = []
    todo = []                  # (row, gate, parameter, exact inverse matrix) per gradient that is wanted
    idx = 1
    slots: list = []           # per parametrised gate, in sweep order: index into `todo` or None
    nrows = 0                  # accumulator rows handed out so far
    for gate in gates:
        inv_prims = gate.inverse().prims()
        if gate.npara > 0:
            p = params[-idx]
            if ctx.needs_input_grad[3 + len(params) - idx]:
                if (len(inv_prims) != 1 or len(inv_prims[0].targets) > 2 or inv_prims[0].kind not in ('gen', 'diag')
                        or inv_prims[0].matrix.ndim != 2):
                    return None
                ip = inv_prims[0]
                slots.append(len(todo))
                # (a gate on two targets -- Rxx, Ryy, Rzz, Rxy of the reference's own test circuit, tests/test_circuit.py:
                # 87-139 -- takes four one-target records: executor.grad_records)
                # (only the sums the gate's derivative can read -- a real / a I + i b X / diagonal matrix has a derivative of
                # the same form -- and ONE sum for a rotation about X: nothing differentiates these brackets again)
                recs, cnt = executor.grad_records(ip.kind, ip.mode, tuple(t + 1 for t in ip.targets),
                                                  tuple(c + 1 for c in ip.controls), nrows, terminal=ip.exact is True)
                prims.extend(recs)
                todo.append((gate, p, ip.matrix, nrows, ip.kind, len(ip.targets)))
                nrows += cnt
            else:
                slots.append(None)
            idx += 1
        for ip in inv_prims:
            prims.append(Prim(ip.kind, ip.matrix, tuple(t + 1 for t in ip.targets), tuple(c + 1 for c in ip.controls), ip.mode))
    if not todo:
        return [None] * len(slots)
    ops = [fusion.PrimOp(q.kind, q.targets, q.controls, 0, q.mode) for q in prims]
    if any(len(q.targets) > 2 for q in prims):
        return None
    # the pair as a sharded state of n + 1 qubits: bit 0 tells psi from lambda, the rank bits are the same
    from .state import DistributedQubitState

    with torch.no_grad():
        work = backend.interleave(phi.amps.reshape(1, -1), lam.amps.reshape(1, -1)).reshape(-1)
        empty = work.new_zeros(0)
        phi.amps = phi.buffer = lam.amps = lam.buffer = empty        # (three states regardless of depth, not five)
        phi._shape = lam._shape = (0,)
        lazy = DistributedQubitState.LAZY_AMPS
        DistributedQubitState.LAZY_AMPS = -1                         # (do not build |0..0> first)
        try:
            pair = DistributedQubitState(phi.nqubit + 1, dtype=work.dtype)
        finally:
            DistributedQubitState.LAZY_AMPS = lazy
        pair.amps, pair.buffer = work, torch.empty_like(work)
        acc = torch.zeros(1, nrows, 8, dtype=torch.float64, device=work.device)
        D._SWEEP['grads'] = acc
        try:
            D.dist_apply_prims(pair, prims, mode='remap', keep_layout=True, force_mode=True)
        finally:
            D._SWEEP['grads'] = None
        stats = dict(D.LAST_RUN)
        if pair.world_size > 1:
            dist.all_reduce(acc, dist.ReduceOp.SUM)
        gsum = torch.view_as_complex(acc.reshape(-1, 4, 2)).reshape(-1, 2, 2)       # G' per row
    LAST_SWEEP.update(fused=True, rows=nrows, remaps=stats.get('remaps', 0), local_flushes=stats.get('local_flushes', 0))
    vals = []
    for gate, p, inv, row0, kind, ntargets in todo:
        with torch.enable_grad():
            du = gate.get_derivative(p.detach())
        du = du.unsqueeze(0).flatten(0, -3).to(torch.complex128)                     # (npara, D, D)
        g = executor.assemble_grad_sums(gsum, row0, kind, ntargets) @ inv.to(torch.complex128).mH       # G = G' U^-dagger
        brackets = (du * g.conj()).sum(dim=(-2, -1))
        vals.append((grad_out * 2 * brackets.real.to(grad_out.dtype)).reshape(p.shape))
    return [None if sl is None else vals[sl] for sl in slots]


#: what the last backward of ``AdjointExpectation`` did (tests / bench)
LAST_SWEEP: dict = {'fused': False}


class AdjointExpectation(Function):
    @staticmethod
    def forward(ctx, state: DistributedQubitState, operators, observable, *parameters: torch.Tensor) -> torch.Tensor:
        ctx.state_phi = state
        ctx.operators = operators
        ctx.state_lambda = deepcopy(state)
        dist_apply_prims(ctx.state_lambda, observable.prims())
        ctx.save_for_backward(*parameters)
        return inner_product_dist(ctx.state_lambda, ctx.state_phi).real

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        return (None, None, None, *_reverse_sweep(ctx, grad_out, list(ctx.saved_tensors)))


def _reverse_sweep(ctx, grad_out: torch.Tensor, params: list) -> list:
    """Gradients w.r.t. ``params`` (forward order) from ``ctx.state_phi`` (the final state) and ``ctx.state_lambda``
    (the observable applied to it): fused passes on the pair where that applies, the reference's gate-by-gate sweep
    (adjoint.py:42-83) otherwise."""
    phi, lam = ctx.state_phi, ctx.state_lambda
    LAST_SWEEP.clear()
    LAST_SWEEP['fused'] = False
    fused = _sweep_fused_sharded(ctx, grad_out, params)
    if fused is not None:
        return fused[::-1]
    grads: list = []
    idx = 1
    with torch.no_grad():
        for gate in reversed(_flatten_gates(ctx.operators)):
            inv_prims = gate.inverse().prims()
            dist_apply_prims(phi, inv_prims)
            if gate.npara > 0:
                p = params[-idx]
                if ctx.needs_input_grad[3 + len(params) - idx]:
                    with torch.enable_grad():
                        du = gate.get_derivative(p.detach())
                    du = du.unsqueeze(0).flatten(0, -3)  # (npara, D, D)
                    vals = [grad_out * 2 * _bracket(lam, phi, gate, d).real.to(grad_out.dtype) for d in du]
                    grads.append(torch.stack(vals).reshape(p.shape))
                else:
                    grads.append(None)
                idx += 1
            dist_apply_prims(lam, inv_prims)
    return grads[::-1]


class AdjointExpectations(Function):
    """ALL observables of a circuit as one node: the values from one read of the shards each (no copy), and in the
    backward ONE reverse sweep with  lambda = sum_k g_k O_k psi  (the cost is linear in the observables; g = the incoming
    gradient).  The reference -- and ``adjoint_expectation`` -- build one (psi, lambda) pair and run one sweep PER
    observable (circuit.py:1706-1738): for the 34 ZZ terms of a QAOA ring on 34 qubits that is 34 sweeps and 68 shards
    held between forward and backward; here it is one sweep and three shards whatever the number of observables."""

    @staticmethod
    def forward(ctx, state: DistributedQubitState, operators, observables, *parameters: torch.Tensor) -> torch.Tensor:
        from .distributed import expect_pauli_dist

        ctx.state_phi = state
        ctx.operators = operators
        ctx.observables = observables
        ctx.save_for_backward(*parameters)
        return torch.stack([expect_pauli_dist(state, ob) for ob in observables], dim=-1)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        phi = ctx.state_phi
        with torch.no_grad():
            lam = None
            for k, ob in enumerate(ctx.observables):
                tmp = deepcopy(phi)
                dist_apply_prims(tmp, ob.prims())
                gk = grad_out[..., k].to(tmp.amps.real.dtype)
                if lam is None:
                    lam = tmp
                    lam.amps.mul_(gk)
                else:
                    lam.amps.add_(tmp.amps * gk)
                del tmp
        ctx.state_lambda = lam
        one = torch.ones((), dtype=grad_out.dtype, device=grad_out.device)
        return (None, None, None, *_reverse_sweep(ctx, one, list(ctx.saved_tensors)))


def adjoint_expectations(state: DistributedQubitState, operators, observables) -> torch.Tensor:
    """Differentiable expectation values of all ``observables`` on the sharded state, shape (len(observables),)."""
    parameters = [_gate_parameters(g) for g in _flatten_gates(operators) if g.npara > 0]
    work = deepcopy(state)
    return AdjointExpectations.apply(work, operators, list(observables), *parameters)


def adjoint_expectation(state: DistributedQubitState, operators, observable) -> torch.Tensor:
    """Differentiable expectation on the sharded state (reference: circuit.py:1706-1738)."""
    parameters = [_gate_parameters(g) for g in _flatten_gates(operators) if g.npara > 0]
    work = deepcopy(state)
    return AdjointExpectation.apply(work, operators, observable, *parameters)
