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
        params = list(ctx.saved_tensors)
        phi, lam = ctx.state_phi, ctx.state_lambda
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
        return (None, None, None, *grads[::-1])


def adjoint_expectation(state: DistributedQubitState, operators, observable) -> torch.Tensor:
    """Differentiable expectation on the sharded state (reference: circuit.py:1706-1738)."""
    parameters = [_gate_parameters(g) for g in _flatten_gates(operators) if g.npara > 0]
    work = deepcopy(state)
    return AdjointExpectation.apply(work, operators, observable, *parameters)
