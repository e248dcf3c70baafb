"""Autograd-aware wrappers around the HIP kernels.

The reference differentiates its dense path with stock autograd through permute/reshape/matmul/cat
(SURVEY section 3C).  Here every gate application is one ``torch.autograd.Function`` whose backward is
again a gate kernel (``U^dagger`` applied to the cotangent) plus a small reduction for the matrix
gradient, so ``loss.backward()`` through ``QubitCircuit`` works unchanged while every sweep over the
state stays a single HBM-bound HIP launch.

Every backward here is itself written with differentiable operations of this module (the matrix gradient is the
``_GateGrad`` node, whose own backward is two gate applications; the cotangent of an expectation value is a gate
application), so ``create_graph=True`` -- ``torch.autograd.functional.hessian`` over a circuit, which the reference
gets from stock autograd (examples/benchmarks/gradient_benchmark.py:147-163) -- differentiates to any order.
"""

from __future__ import annotations

from typing import Sequence

import torch

from torch.autograd import forward_ad as _fwad

from . import _functorch, backend


def _is_batched(t: torch.Tensor) -> bool:
    """True inside a ``torch.vmap`` transform (the tensor is a functorch BatchedTensor)."""
    return _functorch.is_batched(t)


#: PyTorch releases the probes below were checked against (tests/test_api_cpu.py::test_functorch_probes_are_guarded names
#: them too).  They read functorch's interpreter stack through private modules; on a release where those moved, the
#: probes answer "unknown" and every caller takes its conservative route (per-gate nodes, no refusal).
FUNCTORCH_PROBES_CHECKED_ON = _functorch.CHECKED_ON


def transform_stack() -> list[str] | None:
    """The ``torch.func`` transforms this call runs under, outermost first, as 'Vmap' / 'Grad' / 'Jvp' / ... -- or None
    when this PyTorch does not let us look (the private interpreter stack moved)."""
    return _functorch.transform_stack()


def forward_ad_active() -> bool | None:
    """True inside a ``torch.autograd.forward_ad.dual_level()`` (None: cannot tell)."""
    try:
        return _fwad._current_level >= 0
    except Exception:           # noqa: BLE001
        return None


def _single_forward_level() -> None:
    """Called by every ``jvp`` rule.  Forward over forward (``jacfwd(jacfwd(f))``, a ``jvp`` inside a ``jvp``) through ANY
    ``autograd.Function`` gives silently wrong numbers in this PyTorch (2.10: the tensors a rule saved for forward mode
    lose the outer level's tangents -- a two-line Function that multiplies shows it), so it is refused by name; forward
    over reverse (``torch.func.hessian``), reverse over forward and reverse over reverse are right.  (Where the
    interpreter stack cannot be read the rule goes ahead: the refusal is a courtesy, not a correctness device.)"""
    stack = transform_stack()
    if stack is not None and sum(t == 'Jvp' for t in stack) >= 2:
        raise RuntimeError('deepquantum_amd: nested forward-mode differentiation (jacfwd(jacfwd(f)), jvp inside jvp) through '
                           'autograd.Function nodes is not reliable in this PyTorch; use torch.func.hessian(f) (forward over '
                           'reverse) or torch.func.jacrev(torch.func.jacrev(f)) instead.')


def _plain(t: torch.Tensor) -> torch.Tensor:
    """The tensor with no pending conjugation / negation (cotangents that passed through ``conj()`` or ``.real`` of a
    forward-mode rule carry them lazily): what a kernel may read through a raw pointer.  Free when there is none."""
    return t.resolve_conj().resolve_neg()


def _is_wrapped(t: torch.Tensor | None) -> bool:
    """True inside any ``torch.func`` transform (vmap, grad, vjp, jacrev, jvp ...): the tensor is a functorch wrapper -- no
    data pointer, and only the per-gate nodes (``setup_context`` style, with ``vmap`` and ``jvp`` rules) may see it.  Also
    True for a dual tensor of plain forward-mode AD (``torch.autograd.forward_ad``): a raw kernel would drop its tangent."""
    if t is None:
        return False
    if _functorch.is_wrapped_tensor(t):
        return True
    if forward_ad_active() is False:
        return False
    return _fwad.unpack_dual(t).tangent is not None


def _polar_scale(ref: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """Per-sample factor s (a constant: detached) that brings ``d`` to the size of ``ref``, for the forward-mode rules
    that get a bilinear quantity by polarisation, f(ref + s d) - f(ref - s d) = 4 s Re <ref| . |d>: taken with s = 1 the
    difference of two quadratics cancels catastrophically when |d| << |ref| (complex64: a tangent scaled by 1e-7 came
    back 49 % off); with |s d| = |ref| the round-off is that of one reduction, relative to |ref| |d|, whatever the
    tangent's scale -- a JVP has to be linear in its tangent."""
    with torch.no_grad():
        nr = torch.linalg.vector_norm(ref.detach(), dim=-1, keepdim=True)
        nd = torch.linalg.vector_norm(d.detach(), dim=-1, keepdim=True)
        ok = (nd > 0) & (nr > 0)
        return torch.where(ok, nr / torch.where(ok, nd, torch.ones_like(nd)), torch.ones_like(nd))


class _ApplyGate(torch.autograd.Function):
    """y = (U on targets | controls) x  for x: (B, 2**n), U: (Bm, D, D).

    Written in the ``setup_context`` style so that it composes with ``torch.func`` transforms; the
    ``vmap`` rule folds the mapped dimension into the kernels' batch dimension (per-sample matrix stride),
    which is what lets ``torch.vmap`` over a circuit -- the reference's batching mechanism,
    circuit.py:232-240 -- run on the same kernels."""

    @staticmethod
    def forward(state: torch.Tensor, mats: torch.Tensor, targets: tuple, controls: tuple) -> torch.Tensor:
        return backend.apply_gate(_plain(state), _plain(mats), targets, controls)

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, mats, targets, controls = inputs
        ctx.targets, ctx.controls = targets, controls
        ctx.save_for_backward(state, mats)
        ctx.save_for_forward(state, mats)

    @staticmethod
    def jvp(ctx, state_t, mats_t, _targets_t, _controls_t):
        _single_forward_level()
        # forward mode (torch.func.jvp / jacfwd / hessian): d(U x) = U dx + dU x, the second term on the controlled
        # subspace only -- two gate applications
        state, mats = ctx.saved_tensors
        out = None
        if state_t is not None:
            out = apply_gate(state_t, mats, ctx.targets, ctx.controls)
        if mats_t is not None:
            term = apply_masked(state, mats_t, ctx.targets, ctx.controls)
            out = term if out is None else out + term
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        state, mats = ctx.saved_tensors
        gy = gy.contiguous()
        gstate = gmats = None
        if ctx.needs_input_grad[0]:
            # d/dx of y = U x  ->  U^H gy on the same targets / controls (other amplitudes: identity)
            gstate = apply_gate(gy, mats.mH.resolve_conj().contiguous(), ctx.targets, ctx.controls)
        if ctx.needs_input_grad[1]:
            g = gate_grad(state, gy, ctx.targets, ctx.controls)  # (B, D, D) complex128; a node of its own
            if mats.shape[0] == 1 and g.shape[0] > 1:
                g = g.sum(dim=0, keepdim=True)
            gmats = g.to(mats.dtype)
        return gstate, gmats, None, None

    @staticmethod
    def vmap(info, in_dims, state, mats, targets, controls):
        sd, md = in_dims[0], in_dims[1]
        v = info.batch_size
        state = state.movedim(sd, 0) if sd is not None else state.unsqueeze(0).expand(v, *state.shape)
        b = state.shape[1]
        if md is not None:
            mats = mats.movedim(md, 0)
            d = mats.shape[-1]
            mats = mats.expand(v, b, d, d).reshape(v * b, d, d)
        elif mats.shape[0] > 1:
            mats = mats.unsqueeze(0).expand(v, *mats.shape).reshape(v * b, *mats.shape[1:])
        out = _ApplyGate.apply(state.reshape(v * b, -1).contiguous(), mats.resolve_conj().contiguous(), targets, controls)
        return out.reshape(v, b, -1), 0


def apply_gate(
    state: torch.Tensor, mats: torch.Tensor, targets: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """Differentiable single-gate application on a contiguous (B, 2**n) state.

    ``mats`` is (D, D) or (Bm, D, D) with Bm in {1, B}; it is cast to the state's dtype (the reference
    multiplies a matrix and a state of the same dtype because ``.to()`` converts both,
    operation.py:156-169)."""
    if mats.ndim == 2:
        mats = mats.unsqueeze(0)
    if mats.dtype != state.dtype:
        mats = mats.to(state.dtype)
    if not _is_batched(state) and not state.is_contiguous():
        state = state.contiguous()
    return _ApplyGate.apply(state, mats, tuple(int(t) for t in targets), tuple(int(c) for c in controls))


def apply_masked(
    state: torch.Tensor, mats: torch.Tensor, targets: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """``mats`` applied on the amplitudes whose control bits are all set, ZERO elsewhere (a controlled gate leaves
    those alone instead): P_c (M on targets) x.  Differentiable.  With controls it is the difference of two controlled
    applications -- (x + P_c (M - 1) x) - (x - P_c x) -- exact in floating point: outside the controlled subspace both
    terms are x itself, inside the second one is 0."""
    if not controls:
        return apply_gate(state, mats, targets, ())
    return apply_gate(state, mats, targets, controls) - apply_gate(state, torch.zeros_like(mats), targets, controls)


class _GateGrad(torch.autograd.Function):
    """G[b] = sum over the controlled amplitude groups of gy (outer) conj(x): (B, D, D) complex128 -- the matrix
    cotangent of ``_ApplyGate`` (one read of both states, ``dq_gate_grad_*``).  G is linear in gy and anti-linear in
    x, so its own backward is two masked gate applications: the cotangent of gy is (gG on targets) x, the cotangent
    of x is (gG^H on targets) gy, both on the controlled subspace only."""

    @staticmethod
    def forward(x: torch.Tensor, gy: torch.Tensor, targets: tuple, controls: tuple) -> torch.Tensor:
        return backend.gate_grad(_plain(x), _plain(gy), targets, controls)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, gy, ctx.targets, ctx.controls = inputs
        ctx.save_for_backward(x, gy)
        ctx.save_for_forward(x, gy)

    @staticmethod
    def jvp(ctx, x_t, gy_t, _targets_t, _controls_t):
        _single_forward_level()
        x, gy = ctx.saved_tensors           # G is linear in gy and anti-linear in x
        out = None
        if gy_t is not None:
            out = gate_grad(x, gy_t, ctx.targets, ctx.controls)
        if x_t is not None:
            term = gate_grad(x_t, gy, ctx.targets, ctx.controls)
            out = term if out is None else out + term
        return out

    @staticmethod
    def backward(ctx, gg: torch.Tensor):
        x, gy = ctx.saved_tensors
        gx = ggy = None
        m = gg.to(x.dtype)
        if ctx.needs_input_grad[0]:
            gx = apply_masked(gy, m.mH.resolve_conj().contiguous(), ctx.targets, ctx.controls)
        if ctx.needs_input_grad[1]:
            ggy = apply_masked(x, m.contiguous(), ctx.targets, ctx.controls)
        return gx, ggy, None, None

    @staticmethod
    def vmap(info, in_dims, x, gy, targets, controls):
        v = info.batch_size
        x = x.movedim(in_dims[0], 0) if in_dims[0] is not None else x.unsqueeze(0).expand(v, *x.shape)
        gy = gy.movedim(in_dims[1], 0) if in_dims[1] is not None else gy.unsqueeze(0).expand(v, *gy.shape)
        b = x.shape[1]
        out = _GateGrad.apply(x.reshape(v * b, -1).contiguous(), gy.reshape(v * b, -1).contiguous(), targets, controls)
        return out.reshape(v, b, *out.shape[1:]), 0


def gate_grad(
    x: torch.Tensor, gy: torch.Tensor, targets: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """Differentiable sum gy (outer) conj(x) over the amplitude groups of a gate: complex128 (B, D, D)."""
    if not _is_batched(x) and not x.is_contiguous():
        x = x.contiguous()
    if not _is_batched(gy) and not gy.is_contiguous():
        gy = gy.contiguous()
    return _GateGrad.apply(x, gy, tuple(int(t) for t in targets), tuple(int(c) for c in controls))


class _ExpectPauli(torch.autograd.Function):
    """Re <psi|P|psi> per batch sample, P a Pauli string given by bit masks."""

    @staticmethod
    def forward(state: torch.Tensor, xmask: int, zmask: int) -> torch.Tensor:
        return backend.expect_pauli(_plain(state), xmask, zmask).to(state.real.dtype)

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, ctx.xmask, ctx.zmask = inputs
        ctx.save_for_backward(state)
        ctx.save_for_forward(state)

    @staticmethod
    def jvp(ctx, state_t, _xmask_t, _zmask_t):
        _single_forward_level()
        (state,) = ctx.saved_tensors        # d Re<psi|P|psi> = 2 Re <P psi|d psi>
        ppsi = apply_pauli(state, ctx.xmask, ctx.zmask, differentiable=True)
        return (2.0 * (ppsi.conj() * state_t).sum(dim=-1).real).to(state.real.dtype)

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        (state,) = ctx.saved_tensors
        # L = psi^H P psi with P Hermitian: dL/d(conj psi) = P psi; PyTorch's convention for a real loss
        # of a complex tensor is grad = 2 * dL/d(conj psi).
        # (P psi through the differentiable gate applications, so that a second derivative sees it)
        # (under create_graph only; a first-order backward takes all factors in one fused pass)
        legacy = _functorch.is_legacy_batched
        if (not torch.is_grad_enabled() and state.ndim == 2 and not _is_batched(state) and not _is_wrapped(g)
                and not _is_wrapped(state) and not legacy(g) and not legacy(state) and _functorch.no_transforms()):
            return apply_pauli(state, ctx.xmask, ctx.zmask, scale=2.0 * g), None, None
        ppsi = apply_pauli(state, ctx.xmask, ctx.zmask, differentiable=torch.is_grad_enabled())
        return (2.0 * g).to(state.real.dtype).unsqueeze(-1) * ppsi, None, None

    @staticmethod
    def vmap(info, in_dims, state, xmask, zmask):
        v = info.batch_size
        state = state.movedim(in_dims[0], 0)
        b = state.shape[1]
        out = _ExpectPauli.apply(state.reshape(v * b, -1).contiguous(), xmask, zmask)
        return out.reshape(v, b), 0


_PAULI = {}


def _pauli_mats(dtype: torch.dtype, device: torch.device) -> dict[str, torch.Tensor]:
    key = (dtype, device)
    if key not in _PAULI:
        _PAULI[key] = {
            'x': torch.tensor([[0, 1], [1, 0]], dtype=dtype, device=device),
            'y': torch.tensor([[0, -1j], [1j, 0]], dtype=dtype, device=device),
            'z': torch.tensor([[1, 0], [0, -1]], dtype=dtype, device=device),
        }
    return _PAULI[key]


def apply_pauli(state: torch.Tensor, xmask: int, zmask: int, differentiable: bool = False,
                scale: torch.Tensor | None = None) -> torch.Tensor:
    """P|psi> for a Pauli string (``differentiable``: through the autograd-aware gate applications).  ``scale`` (real, one
    number per sample; not with ``differentiable``): s P|psi> -- the factor rides on the matrix of the first Pauli instead
    of costing a read and a write of the state of its own (the cotangent 2 g P psi of an expectation value)."""
    mats = _pauli_mats(state.dtype, state.device)
    n = state.shape[-1].bit_length() - 1
    factors = [(p, 'y' if (xmask >> p) & (zmask >> p) & 1 else ('x' if (xmask >> p) & 1 else 'z'))
               for p in range(n) if ((xmask | zmask) >> p) & 1]
    if scale is not None:
        assert not differentiable and state.ndim == 2 and not _is_batched(state)
        if not factors:
            return state * scale.to(state.real.dtype).reshape(-1, 1)
        from . import executor

        s_ = scale.to(state.real.dtype).reshape(-1, 1, 1)
        kinds = {'x': 'gen', 'y': 'gen', 'z': 'diag'}          # (a scaled X is no bit flip any more)
        first = mats[factors[0][1]] * s_                         # (B, 2, 2), or (1, 2, 2)
        prims = [executor.Prim(kinds[factors[0][1]], first if first.shape[0] > 1 else first[0], (factors[0][0],), (), unitary=False)]
        prims += [executor.Prim({'x': 'x', 'y': 'gen', 'z': 'diag'}[c], mats[c], (p,), ()) for p, c in factors[1:]]
        with torch.no_grad():
            return executor.run(state.detach(), prims)
    if not differentiable and len(factors) >= 2 and not _is_batched(state) and state.ndim == 2:
        # all factors in one fused pass (the reference applies them one after the other, qmath.py:846-856: a read and a
        # write of the state EACH -- <X..X> on n qubits, the observable of its own gradient benchmark, cost n of them)
        from . import executor

        kinds = {'x': 'x', 'y': 'gen', 'z': 'diag'}
        prims = [executor.Prim(kinds[c], mats[c], (p,), ()) for p, c in factors]
        with torch.no_grad():
            return executor.run(state.detach(), prims)
    out = state
    for p, c in factors:
        out = apply_gate(out, mats[c], [p], []) if differentiable else backend.apply_gate(out, mats[c], [p], [])
    return out if out is not state else state.clone()


def expect_pauli(state: torch.Tensor, xmask: int, zmask: int) -> torch.Tensor:
    """Differentiable Re <psi_b|P|psi_b>, real (B,) in the state's real precision."""
    if not _is_batched(state) and not state.is_contiguous():
        state = state.contiguous()
    return _ExpectPauli.apply(state, int(xmask), int(zmask))


class _Marginal(torch.autograd.Function):
    """p[b, k] = sum over amplitudes whose ``bits`` spell k of |psi_b|^2 (bits[0] = MSB of k)."""

    @staticmethod
    def forward(state: torch.Tensor, bits: tuple) -> torch.Tensor:
        return backend.marginal(_plain(state), bits).to(state.real.dtype)

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, ctx.bits = inputs
        ctx.save_for_backward(state)
        ctx.save_for_forward(state)

    @staticmethod
    def jvp(ctx, state_t, _bits_t):
        _single_forward_level()
        (state,) = ctx.saved_tensors        # by polarisation: |psi + s d|^2 - |psi - s d|^2 = 4 s Re conj(psi) d
        s = _polar_scale(state, state_t)    # (|s d| = |psi|: see `_polar_scale`)
        d = state_t * s
        return (marginal(state + d, ctx.bits) - marginal(state - d, ctx.bits)) * (0.5 / s)

    @staticmethod
    def vmap(info, in_dims, state, bits):
        v = info.batch_size
        state = state.movedim(in_dims[0], 0)
        b = state.shape[1]
        out = _Marginal.apply(state.reshape(v * b, -1).contiguous(), bits)
        return out.reshape(v, b, -1), 0

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        (state,) = ctx.saved_tensors
        # d p_k / d conj(psi_i) = psi_i [bits(i) = k]; real loss of a complex tensor: grad = 2 * that,
        # i.e. a diagonal "gate" diag(2 g[b, :]) on the measured bits
        diag = (2.0 * g).to(state.dtype).diag_embed()
        return apply_gate(state, diag.contiguous(), ctx.bits, ()), None


def marginal(state: torch.Tensor, bits: Sequence[int]) -> torch.Tensor:
    """Differentiable marginal distribution over ``bits`` of a contiguous (B, 2**n) state."""
    if not state.is_contiguous():
        state = state.contiguous()
    return _Marginal.apply(state, tuple(int(b) for b in bits))


class _ExpectZMulti(torch.autograd.Function):
    """Re <psi_b| Z-string_k |psi_b> for K Z-type strings: (B, K), one read of the state per 32 strings."""

    @staticmethod
    def forward(state: torch.Tensor, zmasks: tuple) -> torch.Tensor:
        return backend.expect_z_multi(_plain(state), zmasks).to(state.real.dtype)

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, ctx.zmasks = inputs
        ctx.save_for_backward(state)
        ctx.save_for_forward(state)

    @staticmethod
    def jvp(ctx, state_t, _zmasks_t):
        _single_forward_level()
        (state,) = ctx.saved_tensors        # P_k real diagonal: d <psi|P_k|psi> = 2 Re <psi|P_k|d psi>, by polarisation
        s = _polar_scale(state, state_t)    # (|s d| = |psi|: see `_polar_scale`)
        d = state_t * s
        return (expect_z_multi(state + d, ctx.zmasks) - expect_z_multi(state - d, ctx.zmasks)) * (0.5 / s)

    @staticmethod
    def vmap(info, in_dims, state, zmasks):
        v = info.batch_size
        state = state.movedim(in_dims[0], 0)
        b = state.shape[1]
        out = _ExpectZMulti.apply(state.reshape(v * b, -1).contiguous(), zmasks)
        return out.reshape(v, b, -1), 0

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        (state,) = ctx.saved_tensors
        # L = sum_k g_k psi^H P_k psi with P_k diagonal: grad = 2 (sum_k g_k P_k) psi, one read + one write
        return scale_z_signs(state, ctx.zmasks, 2.0 * g), None


class _ScaleZSigns(torch.autograd.Function):
    """out[b, i] = (sum_k w[b, k] s_k(i)) psi[b, i] with s_k(i) = (-1)^popcount(i & zmask_k): the cotangent of
    ``_ExpectZMulti`` as a node of its own, so that second derivatives exist.  Linear in psi (real diagonal factor:
    its cotangent is the same scaling of the incoming cotangent) and in w (d/dw_k = Re <g| P_k |psi>, taken by
    polarisation from the Z-string reduction itself)."""

    @staticmethod
    def forward(state: torch.Tensor, zmasks: tuple, w: torch.Tensor) -> torch.Tensor:
        return backend.scale_z_signs(_plain(state), zmasks, w)

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, ctx.zmasks, w = inputs
        ctx.save_for_backward(state, w)
        ctx.save_for_forward(state, w)

    @staticmethod
    def jvp(ctx, state_t, _zmasks_t, w_t):
        _single_forward_level()
        state, w = ctx.saved_tensors        # linear in psi and in w
        out = None
        if state_t is not None:
            out = scale_z_signs(state_t, ctx.zmasks, w)
        if w_t is not None:
            term = scale_z_signs(state, ctx.zmasks, w_t)
            out = term if out is None else out + term
        return out

    @staticmethod
    def vmap(info, in_dims, state, zmasks, w):
        v = info.batch_size
        state = state.movedim(in_dims[0], 0) if in_dims[0] is not None else state.unsqueeze(0).expand(v, *state.shape)
        w = w.movedim(in_dims[2], 0) if in_dims[2] is not None else w.unsqueeze(0).expand(v, *w.shape)
        b = state.shape[1]
        out = _ScaleZSigns.apply(state.reshape(v * b, -1).contiguous(), zmasks, w.reshape(v * b, -1).contiguous())
        return out.reshape(v, b, -1), 0

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        state, w = ctx.saved_tensors
        g = g.contiguous()
        gstate = gw = None
        if ctx.needs_input_grad[0]:
            gstate = scale_z_signs(g, ctx.zmasks, w)
        if ctx.needs_input_grad[2]:
            # Re <g| P |psi> = (<s g + psi| P |s g + psi> - <s g - psi| P |s g - psi>) / (4 s), |s g| = |psi|
            s = _polar_scale(state, g)
            gs = g * s
            gw = ((expect_z_multi(gs + state, ctx.zmasks) - expect_z_multi(gs - state, ctx.zmasks)) * (0.25 / s)).to(w.dtype)
        return gstate, None, gw


def scale_z_signs(state: torch.Tensor, zmasks: Sequence[int], w: torch.Tensor) -> torch.Tensor:
    """Differentiable (sum_k w_k P_k) psi for Z-type strings P_k, w real (B, K)."""
    if not state.is_contiguous():
        state = state.contiguous()
    return _ScaleZSigns.apply(state, tuple(int(z) for z in zmasks), w)


def expect_z_multi(state: torch.Tensor, zmasks: Sequence[int]) -> torch.Tensor:
    """Differentiable expectation values of several Z-type Pauli strings at once: real (B, K)."""
    if not state.is_contiguous():
        state = state.contiguous()
    return _ExpectZMulti.apply(state, tuple(int(z) for z in zmasks))
