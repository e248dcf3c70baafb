"""Tensor-level entry points of the statevector kernels.

Every function takes PyTorch tensors (device memory owned by the caller), checks shapes on the host the
way the reference asserts them (operation.py:91-107, 285-288), and enqueues the matching libdqhip call
on ``torch.cuda.current_stream()``.  CUDA (= HIP on ROCm) tensors are required: a CPU tensor raises
``RuntimeError`` -- there is no CPU fallback in the product.  The unit tests that exercise host logic
without a GPU install an explicit test double with :func:`set_test_backend` (it lives under ``tests/``
and is built on the oracle); the package itself never imports ``oracle/``.
"""

from __future__ import annotations

import ctypes as C
from typing import Any, Sequence

import torch

from . import _lib

_test_backend: Any = None


class _NoGraph:
    """A test double must not be more capable than the product: the HIP entry points return tensors without an
    autograd graph, so every call into the double runs under ``no_grad`` as well (a double built from differentiable
    torch operations once hid a missing second derivative) -- and with forward-mode tangents switched off: a kernel
    that reads raw buffers carries none, the nodes' ``jvp`` rules do."""

    def __init__(self, obj: Any):
        self._obj = obj

    def __getattr__(self, name: str) -> Any:
        attr = getattr(self._obj, name)
        if not callable(attr):
            return attr

        def call(*args, **kwargs):
            for a in list(args) + list(kwargs.values()):      # (what `_ptr` refuses, the double refuses)
                if isinstance(a, torch.Tensor) and (a.is_conj() or a.is_neg()):
                    raise ValueError('tensor with a pending conjugation / negation: resolve_conj() / resolve_neg() it first')
            was = torch._C._is_fwd_grad_enabled()
            torch._C._set_fwd_grad_enabled(False)
            try:
                with torch.no_grad():
                    return attr(*args, **kwargs)
            finally:
                torch._C._set_fwd_grad_enabled(was)

        return call


def set_test_backend(obj: Any) -> None:
    """Install (or clear with ``None``) a CPU test double.  For ``tests/`` only."""
    global _test_backend
    _test_backend = None if obj is None else _NoGraph(obj)


def get_test_backend() -> Any:
    return None if _test_backend is None else _test_backend._obj


def _suffix(t: torch.Tensor) -> str:
    if t.dtype == torch.complex64:
        return 'c64'
    if t.dtype == torch.complex128:
        return 'c128'
    raise TypeError(f'state must be complex64 or complex128, got {t.dtype}')


def _use_hip(t: torch.Tensor) -> bool:
    if t.is_cuda:
        return True
    if _test_backend is not None:
        return False
    raise RuntimeError(
        'deepquantum_amd: statevector kernels run on MI355X only; got a tensor on '
        f'{t.device}. Move the circuit/state to "cuda" (there is no CPU fallback).'
    )


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    # a lazily conjugated / negated view (x.mH of a column-major matrix is "contiguous" AND unconjugated in memory) must
    # never reach a kernel as a raw pointer
    if t is not None and (t.is_conj() or t.is_neg()):
        raise ValueError('tensor with a pending conjugation / negation: resolve_conj() / resolve_neg() it first')
    return C.c_void_p(0 if t is None else t.data_ptr())


def _nqubit(state: torch.Tensor) -> int:
    dim = state.shape[-1]
    n = dim.bit_length() - 1
    if state.ndim != 2 or (1 << n) != dim:
        raise ValueError(f'state must have shape (batch, 2**n), got {tuple(state.shape)}')
    return n


def _check_mats(state: torch.Tensor, mats: torch.Tensor, k: int) -> tuple[torch.Tensor, int]:
    """Return contiguous (Bm, D, D) matrices in the state's dtype/device and the batch stride."""
    d = 1 << k
    if mats.ndim == 2:
        mats = mats.unsqueeze(0)
    if mats.ndim != 3 or mats.shape[-1] != d or mats.shape[-2] != d:
        raise ValueError(f'matrix must have shape (.., {d}, {d}), got {tuple(mats.shape)}')
    if mats.shape[0] not in (1, state.shape[0]):
        raise ValueError(f'matrix batch {mats.shape[0]} does not match state batch {state.shape[0]}')
    if mats.dtype != state.dtype or mats.device != state.device:
        mats = mats.to(device=state.device, dtype=state.dtype)
    mats = mats.resolve_conj().resolve_neg().contiguous()     # (.mH of a strided user matrix is contiguous with the conj bit set)
    stride = 0 if mats.shape[0] == 1 else d * d
    return mats, stride


MAX_BATCH = 32768     # slices of a batch wider than the 65535 a grid dimension can hold


# ---------------------------------------------------------------------------------------------------
def apply_gate(
    state: torch.Tensor,
    mats: torch.Tensor,
    targets: Sequence[int],
    controls: Sequence[int] = (),
    out: torch.Tensor | None = None,
) -> torch.Tensor:
    """psi' = (U on ``targets``, conditioned on all ``controls`` = 1) psi.

    ``state``: (B, 2**n) contiguous complex; ``targets``/``controls``: bit positions (LSB = 0),
    ``targets[0]`` is the matrix-index MSB.  ``out`` may be ``state`` itself (in place) for k <= 4.
    Replaces qmath.evolve_state / Gate.op_state_control of the reference.
    """
    n = _nqubit(state)
    targets, controls = [int(t) for t in targets], [int(c) for c in controls]
    k = len(targets)
    if not state.is_contiguous():
        raise ValueError('state must be contiguous')
    mats, stride = _check_mats(state, mats, k)
    if out is None:
        out = torch.empty_like(state)
    if not _use_hip(state):
        return _test_backend.apply_gate(state, mats, targets, controls, out)
    if state.shape[0] > MAX_BATCH:       # the batch is a grid dimension (<= 65535): very wide batches go in slices
        for lo in range(0, state.shape[0], MAX_BATCH):
            hi = min(lo + MAX_BATCH, state.shape[0])
            apply_gate(state[lo:hi], mats[lo:hi] if stride else mats, targets, controls, out[lo:hi])
        return out
    lib = _lib.load()
    fn = getattr(lib, f'dq_apply_gate_{_suffix(state)}')
    rc = fn(_ptr(state), _ptr(out), _ptr(mats), stride, n, _lib.int_array(targets), k,
            _lib.int_array(controls), len(controls), state.shape[0], _stream(state))
    _lib.check(rc, 'dq_apply_gate')
    return out


#: records of a pass that travel in the kernel-argument segment (csrc/dq_wave.hip, WAVE_MAX_REC); a pass with more keeps them
#: in device memory (include/dq_hip.h, dq_wave_records / dq_apply_fused_grad_ext_*)
KERNARG_RECORDS = 112
#: Device tensors that launches of a HIP graph UNDER CAPTURE read and that a plan or a cache owns -- the kernel matrix buffer
#: of a circuit with fixed gates, the offsets of its deferred Rx blocks, the records of a long pass: the graph has their
#: addresses baked in, so they must outlive the plan cache's evictions.  Only tensors a cache OWNS are pinned (`own`):
#: what a capture allocates itself -- the matrix buffer of a trainable circuit, built anew inside the capture -- comes
#: from the graph's private pool and lives with the graph; pinning it would keep pool blocks from ever being reused.
#: Pins of a capture made by `utils.CapturedGraph` belong to that object (`_PIN_SINK`) and die with it; those of a raw
#: ``torch.cuda.graph`` capture stay here: a few KiB per captured circuit.
_CAPTURE_PINS: dict = {}
_PIN_SINK: list = []          # the innermost CapturedGraph under construction keeps its pins in _PIN_SINK[-1]


def own(t: torch.Tensor | None) -> torch.Tensor | None:
    """Mark a device tensor as owned by a plan / steady-state cache (what `pin_if_capturing` keeps alive)."""
    if t is not None:
        t._dq_owned = True
    return t


def pin_if_capturing(*tensors: torch.Tensor | None) -> None:
    first = next((t for t in tensors if t is not None), None)
    if first is None or not first.is_cuda or not torch.cuda.is_current_stream_capturing():
        return
    for t in tensors:
        if t is not None and getattr(t, '_dq_owned', False):
            if _PIN_SINK:
                _PIN_SINK[-1].setdefault(id(t), t)
            else:
                _CAPTURE_PINS.setdefault(id(t), t)


def _device_records(desc: _lib.DqFusedPass, n: int, device: torch.device) -> torch.Tensor | None:
    """The records of ``desc`` on ``device`` (a uint8 tensor, cached on the descriptor) when they do not fit the
    kernel-argument segment, else None.  Only passes with many gates can need them: up to 72 gates and reductions (the cap
    of every pass before ABI 24) the planner's bound on layout changes keeps a pass within the segment, and the count is not
    even asked for."""
    if desc.rounds[desc.nrounds - 1].gate_end <= 72:
        return None
    cache = desc.__dict__.setdefault('_dev_records', {})
    key = (n, device)
    hit = cache.get(key, False)
    if hit is False:
        lib = _lib.load()
        nbytes = lib.dq_wave_records(C.byref(desc), n, None, 0)
        if nbytes < 0:
            _lib.check(int(nbytes), 'dq_wave_records')
        if nbytes <= 32 * KERNARG_RECORDS:
            hit = None
        else:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('deepquantum_amd: the first run of a pass with more than 112 records copies them to the '
                                   'device; run the step once before capturing it (dq.CapturedGraph warms up by itself)')
            host = torch.empty(int(nbytes), dtype=torch.uint8)
            got = lib.dq_wave_records(C.byref(desc), n, host.data_ptr(), int(nbytes))
            assert got == nbytes
            hit = own(host.to(device))
        cache[key] = hit
    return hit


def apply_fused(
    state: torch.Tensor,
    mats: torch.Tensor,
    mat_batch_stride: int,
    desc: _lib.DqFusedPass,
    out: torch.Tensor | None = None,
    grads: torch.Tensor | None = None,
    known_zero: int = 0,
    slice_bits: tuple[int, int] | None = None,
) -> torch.Tensor:
    """Run one fused pass (see fusion.py).  ``mats``: flat complex buffer (Bm * stride or stride).
    ``state`` with ONE row and ``out`` with B rows: the single input state is shared by all outputs.
    ``grads`` (float64, (B, rows, 8), added to): a pass of the adjoint method's reverse sweep -- its DQ_FG_GRAD
    records reduce into it (include/dq_hip.h, dq_apply_fused_grad_c64 / _c128).
    ``known_zero``: index bits (read side) known to be |0> in ``state`` -- it is not read where one of them is 1, and
    ``out`` is not written where such a bit outside the tile is 1 (include/dq_hip.h, dq_apply_fused_zext_*).
    ``slice_bits = (mask, value)``: ONE SLICE of the pass -- only the tiles whose index bits ``mask`` (read side, outside the
    tile) equal ``value`` run (include/dq_hip.h, dq_apply_fused_slice_*)."""
    n = _nqubit(state)
    if out is None:
        out = state
    broadcast = state.shape[0] == 1 and out.shape[0] > 1
    if known_zero and grads is not None:
        raise ValueError('a reducing pass takes no known-zero bits')
    if slice_bits is not None and (grads is not None or out.shape[0] > MAX_BATCH):
        raise ValueError('a slice of a pass: no reducing pass, at most MAX_BATCH samples')
    if mats.dtype != state.dtype or mats.device != state.device or not mats.is_contiguous():
        raise ValueError('mats must be a contiguous buffer in the dtype/device of the state')
    if grads is not None:
        if (grads.dtype != torch.float64 or grads.ndim != 3 or grads.shape[0] != out.shape[0] or grads.shape[2] != 8
                or not grads.is_contiguous() or grads.device != state.device or broadcast):
            raise ValueError('grads must be a contiguous float64 (batch, rows, 8) accumulator on the device of the state')
    if not _use_hip(state):
        src = state.expand(out.shape[0], -1) if broadcast else state
        if grads is not None:
            return _test_backend.apply_fused(src, mats, mat_batch_stride, desc, out, grads=grads)
        if slice_bits is not None:
            return _test_backend.apply_fused(src, mats, mat_batch_stride, desc, out, known_zero=known_zero, slice_bits=slice_bits)
        if known_zero:
            return _test_backend.apply_fused(src, mats, mat_batch_stride, desc, out, known_zero=known_zero)
        return _test_backend.apply_fused(src, mats, mat_batch_stride, desc, out)
    if grads is not None:
        if out.shape[0] > MAX_BATCH:
            raise ValueError(f'reverse-sweep passes take at most {MAX_BATCH} samples')
        lib = _lib.load()
        rec = _device_records(desc, n, state.device)
        pin_if_capturing(mats, rec)
        if rec is not None:       # more records than the kernel-argument segment holds: the kernel reads them from device memory
            fn = lib.dq_apply_fused_grad_ext_c128 if state.dtype == torch.complex128 else lib.dq_apply_fused_grad_ext_c64
            rc = fn(_ptr(state), _ptr(out), _ptr(mats), int(mat_batch_stride), n, out.shape[0], C.byref(desc), _ptr(rec),
                    rec.numel(), _ptr(grads), grads.shape[1], _stream(state))
            _lib.check(rc, 'dq_apply_fused_grad_ext')
            return out
        fn = lib.dq_apply_fused_grad_c128 if state.dtype == torch.complex128 else lib.dq_apply_fused_grad_c64
        rc = fn(_ptr(state), _ptr(out), _ptr(mats), int(mat_batch_stride), n, out.shape[0], C.byref(desc), _ptr(grads),
                grads.shape[1], _stream(state))
        _lib.check(rc, 'dq_apply_fused_grad')
        return out
    if out.shape[0] > MAX_BATCH:
        rows = mats.reshape(-1, mat_batch_stride) if mat_batch_stride else None
        for lo in range(0, out.shape[0], MAX_BATCH):
            hi = min(lo + MAX_BATCH, out.shape[0])
            apply_fused(state if broadcast else state[lo:hi], rows[lo:hi].reshape(-1) if rows is not None else mats,
                        mat_batch_stride, desc, out[lo:hi], known_zero=known_zero)
        return out
    lib = _lib.load()
    pin_if_capturing(mats)
    if slice_bits is not None:
        fn = lib.dq_apply_fused_slice_c128 if state.dtype == torch.complex128 else lib.dq_apply_fused_slice_c64
        rc = fn(_ptr(state), 0 if broadcast else 1 << n, _ptr(out), _ptr(mats), int(mat_batch_stride), n, out.shape[0],
                C.byref(desc), int(known_zero), int(slice_bits[0]), int(slice_bits[1]), _stream(state))
        _lib.check(rc, 'dq_apply_fused_slice')
        return out
    if known_zero:
        fn = lib.dq_apply_fused_zext_c128 if state.dtype == torch.complex128 else lib.dq_apply_fused_zext_c64
        rc = fn(_ptr(state), 0 if broadcast else 1 << n, _ptr(out), _ptr(mats), int(mat_batch_stride), n, out.shape[0],
                C.byref(desc), int(known_zero), _stream(state))
        _lib.check(rc, 'dq_apply_fused_zext')
        return out
    fn = getattr(lib, f'dq_apply_fused_{"bcast_" if broadcast else ""}{_suffix(state)}')
    rc = fn(_ptr(state), _ptr(out), _ptr(mats), int(mat_batch_stride), n, out.shape[0], C.byref(desc),
            _stream(state))
    _lib.check(rc, 'dq_apply_fused')
    return out


def fused_geometry(is_c128: bool, variant: int = 0) -> tuple[int, int, int]:
    """(m, slots, threads) of a compiled fused-kernel variant."""
    if _test_backend is not None and not torch.cuda.is_available():
        return _test_backend.fused_geometry(is_c128, variant)
    lib = _lib.load()
    m, s, t = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.dq_fused_geometry(int(is_c128), variant, C.byref(m), C.byref(s), C.byref(t)),
               'dq_fused_geometry')
    return m.value, s.value, t.value


_ws_cache: dict[tuple, torch.Tensor] = {}


def _workspace(state: torch.Tensor) -> torch.Tensor:
    # one workspace per (device, batch, stream): reductions enqueued on two streams must not share scratch
    stream = torch.cuda.current_stream(state.device).cuda_stream if state.is_cuda else 0
    key = (state.device, state.shape[0], stream)
    ws = _ws_cache.get(key)
    if ws is None:
        nbytes = _lib.load().dq_reduce_ws_bytes(state.shape[0])
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=state.device)
        _ws_cache[key] = ws
    return ws


def expect_pauli(state: torch.Tensor, xmask: int, zmask: int) -> torch.Tensor:
    """Re <psi_b|P|psi_b> for the Pauli string (xmask, zmask); a Y sets its bit in both masks.
    Returns float64 (B,)."""
    n = _nqubit(state)
    if not state.is_contiguous():
        raise ValueError('state must be contiguous')
    if not _use_hip(state):
        return _test_backend.expect_pauli(state, xmask, zmask)
    out = torch.empty(state.shape[0], dtype=torch.float64, device=state.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_expect_pauli_{_suffix(state)}')
    rc = fn(_ptr(state), xmask, zmask, n, state.shape[0], _ptr(out), _ptr(_workspace(state)), _stream(state))
    _lib.check(rc, 'dq_expect_pauli')
    return out


def _u64_array(values: Sequence[int]):
    return (C.c_uint64 * max(len(values), 1))(*[int(v) for v in values])


def expect_z_multi(state: torch.Tensor, zmasks: Sequence[int]) -> torch.Tensor:
    """<psi_b| Z-string_k |psi_b> for several Z-type strings in one read of the state per 32 strings:
    float64 (B, K)."""
    n = _nqubit(state)
    if not state.is_contiguous():
        raise ValueError('state must be contiguous')
    if not _use_hip(state):
        return torch.stack([_test_backend.expect_pauli(state, 0, z) for z in zmasks], dim=1)
    lib = _lib.load()
    fn = getattr(lib, f'dq_expect_zmulti_{_suffix(state)}')
    nblocks = max(1, min(2048, (1 << n) // 1024))
    parts = []
    for lo in range(0, len(zmasks), 32):
        grp = zmasks[lo:lo + 32]
        part = torch.empty(state.shape[0], nblocks, len(grp), dtype=torch.float64, device=state.device)
        rc = fn(_ptr(state), _u64_array(grp), len(grp), n, state.shape[0], _ptr(part), nblocks, _stream(state))
        _lib.check(rc, 'dq_expect_zmulti')
        parts.append(part.sum(dim=1))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


def scale_z_signs(state: torch.Tensor, zmasks: Sequence[int], coef: torch.Tensor) -> torch.Tensor:
    """(sum_k coef[b, k] Z-string_k) |psi_b>: (B, 2**n) in the state's dtype; coef real (B, K)."""
    n = _nqubit(state)
    if not _use_hip(state):
        with torch.no_grad():
            return _scale_z_signs_double(state, zmasks, coef, n)
    lib = _lib.load()
    fn = getattr(lib, f'dq_scale_zsigns_{_suffix(state)}')
    coef = coef.to(torch.float64).contiguous()
    out = None
    for lo in range(0, len(zmasks), 32):
        grp = zmasks[lo:lo + 32]
        part = torch.empty_like(state)
        rc = fn(_ptr(state), _ptr(part), _u64_array(grp), len(grp), _ptr(coef[:, lo:lo + 32].contiguous()), n,
                state.shape[0], _stream(state))
        _lib.check(rc, 'dq_scale_zsigns')
        out = part if out is None else out.add_(part)
    return out


def _scale_z_signs_double(state, zmasks, coef, n):
    """``scale_z_signs`` for the CPU test double (tests only)."""
    i = torch.arange(1 << n)
    w = torch.zeros(state.shape[0], 1 << n, dtype=torch.float64)
    for k, z in enumerate(zmasks):
        par = torch.zeros_like(i)
        for p in range(n):
            if (z >> p) & 1:
                par ^= (i >> p) & 1
        w += coef[:, k : k + 1].to(torch.float64) * (1 - 2 * par).to(torch.float64)
    return (state * w.to(state.real.dtype)).contiguous()


def inner(bra: torch.Tensor, ket: torch.Tensor) -> torch.Tensor:
    """<bra_b|ket_b> as complex128 (B,).  Inputs (B, count) contiguous, same dtype."""
    if bra.shape != ket.shape or bra.dtype != ket.dtype or bra.ndim != 2:
        raise ValueError('bra/ket must be matching (B, count) tensors')
    if not (bra.is_contiguous() and ket.is_contiguous()):
        raise ValueError('bra/ket must be contiguous')
    if not _use_hip(bra):
        return _test_backend.inner(bra, ket)
    out = torch.empty(bra.shape[0], 2, dtype=torch.float64, device=bra.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_inner_{_suffix(bra)}')
    rc = fn(_ptr(bra), _ptr(ket), bra.shape[1], bra.shape[0], _ptr(out), _ptr(_workspace(bra)), _stream(bra))
    _lib.check(rc, 'dq_inner')
    return torch.view_as_complex(out)


def probs(state: torch.Tensor) -> torch.Tensor:
    """|psi|^2 elementwise in the state's real precision, same shape."""
    if not state.is_contiguous():
        raise ValueError('state must be contiguous')
    if not _use_hip(state):
        return _test_backend.probs(state)
    out = torch.empty(state.shape, dtype=state.real.dtype, device=state.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_probs_{_suffix(state)}')
    _lib.check(fn(_ptr(state), _ptr(out), state.numel(), _stream(state)), 'dq_probs')
    return out


def marginal(state: torch.Tensor, bits: Sequence[int]) -> torch.Tensor:
    """Marginal distribution over ``bits`` (bits[0] = MSB of the outcome index), float64 (B, 2**nw)."""
    n = _nqubit(state)
    bits = [int(b) for b in bits]
    if not _use_hip(state):
        return _test_backend.marginal(state, bits)
    nw = len(bits)
    b = state.shape[0]
    lib = _lib.load()
    fn = getattr(lib, f'dq_marginal_{_suffix(state)}')
    max_b = 65535                                            # (the batch is a grid axis of its own)
    out = torch.zeros(b, 1 << nw, dtype=torch.float64, device=state.device)
    for lo in range(0, b, max_b):
        hi = min(b, lo + max_b)
        rc = fn(_ptr(state[lo:hi]), n, _lib.int_array(bits), nw, hi - lo, _ptr(out[lo:hi]), _stream(state))
        _lib.check(rc, 'dq_marginal')
    return out


def gate_grad(
    x: torch.Tensor, gy: torch.Tensor, targets: Sequence[int], controls: Sequence[int] = ()
) -> torch.Tensor:
    """gU[b] = sum over controlled amplitude groups of gy (outer) conj(x): complex128 (B, D, D)."""
    n = _nqubit(x)
    targets, controls = [int(t) for t in targets], [int(c) for c in controls]
    k = len(targets)
    if x.shape != gy.shape or x.dtype != gy.dtype:
        raise ValueError('x/gy mismatch')
    if not (x.is_contiguous() and gy.is_contiguous()):
        raise ValueError('x/gy must be contiguous')
    if not _use_hip(x):
        return _test_backend.gate_grad(x, gy, targets, controls)
    d = 1 << k
    if k > 2:
        # Dense blocks on > 2 wires with trainable entries (LatentGate on many wires) are rare; the
        # contraction is a plain GEMM (gy_mat @ x_mat^H) and is left to rocBLAS through torch.
        return _gate_grad_gemm(x, gy, n, targets, controls)
    out = torch.zeros(x.shape[0], d, d, 2, dtype=torch.float64, device=x.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_gate_grad_{_suffix(x)}')
    rc = fn(_ptr(x), _ptr(gy), n, _lib.int_array(targets), k, _lib.int_array(controls), len(controls),
            x.shape[0], _ptr(out), _stream(x))
    _lib.check(rc, 'dq_gate_grad')
    return torch.view_as_complex(out)


def gate_grad_multi(x: torch.Tensor, gy: torch.Tensor, gates: Sequence[tuple[int, Sequence[int]]]) -> torch.Tensor:
    """``gate_grad`` for several single-target gates ``(target, controls)`` on the same pair of states:
    complex128 (B, G, 2, 2).  Gates are packed into as few launches as the kernel's tile allows (each launch reads
    both states once: 8 gates / 7 distinct high targets for complex64, 4 / 7 for complex128); states smaller than a
    tile go gate by gate."""
    n = _nqubit(x)
    if x.shape != gy.shape or x.dtype != gy.dtype or not (x.is_contiguous() and gy.is_contiguous()):
        raise ValueError('x/gy must be contiguous tensors of one shape and dtype')
    is128 = x.dtype == torch.complex128
    tile, per_call, low = (10, 4, 3) if is128 else (11, 8, 4)
    if not _use_hip(x) or n < tile:
        return torch.stack([gate_grad(x, gy, [t], list(c)) for t, c in gates], dim=1)
    nblocks = min(1 << (n - tile), 1536)          # ~6 workgroups per CU striding over the tiles
    parts = []
    lib = _lib.load()
    fn = getattr(lib, f'dq_gate_grad_multi_{_suffix(x)}')
    start = 0
    while start < len(gates):
        stop, high = start, set()
        while stop < len(gates) and stop - start < per_call:
            t = int(gates[stop][0])
            if t >= low and t not in high and len(high) == tile - low:
                break
            if t >= low:
                high.add(t)
            stop += 1
        grp = gates[start:stop]
        begin, bits = [0], []
        for _t, c in grp:
            bits += [int(q) for q in c]
            begin.append(len(bits))
        part = torch.empty(x.shape[0], nblocks, len(grp), 2, 2, 2, dtype=torch.float64, device=x.device)
        rc = fn(_ptr(x), _ptr(gy), n, len(grp), _lib.int_array([int(t) for t, _ in grp]), _lib.int_array(begin),
                _lib.int_array(bits), x.shape[0], _ptr(part), nblocks, _stream(x))
        _lib.check(rc, 'dq_gate_grad_multi')
        parts.append(part.sum(dim=1))
        start = stop
    out = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    return torch.view_as_complex(out.contiguous())


def _gate_grad_gemm(x, gy, n, targets, controls):
    b = x.shape[0]
    wires_t = [n - 1 - t + 1 for t in targets]
    wires_c = [n - 1 - c + 1 for c in controls]
    rest = [i for i in range(1, n + 1) if i not in wires_t and i not in wires_c]
    perm = [0] + wires_t + rest + wires_c
    d = 1 << len(targets)

    def mat(t):
        t = t.reshape([b] + [2] * n).permute(perm).reshape(b, d, -1, 1 << len(controls))
        return t[..., -1]

    return (mat(gy) @ mat(x).mH).to(torch.complex128)


def pack(amps: torch.Tensor, mask: int, value: int) -> torch.Tensor:
    """Gather the sub-cube of each shard whose bits under ``mask`` equal ``value`` into a contiguous
    (B, 2**(nl - popcount(mask))) buffer."""
    nl = _nqubit(amps)
    if not _use_hip(amps):
        return _test_backend.pack(amps, mask, value)
    cnt = 1 << (nl - bin(mask).count('1'))
    out = torch.empty(amps.shape[0], cnt, dtype=amps.dtype, device=amps.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_pack_{_suffix(amps)}')
    _lib.check(fn(_ptr(amps), _ptr(out), nl, mask, value, amps.shape[0], _stream(amps)), 'dq_pack')
    return out


def unpack_axpby(
    amps: torch.Tensor,
    x: torch.Tensor,
    y: torch.Tensor | None,
    coef: torch.Tensor | None,
    mask: int,
    value: int,
) -> torch.Tensor:
    """amps[b, expand(c)] = coef[b,0] * x[b,c] + coef[b,1] * y[b,c] (or = x[b,c] when y is None), where
    expand() re-inserts ``value`` at the ``mask`` bit positions.  In place on ``amps``."""
    nl = _nqubit(amps)
    if not _use_hip(amps):
        return _test_backend.unpack_axpby(amps, x, y, coef, mask, value)
    stride = 0
    if y is not None:
        coef = coef.to(device=amps.device, dtype=amps.dtype).reshape(-1, 2).contiguous()
        if coef.shape[0] not in (1, amps.shape[0]):
            raise ValueError('coef batch mismatch')
        stride = 0 if coef.shape[0] == 1 else 2
    lib = _lib.load()
    fn = getattr(lib, f'dq_unpack_axpby_{_suffix(amps)}')
    rc = fn(_ptr(amps), _ptr(x), _ptr(y), _ptr(coef), stride, nl, mask, value, amps.shape[0], _stream(amps))
    _lib.check(rc, 'dq_unpack_axpby')
    return amps


def defer_rx(flat: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """The deferred form of the uncontrolled Rx-like gates of complex64 passes (include/dq_hip.h, DQ_MODE_RX), in place in
    the kernel matrix buffer ``flat`` (Bm, total): the blocks at the offsets ``index`` (LongTensor on the buffer's
    device; fusion.rx_defer_positions).  One launch (dq_defer_rx_c64); `fusion.defer_rx` is the same rewrite in tensor
    operations (tests and tools compare the two)."""
    if index is None or index.numel() == 0:
        return flat
    if not _use_hip(flat):
        return _test_backend.defer_rx(flat, index)
    assert flat.dtype == torch.complex64 and flat.ndim == 2 and flat.stride(1) == 1 and index.dtype == torch.long
    assert index.device == flat.device and index.is_contiguous()
    pin_if_capturing(index)
    lib = _lib.load()
    for lo in range(0, flat.shape[0], MAX_BATCH):          # (the batch is a grid dimension)
        rows = flat[lo : lo + MAX_BATCH]
        _lib.check(lib.dq_defer_rx_c64(_ptr(rows), flat.stride(0), _ptr(index), index.numel(), rows.shape[0],
                                       _stream(flat)), 'dq_defer_rx_c64')
    return flat


def permute_bits(amps: torch.Tensor, src_of_dst: Sequence[int], out: torch.Tensor | None = None) -> torch.Tensor:
    """out[b, i] = amps[b, sigma(i)], sigma(i) = sum_p bit_p(i) << src_of_dst[p]: re-label the local qubits
    (destination bit p takes the role of source bit src_of_dst[p]).  Out of place."""
    nl = _nqubit(amps)
    src_of_dst = [int(p) for p in src_of_dst]
    if sorted(src_of_dst) != list(range(nl)):
        raise ValueError('src_of_dst must be a permutation of the local bit positions')
    if out is None:
        out = torch.empty_like(amps)
    if not _use_hip(amps):
        return _test_backend.permute_bits(amps, src_of_dst, out)
    lib = _lib.load()
    fn = getattr(lib, f'dq_permute_bits_{_suffix(amps)}')
    _lib.check(fn(_ptr(amps), _ptr(out), nl, _lib.int_array(src_of_dst), amps.shape[0], _stream(amps)), 'dq_permute_bits')
    return out


def interleave(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(B, N), (B, N) -> (B, 2 N) with out[:, 2 i] = a[:, i], out[:, 2 i + 1] = b[:, i]: psi and the cotangent side by side
    along a new index bit 0, as a fused reverse sweep wants them (include/dq_hip.h, dq_interleave_*).  On the test double
    (CPU tensors): torch.stack."""
    if a.shape != b.shape or a.dtype != b.dtype or a.device != b.device or a.ndim != 2:
        raise ValueError('interleave: two (batch, N) tensors of one dtype on one device')
    if not _use_hip(a) or (a.numel() & 1):
        return torch.stack([a, b], dim=-1).reshape(a.shape[0], -1)
    a, b = a.contiguous(), b.contiguous()
    if (a.data_ptr() | b.data_ptr()) & 15:      # a view at an odd complex64 offset: the kernels read 16-byte pieces
        return torch.stack([a, b], dim=-1).reshape(a.shape[0], -1)
    out = torch.empty(a.shape[0], 2 * a.shape[1], dtype=a.dtype, device=a.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_interleave_{_suffix(a)}')
    _lib.check(fn(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream(a)), 'dq_interleave')
    return out


def deinterleave(pair: torch.Tensor, which: int) -> torch.Tensor:
    """(B, 2 N) -> (B, N): out[:, i] = pair[:, 2 i + which] (dq_deinterleave_*)."""
    if pair.ndim != 2 or pair.shape[1] & 1 or which not in (0, 1):
        raise ValueError('deinterleave: a (batch, 2 N) tensor, which = 0 or 1')
    half = pair.shape[1] // 2
    if not _use_hip(pair) or not pair.is_contiguous() or ((pair.shape[0] * half) & 1) or (pair.data_ptr() & 15):
        return pair.reshape(pair.shape[0], -1, 2)[:, :, which].contiguous()
    out = torch.empty(pair.shape[0], half, dtype=pair.dtype, device=pair.device)
    lib = _lib.load()
    fn = getattr(lib, f'dq_deinterleave_{_suffix(pair)}')
    _lib.check(fn(_ptr(pair), _ptr(out), out.numel(), int(which), _stream(pair)), 'dq_deinterleave')
    return out
