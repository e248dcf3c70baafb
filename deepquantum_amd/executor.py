"""Circuit executor: runs a list of kernel-level gates on a (B, 2**n) state.

* No gradient needed: the gate list is scheduled once into fused passes (``fusion.schedule``, cached by circuit
  structure) and each pass is one ``dq_apply_fused_*`` launch working in place on a private copy of the state (the
  first pass of a batched circuit reads the one shared initial state directly).  States smaller than a tile are
  folded into / padded to one tile so that they run the same passes (``_run_small``).
* Gradient needed: the whole list is ONE autograd node (``_AdjointCircuit``): fused forward, reverse sweep with
  recomputation -- O(1) states of memory instead of the one-state-per-gate of stock autograd, which is how the
  reference differentiates (circuit.py:261).  ``CONFIG['grad_mode'] = 'per_gate'`` keeps one differentiable
  ``ops.apply_gate`` per gate (non-reversible primitives -- channels -- and ``torch.vmap`` always use it).

This replaces the Python loop ``nn.Sequential(self.operators)(x)`` of the reference.
"""

from __future__ import annotations

import functools

from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, Sequence

import torch

from . import _functorch, backend, fusion, ops


@dataclass
class Prim:
    """A kernel-level gate with its matrix tensor: (.., D, D), batch dim optional."""

    kind: str                      # 'gen' | 'x' | 'diag'
    matrix: torch.Tensor
    targets: tuple[int, ...]       # bit positions, matrix MSB first
    controls: tuple[int, ...] = ()
    mode: int = 0                  # 2x2 matrix structure known from the gate class: 0 general, 1 real, 2 Rx-like
    unitary: bool = True           # False for channel superoperators: not reversible, per-gate autograd only
    order: tuple[int, ...] = ()    # bits the gate is ordered on without touching them (fusion.PrimOp.order)
    exact: bool = True             # unitary to the rounding of its precision by construction; False: a user-supplied
    #                                matrix that passed only the reference's 1e-4 check (UAnyGate)


@dataclass
class Plan:
    steps: list                    # fusion.Steps
    prim_ops: list[fusion.PrimOp]
    mat_order: list[int]           # prim indices in the order their matrices lie in the kernel buffer
    mat_total: int                 # complex numbers per batch sample in the flat matrix buffer (incl. tail pad)
    n_fused: int
    n_single: int
    rx_defer: list[int] | None = None      # matrix-buffer offsets of the gates on the deferred Rx handlers
    _rx_index: dict | None = None          # ... as a LongTensor per device
    _scale_cache: dict | None = None       # complex128 reverse sweeps: which scalar gates run before which reduction
    _zero_masks: Any = False               # fusion.zero_state_masks of the steps (False: not computed yet)
    _kz_masks: dict | None = None          # ... for a given mask of index bits known to be |0> at the start (sharded state)


_PLAN_CACHE: OrderedDict = OrderedDict()
_PLAN_CACHE_SIZE = 64

# Tunables (debug / benchmarking); None = library default.
CONFIG = {'fuse': True, 'min_low_c64': None, 'min_low_c128': None,
          'max_gates': None, 'max_far': None, 'far_bit': None,
          # gates + reductions in a pass of a reverse sweep (ABI 24: records in device memory; 72 as everywhere: 0 or None)
          'sweep_max_gates': 104,
          # pass planner (fusion._plan_tiles): beam width / tiles tried per state; 0 = first-come tiles, 1 = greedy
          'plan_width': None, 'plan_branch': None, 'plan_restarts': None,
          # permuted stores also re-label the contiguous low bits, so every pass picks all its tile qubits (None = on)
          'free_low': None,
          # out-of-place passes that write the next pass's qubits to cheap index bits (fusion._place_writes): needs a
          # second state buffer; used when both fit in `permute_mem_frac` of the device memory
          'permute_store': True, 'permute_mem_frac': 0.45, 'permute_min_bits': 20,
          # 'adjoint': fused forward + reverse sweep with recomputation (O(1) states of memory);
          # 'per_gate': one autograd node per gate (saves every intermediate state).  Both differentiate to any order:
          # under create_graph=True the adjoint node replays its gates as per-gate nodes (_backward_with_graph)
          'grad_mode': 'adjoint',
          # reverse sweeps run as fused passes over psi and the cotangent interleaved along one extra index bit, the
          # reductions for the trainable gates folded into the passes (_AdjointCircuit._sweep_fused); False: the
          # undo-then-reduce sweep (always used for trainable gates on two or more targets, and for complex128
          # circuits the wave-tile kernel cannot run)
          'fused_sweep': True,
          # the reductions of the fused sweep form only the sums a trainable gate's gradient can need (2 real sums for a
          # Pauli-X rotation instead of 4 complex ones; A/B switch)
          'reduced_grad_sums': True,
          # a backward that records no graph: ONE sum for a rotation about X (`_grad_records`, DQ_FG_GRAD variant 4)
          'terminal_grad_sums': True,
          # sharded adjoint: all observables of a circuit share ONE reverse sweep (lambda = sum_k g_k O_k psi); False: one
          # sweep and one (psi, lambda) pair per observable, the reference's structure (circuit.py:1706-1738)
          'joint_adjoint': True,
          # no-grad forwards of circuits with Z-type observables take <Z..Z> from the registers of the last pass
          # (DQ_FG_EXPZ records; wave-tile kernel): `expectation()` then costs no read of the state
          'fused_expectation': True,
          # reuse merged runs / plan / matrix buffer while a forward sees the same primitive objects at the same versions
          'steady_cache': True,
          # states smaller than a tile: fuse (batch folded into the index, or zero-padded) from this many gates on
          'small_fuse_min_gates': 6,
          # ... and their reverse sweeps run fused on the zero-padded (psi, lambda) pair (A/B switch; False: undo-then-reduce,
          # a pass per circuit layer and a reduction launch per trainable gate)
          'small_fused_sweep': True,
          # backward of a circuit node under create_graph=True (Hessians): 'tangent' -- the sweep is a node whose own backward
          # runs the tangent circuit (_SweepGrads: two sweeps per Hessian row); 'replay' -- the gates as per-gate nodes
          # (two Python nodes per gate and row; always used where 'tangent' does not apply, and for third order)
          'second_order': 'tangent',
          # no-grad runs on states of at least this many amplitudes (batch included) multiply runs of one-qubit gates
          # on the same qubit into one matrix before planning (merge_one_qubit_runs); None = never
          'merge_min_amps': 1 << 27,
          # from this many amplitudes (batch included) on, the pass planner searches wider (make_plan)
          'plan_big_amps': 1 << 31,
          # a circuit started from its own |0..0>: qubits no pass has had in its tile yet factor out as |0>, and the first
          # passes neither read, compute nor write where such an index bit is 1 (fusion.zero_state_masks,
          # dq_apply_fused_zext_*): the first two of the headline's nineteen passes cost next to nothing, the third only
          # its stores.  A/B switch
          'zero_state': True,
          # a call that sees torch.func wrappers (vmap over the circuit, grad / jacrev around it) runs as ONE node whose
          # vmap rules fold the mapped dimension into the kernels' batch (_FusedCircuit); False, or a transform stack it
          # does not take (forward mode, two grad levels): one node per gate, as in round 4
          'fused_transforms': True,
          # a circuit node keeps its INPUT state for the second-order routes of its backward only if that is free (shared,
          # differentiated) or the state is at most this big; otherwise the routes recompute it from the output
          'keep_input_bytes': 64 << 20}

# When enabled, every fused launch is bracketed by HIP events on the launch stream; bench.py reads
# (start, stop, ngates, bytes read + written) to report the kernel's average duration next to its algorithmic bytes.
PROFILE = {'enabled': False, 'events': []}

# The most recent reverse sweep of _AdjointCircuit: which kind, how many fused passes / reduction records.
LAST_SWEEP = {'fused': False, 'passes': 0, 'reductions': 0, 'with_graph': False}
# How often a backward ran under create_graph=True and took the differentiable per-gate route (tests).
GRAPH_BACKWARDS = {'count': 0, 'tangent_rows': 0}     # backwards under create_graph; rows run by the tangent circuit

# Host time spent in the pass planner (fusion.schedule; once per circuit structure, plans are cached) since import.
PLAN_STATS = {'seconds': 0.0, 'plans': 0}

# Statistics of the most recent fused run (for bench.py and tests).
LAST_RUN = {'passes': 0, 'singles': 0, 'gates': 0, 'rounds': 0, 'transposes': 0, 'permute_folded': False, 'zero_passes': 0}


def _geometry(is128: bool) -> fusion.Geometry:
    """The wave tile of the precision with the overrides of ``CONFIG``."""
    g = fusion.default_geometry(is128)
    ml = CONFIG['min_low_c128'] if is128 else CONFIG['min_low_c64']
    if ml is not None:
        g.min_low = ml
    for key in ('max_gates', 'max_far', 'far_bit', 'plan_width', 'plan_branch', 'plan_restarts'):
        if CONFIG[key] is not None:
            setattr(g, key, CONFIG[key])
    if CONFIG['free_low'] is not None:
        g.free_low = CONFIG['free_low'] if CONFIG['free_low'] == 'force' else bool(CONFIG['free_low'])
    return g


# Steady state of an inference loop: the SAME primitive objects (Gate.prims reuses them while a gate's matrix object is
# unchanged) with matrices at the same versions as last time.  What depends only on them -- the merged one-qubit runs,
# the extra reduction records of fused expectation values, the plan, the flat matrix buffer -- is kept per list of
# primitives instead of being rebuilt per forward (config 2, n = 24: the step is host-bound, 3.0 -> ... ms).
_STEADY: OrderedDict = OrderedDict()
_STEADY_SIZE = 8


def _steady(prims: Sequence[Prim]) -> dict | None:
    """The cache entry of exactly these primitives (created empty if new or if a matrix was written in place);
    None when caching is off or a matrix is not a plain tensor (vmap)."""
    if not CONFIG.get('steady_cache', True) or len(prims) < 16:
        return None
    key = tuple(map(id, prims))
    if any(p.matrix is not None and (torch.is_inference(p.matrix) or p.matrix.requires_grad) for p in prims):
        return None                       # (inference tensors track no version: nothing to key a cache on; matrices of a
                                          #  training step are new every time, and the entry would keep their graphs alive)
    try:
        versions = tuple(-1 if p.matrix is None else p.matrix._version for p in prims)
    except Exception:
        return None
    e = _STEADY.get(key)
    if e is not None and e['versions'] == versions:
        _STEADY.move_to_end(key)
        return e
    e = {'prims': list(prims), 'versions': versions, 'plans': {}, 'flat': {}, 'extra': {}}     # (holds the objects: ids stay theirs)
    _STEADY[key] = e
    if len(_STEADY) > _STEADY_SIZE:
        _STEADY.popitem(last=False)
    return e


def make_plan(prims: Sequence[Prim], n: int, is128: bool, permute: bool = False,
              out_perm: Sequence[int] | None = None, amps: int = 0, steady: dict | None = None,
              final_free: tuple = (), pad_untouched: bool = False, known_zero: int | None = None) -> Plan:
    """``amps`` = amplitudes the plan will be run on (batch included): from ``CONFIG['plan_big_amps']`` on a step
    takes long enough (>= 0.1 s) for a wider search of the pass planner to pay for itself within a few steps
    (measured on the headline: 21 -> 20 passes, -2.7 %, 4.6 s of planning once per circuit structure)."""
    geom = _geometry(is128)
    if CONFIG['max_gates'] is None and CONFIG['sweep_max_gates'] and any(p.kind == 'grad' for p in prims):
        # a reverse sweep: a reduction in front of every trainable gate -- the NUMBER of records bounded its passes, not the
        # tile (n = 28, depth 40: 1120 gates + 420 reductions = 32 passes of at most 72, 23 of at most 104).  Such a pass
        # keeps its records in device memory (backend._device_records)
        geom.max_gates = CONFIG['sweep_max_gates']
    if amps >= CONFIG['plan_big_amps']:
        if CONFIG['plan_width'] is None:
            geom.plan_width = 8
        if CONFIG['plan_branch'] is None:
            geom.plan_branch = 4
        if CONFIG['plan_restarts'] is None:
            geom.plan_restarts = 6
    geom.permute_store = permute
    geom.final_free = tuple(final_free) if permute else ()
    geom.pad_last_with_untouched = bool(pad_untouched)
    geom.known_zero = known_zero
    head = (n, is128, geom.m, geom.slots, geom.min_low, geom.max_gates, geom.max_far, geom.far_bit, geom.plan_width,
            geom.plan_branch, geom.plan_restarts, geom.free_low, permute, CONFIG['fuse'], None if out_perm is None else tuple(out_perm),
            geom.final_free, geom.pad_last_with_untouched, geom.known_zero)
    if steady is not None:
        plan = steady['plans'].get(head)
        if plan is not None and _PLAN_CACHE.get(plan[0]) is plan[1]:     # (still the plan the global cache would give)
            return plan[1]
    key = head + (tuple((p.kind, p.targets, p.controls, p.mode, p.order) for p in prims),)
    plan = _PLAN_CACHE.get(key)
    if plan is not None:
        _PLAN_CACHE.move_to_end(key)
        if steady is not None:
            steady['plans'][head] = (key, plan)
        return plan
    prim_ops, off = [], 0
    for p in prims:
        prim_ops.append(fusion.PrimOp(p.kind, tuple(p.targets), tuple(p.controls), off, p.mode, 0, tuple(p.order)))
        if p.kind not in ('grad', 'expz'):            # (reductions have no matrix)
            off += (1 << len(p.targets)) ** 2
    import time

    t0 = time.perf_counter()
    steps = fusion.schedule(prim_ops, n, geom, fuse=CONFIG['fuse'], final_perm=out_perm)
    PLAN_STATS['seconds'] += time.perf_counter() - t0
    PLAN_STATS['plans'] += 1
    order, total = fusion.layout_matrices(steps, prim_ops)
    plan = Plan(steps, prim_ops, order, total,
                sum(isinstance(s, fusion.FusedStep) for s in steps),
                sum(isinstance(s, fusion.SingleStep) for s in steps),
                rx_defer=fusion.rx_defer_positions(steps, prim_ops), _rx_index={})
    _PLAN_CACHE[key] = plan
    if len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
        _PLAN_CACHE.popitem(last=False)
    if steady is not None:
        steady['plans'][head] = (key, plan)
    return plan


def _flat_mats(prims: Sequence[Prim], order: Sequence[int], batch: int, dtype: torch.dtype,
               device: torch.device) -> tuple[torch.Tensor, int]:
    """Concatenate the gate matrices in the plan's buffer order (fusion.layout_matrices) into one
    (Bm, total) buffer with the kernel's tail pad; Bm = batch if any matrix is batched."""
    batched = any(p.matrix is not None and p.matrix.ndim == 3 and p.matrix.shape[0] > 1 for p in prims)
    bm = batch if batched else 1
    rows = []
    for i in order:
        p = prims[i]
        m = p.matrix
        if m.ndim == 2:
            m = m.unsqueeze(0)
        m = m.reshape(m.shape[0], -1)
        if m.shape[0] != bm:
            m = m.expand(bm, -1)
        rows.append(m)
    rows.append(rows[0].new_zeros(bm, _lib_pad()) if rows else torch.zeros(bm, _lib_pad(), dtype=dtype, device=device))
    flat = torch.cat(rows, dim=1).to(device=device, dtype=dtype).contiguous()
    return flat, (flat.shape[1] if batched else 0)


def _lib_pad() -> int:
    from . import _lib

    return _lib.MAT_PAD


def needs_autograd(state: torch.Tensor, prims: Sequence[Prim]) -> bool:
    """Eager per-gate path: when autograd must see every gate, or inside a ``torch.vmap`` transform (the
    per-gate Function carries the vmap rule; raw pointers of BatchedTensors are not available)."""
    if ops._is_wrapped(state) or any(ops._is_wrapped(p.matrix) for p in prims):
        return True
    if not torch.is_grad_enabled():
        return False
    return state.requires_grad or any(p.matrix is not None and p.matrix.requires_grad for p in prims)


def run(state: torch.Tensor, prims: Sequence[Prim], inplace: bool = False, scratch: torch.Tensor | None = None,
        out_perm: Sequence[int] | None = None, amps: int | None = None, grads: torch.Tensor | None = None,
        expect_z: dict | None = None, zero_state: bool | int = False, need_zeros=None,
        slicing: dict | None = None) -> torch.Tensor:
    """Apply ``prims`` in order to ``state`` (B, 2**n) and return the new (B, 2**n) state.

    ``zero_state``: the caller vouches that ``state`` is |0..0> (every row; ``QubitState.is_zero_state``) -- the first
    passes then skip what is known to be zero (CONFIG['zero_state']).  An int (in-place runs of the sharded state only):
    the mask of the index bits that are known to be |0> in an otherwise arbitrary state whose memory holds real zeros
    there (a shard after its first exchange).

    ``scratch`` (no-grad runs): a second buffer like ``state`` that the passes may ping-pong with (permuted stores
    without an allocation; the sharded state passes its receive buffer) -- the result then lives in ``state`` OR in
    ``scratch``, whichever the last pass wrote.  ``out_perm``: afterwards index bit b sits at position out_perm[b]
    (the re-labelling a shard exchange needs); the last pass writes it if it can, else one extra permute pass.
    ``amps``: amplitudes the plan will be run on in total when ``state`` is only a slice of them (the sample groups of
    the sharded state): the planner's effort goes by the whole.
    ``need_zeros`` (the sharded state, round 6): ``state`` holds garbage where ``zero_state`` says it is zero; the callable
    clears it and is called before anything could read there -- i.e. unless the known-zero masks apply to this schedule
    from its first pass to its last (`fusion.zero_state_masks` vouches that the result is then written completely).
    ``slicing`` (the sharded state's exchange overlap, round 6; in-place runs with ``scratch``): the FIRST and / or the LAST
    pass in slices by index bits outside its tile (`backend.apply_fused(slice_bits=...)`), with a callback per PROTOCOL
    slice -- the 2^B values of B agreed bits: ``{'first': (read positions of the B bits, before(j)), 'last': (positions of
    the B bits AFTER the run, after(j, where))}``.  ``before(j)`` is called once for every j before the first launch that
    reads slice j, ``after(j, where)`` once for every j behind the last launch that writes it (``where``: the buffer the
    result is written to -- ``state`` or ``scratch``); a bit that lies inside the pass's tile
    (or is known zero) is not sliced by: the launches then cover several protocol slices each (none sliceable: ``before``
    for all j, the whole pass, ``after`` for all j).  ``slicing['done']`` reports what happened."""
    if len(prims) == 0 and out_perm is None:
        if need_zeros is not None:
            need_zeros()
        _slicing_all(slicing, 'first')
        _slicing_all(slicing, 'last', state)
        return state
    if state.ndim != 2:
        raise ValueError('state must be (batch, 2**n)')
    if grads is not None:       # a stretch of a reverse sweep ('grad' prims reduce into ``grads``; the sharded state's)
        if need_zeros is not None:
            need_zeros()
        return _run_nograd(state, prims, inplace=inplace, scratch=scratch, out_perm=out_perm, grads=grads, amps=amps)
    if needs_autograd(state, prims):
        if need_zeros is not None:
            need_zeros()
        assert scratch is None and out_perm is None, 'scratch / out_perm are for no-grad runs'
        # (inside a torch.func transform -- vmap, grad, jacrev -- only the per-gate nodes compose)
        vmapped = ops._is_wrapped(state) or any(ops._is_wrapped(p.matrix) for p in prims)
        if CONFIG['grad_mode'] == 'adjoint' and not vmapped and all(p.unitary for p in prims):
            meta = _Meta(((p.kind, tuple(p.targets), tuple(p.controls), p.mode, p.exact) for p in prims), zero_state)
            meta.prims = prims          # (the node's forward runs these very primitives: not one more of each, `_AdjointCircuit.forward`)
            return _AdjointCircuit.apply(state, meta, *[p.matrix for p in prims])
        if vmapped and CONFIG['grad_mode'] == 'adjoint' and CONFIG['fused_transforms'] and _fused_under_transforms(state, prims):
            # torch.vmap over the circuit -- the reference's own batching, circuit.py:232-240 -- and one level of
            # torch.func.grad / vjp / jacrev around it: ONE node with vmap rules of its own (the mapped dimension folds
            # into the kernels' batch), so the gates keep their fused passes
            meta = tuple((p.kind, tuple(p.targets), tuple(p.controls), p.mode, p.exact) for p in prims)
            LAST_RUN['fused_transform_nodes'] = LAST_RUN.get('fused_transform_nodes', 0) + 1
            return _FusedCircuit.apply(state, _Meta(meta, zero_state), *[p.matrix for p in prims])
        x = state
        for p in prims:
            x = ops.apply_gate(x, p.matrix, p.targets, p.controls)
        return x
    if (CONFIG['merge_min_amps'] is not None and CONFIG['fuse'] and state.numel() >= CONFIG['merge_min_amps']
            and not ops._is_batched(state)):
        e = _steady(prims)
        if e is not None and e.get('merged') is not None:
            prims = e['merged']
        else:
            prims = merge_one_qubit_runs(prims)
            if e is not None:
                e['merged'] = prims
    if expect_z is not None and expect_z.get('masks') and CONFIG['fused_expectation'] and CONFIG['fuse'] and out_perm is None:
        # ``expect_z = {'masks': [zmask, ..]}``: the Z strings' expectation values of the FINAL state come out of the last
        # pass (``expect_z['values']``, float64 (B, K)) when the wave-tile kernel runs the circuit; untouched otherwise
        n = state.shape[-1].bit_length() - 1
        is128 = state.dtype == torch.complex128
        g_ = _geometry(is128)
        e = _steady(prims)
        wave_ok = e.get(('wave_ok', is128)) if e is not None else None
        if wave_ok is None:
            wave_ok = fusion.wave_supports([fusion.PrimOp(p.kind, tuple(p.targets), tuple(p.controls), 0, p.mode) for p in prims],
                                           is128)
            if e is not None:
                e[('wave_ok', is128)] = wave_ok
        if (n >= g_.m and len(prims) > 0 and wave_ok and not ops._is_batched(state)
                and state.shape[0] <= backend.MAX_BATCH):
            every = tuple(range(n))
            mkey = (n, tuple(int(z) for z in expect_z['masks']))
            both = e['extra'].get(mkey) if e is not None else None
            if both is None:
                extra = [Prim('expz', None, (), tuple(q for q in range(n) if (int(z) >> q) & 1), r, order=every)
                         for r, z in enumerate(expect_z['masks'])]
                both = list(prims) + extra
                if e is not None:
                    e['extra'][mkey] = both
            nextra = len(expect_z['masks'])
            acc = torch.zeros(state.shape[0], nextra, 8, dtype=torch.float64, device=state.device)
            out = _run_nograd(state, both, inplace, scratch, out_perm, grads=acc, amps=amps, zero_state=zero_state,
                              need_zeros=need_zeros, slicing=slicing)
            expect_z['values'] = acc[:, :, 0]
            return out
    return _run_nograd(state, prims, inplace, scratch, out_perm, amps=amps, zero_state=zero_state, need_zeros=need_zeros,
                       slicing=slicing)


class _Meta(tuple):
    """The gate list of an ``_AdjointCircuit`` node (kind, targets, controls, mode, exact per gate) plus what the caller
    knows about the input state (``zero_state``: it is |0..0>)."""

    zero_state = False
    tangent = False           # a tangent circuit (``_SweepGrads.backward``): trainable gates that are not unitary

    def __new__(cls, items, zero_state: bool = False, tangent: bool = False):
        self = super().__new__(cls, items)
        self.zero_state = bool(zero_state)
        self.tangent = bool(tangent)
        return self


# Issue slots of a wave per one-qubit gate, by matrix structure, measured on the workgroup-tile kernels of round 2 (DESIGN
# 5-r2).  Kept for the wave-tile kernel: with ITS costs (64 / 192 / 128 / 256 packed operations + ~12 slots of dispatch) Hadamard x Rx
# would merge into a general matrix -- 20 % fewer gates, the same arithmetic -- and the step gets 5 % SLOWER (measured,
# one box: 282 vs 268 ms): a general matrix commutes with nothing, so the scheduler loses the freedom the Rx-like factor had.
_MERGE_COST = {3: 30, 2: 37, 1: 45, 0: 79}
SCALAR_MODE = 4          # (host-side only: a product of an even number of Hadamard-like matrices, c I)


_MERGE_CACHE: OrderedDict = OrderedDict()


def _merge_structure(prims: Sequence[Prim]):
    """Which gates of ``prims`` multiply into which product (a function of the circuit structure only; cached):
    (groups [members in order of application, mode of the product], order of the output list)."""
    cost = _MERGE_COST
    key = tuple((p.kind, p.targets, p.controls, p.mode, p.unitary) for p in prims)
    hit = _MERGE_CACHE.get(key)
    if hit is not None:
        _MERGE_CACHE.move_to_end(key)
        return hit
    groups: list[list] = []            # [members (prim indices, in order of application), mode, Hadamard-like members or -1]
    order: list[tuple[str, int]] = []  # ('g', group) | ('p', prim) | ('s', group whose product is a SCALAR: see below)
    last: dict[int, int] = {}          # qubit -> open group
    nothing: dict[int, bool] = {}      # no gate has touched the qubit since the group's last member
    only_x: dict[int, bool] = {}       # ... only X-type actions have
    for i, p in enumerate(prims):
        if p.kind == 'gen' and len(p.targets) == 1 and not p.controls and p.unitary:
            q = p.targets[0]
            g = last.get(q)
            if g is not None:
                mode, nh = groups[g][1], groups[g][2]
                if nothing[q] or (only_x[q] and mode == 2 and p.mode == 2):
                    if nh >= 0 and p.mode == 3:
                        # Hadamard-like times Hadamard-like: (s1 H0)(s2 H0) = 2 s1 s2 I EXACTLY in floating point (equal
                        # products added, equal products subtracted) -- a scalar (SCALAR_MODE) -- and a third factor
                        # makes it Hadamard-like again.  Round 4 applied the pair as a "real" matrix: 128 packed operations
                        # per tile for a multiplication by one number
                        groups[g][0].append(i)
                        groups[g][2] = nh + 1
                        groups[g][1] = 3 if (nh + 1) % 2 else SCALAR_MODE
                        continue
                    if mode == SCALAR_MODE:            # c I times anything: the other factor's structure
                        groups[g][0].append(i)
                        groups[g][1], groups[g][2] = p.mode, -1
                        continue
                    new = 2 if (mode == 2 and p.mode == 2) else 1 if (mode in (1, 3) and p.mode in (1, 3)) else 0
                    if cost[new] <= cost[mode] + cost[p.mode] - 5:
                        groups[g][0].append(i)
                        groups[g][1], groups[g][2] = new, -1
                        continue
            groups.append([[i], p.mode, 1 if p.mode == 3 else -1])
            order.append(('g', len(groups) - 1))
            last[q], nothing[q], only_x[q] = len(groups) - 1, True, True
            continue
        order.append(('p', i))
        x_target = p.kind == 'x' or (p.kind == 'gen' and len(p.targets) == 1 and p.mode == 2)
        for q in p.controls:
            last.pop(q, None)
        for q in p.targets:
            if x_target and q in last:
                nothing[q] = False
            else:
                last.pop(q, None)
    # a product that is a scalar c I is no gate at all: it commutes with everything and belongs to the whole state, so it is
    # multiplied into the matrix of a CARRIER -- the nearest following product or one-qubit gate (any structure survives a
    # real factor), else the nearest one before -- and only a circuit with no such gate at all keeps it as a real matrix
    carriers = [k for k, (kind, idx) in enumerate(order) if kind == 'g' and groups[idx][1] != SCALAR_MODE]
    for k, (kind, idx) in enumerate(order):
        if kind == 'g' and groups[idx][1] == SCALAR_MODE:
            after = [c for c in carriers if c > k]
            before = [c for c in carriers if c < k]
            if after or before:
                order[k] = ('s', idx)
                groups[idx].append(order[after[0] if after else before[-1]][1])       # [3] = the carrier group
            else:
                groups[idx][1] = 1
    # the products with most factors first: at every level of the stacked multiplication the groups still growing are
    # then a PREFIX of the stack (a slice: no index tensor, which would be a host-to-device copy and a stream sync)
    multi = sorted((gi for gi, g in enumerate(groups) if len(g[0]) > 1), key=lambda gi: -len(groups[gi][0]))
    levels: list[list[int]] = []       # levels[l] = the l-th factor of every product that has one, in stack order
    for lv in range(max((len(groups[gi][0]) for gi in multi), default=0)):
        levels.append([groups[gi][0][lv] for gi in multi if len(groups[gi][0]) > lv])
    hit = (groups, order, multi, levels)
    _MERGE_CACHE[key] = hit
    if len(_MERGE_CACHE) > _PLAN_CACHE_SIZE:
        _MERGE_CACHE.popitem(last=False)
    return hit


def merge_one_qubit_runs(prims: Sequence[Prim]) -> list[Prim]:
    """Multiply runs of uncontrolled one-qubit gates on the same qubit into one 2x2 matrix (what qsim / Aer call
    gate fusion, at its smallest): gates with nothing else on the qubit in between, and -- because functions of X
    commute -- all Rx-like gates of a stretch in which the qubit only sees X-type actions (CNOT targets, X).  A
    merge happens only when the product is cheaper for the kernel than its factors (two Hadamards -> one real
    matrix, two Rx -> one Rx-like matrix, anything into a general matrix; NOT Hadamard + Rx -> general).  The
    product matrix is always applied, also when it is (numerically almost) the identity.  The matrices of all
    groups are multiplied level by level in stacked matmuls: ONE stack of all factors, then a matmul and a copy per
    level -- a handful of launches whatever the circuit, no host-device synchronisation."""
    groups, order, multi, levels = _merge_structure(prims)
    if not multi:
        return list(prims)
    members = [i for lv in levels for i in lv]
    bm = max(prims[i].matrix.shape[0] if prims[i].matrix.ndim == 3 else 1 for i in members)

    def mat(i: int) -> torch.Tensor:
        m = prims[i].matrix
        return (m if m.ndim == 3 else m.unsqueeze(0)).expand(bm, 2, 2)

    stack = torch.stack([mat(i) for i in members])                      # (all factors, bm, 2, 2), level by level
    acc = stack[: len(multi)].clone()                                   # (G, bm, 2, 2): the first factors
    off = len(multi)
    for lv in levels[1:]:
        k = len(lv)
        acc[:k] = torch.matmul(stack[off : off + k], acc[:k])           # the later gate multiplies from the left
        off += k

    def batched(g) -> bool:
        return any(prims[i].matrix.ndim == 3 and prims[i].matrix.shape[0] > 1 for i in g[0])

    merged = {gi: acc[k] if batched(groups[gi]) else acc[k, 0] for k, gi in enumerate(multi)}
    # scalar products (an even number of Hadamard-like factors: c I exactly) ride on their carrier's matrix
    scale: dict[int, torch.Tensor] = {}
    for kind, idx in order:
        if kind == 's':
            c = merged[idx][..., 0, 0]
            tgt = groups[idx][3]
            scale[tgt] = c if tgt not in scale else scale[tgt] * c
    out: list[Prim] = []
    for kind, idx in order:
        if kind == 'p':
            out.append(prims[idx])
        elif kind == 'g':
            g = groups[idx]
            first = prims[g[0][0]]
            m = first.matrix if len(g[0]) == 1 else merged[idx]
            c = scale.get(idx)
            if c is not None:
                m = m * (c if c.ndim == 0 else c.reshape(-1, 1, 1))
            if len(g[0]) == 1 and c is None:
                out.append(first)
            else:
                out.append(Prim('gen', m, first.targets, (), g[1]))
    return out


def _run_small(state: torch.Tensor, prims: Sequence[Prim], n: int, m: int) -> torch.Tensor | None:
    """States smaller than a tile (n < m): run the fused passes anyway instead of one launch per gate.
    Shared matrices and a power-of-two batch: the batch index is just more (idle) high qubits of one
    (n + log2 B)-qubit state.  Otherwise every sample is embedded in an m-qubit state |0..0> (x) |psi>
    (zero padding; the pad qubits are never touched), one workgroup per sample."""
    b = state.shape[0]
    batched_mats = any(p.matrix.ndim == 3 and p.matrix.shape[0] > 1 for p in prims)
    if not batched_mats and b & (b - 1) == 0 and n + b.bit_length() - 1 >= m:
        out = _run_nograd(state.detach().reshape(1, -1), prims, inplace=False)
        return out.reshape(b, -1)
    padded = state.new_zeros(b, 1 << m)
    padded[:, : 1 << n] = state.detach()
    out = _run_nograd(padded, prims, inplace=True)
    return out[:, : 1 << n].contiguous()


def _permute_after(x: torch.Tensor, out_perm: Sequence[int], scratch: torch.Tensor | None) -> torch.Tensor:
    """Index bit b -> position out_perm[b] as a pass of its own (destination bit d takes source bit src_of_dst[d])."""
    if list(out_perm) == list(range(len(out_perm))):
        return x
    src_of_dst = [0] * len(out_perm)
    for b, d in enumerate(out_perm):
        src_of_dst[d] = b
    dst = scratch if scratch is not None and scratch.data_ptr() != x.data_ptr() else torch.empty_like(x)
    return backend.permute_bits(x, src_of_dst, out=dst)


def tail_of_last_pass(state: torch.Tensor, prims: Sequence[Prim], amps: int, max_gates: int) -> list[int] | None:
    """The indices (into ``prims``) of the gates the LAST fused pass of `run(state, prims, scratch=...)` would hold, if that
    pass is under-filled -- at most ``max_gates`` kernel gates -- and not the only one; None otherwise.  Structure only (no
    matrices are touched): the same merge of one-qubit runs and the same planner as the run itself.  The sharded state moves
    such a tail behind the exchange that follows the stretch (`distributed._defer_tail`): a pass of 3-7 gates costs a whole
    read and write of the shard."""
    n = state.shape[-1].bit_length() - 1
    is128 = state.dtype == torch.complex128
    if not CONFIG['fuse'] or n < _geometry(is128).m or len(prims) < 2:
        return None
    members = [[i] for i in range(len(prims))]
    struct = list(prims)
    if CONFIG['merge_min_amps'] is not None and state.numel() >= CONFIG['merge_min_amps']:
        groups, order, _multi, _levels = _merge_structure(prims)
        riders: dict = {}                   # carrier group -> the factors of the scalar products that ride on its matrix
        for kind, idx in order:
            if kind == 's':                 # (c I commutes with everything: its factors go wherever their carrier goes)
                riders.setdefault(groups[idx][3], []).extend(groups[idx][0])
        members, struct = [], []
        for kind, idx in order:
            if kind == 's':
                continue
            if kind == 'p':
                members.append([idx])
                struct.append(prims[idx])
            else:
                members.append(list(groups[idx][0]) + riders.get(idx, []))
                struct.append(Prim('gen', None, prims[groups[idx][0][0]].targets, (), groups[idx][1]))
    permute = CONFIG['permute_store'] and n >= CONFIG['permute_min_bits']
    plan = make_plan(struct, n, is128, permute, None, amps=max(state.numel(), amps or 0))
    fused = [st for st in plan.steps if isinstance(st, fusion.FusedStep)]
    if len(fused) < 2 or not isinstance(plan.steps[-1], fusion.FusedStep) or len(fused[-1].ops) > max_gates:
        return None
    return sorted(m for oi in fused[-1].ops for m in members[oi])


def _slicing_all(slicing: dict | None, which: str, where: torch.Tensor | None = None) -> None:
    """Every protocol slice's callback of ``slicing[which]`` (a run, or a pass, that is not sliced); ``where``: the buffer
    that holds the result (``after`` callbacks)."""
    if slicing and which in slicing:
        bits, cb = slicing[which]
        for j in range(1 << len(bits)):
            if which == 'last':
                cb(j, where)
            else:
                cb(j)


def _slice_launches(desc, kz: int, proto_reads: Sequence[int | None]) -> list[tuple[int, int, list[int]]]:
    """The launches of ONE pass cut by the protocol bits that lie outside its tile: [(mask, value, protocol slices the
    launch covers)] -- ``proto_reads[i]`` = READ position of protocol bit i (None: not a tile-number bit on the side that
    matters).  No bit available: one launch, mask 0, covering every protocol slice."""
    tile = set(range(desc.L)) | {desc.high_pos[i] for i in range(desc.h)}
    avail = [(i, r) for i, r in enumerate(proto_reads) if r is not None and r not in tile and not (kz >> r) & 1]
    nb = len(proto_reads)
    mask = sum(1 << r for _, r in avail)
    out = []
    for a in range(1 << len(avail)):
        value = sum(((a >> t) & 1) << r for t, (_, r) in enumerate(avail))
        covered = [j for j in range(1 << nb) if all(((j >> i) & 1) == ((a >> t) & 1) for t, (i, _) in enumerate(avail))]
        out.append((mask, value, covered))
    return out


def _run_nograd(state: torch.Tensor, prims: Sequence[Prim], inplace: bool = False, scratch: torch.Tensor | None = None,
                out_perm: Sequence[int] | None = None, grads: torch.Tensor | None = None,
                amps: int | None = None, zero_state: bool | int = False, need_zeros=None,
                slicing: dict | None = None) -> torch.Tensor:
    """``grads``: the accumulator of the 'grad' prims (the reverse sweep of ``_AdjointCircuit``; complex64, n >= a tile).
    ``zero_state``: ``state`` is |0..0> (see ``run``)."""
    n = state.shape[-1].bit_length() - 1
    with torch.no_grad():
        is128 = state.dtype == torch.complex128
        g_ = _geometry(is128)
        m = g_.m                 # the tile a fused pass runs on
        if len(prims) == 0:
            LAST_RUN['permute_folded'] = False
            if need_zeros is not None:
                need_zeros()
            _slicing_all(slicing, 'first')
            res = _permute_after(state, out_perm, scratch)
            _slicing_all(slicing, 'last', res)
            return res
        if grads is not None:
            assert n >= m and CONFIG['fuse'], 'the fused reverse sweep needs a state of at least one tile'
        if (n < m and CONFIG['fuse'] and len(prims) >= CONFIG['small_fuse_min_gates']
                and all(len(p.targets) <= 2 for p in prims)):
            if need_zeros is not None:
                need_zeros()
            _slicing_all(slicing, 'first')
            out = _run_small(state, prims, n, m)
            out = out if out_perm is None else _permute_after(out, out_perm, scratch)
            _slicing_all(slicing, 'last', out)
            return out
        permute = False
        if scratch is not None:
            assert scratch.shape == state.shape and scratch.dtype == state.dtype and scratch.is_contiguous()
            permute = CONFIG['permute_store'] and n >= CONFIG['permute_min_bits'] and state.is_contiguous()
        elif CONFIG['permute_store'] and not inplace and n >= CONFIG['permute_min_bits']:
            permute = True
            if state.is_cuda:     # the second buffer must fit: a share of the device, and what is free right now
                nbytes = state.numel() * state.element_size()
                free, total = torch.cuda.mem_get_info(state.device)
                free += torch.cuda.memory_reserved(state.device) - torch.cuda.memory_allocated(state.device)
                # what still has to be allocated: the second buffer, and the private working copy unless the caller's
                # state is updated in place (never here: `inplace` runs do not permute)
                permute = 2 * nbytes <= CONFIG['permute_mem_frac'] * total and 2.05 * nbytes <= free
        steady = _steady(prims)
        plan = make_plan(prims, n, is128, permute, out_perm if permute else None, amps=max(state.numel(), amps or 0),
                         steady=steady, final_free=tuple(slicing['last'][0]) if (slicing and 'last' in slicing) else (),
                         pad_untouched=bool(zero_state) and bool(CONFIG['zero_state']),
                         known_zero=int(zero_state) if (zero_state is not True and zero_state and CONFIG['zero_state']) else None)
        # one initial state expanded over the batch (stride 0) and a fused first step: that pass reads the single
        # state directly and writes the B results -- no B materialised copies
        shared_in = None
        # |0..0> in: the masks of the index bits still known to be zero, per step (None: not applicable)
        zmasks = None
        if zero_state and CONFIG['zero_state'] and CONFIG['fuse'] and n >= m and plan.steps:
            if zero_state is True:
                if plan._zero_masks is False:
                    plan._zero_masks = fusion.zero_state_masks(plan.steps, n)
                zmasks = plan._zero_masks
            else:                   # some index bits known to be |0> (the memory there holds real zeros)
                assert inplace, 'a mask of known-zero bits comes with an in-place run'
                if plan._kz_masks is None:
                    plan._kz_masks = {}
                key = int(zero_state)
                if key not in plan._kz_masks:
                    plan._kz_masks[key] = fusion.zero_state_masks(plan.steps, n, key)
                zmasks = plan._kz_masks[key]
            if zmasks is not None and grads is not None and any(
                    zk and any(plan.prim_ops[oi].kind in ('grad', 'expz') for oi in st.ops)
                    for zk, st in zip(zmasks, plan.steps) if isinstance(st, fusion.FusedStep)):
                zmasks = None           # (a reducing pass sums over the whole buffer)
        if need_zeros is not None and (zmasks is None or not (permute and plan.steps.applied_final_perm or out_perm is None)
                                       or not all(isinstance(st_, fusion.FusedStep) for st_ in plan.steps)):
            need_zeros()                # (the masks do not carry this schedule from its first pass to its last: real zeros)
        if (state.shape[0] > 1 and state.stride(0) == 0 and state.stride(1) == 1 and plan.steps
                and isinstance(plan.steps[0], fusion.FusedStep)):
            shared_in = state.detach()[:1]
            x = torch.empty(state.shape, dtype=state.dtype, device=state.device)
        elif zmasks is not None and not inplace and state.is_contiguous():
            # the first pass reads the caller's |0..0> itself -- a few amplitudes of it -- and writes the working buffer: no copy
            shared_in = state.detach()
            x = torch.empty_like(shared_in)
        elif inplace and state.is_contiguous():
            x = state
        else:
            x = state.detach().clone(memory_format=torch.contiguous_format)
        fkey = (id(plan), x.shape[0], x.dtype, x.device)
        cached = steady['flat'].get(fkey) if steady is not None else None
        if cached is not None and cached[0] is plan:
            flat, stride = cached[1], cached[2]
        else:
            flat, stride = _flat_mats(prims, plan.mat_order, x.shape[0], x.dtype, x.device)
            if plan.rx_defer:       # uncontrolled Rx-like gates of complex64 passes: the deferred form (fusion.defer_rx)
                idx = plan._rx_index.get(x.device)
                if idx is None:
                    idx = plan._rx_index[x.device] = backend.own(torch.tensor(plan.rx_defer, dtype=torch.long, device=x.device))
                backend.defer_rx(flat, idx)
            if steady is not None:
                steady['flat'] = {fkey: (plan, backend.own(flat), stride)}       # (one buffer per entry: the latest shape)
        stats = {'passes': 0, 'singles': 0, 'gates': len(prims), 'rounds': 0, 'transposes': 0, 'zero_passes': 0}
        spare = scratch                      # the caller's second buffer (if any)
        other = spare if permute else None
        scratch = None
        # the first / last pass in slices (``slicing``): decided per pass below; whatever cannot be sliced still gets its
        # callbacks -- all ``before`` in front of the pass, all ``after`` behind it
        last_si = len(plan.steps) - 1
        fold_ok = out_perm is None or (permute and plan.steps.applied_final_perm)
        sliced = {'first': 0, 'last': 0}
        if slicing is not None and plan.steps and not isinstance(plan.steps[0], fusion.FusedStep):
            _slicing_all(slicing, 'first')
        for si, st in enumerate(plan.steps):
            if isinstance(st, fusion.FusedStep):
                src, shared_in = (shared_in, None) if shared_in is not None else (x, None)
                kz = zmasks[si] if zmasks is not None else 0
                dst = x
                if st.permutes and src is x:     # writes to other index bits than it reads: the other buffer
                    if other is None:
                        other = torch.empty_like(x)
                    dst = other
                gr = grads if grads is not None and any(plan.prim_ops[oi].kind in ('grad', 'expz') for oi in st.ops) else None
                if gr is not None and src is not x:      # (a reducing pass takes no shared input state: materialise it)
                    x.copy_(src.expand_as(x))
                    src = x
                if kz:
                    stats['zero_passes'] += 1
                launches = [(0, 0, None)]
                before = after = None
                if slicing is not None and (si == 0 or si == last_si):
                    want_first = si == 0 and 'first' in slicing
                    want_last = si == last_si and 'last' in slicing and fold_ok
                    can = gr is None and src is x and x.shape[0] <= backend.MAX_BATCH
                    d = st.desc
                    if want_last:           # the bits are given where they lie AFTER the run: the write side of this pass
                        tile_r = set(range(d.L)) | {d.high_pos[i] for i in range(d.h)}
                        blk = [p_ for p_ in range(d.L, n) if p_ not in tile_r]
                        read_of_write = {d.store_blk_pos[j]: p_ for j, p_ in enumerate(blk)}
                        after = slicing['last'][1]
                        if can:
                            launches = _slice_launches(d, kz, [read_of_write.get(w) for w in slicing['last'][0]])
                            sliced['last'] = len(launches)
                        else:
                            launches = [(0, 0, list(range(1 << len(slicing['last'][0]))))]
                        if want_first:      # (a stretch of ONE pass: everything has to be there before it starts)
                            _slicing_all(slicing, 'first')
                    elif want_first:
                        before = slicing['first'][1]
                        if can:
                            launches = _slice_launches(d, kz, list(slicing['first'][0]))
                            sliced['first'] = len(launches)
                        else:
                            launches = [(0, 0, list(range(1 << len(slicing['first'][0]))))]
                for smask, svalue, covered in launches:
                    if before is not None:
                        for j in covered:
                            before(j)
                    sb = (smask, svalue) if smask else None
                    if PROFILE['enabled'] and x.is_cuda:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        backend.apply_fused(src, flat, stride, st.desc, out=dst, grads=gr, known_zero=kz, slice_bits=sb)
                        e1.record()
                        # bytes the pass has to move: what it reads (not where a known-zero bit is 1) + what it writes (not
                        # the tiles in which such a bit outside the tile is 1); a slice: its share
                        nz = bin(kz).count('1')
                        nzo = nz - sum(1 for i in range(st.desc.h) if (kz >> st.desc.high_pos[i]) & 1)
                        ns = bin(smask).count('1')
                        PROFILE['events'].append((e0, e1, len(st.ops), (((src.numel() >> nz) + (dst.numel() >> nzo)) >> ns) * x.element_size()))
                    else:
                        backend.apply_fused(src, flat, stride, st.desc, out=dst, grads=gr, known_zero=kz, slice_bits=sb)
                    if after is not None:
                        for j in covered:
                            after(j, dst)
                if dst is not x:
                    x, other = dst, x
                stats['passes'] += 1
                stats['rounds'] += st.nrounds
                stats['transposes'] += st.ntranspose
            else:
                op = plan.prim_ops[st.op]
                assert op.kind not in ('grad', 'expz'), 'a reduction can only run inside a fused pass'
                d = 1 << op.k
                mat = flat[:, op.pos : op.pos + d * d].reshape(-1, d, d)
                if op.k <= 4:
                    backend.apply_gate(x, mat, op.targets, op.controls, out=x)
                else:
                    if scratch is None:
                        scratch = torch.empty_like(x)
                    backend.apply_gate(x, mat, op.targets, op.controls, out=scratch)
                    x, scratch = scratch, x
                stats['singles'] += 1
        stats['permute_folded'] = out_perm is not None and permute and plan.steps.applied_final_perm
        stats['plan'] = plan
        LAST_RUN.update(stats)
        if out_perm is not None and not (permute and plan.steps.applied_final_perm):
            x = _permute_after(x, out_perm, other if other is not None else spare)
        if slicing is not None:
            if 'last' in slicing and not (plan.steps and isinstance(plan.steps[-1], fusion.FusedStep) and fold_ok):
                _slicing_all(slicing, 'last', x)     # (the last step was no fused pass, or the re-labelling a pass of its own)
            slicing['done'] = sliced
        return x


def _inverse(kind: str, m: torch.Tensor) -> torch.Tensor:
    """Exact inverse of a (.., D, D) gate matrix (the reference's fixed matrices are float32-rounded, so
    U^dagger is an inverse only to 1e-8: recomputation must not drift)."""
    if kind == 'diag':
        return torch.diag_embed(1.0 / m.diagonal(dim1=-2, dim2=-1))
    if m.shape[-1] == 2:
        a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
        det = a * d - b * c
        return torch.stack([torch.stack([d, -b], dim=-1), torch.stack([-c, a], dim=-1)], dim=-2) / det[..., None, None]
    return torch.linalg.inv_ex(m)[0]


def _inverse_block(m: torch.Tensor) -> torch.Tensor:
    """Inverse of the blocks [[U, 0], [C, U]] of a tangent circuit (``_SweepGrads``): [[V, 0], [-V C V, V]], V = U^-1 --
    closed form for one-target gates (a batched LU of hundreds of 4x4 matrices costs more than the sweep it serves)."""
    d = m.shape[-1] // 2
    v = _inverse('gen', m[..., :d, :d])
    low = -(v @ m[..., d:, :d] @ v)
    return torch.cat([torch.cat([v, torch.zeros_like(v)], dim=-1), torch.cat([low, v], dim=-1)], dim=-2)


def grad_records(kind: str, mode: int, targets: Sequence[int], controls: Sequence[int], row0: int,
                 reduced: bool = True, terminal: bool = False) -> tuple[list[Prim], int]:
    """`_grad_records`, remembered per argument tuple: the records carry no matrix, and a Hessian by rows asks for the same
    two thousand of them in every row (a third of a row's host time was their construction)."""
    reduced = bool(reduced and CONFIG['reduced_grad_sums'])
    recs, cnt = _grad_records(kind, int(mode), tuple(targets), tuple(controls), int(row0), reduced,
                              bool(terminal and reduced and CONFIG['terminal_grad_sums']))
    return list(recs), cnt


@functools.lru_cache(maxsize=1 << 16)
def _grad_records(kind: str, mode: int, targets: tuple, controls: tuple, row0: int, reduced: bool,
                  terminal: bool = False) -> tuple[tuple, int]:
    """The reduction records of a reverse sweep for ONE trainable gate (bit positions of the (psi, lambda) pair: bit 0
    tells the two apart), in front of which they go, and how many accumulator rows they fill from ``row0`` on.

    One target: one DQ_FG_GRAD record, G = sum lambda (x) conj(psi) on the target (with the variant that forms only the
    sums a rotation's gradient needs).  ``terminal``: the cotangent is only ever contracted with dM/dtheta (a backward that
    records no graph).  For a unitary M = a I + i b X that tangent is -i X M / 2, and in <cotangent, dM> = <G, dM M^-1> the
    trace part Re (G00 + G11) drops out: ONE real sum per gate (variant 4) instead of two.  With a graph the value of the
    cotangent itself is differentiated again (d2M/dtheta2 = -M / 4 is radial, not tangent), so the full variant stays.  Two targets (t1, t2): the 4x4 sum in 2x2 blocks over t1, by the values of t2
    in lambda (a2) and psi (b2) -- all from one-target records: a record controlled by t2 gives the block a2 = b2 = 1,
    an uncontrolled one the sum of the two blocks a2 = b2; with t2 flipped on the LAMBDA half (an X controlled by bit
    0, undone afterwards) the same two records give the block a2 = 0, b2 = 1 and the sum of the two blocks a2 != b2.
    A diagonal gate only needs the diagonal: two records, no flips.  `assemble_grad_sums` puts the rows together."""
    t = tuple(targets)
    c = tuple(controls)
    if len(t) == 1:
        variant = 3 if kind == 'diag' else (mode if kind == 'gen' and mode in (1, 2) else 0)
        if not reduced:
            variant = 0
        elif terminal and variant == 2:
            variant = 4
        return (Prim('grad', None, (t[0], 0), c, row0 | (variant << fusion.GRAD_VARIANT_SHIFT)),), 1
    assert len(t) == 2, 'reductions inside the passes: trainable gates on one or two targets'
    v = (3 << fusion.GRAD_VARIANT_SHIFT) if (kind == 'diag' and reduced) else 0
    same = (Prim('grad', None, (t[0], 0), c, row0 | v), Prim('grad', None, (t[0], 0), c + (t[1],), (row0 + 1) | v))
    if kind == 'diag':
        return same, 2
    flip = Prim('x', None, (t[1],), (0,))
    return same + (flip, Prim('grad', None, (t[0], 0), c, row0 + 2), Prim('grad', None, (t[0], 0), c + (t[1],), row0 + 3),
                   Prim('x', None, (t[1],), (0,))), 4


def assemble_grad_sums(g: torch.Tensor, row0: int, kind: str, ntargets: int) -> torch.Tensor:
    """``g``: (.., rows, 2, 2) complex sums of the records of `grad_records` -> the gate's (.., D, D) sum
    lambda (x) conj(psi) (matrix index = 2 * bit(t1) + bit(t2))."""
    if ntargets == 1:
        return g[..., row0, :, :]
    both, one = g[..., row0, :, :], g[..., row0 + 1, :, :]            # blocks a2 = b2: their sum, and a2 = b2 = 1
    out = g.new_zeros(g.shape[:-3] + (4, 4))
    out[..., 0::2, 0::2] = both - one
    out[..., 1::2, 1::2] = one
    if kind != 'diag':
        cross, up = g[..., row0 + 2, :, :], g[..., row0 + 3, :, :]    # a2 != b2: their sum, and a2 = 0, b2 = 1
        out[..., 0::2, 1::2] = up
        out[..., 1::2, 0::2] = cross - up
    return out


class _SweepGrads(torch.autograd.Function):
    """F(gy, state, U_1..U_K) = (U^dagger gy, [sum lambda_j (x) conj(psi_{j-1})]_j): the backward of a circuit node as a node
    of its own, for ``create_graph=True``.  Forward: the (fused) reverse sweep, as in a first-order backward.  Backward --
    one row of a Hessian: F is the gradient of Re<gy, U state>, so its vector-Jacobian product with cotangents (c_0, C_j) is
    the gradient of Re<gy, alpha_K>, alpha_K = d/de U(U_j + e C_j)(state + e c_0) -- the output of the TANGENT CIRCUIT:
    the pair (psi, alpha) as one state with one more qubit on top, gate j as the block matrix [[U_j, 0], [C_j, U_j]] on
    (that qubit, the gate's targets), built from U_j and C_j by tensor operations.  One forward of that circuit gives
    alpha_K (the cotangent of gy), one reverse sweep of it -- with exact inverses, the blocks are not unitary -- the
    cotangents of the state and of every U_j (autograd reads them off the blocks): two circuit nodes per row instead of
    two Python nodes per gate (reference: stock autograd through qmath.py:503-505, one node per gate and order).  Under
    ``create_graph=True`` (third order) the backward differentiates the per-gate formulation instead."""

    @staticmethod
    def takes(meta, mats, need) -> bool:
        # trainable gates: dense or diagonal matrices (a block on one more target must still be a gate the sweeps take)
        return all(not nd or (kind in ('gen', 'diag') and len(t) <= 2) for (kind, t, _c, _m, _e), nd in zip(meta, need, strict=True))

    @staticmethod
    def forward(ctx, gy, state, out, meta, need_state, need, *mats):
        with torch.no_grad():
            gstate, grads = _AdjointCircuit._first_order(out, gy, meta, list(mats), need_state, list(need))
        ctx.meta, ctx.need_state, ctx.need = meta, need_state, need
        ctx.save_for_backward(gy, state, *mats)
        res = ([gstate] if need_state else []) + [g for g in grads if g is not None]
        return tuple(res)

    @staticmethod
    def _replay(gy, state, meta, mats, need_state, need):
        """F by per-gate nodes (differentiable to any order)."""
        mats = [m.view_as(m) if nd else m for m, nd in zip(mats, need, strict=True)]
        wanted = ([state] if need_state else []) + [m for m, nd in zip(mats, need, strict=True) if nd]
        x = state
        for (_kind, targets, controls, _mode, _e), m in zip(meta, mats, strict=True):
            x = ops.apply_gate(x, m, targets, controls)
        return list(torch.autograd.grad(x, wanted, grad_outputs=gy.to(x.dtype), create_graph=True, allow_unused=True))

    @staticmethod
    def backward(ctx, *cots):
        gy, state, *mats = ctx.saved_tensors
        meta, need_state, need = ctx.meta, ctx.need_state, ctx.need
        k = len(mats)
        wants = ctx.needs_input_grad                 # (gy, state, out, meta, need_state, need, *mats)
        if torch.is_grad_enabled():
            # third order and beyond: differentiate the per-gate formulation of F
            with torch.enable_grad():
                # (an alias per slot: one tensor may serve several gates, and every slot gets its own cotangent instead
                # of the sum over all of them)
                mats = [m.view_as(m) for m in mats]
                outs = _SweepGrads._replay(gy, state, meta, mats, need_state, need)
                pairs = [(o, c) for o, c in zip(outs, cots, strict=True) if o is not None and c is not None]
                inputs = [gy, state] + list(mats)
                sel = [i for i, t in enumerate(inputs) if wants[0 if i == 0 else (1 if i == 1 else 6 + i - 2)]]
                got = torch.autograd.grad([o for o, _ in pairs], [inputs[i] for i in sel],
                                          grad_outputs=[c for _, c in pairs], create_graph=True, allow_unused=True)
            full = [None] * len(inputs)
            for i, g in zip(sel, got, strict=True):
                full[i] = g
            return (full[0], full[1], None, None, None, None, *full[2:])
        GRAPH_BACKWARDS['tangent_rows'] += 1
        cots = list(cots)
        c0 = cots.pop(0) if need_state else None
        cmat = [cots.pop(0) if nd else None for nd in need]
        b, dim = state.shape
        n = dim.bit_length() - 1
        dt = state.dtype
        with torch.enable_grad():
            state_l = state.detach().requires_grad_(bool(wants[1]))
            mats_l = [m.detach().requires_grad_(bool(wants[6 + j])) for j, m in enumerate(mats)]
            alpha0 = torch.zeros_like(state) if c0 is None else c0.to(dt).expand_as(state)
            pair = torch.cat([state_l, alpha0], dim=-1)              # index bit n: psi | alpha
            # (a gate without a cotangent block keeps its own `exact` flag: a user matrix that is unitary to 1e-4 only is
            # undone with its inverse here as in the first-order sweep)
            meta2 = [(kind, targets, controls, mode, _e) for kind, targets, controls, mode, _e in meta]
            mats2 = list(mats_l)              # (a gate without a cotangent: the same gate on both halves)
            groups: dict = {}                 # the blocks [[U, 0], [C, U]] of all gates of one shape in a few calls
            for j, ((kind, targets, _c, _m, _e), m) in enumerate(zip(meta, mats_l, strict=True)):
                if cmat[j] is not None:
                    nb = max(m.shape[0] if m.ndim == 3 else 1, cmat[j].shape[0] if cmat[j].ndim == 3 else 1)
                    groups.setdefault((kind == 'diag', m.shape[-1], nb, m.shape, cmat[j].shape, m.dtype, cmat[j].dtype), []).append(j)
            for (diag, d, nb, _su, _sc, _du, _dc), js in groups.items():
                # (ONE stack per group and side: its backward hands every gate its cotangent as a view)
                us = torch.stack([mats_l[j] for j in js]).to(dt).reshape(len(js), -1, d, d).expand(len(js), nb, d, d)
                cs = torch.stack([cmat[j] for j in js]).to(dt).reshape(len(js), -1, d, d).expand(len(js), nb, d, d)
                if diag:                      # (F's output for a diagonal gate has no off-diagonal entries)
                    cs = torch.diag_embed(cs.diagonal(dim1=-2, dim2=-1))
                top = torch.cat([us, torch.zeros_like(us)], dim=-1)
                blk = torch.cat([top, torch.cat([cs, us], dim=-1)], dim=-2)     # on (bit n, the gate's targets)
                parts = (blk if nb > 1 else blk[:, 0]).unbind(0)
                for j, part in zip(js, parts, strict=True):
                    meta2[j] = ('gen', (n,) + tuple(meta[j][1]), meta[j][2], 0, 'block')
                    mats2[j] = part
            out2 = _AdjointCircuit.apply(pair, _Meta(tuple(meta2), tangent=True), *mats2)
            g_gy = out2[:, dim:] if wants[0] else None                        # alpha_K
            leaves = ([state_l] if wants[1] else []) + [m for j, m in enumerate(mats_l) if wants[6 + j]]
            got = []
            if leaves:
                seed = torch.cat([torch.zeros_like(gy, dtype=dt), gy.to(dt)], dim=-1)
                with torch.no_grad():
                    got = list(torch.autograd.grad(out2, leaves, grad_outputs=seed, allow_unused=True))
        g_state = got.pop(0) if wants[1] else None
        g_mats = [got.pop(0) if wants[6 + j] else None for j in range(k)]
        if g_gy is not None:
            g_gy = g_gy.detach().to(gy.dtype)
        return (g_gy, g_state, None, None, None, None, *g_mats)


def _fused_under_transforms(state: torch.Tensor, prims: Sequence[Prim]) -> bool:
    """May a call that sees functorch wrappers run as `_FusedCircuit`?  Only under transform stacks its rules cover:
    any number of ``vmap`` levels and at most TWO differentiation levels -- reverse (``grad`` / ``vjp`` / ``jacrev``) or
    forward (``torch.func.jvp`` / ``jacfwd``, at most one), in any order (plain ``torch.autograd.forward_ad`` keeps the per-gate
    nodes): the second level runs the tangent
    circuit (`_second_order`).  Third order: the per-gate nodes, which differentiate to any order.  Unknown stack: no."""
    stack = ops.transform_stack()
    fwad = ops.forward_ad_active()
    if stack is None or fwad is None:
        return False
    if fwad and 'Jvp' not in stack:          # (plain torch.autograd.forward_ad, no torch.func.jvp around it: the per-gate nodes)
        return False
    levels = sum(t in ('Grad', 'Jvp') for t in stack)
    if any(t not in ('Vmap', 'Grad', 'Jvp') for t in stack) or levels > 2 or sum(t == 'Jvp' for t in stack) > 1:
        return False
    if not all(p.unitary and len(p.targets) <= 2 for p in prims):
        return False
    n = state.shape[-1].bit_length() - 1
    return n + 1 >= _geometry(state.dtype == torch.complex128).m or len(prims) >= CONFIG['small_fuse_min_gates']


def _fold(t: torch.Tensor, dim: int | None, v: int, rows: int, lead: int) -> torch.Tensor:
    """A (possibly mapped) operand as ``v * rows`` consecutive samples: ``t`` has ``lead`` trailing dims that are not
    batch ((2^n,) -> 1, (D, D) -> 2); its own batch dim, if any, is 1 or ``rows``."""
    if dim is not None:
        t = t.movedim(dim, 0)
    else:
        t = t.unsqueeze(0)
    if t.ndim == lead + 1:                       # no batch dim of its own
        t = t.unsqueeze(1)
    tail = t.shape[2:]
    return t.expand(v, rows, *tail).reshape(v * rows, *tail)


def _tangent_circuit(meta, mats, state, c0, dirs):
    """The TANGENT CIRCUIT of (state, U_1 .. U_K) in the direction (c0, C_1 .. C_K) (None = no direction): the pair
    (psi, alpha) as one state with one more qubit on top, alpha_0 = c0, gate j as the block [[U_j, 0], [C_j, U_j]] on (that
    qubit, the gate's targets) -- its output is (U psi, d/de U(U_j + e C_j)(psi + e c0)).  Tensor algebra only (wrappers of
    ``torch.func`` transforms pass through): the blocks of all gates of one shape in a few stack / cat calls.  Returns
    (pair, meta of the tangent circuit, its matrices)."""
    dim = state.shape[-1]
    n = dim.bit_length() - 1
    dt = state.dtype
    alpha0 = torch.zeros_like(state) if c0 is None else c0.to(dt).expand_as(state)
    pair = torch.cat([state, alpha0], dim=-1)                # index bit n: psi | alpha
    meta2 = [(kind, targets, controls, mode, _e) for kind, targets, controls, mode, _e in meta]
    mats2 = list(mats)                   # (a gate without a direction: the same gate on both halves)
    groups: dict = {}
    for j, ((kind, _t, _c, _m, _e), m) in enumerate(zip(meta, mats, strict=True)):
        if dirs[j] is not None:
            nb = max(m.shape[0] if m.ndim == 3 else 1, dirs[j].shape[0] if dirs[j].ndim == 3 else 1)
            groups.setdefault((kind == 'diag', m.shape[-1], nb, tuple(m.shape), tuple(dirs[j].shape)), []).append(j)
    for (diag, d, nb, _su, _sc), js in groups.items():
        us = torch.stack([mats[j] for j in js]).to(dt).reshape(len(js), -1, d, d).expand(len(js), nb, d, d)
        cs = torch.stack([dirs[j] for j in js]).to(dt).reshape(len(js), -1, d, d).expand(len(js), nb, d, d)
        if diag:                         # (a diagonal gate moves along its diagonal only)
            cs = torch.diag_embed(cs.diagonal(dim1=-2, dim2=-1))
        blk = torch.cat([torch.cat([us, torch.zeros_like(us)], dim=-1), torch.cat([cs, us], dim=-1)], dim=-2)
        parts = (blk if nb > 1 else blk[:, 0]).unbind(0)
        for j, part in zip(js, parts, strict=True):
            meta2[j] = ('gen', (n,) + tuple(meta[j][1]), meta[j][2], 0, 'block')
            mats2[j] = part
    return pair, _Meta(tuple(meta2), tangent=True), mats2


def _second_order(gy, state, meta, mats, c0, dirs, want_alpha: bool, want_state: bool, want_mats: Sequence[bool]):
    """With L = Re <gy, U_K .. U_1 state>: the gradient, with respect to (state, U_j), of the derivative of L in the direction
    (c0, C_j) -- one forward of the tangent circuit (alpha_K, which is also the gradient with respect to gy) and one reverse
    sweep of it, read off the blocks.  The Hessian of L is symmetric, so this is both the vector-Jacobian product of the
    sweep node F = grad L (`_FusedSweep.backward`: the direction is the incoming cotangent) and its Jacobian-vector product
    (`_FusedSweep.jvp`: the direction is the incoming tangent).  Returns (alpha_K or None, g_state or None, [g_U_j or None])."""
    dim = state.shape[-1]
    dt = state.dtype
    pair, meta2, mats2 = _tangent_circuit(meta, mats, state, c0, dirs)
    out2 = _FusedCircuit.apply(pair, meta2, *mats2)
    alpha = out2[..., dim:].to(gy.dtype) if want_alpha else None
    g_state, g_mats = None, [None] * len(mats)
    if want_state or any(want_mats):
        seed = torch.cat([torch.zeros_like(gy, dtype=dt), gy.to(dt)], dim=-1)
        mask2 = sum(1 << j for j, w in enumerate(want_mats) if w)
        res = list(_FusedSweep.apply(seed, pair, out2, meta2, bool(want_state), mask2, *mats2))
        if want_state:
            g_state = res.pop(0)[..., :dim]
        for j, w in enumerate(want_mats):
            if not w:
                continue
            g = res.pop(0)
            if dirs[j] is not None:      # U_j sits on both diagonal blocks of [[U, 0], [C, U]]
                d = mats[j].shape[-1]
                g = g[..., :d, :d] + g[..., d:, d:]
                if g.ndim == 3 and g.shape[0] > 1 and (mats[j].ndim == 2 or mats[j].shape[0] == 1):
                    g = g.sum(dim=0, keepdim=mats[j].ndim == 3)      # (a shared U under per-sample directions)
                if meta[j][0] == 'diag':
                    g = torch.diag_embed(g.diagonal(dim1=-2, dim2=-1))
            g_mats[j] = g.reshape(mats[j].shape).to(mats[j].dtype)
    return alpha, g_state, g_mats


class _FusedCircuit(torch.autograd.Function):
    """`_AdjointCircuit` for calls that run under ``torch.func`` transforms: the same fused forward and the same fused
    reverse sweep, written in the ``setup_context`` style with a ``vmap`` rule -- the mapped dimension is folded into the
    kernels' batch dimension and per-sample matrix stride, exactly as `ops._ApplyGate.vmap` does for one gate -- so that
    ``torch.vmap(circuit)`` (the reference's batching mechanism, circuit.py:232-240) and ``jacrev`` / ``grad`` around a
    circuit keep their passes instead of one launch per gate.  The backward is the node `_FusedSweep`, which has the same
    kind of rule (``jacrev`` maps over cotangents).  First order only: `_fused_under_transforms` sends everything else to
    the per-gate nodes."""

    @staticmethod
    def forward(state, meta, *mats):
        prims = [Prim(k, m, t, c, mode) for (k, t, c, mode, _e), m in zip(meta, mats, strict=True)]
        with torch.no_grad():
            if (CONFIG['merge_min_amps'] is not None and CONFIG['fuse'] and state.numel() >= CONFIG['merge_min_amps']):
                prims = merge_one_qubit_runs(prims)
            return _run_nograd(ops._plain(state), prims, zero_state=getattr(meta, 'zero_state', False))

    @staticmethod
    def setup_context(ctx, inputs, output):
        state, meta, *mats = inputs
        ctx.meta = meta
        ctx.kept_input = (ops._is_wrapped(state) or bool(ops.transform_stack()) or ops.forward_ad_active() is not False
                          or _keep_input(state, ctx.needs_input_grad[0]))
        ctx.save_for_backward(state if ctx.kept_input else state.new_empty(0), output, *mats)
        ctx.save_for_forward(state if ctx.kept_input else state.new_empty(0), output, *mats)

    @staticmethod
    def jvp(ctx, state_t, _meta_t, *mats_t):
        # forward mode (torch.func.jvp / jacfwd, forward_ad): the tangent of U state is the tangent circuit's alpha_K -- ONE
        # fused forward on one more qubit, directions (state_t, dU_j)
        ops._single_forward_level()
        state, _out, *mats = ctx.saved_tensors
        assert state.numel() > 0, 'forward mode needs the input state of the node'
        pair, meta2, mats2 = _tangent_circuit(ctx.meta, mats, state, state_t, list(mats_t))
        return _FusedCircuit.apply(pair, meta2, *mats2)[..., state.shape[-1]:].contiguous()      # (a tangent is no view of anything)

    @staticmethod
    def vmap(info, in_dims, state, meta, *mats):
        v = info.batch_size
        rows = state.shape[0] if in_dims[0] is None else state.movedim(in_dims[0], 0).shape[1]
        fstate = _fold(state, in_dims[0], v, rows, 1)
        mdims = in_dims[2:]
        fmats = []
        for m, d in zip(mats, mdims, strict=True):
            if d is None and (m.ndim == 2 or m.shape[0] == 1):
                fmats.append(m)                  # shared by every sample: stays one matrix
            else:
                fmats.append(_fold(m, d, v, rows, 2).contiguous())
        meta2 = meta
        if getattr(meta, 'zero_state', False) and in_dims[0] is not None:
            meta2 = _Meta(tuple(meta), False)    # (only the circuit's own un-mapped |0..0> is vouched for)
        out = _FusedCircuit.apply(fstate, meta2, *fmats)
        return out.reshape(v, rows, out.shape[-1]), 0

    @staticmethod
    def backward(ctx, gy):
        state, out, *mats = ctx.saved_tensors
        need_state = bool(ctx.needs_input_grad[0])
        need = tuple(bool(ctx.needs_input_grad[2 + j]) for j in range(len(mats)))
        wrapped = ops._is_wrapped(gy) or ops._is_wrapped(out) or any(ops._is_wrapped(m) for m in mats)
        if not wrapped:
            if torch.is_grad_enabled() and not ops.transform_stack():
                # plain autograd with create_graph=True on a node that was made under vmap: the differentiable routes
                return _AdjointCircuit._backward_with_graph(ctx, gy)
            terminal = not torch.is_grad_enabled() and not ops.transform_stack()
            with torch.no_grad():
                gstate, grads = _AdjointCircuit._first_order(out, gy.contiguous(), ctx.meta, list(mats), need_state, list(need),
                                                             terminal=terminal)
            return (gstate, None, *grads)
        mask = sum(1 << j for j, nd in enumerate(need) if nd)       # (an int: a leaf for the transforms' pytrees)
        res = list(_FusedSweep.apply(gy, state, out, ctx.meta, need_state, mask, *mats))
        gstate = res.pop(0) if need_state else None
        grads = [res.pop(0) if nd else None for nd in need]
        return (gstate, None, *grads)


class _FusedSweep(torch.autograd.Function):
    """F(gy, state, U_1..U_K) = (U^dagger gy, the matrix cotangents) by the fused reverse sweep, as a node with a ``vmap`` rule:
    ``jacrev`` maps over cotangents -- every basis cotangent becomes a sample of ONE sweep (per-sample matrices, so that every
    sample gets its own matrix cotangents).  ``state`` (the circuit's input) and ``out`` (= U state) ride along: the sweep
    itself starts from ``out``; ``state`` is what the node's own backward needs.

    Its backward -- a second reverse level: ``jacrev(jacrev(f))``, all rows of a Hessian in one traversal -- is the TANGENT
    CIRCUIT of `_SweepGrads` (the pair (psi, alpha) on one more qubit, gate j as the block [[U_j, 0], [C_j, U_j]] of U_j and
    the incoming cotangent C_j), written with the two fused nodes themselves: one `_FusedCircuit` forward gives alpha_K (the
    cotangent of gy), one `_FusedSweep` of it the cotangents of the state and -- read off the blocks -- of every U_j.
    Everything in between is tensor algebra, so the transforms' wrappers pass through (the rows of the outer ``jacrev``
    become samples of both)."""

    @staticmethod
    def forward(gy, state, out, meta, need_state, mask, *mats):
        need = [bool((mask >> j) & 1) for j in range(len(mats))]
        with torch.no_grad():
            gstate, grads = _AdjointCircuit._first_order(ops._plain(out), ops._plain(gy).contiguous(), meta, list(mats),
                                                         need_state, list(need))
        return tuple(([gstate] if need_state else []) + [g for g in grads if g is not None])

    @staticmethod
    def setup_context(ctx, inputs, output):
        gy, state, out, ctx.meta, ctx.need_state, ctx.mask, *mats = inputs
        ctx.save_for_backward(gy, state, out, *mats)
        ctx.save_for_forward(gy, state, out, *mats)

    @staticmethod
    def jvp(ctx, gy_t, state_t, _out_t, _meta_t, _ns_t, _mask_t, *mats_t):
        # forward over reverse (torch.func.hessian = jacfwd(jacrev)): F = grad L is linear in gy, and its derivative in the
        # direction (state_t, dU_j) is -- the Hessian of L being symmetric -- what `backward` computes from cotangents
        ops._single_forward_level()
        gy, state, out, *mats = ctx.saved_tensors
        meta, need_state, mask = ctx.meta, ctx.need_state, ctx.mask
        need = [bool((mask >> j) & 1) for j in range(len(mats))]
        total = None
        if gy_t is not None:
            total = list(_FusedSweep.apply(gy_t, state, out, meta, need_state, mask, *mats))
        if state_t is not None or any(t is not None for t in mats_t):
            assert state.numel() > 0 and not getattr(meta, 'tangent', False), 'forward mode needs the input state of the node'
            _a, g_state, g_mats = _second_order(gy, state, meta, mats, state_t, list(mats_t), False, need_state, need)
            part = ([g_state] if need_state else []) + [g for g, nd in zip(g_mats, need) if nd]
            total = part if total is None else [x + y for x, y in zip(total, part, strict=True)]
        return tuple(total)

    @staticmethod
    def vmap(info, in_dims, gy, state, out, meta, need_state, mask, *mats):
        need = [bool((mask >> j) & 1) for j in range(len(mats))]
        v = info.batch_size
        rows = out.shape[0] if in_dims[2] is None else out.movedim(in_dims[2], 0).shape[1]
        fgy = _fold(gy, in_dims[0], v, rows, 1).contiguous()
        fstate = state if state.numel() == 0 else _fold(state, in_dims[1], v, rows, 1)
        fout = _fold(out, in_dims[2], v, rows, 1).contiguous()
        mdims = in_dims[6:]
        fmats, shared = [], []
        for m, d, nd in zip(mats, mdims, need, strict=True):
            logical = m.ndim - (d is not None)                     # (D, D) or (bm, D, D) as the caller sees it
            bm = 1 if logical == 2 else (m.shape[0] if d is None else m.movedim(d, 0).shape[1])
            one = bm == 1
            if d is None and one and not nd:
                fmats.append(m)
                shared.append(None)
                continue
            # (its cotangent is summed over the rows of a sample, never over the mapped dimension; True: keep a batch dim of 1)
            shared.append((logical != 2) if one else None)
            fmats.append(_fold(m, d, v, rows, 2).contiguous())
        res = list(_FusedSweep.apply(fgy, fstate, fout, meta, need_state, mask, *fmats))
        outs = []
        if need_state:
            g = res.pop(0)
            outs.append(g.reshape(v, rows, g.shape[-1]))
        for nd, keep in zip(need, shared, strict=True):
            if not nd:
                continue
            g = res.pop(0)
            g = g.reshape(v, rows, *g.shape[-2:])
            if keep is not None:
                g = g.sum(dim=1, keepdim=keep)
            outs.append(g)
        return tuple(outs), tuple(0 for _ in outs)

    @staticmethod
    def backward(ctx, *cots):
        gy, state, out, *mats = ctx.saved_tensors
        meta, need_state, mask = ctx.meta, ctx.need_state, ctx.mask
        need = [bool((mask >> j) & 1) for j in range(len(mats))]
        if state.numel() == 0 or getattr(meta, 'tangent', False) or not _SweepGrads.takes(meta, mats, need):
            raise NotImplementedError('deepquantum_amd: this derivative of a circuit node that ran under torch.func transforms is '
                                      'not available on the fused route (third order, or a trainable gate on more than two '
                                      'targets at second order): set executor.CONFIG["fused_transforms"] = False -- the gates '
                                      'then run as per-gate nodes, which differentiate to any order')
        GRAPH_BACKWARDS['tangent_rows'] += 1
        wants = ctx.needs_input_grad                 # (gy, state, out, meta, need_state, mask, *mats)
        cots = list(cots)
        c0 = cots.pop(0) if need_state else None
        cmat = [cots.pop(0) if nd else None for nd in need]
        want_mats = [bool(wants[6 + j]) for j in range(len(mats))]
        g_gy, g_state, g_mats = _second_order(gy, state, meta, mats, c0, cmat, bool(wants[0]), bool(wants[1]), want_mats)
        return (g_gy, g_state, None, None, None, None, *g_mats)


def _keep_input(state: torch.Tensor, differentiated: bool) -> bool:
    return bool(differentiated or state.requires_grad or (state.ndim == 2 and state.stride(0) == 0)
                or state.numel() * state.element_size() <= CONFIG['keep_input_bytes'])


def _input_of(ctx, state: torch.Tensor, out: torch.Tensor, mats: Sequence[torch.Tensor]) -> torch.Tensor:
    """The input state of a circuit node for its second-order routes: the saved one, or -- where the forward did not pin
    it (`_keep_input`) -- recomputed from the output by the exact inverses of the gates in reverse order (fused passes;
    equal to the input to the rounding of the state's precision)."""
    if getattr(ctx, 'kept_input', True):
        return state
    with torch.no_grad():
        prims = []
        for (kind, targets, controls, mode, _e), m in reversed(list(zip(ctx.meta, mats, strict=True))):
            u = m.detach() if m.ndim == 3 else m.detach().unsqueeze(0)
            prims.append(Prim(kind, u if kind == 'x' else _inverse(kind, u.to(out.dtype)), targets, controls, mode))
        return _run_nograd(out.detach(), prims)


class _AdjointCircuit(torch.autograd.Function):
    """y = U_K ... U_1 x for reversible gates as ONE autograd node.  Forward: the fused passes.  Backward: a
    reverse sweep over two states stacked as one batch -- psi_j recomputed with the exact inverses, the
    cotangent lambda_j carried with the adjoints -- and, for every matrix that needs a gradient,
    dL/dU_j = [sum lambda_j (x) conj(psi_j)] U_j^{-dagger} from the gate-gradient kernel.  Memory: two extra
    states whatever the depth (stock autograd, as in the reference, keeps one state per gate); stretches of
    gates between two trainable ones are undone by fused passes."""

    @staticmethod
    def forward(ctx, state, meta, *mats):
        prims = meta.__dict__.pop('prims', None)       # (the caller's primitives, for this one use: the node must not hold them)
        if prims is None or len(prims) != len(mats) or any(p.matrix is not m for p, m in zip(prims, mats)):
            prims = [Prim(k, m, t, c, mode) for (k, t, c, mode, _e), m in zip(meta, mats, strict=True)]
        # the sweep recomputes from `out` gate by gate, so the forward itself may run the merged gate list
        if (CONFIG['merge_min_amps'] is not None and CONFIG['fuse'] and state.numel() >= CONFIG['merge_min_amps']):
            with torch.no_grad():
                prims = merge_one_qubit_runs(prims)
        out = _run_nograd(state, prims, zero_state=getattr(meta, 'zero_state', False))
        ctx.meta = meta
        # The input is kept for the second-order route of ``backward`` only (the sweep itself needs ``out`` alone) -- and
        # only where keeping it is free or needed: a shared (stride-0) or small state, or one the caller differentiates
        # with respect to.  A big input of somebody else's -- an amplitude-encoded batch, the state behind a Reset -- would
        # be pinned for the whole first-order training step (2-4 GB at n >= 28); the second-order route
        # recomputes it from ``out`` with the exact inverses instead (`_input_of`).
        ctx.kept_input = _keep_input(state, ctx.needs_input_grad[0])
        ctx.save_for_backward(state if ctx.kept_input else state.new_empty(0), out, *mats)
        return out

    @staticmethod
    def _backward_with_graph(ctx, gy):
        """``backward`` under ``create_graph=True`` (grad mode is on while the engine runs us): the cotangents must be
        differentiable functions of ``gy``, the input state and the matrices, which the recomputing sweep is not.  The
        gates are replayed from the saved input as per-gate nodes -- each of them differentiable to any order
        (``ops._ApplyGate`` / ``ops._GateGrad``) -- and differentiated there; this is the reference's own cost model
        (stock autograd, one state per gate, qmath.py:503-505) and is only paid when a graph of the backward is asked
        for: Hessians (examples/benchmarks/gradient_benchmark.py:147-163), gradient penalties.  First-order
        ``backward()`` never comes here."""
        state, out, *mats = ctx.saved_tensors
        state = _input_of(ctx, state, out, mats)
        meta = ctx.meta
        need = tuple(bool(ctx.needs_input_grad[2 + j]) for j in range(len(mats)))
        if CONFIG['second_order'] == 'tangent' and _SweepGrads.takes(meta, mats, need) and not getattr(meta, 'tangent', False):
            # the sweep as a node of its own: its value by the fused sweep, ITS backward -- what a Hessian row costs -- by
            # one forward and one reverse sweep of the tangent circuit instead of two Python nodes per gate
            res = _SweepGrads.apply(gy, state, out, meta, bool(ctx.needs_input_grad[0]), need, *mats)
            LAST_SWEEP.update(with_graph=True)
            GRAPH_BACKWARDS['count'] += 1
            res = list(res)
            gstate = res.pop(0) if ctx.needs_input_grad[0] else None
            grads = [res.pop(0) if nd else None for nd in need]
            return (gstate, None, *grads)
        with torch.enable_grad():
            # (an alias per slot: one tensor may serve several gates, and every slot gets its own gate's cotangent)
            mats = [m.view_as(m) if ctx.needs_input_grad[2 + j] else m for j, m in enumerate(mats)]
            wanted = [state] if ctx.needs_input_grad[0] else []
            wanted += [m for j, m in enumerate(mats) if ctx.needs_input_grad[2 + j]]
            x = state
            for (_kind, targets, controls, _mode, _e), m in zip(meta, mats, strict=True):
                x = ops.apply_gate(x, m, targets, controls)
            got = list(torch.autograd.grad(x, wanted, grad_outputs=gy.to(x.dtype), create_graph=True, allow_unused=True))
        LAST_SWEEP.update(fused=False, passes=0, reductions=0, with_graph=True)
        GRAPH_BACKWARDS['count'] += 1
        gstate = got.pop(0) if ctx.needs_input_grad[0] else None
        grads = [got.pop(0) if ctx.needs_input_grad[2 + j] else None for j in range(len(mats))]
        return (gstate, None, *grads)

    @staticmethod
    def backward(ctx, gy):
        if _functorch.is_legacy_batched(gy) or ops._is_wrapped(gy):
            raise RuntimeError('deepquantum_amd: a batch of cotangents reached the circuit node (is_grads_batched / '
                               'torch.autograd.functional.*(vectorize=True)): its sweep runs on raw buffers.  Use '
                               'torch.func.jacrev / torch.func.vmap over the function instead -- inside those transforms '
                               'the gates run as per-gate nodes with vmap rules.')
        if torch.is_grad_enabled():
            return _AdjointCircuit._backward_with_graph(ctx, gy)
        _state, out, *mats = ctx.saved_tensors
        need = [ctx.needs_input_grad[2 + j] for j in range(len(mats))]
        gstate, grads = _AdjointCircuit._first_order(out, gy, ctx.meta, mats, ctx.needs_input_grad[0], need, terminal=True)
        return (gstate, None, *grads)

    @staticmethod
    def _first_order(out, gy, meta, mats, need_state, need, terminal=False):
        """The reverse sweep from the saved output: (cotangent of the input state or None, [cotangent of every matrix or
        None]).  No graph is built (``backward`` of the circuit node; forward of ``_SweepGrads``).  ``terminal``: nothing will
        differentiate the result again (`grad_records`)."""
        b = out.shape[0]
        # Inverses / adjoints of ALL gates in a few vectorised calls (grouped by kind, size and batchness): at
        # launch-bound sizes a handful of tiny kernels per gate would dominate the whole sweep.
        groups: dict = {}
        tangent = getattr(meta, 'tangent', False)     # (a tangent circuit's blocks are trainable AND not unitary)
        for j, ((kind, _t, _c, _mode, exact), m) in enumerate(zip(meta, mats, strict=True)):
            u = m if m.ndim == 3 else m.unsqueeze(0)
            # (the blocks [[U, 0], [C, U]] of a tangent circuit: a group of their own, inverted block-wise)
            groups.setdefault((kind, u.shape[-1], u.shape[0]) + (('block',) if tangent and exact == 'block' else ()), []).append((j, u))
        undo: list = [None] * len(mats)       # (2b, D, D): rows [0, b) the inverse, rows [b, 2b) the adjoint
        inv_h: dict = {}                      # group key -> (positions, inverse^dagger stack) for the gradients
        is128 = out.dtype == torch.complex128
        corr: dict = {}
        for key, members in groups.items():
            kind, d, nb = key[:3]
            us = torch.stack([u for _, u in members]).to(out.dtype)           # (K, nb, D, D)
            inv = us if kind == 'x' else (_inverse_block(us) if len(key) > 3 else _inverse(kind, us))
            both = torch.cat([inv.expand(-1, b, d, d), us.mH.expand(-1, b, d, d)], dim=1).contiguous()
            for k, (j, _u) in enumerate(members):
                undo[j] = both[k]
            # U^dagger U: what the fused sweep multiplies lambda by after U^-1 (complex128; user matrices in any precision)
            if kind != 'x' and (is128 or any(meta[j][4] is not True for j, _ in members)):
                cs = us.mH @ us
                for k, (j, _u) in enumerate(members):
                    corr[j] = cs[k]
            if any(need[j] for j, _ in members):
                inv_h[key] = ({j: k for k, (j, _u) in enumerate(members)}, inv.mH.to(torch.complex128))

        n = out.shape[-1].bit_length() - 1
        g_ = _geometry(is128)
        # the sweep as fused passes: every gate one the pass kernel takes (at most two targets), the (psi, lambda) pair at
        # least a tile, every trainable gate on one target
        fusable = all(len(targets) <= 2 for _k, targets, _c, _m, _e in meta)
        # (a pair smaller than a tile is zero-padded to one, like the forward of such a state (`_run_small`): one launch
        # with the reductions inside instead of a pass per circuit layer and a reduction kernel per trainable gate)
        fused = (CONFIG['fused_sweep'] and CONFIG['fuse'] and fusable and b <= backend.MAX_BATCH
                 and (n + 1 >= g_.m or (CONFIG['small_fused_sweep'] and len(meta) >= CONFIG['small_fuse_min_gates'])))
        tangent = getattr(meta, 'tangent', False)     # (a tangent circuit's blocks are trainable AND not unitary)
        if fused:
            # complex128: a matrix that is not computed from parameters or data may be unitary only to float32 rounding
            # (the reference's fixed matrices are, after .to(torch.double)): the sweep then tells U^-1 from U^dagger
            # -- and a user-supplied matrix (UAnyGate: unitary to 1e-4 is all the reference asks) in any precision
            inexact = [j in corr and ((not need[j] and m.grad_fn is None and not m.requires_grad
                                       and (is128 or meta[j][4] is not True)) or (tangent and meta[j][4] == 'block'))
                       for j, m in enumerate(mats)]
            raw, lam = _AdjointCircuit._sweep_fused(out, gy, meta, undo, need, b,
                                                    [m.ndim == 2 or m.shape[0] == 1 for m in mats], corr, inexact,
                                                    terminal=terminal and not tangent)
        else:
            raw, lam = _AdjointCircuit._sweep_undo_reduce(out, gy, meta, undo, need, b)

        grads: list = [None] * len(mats)
        for key, (pos, ih) in inv_h.items():
            kind, d, nb = key[:3]
            js = [j for j in pos if need[j]]
            ks = [pos[j] for j in js]
            ihs = ih if ks == list(range(ih.shape[0])) else torch.stack([ih[k] for k in ks])   # no host index tensors
            g = torch.stack([raw[j] for j in js]) @ ihs                              # (K', b, D, D)
            if kind == 'diag':
                g = torch.diag_embed(g.diagonal(dim1=-2, dim2=-1))      # the kernels ignore off-diagonal entries
            if nb == 1 and b > 1:
                g = g.sum(dim=1, keepdim=True)
            if all(mats[j].dtype == mats[js[0]].dtype for j in js):
                g = g.to(mats[js[0]].dtype)         # one conversion for the group, not one per gate
            for k, j in enumerate(js):
                grads[j] = g[k].to(mats[j].dtype).reshape(mats[j].shape)
        gstate = lam() if need_state else None
        return gstate, grads

    @staticmethod
    def _sweep_undo_reduce(out, gy, meta, undo, need, b):
        """Reverse sweep over psi and lambda stacked as one batch: gates are undone lazily (fused passes) and a
        trainable gate's sum lambda (x) conj(psi) is reduced from a snapshot in which no pending gate touches its
        qubits (gate-gradient kernels).  Returns ({gate: (b, D, D) sums}, thunk for lambda_0)."""
        work = torch.cat([out, gy.to(out.dtype)]).contiguous()        # rows [0, b): psi, rows [b, 2b): lambda
        LAST_SWEEP.update(fused=False, passes=0, reductions=sum(need), with_graph=False)
        raw: dict = {}                                                  # j -> sum lambda_j (x) conj(psi_j)
        pending: list[Prim] = []
        touched: set[int] = set()             # qubits the not-yet-undone gates act on

        def flush():
            nonlocal work
            if pending:
                work = _run_nograd(work, pending, inplace=True)
                pending.clear()
                touched.clear()

        waiting: list[int] = []               # single-target trainable gates whose snapshot is the current `work`

        def reduce_waiting():
            # all of them in as few reads of the two states as the multi-gate kernel allows
            if waiting:
                g = backend.gate_grad_multi(work[:b], work[b:], [(meta[j][1][0], meta[j][2]) for j in waiting])
                for k, j in enumerate(waiting):
                    raw[j] = g[:, k]
                waiting.clear()

        for j in range(len(meta) - 1, -1, -1):
            kind, targets, controls, mode, _exact = meta[j]
            mine = set(targets) | set(controls)
            if need[j]:
                # sum lambda (x) conj(psi) reduced to this gate's qubits does not change when BOTH states are
                # pulled back through gates on other qubits (psi with U^-1, lambda with U^dagger: the partial
                # trace sees U^dagger U^-dagger = 1 exactly), so the snapshot is stale only if a pending gate
                # shares a qubit with this one.  On layered circuits this is one flush per layer, not per gate.
                if mine & touched:
                    reduce_waiting()
                    flush()
                if len(targets) == 1:
                    waiting.append(j)
                else:
                    raw[j] = backend.gate_grad(work[:b], work[b:], targets, controls)
            pending.append(Prim(kind, undo[j], targets, controls, mode))
            touched |= mine
        reduce_waiting()
        flush()
        return raw, lambda: work[b:].clone()

    @staticmethod
    def _sweep_fused(out, gy, meta, undo, need, b, shared, corr=None, inexact=None, terminal=False):
        """The same sweep as ONE gate list run in fused passes: psi and lambda are interleaved along an
        extra index bit 0 -- every thread of a pass then holds both halves of an amplitude pair of both states -- every
        gate's adjoint acts on both at once (adjoint = inverse to the rounding of the state's precision, the precision psi
        is recomputed in anyway), and in front of the adjoint of every trainable gate a 'grad' prim reduces sum lambda (x) conj(psi) on the
        gate's target inside the pass that holds the qubit (DQ_FG_GRAD, include/dq_hip.h): no snapshots, no separate
        reduction kernels, a pass per ~80 records instead of two per circuit layer.  complex128 holds the 1e-10 bar
        against gates that are unitary only to float32 rounding (``inexact``): those are undone with the exact inverse
        on both halves, and lambda -- the half with bit 0 set -- is multiplied by U^dagger U afterwards (``corr``, a
        gate controlled by bit 0): U^dagger lambda exactly, psi never drifts."""
        work = backend.interleave(out.reshape(b, -1), gy.to(out.dtype).reshape(b, -1))        # bit 0: psi | lambda
        pair = work.shape[-1]
        tile = 1 << _geometry(out.dtype == torch.complex128).m
        if pair < tile:                   # launch-bound sizes: |0..0> (x) the pair, the pad qubits are never touched
            padded = work.new_zeros(b, tile)
            padded[:, :pair] = work
            work = padded
        rows: dict[int, int] = {}         # gate -> first accumulator row of its reduction records
        nrows = 0
        prims: list[Prim] = []
        scalars: dict[int, int] = {}      # prim index -> gate, for the gates whose U^dagger U is a scalar != 1
        grad_at: dict[int, int] = {}      # prim index of a reduction -> its row
        variants: dict[int, int] = {}     # row -> which sums its record forms (DQ_FG_GRAD `loc`)
        for j in range(len(meta) - 1, -1, -1):
            kind, targets, controls, mode, _exact = meta[j]
            t1, c1 = tuple(t + 1 for t in targets), tuple(c + 1 for c in controls)
            if need[j]:
                # (which sums this gate's gradient can need -- DQ_FG_GRAD variants, include/dq_hip.h: a matrix the gate
                # class promises to be real / of the form a I + i b X / diagonal has no gradient component in the
                # others; a gate on two targets takes four one-target records, `grad_records`)
                rows[j] = nrows
                # (one sum instead of two for a rotation about X: only a gate that IS unitary to the working precision -- a
                # matrix computed from parameters, `exact` -- and only when nothing differentiates the cotangent again)
                recs, cnt = grad_records(kind, mode, t1, c1, nrows, terminal=terminal and _exact is True and not (
                    inexact is not None and inexact[j]))
                for q in recs:
                    if q.kind == 'grad':
                        grad_at[len(prims)] = q.mode & fusion.GRAD_ROW_MASK
                        variants[q.mode & fusion.GRAD_ROW_MASK] = q.mode >> fusion.GRAD_VARIANT_SHIFT
                    prims.append(q)
                nrows += cnt
            if (inexact is not None and inexact[j] and kind == 'gen' and mode == 3 and not controls and shared[j]
                    and len(targets) == 1):
                # Hadamard-like, s [[1, 1], [1, -1]] with 2 s^2 = 1 only to float32 rounding: U^dagger U = 2 s^2 exactly,
                # a SCALAR.  The adjoint goes over both halves as in complex64; psi is then 2 s^2 times what it should
                # be from here on, and so is every sum reduced later in the sweep: divided out below, by the order the
                # plan executes things in.  No correction gate, no ordering constraint.
                scalars[len(prims)] = j
                prims.append(Prim(kind, undo[j][b : b + 1], t1, c1, mode))
                continue
            if inexact is not None and inexact[j]:
                # (ordered on the psi / lambda bit like its correction: no reduction may come between the two)
                prims.append(Prim(kind, undo[j][0:1] if shared[j] else undo[j][:b], t1, c1, 0 if kind == 'gen' else mode,
                                  order=(0,)))
                prims.append(Prim(kind, corr[j][0:1] if shared[j] else corr[j].expand(b, -1, -1), t1, c1 + (0,), 0))
                continue
            prims.append(Prim(kind, undo[j][b : b + 1] if shared[j] else undo[j][b:], t1, c1, mode))
        acc = torch.zeros(b, max(nrows, 1), 8, dtype=torch.float64, device=out.device)
        scratch = None                    # the partner buffer of the permuted stores, when it fits what is free now
        if work.is_cuda:
            free, _total = torch.cuda.mem_get_info(work.device)
            free += torch.cuda.memory_reserved(work.device) - torch.cuda.memory_allocated(work.device)
            if 1.05 * work.numel() * work.element_size() <= free:
                scratch = torch.empty_like(work)
        work = _run_nograd(work, prims, inplace=True, scratch=scratch, grads=acc)
        LAST_SWEEP.update(fused=True, passes=LAST_RUN['passes'], reductions=nrows, with_graph=False,
                          variants=tuple(sorted(set(variants.values()))))
        if CONFIG.get('check_grad_rows'):
            # (tests) the ABI's promise for reduced records: the components a variant does not form are left untouched --
            # zero, since the accumulator was zeroed -- which is what lets `_first_order` multiply whole rows by U^-dagger
            keep = {0: range(8), 1: (0, 2, 4, 6), 2: (0, 3), 3: (0, 1, 6, 7), 4: (3,)}       # (Re, Im) of G00 G01 G10 G11
            for r, v_ in variants.items():
                idle = [c for c in range(8) if c not in keep[v_]]
                if idle and float(acc[:, r, idle].abs().max()) != 0.0:
                    raise AssertionError(f'DQ_FG_GRAD variant {v_}, row {r}: components {idle} were written to')
            LAST_SWEEP['checked_rows'] = len(variants)
        g = torch.view_as_complex(acc.reshape(b, -1, 4, 2)).reshape(b, -1, 2, 2)
        if scalars and rows:
            # row r was reduced from a psi that is  prod 2 s_k^2  (over the scalar gates executed before it) too large
            plan = LAST_RUN['plan']
            key = tuple(scalars)
            before = plan._scale_cache.get((key, work.device)) if plan._scale_cache is not None else None
            if before is None:
                rank = {}
                for st in plan.steps:
                    for oi in (st.ops if isinstance(st, fusion.FusedStep) else [st.op]):
                        rank[oi] = len(rank)
                sc = sorted(scalars)
                bmat = torch.zeros(max(nrows, 1), len(sc), dtype=torch.float64)
                for pi, r in grad_at.items():
                    for k, si in enumerate(sc):
                        if rank[si] < rank[pi]:
                            bmat[r, k] = 1.0
                before = backend.own(bmat.to(work.device))
                if plan._scale_cache is None:
                    plan._scale_cache = {}
                plan._scale_cache[(key, work.device)] = before
            backend.pin_if_capturing(before)
            s00 = torch.stack([undo[scalars[si]][b, 0, 0] for si in sorted(scalars)])      # s of every scalar gate
            logc = torch.log(2.0 * (s00.real * s00.real + s00.imag * s00.imag))
            g = g / torch.exp(before @ logc)[None, :, None, None]
        raw = {}
        # dense gates on two targets whose records lie four rows apart (every block of a tangent circuit, the Rxx layers of
        # an ansatz): assembled together -- a handful of launches for all of them instead of six per gate
        two = [(j, r) for j, r in rows.items() if len(meta[j][1]) == 2 and meta[j][0] != 'diag']
        if len(two) >= 4 and all(r == two[0][1] + 4 * k for k, (_j, r) in enumerate(two)):
            r0, cnt = two[0][1], len(two)
            g4 = g[:, r0 : r0 + 4 * cnt].reshape(b, cnt, 4, 2, 2)
            both, one, cross, up = g4[:, :, 0], g4[:, :, 1], g4[:, :, 2], g4[:, :, 3]
            full = g.new_zeros(b, cnt, 4, 4)
            full[..., 0::2, 0::2] = both - one
            full[..., 1::2, 1::2] = one
            full[..., 0::2, 1::2] = up
            full[..., 1::2, 0::2] = cross - up
            for k, (j, _r) in enumerate(two):
                raw[j] = full[:, k]
        for j, r in rows.items():
            if j not in raw:
                raw[j] = assemble_grad_sums(g, r, meta[j][0], len(meta[j][1]))
        return raw, lambda: backend.deinterleave(work[:, :pair] if work.shape[1] != pair else work, 1)
