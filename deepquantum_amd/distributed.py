"""Gate application on the index-bit-partitioned statevector (one shard per GPU / process).

Layout (same as the reference, state.py:342-383): with W = 2^g ranks and L = n - g local qubits, rank r
owns the amplitudes whose top g index bits equal r; index bit p < L is "local", bit p >= L is rank bit
p - L ("global").  Semantics follow arXiv:2311.01512 Alg. 6-10 as the reference implements them
(distributed.py:15-202), but the data path is different:

* local work goes through the same HIP kernels as the single-GPU path, and consecutive gates that need
  no exchange are fused into HBM passes by ``executor.run`` (the reference runs one permute/matmul or
  one arange+mask gather per gate);
* a control on a global qubit is a per-rank predicate and never moves data; a diagonal gate on a global
  qubit is a per-rank phase and never moves data either (the reference exchanges the full shard for,
  e.g., Rz on a global qubit);
* the sub-cube that has to travel is gathered / scattered by the pack kernels instead of boolean-mask
  indexing, and complex amplitudes cross the wire as interleaved reals through one
  ``all_to_all_single`` with a single non-zero split (RCCL send/recv pair over one xGMI link).

Every rank executes the same sequence of exchange steps (ranks with nothing to move take part with an
empty message), which keeps the collective matched exactly like the reference does
(distributed.py:89-96).

Two execution modes (``CONFIG['mode']``):

* ``'pairwise'`` -- reference semantics gate by gate: a non-diagonal gate on a global qubit costs one
  pairwise exchange of (part of) the shard over ONE xGMI link (Alg. 6-10).  Used for short gate lists
  (single ``Gate.forward`` calls, the adjoint sweep).
* ``'remap'`` (default for whole circuits) -- the state carries a logical->physical qubit permutation.
  When a stretch of gates needs a qubit that currently is a rank bit, ONE all-to-all swaps k global
  qubits with the k local qubits whose next non-diagonal use lies farthest in the future (Belady),
  each GPU sending 1/2^k of its shard to each of its 2^k - 1 group peers concurrently -- all xGMI links
  at once -- and the following gates run as fused local passes until a rank bit is needed again.  The
  canonical layout is restored before anything reads ``state.amps``.
"""

from __future__ import annotations

from collections import Counter
from dataclasses import replace
from typing import Sequence

import torch
import torch.distributed as dist

from . import backend, executor
from .bitmath import get_bit
from .communication import comm_exchange_arrays, exchange_chunks, exchange_pieces
from .executor import Prim
from .qmath import block_sample, measure
from .state import DistributedQubitState


#: 'remap' | 'pairwise'; gate lists shorter than ``remap_min_prims`` always run pairwise
#: ``overlap_groups``: a batched shard is cut into this many groups of samples; group g's exchange (RCCL, its own
#: HIP stream) runs while group g + 1 still computes its local passes.  ``fold_permute``: the re-labelling of the
#: local qubits that an exchange needs is written by the last fused pass before it instead of a pass of its own.
CONFIG = {'mode': 'remap', 'remap_min_prims': 8, 'horizon': 1 << 14, 'overlap_groups': 4, 'fold_permute': True,
          # 'remap' mode: gates are first re-ordered along the commutation DAG so that everything that is local under the
          # current placement runs before the next exchange (_order_for_remaps)
          'reorder': True,
          # an UN-BATCHED shard treats its top ``virtual_bits`` local index bits as rank bits of a virtual world: the shard
          # becomes 2^v rows that move through a remap like the samples of a batch -- row r's amplitudes are on the links
          # while row r + 1 runs its passes -- at the price of extra stretches: a gate that targets a virtual bit needs a
          # (local, free) re-labelling of the shard first (`_remap_virtual`).  0 = off.  Forward circuits in 'remap' mode.
          # None = 2 where an exchange can overlap with compute at all -- RCCL on device shards (asynchronous, on the
          # group's own stream) -- and 0 elsewhere (gloo: synchronous and host-staged, the extra stretches would only cost)
          'virtual_bits': None,
          # REHEARSAL of one rank of a world that is not there (bench.py --rehearse-rank; `DistributedQubitState.REHEARSE`):
          # every exchange and every all-reduce is left out -- the receive buffer keeps whatever it held, so the amplitudes
          # are meaningless -- while schedule, passes, re-labellings and streams are exactly this rank's: the timing of the
          # kernels does not depend on the data, so the compute half of a multi-GPU step can be MEASURED on one GPU
          'elide_exchange': False,
          # ... and with 'loopback' as its value the slices of a sliced exchange (`_remap_sliced`) are COPIED from the send
          # buffer to the receive buffer on the exchange stream instead of being left out: the same bytes cross this GPU's HBM
          # as when they leave for / arrive from the peers (only faster than any link), so the rehearsal measures what the
          # hidden wire costs the passes it runs beside (amplitudes still meaningless)
          # a circuit that starts from reset() picks its FIRST qubit placement freely (`initial_placement`: |0..0> is the
          # same vector under every permutation of the qubits): the qubits needed last start on the rank bits.  A/B switch
          'initial_placement': True,
          # the FIRST exchange behind reset() without the wire (round 6): only rank 0 holds anything before it, so every
          # rank computes rank 0's first stretch itself (same |0..0>, same gates as rank 0 sees them, the known-zero
          # masks make it cheap) and keeps the chunk the all-to-all would have brought it; the other chunk slots are the
          # zeros the other ranks would have sent.  No collective, no bytes on the links.  A/B switch
          'first_exchange_local': True,
          # remaps evict to the rank bits only qubits whose move can ride on a fused pass (not on the contiguous low bits of
          # a tile) while others are left: saves the re-labelling pass in front of such an exchange, at times for one more
          # exchange.  None = per circuit, whichever the dry-run model prices lower (`choose_eviction`)
          'evict_foldable': None,
          # EXCHANGE OVERLAP FOR AN UN-BATCHED SHARD WITHOUT EXTRA STRETCHES (round 6): the last pass in front of a remap and
          # the first pass behind it run in 2^B slices by B index bits right below the chunk bits (`_remap_sliced`,
          # dq_apply_fused_slice_*): slice j of every chunk leaves for its peer while slice j + 1 computes, and the next
          # stretch starts on slice j as soon as it has arrived.  The B qubits are the local ones needed last after the
          # evicted ones -- outside the tiles of both passes on (nearly) every rank; where one is not, that rank launches
          # fewer, bigger slices and the protocol stays the same.  Costs a third shard-sized buffer (the first pass behind
          # the exchange must not write where slices are still being sent from).  None = 3 where an exchange can overlap
          # with compute at all (RCCL on device shards, or the rehearsal of such a job), 0 elsewhere; an int forces it
          'slice_exchange': None,
          # an under-filled LAST pass of a stretch (at most this many kernel gates) is not run at all: its gates move behind
          # the exchange, where the next stretch's first passes take them in (legal when none of them targets a qubit that
          # leaves for the rank bits; `_defer_tail`).  A pass of 3-7 gates costs a whole read and write of the shard; dry run of
          # the n = 34 / 8-rank step: 33 -> 32 passes.  0 = off
          'defer_tail': 12}

#: the accumulator of the DQ_FG_GRAD reductions while a fused reverse sweep runs on a sharded (psi, lambda) pair
#: (adjoint._sweep_fused_sharded): every local stretch hands its rows to the passes
_SWEEP: dict = {'grads': None}

#: statistics of the last ``dist_apply_prims`` call (bench / tests)
LAST_RUN = {'remaps': 0, 'pairwise_exchanges': 0, 'local_flushes': 0, 'folded_permutes': 0, 'permute_passes': 0,
            'wire_bytes': 0, 'groups': 1, 'virtual_bits': 0, 'virtual_remaps': 0, 'zero_shard_stretches': 0,
            'known_zero_stretches': 0, 'local_first_exchanges': 0, 'zero_fills': 0, 'sliced_remaps': 0,
            'slice_launches_last': 0, 'slice_launches_first': 0, 'deferred_tails': 0, 'deferred_gates': 0}


# ---------------------------------------------------------------------------------------------------
# helpers on one shard
class _raw:
    """While one of these is open on a state its ``amps`` is the raw shard in whatever qubit order the last remap left
    (DistributedQubitState.__getattr__ restores the canonical order for everybody else).  Re-entrant."""

    def __init__(self, *states: DistributedQubitState):
        self.states = states

    def __enter__(self):
        for st in self.states:
            st.__dict__['_raw'] = st.__dict__.get('_raw', 0) + 1

    def __exit__(self, *exc):
        for st in self.states:
            st.__dict__['_raw'] -= 1
        return False


def _vbits(state: DistributedQubitState) -> int:
    """Virtual rank bits the state currently runs with (CONFIG['virtual_bits']; only inside `_dist_apply_prims`)."""
    return state.__dict__.get('_vbits', 0)


def _view(state: DistributedQubitState) -> torch.Tensor:
    """(batch, 2^L) view of the shard(s) -- with virtual rank bits: (2^v rows, 2^(L - v)) of the one shard."""
    return state.amps.view(-1, state.num_amps_per_node >> _vbits(state))


def _bview(state: DistributedQubitState) -> torch.Tensor:
    return state.buffer.view(-1, state.num_amps_per_node >> _vbits(state))


def _live(state: DistributedQubitState) -> bool:
    """More than one rank, and either a process group or a rehearsal of one of its ranks (CONFIG['elide_exchange'])."""
    return state.world_size > 1 and (dist.is_initialized() or bool(CONFIG['elide_exchange']))


def _all_reduce(t: torch.Tensor) -> None:
    """Sum over the ranks (left out in a rehearsal: this rank's share stands for the whole)."""
    if not CONFIG['elide_exchange']:
        dist.all_reduce(t, dist.ReduceOp.SUM)


def _materialize_zeros(state: DistributedQubitState) -> None:
    """Real zeros where the shard is only LOGICALLY zero (``_lazy_zero`` of `DistributedQubitState.reset(lazy=True)`:
    everything behind the first LAZY_HEAD amplitudes of every row; ``_zeros_owed = ('top', k)`` behind the first
    exchange without the wire: chunk slots 1 .. 2^k - 1).  Called by whoever is about to read the shard without the
    known-zero masks that make the garbage unreachable."""
    d = state.__dict__
    fresh = d.pop('_lazy_zero', False)
    owed = d.pop('_zeros_owed', None)
    if not fresh and owed is None:
        return
    rows = state._buffers['amps'].view(-1, state.num_amps_per_node)
    if fresh or owed[0] == 'fresh':
        rows[:, state.LAZY_HEAD:].zero_()
    else:
        rows[:, state.num_amps_per_node >> owed[1]:].zero_()
    LAST_RUN['zero_fills'] += 1


def _rank_controls_ok(state: DistributedQubitState, controls: Sequence[int]) -> bool:
    L = state.log_num_amps_per_node
    return all(get_bit(state.rank, c - L) for c in controls if c >= L)


def _localize(state: DistributedQubitState, p: Prim) -> Prim | None | str:
    """Translate a primitive on global bit positions into what THIS rank has to do locally.

    Returns a local ``Prim``, ``None`` (nothing to do on this rank) or ``'exchange'`` (a non-diagonal
    target is a global qubit: data has to move).  With virtual rank bits every row of the shard is a rank of the
    virtual world: a primitive whose predicates or phases depend on a virtual bit comes back with one matrix per row
    (the identity on the rows it does not act on)."""
    vb = _vbits(state)
    if vb == 0:
        # (the first stretch behind reset() with CONFIG['first_exchange_local']: every rank computes what RANK 0 holds)
        return _localize_at(state.log_num_amps_per_node, 0 if state.__dict__.get('_as_rank0') else state.rank, p)
    L = state.log_num_amps_per_node
    lr = L - vb
    if p.kind != 'diag' and any(t >= lr for t in p.targets):
        return 'exchange'
    on_virtual = any(lr <= c < L for c in p.controls) or (p.kind == 'diag' and any(lr <= t < L for t in p.targets))
    if not on_virtual:          # (every row sees the same thing: no per-row matrices, no comparison of device tensors)
        return _localize_at(lr, state.rank << vb, p)
    per_row = [_localize_at(lr, (state.rank << vb) | r, p) for r in range(1 << vb)]
    live = [q for q in per_row if q is not None]
    if not live:
        return None
    first = live[0]
    d = 1 << len(first.targets)
    dtype, device = state.amps.dtype, state.amps.device
    if first.kind == 'x':       # an X on some rows only: a dense (real) matrix per row
        flip = torch.tensor([[0, 1], [1, 0]], dtype=dtype, device=device)
        eye = torch.eye(2, dtype=dtype, device=device)
        mats = torch.stack([eye if q is None else flip for q in per_row])
        return Prim('gen', mats, first.targets, first.controls, 1)
    eye = torch.eye(d, dtype=dtype, device=device)
    mats = torch.stack([eye if q is None else q.matrix.reshape(d, d).to(dtype) for q in per_row])
    return Prim(first.kind, mats, first.targets, first.controls, 0)


def _localize_at(L: int, rank: int, p: Prim) -> Prim | None | str:
    """`_localize` for the rank ``rank`` of a world whose ranks hold 2^L amplitudes."""
    lc = tuple(c for c in p.controls if c < L)
    if p.kind != 'diag' and any(t >= L for t in p.targets):
        return 'exchange'
    if not all(get_bit(rank, c - L) for c in p.controls if c >= L):
        return None
    if p.kind != 'diag' or all(t < L for t in p.targets):
        return Prim(p.kind, p.matrix, p.targets, lc, p.mode)
    # diagonal gate with global target(s): the rank bits select a sub-block of the diagonal
    diag = p.matrix.diagonal(dim1=-2, dim2=-1)             # (.., D); a leading dim = one matrix per sample
    k = len(p.targets)
    local_t = [t for t in p.targets if t < L]
    sel = []
    for idx in range(1 << len(local_t)):
        full, li = 0, 0
        for i, t in enumerate(p.targets):
            if t >= L:
                bit = get_bit(rank, t - L)
            else:
                bit = (idx >> (len(local_t) - 1 - li)) & 1
                li += 1
            full |= bit << (k - 1 - i)
        sel.append(diag[..., full])
    if local_t:
        return Prim('diag', torch.stack(sel, dim=-1).diag_embed(), tuple(local_t), lc)
    phase = sel[0]
    if lc:  # phase on the controlled sub-cube = diag(1, phase) on one control bit, controlled by the rest
        one = torch.ones_like(phase)
        return Prim('diag', torch.stack([one, phase], dim=-1).diag_embed(), (lc[0],), lc[1:])
    return Prim('diag', torch.stack([phase, phase], dim=-1).diag_embed(), (0,), ())


def _row_groups(state: DistributedQubitState) -> list[slice]:
    """Groups of samples that move through a remap independently (each on its own stream when the shards are on a
    GPU): equal sizes, so that every group takes the same plan."""
    rows = _view(state).shape[0]
    g = max(1, min(int(CONFIG['overlap_groups']), rows))
    while rows % g:
        g -= 1
    if state.world_size == 1:
        g = 1
    step = rows // g
    return [slice(i * step, (i + 1) * step) for i in range(g)]


_STREAMS: dict = {}


def _group_streams(state: DistributedQubitState, ngroups: int) -> list:
    """One side stream per group (GPU shards with more than one group; None otherwise = the current stream)."""
    if ngroups == 1 or not state.amps.is_cuda:
        return [None] * ngroups
    key = (state.amps.device, ngroups)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=state.amps.device) for _ in range(ngroups)]
    return _STREAMS[key]


class _on:
    """``with _on(stream):`` -- run on a side stream that first waits for everything enqueued so far on the current
    one; ``None`` = stay on the current stream."""

    def __init__(self, stream):
        self.stream = stream
        self.ctx = None

    def __enter__(self):
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


#: per-exchange timing (bench.py --gpus N): with ``enabled`` every remap records HIP events on the stream of its sample
#: group -- before its local passes, when its exchange is issued, and when the wait for it has been passed -- so that
#: `remap_timings` can tell the last local stretch from the time on the wire
TIMING: dict = {'enabled': False, 'remaps': []}


def _mark(stream):
    """A timing event on the group's stream (None unless timing is on and the state lives on a GPU)."""
    if not TIMING['enabled'] or not torch.cuda.is_available():
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(stream if stream is not None else torch.cuda.current_stream())
    return ev


def _wait(work, stream) -> None:
    """Pass the wait for an exchange in flight (stream-ordered on RCCL) and note when the stream got past it."""
    work.wait()
    if TIMING['enabled']:
        for rec in TIMING['remaps']:
            if rec.get('exchange') is work and 'done' not in rec:
                rec['done'] = _mark(stream)


def remap_timings() -> list[dict]:
    """What `TIMING` recorded, in milliseconds (call after a device synchronisation): per remap and sample group the
    local passes in front of the exchange (``local_ms``) and issue -> wait passed (``wire_ms``: the exchange itself plus
    whatever the stream did in between -- with several groups that is the overlap)."""
    out = []
    for rec in TIMING['remaps']:
        row = {k: rec[k] for k in ('remap', 'rows', 'k', 'bytes')}
        for k_ in ('slices', 'slices_last', 'slices_first'):       # (a sliced exchange: protocol slices, launches of the two passes)
            if k_ in rec:
                row[k_] = rec[k_]
        if rec.get('start') is not None and rec.get('issued') is not None:
            row['local_ms'] = rec['start'].elapsed_time(rec['issued'])
        if rec.get('issued') is not None and rec.get('done') is not None:
            row['wire_ms'] = rec['issued'].elapsed_time(rec['done'])
        out.append(row)
    return out


def _settle(state: DistributedQubitState) -> None:
    """Join the group streams: everything in flight for this state (exchanges included) is ordered before whatever the
    current stream does next."""
    _arrivals_done(state, state.__dict__.pop('_arrivals', None))
    keep = state.__dict__.pop('_inflight_keep', None)
    for stream, works in state.__dict__.pop('_inflight', []):
        if stream is not None:
            with torch.cuda.stream(stream):
                for w in works:
                    _wait(w, stream)
            torch.cuda.current_stream(stream.device).wait_stream(stream)
        else:
            for w in works:
                _wait(w, None)
    del keep        # (the matrices the group streams were reading: only now may their memory be reused)


class _EventWait:
    """``wait()``: the current stream waits for an event of the exchange stream (a slice that has arrived)."""

    __slots__ = ('event',)

    def __init__(self, event) -> None:
        self.event = event

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


def _arrivals_done(state: DistributedQubitState, arr: dict | None) -> None:
    """Join a sliced exchange (`_remap_sliced`): every slice is ordered before what the current stream does next, and the
    buffer the slices were sent from becomes the state's spare third buffer again."""
    if arr is None:
        return
    for j, w in enumerate(arr['works']):
        if w is not None:
            w.wait()
            arr['works'][j] = None
    if TIMING['enabled'] and arr.get('timing') is not None and 'done' not in arr['timing']:
        arr['timing']['done'] = _mark(None)
    state.__dict__['_spare'] = arr.pop('src', None)


def _first_slicing(state: DistributedQubitState, arr: dict | None) -> dict | None:
    """``executor.run(slicing=...)`` for the first pass behind a sliced exchange: slice j is waited for right before the
    first launch that reads it."""
    if arr is None:
        return None

    def before(j: int) -> None:
        w = arr['works'][j]
        if w is not None:
            w.wait()
            arr['works'][j] = None

    return {'first': (arr['bits'], before)}


def _rows_of(pending: Sequence[Prim], rows: slice, total: int) -> list[Prim]:
    """The primitives as group ``rows`` of the batch sees them (per-sample matrices are sliced)."""
    if rows.start == 0 and rows.stop == total:
        return list(pending)
    out = []
    for p in pending:
        m = p.matrix
        if m is not None and m.ndim == 3 and m.shape[0] == total and total > 1:
            m = m[rows]
        out.append(p if m is p.matrix else replace(p, matrix=m))     # (an unsliced primitive stays the same object:
    return out                                                         #  the executor's steady cache keys on identity)


def _run_rows(a: torch.Tensor, b: torch.Tensor, pending: Sequence[Prim], rows: slice,
              out_perm: Sequence[int] | None = None, expect_z: dict | None = None, zero: bool = False,
              need_zeros=None, slicing: dict | None = None) -> bool:
    """Fused local passes on rows ``rows`` of the shard ``a`` with the receive buffer ``b`` as the second buffer of the
    permuted stores; afterwards local bit q sits at position out_perm[q].  Returns True if the result lives in ``b``.
    ``zero``: the rows are |0..0> (rank 0's shard right after ``reset()``): the first passes skip what is still known to be
    zero (executor.CONFIG['zero_state'])."""
    total = a.shape[0]
    x, y = a[rows], b[rows]
    if need_zeros is not None and (_SWEEP['grads'] is not None or not (CONFIG['fold_permute'] or out_perm is None)):
        need_zeros()                              # (routes below that take no masks)
        need_zeros = None
    if slicing is not None and (_SWEEP['grads'] is not None or not (CONFIG['fold_permute'] or out_perm is None)):
        executor._slicing_all(slicing, 'first')   # (routes below that take no slices: everything has to be there first)
        slicing = {k_: v_ for k_, v_ in slicing.items() if k_ != 'first'}
        post = slicing
        slicing = None
    else:
        post = None
    if _SWEEP['grads'] is not None:               # a stretch of a fused reverse sweep: reductions inside the passes
        out = executor.run(x, _rows_of(pending, rows, total), inplace=True, scratch=y, out_perm=out_perm, amps=a.numel(),
                           grads=_SWEEP['grads'][rows])
    elif CONFIG['fold_permute'] or out_perm is None:
        # (``need_zeros``: the shard holds garbage where it is logically zero; the executor calls it before anything reads
        # there -- i.e. unless the known-zero masks of ``zero`` apply to this schedule from its first pass to its last)
        out = executor.run(x, _rows_of(pending, rows, total), inplace=True, scratch=y, out_perm=out_perm, amps=a.numel(),
                           expect_z=expect_z, zero_state=zero, need_zeros=need_zeros, slicing=slicing)
    else:                                         # A/B: the re-labelling as a pass of its own
        out = executor.run(x, _rows_of(pending, rows, total), inplace=True, scratch=y)
        if out.data_ptr() not in (x.data_ptr(), y.data_ptr()):
            x.copy_(out)
            out = x
        out = executor.run(out, [], inplace=True, scratch=y if out.data_ptr() == x.data_ptr() else x, out_perm=out_perm)
    if post is not None:
        executor._slicing_all(post, 'last', out)
    if out.data_ptr() == y.data_ptr():
        return True
    if out.data_ptr() != x.data_ptr():        # (states smaller than a tile come back in a fresh tensor)
        x.copy_(out)
    return False


def _flush(state: DistributedQubitState, pending: list[Prim], expect_z: dict | None = None) -> None:
    arr = state.__dict__.pop('_arrivals', None)      # (a sliced exchange in flight: the first pass below takes it slice by slice)
    if arr is not None and not pending:
        _arrivals_done(state, arr)
        arr = None
    _settle(state)
    fresh = state.__dict__.pop('_fresh_zero', False)
    state.__dict__.pop('_zero_shard', None)      # (a shard of zeros runs its passes here like anybody: zeros in, zeros out)
    state.__dict__.pop('_as_rank0', None)        # (no exchange came: ranks != 0 ran rank 0's gates on zeros -- zeros out)
    kz = state.__dict__.pop('_known_zero_local', 0)
    zero = (fresh and state.rank == 0) or kz
    lazy = bool(state.__dict__.get('_lazy_zero') or state.__dict__.get('_zeros_owed'))
    if lazy and not (pending and zero):
        _materialize_zeros(state)                # (nothing runs, or it runs unmasked: the logical zeros become real ones)
        lazy = False
    if not pending:
        return
    LAST_RUN['local_flushes'] += 1
    LAST_RUN['known_zero_stretches'] += bool(kz)
    a, b = _view(state), _bview(state)
    slicing = _first_slicing(state, arr)
    if _run_rows(a, b, pending, slice(0, a.shape[0]), expect_z=expect_z, zero=zero,
                 need_zeros=(lambda: _materialize_zeros(state)) if lazy else None, slicing=slicing):
        state.amps, state.buffer = state.buffer, state.amps
    if arr is not None:
        LAST_RUN['slice_launches_first'] += slicing.get('done', {}).get('first', 0)
        if TIMING['enabled'] and arr.get('timing') is not None:
            arr['timing']['slices_first'] = slicing.get('done', {}).get('first', 0)
        _arrivals_done(state, arr)
    state.__dict__.pop('_lazy_zero', None)       # (masked passes that ran through have written everything)
    state.__dict__.pop('_zeros_owed', None)
    pending.clear()


def _expect_z_local(state: DistributedQubitState, zmasks: Sequence[int]) -> tuple[dict, list[float]]:
    """The Z strings ``zmasks`` (logical qubits) as the last local stretch of a circuit sees them: the factors on local
    bits (masks in physical positions, for executor.run(expect_z=...)) and this rank's sign from the factors on rank bits."""
    L, ph = state.log_num_amps_per_node, _phys(state)
    local, signs = [], []
    for zm in zmasks:
        pz = sum(1 << ph[q] for q in range(state.nqubit) if (int(zm) >> q) & 1)
        local.append(pz & ((1 << L) - 1))
        signs.append(-1.0 if bin((pz >> L) & state.rank).count('1') & 1 else 1.0)
    return {'masks': local}, signs


def _expect_z_finish(state: DistributedQubitState, zmasks: Sequence[int], holder: dict | None, signs: list[float] | None) -> None:
    """Every rank adds what its last pass reduced (times its sign) -- or says that it reduced nothing, in which case all
    ranks drop the values alike (a rank whose last stretch was empty, or did not run on the wave-tile kernel) -- and the
    sums are cached on the state for ``expectation()``: <Z..Z> of a sharded circuit without another read of the shards."""
    rows = _view(state).shape[0]
    k = len(zmasks)
    buf = torch.zeros(rows, k + 1, dtype=torch.float64, device=state.amps.device)
    if holder is not None and holder.get('values') is not None:
        buf[:, :k] = holder['values'] * torch.tensor(signs, dtype=torch.float64, device=buf.device)
        buf[:, k] = 1.0
    if state.world_size > 1 and dist.is_initialized():
        _all_reduce(buf)
    # (whether every rank contributed is a number on the device: `expectation()` looks at it -- no host sync in the forward)
    state.__dict__['_expz'] = {'masks': [int(z) for z in zmasks], 'values': buf[:, :k], 'ranks': buf[0, k]}


def cached_expect_z(state: DistributedQubitState) -> dict | None:
    """The Z-string values the last circuit's final pass left on the state, if every rank took part."""
    ez = state.__dict__.get('_expz')
    if ez is None:
        return None
    if 'ok' not in ez:
        ez['ok'] = float(ez['ranks']) == state.world_size
    return ez if ez['ok'] else None


# ---------------------------------------------------------------------------------------------------
# exchange steps
def _swap_local_global(state: DistributedQubitState, lbit: int, gbit: int) -> None:
    """SWAP of local qubit ``lbit`` with global qubit ``gbit``: each rank trades the half of its shard
    whose local bit differs from its rank bit (Alg. 9; reference: distributed.py:148-158)."""
    L = state.log_num_amps_per_node
    rb = gbit - L
    b = get_bit(state.rank, rb)
    pair = state.rank ^ (1 << rb)
    mask = 1 << lbit
    value = (1 - b) << lbit
    send = backend.pack(_view(state), mask, value)
    recv = state.buffer.view(-1)[: send.numel()].view(send.shape)
    comm_exchange_arrays(send, recv, pair)
    backend.unpack_axpby(_view(state), recv, None, None, mask, value)


def _swap_global_global(state: DistributedQubitState, g1: int, g2: int) -> None:
    L = state.log_num_amps_per_node
    r1, r2 = g1 - L, g2 - L
    if get_bit(state.rank, r1) != get_bit(state.rank, r2):
        pair = state.rank ^ (1 << r1) ^ (1 << r2)
        comm_exchange_arrays(state.amps, state.buffer, pair)
        state.amps, state.buffer = state.buffer, state.amps
    else:
        comm_exchange_arrays(state.amps, state.buffer, None)


def _one_target_global(state: DistributedQubitState, p: Prim, derivative: bool = False) -> None:
    """Single-qubit (possibly controlled) gate whose target is a global qubit (Alg. 6-8; reference:
    distributed.py:57-127): exchange the (controlled part of the) shard with the partner rank and
    combine  amps <- M[b,b] * amps + M[b,1-b] * received."""
    L = state.log_num_amps_per_node
    t = p.targets[0]
    rb = t - L
    lc = [c for c in p.controls if c < L]
    if not _rank_controls_ok(state, p.controls):
        if derivative:
            state.amps.zero_()
        comm_exchange_arrays(state.amps, state.buffer, None)
        return
    b = get_bit(state.rank, rb)
    pair = state.rank ^ (1 << rb)
    m = p.matrix
    if p.kind == 'x':
        m = m.new_tensor([[0, 1], [1, 0]])
    coef = torch.stack([m[..., b, b], m[..., b, 1 - b]], dim=-1).to(state.amps.dtype)   # (2,) or (B, 2)
    mask = 0
    for c in lc:
        mask |= 1 << c
    if mask == 0:
        comm_exchange_arrays(state.amps, state.buffer, pair)
        backend.unpack_axpby(_view(state), _view(state), _bview(state), coef, 0, 0)
        return
    send = backend.pack(_view(state), mask, mask)
    recv = state.buffer.view(-1)[: send.numel()].view(send.shape)
    comm_exchange_arrays(send, recv, pair)
    if derivative:
        state.amps.zero_()
    backend.unpack_axpby(_view(state), send, recv, coef, mask, mask)


def _many_target_global(state: DistributedQubitState, p: Prim) -> None:
    """Multi-qubit gate with global targets: swap each global target with a free local qubit, apply
    locally, swap back (Alg. 10; reference: distributed.py:162-202).  Controls stay where they are."""
    L = state.log_num_amps_per_node
    glob_t = [t for t in p.targets if t >= L]
    loc_t = set(t for t in p.targets if t < L)
    loc_c = set(c for c in p.controls if c < L)
    # host qubits: local bits that are not targets, bits that are not controls either first (a control
    # bit may host a target too: the control then sits on the global position for the duration)
    hosts = [q for q in range(L) if q not in loc_t and q not in loc_c] + [q for q in range(L) if q in loc_c]
    assert len(hosts) >= len(glob_t), 'not enough local qubits to host the gate'
    subst = dict(zip(glob_t, hosts))
    for g, l in subst.items():
        _swap_local_global(state, l, g)
    # after the swaps the former local qubits l sit on the global positions: a control that used one
    # of them moves with it
    rev = {l: g for g, l in subst.items()}
    new_t = tuple(subst.get(t, t) for t in p.targets)
    new_c = tuple(rev.get(c, c) for c in p.controls)
    local = _localize(state, Prim(p.kind, p.matrix, new_t, new_c))
    if isinstance(local, Prim):
        _flush(state, [local])
    for g, l in reversed(list(subst.items())):
        _swap_local_global(state, l, g)


def _exchange_prim(state: DistributedQubitState, p: Prim) -> None:
    L = state.log_num_amps_per_node
    LAST_RUN['pairwise_exchanges'] += 1
    if len(p.targets) == 1:
        _one_target_global(state, p)
    else:
        # free local slots must not collide with local controls: handled inside
        _many_target_global(state, p)
    del L


# ---------------------------------------------------------------------------------------------------
# qubit remap: logical -> physical permutation + k-qubit all-to-all
def _phys(state: DistributedQubitState) -> list[int]:
    """phys[logical bit] = physical bit (physical bit p >= L is rank bit p - L)."""
    ph = state.__dict__.get('_phys')
    if ph is None:
        ph = list(range(state.nqubit))
        state.__dict__['_phys'] = ph
    return ph


def _is_canonical(state: DistributedQubitState) -> bool:
    ph = state.__dict__.get('_phys')
    return ph is None or all(p == q for q, p in enumerate(ph))


def _translate(p: Prim, ph: list[int]) -> Prim:
    return replace(p, targets=tuple(ph[t] for t in p.targets), controls=tuple(ph[c] for c in p.controls),
                   order=tuple(ph[o] for o in p.order))


def _permute_local(state: DistributedQubitState, src_of_dst: list[int]) -> None:
    """Re-label the local qubits (one read + one write): destination bit d <- source bit src_of_dst[d]."""
    _settle(state)
    if src_of_dst == list(range(len(src_of_dst))):
        return
    backend.permute_bits(_view(state), src_of_dst, out=_bview(state))
    state.amps, state.buffer = state.buffer, state.amps
    ph = _phys(state)
    dst_of_src = {sp: d for d, sp in enumerate(src_of_dst)}
    for q, p in enumerate(ph):
        if p < state.log_num_amps_per_node:
            ph[q] = dst_of_src[p]


def _exchange_qubits(state: DistributedQubitState, pairs: list[tuple[int, int]]) -> None:
    """Swap k global qubits with k local ones in one all-to-all.  ``pairs`` = [(leaving logical qubit,
    entering logical qubit)]: the entering qubit takes over the rank bit of the leaving one."""
    _remap(state, pairs, [])
    _settle(state)


def _remap(state: DistributedQubitState, pairs: list[tuple[int, int]], pending: list[Prim],
           slice_qubits: Sequence[int] = ()) -> None:
    """The local gates ``pending`` (physical positions), then the exchange ``pairs``.  Per group of samples
    (CONFIG['overlap_groups']; each on its own stream): fused passes whose LAST one also moves the entering qubits to
    the top k local bits (executor ``out_perm``), then one all-to-all per sample among the 2^k ranks of the group --
    chunk c of the shard goes to the peer whose rank bits spell c -- issued asynchronously, so that the next group's
    passes run while this group's amplitudes are on the links.  Nothing waits here: ``_settle`` (or the next remap of
    the same group) does."""
    vb = _vbits(state)
    L, W = state.log_num_amps_per_node - vb, state.world_size       # (with virtual rank bits: L = bits of a ROW)
    ph = _phys(state)
    pairs = sorted(pairs, key=lambda pr: ph[pr[0]])          # ascending rank bit -> ascending peer rank
    k = len(pairs)
    rbits = [ph[lq] - L for lq, _ in pairs]                 # bits of the (virtual) world's rank: the low vb are rows
    assert all(0 <= r < state.log_num_nodes + vb for r in rbits) and all(ph[eq] < L for _, eq in pairs)
    # a sliced exchange in flight (`_remap_sliced`): the first pass of THIS stretch takes it slice by slice
    arr = state.__dict__.pop('_arrivals', None)
    # 1. the entering qubits go to the top k local bits (chunk index = their joint value): destination bit d takes
    #    source bit src_of_dst[d]
    ent_bits = [ph[eq] for _, eq in pairs]
    sq = [q for q in slice_qubits if ph[q] < L and ph[q] not in ent_bits]
    sliced = (len(sq) > 0 and vb == 0 and _view(state).shape[0] == 1 and _live(state) and L - k - len(sq) >= 12
              and _SWEEP['grads'] is None and CONFIG['fold_permute'] and not state.__dict__.get('_as_rank0')
              and not state.__dict__.get('_behind_reset'))
    if not sliced:
        sq = []
    sbits = [ph[q] for q in sq]
    #    (with slices: the slice qubits right below them -- protocol bit i on local bit L - k - B + i)
    src_of_dst = [b for b in range(L) if b not in ent_bits and b not in sbits] + sbits + ent_bits
    out_perm = [0] * L
    for d, sp in enumerate(src_of_dst):
        out_perm[sp] = d
    identity = out_perm == list(range(L))
    # 2. chunk c goes to the peer whose rank bits `rbits` spell c; what comes back from that peer lands in
    #    the same chunk slot.  Peers outside the 2^k group get empty messages.
    chunk = (1 << (L - k))
    if vb:
        # (the planner re-fills ONE class of far positions per remap, `_plan_remap`)
        assert all(r < vb for r in rbits) or all(r >= vb for r in rbits), 'a remap trades real OR virtual rank bits'
        if rbits[0] < vb:               # virtual bits trade places: a re-labelling of the shard, no exchange
            _remap_virtual(state, pairs, pending)
            return
        pending[:] = [q for q in (_localize(state, p) for p in pending) if q is not None]      # row by row
    peers = []
    for c in range(1 << k):
        peer = state.rank
        for i, r in enumerate(rbits):
            peer = (peer & ~(1 << (r - vb))) | (((c >> i) & 1) << (r - vb))
        peers.append(peer)
    a, b = _view(state), _bview(state)
    groups = _row_groups(state)
    streams = _group_streams(state, len(groups))
    if sliced or arr is not None:
        if _remap_sliced(state, pairs, rbits, pending, out_perm, identity, k, chunk, peers, len(sq), arr):
            _remap_bookkeeping(ph, pairs, rbits, out_perm, L)
            return
        a, b = _view(state), _bview(state)        # (the stretch may have run and changed the buffers' roles)
    # the first stretch behind reset(): rank 0 holds |0..0> -- its first passes skip what is still known to be zero --
    # and every other rank holds nothing but zeros, which stay zeros under any gates and in any layout: no pass at all
    # (with virtual rank bits the rows of rank 0's shard are |0..0> and zeros: the masks hold for both; the other ranks
    # have skipped every stretch since reset() -- `_remap_virtual` -- and receive their first amplitudes now)
    fresh = state.__dict__.pop('_fresh_zero', False)
    as0 = state.__dict__.pop('_as_rank0', False)
    zeros = state.rank != 0 and (fresh or state.__dict__.pop('_zero_shard', False))
    state.__dict__.pop('_zero_shard', None)
    kz = state.__dict__.pop('_known_zero_local', 0)
    first_exchange = state.__dict__.pop('_behind_reset', False)
    if as0 and fresh and first_exchange and vb == 0 and _live(state):
        _first_exchange_local(state, pairs, rbits, pending, out_perm, identity, k, chunk)
        _remap_bookkeeping(ph, pairs, rbits, out_perm, L)
        state.__dict__['_known_zero_local'] = ((1 << k) - 1) << (L - k)
        return
    if zeros:
        LAST_RUN['zero_shard_stretches'] += 1
    elif pending:
        LAST_RUN['known_zero_stretches'] += bool(kz)
    # logical zeros (a lazy reset, the first exchange without the wire): the masked passes of ONE group of rows may run on
    # them -- the executor clears them itself if the masks do not apply; every other case gets real zeros first
    lazy = bool(state.__dict__.get('_lazy_zero') or state.__dict__.get('_zeros_owed'))
    if lazy and not (len(groups) == 1 and not zeros and (pending or not identity) and (fresh and state.rank == 0 or kz)
                     and not (state.__dict__.get('_lazy_zero') and not fresh)):
        _materialize_zeros(state)
        lazy = False
    need_zeros = (lambda: _materialize_zeros(state)) if lazy else None
    inflight_prev = {id(st): (st, works) for st, works in state.__dict__.pop('_inflight', [])}
    inflight, landed_in_a = [], []
    if pending:
        LAST_RUN['local_flushes'] += 1
    for rows, stream in zip(groups, streams):
        with _on(stream):
            for w in inflight_prev.pop(id(stream), (None, []))[1]:    # this group's previous exchange
                _wait(w, stream)
            t_start = _mark(stream)
            if zeros:
                in_b = False
            else:
                in_b = (_run_rows(a, b, pending, rows, None if identity else out_perm, zero=fresh or kz, need_zeros=need_zeros)
                        if (pending or not identity) else False)
                state.__dict__.pop('_lazy_zero', None)      # (ran through masked -- or were cleared: all is written)
                state.__dict__.pop('_zeros_owed', None)
            src, dst = (b, a) if in_b else (a, b)
            works = []
            if _live(state):
                send, recv = torch.view_as_real(src[rows]), torch.view_as_real(dst[rows])     # (rows, 2^L, 2)
                nbytes = send.shape[0] * ((1 << k) - 1) * chunk * send.element_size() * 2
                t_issue = _mark(stream)
                ex = None
                if not CONFIG['elide_exchange']:
                    # ONE coalesced exchange for all samples of the group (communication.exchange_chunks): every chunk is
                    # contiguous where it lies, (rows x (2^k - 1)) send / receive pairs in one group call
                    ex = exchange_chunks(recv.reshape(send.shape[0], -1), send.reshape(send.shape[0], -1), peers, chunk * 2,
                                         what=(f'shard exchange of remap {LAST_RUN["remaps"] + 1} (logical qubits leaving / entering '
                                               f'{pairs}, rank bits {rbits}, samples {rows.start}:{rows.stop}, peers '
                                               f'{sorted(set(peers) - {state.rank})}, {nbytes} bytes each way)'),
                                         async_op=stream is not None)
                if ex is not None:
                    works.append(ex)
                LAST_RUN['wire_bytes'] += nbytes
                if TIMING['enabled']:
                    TIMING['remaps'].append({'remap': LAST_RUN['remaps'] + 1, 'rows': (rows.start, rows.stop), 'k': k,
                                             'bytes': nbytes, 'start': t_start, 'issued': t_issue, 'exchange': ex,
                                             'stream': stream})
            else:
                dst[rows].copy_(src[rows])
            landed_in_a.append(in_b)
            inflight.append((stream, works))
    for st, works in inflight_prev.values():                          # (streams of another grouping: join them)
        inflight.append((st, works))
    state.__dict__['_inflight'] = inflight
    # The passes queued on the group streams read the matrices of `pending` -- some of them temporaries that
    # `_localize` made on the main stream (rank-selected phases of diagonal gates on global qubits).  Dropping the last
    # reference here would hand their memory back to the caching allocator while that work is still queued, and the
    # next main-stream allocation could overwrite it: they stay referenced until `_settle` has joined the streams.
    if any(st is not None for st, _ in inflight):
        state.__dict__.setdefault('_inflight_keep', []).append(list(pending))
    if not all(landed_in_a):
        if any(landed_in_a):         # groups disagree on the buffer they ended in (never with equal group sizes)
            _settle(state)
            for rows, in_a in zip(groups, landed_in_a):
                if in_a:
                    b[rows].copy_(a[rows])
        state.amps, state.buffer = state.buffer, state.amps
    pending.clear()
    if not identity:
        LAST_RUN['folded_permutes' if executor.LAST_RUN.get('permute_folded') else 'permute_passes'] += 1
    LAST_RUN['groups'] = len(groups)
    _remap_bookkeeping(ph, pairs, rbits, out_perm, L)
    if first_exchange and _live(state):
        # The first exchange behind reset(): only rank 0 had anything to send, so on every rank (and in every row) what
        # arrived lies in chunk 0 and the other chunks hold the zeros the other ranks sent -- the qubits that came from
        # the rank bits, now on the top k local bits, are still |0>, and the next stretch starts with their mask
        # (executor.run(zero_state=mask): its first passes move 2^-k of the shard)
        state.__dict__['_known_zero_local'] = ((1 << k) - 1) << (L - k)


def _remap_sliced(state: DistributedQubitState, pairs, rbits, pending: list[Prim], out_perm, identity: bool, k: int,
                  chunk: int, peers: list[int], nb: int, arr: dict | None) -> bool:
    """A remap of an un-batched shard with its exchange in 2^nb slices (CONFIG['slice_exchange']; ``nb`` = 0: an ordinary
    exchange, but the first pass of this stretch takes the PREVIOUS sliced exchange ``arr`` slice by slice).

    The last pass of ``pending`` -- which also writes the re-labelling ``out_perm``: entering qubits on the top k local
    bits, the nb slice qubits right below them -- is launched slice by slice (`executor.run(slicing=...)`); behind every
    protocol slice j (the value of the nb bits) its part of every chunk -- contiguous: chunk c, slice j -- leaves for peer c
    on the exchange stream while the next launch computes.  Nothing waits here: the next stretch's first pass waits for
    slice j right before it reads it (`_first_slicing`), `_settle` for everything.  The pass behind the exchange must not
    write where slices are still being sent from, so the state's buffers rotate through a THIRD one: received -> ``amps``,
    spare -> ``buffer``, and the buffer the slices leave from becomes the spare when the last slice has gone.
    Returns False when there is nothing sliced to do (the caller's ordinary route runs)."""
    if nb == 0 and (arr is None or not pending):
        _arrivals_done(state, arr)
        return False
    for key in ('_fresh_zero', '_zero_shard', '_behind_reset'):
        state.__dict__.pop(key, None)
    kz = state.__dict__.pop('_known_zero_local', 0)
    lazy = bool(state.__dict__.get('_lazy_zero') or state.__dict__.get('_zeros_owed'))
    if lazy and not (kz and (pending or not identity)):
        _materialize_zeros(state)
        lazy = False
    if nb == 0:
        # an ordinary exchange behind a sliced one: run the stretch here (its first pass in slices), then fall back
        a, b = _view(state), _bview(state)
        slicing = _first_slicing(state, arr)
        LAST_RUN['local_flushes'] += 1
        LAST_RUN['known_zero_stretches'] += bool(kz)
        if _run_rows(a, b, pending, slice(0, 1), zero=kz, need_zeros=(lambda: _materialize_zeros(state)) if lazy else None,
                     slicing=slicing):
            state.amps, state.buffer = state.buffer, state.amps
        state.__dict__.pop('_lazy_zero', None)
        state.__dict__.pop('_zeros_owed', None)
        LAST_RUN['slice_launches_first'] += slicing.get('done', {}).get('first', 0)
        if TIMING['enabled'] and arr.get('timing') is not None:
            arr['timing']['slices_first'] = slicing.get('done', {}).get('first', 0)
        _arrivals_done(state, arr)
        pending.clear()
        return False
    _settle(state)
    L = state.log_num_amps_per_node
    nsl = 1 << nb
    sub = chunk >> nb
    a, b = _view(state), _bview(state)
    spare = state.__dict__.pop('_spare', None)
    if spare is None or spare.shape != state.amps.shape or spare.dtype != state.amps.dtype or spare.device != state.amps.device:
        spare = torch.empty_like(state.amps)
    on_gpu = a.is_cuda
    xs = _group_streams(state, 2)[1] if on_gpu else None          # the exchange stream
    works: list = [None] * nsl
    bufs: dict = {}
    rec = None
    if TIMING['enabled']:
        rec = {'remap': LAST_RUN['remaps'] + 1, 'rows': (0, 1), 'k': k, 'bytes': ((1 << k) - 1) * chunk * a.element_size(),
               'start': _mark(None), 'issued': None, 'exchange': None, 'stream': None, 'slices': nsl}
        TIMING['remaps'].append(rec)

    def pieces(t: torch.Tensor, j: int) -> list[torch.Tensor]:
        v = torch.view_as_real(t[0]).reshape(1 << k, nsl, sub * 2)
        return [v[c, j] for c in range(1 << k)]

    ready: set = set()
    nxt = [0]

    def after(j: int, where: torch.Tensor) -> None:
        # (the executor calls this behind the last launch that writes protocol slice j, with the buffer it wrote to.  A rank
        # whose last pass could be cut by fewer bits finishes the slices in another order than its peers: point-to-point
        # operations between two ranks match BY ORDER, so every rank issues the slices in protocol order, 0, 1, 2, ..)
        bufs['src'] = where
        ready.add(j)
        while nxt[0] in ready:
            issue(nxt[0], where)
            nxt[0] += 1

    def issue(j: int, where: torch.Tensor) -> None:
        src, dst = where, bufs['dst']
        nbytes = ((1 << k) - 1) * sub * a.element_size()
        LAST_RUN['wire_bytes'] += nbytes
        if rec is not None and rec['issued'] is None:
            rec['issued'] = _mark(None)
        if CONFIG['elide_exchange']:
            if CONFIG['elide_exchange'] == 'loopback' and xs is not None:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(xs):
                    xs.wait_event(ev)
                    for c, (pr, ps) in enumerate(zip(pieces(dst, j), pieces(src, j))):
                        if peers[c] != state.rank:
                            pr.copy_(ps)
                    done = torch.cuda.Event()
                    done.record(xs)
                works[j] = _EventWait(done)
            return
        what = (f'slice {j} of {nsl} of the shard exchange of remap {LAST_RUN["remaps"] + 1} (logical qubits leaving / entering '
                f'{pairs}, rank bits {rbits}, peers {sorted(set(peers) - {state.rank})}, {nbytes} bytes each way)')
        if xs is not None:
            # on the exchange stream, behind the launch that finished the slice; whoever needs the slice later waits for
            # `done` -- recorded on that stream behind the exchange's own completion (RCCL: a stream-ordered wait, no host
            # block; gloo on device memory: the staged copies back run on this stream) -- never for the handle itself
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(xs):
                xs.wait_event(ev)
                ex = exchange_pieces(pieces(dst, j), pieces(src, j), peers, what, async_op=True)
                if ex is not None:
                    ex.wait()
                done = torch.cuda.Event()
                done.record(xs)
            works[j] = _EventWait(done)
        else:
            works[j] = exchange_pieces(pieces(dst, j), pieces(src, j), peers, what, async_op=False)

    slicing = {'last': ([L - k - nb + i for i in range(nb)], after)}
    if arr is not None:
        slicing.update(_first_slicing(state, arr))
    bufs['src'], bufs['dst'] = None, spare.view(a.shape)
    LAST_RUN['known_zero_stretches'] += bool(kz)
    if pending:
        LAST_RUN['local_flushes'] += 1
    in_b = _run_rows(a, b, pending, slice(0, 1), None if identity else out_perm, zero=kz,
                     need_zeros=(lambda: _materialize_zeros(state)) if lazy else None, slicing=slicing)
    assert bufs['src'] is not None and (bufs['src'].data_ptr() == b.data_ptr()) == in_b, 'slices left from the wrong buffer'
    state.__dict__.pop('_lazy_zero', None)
    state.__dict__.pop('_zeros_owed', None)
    done = slicing.get('done', {})
    LAST_RUN['sliced_remaps'] += 1
    LAST_RUN['slice_launches_last'] += done.get('last', 0)
    LAST_RUN['slice_launches_first'] += done.get('first', 0)
    if arr is not None:
        if TIMING['enabled'] and arr.get('timing') is not None:
            arr['timing']['slices_first'] = done.get('first', 0)
        _arrivals_done(state, arr)
    if rec is not None:
        rec['slices_last'] = done.get('last', 0)
    # received -> amps, the old spare's place is taken by the buffer that was NOT the source, the source stays out of
    # reach until its slices have gone
    src_t, other_t = (state.buffer, state.amps) if in_b else (state.amps, state.buffer)
    state.amps, state.buffer = spare, other_t
    state.__dict__['_arrivals'] = {'bits': [L - k - nb + i for i in range(nb)], 'works': works, 'src': src_t, 'timing': rec}
    pending.clear()
    if not identity:
        LAST_RUN['folded_permutes' if executor.LAST_RUN.get('permute_folded') else 'permute_passes'] += 1
    LAST_RUN['groups'] = 1
    return True


def _first_exchange_local(state: DistributedQubitState, pairs, rbits, pending: list[Prim], out_perm, identity: bool,
                          k: int, chunk: int) -> None:
    """The first exchange behind reset() without the wire (CONFIG['first_exchange_local']).  Before it only rank 0 holds
    anything: the all-to-all would bring rank r chunk c(r) of RANK 0's shard (c(r) = r's bits on the exchanged rank bits),
    landing in chunk slot 0 (the sender's code), and zeros from everybody else.  ``pending`` was localized as rank 0 sees
    the gates (`_localize`, ``_as_rank0``), so every rank of rank 0's group computes that shard itself -- |0..0> in, the
    known-zero masks on: a fraction of a pass -- keeps its chunk and zeroes the other slots.  Ranks of a group without
    rank 0 (k < log2 W) would receive zeros only: they keep their zeros and run nothing."""
    _settle(state)
    leader = state.rank
    code = 0
    for i, r in enumerate(rbits):
        code |= ((state.rank >> r) & 1) << i
        leader &= ~(1 << r)
    LAST_RUN['local_first_exchanges'] += 1
    if leader != 0:
        LAST_RUN['zero_shard_stretches'] += 1
        _materialize_zeros(state)         # (a shard of zeros from here on: real ones)
        pending.clear()
        return
    a, b = _view(state), _bview(state)
    rows = slice(0, a.shape[0])
    if state.rank != 0:
        a[:, 0] = 1                       # rank 0's input: |0..0> (the rest of the shard is zero -- really, or logically)
    if pending:
        LAST_RUN['local_flushes'] += 1
    lazy = bool(state.__dict__.get('_lazy_zero'))
    if pending or not identity:
        in_b = _run_rows(a, b, pending, rows, None if identity else out_perm, zero=True,
                         need_zeros=(lambda: _materialize_zeros(state)) if lazy else None)
    else:
        _materialize_zeros(state)
        in_b = False
    was_lazy = lazy and state.__dict__.pop('_lazy_zero', False)     # (still set: the masked passes ran through, all is written)
    src, dst = (b, a) if in_b else (a, b)
    dst[:, :chunk].copy_(src[:, code * chunk:(code + 1) * chunk])
    if not in_b:                          # the new shard lies in the receive buffer
        state.amps, state.buffer = state.buffer, state.amps
    if was_lazy:
        # the other chunk slots -- the zeros the other ranks would have sent -- stay un-cleared: the next stretch starts
        # with the known-zero mask of the k qubits that came from the rank bits and reads nothing there; whoever cannot
        # vouch for that clears them first (`_materialize_zeros`)
        state.__dict__['_zeros_owed'] = ('top', k)
    else:
        dst[:, chunk:].zero_()
    pending.clear()
    if not identity:
        LAST_RUN['folded_permutes' if executor.LAST_RUN.get('permute_folded') else 'permute_passes'] += 1
    LAST_RUN['groups'] = 1


def _remap_bookkeeping(ph: list[int], pairs, rbits, out_perm, L: int) -> None:
    """Local qubits moved with the permutation; entering qubit i now is rank bit rbits[i]; leaving qubit i is local bit
    L - k + i."""
    k = len(pairs)
    for q, p_ in enumerate(ph):
        if p_ < L:
            ph[q] = out_perm[p_]
    for i, (lq, eq) in enumerate(pairs):
        ph[eq] = L + rbits[i]
        ph[lq] = L - k + i
    LAST_RUN['remaps'] += 1


def _remap_virtual(state: DistributedQubitState, pairs, pending: list[Prim]) -> None:
    """A remap that trades VIRTUAL rank bits only (CONFIG['virtual_bits']): nothing leaves the GPU -- the qubits on the
    virtual bits and the entering row-local qubits swap index positions of the ONE shard, and that re-labelling rides on
    the last pass of the stretch in front of it, which therefore runs on the whole shard (gates controlled by a virtual
    bit are ordinary local controls there) instead of row by row."""
    vb = state.__dict__.pop('_vbits')
    try:
        _settle(state)
        _materialize_zeros(state)
        L = state.log_num_amps_per_node
        ph = _phys(state)
        out_perm = list(range(L))
        for lq, eq in pairs:
            assert L - vb <= ph[lq] < L and ph[eq] < L - vb
            out_perm[ph[lq]], out_perm[ph[eq]] = ph[eq], ph[lq]
        # behind reset(): rank 0's shard is |0..0> (the first stretch runs with the known-zero masks), everybody else's is
        # all zeros and stays so -- under any gates, in any order of the index bits -- until the first REAL exchange
        fresh = state.__dict__.pop('_fresh_zero', False)
        kz = state.__dict__.pop('_known_zero_local', 0)      # (positions of a row: the same bits of the whole shard)
        zeros = state.rank != 0 and (fresh or state.__dict__.get('_zero_shard', False))
        if pending:
            LAST_RUN['local_flushes'] += 1
        if zeros:
            state.__dict__['_zero_shard'] = True
            LAST_RUN['zero_shard_stretches'] += 1
        else:
            LAST_RUN['known_zero_stretches'] += bool(kz)
            local = [q for q in (_localize(state, p) for p in pending) if q is not None]
            a, b = _view(state), _bview(state)
            if _run_rows(a, b, local, slice(0, a.shape[0]), out_perm, zero=fresh or kz):
                state.amps, state.buffer = state.buffer, state.amps
        pending.clear()
        LAST_RUN['folded_permutes' if executor.LAST_RUN.get('permute_folded') else 'permute_passes'] += 1
        for lq, eq in pairs:
            ph[lq], ph[eq] = ph[eq], ph[lq]
        LAST_RUN['remaps'] += 1
        LAST_RUN['virtual_remaps'] += 1
    finally:
        state.__dict__['_vbits'] = vb


#: the eviction policy in force (set per call by `_dist_apply_prims` from CONFIG['evict_foldable']; None there = whichever
#: the dry-run model prefers for the circuit at hand, `choose_eviction`)
_EVICT = [True]


#: index bits below this are the contiguous low bits of a complex64 tile (fusion.default_geometry: min_low = 4; 3 for
#: complex128 -- the stricter bound serves both): `fusion._place_writes` folds a final permutation only if it fixes them
_UNFOLDABLE_BELOW = 4


def _next_use(prims: Sequence[Prim], start: int, n: int) -> list[int]:
    """Index of the next gate (>= start) acting NON-diagonally on each logical qubit (inf if none soon)."""
    inf = 1 << 60
    nxt = [inf] * n
    left = n
    stop = min(len(prims), start + CONFIG['horizon'])
    for j in range(start, stop):
        p = prims[j]
        if p.kind == 'diag':
            continue
        for t in p.targets:
            if nxt[t] == inf:
                nxt[t] = j
                left -= 1
        if left == 0:
            break
    return nxt


def _plan_remap(ph: list[int], prims: Sequence[Prim], i: int, n: int, L: int, v: int = 0) -> list[tuple[int, int]]:
    """Which qubits trade places so that gate ``i`` becomes local: evict to the rank bits the qubits whose
    next non-diagonal use is farthest away (Belady).  Pure function of the gate list: every rank computes
    the same plan.

    ``v`` virtual rank bits (positions L .. L + v - 1; L = bits of a row): the two classes of far positions are
    re-filled SEPARATELY -- a remap either trades real rank bits only (its wire time hides behind the other rows'
    passes) or virtual bits only (chunks move between rows of the shard: no wire at all) -- virtual bits first when
    gate ``i`` waits for one of them; the loop comes back for the other class if the gate still is not local."""
    nxt = _next_use(prims, i, n)
    needed = {t for t in prims[i].targets} if prims[i].kind != 'diag' else set()
    if v:
        on_virtual = any(L <= ph[t] < L + v for t in needed)
        mine = (lambda p_: L <= p_ < L + v) if on_virtual else (lambda p_: p_ >= L + v)
    else:
        mine = lambda p_: p_ >= L                 # noqa: E731
    g = sum(1 for q in range(n) if mine(ph[q]))   # far positions of the class that is re-filled
    is_glob = [mine(ph[q]) for q in range(n)]
    frozen = [ph[q] >= L and not mine(ph[q]) for q in range(n)]       # the other class: stays where it is
    # farthest next use first; ties: keep what already is global (less traffic), then qubits above the contiguous run
    # of a tile (moving a lower bit cannot ride on a fused pass's permuted store), then canonical order
    # (a local qubit on the contiguous low bits of a tile cannot be moved by a fused pass's permuted store -- its remap
    # would cost a re-labelling pass of its own: with CONFIG['evict_foldable'] such a qubit is evicted only when nothing else is left)
    low = _UNFOLDABLE_BELOW if (_EVICT[0] and L >= 12) else 0      # (shards of at least a tile)
    order = sorted((q for q in range(n) if not frozen[q]),
                   key=lambda q: (1 if (not is_glob[q] and ph[q] < low) else 0, -nxt[q], 0 if is_glob[q] else 1, 0 if ph[q] >= 4 else 1, -q))
    new_global = set(order[:g])
    assert not ({t for t in needed if not frozen[t]} & new_global), 'gate needs more local qubits than a shard has'
    leaving = [q for q in range(n) if is_glob[q] and q not in new_global]
    entering = [q for q in new_global if not is_glob[q]]
    assert len(leaving) == len(entering) and leaving, 'remap requested although the gate is local'
    # a canonical global qubit (logical bit L + j) prefers its own rank bit j: cheaper to canonicalise later
    pairs, free_enter = [], list(entering)
    for lq in leaving:
        own = ph[lq]                                   # physical rank bit being vacated
        pick = next((eq for eq in free_enter if eq == own), free_enter[0])
        free_enter.remove(pick)
        pairs.append((lq, pick))
    return pairs


_ORDERS: dict = {}


def _structure(prims: Sequence[Prim]) -> tuple:
    """What the exchange schedule of a gate list depends on (no matrices): the key of the schedule caches."""
    return tuple((p.kind, tuple(p.targets), tuple(p.controls), p.mode, tuple(p.order)) for p in prims)


def _order_for_remaps(prims: Sequence[Prim], ph0: Sequence[int], n: int, L: int, v: int = 0,
                      structure: tuple | None = None) -> list[Prim]:
    """`_order_indices` applied; the order is cached by the gate list's structure and the starting placement (round 6: it
    is 10 ms of host time for the 1360 gates of the n = 34 benchmark circuit, in front of the step's first launch)."""
    key = (structure if structure is not None else _structure(prims), tuple(ph0), n, L, v, _EVICT[0])
    order = _ORDERS.get(key)
    if order is None:
        order = _order_indices(prims, ph0, n, L, v)
        if len(_ORDERS) >= 16:
            _ORDERS.pop(next(iter(_ORDERS)))
        _ORDERS[key] = order
    return [prims[i] for i in order]


def _order_indices(prims: Sequence[Prim], ph0: Sequence[int], n: int, L: int, v: int = 0) -> list[int]:
    """The gate list in an order that needs far fewer exchanges: list scheduling over the commutation DAG of the
    circuit (`fusion._Dag`: two gates commute when on every shared qubit both act diagonally, or both as functions of
    X) -- every gate that is ready and local under the current placement runs; only when ALL ready gates wait for a
    qubit on the rank bits does a remap happen (simulated here with the same farthest-next-use rule as `_plan_remap`).
    In program order a gate on a global qubit stops everything behind it, although most of what follows neither
    depends on it nor touches that qubit: on the benchmark circuit (depth 40) the exchange steps go 15 -> 4 (2 ranks),
    20 -> 5 (4), 22 -> 5 (8 ranks) and the bytes on the wire down by 73-78 %.  The re-ordering is exact (commuting
    operators), a pure function of the gate list and the starting placement: every rank computes the same order."""
    from . import fusion

    g = n - L
    ops = [fusion.PrimOp(p.kind, tuple(p.targets), tuple(p.controls), 0, p.mode) for p in prims]
    dag = fusion._Dag(ops, n)
    ph = list(ph0)
    retired = [False] * len(prims)
    order: list[int] = []
    inf = 1 << 60
    while dag.done < dag.n_ops:
        progressed = True
        while progressed:
            progressed = False
            for i in list(dag.ready):
                p = prims[i]
                if p.kind == 'diag' or all(ph[t] < L for t in p.targets):
                    order.append(i)
                    dag.retire(i)
                    retired[i] = True
                    progressed = True
        if dag.done >= dag.n_ops:
            break
        # every ready gate has a target on the rank bits: new global qubits = the ones not needed for longest
        nxt = [inf] * n
        left = n
        for j in range(dag.ready[0], len(prims)):
            if retired[j] or prims[j].kind == 'diag':
                continue
            for t in prims[j].targets:
                if nxt[t] == inf:
                    nxt[t] = j
                    left -= 1
            if left == 0:
                break
        if v:      # (virtual rank bits: one class of far positions is re-filled at a time, as in `_plan_remap`)
            waits = {t for i in dag.ready if prims[i].kind != 'diag' for t in prims[i].targets}
            on_virtual = any(L <= ph[t] < L + v for t in waits)
            mine = (lambda p_: L <= p_ < L + v) if on_virtual else (lambda p_: p_ >= L + v)
        else:
            mine = lambda p_: p_ >= L             # noqa: E731
        is_glob = [mine(ph[q]) for q in range(n)]
        frozen = [ph[q] >= L and not mine(ph[q]) for q in range(n)]
        low = _UNFOLDABLE_BELOW if (_EVICT[0] and L >= 12) else 0
        cand = sorted((q for q in range(n) if not frozen[q]),
                      key=lambda q: (1 if (not is_glob[q] and ph[q] < low) else 0, -nxt[q], 0 if is_glob[q] else 1, 0 if ph[q] >= 4 else 1, -q))
        new_global = set(cand[:sum(is_glob)])
        leaving = [q for q in range(n) if is_glob[q] and q not in new_global]
        entering = [q for q in new_global if not is_glob[q]]
        if not leaving:          # (cannot happen: some ready gate has a global target, and its next use is now)
            i = dag.ready[0]
            order.append(i)
            dag.retire(i)
            retired[i] = True
            continue
        for lq, eq in zip(leaving, entering):
            ph[lq], ph[eq] = ph[eq], ph[lq]
    return order


_PLACEMENTS: dict = {}


def _dry_canonicalize(ph: list[int], n: int, L: int) -> tuple[int, float]:
    """(exchanges, volume in shards) that `_canonicalize` would need from placement ``ph`` (no data; ``ph`` is updated):
    the same rounds -- every misplaced rank bit trades with its owner, or with a filler while the owner itself sits on
    another rank bit."""
    steps, vol = 0, 0.0
    for _ in range(4):
        misplaced = [q for q in range(n) if ph[q] >= L and ph[q] != q]
        if not misplaced:
            break
        used: set = set()
        pairs = []
        for lq in misplaced:
            owner = ph[lq]
            pick = owner if (ph[owner] < L and owner not in used) else next(q for q in range(L) if ph[q] < L and q not in used)
            used.add(pick)
            pairs.append((lq, pick))
        for lq, pick in pairs:
            ph[lq], ph[pick] = ph[pick], ph[lq]
        steps += 1
        vol += 1 - 0.5 ** len(pairs)
    return steps, vol


def _dry_remaps(prims: Sequence[Prim], ph0: Sequence[int], n: int, lr: int, v: int, restore: bool = False,
                trace: list | None = None) -> tuple[int, float]:
    """(exchanges of real rank bits, their volume in shards) of the remap schedule started from placement ``ph0`` -- the
    loop of `count_exchange_steps` without the statistics.  ``restore``: plus what the canonicalisation at the end of a
    drop-in forward (``keep_layout=False``) costs from where the schedule leaves the qubits.  ``trace``: a list that
    receives one (k, trades real rank bits) per remap of the gate schedule, in order."""
    L = lr + v
    ph = list(ph0)
    order = _order_for_remaps(prims, ph, n, lr, v)
    steps, vol, i = 0, 0.0, 0
    while i < len(order):
        p = order[i]
        if p.kind != 'diag' and any(ph[t] >= lr for t in p.targets):
            pairs = sorted(_plan_remap(ph, order, i, n, lr, v), key=lambda pr: ph[pr[0]])
            k = len(pairs)
            rb = [ph[lq] for lq, _ in pairs]
            ent = [ph[eq] for _, eq in pairs]
            new_local = {sp: d for d, sp in enumerate([b for b in range(lr) if b not in ent] + ent)}
            for q in range(n):
                if ph[q] < lr:
                    ph[q] = new_local[ph[q]]
            for j, (lq, eq) in enumerate(pairs):
                ph[eq], ph[lq] = rb[j], lr - k + j
            if trace is not None:
                # (k, trades real rank bits, the re-labelling in front of it can ride on a pass: no entering qubit on the
                # contiguous low bits of a tile)
                trace.append((k, rb[0] >= L, all(e >= _UNFOLDABLE_BELOW for e in ent) or lr < 12))
            if rb[0] >= L:
                steps += 1
                vol += 1 - 0.5**k
            continue
        i += 1
    if restore:
        cs, cv = _dry_canonicalize(ph, n, L)
        steps, vol = steps + cs, vol + cv
    return steps, vol


def initial_placement(prims: Sequence[Prim], n: int, L: int, v: int = 0, restore: bool = False,
                      structure: tuple | None = None) -> list[int]:
    """Where the qubits of a circuit that starts from |0..0> should sit at the start: |0..0> is the same vector under
    every permutation of the qubits (rank 0 holds the one non-zero amplitude at local index 0 in any of them), so the
    FIRST placement costs nothing -- no exchange, not even a re-labelling pass.  Candidates: the reference layout
    (wires 0 .. g-1 on the rank bits -- a layered circuit needs them within its first layer) and the placements that put
    g of the g + 3 qubits whose first non-diagonal gate comes last (farthest next use, asked at gate 0) on the rank bits
    and the next v on the virtual ones (CONFIG['virtual_bits']); each is dry-run through the whole remap schedule
    (`_dry_remaps`, ~10 ms) and the one with the fewest exchanges wins -- the reference layout unless another one saves a
    whole exchange.  Never worse than the reference start, typically one exchange and one stretch boundary less (n = 34 on 8 ranks:
    5 -> 4 exchanges, 35 -> 33 passes).  A pure function of the gate list, cached by its structure: every rank computes
    the same placement.  ``canonicalize`` restores the reference's order whenever somebody asks for it; ``restore`` (a
    drop-in forward, ``keep_layout=False``) charges every candidate the exchanges of that canonicalisation too, so that
    "never worse than the reference start" holds for the step as it runs."""
    from itertools import combinations

    g = n - L
    canonical = list(range(n))
    if g <= 0 or not prims:
        return canonical
    # (the tuple itself: a hash of strings is randomised per process, and a collision on one rank only would give the
    # ranks different placements)
    key = (n, L, v, bool(restore), CONFIG['horizon'], CONFIG['reorder'], _EVICT[0],
           structure if structure is not None else _structure(prims))
    hit = _PLACEMENTS.get(key)
    if hit is not None:
        return list(hit)
    nxt = _next_use(prims, 0, n)
    order = sorted(range(n), key=lambda q: (-nxt[q], 0 if q >= L else 1, -q))
    lr = L - v
    best = (_dry_remaps(prims, canonical, n, lr, v, restore), 0, canonical)
    for ci, pick in enumerate(combinations(order[:g + 3], g)):
        ph = list(canonical)
        rest = [q for q in order if q not in pick]
        for positions, want in ((range(L, n), list(pick)), (range(lr, L), rest[:v])):
            have = [q for q in range(n) if ph[q] in positions]
            leaving = [q for q in have if q not in want]
            entering = [q for q in want if q not in have]
            for lq, eq in zip(leaving, entering):
                ph[lq], ph[eq] = ph[eq], ph[lq]
        cand = (_dry_remaps(prims, ph, n, lr, v, restore), ci + 1, ph)
        # (fewer EXCHANGES, not merely less volume: a placement that only trims the volume was measured to cost more in
        # passes and un-folded re-labellings than it saves on the wire -- rehearsal of n = 34 / 8 ranks with virtual bits)
        if cand[0][0] < best[0][0] or (cand[0][0] == best[0][0] and best[1] > 0 and cand[0] < best[0]):
            best = cand
    if len(_PLACEMENTS) >= 32:
        _PLACEMENTS.pop(next(iter(_PLACEMENTS)))
    _PLACEMENTS[key] = list(best[2])
    return list(best[2])


#: The dry-run cost model behind `choose_virtual_bits`, in units of ONE PASS over the shard (read + write at the rate the
#: pass kernel reaches on a shard, 5.5 TB/s: profiles/r05/strong_rehearsal.txt).  A stretch boundary -- every remap, real
#: or virtual -- costs `boundary_passes` under-filled passes (measured: 35 passes in 6 stretches against 28 for the
#: unsharded plan of the same circuit); the wire of a k-qubit exchange of real rank bits takes shard / 2^k per link at
#: `link_GBs`, of which only the first row's share 2^-v is exposed with v virtual bits (all of it with v = 0).
MODEL = {'pass_GBs': 5500.0, 'link_GBs': 153.0, 'boundary_passes': 1.7}

_VBITS: dict = {}


def modelled_cost(trace: Sequence[tuple[int, bool]], v: int, first_is_local: bool) -> float:
    """Cost of a remap schedule (`_dry_remaps(trace=...)`) in passes over the shard: see `MODEL`."""
    wire_pass = MODEL['pass_GBs'] / (2.0 * MODEL['link_GBs'])       # one shard over ONE link, in passes
    cost, first = 0.0, first_is_local and v == 0
    for k, real, *rest in trace:
        cost += MODEL['boundary_passes']
        if rest and not rest[0]:
            cost += 1.0             # a re-labelling pass of its own in front of the exchange
        if real:
            if first:               # the first exchange behind reset() without the wire: a copy of 2^-k and a memset
                cost += 0.5
            else:
                cost += wire_pass / (1 << k) * (0.5 ** v)
            first = False
    return cost


_EVICTIONS: dict = {}


def choose_eviction(prims: Sequence[Prim], n: int, L: int, v: int = 0, fresh: bool = False, restore: bool = False,
                    structure: tuple | None = None) -> bool:
    """CONFIG['evict_foldable'] = None: both eviction rules dry-run through the whole schedule, the cheaper one by
    `modelled_cost` wins (ties: the foldable rule).  A pure function of the gate list: every rank chooses alike."""
    key = (n, L, v, fresh, restore, CONFIG['first_exchange_local'], CONFIG['initial_placement'], tuple(sorted(MODEL.items())),
           structure if structure is not None else _structure(prims))
    hit = _EVICTIONS.get(key)
    if hit is not None:
        return hit
    keep = _EVICT[0]
    costs = {}
    try:
        for rule in (True, False):
            _EVICT[0] = rule
            ph = (initial_placement(prims, n, L, v, restore=restore, structure=structure)
                  if (fresh and CONFIG['initial_placement']) else list(range(n)))
            trace: list = []
            _dry_remaps(prims, ph, n, L - v, v, trace=trace)
            costs[rule] = modelled_cost(trace, v, fresh and CONFIG['first_exchange_local'])
    finally:
        _EVICT[0] = keep
    if len(_EVICTIONS) >= 16:
        _EVICTIONS.pop(next(iter(_EVICTIONS)))
    _EVICTIONS[key] = costs[True] <= costs[False] + 1e-9
    return _EVICTIONS[key]


def choose_virtual_bits(prims: Sequence[Prim], n: int, L: int, candidates: Sequence[int] = (0, 1, 2), fresh: bool = False,
                        restore: bool = False, structure: tuple | None = None) -> int:
    """CONFIG['virtual_bits'] = None: v from the dry-run model (`modelled_cost`), never one the model puts behind v = 0
    (round 5 defaulted to 2 under RCCL; its own rehearsal of n = 34 on 8 ranks had v = 2 SLOWER than v = 0 for the slowest
    rank: the hidden wire was paid for with three times the launches and 60-70 ms of compute).  Every candidate's schedule
    is dry-run from the placement it would start from; ties go to the smaller v.  A pure function of the gate list and
    CONFIG: every rank chooses alike."""
    return _choose_virtual_bits(prims, n, L, candidates, fresh, restore, structure)[0]


def _choose_virtual_bits(prims, n, L, candidates=(0, 1, 2), fresh=False, restore=False, structure=None) -> tuple[int, dict]:
    key = (n, L, tuple(candidates), fresh, restore, CONFIG['first_exchange_local'], CONFIG['initial_placement'], _EVICT[0],
           tuple(sorted(MODEL.items())), structure if structure is not None else _structure(prims))
    hit = _VBITS.get(key)
    if hit is not None:
        return hit
    best, table = None, {}
    for v in candidates:
        if L - v < 1:
            continue
        ph = (initial_placement(prims, n, L, v, restore=restore, structure=structure)
              if (fresh and CONFIG['initial_placement']) else list(range(n)))
        trace: list = []
        _dry_remaps(prims, ph, n, L - v, v, trace=trace)
        cost = modelled_cost(trace, v, fresh and CONFIG['first_exchange_local'])
        table[v] = {'cost_in_passes': cost, 'remaps_real': sum(1 for t_ in trace if t_[1]),
                    'remaps_virtual': sum(1 for t_ in trace if not t_[1]),
                    'relabelling_passes': sum(1 for t_ in trace if not t_[2])}
        if best is None or cost < best[1] - 1e-9:
            best = (v, cost)
    if len(_VBITS) >= 16:
        _VBITS.pop(next(iter(_VBITS)))
    _VBITS[key] = (best[0] if best else 0, table)
    return _VBITS[key]


def virtual_bits_table(prims: Sequence[Prim], n: int, L: int, **kw) -> dict:
    """The model's table behind `choose_virtual_bits` (bench.py prints it): {v: cost in passes, remaps}, and the choice."""
    v, table = _choose_virtual_bits(prims, n, L, **kw)
    return {'chosen': v, 'candidates': table, 'model': dict(MODEL)}


def slice_bits_wanted(state: DistributedQubitState) -> int:
    """CONFIG['slice_exchange'] resolved: how many bits the passes around an exchange are sliced by (0 = off)."""
    nb = CONFIG['slice_exchange']
    if nb is None:
        overlaps = state.amps.is_cuda and (CONFIG['elide_exchange'] or (dist.is_initialized() and dist.get_backend() == 'nccl'))
        nb = 3 if overlaps else 0           # (three bits against two, rehearsal of n = 34 / 8 ranks: −4 .. −7 ms per step)
    return int(nb) if (state.batch is None and _vbits(state) == 0) else 0


def _slice_qubits(ph: list[int], prims: Sequence[Prim], i: int, n: int, L: int, pairs, nbits: int) -> list[int]:
    """The qubits the passes around the exchange ``pairs`` are sliced by: local, staying local, movable by a permuted store,
    and -- after the evicted ones -- needed LAST (farthest next non-diagonal use from gate ``i`` on): neither the gates
    left for the last pass in front of the exchange nor the first ones behind it have any business with them.  A pure
    function of the gate list and the placement: every rank picks the same."""
    if nbits <= 0:
        return []
    nxt = _next_use(prims, i, n)
    leaving_local = {eq for _, eq in pairs}
    cand = sorted((q for q in range(n) if _UNFOLDABLE_BELOW <= ph[q] < L and q not in leaving_local), key=lambda q: (-nxt[q], -q))
    return cand[:nbits]


class _Pending(list):
    """The local gates of the stretch under way (physical positions, localized) with, beside each, the LOGICAL primitive it
    was made from (``src``), and ALL logical primitives of the stretch in order (``every``: also those that are nothing on
    this rank -- a control on a 0 rank bit) -- what `_defer_tail` decides on and re-queues behind the exchange.  ``src`` is
    None once the list has been rewritten wholesale (virtual rank bits re-localize it): nothing is deferred then."""

    def __init__(self) -> None:
        super().__init__()
        self.src: list | None = []
        self.every: list = []

    def add(self, local: Prim | None, logical: Prim) -> None:
        self.every.append(logical)
        if local is not None:
            super().append(local)
            if self.src is not None:
                self.src.append(logical)

    def clear(self) -> None:
        super().clear()
        self.src = []
        self.every = []

    def __setitem__(self, key, value) -> None:
        super().__setitem__(key, value)
        self.src = None


def _defer_tail(state: DistributedQubitState, pending: list[Prim], pairs) -> list[Prim]:
    """CONFIG['defer_tail']: if the last pass of the stretch would be under-filled and none of its gates targets a qubit that
    is about to leave for the rank bits (``pairs``), take those gates out of ``pending`` and return their logical primitives:
    the caller queues them again behind the exchange.

    EVERY RANK MUST DEFER THE SAME GATES -- a gate one rank applies before the exchange and its peer behind it is applied
    twice to some chunks and never to others -- while the plans differ between ranks (a control on a rank bit makes a gate
    nothing on half of them).  So the decision is taken on the stretch as the rank with ALL rank bits set sees it (every
    rank-controlled gate present: a superset of everybody's gates; `executor.tail_of_last_pass` on that list), from
    quantities every rank has alike: the logical gate list, the placement, the number of exchanges so far.  The tail of a
    superset's schedule is closed under "comes later and does not commute" for every subset, so the move is legal on
    every rank whether or not the gates form ITS last pass."""
    cap = int(CONFIG['defer_tail'] or 0)
    if (not cap or not isinstance(pending, _Pending) or pending.src is None or len(pending.every) < 32
            or _vbits(state) or _SWEEP['grads'] is not None
            or _view(state).shape[0] != 1      # (batched shards: measured neutral -- weak series n = 31: 350 / 365 ms with, 350 / 357 without)
            or LAST_RUN['remaps'] < 2):         # (the first two stretches run behind |0..0> / with known-zero masks: cheap anyway)
        return []
    ph = _phys(state)
    L = state.log_num_amps_per_node
    leaving = {ph[eq] for _, eq in pairs}
    ones = state.world_size - 1
    canon, index = [], []
    for m, q in enumerate(pending.every):
        loc = _localize_at(L, ones, _translate(q, ph))
        if isinstance(loc, Prim):
            canon.append(loc)
            index.append(m)
    x = _view(state)
    x = x[_row_groups(state)[0]]            # (a batched shard runs its stretches group by group: the plan is the group's)
    tail = executor.tail_of_last_pass(x, canon, amps=_view(state).numel(), max_gates=cap)
    if not tail or any(set(canon[m].targets) & leaving for m in tail):
        return []
    carry = [pending.every[index[m]] for m in tail]
    gone = {id(q) for q in carry}
    rest = [(p_, s_) for p_, s_ in zip(pending, pending.src) if id(s_) not in gone]
    every = [q for q in pending.every if id(q) not in gone]
    pending.clear()
    for p_, s_ in rest:
        list.append(pending, p_)
        pending.src.append(s_)
    pending.every = every
    LAST_RUN['deferred_tails'] += 1
    LAST_RUN['deferred_gates'] += len(carry)
    return carry


def _remap_for(state: DistributedQubitState, prims: Sequence[Prim], i: int, pending: list[Prim]) -> list[Prim]:
    """The remap that makes gate ``i`` local, behind the local gates ``pending``.  Returns the logical primitives of a
    deferred tail (`_defer_tail`): the caller localizes them under the new layout before anything else."""
    L = state.log_num_amps_per_node - _vbits(state)
    pairs = _plan_remap(_phys(state), prims, i, state.nqubit, L, _vbits(state))
    carry = _defer_tail(state, pending, pairs)
    nb = slice_bits_wanted(state)
    _remap(state, pairs, pending, _slice_qubits(_phys(state), prims, i, state.nqubit, L, pairs, nb) if nb else ())
    return carry


def count_exchange_steps(prims: Sequence[Prim], n: int, g: int, virtual_bits: int = 0, reorder: bool = False,
                         placement: bool = False) -> dict:
    """Dry run of both modes on a gate list (no data): number of exchange steps and the volume each rank
    sends, in units of one shard.  Used by tests and to size the design.

    ``virtual_bits`` = v (CONFIG['virtual_bits']): the remap mode with the top v local bits of every shard as rank bits
    of a virtual world.  ``remap_steps`` then counts both kinds of step: ``virtual_steps`` of them trade virtual bits
    only (a re-labelling of the shard that rides on a pass: nothing on the wire), the others trade real rank bits, and
    their wire volume splits into ``hidden_wire_volume`` -- it travels while the other rows of the shard compute: all
    but the first row's share, 1 - 2^-v of it -- and ``exposed_wire_volume``.  v = 0: everything is exposed (an
    un-batched shard has nothing to overlap with).
    ``reorder``: the gate list in commutation-DAG order first, as `dist_run` runs it.  ``placement``: the free first
    placement of a circuit that starts from |0..0> (`initial_placement`)."""
    L = n - g
    pw_steps, pw_vol = 0, 0.0
    for p in prims:
        if p.kind != 'diag' and any(t >= L for t in p.targets):
            nglob = sum(1 for t in p.targets if t >= L)
            lc = sum(1 for c in p.controls if c < L)
            if len(p.targets) == 1:
                pw_steps += 1
                pw_vol += 0.5**lc
            else:
                pw_steps += 2 * nglob
                pw_vol += 2 * nglob * 0.5
    v = int(virtual_bits)
    lr = L - v                                   # bits of a row
    ph = initial_placement(prims, n, L, v) if placement else list(range(n))
    if reorder:
        prims = _order_for_remaps(prims, ph, n, lr, v)
    rm_steps, rm_vol, i = 0, 0.0, 0
    v_steps, hidden, exposed = 0, 0.0, 0.0
    while i < len(prims):
        p = prims[i]
        if p.kind != 'diag' and any(ph[t] >= lr for t in p.targets):
            pairs = _plan_remap(ph, prims, i, n, lr, v)
            pairs = sorted(pairs, key=lambda pr: ph[pr[0]])
            k = len(pairs)
            rb = [ph[lq] for lq, _ in pairs]
            ent = [ph[eq] for _, eq in pairs]
            rest = [b for b in range(lr) if b not in ent]
            new_local = {sp: d for d, sp in enumerate(rest + ent)}
            for q in range(n):
                if ph[q] < lr:
                    ph[q] = new_local[ph[q]]
            for j, (lq, eq) in enumerate(pairs):
                ph[eq], ph[lq] = rb[j], lr - k + j
            k_real = sum(1 for r in rb if r >= L)          # real rank bits among the leaving positions
            assert k_real in (0, k), 'a remap trades real OR virtual rank bits'
            rm_steps += 1
            if k_real == 0:                                  # a local re-labelling (rides on a pass): nothing on the wire
                v_steps += 1
                continue
            wire = 1 - 0.5**k_real                           # of every row, i.e. of the shard
            rm_vol += wire
            hidden += wire * (1 - 0.5**v)
            exposed += wire * 0.5**v
            continue
        i += 1
    return {'pairwise_steps': pw_steps, 'pairwise_volume': pw_vol, 'remap_steps': rm_steps, 'remap_volume': rm_vol,
            'virtual_bits': v, 'virtual_steps': v_steps, 'hidden_wire_volume': hidden, 'exposed_wire_volume': exposed}


def canonicalize(state: DistributedQubitState) -> DistributedQubitState:
    """Restore phys[q] == q: at most two all-to-all steps for the rank bits, then one local re-labelling."""
    if _is_canonical(state):
        return state
    with _raw(state):
        return _canonicalize(state)


def _canonicalize(state: DistributedQubitState) -> DistributedQubitState:
    n, L = state.nqubit, state.log_num_amps_per_node
    ph = _phys(state)
    for _ in range(4):
        glob = [q for q in range(n) if ph[q] >= L]
        misplaced = [q for q in glob if ph[q] != q]
        if not misplaced:
            break
        pairs, used = [], set()
        for lq in misplaced:
            owner = ph[lq]                              # logical qubit that belongs on this rank bit
            if ph[owner] < L and owner not in used:
                pick = owner
            else:                                       # owner itself is on the move: park a filler there
                pick = next(q for q in range(L) if ph[q] < L and q not in used)
            used.add(pick)
            pairs.append((lq, pick))
        _exchange_qubits(state, pairs)
    assert all(ph[q] == q for q in range(L, n)), 'rank bits not canonical after 4 exchange rounds'
    src_of_dst = [0] * L
    for q in range(L):
        src_of_dst[q] = ph[q]                           # destination bit q must receive logical qubit q
    _permute_local(state, src_of_dst)
    assert _is_canonical(state)
    return state


# ---------------------------------------------------------------------------------------------------
# public entry points
def dist_apply_prims(state: DistributedQubitState, prims: Sequence[Prim], mode: str | None = None,
                     keep_layout: bool = False, force_mode: bool = False,
                     expect_z: Sequence[int] | None = None, fresh_zero: bool = False) -> DistributedQubitState:
    """Apply kernel primitives (logical bit positions) to the sharded state, fusing local stretches.
    Unless ``keep_layout`` is set the canonical qubit order is restored before returning.  ``force_mode``: ``mode`` also
    for short gate lists (which otherwise go gate by gate, pairwise exchanges).  ``fresh_zero``: the caller has just
    ``reset()`` the state -- it is |0..0>: rank 0 holds one 1, everybody else zeros -- and the first local stretch makes
    use of it (`_remap`, `_flush`)."""
    with _raw(state):
        if fresh_zero and executor.CONFIG['zero_state'] and _SWEEP['grads'] is None and _is_canonical(state):
            state.__dict__['_fresh_zero'] = True
            state.__dict__['_behind_reset'] = True      # (until the first exchange of real rank bits)
        elif state.__dict__.get('_lazy_zero'):
            _materialize_zeros(state)                   # (a lazy reset() nobody takes up: real zeros)
        try:
            return _dist_apply_prims(state, prims, mode, keep_layout, force_mode, expect_z)
        finally:
            _materialize_zeros(state)                   # (a no-op unless logical zeros are left: nobody outside sees them)
            for key in ('_fresh_zero', '_zero_shard', '_behind_reset', '_known_zero_local'):
                state.__dict__.pop(key, None)


def _dist_apply_prims(state: DistributedQubitState, prims: Sequence[Prim], mode: str | None, keep_layout: bool,
                      force_mode: bool = False, expect_z: Sequence[int] | None = None) -> DistributedQubitState:
    state.__dict__.pop('_expz', None)        # (cached expectation values belong to the state as it was)
    for k in LAST_RUN:
        LAST_RUN[k] = 0
    mode = mode or CONFIG['mode']
    if (len(prims) < CONFIG['remap_min_prims'] and not force_mode) or state.world_size == 1:
        mode = 'pairwise'
    if mode == 'pairwise' and not _is_canonical(state):
        canonicalize(state)
    # virtual rank bits: an un-batched shard of a forward circuit, rows of at least one tile
    _EVICT[0] = True if CONFIG['evict_foldable'] is None else bool(CONFIG['evict_foldable'])
    vb = CONFIG['virtual_bits']
    tile = executor._geometry(state.amps.dtype == torch.complex128).m
    eligible = mode == 'remap' and state.batch is None and _SWEEP['grads'] is None and state.amps.ndim == 1
    if vb is None:
        # from the dry-run model, and only where an exchange can overlap with compute at all -- RCCL on device shards
        # (asynchronous, on the group's own stream), or the rehearsal of such a job; gloo is synchronous and host-staged
        overlaps = state.amps.is_cuda and (CONFIG['elide_exchange'] or (dist.is_initialized() and dist.get_backend() == 'nccl'))
        vb = 0
        if overlaps and eligible:
            cands = [v for v in (0, 1, 2) if state.log_num_amps_per_node - v >= tile]
            vb = choose_virtual_bits(prims, state.nqubit, state.log_num_amps_per_node, cands,
                                     fresh=bool(state.__dict__.get('_fresh_zero')), restore=not keep_layout)
    vb = int(vb or 0)
    if vb and not (eligible and state.log_num_amps_per_node - vb >= tile):
        vb = 0
    if CONFIG['evict_foldable'] is None and mode == 'remap' and state.world_size > 1 and state.log_num_amps_per_node - vb >= 12:
        _EVICT[0] = choose_eviction(prims, state.nqubit, state.log_num_amps_per_node, vb,
                                    fresh=bool(state.__dict__.get('_fresh_zero')), restore=not keep_layout)
    LAST_RUN['virtual_bits'] = vb
    if mode != 'remap':
        state.__dict__.pop('_fresh_zero', None)       # (gate-by-gate exchanges: not for them)
        state.__dict__.pop('_behind_reset', None)
    if vb:
        _settle(state)
        state.__dict__['_vbits'] = vb
    if (mode == 'remap' and vb == 0 and CONFIG['first_exchange_local'] and state.__dict__.get('_fresh_zero')
            and state.world_size > 1):
        state.__dict__['_as_rank0'] = True       # (until the first exchange: `_remap` / `_flush` take it off)
    elif state.__dict__.get('_lazy_zero') and not (state.rank == 0 and state.__dict__.get('_fresh_zero') and mode == 'remap' and vb == 0):
        # a lazily reset shard somebody will read unmasked: the ranks that hold zeros and send them over the wire, rows of
        # virtual rank bits, gate-by-gate exchanges
        _materialize_zeros(state)
    try:
        return _dist_apply_loop(state, prims, mode, keep_layout, expect_z)
    finally:
        state.__dict__.pop('_as_rank0', None)
        if vb:
            _settle(state)
            state.__dict__.pop('_vbits', None)


def _dist_apply_loop(state: DistributedQubitState, prims: Sequence[Prim], mode: str, keep_layout: bool,
                     expect_z: Sequence[int] | None) -> DistributedQubitState:
    vb = _vbits(state)
    structure = _structure(prims) if mode == 'remap' else None
    if (mode == 'remap' and CONFIG['initial_placement'] and state.__dict__.get('_fresh_zero') and _is_canonical(state)
            and state.world_size > 1):
        # behind reset(): the first placement is free (see `initial_placement`)
        state.__dict__['_phys'] = initial_placement(prims, state.nqubit, state.log_num_amps_per_node, vb,
                                                    restore=not keep_layout, structure=structure)
    if mode == 'remap' and CONFIG['reorder']:
        prims = _order_for_remaps(prims, _phys(state), state.nqubit, state.log_num_amps_per_node - vb, vb, structure)
    pending = _Pending()
    i, nprims = 0, len(prims)
    while i < nprims:
        p = prims[i] if mode == 'pairwise' else _translate(prims[i], _phys(state))
        if vb:
            # virtual rank bits: what a gate is on THIS rank depends on how its stretch ends -- row by row in front of
            # an exchange of real rank bits, on the whole shard otherwise -- so it is localized when the stretch runs
            if p.kind != 'diag' and any(t >= state.log_num_amps_per_node - vb for t in p.targets):
                _remap_for(state, prims, i, pending)
            else:
                pending.add(p, prims[i])
                i += 1
            continue
        local = _localize(state, p)
        if local is None:
            pending.add(None, prims[i])      # (nothing on this rank -- but part of the stretch: `_defer_tail`)
            i += 1
            continue
        if isinstance(local, Prim):
            pending.add(local, prims[i])
            i += 1
            continue
        if mode == 'pairwise':
            _flush(state, pending)
            _exchange_prim(state, p)
            i += 1
        else:
            carry = _remap_for(state, prims, i, pending)     # local gates so far + exchange; then gate i under the new layout
            for q in carry:                                  # (a deferred tail: first in line behind the exchange)
                loc = _localize(state, _translate(q, _phys(state)))
                assert loc != 'exchange', 'a deferred gate targets a qubit that left for the rank bits'
                pending.add(loc if isinstance(loc, Prim) else None, q)
    if vb:                        # back to ONE shard of 2^L amplitudes (the rows are its top local index bits): the last
        _settle(state)            # stretch runs on the whole shard
        state.__dict__.pop('_vbits', None)
        pending[:] = [q for q in (_localize(state, p) for p in pending) if q is not None]
    if expect_z:
        # the Z-type observables of the circuit: reduced from the registers of the last local pass (DQ_FG_EXPZ)
        holder, signs = _expect_z_local(state, expect_z)
        had = bool(pending)
        _flush(state, pending, holder)
        _settle(state)
        _expect_z_finish(state, expect_z, holder if had else None, signs)
    else:
        _flush(state, pending)
        _settle(state)
    if not keep_layout:
        canonicalize(state)
    return state


def dist_gate(state: DistributedQubitState, gate) -> DistributedQubitState:
    """``Gate.forward`` on a DistributedQubitState (reference: operation.py:265-281, gate.py:77-85,
    gate.py:2008-2013)."""
    with torch.no_grad():
        return dist_apply_prims(state, gate.prims(decompose=True))


def dist_run(state: DistributedQubitState, operators, keep_layout: bool = False,
             expect_z: Sequence[int] | None = None, fresh_zero: bool = False) -> DistributedQubitState:
    """Whole circuit on the sharded state (reference: circuit.py:1655-1675).  ``keep_layout``: the qubits stay where
    the last remap put them; ``state.amps`` restores the reference's order when somebody reads it."""
    prims: list[Prim] = []
    for op in operators:
        prims.extend(op.prims(decompose=True))
    with torch.no_grad():
        return dist_apply_prims(state, prims, keep_layout=keep_layout, expect_z=expect_z, fresh_zero=fresh_zero)


def dist_swap_gate(state: DistributedQubitState, qb1: int, qb2: int) -> DistributedQubitState:
    """SWAP of two qubits given as bit positions (reference: distributed.py:130-159)."""
    if qb1 > qb2:
        qb1, qb2 = qb2, qb1
    L = state.log_num_amps_per_node
    if qb2 < L:
        x = state.amps.new_tensor([[0, 1], [1, 0]])
        _flush(state, [Prim('x', x, (qb2,), (qb1,)), Prim('x', x, (qb1,), (qb2,)), Prim('x', x, (qb2,), (qb1,))])
    elif qb1 >= L:
        _swap_global_global(state, qb1, qb2)
    else:
        _swap_local_global(state, qb1, qb2)
    return state


def expect_pauli_dist(state: DistributedQubitState, observable) -> torch.Tensor:
    """Re <psi|P|psi> of a Pauli-string observable on the sharded state without autograd and without
    a copy of the state when no X / Y factor sits on a global qubit: Z factors on global qubits are a
    sign that depends on the rank only.  Otherwise P|psi> is built through the exchange path.
    Works on batched shards; returns a 0-dim tensor (or (B,) for batched shards)."""
    L = state.log_num_amps_per_node
    xmask, zmask = observable.pauli_masks()
    ph = _phys(state)            # the masks are in logical qubits; the shard may be in another order (keep_layout)

    def placed(mask: int) -> int:
        return sum(1 << ph[q] for q in range(state.nqubit) if (mask >> q) & 1)

    px, pz = placed(xmask), placed(zmask)
    if px >> L == 0:
        with _raw(state):
            sign = -1.0 if bin((pz >> L) & state.rank).count('1') & 1 else 1.0
            val = backend.expect_pauli(_view(state), px, pz & ((1 << L) - 1)) * sign     # (B,) float64
            if state.world_size > 1:
                _all_reduce(val)
            val = val.to(state.amps.real.dtype)
        return val[0] if state.batch is None else val
    from copy import deepcopy

    canonicalize(state)

    lam = deepcopy(state)
    dist_apply_prims(lam, observable.prims())
    return inner_product_dist(lam, state).real


def inner_product_dist(bra: DistributedQubitState, ket: DistributedQubitState) -> torch.Tensor:
    """<bra|ket> over all shards (reference: distributed.py:288-294)."""
    val = backend.inner(_view(bra), _view(ket))                       # (B,) complex128
    if bra.world_size > 1:
        buf = torch.view_as_real(val.clone())
        _all_reduce(buf)
        val = torch.view_as_complex(buf)
    val = val.to(bra.amps.dtype)
    return val[0] if bra.batch is None else val


def measure_dist(state: DistributedQubitState, shots: int = 1024, with_prob: bool = False,
                 wires: int | list[int] | None = None, block_size: int = 2**24) -> dict:
    """Sample bit strings from the sharded state; the result lives on rank 0 (other ranks return {})
    (reference: distributed.py:205-285)."""
    assert state.batch is None, 'measure_dist works on an un-batched sharded state'
    if state.world_size == 1:
        return measure(state.amps, shots, with_prob, wires, False, block_size)
    n, L, g = state.nqubit, state.log_num_amps_per_node, state.log_num_nodes
    if isinstance(wires, int):
        wires = [wires]
    wires = sorted(wires) if wires is not None else list(range(n))
    nb = len(wires)
    bits = [n - 1 - w for w in wires]                     # MSB first
    local_bits = [b for b in bits if b < L]
    glob_bits = [b for b in bits if b >= L]
    assert len(local_bits) <= 24, 'too many measured local wires for a replicated marginal'
    # marginal over the measured local bits on this rank, then scatter into the slot the rank bits select
    view = _view(state)
    if len(local_bits) == L:
        part = backend.probs(view).to(torch.float64).reshape(-1)
    elif local_bits and len(local_bits) <= 12:
        part = backend.marginal(view, local_bits).reshape(-1)
    elif local_bits:
        p = backend.probs(view).reshape([2] * L)
        axes = [L - 1 - b for b in local_bits]
        rest = [i for i in range(L) if i not in axes]
        part = p.permute(axes + rest).reshape(1 << len(local_bits), -1).sum(-1).to(torch.float64)
    else:
        part = backend.probs(view).sum().reshape(1).to(torch.float64)
    probs = torch.zeros(1 << nb, dtype=torch.float64, device=part.device)
    gsel = 0
    for b in glob_bits:                                   # glob_bits are the leading outcome bits
        gsel = (gsel << 1) | get_bit(state.rank, b - L)
    width = 1 << len(local_bits)
    probs[gsel * width : (gsel + 1) * width] = part
    dist.all_reduce(probs, dist.ReduceOp.SUM)
    if state.rank != 0:
        return {}
    samples = Counter(block_sample(probs.to(state.amps.real.dtype), shots, block_size))
    results = {bin(k)[2:].zfill(nb): v for k, v in samples.items()}
    if with_prob:
        for k in results:
            results[k] = results[k], probs[int(k, 2)].item()
    return results
