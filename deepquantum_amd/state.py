"""State containers: ``QubitState`` (dense) and ``DistributedQubitState`` (index-bit sharded),
API-compatible with the reference's state.py:14-78 and :342-383."""

from __future__ import annotations

from typing import Any

import torch
from torch import nn

from . import _functorch
from .bitmath import is_power_of_2, log_base2, power_of_2
from .communication import comm_get_rank, comm_get_world_size
from .qmath import amplitude_encoding, is_density_matrix
from .utils import complex_apply


def tensor_version(t: torch.Tensor):
    """``t._version``, or None for a tensor that has none (inference tensors)."""
    return None if torch.is_inference(t) else t._version


_PRIVATE_REFCOUNT: list = []


def _refcount_of_a_private_tensor() -> int:
    """``sys.getrefcount`` of a tensor held by a dict and one local only -- the situation `_rearm` tests for -- measured
    once on a throwaway tensor instead of assuming CPython's "3" (interpreters with borrowed stack references count
    differently)."""
    if not _PRIVATE_REFCOUNT:
        import sys

        def probe() -> int:
            bufs = {'state': torch.zeros(1)}
            t = dict.get(bufs, 'state')
            return sys.getrefcount(t)

        _PRIVATE_REFCOUNT.append(probe())
    return _PRIVATE_REFCOUNT[0]


class _ComplexBuffers(nn.Module):
    """nn.Module whose complex buffers follow ``.to(real dtype)`` to the matching complex dtype."""

    _complex_names: tuple[str, ...] = ()

    def _apply(self, fn: Any, *args, **kwargs):
        held = {k: self._buffers.pop(k) for k in self._complex_names if self._buffers.get(k) is not None}
        super()._apply(fn, *args, **kwargs)
        for key, value in complex_apply(fn, held).items():
            self.register_buffer(key, value)
        return self


class QubitState(_ComplexBuffers):
    """|psi> of ``nqubit`` qubits as a complex64 (2**n, 1) buffer (or rho, (2**n, 2**n), with ``den_mat``): ``'zeros'``, ``'equal'``,
    ``'entangle'``/``'GHZ'``/``'ghz'`` or user amplitudes (amplitude-encoded)."""

    _complex_names = ('state',)

    def __init__(self, nqubit: int = 1, state: Any = 'zeros', den_mat: bool = False) -> None:
        super().__init__()
        self.__dict__['_buffers'] = _WatchedBuffers()       # (who is handed the buffer decides the |0..0> claim: below)
        self.nqubit = nqubit
        self.den_mat = den_mat
        dim = 2**nqubit
        if isinstance(state, str):
            if state == 'zeros':
                vec = torch.zeros((dim, 1), dtype=torch.cfloat)
                vec[0] = 1
            elif state == 'equal':
                vec = nn.functional.normalize(torch.ones((dim, 1), dtype=torch.cfloat), p=2, dim=-2)
            elif state in ('entangle', 'GHZ', 'ghz'):
                vec = torch.zeros((dim, 1), dtype=torch.cfloat)
                vec[0] = 1 / 2**0.5
                vec[-1] = 1 / 2**0.5
            else:
                raise ValueError(f'unknown state name {state!r}')
        else:
            if not isinstance(state, torch.Tensor):
                state = torch.tensor(state, dtype=torch.cfloat)
            if den_mat and state.shape[-1] == dim and is_density_matrix(state):
                self.register_buffer('state', state)      # already a density matrix (reference: state.py:57-58)
                return
            ndim = state.ndim
            vec = amplitude_encoding(data=state, nqubit=nqubit)
            if vec.ndim > ndim:
                vec = vec.squeeze(0)
        if den_mat:
            vec = vec @ vec.mH
        self.register_buffer('state', vec)
        if isinstance(state, str) and state == 'zeros':
            self._mark_zero_state()

    def forward(self) -> None:
        pass

    # |0..0> (or |0..0><0..0|) as made by the constructor is worth knowing to the executor: the first fused passes of a
    # circuit skip everything that is still known to be zero (executor.CONFIG['zero_state']).  The claim has to survive
    # what a version counter does not see -- ``state.data[i] = 1``, a numpy alias: writes that bump nothing -- so it is
    # tied to WHO HOLDS THE TENSOR instead: the buffer lives in a watched ``_buffers`` dict (`_WatchedBuffers`), and
    # handing it out to anybody but this package (``qs.state``, ``buffers()``, ``state_dict()``, ``_buffers['state']``)
    # or replacing it ends the claim.  It comes back at the next forward only if nobody holds the tensor or an alias of
    # its storage any more AND a look at the device memory says it still is |0..0> (`_rearm`: one reduction and one
    # host synchronisation, once per such event).
    def _mark_zero_state(self) -> None:
        import weakref

        bufs = self.__dict__['_buffers']
        t = dict.get(bufs, 'state')
        if t is None or _functorch.is_wrapped_tensor(t):      # (made inside a torch.func transform)
            bufs.zero_mark = None
            return
        bufs.zero_mark = (weakref.ref(t), tensor_version(t), t.data_ptr())
        bufs.rearm = False
        bufs.was_zero = True

    def is_zero_state(self) -> bool:
        """True while ``state`` is still the |0..0> the constructor made (``state='zeros'``), on whatever device / in
        whatever precision ``.to()`` has put it since, and nobody outside the package has been handed the tensor."""
        bufs = self.__dict__['_buffers']
        if getattr(bufs, 'zero_mark', None) is None and getattr(bufs, 'rearm', False):
            self._rearm()
        mark = getattr(bufs, 'zero_mark', None)
        t = dict.get(bufs, 'state')
        return (mark is not None and t is not None and mark[0]() is t and mark[1] is not None and mark[1] == tensor_version(t)
                and mark[2] == t.data_ptr())

    def invalidate(self) -> None:
        """Say that the buffer may have been written to behind PyTorch's back: ends the |0..0> claim until a forward
        has verified the memory again (see `_rearm`).  Never needed for writes through the tensor API."""
        bufs = self.__dict__['_buffers']
        bufs.zero_mark = None
        bufs.rearm = bufs.was_zero

    def _rearm(self) -> None:
        """The claim was ended by somebody who looked at the buffer.  It is taken up again iff (i) nobody holds the tensor
        object (Python reference count) or another view of its storage (storage use count) any more -- so that from here
        on every way to the memory leads through the watched dict again -- and (ii) the memory IS |0..0>: exactly one
        non-zero real component, and it is the real part of element 0, equal to 1."""
        import sys

        bufs = self.__dict__['_buffers']
        t = dict.get(bufs, 'state')
        use_count = getattr(torch._C, '_storage_Use_Count', None)
        if (t is None or use_count is None or t.numel() == 0 or not t.is_complex() or torch.is_inference(t)
                or _functorch.is_wrapped_tensor(t) or t.requires_grad):
            bufs.rearm = False
            return
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            return                                   # (stays pending: the look at the memory is a host synchronisation)
        if sys.getrefcount(t) != _refcount_of_a_private_tensor():      # the dict, ``t`` and the argument: somebody else holds it
            return                                   # (stays pending: cheap, no device work)
        storage = t.untyped_storage()
        if use_count(storage._cdata) != 2:           # the tensor and ``storage``: a view / .data / numpy alias is alive
            return
        del storage
        bufs.rearm = False
        flat = torch.view_as_real(t.detach().reshape(-1))
        ok = bool(((torch.count_nonzero(flat) == 1) & (flat[0, 0] == 1)).item())
        del flat
        if ok:
            self._mark_zero_state()

    def _state_and_claim(self) -> tuple[torch.Tensor, bool]:
        """(the buffer, whether it is known to be |0..0>) for the package's own use: does not count as handing it out."""
        zero = self.is_zero_state()
        return dict.get(self.__dict__['_buffers'], 'state'), zero

    def _apply(self, fn: Any, *args, **kwargs):
        was = self.is_zero_state()
        bufs = self.__dict__['_buffers']
        with bufs.internal():
            super()._apply(fn, *args, **kwargs)
        bufs.zero_mark = None
        if was:                      # (moving or converting |0..0> keeps it |0..0>)
            self._mark_zero_state()
        return self

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_buffers'] = _WatchedBuffers(dict.items(d['_buffers']))    # (no claim travels: a weak reference does not
        return d                                                        #  pickle; a copy starts without one)

    def __setstate__(self, state):
        super().__setstate__(state)
        if not isinstance(self.__dict__['_buffers'], _WatchedBuffers):
            self.__dict__['_buffers'] = _WatchedBuffers(self.__dict__['_buffers'])


class _WatchedBuffers(dict):
    """``_buffers`` of a `QubitState`.  Reading the ``'state'`` entry, iterating over the values or replacing the entry
    from outside an `internal` section ends the owner's |0..0> claim (``zero_mark``) and asks for a verification at
    the next forward (``rearm``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.zero_mark = None       # (weak reference to the tensor, its version, its address) while the claim stands
        self.rearm = False          # the claim was ended from outside: verify at the next forward
        self.was_zero = False       # the owner has held |0..0> at some point
        self.depth = 0              # nesting of `internal` sections

    class _Section:
        def __init__(self, d):
            self.d = d

        def __enter__(self):
            self.d.depth += 1

        def __exit__(self, *exc):
            self.d.depth -= 1
            return False

    def internal(self):
        return _WatchedBuffers._Section(self)

    def _handed_out(self, key='state') -> None:
        if key == 'state' and not self.depth:
            self.zero_mark = None
            self.rearm = self.was_zero       # (a state that never was |0..0> is not looked at)

    def __getitem__(self, key):
        self._handed_out(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._handed_out(key)
        return dict.get(self, key, default)

    def pop(self, key, *default):
        self._handed_out(key)
        return dict.pop(self, key, *default)

    def __setitem__(self, key, value):
        self._handed_out(key)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._handed_out(key)
        dict.__delitem__(self, key)

    def items(self):
        self._handed_out()
        return dict.items(self)

    def values(self):
        self._handed_out()
        return dict.values(self)

    def copy(self):
        self._handed_out()
        return dict(dict.items(self))

    def __reduce__(self):
        self._handed_out()
        return (_WatchedBuffers, (list(dict.items(self)),))


class DistributedQubitState(_ComplexBuffers):
    """Shard of an n-qubit state: rank r owns global indices [r * 2^L, (r+1) * 2^L), L = n - log2(W),
    i.e. wires 0..log2(W)-1 are the global qubits.  ``amps`` is the shard, ``buffer`` the receive
    buffer of the pairwise exchanges (reference: state.py:342-383)."""

    _complex_names = ('amps', 'buffer')

    #: shards of more amplitudes than this are not built until ``reset()`` (the first forward), on the device the
    #: state has been moved to by then: a 2^31-amplitude shard (16 GiB + 16 GiB receive buffer) must not be
    #: allocated in host memory by every rank first and copied over PCIe
    LAZY_AMPS = 1 << 24

    #: (world size, rank) of a world that is NOT there: states made while this is set are the shard of that rank -- for
    #: rehearsing one rank's schedule of a multi-GPU run on one GPU with the exchanges left out
    #: (distributed.CONFIG['elide_exchange'], bench.py --rehearse-rank).  None = ask the process group.
    REHEARSE: tuple[int, int] | None = None

    def __init__(self, nqubit: int, batch: int | None = None, device: Any = None, dtype: torch.dtype = torch.cfloat) -> None:
        super().__init__()
        if self.REHEARSE is not None:
            self.world_size, self.rank = self.REHEARSE
        else:
            self.world_size = comm_get_world_size()
            self.rank = comm_get_rank()
        assert is_power_of_2(self.world_size)
        assert power_of_2(nqubit) >= self.world_size
        assert 0 <= self.rank < self.world_size
        self.nqubit = nqubit
        self.batch = batch          # None: 1-D shard as in the reference; B: (B, 2^L) shards, one per sample
        self.log_num_nodes = log_base2(self.world_size)
        self.log_num_amps_per_node = nqubit - self.log_num_nodes
        self.num_amps_per_node = power_of_2(self.log_num_amps_per_node)
        self._shape = (self.num_amps_per_node,) if batch is None else (batch, self.num_amps_per_node)
        empty = torch.zeros(0, dtype=dtype, device=device)
        self.register_buffer('amps', empty)
        self.register_buffer('buffer', empty.clone())
        if (batch or 1) * self.num_amps_per_node <= self.LAZY_AMPS or device is not None:
            self.reset()

    def __getattr__(self, name: str):
        # ``amps`` read from outside the sharded kernels is always in the reference's layout: a circuit leaves the
        # qubits wherever its last remap put them (distributed.dist_run(keep_layout=True): the exchange back is only
        # paid by who looks at the amplitudes -- expectation values of Pauli strings do not), and the canonical order
        # is restored here, on first access.  The routines of distributed.py work on the raw shard (`_raw`).
        if name in ('amps', 'buffer'):
            # a big shard is not built by the constructor (LAZY_AMPS): whoever touches it first builds |0...0> on the
            # device the state lives on by then -- gate routines, `cir(state=s)`, load_state_dict all see a usable shard
            d = self.__dict__
            t = super().__getattr__(name)
            if t.numel() == 0 and not d.get('_building'):       # (the LAZY marker; a shard of any other shape is the
                d['_building'] = True                           #  caller's business and is never replaced)
                try:
                    self.reset()
                finally:
                    d['_building'] = False
        if name == 'amps':
            d = self.__dict__
            if (d.get('_lazy_zero') or d.get('_zeros_owed')) and not d.get('_raw', 0):
                from .distributed import _materialize_zeros      # (somebody outside the sharded routines looks: real zeros)

                _materialize_zeros(self)
            ph = d.get('_phys')
            if ph is not None and not d.get('_raw', 0) and any(p != q for q, p in enumerate(ph)):
                from .distributed import canonicalize

                canonicalize(self)
        return super().__getattr__(name)

    def __deepcopy__(self, memo):
        # a copy is a state of its own: exchanges in flight on the group streams are joined first (the copy kernels run
        # on the current stream), and the bookkeeping of whoever holds the original open (`_raw` nesting, cached
        # expectation values, stream hand-offs) is not inherited
        from copy import deepcopy

        from .distributed import _materialize_zeros, _settle

        _settle(self)
        _materialize_zeros(self)       # (a copy is read by whoever made it: logical zeros become real ones first)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, value in self.__dict__.items():
            if key in ('_raw', '_inflight', '_inflight_keep', '_expz', '_building', '_spare', '_arrivals'):      # (the third buffer of a
                #  sliced exchange is scratch: a copy gets its own when it needs one)
                continue
            new.__dict__[key] = deepcopy(value, memo)
        return new

    def state_dict(self, *args, **kwargs):
        _ = self.amps          # (a saved shard is in the reference's qubit order)
        return super().state_dict(*args, **kwargs)

    #: amplitudes at the start of every row that a LAZY reset really zeroes (a complex64 tile; more than any first pass
    #: reads of |0..0> -- one 16-byte piece)
    LAZY_HEAD = 1 << 12
    #: tests: a lazy reset fills what it leaves un-zeroed with NaN, so that a read of it cannot go unnoticed
    POISON_LAZY = False

    def reset(self, lazy: bool = False) -> None:
        """|0..0>: rank 0 holds a single 1, everybody else zeros (reference: state.py:358-383, circuit.py:1655-1675).

        ``lazy`` (round 6; only ``DistributedQubitCircuit.forward``, which hands the state to ``dist_run(fresh_zero=True)``
        right away): the shard is NOT cleared -- a 16-GiB memset per step -- beyond its first `LAZY_HEAD` amplitudes; the
        flag ``_lazy_zero`` tells `distributed.dist_apply_prims` that the rest is logically zero but holds whatever the
        last step left.  The passes behind |0..0> neither read nor keep it (known-zero masks); whoever cannot vouch for
        that clears it first (`distributed._materialize_zeros`)."""
        self.__dict__.pop('_phys', None)   # canonical qubit order (first: ``amps`` below must not trigger an exchange)
        self.__dict__.pop('_expz', None)   # (expectation values cached by a circuit's last pass)
        self.__dict__.pop('_lazy_zero', None)
        self.__dict__.pop('_zeros_owed', None)
        cur = self._buffers['amps']       # (read past the lazy-build hook of __getattr__: it calls us)
        if tuple(cur.shape) != tuple(self._shape):
            self.amps = torch.zeros(self._shape, dtype=cur.dtype, device=cur.device)
            self.buffer = torch.zeros_like(self._buffers['amps'])
        elif lazy and self.num_amps_per_node > self.LAZY_HEAD:
            rows = cur.view(-1, self.num_amps_per_node)
            if self.POISON_LAZY:
                rows[:, self.LAZY_HEAD:] = float('nan')
            rows[:, :self.LAZY_HEAD] = 0
            self.__dict__['_lazy_zero'] = True
        else:
            cur.zero_()             # (the receive buffer is scratch: every use writes all of what it then reads)
        if self.rank == 0:
            self._buffers['amps'][..., 0] = 1.0
