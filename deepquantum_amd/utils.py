"""dtype / device plumbing shared by the module classes."""

from __future__ import annotations

from typing import Any, Callable

import torch

from . import _functorch

# real dtype requested through nn.Module.to(...) -> complex dtype the buffers must take
dtype_map = {torch.float: torch.cfloat, torch.double: torch.cdouble}


def complex_apply(fn: Callable, tensors: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """``module.to(torch.double)`` must turn complex64 buffers into complex128 instead of dropping
    their imaginary part.  Probe ``fn`` with a real tensor of the matching precision to learn the
    requested real dtype / device, then convert to the complex counterpart (same contract as the
    reference's utils.apply_complex_fix, utils.py:45-50)."""
    if not tensors:
        return {}
    bulk = getattr(fn, 'bulk', None)
    if bulk is not None and all(id(v) in bulk for v in tensors.values()):
        return {k: bulk[id(v)] for k, v in tensors.items()}           # (moved with the circuit's other small buffers)
    first = next(iter(tensors.values()))
    probe = fn(torch.empty(0, dtype=first.real.dtype, device=first.device))
    target = dtype_map.get(probe.dtype, probe.dtype)
    return {k: v.to(probe.device, target) for k, v in tensors.items()}


class BulkMove:
    """``module.to(device)`` for a module tree with hundreds of tiny buffers -- a circuit: one angle per gate, a 4x4 matrix per
    CNOT -- as ONE host-to-device copy per dtype instead of one per buffer (the reference's gradient benchmark builds a new
    circuit per call: 376 copies of a few bytes were 5 of its 30 ms here).  Wraps the ``fn`` that ``nn.Module._apply``
    hands down: the small CPU buffers of the whole tree are concatenated per dtype, converted once (complex buffers by the
    rule of `complex_apply`), and every module then receives its slice -- a view of the moved block -- when it asks for
    its own buffer.  Only when the call MOVES buffers to another device; only buffers (never parameters), of at most 64
    elements; anything else goes through ``fn`` as before."""

    LIMIT = 64

    def __init__(self, fn: Callable, root: torch.nn.Module) -> None:
        self.fn = fn
        self.bulk: dict[int, torch.Tensor] = {}
        groups: dict = {}
        for mod in root.modules():
            for t in mod._buffers.values() if type(mod._buffers) is dict else ():
                if (t is not None and t.device.type == 'cpu' and 0 < t.numel() <= self.LIMIT and id(t) not in self.bulk
                        and not _functorch.is_wrapped_tensor(t)):
                    groups.setdefault(t.dtype, []).append(t)
                    self.bulk[id(t)] = t
        self.bulk.clear()
        for dtype, ts in groups.items():
            if len(ts) < 8:
                continue
            probe = fn(torch.empty(0, dtype=torch.empty(0, dtype=dtype).real.dtype if dtype.is_complex else dtype))
            if probe.device.type == 'cpu':
                self.bulk.clear()
                return                                   # (not a move to a device: nothing to batch)
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            moved = flat.to(probe.device, dtype_map.get(probe.dtype, probe.dtype)) if dtype.is_complex else fn(flat)
            off = 0
            for t in ts:
                self.bulk[id(t)] = moved[off:off + t.numel()].reshape(t.shape)
                off += t.numel()
        self._keep = [t for ts in groups.values() for t in ts]       # (ids stay theirs while the call runs)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        out = self.bulk.get(id(t))
        return self.fn(t) if out is None else out


def to_list(x: Any) -> list:
    return list(x) if isinstance(x, (list, tuple)) else [x]


class CapturedGraph:
    """A whole circuit evaluation -- or a whole training step: forward, expectation, ``backward()`` -- captured
    into ONE HIP graph, so that every further run is a single host call instead of hundreds of kernel launches
    (below ~20 qubits a circuit is launch-bound: ~10 us of Python and launch overhead per gate, microseconds of
    GPU work).  All kernels of this library enqueue on ``torch.cuda.current_stream()`` and take their scratch
    memory from PyTorch's allocator, which is what makes them capturable.

    ``fn`` takes no arguments: it must read its inputs from tensors that stay alive (update them in place with
    ``copy_`` before :meth:`replay`) and its return value is kept as the static output.  For a training step,
    call ``module.zero_grad(set_to_none=True)`` before constructing the object so that the ``.grad`` tensors are
    allocated inside the graph, and read / apply them after every replay.

        data = torch.zeros(64, cir.ndata, device='cuda')
        graph = dq.CapturedGraph(lambda: (cir(data), cir.expectation())[1])
        data.copy_(batch); ev = graph.replay()
    """

    def __init__(self, fn, warmup: int = 3) -> None:
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):      # plans, lazy handles and allocator pools settle outside the capture
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # cache-owned device tensors the captured launches read (backend.pin_if_capturing): they live as long as this object
        from . import backend

        self._pins: dict = {}
        backend._PIN_SINK.append(self._pins)
        try:
            with torch.cuda.graph(self.graph):
                self.output = fn()
        finally:
            backend._PIN_SINK.pop()

    def replay(self):
        self.graph.replay()
        return self.output
