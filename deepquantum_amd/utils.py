"""dtype / device plumbing shared by the module classes."""

from __future__ import annotations

from typing import Any, Callable

import torch

# real dtype requested through nn.Module.to(...) -> complex dtype the buffers must take
dtype_map = {torch.float: torch.cfloat, torch.double: torch.cdouble}


def complex_apply(fn: Callable, tensors: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """``module.to(torch.double)`` must turn complex64 buffers into complex128 instead of dropping
    their imaginary part.  Probe ``fn`` with a real tensor of the matching precision to learn the
    requested real dtype / device, then convert to the complex counterpart (same contract as the
    reference's utils.apply_complex_fix, utils.py:45-50)."""
    if not tensors:
        return {}
    first = next(iter(tensors.values()))
    probe = fn(torch.empty(0, dtype=first.real.dtype, device=first.device))
    target = dtype_map.get(probe.dtype, probe.dtype)
    return {k: v.to(probe.device, target) for k, v in tensors.items()}


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
        with torch.cuda.graph(self.graph):
            self.output = fn()

    def replay(self):
        self.graph.replay()
        return self.output
