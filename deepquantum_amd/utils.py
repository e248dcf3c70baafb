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
