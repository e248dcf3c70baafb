"""The ONE place that looks into functorch's private interface.

``torch.vmap`` over a circuit is the reference's own batching (circuit.py:232-240), and the fused nodes, the |0..0> claim
of a state and the encoders have to know whether a tensor is a functorch wrapper (no data pointer), a BatchedTensor, and
which transforms a call runs under.  PyTorch offers no public probe for any of that; the private ones used here
(``torch._C._functorch.*``, ``torch._functorch.pyfunctorch.retrieve_all_functorch_interpreters``) were checked on the
releases in `CHECKED_ON`.  On a release where one of them is gone every probe falls back to a slower public test or to
"unknown" -- callers then take their conservative route (per-gate nodes) -- and says so ONCE in a warning that names the
probe, instead of silently turning the fused batching of SURVEY row a5 into the per-gate route."""

from __future__ import annotations

import warnings

import torch

#: PyTorch releases the probes were checked against (tests/test_api_cpu.py::test_functorch_probes_are_guarded)
CHECKED_ON = ('2.10',)

_C = getattr(torch._C, '_functorch', None)
_WARNED: set = set()


def _missing(name: str) -> None:
    if name not in _WARNED:
        _WARNED.add(name)
        warnings.warn(f'deepquantum_amd: torch {torch.__version__} has no private probe {name!r} (checked on {CHECKED_ON}); '
                      f'falling back to a conservative test -- circuits under torch.func transforms may take the per-gate '
                      f'route instead of fused passes', RuntimeWarning, stacklevel=3)


def _probe(name: str):
    fn = getattr(_C, name, None) if _C is not None else None
    if fn is None:
        _missing('torch._C._functorch.' + name)
    return fn


def is_wrapped_tensor(t: torch.Tensor) -> bool:
    """A functorch wrapper of any transform (vmap, grad, jvp ...): no data pointer."""
    fn = _probe('is_functorch_wrapped_tensor')
    if fn is not None:
        return fn(t)
    try:                                   # public fallback: a wrapper refuses to show its memory
        t.data_ptr()
        return False
    except RuntimeError:
        return True


def is_batched(t: torch.Tensor) -> bool:
    """A BatchedTensor of ``torch.vmap``."""
    fn = _probe('is_batchedtensor')
    if fn is not None:
        return fn(t)
    return is_wrapped_tensor(t)            # conservative: any wrapper counts


def is_legacy_batched(t: torch.Tensor) -> bool:
    """A BatchedTensor of the legacy ``torch._vmap_internals`` (what ``autograd.functional.jacobian(vectorize=True)`` uses)."""
    fn = _probe('is_legacy_batchedtensor')
    return bool(fn(t)) if fn is not None else False


def transform_stack() -> list[str] | None:
    """The ``torch.func`` transforms this call runs under, outermost first, as 'Vmap' / 'Grad' / 'Jvp' / ... -- or None
    when this PyTorch does not let us look (callers treat None as "unknown": the conservative route)."""
    try:
        from torch._functorch.pyfunctorch import retrieve_all_functorch_interpreters

        return [str(it.key()).rsplit('.', 1)[-1] for it in retrieve_all_functorch_interpreters()]
    except Exception:           # noqa: BLE001  (ImportError, AttributeError, a changed signature ...)
        _missing('torch._functorch.pyfunctorch.retrieve_all_functorch_interpreters')
        return None


def no_transforms() -> bool:
    """True only when the interpreter stack could be read AND is empty ("unknown" is not "none")."""
    stack = transform_stack()
    return stack is not None and not stack
