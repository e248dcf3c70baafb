"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/dq_hip.h
declares (no compute calls here)."""

import ctypes
import os
import re

import pytest

from deepquantum_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'dq_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dq_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in dq_hip.h but not exported'
    assert sorted(_lib.exported_symbols()) == declared, 'python binding table and header disagree'
    assert lib.dq_abi_version() == _lib.ABI_VERSION
    assert lib.dq_last_error() is not None


def test_struct_layout_matches_header():
    """ctypes structs against what the compiled library reports (dq_struct_layout), plus the fields the kernels decode
    by offset."""
    lib = _lib.load()
    got = (ctypes.c_int * 16)()
    cnt = lib.dq_struct_layout(got, 16)
    p = _lib.DqFusedPass
    mine = [ctypes.sizeof(_lib.DqFusedGate), ctypes.sizeof(_lib.DqFusedRound), ctypes.sizeof(p), p.rounds.offset, p.gates.offset,
            p.load_slot_off.offset, p.store_high_pos.offset, p.store_tb.offset, p.slots.offset]
    assert cnt == len(mine) and list(got[:cnt]) == mine
    assert ctypes.sizeof(_lib.DqFusedGate) == 32
    assert _lib.DqFusedGate.fast.offset == 12 and _lib.DqFusedGate.mat_advance.offset == 24
    assert _lib.DqFusedGate.out_cmask.offset == 16 and _lib.DqFusedGate.mat.offset == 8


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers are rejected on the host before any HIP call
    rc = lib.dq_apply_gate_c64(None, None, None, 0, 3, _lib.int_array([0]), 1, _lib.int_array([]), 0, 1, None)
    assert rc == -1 and b'null' in lib.dq_last_error()
    rc = lib.dq_fused_geometry(0, 9, None, None, None)
    assert rc == -1
    m, s, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dq_fused_geometry(0, 0, ctypes.byref(m), ctypes.byref(s), ctypes.byref(t)) == 0
    assert (m.value, s.value, t.value) == (12, 6, 64)      # complex64 default: the wave tile
    assert lib.dq_reduce_ws_bytes(4) == 4 * 1024 * 16


def test_missing_library_is_a_loud_error(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdqhip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()
