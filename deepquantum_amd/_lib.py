"""ctypes binding of libdqhip.so (the C ABI declared in include/dq_hip.h).

The library is the product: there is no CPU fallback.  If it is missing, every compute entry point
raises ``RuntimeError`` telling the user how to build it (``python -c "import __graft_entry__ as g;
g.build()"`` or ``deepquantum_amd/csrc/build.sh``).
"""

from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DQHIP_LIBRARY: tools/ablate.sh points it at experimental builds of the same ABI)
LIB_PATH = os.environ.get('DQHIP_LIBRARY') or os.path.join(_HERE, 'libdqhip.so')

DQ_OK = 0
ABI_VERSION = 25

# enum DqFusedKind / DqBitLoc (include/dq_hip.h)
FG_GEN1, FG_X1, FG_DIAG1, FG_GEN2, FG_DIAG2, FG_RESERVED5, FG_GRAD, FG_EXPZ = range(8)
LOC_REG, LOC_THR, LOC_OUT = range(3)

FUSED_MAX_HIGH = 12
FUSED_MAX_LOW = 8
FUSED_MAX_ROUNDS = 24
FUSED_MAX_GATES = 128
FUSED_MAX_SLOTS = 6
FUSED_MAX_TBITS = 9
FUSED_MAX_BLK = 24
ROUND_TRANSPOSE = 0x01
ROUND_TRANSPOSE_AFTER = 0x02
FAST_NONE = 0xFFFFFFFF
MAT_PAD = 16


class DqFusedGate(C.Structure):
    _fields_ = [
        ('kind', C.c_uint8),
        ('q', C.c_uint8),
        ('q2', C.c_uint8),
        ('loc', C.c_uint8),
        ('loc2', C.c_uint8),
        ('reg_cmask', C.c_uint8),
        ('thr_cmask', C.c_uint16),
        ('mat', C.c_uint32),
        ('fast', C.c_uint32),
        ('out_cmask', C.c_uint64),
        ('mat_advance', C.c_uint32),
        ('reserved', C.c_uint32),
    ]


class DqFusedRound(C.Structure):
    _fields_ = [
        ('rb', C.c_uint8 * FUSED_MAX_SLOTS),
        ('tb', C.c_uint8 * FUSED_MAX_TBITS),
        ('flags', C.c_uint8),
        ('gate_begin', C.c_uint8),
        ('gate_end', C.c_uint8),
    ]


class DqFusedPass(C.Structure):
    _fields_ = [
        ('m', C.c_uint8),
        ('L', C.c_uint8),
        ('h', C.c_uint8),
        ('nrounds', C.c_uint8),
        ('high_pos', C.c_uint8 * FUSED_MAX_HIGH),
        ('high_sorted', C.c_uint8 * FUSED_MAX_HIGH),
        ('load_rb', C.c_uint8 * FUSED_MAX_SLOTS),
        ('store_rb', C.c_uint8 * FUSED_MAX_SLOTS),
        ('rounds', DqFusedRound * FUSED_MAX_ROUNDS),
        ('mat_base', C.c_uint32),
        ('gates', DqFusedGate * FUSED_MAX_GATES),
        ('load_slot_off', C.c_uint64 * FUSED_MAX_SLOTS),
        ('store_slot_off', C.c_uint64 * FUSED_MAX_SLOTS),
        ('store_high_pos', C.c_uint8 * FUSED_MAX_HIGH),
        ('store_blk_pos', C.c_uint8 * FUSED_MAX_BLK),
        ('store_low_pos', C.c_uint8 * FUSED_MAX_LOW),
        ('store_tb', C.c_uint8 * FUSED_MAX_TBITS),
        ('slots', C.c_uint8),
    ]


_vp, _i, _i64, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); `{s}` expands to c64 / c128
_SIGNATURES = {
    'dq_abi_version': (_i, []),
    'dq_last_error': (C.c_char_p, []),
    'dq_struct_layout': (_i, [_ip, _i]),
    'dq_device_info': (_i, [_ip, C.POINTER(_i64), C.POINTER(_i64)]),
    'dq_fused_geometry': (_i, [_i, _i, _ip, _ip, _ip]),
    'dq_wave_descriptor': (_i, [C.POINTER(DqFusedPass), _i, _u64, _vp, _i]),
    'dq_dag_create': (_vp, [_i, _vp, _vp, _vp, _vp]),
    'dq_dag_destroy': (None, [_vp]),
    'dq_dag_closure': (_i, [_vp, _u64, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    'dq_dag_rank': (_i, [_vp, _u64, _i, _vp, _vp, _i, _vp, _i, _vp]),
    'dq_dag_grow_step': (_i, [_vp, _u64, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'dq_set_dense_path': (_i, [_i]),
    'dq_reduce_ws_bytes': (_i64, [_i64]),
    'dq_apply_gate_{s}': (_i, [_vp, _vp, _vp, _i64, _i, _ip, _i, _ip, _i, _i64, _vp]),
    'dq_apply_fused_{s}': (_i, [_vp, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _vp]),
    'dq_apply_fused_bcast_{s}': (_i, [_vp, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _vp]),
    'dq_apply_fused_zext_{s}': (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _u64, _vp]),
    'dq_apply_fused_slice_{s}': (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _u64, _u64, _u64, _vp]),
    'dq_defer_rx_c64': (_i, [_vp, _i64, _vp, _i64, _i64, _vp]),
    'dq_apply_fused_grad_c64': (_i, [_vp, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _vp, _i64, _vp]),
    'dq_apply_fused_grad_c128': (_i, [_vp, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _vp, _i64, _vp]),
    'dq_wave_records': (_i64, [C.POINTER(DqFusedPass), _i, _vp, _i64]),
    'dq_apply_fused_grad_ext_{s}': (_i, [_vp, _vp, _vp, _i64, _i, _i64, C.POINTER(DqFusedPass), _vp, _i64, _vp, _i64, _vp]),
    'dq_expect_pauli_{s}': (_i, [_vp, _u64, _u64, _i, _i64, _vp, _vp, _vp]),
    'dq_inner_{s}': (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'dq_probs_{s}': (_i, [_vp, _vp, _i64, _vp]),
    'dq_marginal_{s}': (_i, [_vp, _i, _ip, _i, _i64, _vp, _vp]),
    'dq_expect_zmulti_{s}': (_i, [_vp, C.POINTER(C.c_uint64), _i, _i, _i64, _vp, _i, _vp]),
    'dq_scale_zsigns_{s}': (_i, [_vp, _vp, C.POINTER(C.c_uint64), _i, _vp, _i, _i64, _vp]),
    'dq_gate_grad_{s}': (_i, [_vp, _vp, _i, _ip, _i, _ip, _i, _i64, _vp, _vp]),
    'dq_gate_grad_multi_{s}': (_i, [_vp, _vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i64, _vp, _i, _vp]),
    'dq_pack_{s}': (_i, [_vp, _vp, _i, _u64, _u64, _i64, _vp]),
    'dq_unpack_axpby_{s}': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _u64, _u64, _i64, _vp]),
    'dq_permute_bits_{s}': (_i, [_vp, _vp, _i, _ip, _i64, _vp]),
    'dq_interleave_{s}': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'dq_deinterleave_{s}': (_i, [_vp, _vp, _i64, _i, _vp]),
}


def exported_symbols() -> list[str]:
    """Every symbol include/dq_hip.h declares."""
    names = []
    for name in _SIGNATURES:
        if '{s}' in name:
            names += [name.format(s='c64'), name.format(s='c128')]
        else:
            names.append(name)
    return names


_lock = threading.Lock()
_lib = None


def load() -> C.CDLL:
    """Load libdqhip.so (once) and type its entry points.  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'deepquantum_amd: HIP library not found at {LIB_PATH}. Build it with '
                f'`bash {os.path.join(_HERE, "csrc", "build.sh")}` (hipcc, gfx950). '
                'There is no CPU fallback for the statevector kernels.'
            )
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as exc:  # missing libamdhip64 etc.
            raise RuntimeError(f'deepquantum_amd: cannot load {LIB_PATH}: {exc}') from exc
        for name, (res, args) in _SIGNATURES.items():
            for full in ([name.format(s='c64'), name.format(s='c128')] if '{s}' in name else [name]):
                fn = getattr(lib, full)
                fn.restype = res
                fn.argtypes = args
        got = lib.dq_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f'deepquantum_amd: libdqhip ABI {got} != expected {ABI_VERSION}; rebuild')
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != DQ_OK:
        msg = load().dq_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed (status {rc}): {msg}')


def int_array(values) -> C.Array:
    values = list(values)
    return (C.c_int * max(len(values), 1))(*values)
