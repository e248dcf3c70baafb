"""CPU emulator of the wave-tile kernel (csrc/dq_wave.hip) -- TEST INFRASTRUCTURE ONLY.

Executes the kernel-side descriptor the library derives from a ``DqFusedPass`` (``dq_wave_descriptor``, no GPU needed)
the way the kernel's assembly does: 64 lanes x 64 register-resident amplitudes per tile, loads and stores through the
per-lane / per-slot byte offsets, gates on physical register slots, layout changes through a simulated wave-private LDS
buffer.  Everything that moves data between registers -- the trips' ``ds_write_b64`` / ``ds_read_b64`` sequences with
their immediates, X gates, slot swaps, the masked pairs of register-controlled gates -- is taken instruction by
instruction from the generator that writes the kernel (tools/gen_wave_asm.py), so a disagreement between the
translator (C++) and the generated code shows up here, before a kernel runs.  The arithmetic of the 2x2 bodies is
restated (complex multiply-adds), not interpreted.
"""

from __future__ import annotations

import ctypes as C
import importlib.util
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepquantum_amd import _lib  # noqa: E402

_gen = None


def gen():
    """tools/gen_wave_asm.py as a module (its output goes to a scratch file)."""
    global _gen
    if _gen is None:
        os.environ['DQ_ASM_OUT'] = os.path.join('/tmp', f'dq_wave_asm_{os.getpid()}.inc')
        spec = importlib.util.spec_from_file_location('gen_wave_asm', os.path.join(ROOT, 'tools', 'gen_wave_asm.py'))
        _gen = importlib.util.module_from_spec(spec)
        stdout = sys.stdout
        sys.stdout = open(os.devnull, 'w')
        try:
            spec.loader.exec_module(_gen)
        finally:
            sys.stdout = stdout
            os.environ.pop('DQ_ASM_OUT', None)
    return _gen


class WaveKernPass(C.Structure):
    _fields_ = [('load_off', C.c_uint64 * 5), ('store_off', C.c_uint64 * 5), ('load_lane_shift', C.c_uint32 * 6),
                ('store_lane_shift', C.c_uint32 * 6), ('tb_contrib', C.c_uint32 * 6), ('nrec_bytes', C.c_uint32),
                ('mat_base_bytes', C.c_uint32), ('read_blk_pos', C.c_uint8 * 24), ('store_blk_pos', C.c_uint8 * 24),
                ('rec', (C.c_uint32 * 8) * 112)]


def descriptor(desc, n) -> WaveKernPass:
    lib = _lib.load()
    kp = WaveKernPass()
    rc = lib.dq_wave_descriptor(C.byref(desc), n, C.byref(kp), C.sizeof(kp))
    if rc < 0:
        raise RuntimeError(lib.dq_last_error().decode())
    assert rc == WaveKernPass.rec.offset + kp.nrec_bytes
    return kp


_AMP = re.compile(r'v\[(\d+):(\d+)\]')


def _amp_index(tok: str) -> int:
    lo = int(_AMP.fullmatch(tok).group(1))
    g = gen()
    assert lo >= g.AMP0 and (lo - g.AMP0) % 2 == 0
    return (lo - g.AMP0) // 2


def _run_moves(lines, a, active):
    """Interpret v_mov_b64 lines (X gates, slot swaps) on the register file a[lane, reg]; `active` = exec mask."""
    tmp = {}
    for ln in lines:
        op, rest = ln.split(' ', 1)
        assert op == 'v_mov_b64', ln
        dst, src = [t.strip() for t in rest.split(',')]
        val = tmp[src] if src in tmp else a[:, _amp_index(src)].copy()
        if _AMP.fullmatch(dst) and int(_AMP.fullmatch(dst).group(1)) >= gen().AMP0:
            j = _amp_index(dst)
            a[active, j] = val[active]
        else:
            tmp[dst] = val


def _apply2(a, lo, hi, m, active):
    x0, x1 = a[:, lo].copy(), a[:, hi].copy()
    a[active, lo] = (m[0] * x0 + m[1] * x1)[active]
    a[active, hi] = (m[2] * x0 + m[3] * x1)[active]


def run_pass(desc, n, state, mats, mat_batch_stride):
    """state (B, 2^n) complex64 numpy -> the state after the pass, as the wave-tile kernel computes it."""
    g = gen()
    kp = descriptor(desc, n)
    nrec = kp.nrec_bytes // 32
    rec = [list(kp.rec[i]) for i in range(nrec)]
    out = np.empty_like(state)
    lanes = np.arange(64)
    lb = [(lanes >> b) & 1 for b in range(6)]
    lld = sum(lb[b].astype(np.int64) << kp.load_lane_shift[b] for b in range(6))
    lst = sum(lb[b].astype(np.int64) << kp.store_lane_shift[b] for b in range(6))
    ntiles = 1 << (n - 12)
    flat_m = np.asarray(mats).reshape(-1)
    trip_of = {g.ID_TRIP + i: (bin(mk).count('1'), mk) for i, mk in enumerate(g.TRIP_MASKS)}
    trip_of[g.ID_TRIP0] = (0, 0)
    swap_of = {g.ID_SWAP + i: pr for i, pr in enumerate(g.SWAP_PAIRS)}
    for b in range(state.shape[0]):
        flat_in = state[b]
        mb = flat_m[b * mat_batch_stride:] if mat_batch_stride else flat_m
        for tile in range(ntiles):
            tg = sum(((tile >> j) & 1) << kp.read_blk_pos[j] for j in range(24))
            tw = sum(((tile >> j) & 1) << kp.store_blk_pos[j] for j in range(24))
            a = np.zeros((64, 64), dtype=state.dtype)
            for piece in range(32):
                off = sum(kp.load_off[s] for s in range(5) if (piece >> s) & 1)
                addr = (tg * 8 + off + lld) // 8
                a[:, 2 * piece] = flat_in[addr]
                a[:, 2 * piece + 1] = flat_in[addr + 1]
            tb = sum(lb[b_] * kp.tb_contrib[b_] for b_ in range(6))
            scale = np.complex128(1.0)
            moff = kp.mat_base_bytes // 8
            i = 0
            while i < nrec:
                w = rec[i]
                i += 1
                hid = w[0]
                m = mb[moff:moff + 4]
                moff += w[4]
                oc = w[2] | (w[3] << 32)
                tile_ok = (tg & oc) == oc
                active = (tb & w[1]) == w[1]
                if hid in trip_of:
                    k, mask = trip_of[hid]
                    wb2 = rec[i]
                    i += 1
                    tbw = [w[1], w[2], w[3], w[5], w[6], w[7]]
                    tb = sum(lb[b_] * tbw[b_] for b_ in range(6))
                    packed = sum(lb[b_].astype(np.int64) * wb2[b_] for b_ in range(6))
                    wbase, rbase = packed & 0xFFFF, packed >> 16
                    lds = {}
                    for ln in g.trip(k, mask):
                        if ln.startswith('ds_write_b64'):
                            mm = re.fullmatch(r'ds_write_b64 v\d+, (v\[\d+:\d+\])(?: offset:(\d+))?', ln)
                            j, imm = _amp_index(mm.group(1)), int(mm.group(2) or 0)
                            for lane in range(64):
                                ad = int(wbase[lane]) + imm
                                assert ad % 8 == 0 and ad < 8448, 'LDS write outside the wave\'s region'
                                lds[ad] = a[lane, j]
                        elif ln.startswith('ds_read_b64'):
                            mm = re.fullmatch(r'ds_read_b64 (v\[\d+:\d+\]), v\d+(?: offset:(\d+))?', ln)
                            j, imm = _amp_index(mm.group(1)), int(mm.group(2) or 0)
                            for lane in range(64):
                                a[lane, j] = lds[int(rbase[lane]) + imm]      # KeyError: reads what nobody wrote
                    continue
                if hid in swap_of:
                    _run_moves(g.slotswap(*swap_of[hid]), a, np.ones(64, bool))
                    continue
                if not tile_ok:
                    continue
                if hid >= g.ID_DIAG1:
                    # diagonal gate (tools/gen_wave_asm.py, diag_code): four phases, candidates by the indices in w5,
                    # two per-lane selectors, PH0 / PH1 by the bit of one register slot, an optional register mask
                    four = hid >= g.ID_DIAG2
                    variant = (hid - g.ID_DIAG1) % 14
                    if four:
                        blk = mb[moff - 16:moff]
                        d = [blk[0], blk[5], blk[10], blk[15]]
                    else:
                        d = [m[0], m[3], m[0], m[3]]
                    w5 = w[5]

                    def sel(byte):
                        kind, pos = (byte >> 6) & 3, byte & 63
                        if kind == 1:
                            return ((tb >> pos) & 1).astype(bool)
                        if kind == 2:
                            return np.full(64, bool((tg >> pos) & 1))
                        return np.zeros(64, bool)

                    sa, sb = sel(w5 & 0xFF), sel((w5 >> 8) & 0xFF)
                    c0 = [d[(w5 >> (16 + 2 * k)) & 3] for k in range(4)]
                    c1 = {k: d[(w5 >> (24 + 2 * k)) & 3] for k in (0, 2)}
                    ph0 = np.where(sa, np.where(sb, c0[3], c0[2]), np.where(sb, c0[1], c0[0]))
                    ph1 = np.where(sa, c1[2], c1[0])
                    masked, v = variant >= 7, variant % 7
                    rmask = w[6] | (w[7] << 32)
                    for j in range(64):
                        if masked and not (rmask >> j) & 1:
                            continue
                        ph = ph0 if v == 0 or not (j >> (v - 1)) & 1 else ph1
                        a[active, j] = (a[:, j] * ph.astype(a.dtype))[active]
                    continue
                if g.ID_GEN_U <= hid < g.ID_GEN_C:
                    mode, q = divmod(hid - g.ID_GEN_U, 6)
                    everyone = np.ones(64, bool)
                    if mode == 2:       # the deferred Rx block { f, (0, t), -, (flag, -) }
                        f, it, flag = m[0], m[1], m[3].real
                        assert it.real == 0 and flag in (0.0, 1.0)
                        mat = np.array([1, it, it, 1]) if flag == 0 else np.array([it, 1, 1, it])
                        scale = scale * (f.real if flag == 0 else 1j * f.imag)
                    elif mode == 3:     # Hadamard-like: sums and differences, the factor deferred
                        mat = np.array([1, 1, 1, -1])
                        scale = scale * m[0].real
                    else:
                        mat = m
                    for lo, hi in g.pairs(q):
                        _apply2(a, lo, hi, mat.astype(a.dtype), everyone)
                elif g.ID_GEN_C <= hid < g.ID_GEN_R:
                    for lo, hi in g.pairs(hid - g.ID_GEN_C):
                        _apply2(a, lo, hi, m, active)
                elif g.ID_GEN_R <= hid < g.ID_X_U:
                    for pi, (lo, hi) in enumerate(g.pairs(hid - g.ID_GEN_R)):
                        if (w[5] >> pi) & 1:
                            _apply2(a, lo, hi, m, active)
                elif g.ID_X_U <= hid < g.ID_X_C:
                    _run_moves(g.xlines(hid - g.ID_X_U), a, np.ones(64, bool))
                elif g.ID_X_C <= hid < g.ID_X_R:
                    _run_moves(g.xlines(hid - g.ID_X_C), a, active)
                elif g.ID_X_R <= hid < g.ID_X_R1:
                    q = hid - g.ID_X_R
                    for pi, (lo, hi) in enumerate(g.pairs(q)):
                        if (w[5] >> pi) & 1:
                            x0 = a[:, lo].copy()
                            a[active, lo] = a[active, hi]
                            a[active, hi] = x0[active]
                elif g.ID_X_R1 <= hid < g.ID_TRIP0:
                    q, cc = divmod(hid - g.ID_X_R1, 5)
                    c = cc if cc < q else cc + 1
                    _run_moves(g.xlines(q, 1 << c), a, active)
                else:
                    raise AssertionError(f'unknown handler id {hid}')
            a = (a * scale).astype(state.dtype)
            base = b
            for piece in range(32):
                off = sum(kp.store_off[s] for s in range(5) if (piece >> s) & 1)
                addr = (tw * 8 + off + lst) // 8
                out[base][addr] = a[:, 2 * piece]
                out[base][addr + 1] = a[:, 2 * piece + 1]
    return out
