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

_gens = {}


def gen(c128: bool = False):
    """tools/gen_wave_asm.py (complex64) / gen_wave_asm64.py (complex128) as a module (output to a scratch file)."""
    if c128 not in _gens:
        name = 'gen_wave_asm64' if c128 else 'gen_wave_asm'
        os.environ['DQ_ASM_OUT'] = os.path.join('/tmp', f'dq_{name}_{os.getpid()}.inc')
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        stdout = sys.stdout
        sys.stdout = open(os.devnull, 'w')
        try:
            spec.loader.exec_module(mod)
        finally:
            sys.stdout = stdout
            os.environ.pop('DQ_ASM_OUT', None)
        _gens[c128] = mod
    return _gens[c128]


class WaveKernPass(C.Structure):
    _fields_ = [('load_off', C.c_uint64 * 5), ('store_off', C.c_uint64 * 5), ('load_lane_shift', C.c_uint32 * 6),
                ('store_lane_shift', C.c_uint32 * 6), ('tb_contrib', C.c_uint32 * 6), ('nrec_bytes', C.c_uint32),
                ('mat_base_bytes', C.c_uint32), ('read_blk_pos', C.c_uint8 * 24), ('store_blk_pos', C.c_uint8 * 24),
                ('zext', C.c_uint32), ('fix_read', C.c_uint32 * 2), ('fix_write', C.c_uint32 * 2), ('reserved', C.c_uint32 * 3),
                ('rec', (C.c_uint32 * 8) * 256)]     # (WAVE_EXT_REC: the test hook's cap)


def descriptor(desc, n, known_zero: int = 0) -> WaveKernPass:
    lib = _lib.load()
    kp = WaveKernPass()
    rc = lib.dq_wave_descriptor(C.byref(desc), n, known_zero, C.byref(kp), C.sizeof(kp))
    if rc < 0:
        raise RuntimeError(lib.dq_last_error().decode())
    assert rc == WaveKernPass.rec.offset + kp.nrec_bytes
    return kp


_AMP = re.compile(r'v\[(\d+):(\d+)\]')


class _Regs:
    """The amplitude registers of a wave as the generated code names them: complex64 -- amplitude j = the 64-bit pair
    v[40 + 2j : 41 + 2j]; complex128 -- re = v[40 + 4j : 41 + 4j], im = v[42 + 4j : 43 + 4j], the whole amplitude the
    quad.  `a` is the (64 lanes, NA) complex array; temporaries of v_mov_b64 sequences live in `tmp`."""

    def __init__(self, g, a):
        self.g, self.a, self.c128 = g, a, getattr(g, 'ELEM', 8) == 16

    def amp(self, tok):
        lo, hi = (int(x) for x in _AMP.fullmatch(tok).groups())
        assert lo >= self.g.AMP0
        if self.c128:
            assert hi - lo == 3 and (lo - self.g.AMP0) % 4 == 0
            return (lo - self.g.AMP0) // 4
        assert hi - lo == 1 and (lo - self.g.AMP0) % 2 == 0
        return (lo - self.g.AMP0) // 2

    def is_amp_reg(self, tok):
        m = _AMP.fullmatch(tok)
        return bool(m) and int(m.group(1)) >= self.g.AMP0

    def get64(self, tok):
        lo = int(_AMP.fullmatch(tok).group(1))
        if not self.c128:
            return self.a[:, (lo - self.g.AMP0) // 2].copy()
        j, comp = divmod((lo - self.g.AMP0) // 2, 2)
        return (self.a[:, j].imag if comp else self.a[:, j].real).copy()

    def set64(self, tok, val, active):
        lo = int(_AMP.fullmatch(tok).group(1))
        if not self.c128:
            self.a[active, (lo - self.g.AMP0) // 2] = val[active]
            return
        j, comp = divmod((lo - self.g.AMP0) // 2, 2)
        col = self.a[:, j].copy()
        if comp:
            col.imag[active] = val[active]
        else:
            col.real[active] = val[active]
        self.a[:, j] = col


def _run_moves(g, lines, a, active):
    """Interpret v_mov_b64 lines (X gates, slot swaps) on the register file; `active` = exec mask."""
    regs, tmp = _Regs(g, a), {}
    for ln in lines:
        op, rest = ln.split(' ', 1)
        assert op == 'v_mov_b64', ln
        dst, src = [t.strip() for t in rest.split(',')]
        val = tmp[src] if src in tmp else regs.get64(src)
        if regs.is_amp_reg(dst):
            regs.set64(dst, val, active)
        else:
            tmp[dst] = val


def _apply2(a, lo, hi, m, active):
    x0, x1 = a[:, lo].copy(), a[:, hi].copy()
    a[active, lo] = (m[0] * x0 + m[1] * x1)[active]
    a[active, hi] = (m[2] * x0 + m[3] * x1)[active]


def run_pass(desc, n, state, mats, mat_batch_stride, grads=None, known_zero: int = 0, out=None):
    """state (B, 2^n) complex numpy -> the state after the pass, as the wave-tile kernel of its precision computes it.
    ``known_zero`` (dq_apply_fused_zext_*): the index bits known to be |0> in the input -- the kernel's loads and its tile
    count follow the descriptor's `zext` word; ``out`` (B, 2^n) is then written only where the kernel writes."""
    c128 = state.dtype == np.complex128
    g = gen(c128)
    NA, R, EL, M_ = g.NA, g.R, (16 if c128 else 8), (11 if c128 else 12)
    NV = 2 * (R + 1)
    regs_of = lambda arr: _Regs(g, arr)      # noqa: E731
    kp = descriptor(desc, n, known_zero)
    nrec = kp.nrec_bytes // 32
    rec = [list(kp.rec[i]) for i in range(nrec)]
    out = np.empty_like(state) if out is None else out
    lanes = np.arange(64)
    lb = [(lanes >> b) & 1 for b in range(6)]
    lld = sum(lb[b].astype(np.int64) << kp.load_lane_shift[b] for b in range(6))
    lst = sum(lb[b].astype(np.int64) << kp.store_lane_shift[b] for b in range(6))
    ntiles = 1 << (kp.zext & 63)
    dead_slots, dead_lanes = (kp.zext >> 8) & 63, (kp.zext >> 16) & 63
    assert known_zero or (ntiles == 1 << (n - M_) and not dead_slots and not dead_lanes)
    live_lanes = (lanes & dead_lanes) == 0
    flat_m = np.asarray(mats).reshape(-1)
    trip_of = {g.ID_TRIP + i: (bin(mk).count('1'), mk) for i, mk in enumerate(g.TRIP_MASKS)}
    trip_of[g.ID_TRIP0] = (0, 0)
    swap_of = {g.ID_SWAP + i: pr for i, pr in enumerate(g.SWAP_PAIRS)}
    for b in range(state.shape[0]):
        flat_in = state[b]
        mb = flat_m[b * mat_batch_stride:] if mat_batch_stride else flat_m
        for tile in range(ntiles):
            tg = sum(((tile >> j) & 1) << kp.read_blk_pos[j] for j in range(24)) | kp.fix_read[0] | (kp.fix_read[1] << 32)
            tw = sum(((tile >> j) & 1) << kp.store_blk_pos[j] for j in range(24)) | kp.fix_write[0] | (kp.fix_write[1] << 32)
            a = np.zeros((64, NA), dtype=state.dtype)
            for piece in range(32):
                # (zero-extended loads: a piece whose slot pattern has a known-zero bit set is not loaded, nor are the
                # lanes with such a lane bit -- tools/gen_wave_asm.py, zext_load)
                if (piece << (0 if c128 else 1)) & dead_slots:
                    continue
                off = sum(kp.load_off[s] for s in range(5) if (piece >> s) & 1)
                addr = ((tg * EL + off + lld) // EL)[live_lanes]
                if c128:
                    a[live_lanes, piece] = flat_in[addr]
                else:
                    a[live_lanes, 2 * piece] = flat_in[addr]
                    a[live_lanes, 2 * piece + 1] = flat_in[addr + 1]
            tb = sum(lb[b_] * kp.tb_contrib[b_] for b_ in range(6))
            scale = np.complex128(1.0)
            moff = kp.mat_base_bytes // EL
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
                    rg = regs_of(a)
                    for ln in g.trip(k, mask):
                        if ln.startswith('ds_write_b'):
                            mm = re.fullmatch(r'ds_write_b(?:64|128) v\d+, (v\[\d+:\d+\])(?: offset:(\d+))?', ln)
                            j, imm = rg.amp(mm.group(1)), int(mm.group(2) or 0)
                            for lane in range(64):
                                ad = int(wbase[lane]) + imm
                                assert ad % EL == 0 and ad + EL <= (8704 if c128 else 8448), 'LDS write outside the wave\'s region'
                                lds[ad] = a[lane, j]
                        elif ln.startswith('ds_read_b'):
                            mm = re.fullmatch(r'ds_read_b(?:64|128) (v\[\d+:\d+\]), v\d+(?: offset:(\d+))?', ln)
                            j, imm = rg.amp(mm.group(1)), int(mm.group(2) or 0)
                            for lane in range(64):
                                a[lane, j] = lds[int(rbase[lane]) + imm]      # KeyError: reads what nobody wrote
                    continue
                if hid in swap_of:
                    _run_moves(g, g.slotswap(*swap_of[hid]), a, np.ones(64, bool))
                    continue
                if hid == getattr(g, 'ID_EXPZ', -1):
                    # <Z..Z> from the registers (expz_code): register signs in w5 / w7, lane parity mask w1, tile parity mask
                    # w2:w3 -- parity masks, not controls: every tile and every lane contributes
                    rs = np.array([(w[5 if j < 32 else 7] >> (j % 32)) & 1 for j in range(NA)])
                    lane_par = np.array([bin(int(tb[ln]) & w[1]).count('1') & 1 for ln in range(64)])
                    tile_par = bin(tg & oc).count('1') & 1
                    sign = 1.0 - 2.0 * ((lane_par[:, None] ^ rs[None, :]) ^ tile_par)
                    v = a.astype(np.complex128)
                    assert grads is not None, 'an expectation record outside a dq_apply_fused_grad call'
                    grads[b, w[6], 0] += float((sign * (v.real ** 2 + v.imag ** 2)).sum()) * abs(scale) ** 2
                    continue
                if not tile_ok:
                    continue
                if getattr(g, 'ID_GEN2', 1 << 30) <= hid < getattr(g, 'ID_GEN2', 1 << 30) + len(g.SWAP_PAIRS) * (4 if hasattr(g, 'ID_GEN2XC') else 3 if hasattr(g, 'ID_GEN2X') else 2 if hasattr(g, 'ID_GEN2R') else 1):
                    # dense gate on two slots a < b (gen2_code): matrix index = 2 * bit(b) + bit(a), w6 = swap the two
                    # index bits of the matrix first, w5 = group mask; the second range of ids: the bodies for a matrix
                    # promised real (gen2_body_real: the imaginary parts are never read)
                    real_body, pair = divmod(hid - g.ID_GEN2, len(g.SWAP_PAIRS))
                    a_, b_ = g.SWAP_PAIRS[pair]
                    m4 = mb[moff - 16:moff].reshape(4, 4)
                    if real_body in (1, 2):
                        m4 = m4.real.astype(m4.dtype)
                    if real_body in (2, 3):       # gen2_body_xreal: only the blocks (00, 11) and (01, 10) are read
                        m4 = np.where(np.array([[(i ^ j) in (0, 3) for j in range(4)] for i in range(4)]), m4, 0).astype(m4.dtype)
                    if w[6]:
                        perm = [0, 2, 1, 3]
                        m4 = m4[perm][:, perm]
                    for gi, j in enumerate(g.gen2_groups(a_, b_)):
                        if not (w[5] >> gi) & 1:
                            continue
                        regs = [j | (((r >> 1) & 1) << b_) | ((r & 1) << a_) for r in range(4)]
                        old = [a[:, r].copy() for r in regs]
                        for r in range(4):
                            new = sum(m4[r, c].astype(a.dtype) * old[c] for c in range(4))
                            a[active, regs[r]] = new[active]
                    continue
                if getattr(g, 'ID_GRAD', 1 << 30) <= hid < getattr(g, 'ID_GRAD', 1 << 30) + (g.R - 1) * getattr(g, 'GRAD_VARIANTS', 1):
                    # reduction of the reverse sweep (gen_wave_asm.py, grad_code / grad_code_reduced): target slot q,
                    # psi / lambda on slot 0; the variant says which sums the handler forms
                    variant, q = divmod(hid - g.ID_GRAD, g.R - 1)
                    q += 1
                    acc = np.zeros((2, 2), dtype=np.complex128)
                    for gi, j in enumerate(g.grad_groups(q)):
                        if not (w[5] >> gi) & 1:
                            continue
                        p_ = [a[active, j], a[active, j | (1 << q)]]
                        l_ = [a[active, j | 1], a[active, j | (1 << q) | 1]]
                        for a_ in range(2):
                            for b_ in range(2):
                                acc[a_, b_] += np.sum(l_[a_].astype(np.complex128) * np.conj(p_[b_].astype(np.complex128)))
                    acc *= abs(scale) ** 2
                    assert grads is not None, 'a reduction record outside a reverse-sweep pass'
                    comp = np.array([acc[0, 0].real, acc[0, 0].imag, acc[0, 1].real, acc[0, 1].imag,
                                     acc[1, 0].real, acc[1, 0].imag, acc[1, 1].real, acc[1, 1].imag])
                    if variant == 1:
                        comp[1::2] = 0.0
                    elif variant == 2:
                        comp = np.array([(acc[0, 0] + acc[1, 1]).real, 0, 0, (acc[0, 1] + acc[1, 0]).imag, 0, 0, 0, 0])
                    elif variant == 3:
                        comp[2:6] = 0.0
                    elif variant == 4:
                        comp = np.array([0, 0, 0, (acc[0, 1] + acc[1, 0]).imag, 0, 0, 0, 0])
                    grads[b, w[6]] += comp
                    continue
                if hid >= g.ID_DIAG1:
                    # diagonal gate (tools/gen_wave_asm.py, diag_code): four phases, candidates by the indices in w5,
                    # two per-lane selectors, PH0 / PH1 by the bit of one register slot, an optional register mask
                    four = hid >= g.ID_DIAG2
                    variant = (hid - g.ID_DIAG1) % NV
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
                    masked, v = variant >= R + 1, variant % (R + 1)
                    rmask = w[6] | (w[7] << 32)
                    for j in range(NA):
                        if masked and not (rmask >> j) & 1:
                            continue
                        ph = ph0 if v == 0 or not (j >> (v - 1)) & 1 else ph1
                        a[active, j] = (a[:, j] * ph.astype(a.dtype))[active]
                    continue
                if g.ID_GEN_U <= hid < g.ID_GEN_C:
                    mode, q = divmod(hid - g.ID_GEN_U, R)
                    everyone = np.ones(64, bool)
                    if mode == 2 and not c128:       # the deferred Rx block { f, (0, t), -, (flag, -) }
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
                    _run_moves(g, g.xlines(hid - g.ID_X_U), a, np.ones(64, bool))
                elif g.ID_X_C <= hid < g.ID_X_R:
                    _run_moves(g, g.xlines(hid - g.ID_X_C), a, active)
                elif g.ID_X_R <= hid < g.ID_X_R1:
                    q = hid - g.ID_X_R
                    for pi, (lo, hi) in enumerate(g.pairs(q)):
                        if (w[5] >> pi) & 1:
                            x0 = a[:, lo].copy()
                            a[active, lo] = a[active, hi]
                            a[active, hi] = x0[active]
                elif g.ID_X_R1 <= hid < g.ID_TRIP0:
                    q, cc = divmod(hid - g.ID_X_R1, R - 1)
                    c = cc if cc < q else cc + 1
                    _run_moves(g, g.xlines(q, 1 << c), a, active)
                else:
                    raise AssertionError(f'unknown handler id {hid}')
            a = (a * scale).astype(state.dtype)
            base = b
            for piece in range(32):
                off = sum(kp.store_off[s] for s in range(5) if (piece >> s) & 1)
                addr = (tw * EL + off + lst) // EL
                if c128:
                    out[base][addr] = a[:, piece]
                else:
                    out[base][addr] = a[:, 2 * piece]
                    out[base][addr + 1] = a[:, 2 * piece + 1]
    return out
