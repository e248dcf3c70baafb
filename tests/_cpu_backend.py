"""CPU test double for ``deepquantum_amd.backend`` -- TEST INFRASTRUCTURE ONLY.

Installed explicitly by tests (``backend.set_test_backend``) so that host logic -- the gate library,
the circuit driver, the fusion scheduler and the distributed routines -- can be checked in a container
without a GPU.  Gate application and reductions are delegated to the oracle; ``apply_fused`` is an
independent interpreter of the ``DqFusedPass`` descriptor that follows the semantics documented in
``include/dq_hip.h`` (tile = low L bits + gathered bits, rounds with register slots / thread bits,
control masks split into reg / thread / outside-tile), so descriptor bugs show up here before a kernel
ever runs.
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import statevec_oracle as oracle  # noqa: E402

from deepquantum_amd import _lib  # noqa: E402


class CpuTestBackend:
    def __init__(self):
        self.fused_calls = 0
        self.single_calls = 0

    # ---- single gate -------------------------------------------------------------------------------
    def apply_gate(self, state, mats, targets, controls, out):
        self.single_calls += 1
        res = oracle.apply_gate_bits(state, mats, targets, controls)
        out.copy_(res)
        return out

    def fused_geometry(self, is_c128, variant):
        table = {(False, 0): (12, 6, 64), (True, 0): (11, 5, 64)}
        return table[(bool(is_c128), variant)]

    # ---- fused pass interpreter --------------------------------------------------------------------
    def apply_fused(self, state, mats, mat_batch_stride, desc, out, grads=None, known_zero=0, slice_bits=None):
        self.fused_calls += 1
        n = state.shape[-1].bit_length() - 1
        bsz = state.shape[0]
        is128 = state.dtype == torch.complex128
        m, L, h = desc.m, desc.L, desc.h
        R = desc.slots
        assert (is128, m, R) in ((False, 12, 6), (True, 11, 5)), 'no such kernel (the wave tile of the precision)'
        vb = 0 if is128 else 1
        logt = m - R
        assert L + h == m and n >= m
        high_pos = [desc.high_pos[i] for i in range(h)]
        high_sorted = [desc.high_sorted[i] for i in range(h)]
        assert sorted(high_pos) == high_sorted and all(L <= p < n for p in high_pos)
        assert len(set(high_pos)) == h
        sl = [desc.load_rb[s] for s in range(R)]
        assert sl == sorted(set(sl)), 'load slots must be ascending and distinct'
        assert all((q == s) if s < vb else (L <= q < m) for s, q in enumerate(sl)), 'load layout not coalesced'
        # the store layout is explicit (include/dq_hip.h): any slots, the other tile bits as thread bits
        store_sl = [desc.store_rb[s] for s in range(R)]
        store_tb = [desc.store_tb[i] for i in range(logt)]
        assert sorted(store_sl + store_tb) == list(range(m)), 'store layout does not cover the tile'

        store_low = [desc.store_low_pos[i] for i in range(L)]
        store_high = [desc.store_high_pos[i] for i in range(h)]
        store_blk = [desc.store_blk_pos[j] for j in range(n - m)]
        assert sorted(store_low + store_high + store_blk) == list(range(n)), 'write positions are not a permutation of [0, n)'
        wpos = store_low + store_high                     # tile bit -> global bit on the write side
        if vb:
            assert wpos[store_sl[0]] == 0, 'complex64: store slot 0 must be written to bit 0 (two adjacent amplitudes per lane)'
        for sl_ in range(R):
            tl = desc.load_rb[sl_]
            assert desc.load_slot_off[sl_] == 1 << (tl if tl < L else high_pos[tl - L]), 'slot offset table wrong'
            assert desc.store_slot_off[sl_] == 1 << wpos[store_sl[sl_]], 'slot offset table wrong'
        # tile-local index -> global offset, tile index -> base
        e = np.arange(1 << m, dtype=np.int64)
        glob = e & ((1 << L) - 1)
        for i in range(h):
            glob |= ((e >> (L + i)) & 1) << high_pos[i]
        tiles = np.arange(1 << (n - m), dtype=np.int64) << L
        for p in high_sorted:
            tiles = ((tiles >> p) << (p + 1)) | (tiles & ((1 << p) - 1))
        idx = tiles[:, None] | glob[None, :]            # (ntiles, 2^m) global amplitude indices
        assert np.array_equal(np.sort(idx.reshape(-1)), np.arange(1 << n)), 'tiles do not partition the state'
        # write side: tile bit L + i -> store_high_pos[i], block-index bit j -> store_blk_pos[j]
        globw = np.zeros_like(e)
        for i in range(m):
            globw |= ((e >> i) & 1) << wpos[i]
        blk = np.arange(1 << (n - m), dtype=np.int64)
        tilesw = np.zeros_like(blk)
        for j in range(n - m):
            tilesw |= ((blk >> j) & 1) << store_blk[j]
        idxw = tilesw[:, None] | globw[None, :]
        assert np.array_equal(np.sort(idxw.reshape(-1)), np.arange(1 << n)), 'written tiles do not partition the state'
        if not np.array_equal(idx, idxw):
            assert out.data_ptr() != state.data_ptr(), 'a permuting pass needs in != out'

        # matrices of the pass lie back to back in gate order, readable MAT_PAD entries past the end
        run = desc.mat_base
        ngates = desc.rounds[desc.nrounds - 1].gate_end if desc.nrounds else 0
        for gi in range(ngates):
            g = desc.gates[gi]
            assert g.mat == run, 'matrix layout is not sequential'
            size = {_lib.FG_GEN1: 4, _lib.FG_X1: 0, _lib.FG_DIAG1: 4, _lib.FG_GEN2: 16, _lib.FG_DIAG2: 16, _lib.FG_GRAD: 0, _lib.FG_EXPZ: 0}[g.kind]
            assert g.mat_advance == size
            run += size
        per_sample = mats.shape[-1] if mats.ndim == 2 else mats.numel()
        assert run + _lib.MAT_PAD <= per_sample, 'matrix buffer lacks the prefetch pad'

        x = state.detach().numpy().copy()
        live_rows = np.ones(len(tiles), dtype=bool)
        if known_zero:
            # include/dq_hip.h, dq_apply_fused_zext_*: the input is zero -- and not read: it may hold anything -- where a
            # known-zero index bit is 1; the tiles in which such a bit OUTSIDE the tile is 1 are not written either
            assert grads is None and known_zero >> n == 0 and known_zero & ((1 << L) - 1) == 0
            x[:, (np.arange(1 << n, dtype=np.int64) & known_zero) != 0] = 0
            live_rows = (tiles & known_zero) == 0
        if slice_bits is not None:
            # include/dq_hip.h, dq_apply_fused_slice_*: only the tiles whose index bits `mask` (read side, outside the tile,
            # not known zero) equal `value` run; what the other slices wrote (or will write) in `out` is not touched
            smask, svalue = int(slice_bits[0]), int(slice_bits[1])
            tile_bits = ((1 << L) - 1) | sum(1 << p for p in high_pos)
            assert grads is None and smask >> n == 0 and smask & tile_bits == 0 and smask & known_zero == 0 and svalue & ~smask == 0
            live_rows = live_rows & ((tiles & smask) == svalue)
        flat_m = mats.detach().numpy().reshape(-1)
        for b in range(bsz):
            t = x[b][idx]                               # (ntiles, 2^m)
            mb = flat_m[b * mat_batch_stride :] if mat_batch_stride else flat_m
            # layout the registers are in: (slots, thread bits); the load layout has ascending thread bits
            lay = ([desc.load_rb[s] for s in range(R)],
                   [q for q in range(m) if q not in [desc.load_rb[s] for s in range(R)]])
            for r in range(desc.nrounds):
                rd = desc.rounds[r]
                rb = [rd.rb[s] for s in range(R)]
                tb = [rd.tb[i] for i in range(logt)]
                assert len(set(rb)) == R and all(q < m for q in rb)
                first = rd.gate_begin
                assert bool(rd.flags & _lib.ROUND_TRANSPOSE) == ((rb, tb) != lay), 'transposition flag wrong'
                assert rd.flags & ~(_lib.ROUND_TRANSPOSE | _lib.ROUND_TRANSPOSE_AFTER) == 0
                lay = (rb, tb)
                after = r == desc.nrounds - 1 and lay != (store_sl, store_tb)
                assert bool(rd.flags & _lib.ROUND_TRANSPOSE_AFTER) == after, 'final transposition flag wrong'
                assert sorted(rb + tb) == list(range(m)), 'slots + thread bits must cover the tile exactly'
                slotmask = sum(1 << q for q in rb)
                for gi in range(first, rd.gate_end):
                    g = desc.gates[gi]
                    assert g.thr_cmask & slotmask == 0, 'thread-control on a slot bit'
                    assert g.kind in (_lib.FG_GEN1, _lib.FG_X1, _lib.FG_DIAG1, _lib.FG_DIAG2, _lib.FG_GEN2, _lib.FG_GRAD, _lib.FG_EXPZ), 'the pass kernel takes one- and two-target dense gates, X, diagonal gates and the reductions'
                    assert g.fast == _lib.FAST_NONE, 'no handler ids since ABI 21'
                    assert (g.reg_cmask >> R) == 0
                    cm = g.thr_cmask
                    for s in range(R):
                        if (g.reg_cmask >> s) & 1:
                            cm |= 1 << rb[s]
                    tile_ok = (tiles & g.out_cmask) == g.out_cmask          # (ntiles,)
                    outside = g.out_cmask
                    assert all(((outside >> p) & 1) == 0 for p in range(L)) and all(
                        ((outside >> p) & 1) == 0 for p in high_pos
                    ), 'outside-control on a tile bit'
                    if g.kind == _lib.FG_EXPZ:
                        # include/dq_hip.h, DQ_FG_EXPZ: sum (-1)^popc(index & zmask) |a|^2 over the whole state; the Z
                        # bits come as control masks (cm = the tile-local ones, `outside` the others); row `reserved`
                        assert grads is not None and g.reserved < grads.shape[1], 'expectation record outside a dq_apply_fused_grad call'
                        par_e = np.zeros(e.shape, dtype=np.int64)
                        for bit in range(m):
                            if (cm >> bit) & 1:
                                par_e ^= (e >> bit) & 1
                        par_t = np.zeros(tiles.shape, dtype=np.int64)
                        for bit in range(n):
                            if (outside >> bit) & 1:
                                par_t ^= (tiles >> bit) & 1
                        sign = 1.0 - 2.0 * (par_t[:, None] ^ par_e[None, :])
                        v = t.astype(np.complex128)
                        grads[b, g.reserved, 0] += float((sign * (v.real ** 2 + v.imag ** 2)).sum())
                        continue
                    el_ok = (e & cm) == cm
                    if g.kind == _lib.FG_GRAD:
                        # include/dq_hip.h, DQ_FG_GRAD: G[a][b] = sum lambda[target = a] conj(psi[target = b]) over the
                        # controls' 1-subspace; register slot q2 tells psi (0) from lambda (1); row `reserved`
                        assert grads is not None, 'reduction record outside a reverse-sweep pass'
                        assert g.fast == _lib.FAST_NONE and g.q != g.q2 and g.q < R and g.q2 < R and g.reserved < grads.shape[1]
                        tbit, sbit = rb[g.q], rb[g.q2]
                        assert not (cm >> tbit) & 1 and not (cm >> sbit) & 1
                        e00 = e[el_ok & (((e >> tbit) & 1) == 0) & (((e >> sbit) & 1) == 0)]
                        psi = [t[tile_ok][:, e00].astype(np.complex128), t[tile_ok][:, e00 | (1 << tbit)].astype(np.complex128)]
                        lam = [t[tile_ok][:, e00 | (1 << sbit)].astype(np.complex128),
                               t[tile_ok][:, e00 | (1 << sbit) | (1 << tbit)].astype(np.complex128)]
                        G = np.array([[np.sum(lam[a_] * np.conj(psi[b_])) for b_ in range(2)] for a_ in range(2)])
                        # `loc`: the sums the caller will read; like the kernels, the double forms no others
                        comp = np.array([G[0, 0].real, G[0, 0].imag, G[0, 1].real, G[0, 1].imag,
                                         G[1, 0].real, G[1, 0].imag, G[1, 1].real, G[1, 1].imag])
                        assert g.loc in (0, 1, 2, 3, 4)
                        if g.loc == 4:
                            comp = np.array([0, 0, 0, (G[0, 1] + G[1, 0]).imag, 0, 0, 0, 0])
                        elif g.loc == 1:
                            comp[1::2] = 0.0
                        elif g.loc == 2:
                            comp = np.array([(G[0, 0] + G[1, 1]).real, 0, 0, (G[0, 1] + G[1, 0]).imag, 0, 0, 0, 0])
                        elif g.loc == 3:
                            comp[2:6] = 0.0
                        grads[b, g.reserved] += torch.from_numpy(comp)
                        continue
                    if g.kind in (_lib.FG_GEN1, _lib.FG_X1):
                        tbit = rb[g.q]
                        assert not (cm >> tbit) & 1
                        free = g.reg_cmask == 0 and g.thr_cmask == 0 and g.out_cmask == 0
                        mat = mb[g.mat : g.mat + 4].reshape(2, 2)
                        if g.kind == _lib.FG_GEN1 and g.loc == 2 and not is128 and free:
                            # uncontrolled Rx-like gate of a complex64 pass: the deferred form (include/dq_hip.h,
                            # DQ_MODE_RX): { f, i t, -, flag } stands for f [[1, it], [it, 1]] (flag 0, f real) or
                            # f [[it, 1], [1, it]] (flag 1, f imaginary), |t| <= 1
                            f, it, flag = mat[0, 0], mat[0, 1], mat[1, 1].real
                            assert it.real == 0 and abs(it.imag) <= 1 and flag in (0.0, 1.0), 'deferred Rx block malformed'
                            assert (f.imag == 0) if flag == 0 else (f.real == 0)
                            mat = f * (np.array([[1, it], [it, 1]]) if flag == 0 else np.array([[it, 1], [1, it]]))
                            mat = mat.astype(mb.dtype)
                        if g.kind == _lib.FG_GEN1 and g.loc == 1:
                            assert np.all(mat.imag == 0), 'gate promised a real matrix'
                        if g.kind == _lib.FG_GEN1 and g.loc == 3:
                            assert np.all(mat.imag == 0) and mat[0, 0] == mat[0, 1] == mat[1, 0] == -mat[1, 1], \
                                'gate promised a Hadamard-like matrix'
                        if g.kind == _lib.FG_GEN1 and g.loc == 2:
                            assert mat[0, 0].imag == 0 and mat[1, 1].imag == 0 and mat[0, 1].real == 0 and mat[1, 0].real == 0
                        if g.kind == _lib.FG_X1:
                            mat = np.array([[0, 1], [1, 0]], dtype=mat.dtype)
                        e0 = e[el_ok & (((e >> tbit) & 1) == 0)]
                        e1 = e0 | (1 << tbit)
                        a0, a1 = t[:, e0].copy(), t[:, e1].copy()
                        n0 = mat[0, 0] * a0 + mat[0, 1] * a1
                        n1 = mat[1, 0] * a0 + mat[1, 1] * a1
                        t[:, e0] = np.where(tile_ok[:, None], n0, a0)
                        t[:, e1] = np.where(tile_ok[:, None], n1, a1)
                    elif g.kind == _lib.FG_GEN2:
                        b1, b2 = rb[g.q], rb[g.q2]
                        assert b1 != b2 and not (cm >> b1) & 1 and not (cm >> b2) & 1
                        mat = mb[g.mat : g.mat + 16].reshape(4, 4)
                        if g.loc in (1, 4):
                            assert np.all(mat.imag == 0), 'gate promised a real 4x4 matrix'
                        if g.loc in (4, 5):      # (DQ_MODE_XREAL / _XCPLX: the entries off the two 2x2 blocks are never read)
                            keep = np.array([[(i ^ j) in (0, 3) for j in range(4)] for i in range(4)])
                            mat = np.where(keep, mat, 0)
                        e00 = e[el_ok & (((e >> b1) & 1) == 0) & (((e >> b2) & 1) == 0)]
                        es = [e00, e00 | (1 << b2), e00 | (1 << b1), e00 | (1 << b1) | (1 << b2)]
                        a = [t[:, q].copy() for q in es]
                        for i in range(4):
                            ni = sum(mat[i, j] * a[j] for j in range(4))
                            t[:, es[i]] = np.where(tile_ok[:, None], ni, a[i])
                    else:
                        two = g.kind == _lib.FG_DIAG2
                        assert g.kind in (_lib.FG_DIAG1, _lib.FG_DIAG2)

                        def bit_of(loc, q):
                            if loc == _lib.LOC_REG:
                                return ((e >> rb[q]) & 1)[None, :] * np.ones((len(tiles), 1), dtype=np.int64)
                            if loc == _lib.LOC_THR:
                                assert not (slotmask >> q) & 1 and q < m
                                return ((e >> q) & 1)[None, :] * np.ones((len(tiles), 1), dtype=np.int64)
                            assert loc == _lib.LOC_OUT
                            return ((tiles >> q) & 1)[:, None] * np.ones((1, len(e)), dtype=np.int64)

                        if two:
                            d = mb[g.mat : g.mat + 16].reshape(4, 4).diagonal()
                            sel = bit_of(g.loc, g.q) * 2 + bit_of(g.loc2, g.q2)
                        else:
                            d = mb[g.mat : g.mat + 4].reshape(2, 2).diagonal()
                            sel = bit_of(g.loc, g.q)
                        ph = d[sel]
                        ok = tile_ok[:, None] & el_ok[None, :]
                        t = np.where(ok, ph * t, t)
            x[b][idxw] = t          # (every index is written exactly once: idxw partitions the state)
        if slice_bits is not None:
            keep = out.detach().numpy().copy()
            assert out.data_ptr() != state.data_ptr() or np.array_equal(idx, idxw)
            for b in range(bsz):
                keep[b][idxw[live_rows].reshape(-1)] = x[b][idxw[live_rows].reshape(-1)]
            x = keep
        elif known_zero:
            # (what the kernel leaves untouched is poisoned here: a later pass that reads it -- a wrong mask -- shows)
            keep = np.full_like(x, complex(float('nan'), float('nan')))
            for b in range(bsz):
                keep[b][idxw[live_rows].reshape(-1)] = x[b][idxw[live_rows].reshape(-1)]
            x = keep
        out.copy_(torch.from_numpy(x))
        return out

    def defer_rx(self, flat, index):
        from deepquantum_amd import fusion

        return fusion.defer_rx(flat, index)          # (the tensor formulation the kernel is tested against)

    # ---- reductions ---------------------------------------------------------------------------------
    def expect_pauli(self, state, xmask, zmask):
        n = state.shape[-1].bit_length() - 1
        wires, basis = [], ''
        for p in range(n):
            xb, zb = (xmask >> p) & 1, (zmask >> p) & 1
            if xb or zb:
                wires.append(n - 1 - p)
                basis += 'y' if (xb and zb) else ('x' if xb else 'z')
        if not wires:
            return (torch.abs(state) ** 2).sum(-1).to(torch.float64)
        return oracle.expectation_pauli(state, wires, basis).to(torch.float64).clone()     # (a fresh tensor, like a kernel's output)

    def inner(self, bra, ket):
        return (bra.conj() * ket).sum(-1).to(torch.complex128)

    def probs(self, state):
        return torch.abs(state) ** 2

    def marginal(self, state, bits):
        n = state.shape[-1].bit_length() - 1
        wires = [n - 1 - b for b in bits]
        order = sorted(range(len(wires)), key=lambda i: wires[i])
        p = oracle.probabilities(state, wires).reshape([state.shape[0]] + [2] * len(wires))
        # oracle returns outcomes indexed by sorted wires; re-order axes to the requested bit order
        inv = [order.index(i) + 1 for i in range(len(wires))]
        return p.permute([0] + inv).reshape(state.shape[0], -1).to(torch.float64).clone()

    def gate_grad(self, x, gy, targets, controls):
        n = x.shape[-1].bit_length() - 1
        b = x.shape[0]
        wt = [n - 1 - t + 1 for t in targets]
        wc = [n - 1 - c + 1 for c in controls]
        rest = [i for i in range(1, n + 1) if i not in wt and i not in wc]
        perm = [0] + wt + rest + wc
        d = 1 << len(targets)

        def mat(t):
            t = t.reshape([b] + [2] * n).permute(perm).reshape(b, d, -1, 1 << len(controls))
            return t[..., -1]

        return (mat(gy) @ mat(x).mH).to(torch.complex128)

    # ---- shard helpers ------------------------------------------------------------------------------
    @staticmethod
    def _expand(nl, mask, value):
        c = np.arange(1 << (nl - bin(mask).count('1')), dtype=np.int64)
        out = np.zeros_like(c)
        src = 0
        for p in range(nl):
            if not (mask >> p) & 1:
                out |= ((c >> src) & 1) << p
                src += 1
        return torch.from_numpy(out | value)

    def pack(self, amps, mask, value):
        nl = amps.shape[-1].bit_length() - 1
        return amps[:, self._expand(nl, mask, value)].contiguous()

    def unpack_axpby(self, amps, x, y, coef, mask, value):
        nl = amps.shape[-1].bit_length() - 1
        idx = self._expand(nl, mask, value)
        if y is None:
            amps[:, idx] = x
        else:
            coef = coef.to(amps.dtype).reshape(-1, 2)
            amps[:, idx] = coef[:, 0:1] * x + coef[:, 1:2] * y
        return amps

    def permute_bits(self, amps, src_of_dst, out):
        nl = amps.shape[-1].bit_length() - 1
        i = np.arange(1 << nl, dtype=np.int64)
        sidx = np.zeros_like(i)
        for p, sp in enumerate(src_of_dst):
            sidx |= ((i >> p) & 1) << sp
        out.copy_(amps[:, torch.from_numpy(sidx)])
        return out
