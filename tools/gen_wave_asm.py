"""Generate csrc/dq_wave_asm.inc: the body of the wave-tile pass kernel (complex64).

One wavefront owns a 12-bit tile: 64 lanes x 64 amplitudes in v[40:167] (amplitude j = register-slot pattern j at
v[40 + 2j : 41 + 2j]).  The whole life of a tile -- 32 x 16-byte loads, the record loop (gates, layout changes), the
deferred scale, 32 stores -- is ONE asm statement with fixed registers; C++ only computes the addresses it starts from
(csrc/dq_wave.hip).  No workgroup barrier anywhere: a layout change ("trip") goes through a small wave-private LDS
buffer, sub-tile by sub-tile (2^k registers x 64 lanes), addressed with immediates.

Records (32 bytes, made by the translator in csrc/dq_wave.hip from the host's DqFusedPass):
  w0 handler id   w1 thread-control mask (tile-local bits)   w2:w3 outside-control mask   w4 matrix advance (complex
  numbers)   w5 pair mask (register-controlled gates)   w6, w7 unused
  trip: two records -- A: w0 id, w4 0, (w1 w2 w3 w5 w6 w7) = what lane bits 0..5 add to the thread's tile-local base in
  the new layout;  B: w0..w5 = what lane bit b adds to the LDS write address (low half) and read address (high half).
"""
import os

R = 6
NA = 1 << R
AMP0 = 40
T0, U0, T1, U1 = 'v[10:11]', 'v[12:13]', 'v[14:15]', 'v[16:17]'
TT, HR, HI, TB, LANE, WB, RB = 'v18', 'v19', 'v20', 'v21', 'v22', 'v23', 'v24'
LB = [f'v{25 + b}' for b in range(6)]            # the lane's bits as masks: 0 / 0xffffffff
ADDR = ['v[32:33]', 'v[34:35]', 'v[36:37]', 'v[38:39]']
LLD, LST = 'v[2:3]', 'v[4:5]'                    # the lane's byte offset in the load / store layout
SLOTOFF = [f's[{40 + 2 * i}:{41 + 2 * i}]' for i in range(5)]     # what slots 1..5 add (bytes)
RUN, SAVE, TABLE, HADC = 's[50:51]', 's[52:53]', 's[54:55]', 's[56:57]'
GOFF, GEND, MOFF, STMP = 's58', 's59', 's60', 's61'
MB, KG, TG, LDSB = 's[62:63]', 's[64:65]', 's[66:67]', 's68'
REC, REC2, MAT = 72, 80, 88
NMAT = 40            # the next record's matrix, fetched one record ahead like the next record itself (REC2); the slot
                     # offsets that live in s[40:49] are only needed by the loads and the stores, outside the record loop
M = [f's[{MAT + 2 * i}:{MAT + 2 * i + 1}]' for i in range(4)]     # m00 m01 m10 m11


def A(j):
    return f'v[{AMP0 + 2 * j}:{AMP0 + 2 * j + 1}]'


def pairs(q, cmask=0):
    return [(j, j | (1 << q)) for j in range(NA) if not (j >> q) & 1 and (j & cmask) == cmask]


RE2, RE3 = 'op_sel_hi:[1,0]', 'op_sel_hi:[1,0,1]'
I2 = 'op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]'
I3 = 'op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]'


def pair_general(lo, hi, t, u):
    a_, b_ = A(lo), A(hi)
    return [f'v_pk_mul_f32 {t}, {b_}, {M[1]} {RE2}', f'v_pk_mul_f32 {u}, {a_}, {M[2]} {RE2}',
            f'v_pk_fma_f32 {t}, {b_}, {M[1]}, {t} {I3}', f'v_pk_fma_f32 {u}, {a_}, {M[2]}, {u} {I3}',
            f'v_pk_fma_f32 {t}, {a_}, {M[0]}, {t} {I3}', f'v_pk_fma_f32 {u}, {b_}, {M[3]}, {u} {I3}',
            f'v_pk_fma_f32 {a_}, {a_}, {M[0]}, {t} {RE3}', f'v_pk_fma_f32 {b_}, {b_}, {M[3]}, {u} {RE3}']


def interleave(seqs):
    """Instruction lists of independent computations merged round-robin: no instruction reads what the one right before
    it wrote (a dependent packed operation waits ~2 issue slots of its wave; the H and Rx bodies have no such pairs)."""
    out_, i = [], 0
    while any(i < len(s_) for s_ in seqs):
        out_ += [s_[i] for s_ in seqs if i < len(s_)]
        i += 1
    return out_


def body(mode, q):
    lines = []
    ps = pairs(q)
    if mode == 3:      # s [[1, 1], [1, -1]], s deferred: all sums first, then all differences B' = A' - 2 B
        for lo, hi in ps:
            lines.append(f'v_pk_add_f32 {A(lo)}, {A(lo)}, {A(hi)}')
        for lo, hi in ps:
            lines.append(f'v_pk_fma_f32 {A(hi)}, {A(hi)}, {HADC}, {A(lo)}')
        return lines
    for k in range(0, len(ps), 2):      # two pairs at a time: four independent chains
        seqs = []
        for (lo, hi), (t, u) in zip(ps[k:k + 2], ((T0, U0), (T1, U1))):
            a_, b_ = A(lo), A(hi)
            if mode == 1:    # all entries real
                seqs += [[f'v_pk_mul_f32 {t}, {b_}, {M[1]} {RE2}', f'v_pk_fma_f32 {a_}, {a_}, {M[0]}, {t} {RE3}'],
                         [f'v_pk_mul_f32 {u}, {a_}, {M[2]} {RE2}', f'v_pk_fma_f32 {b_}, {b_}, {M[3]}, {u} {RE3}']]
            else:
                g_ = pair_general(lo, hi, t, u)
                seqs += [g_[0::2], g_[1::2]]
        if mode == 1:
            # (the real body overwrites A before the second chain has read it: keep the reads first)
            lines += [s_[0] for s_ in seqs] + [s_[1] for s_ in seqs]
        else:
            lines += interleave(seqs)
    return lines


def body_rx_deferred(q, tag):
    """f [[1, it], [it, 1]] (flag 0) or f [[it, 1], [1, it]] (flag 1): the host's deferred block (include/dq_hip.h,
    DQ_MODE_RX); f multiplies the pass's deferred factor hr + i hi."""
    it, fr, fi, flag = M[1], f's{MAT}', f's{MAT + 1}', f's{MAT + 6}'
    f0, f1 = [], []
    for k, (lo, hi) in enumerate(pairs(q)):
        t = T0 if k % 2 == 0 else T1
        a_, b_ = A(lo), A(hi)
        f0 += [f'v_mov_b64 {t}, {a_}', f'v_pk_fma_f32 {a_}, {b_}, {it}, {a_} {I3}', f'v_pk_fma_f32 {b_}, {t}, {it}, {b_} {I3}']
        f1 += [f'v_mov_b64 {t}, {a_}', f'v_pk_fma_f32 {a_}, {a_}, {it}, {b_} {I3}', f'v_pk_fma_f32 {b_}, {b_}, {it}, {t} {I3}']
    l1, l2 = f'.Lrx1_{tag}_%=', f'.Lrx2_{tag}_%='
    return ([f's_cmp_eq_u32 {flag}, 0', f's_cbranch_scc0 {l1}', f'v_mul_f32 {HR}, {HR}, {fr}', f'v_mul_f32 {HI}, {HI}, {fr}'] + f0 +
            [f's_branch {l2}', f'{l1}:',
             f'v_mul_f32 {TT}, {HR}, {fi}', f'v_mul_f32 {HR}, {HI}, {fi}', f'v_xor_b32 {HR}, 0x80000000, {HR}',
             f'v_mov_b32 {HI}, {TT}'] + f1 + [f'{l2}:'])


def xlines(q, cmask=0):
    out_ = []
    for k, (lo, hi) in enumerate(pairs(q, cmask)):
        t = T0 if k % 2 == 0 else T1
        out_ += [f'v_mov_b64 {t}, {A(lo)}', f'v_mov_b64 {A(lo)}, {A(hi)}', f'v_mov_b64 {A(hi)}, {t}']
    return out_


def masked(q, tag, per_pair):
    """Pair i of slot q (in the order of pairs(q)) acts only if bit i of the record's pair mask (w5) is set."""
    out_ = []
    for i, (lo, hi) in enumerate(pairs(q)):
        t, u = (T0, U0) if i % 2 == 0 else (T1, U1)
        out_ += [f's_bitcmp1_b32 s{REC + 5}, {i}', f's_cbranch_scc0 .Lm{tag}_{i}_%='] + per_pair(lo, hi, t, u) + [f'.Lm{tag}_{i}_%=:']
    return out_


def slotswap(i, j):
    out_ = []
    k = 0
    for r in range(NA):
        if (r >> i) & 1 and not (r >> j) & 1:
            o = r ^ (1 << i) ^ (1 << j)
            t = T0 if k % 2 == 0 else T1
            out_ += [f'v_mov_b64 {t}, {A(r)}', f'v_mov_b64 {A(r)}, {A(o)}', f'v_mov_b64 {A(o)}, {t}']
            k += 1
    return out_


def prefetch(tag, mat=True):
    """Fetch the record at GOFF (and its matrix at MOFF) into the look-ahead registers, if there is one: the scalar
    loads' latency (~300 cycles each record otherwise) hides behind the current record's work."""
    return [f's_cmp_lt_u32 {GOFF}, {GEND}', f's_cbranch_scc0 .Lnp{tag}_%=',
            f's_load_dwordx8 s[{REC2}:{REC2 + 7}], {KG}, {GOFF}'] + \
        ([f's_load_dwordx8 s[{NMAT}:{NMAT + 7}], {MB}, {MOFF}'] if mat else []) + [f'.Lnp{tag}_%=:']


def deposit(val, positions):
    return sum(((val >> i) & 1) << p for i, p in enumerate(positions))


def trip(k, mask):
    """The slots of `mask` (k of them) trade places with k lane bits; all lanes may be re-ordered.  Sub-tile element
    (x = pattern of the outgoing slot bits, y = of the incoming bits, z = of the lane bits that stay) lives at slot
    x * S + (y << a) + F(z), S = 64 + 2^a, a = 5 - k: written with immediate x * S, read with immediate y << a."""
    tbw = [REC + 1, REC + 2, REC + 3, REC + 5, REC + 6, REC + 7]
    pre = [f'v_and_b32 {TB}, s{tbw[0]}, {LB[0]}'] + [f'v_and_or_b32 {TB}, {LB[b]}, s{tbw[b]}, {TB}' for b in range(1, 6)]
    pre += ['s_waitcnt lgkmcnt(0)']           # record B = the look-ahead record
    pre += [f'v_and_b32 {WB}, s{REC2}, {LB[0]}'] + [f'v_and_or_b32 {WB}, {LB[b]}, s{REC2 + b}, {WB}' for b in range(1, 6)]
    pre += [f's_add_u32 {GOFF}, {GOFF}, 32'] + prefetch(f't{mask}', mat=False)      # (trips advance no matrix: NMAT stays)
    pre += [f'v_lshrrev_b32 {RB}, 16, {WB}', f'v_and_b32 {WB}, 0xffff, {WB}', f'v_add_u32 {WB}, {LDSB}, {WB}', f'v_add_u32 {RB}, {LDSB}, {RB}']
    if k == 0:
        body_ = []
        for j in range(NA):
            body_ += [f'ds_write_b64 {WB}, {A(j)}', f'ds_read_b64 {A(j)}, {RB}']
        return pre + body_
    a = 5 - k
    S = 64 + (1 << a)
    moving = [s for s in range(R) if (mask >> s) & 1]
    staying = [s for s in range(R) if not (mask >> s) & 1]
    body_ = []
    for g in range(1 << (R - k)):
        base = deposit(g, staying)
        for x in range(1 << k):
            body_.append(f'ds_write_b64 {WB}, {A(base | deposit(x, moving))} offset:{8 * S * x}')
        for y in range(1 << k):
            body_.append(f'ds_read_b64 {A(base | deposit(y, moving))}, {RB} offset:{8 * (y << a)}')
    return pre + body_


MAXK = 4
TRIP_MASKS = [m for m in range(1, 64) if bin(m).count('1') <= MAXK]
SWAP_PAIRS = [(i, j) for i in range(R) for j in range(i + 1, R)]

# handler ids
ID_GEN_U = 0        # + 6 * mode + q          mode: 0 general, 1 real, 2 Rx-like (deferred block), 3 Hadamard-like
ID_GEN_C = 24       # + q   any 2x2 with thread / outside controls
ID_GEN_R = 30       # + q   ... with a register-pair mask as well
ID_X_U = 36         # + q
ID_X_C = 42         # + q
ID_X_R = 48         # + q   pair mask
ID_X_R1 = 54        # + 5 * q + c'   one register control c (c' = c, or c - 1 above q), thread / outside controls too
ID_TRIP0 = 84
ID_TRIP = 85        # + index in TRIP_MASKS
ID_SWAP = ID_TRIP + len(TRIP_MASKS)       # + index in SWAP_PAIRS
# diagonal gates: + variant (0: one phase for every register, 1 + q: by the bit of slot q, 7 / 8 + q: the same with a
# register mask); ID_DIAG2: the four phases are the diagonal of a 4x4 block that ends at the current matrix offset
ID_DIAG1 = ID_SWAP + len(SWAP_PAIRS)
ID_DIAG2 = ID_DIAG1 + 14
# reduction of the adjoint method's reverse sweep: target on slot 1 + (id - ID_GRAD), psi / lambda told apart by slot 0
ID_GRAD = ID_DIAG2 + 14
# + 5 * variant + (slot - 1).  Variants (DqFusedGate::loc of a DQ_FG_GRAD record: what the trainable gate's matrix can
# depend on decides which sums its gradient needs): 0 all of G; 1 Re G only (a real matrix: Ry); 2 Re (G00 + G11) and
# Im (G01 + G10) only (a I + i b X: Rx, CRx); 3 the diagonal G00, G11 only (a diagonal gate on a slot); 4 Im (G01 + G10)
# alone (a UNITARY a I + i b X whose cotangent is only ever contracted with a tangent of the rotation: dM M^-1 = -i X / 2, the
# trace part drops out -- a first-order backward that records no graph)
GRAD_VARIANTS = 5
# expectation value of a Z string, reduced from the registers (DQ_FG_EXPZ): same accumulators as the reductions above
ID_EXPZ = ID_GRAD + 5 * GRAD_VARIANTS
# dense gate on two register slots a < b (index in SWAP_PAIRS): a 4x4 matrix, index = 2 * (bit of slot b) + (bit of slot a)
ID_GEN2 = ID_EXPZ + 1
# ... + 15: the same for a matrix the gate class promises to be REAL (DqFusedGate::loc = 1: the superoperators of noise
# channels): one packed operation per entry instead of two
ID_GEN2R = ID_GEN2 + 15
# ... + 15: REAL and X-SHAPED (DqFusedGate::loc = 4): non-zero only where row and column index agree in parity of their two
# bits -- a real 2x2 on (00, 11) and one on (01, 10).  Every non-diagonal channel of the reference's menu (bit flip,
# bit-phase flip, depolarizing, Pauli, amplitude damping, generalized amplitude damping) has such a superoperator on its
# (row, column) bit pair: ten packed operations per register group instead of twenty
ID_GEN2X = ID_GEN2R + 15
# ... + 15: X-SHAPED, complex (DqFusedGate::loc = 5): the same two 2x2 blocks with complex entries -- exp(-i theta XX / 2),
# exp(-i theta YY / 2), exp(-i theta (XX + YY) / 4) and their controlled forms: twenty operations per group instead of forty
ID_GEN2XC = ID_GEN2X + 15
NIDS = ID_GEN2XC + 15
ACC_BASE = 4 * 8448       # LDS offset of the reduction accumulators: behind the four waves' staging buffers


def handlers():
    h = {}
    for q in range(R):
        h[ID_GEN_U + 0 * 6 + q] = (False, body(0, q))
        h[ID_GEN_U + 1 * 6 + q] = (False, body(1, q))
        h[ID_GEN_U + 2 * 6 + q] = (False, body_rx_deferred(q, f'u{q}'))
        h[ID_GEN_U + 3 * 6 + q] = (False, [f'v_mul_f32 {HR}, {HR}, s{MAT}', f'v_mul_f32 {HI}, {HI}, s{MAT}'] + body(3, q))
        h[ID_GEN_C + q] = (True, body(0, q))
        h[ID_GEN_R + q] = (True, masked(q, f'g{q}', pair_general))
        h[ID_X_U + q] = (False, xlines(q))
        h[ID_X_C + q] = (True, xlines(q))
        h[ID_X_R + q] = (True, masked(q, f'x{q}', lambda lo, hi, t, u: [f'v_mov_b64 {t}, {A(lo)}', f'v_mov_b64 {A(lo)}, {A(hi)}', f'v_mov_b64 {A(hi)}, {t}']))
        for c in range(R):
            if c != q:
                h[ID_X_R1 + 5 * q + (c if c < q else c - 1)] = (True, xlines(q, 1 << c))
    for q in range(1, R):
        h[ID_GRAD + q - 1] = (False, grad_code(q))
        for v in range(1, GRAD_VARIANTS):
            h[ID_GRAD + 5 * v + q - 1] = (False, grad_code_reduced(q, v))
    h[ID_EXPZ] = (False, expz_code())
    h[ID_TRIP0] = (False, trip(0, 0))
    for i, m in enumerate(TRIP_MASKS):
        h[ID_TRIP + i] = (False, trip(bin(m).count('1'), m))
    for i, (a_, b_) in enumerate(SWAP_PAIRS):
        h[ID_SWAP + i] = (False, slotswap(a_, b_))
    return h


PH0, PH1 = 'v[10:11]', 'v[12:13]'


def cmul_inplace(j, ph, t):
    return [f'v_pk_mul_f32 {t}, {A(j)}, {ph} op_sel_hi:[1,0]', f'v_pk_fma_f32 {A(j)}, {A(j)}, {ph}, {t} {I3}']


def diag_body(variant):
    """variant 0: PH0 for every register; 1 + q: PH0 / PH1 by bit q of the register index; + 7: only the registers whose
    bit is set in the record's 64-bit mask (w6:w7)."""
    masked, v = variant >= 7, variant % 7
    out_ = []
    for j in range(NA):
        ph = PH0 if v == 0 or not (j >> (v - 1)) & 1 else PH1
        t = 'v[14:15]' if j % 2 == 0 else 'v[16:17]'
        if masked:
            out_ += [f's_bitcmp1_b64 s[{REC + 6}:{REC + 7}], {j}', f's_cbranch_scc0 .Ldg{variant}_{j}_%=']
        out_ += cmul_inplace(j, ph, t)
        if masked:
            out_.append(f'.Ldg{variant}_{j}_%=:')
    return out_


def diag_code():
    """Entry of every diagonal record: controls, the four phases d0..d3 -> s[40:47], the candidates of PH0 / PH1 (indices
    in w5 bits 16..31) -> s[80:87] / s[96:99], the two per-lane selectors (w5 bytes 0, 1: kind << 6 | position;
    kind 1 = a tile-local bit of the thread, 2 = an index bit outside the tile) -> s[50:51], s[70:71], then
    PH = selA ? (selB ? c3 : c2) : (selB ? c1 : c0) per lane, and on to the body of the variant."""
    W5 = f's{REC + 5}'
    t = ['.Ldiag_%=:', 's_waitcnt lgkmcnt(0)',        # (the look-ahead matrix is on its way into s[40:47])
         f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=',
         f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
         's_cbranch_execz .Lrestore_%=',
         f's_cmp_ge_u32 s{REC}, {ID_DIAG2}', 's_cbranch_scc1 .Ldiag4_%=',
         f's_mov_b64 s[40:41], {M[0]}', f's_mov_b64 s[42:43], {M[3]}', f's_mov_b64 s[44:45], {M[0]}', f's_mov_b64 s[46:47], {M[3]}',
         's_branch .Ldiagsel_%=', '.Ldiag4_%=:', f's_sub_u32 s69, {MOFF}, 128']
    for k in range(4):
        t += [f's_load_dwordx2 s[{40 + 2 * k}:{41 + 2 * k}], {MB}, s69'] + (['s_add_u32 s69, s69, 40'] if k < 3 else [])
    t += ['s_waitcnt lgkmcnt(0)', '.Ldiagsel_%=:']
    cand0 = ['s[80:81]', 's[82:83]', 's[84:85]', 's[86:87]']
    cand1 = {0: 's[96:97]', 2: 's[98:99]'}      # (PH1 exists where a slot bit is one of the targets: one selector at most)
    for k in range(4):
        t += [f's_bfe_u32 s69, {W5}, {(16 + 2 * k) | (2 << 16)}', 's_lshl_b32 m0, s69, 1', 's_nop 0', f's_movrels_b64 {cand0[k]}, s[40:41]']      # (one wait state between a write of M0 and s_movrel)
    for k in (0, 2):
        t += [f's_bfe_u32 s69, {W5}, {(24 + 2 * k) | (2 << 16)}', 's_lshl_b32 m0, s69, 1', 's_nop 0', f's_movrels_b64 {cand1[k]}, s[40:41]']
    for name, byte, dst in (('a', 0, 's[50:51]'), ('b', 8, 's[70:71]')):
        t += [f's_bfe_u32 s69, {W5}, {byte | (6 << 16)}', f's_bfe_u32 vcc_lo, {W5}, {(byte + 6) | (2 << 16)}', f's_mov_b64 {dst}, 0',
              's_cmp_eq_u32 vcc_lo, 1', f's_cbranch_scc0 .Lsel{name}o_%=',
              f'v_lshrrev_b32 {TT}, s69, {TB}', f'v_and_b32 {TT}, 1, {TT}', f'v_cmp_ne_u32 {dst}, 0, {TT}', f's_branch .Lsel{name}d_%=',
              f'.Lsel{name}o_%=:', 's_cmp_eq_u32 vcc_lo, 2', f's_cbranch_scc0 .Lsel{name}d_%=',
              f's_lshr_b64 vcc, {TG}, s69', 's_bitcmp1_b32 vcc_lo, 0', f's_cselect_b64 {dst}, -1, 0', f'.Lsel{name}d_%=:']
    for half in (0, 1):
        regs = [f's{int(c[2:c.index(":")]) + half}' for c in cand0]
        t += [f'v_mov_b32 v14, {regs[0]}', f'v_mov_b32 v15, {regs[1]}', f'v_mov_b32 v16, {regs[2]}', f'v_mov_b32 v17, {regs[3]}',
              'v_cndmask_b32 v14, v14, v15, s[70:71]', 'v_cndmask_b32 v16, v16, v17, s[70:71]',
              f'v_cndmask_b32 v{10 + half}, v14, v16, s[50:51]']
        r0, r2 = (f's{int(cand1[k][2:cand1[k].index(":")]) + half}' for k in (0, 2))
        t += [f'v_mov_b32 v14, {r0}', f'v_mov_b32 v16, {r2}', f'v_cndmask_b32 v{12 + half}, v14, v16, s[50:51]']
    t += prefetch('dg')      # (s[80:87] held the candidates, s[40:47] the phases: fetch the look-ahead record and matrix again)
    # second-level table: the body of the variant
    t += [f's_sub_u32 s69, s{REC}, {ID_DIAG1}', 's_cmp_ge_u32 s69, 14', 's_cbranch_scc0 .Ldiagv_%=', 's_sub_u32 s69, s69, 14', '.Ldiagv_%=:',
          's_getpc_b64 vcc', '.Ldiaganchor_%=:', 's_lshl2_add_u32 vcc_lo, s69, vcc_lo', 's_addc_u32 vcc_hi, vcc_hi, 0',
          's_add_u32 vcc_lo, vcc_lo, .Ldiagtable_%=-.Ldiaganchor_%=', 's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc', '.Ldiagtable_%=:']
    t += [f's_branch .Ldiagb{v}_%=' for v in range(14)]
    for v in range(14):
        t += [f'.Ldiagb{v}_%=:'] + diag_body(v) + [f's_mov_b64 exec, {SAVE}', 's_branch .Lnext_%=']
    return t


def grad_groups(q):
    return [j for j in range(NA) if not (j >> q) & 1 and not j & 1]


def grad_code(q):
    """DQ_FG_GRAD with the target on slot q, psi (0) / lambda (1) on slot 0: G[a][b] = sum lambda[target = a] conj(psi[target
    = b]) over the thread's register groups (w5 = mask of the groups whose register controls are set), lanes that fail the
    thread controls contribute nothing; |f|^2 of the pass's deferred factor; reduce-scatter over each row of 16 lanes (16
    DPP adds), then 32 lanes add to the record's eight accumulators in LDS (ACC_BASE + 32 * record number)."""
    G = ['v[10:11]', 'v[12:13]', 'v[14:15]', 'v[16:17]']
    C2 = 'op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]'
    t = [f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=']
    t += [f'v_mov_b32 v{r}, 0' for r in range(10, 18)]
    t += [f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
          f's_cbranch_execz .Lgz{q}_%=']
    for i, j in enumerate(grad_groups(q)):
        p0, l0, p1, l1 = A(j), A(j | 1), A(j | (1 << q)), A(j | (1 << q) | 1)
        t += [f's_bitcmp1_b32 s{REC + 5}, {i}', f's_cbranch_scc0 .Lgg{q}_{i}_%=']
        for g_, l_, p_ in ((G[0], l0, p0), (G[1], l0, p1), (G[2], l1, p0), (G[3], l1, p1)):
            t.append(f'v_pk_fma_f32 {g_}, {l_}, {p_}, {g_} {RE3}')
        for g_, l_, p_ in ((G[0], l0, p0), (G[1], l0, p1), (G[2], l1, p0), (G[3], l1, p1)):
            t.append(f'v_pk_fma_f32 {g_}, {l_}, {p_}, {g_} {C2}')
        t.append(f'.Lgg{q}_{i}_%=:')
    g = [f'v{r}' for r in range(10, 18)]
    t += [f'.Lgz{q}_%=:', f's_mov_b64 exec, {SAVE}',
          f'v_mul_f32 {TT}, {HR}, {HR}', f'v_fma_f32 {TT}, {HI}, {HI}, {TT}', 's_nop 1',
          # lane bit 2: lanes with the bit clear take over the even-numbered sum of each pair, the others the odd one
          f'v_add_f32_dpp {g[0]}, {g[0]}, {g[0]} row_shl:4 row_mask:0xf bank_mask:0x5', f'v_add_f32_dpp {g[2]}, {g[2]}, {g[2]} row_shl:4 row_mask:0xf bank_mask:0x5',
          f'v_add_f32_dpp {g[4]}, {g[4]}, {g[4]} row_shl:4 row_mask:0xf bank_mask:0x5', f'v_add_f32_dpp {g[6]}, {g[6]}, {g[6]} row_shl:4 row_mask:0xf bank_mask:0x5',
          f'v_add_f32_dpp {g[0]}, {g[1]}, {g[1]} row_shr:4 row_mask:0xf bank_mask:0xa', f'v_add_f32_dpp {g[2]}, {g[3]}, {g[3]} row_shr:4 row_mask:0xf bank_mask:0xa',
          f'v_add_f32_dpp {g[4]}, {g[5]}, {g[5]} row_shr:4 row_mask:0xf bank_mask:0xa', f'v_add_f32_dpp {g[6]}, {g[7]}, {g[7]} row_shr:4 row_mask:0xf bank_mask:0xa',
          # lane bit 3
          f'v_add_f32_dpp {g[0]}, {g[0]}, {g[0]} row_shl:8 row_mask:0xf bank_mask:0x3', f'v_add_f32_dpp {g[4]}, {g[4]}, {g[4]} row_shl:8 row_mask:0xf bank_mask:0x3',
          's_nop 1',
          f'v_add_f32_dpp {g[0]}, {g[2]}, {g[2]} row_shr:8 row_mask:0xf bank_mask:0xc', f'v_add_f32_dpp {g[4]}, {g[6]}, {g[6]} row_shr:8 row_mask:0xf bank_mask:0xc',
          # lane bit 0: even lanes keep g0, odd lanes g4
          's_mov_b32 vcc_lo, 0xaaaaaaaa', 's_mov_b32 vcc_hi, 0xaaaaaaaa', 's_nop 1',
          f'v_add_f32_dpp v32, {g[0]}, {g[0]} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf',
          f'v_add_f32_dpp v33, {g[4]}, {g[4]} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf',
          f'v_cndmask_b32 {g[0]}, v32, v33, vcc',
          # lane bit 1: plain add
          's_nop 1', f'v_add_f32_dpp {g[0]}, {g[0]}, {g[0]} quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf',
          f'v_mul_f32 {g[0]}, {g[0]}, {TT}',
          # the lane with bits (b0, b3, b2) holds sum number 4 b0 + 2 b3 + b2; lanes that differ in bit 1 hold the same
          f'v_and_b32 v32, 1, {LANE}', 'v_lshlrev_b32 v32, 2, v32', f'v_lshrrev_b32 v33, 2, {LANE}', 'v_and_b32 v34, 2, v33', 'v_and_b32 v33, 1, v33',
          'v_or3_b32 v32, v32, v34, v33', 'v_lshlrev_b32 v32, 2, v32', f'v_add_u32 v32, {GOFF}, v32', f'v_add_u32 v32, {ACC_BASE - 32}, v32',
          's_mov_b32 exec_lo, 0x33333333', 's_mov_b32 exec_hi, 0x33333333',
          f'ds_add_f32 v32, {g[0]}',
          's_mov_b64 exec, -1']
    return t


def grad_code_reduced(q, variant):
    """The reduction of `grad_code` for a trainable gate whose matrix is known to be real (variant 1: only Re G is needed,
    four real sums), of the form a I + i b X (variant 2: only Re (G00 + G11) and Im (G01 + G10), two real sums -- its
    gradient has no other component) or diagonal (variant 3: G00 and G11).  Per register group four packed operations
    instead of eight, 2 or 4 sums per lane instead of 8: a reduce-scatter over one or two lane bits, butterflies over the
    others of a row of 16 lanes, and 2 or 4 lanes of every row add to the record's accumulators.  The sums land in the
    components of the full layout (G00 G01 G10 G11) x (re, im): variant 2 leaves Re (G00 + G11) in Re G00 and
    Im (G01 + G10) in Im G01 -- multiplied by the inverse of a matrix of the same form that is all the chain rule reads."""
    ACC = ['v[10:11]', 'v[12:13]', 'v[14:15]', 'v[16:17]']
    CROSS = 'op_sel:[0,1,0] op_sel_hi:[1,0,1]'        # (l.re p.im, l.im p.re): Im (l conj p) = hi - lo
    tag = f'{q}v{variant}'
    t = [f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=']
    t += [f'v_mov_b32 v{r}, 0' for r in ((12, 13, 16, 17) if variant == 4 else range(10, 18))]
    t += [f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
          f's_cbranch_execz .Lgz{tag}_%=']
    for i, j in enumerate(grad_groups(q)):
        p0, l0, p1, l1 = A(j), A(j | 1), A(j | (1 << q)), A(j | (1 << q) | 1)
        t += [f's_bitcmp1_b32 s{REC + 5}, {i}', f's_cbranch_scc0 .Lgg{tag}_{i}_%=']
        if variant == 1:
            t += [f'v_pk_fma_f32 {g_}, {l_}, {p_}, {g_}' for g_, l_, p_ in ((ACC[0], l0, p0), (ACC[1], l0, p1), (ACC[2], l1, p0), (ACC[3], l1, p1))]
        elif variant == 4:
            t += [f'v_pk_fma_f32 {ACC[1]}, {l0}, {p1}, {ACC[1]} {CROSS}', f'v_pk_fma_f32 {ACC[3]}, {l1}, {p0}, {ACC[3]} {CROSS}']
        elif variant == 2:
            t += [f'v_pk_fma_f32 {ACC[0]}, {l0}, {p0}, {ACC[0]}', f'v_pk_fma_f32 {ACC[1]}, {l0}, {p1}, {ACC[1]} {CROSS}',
                  f'v_pk_fma_f32 {ACC[2]}, {l1}, {p1}, {ACC[2]}', f'v_pk_fma_f32 {ACC[3]}, {l1}, {p0}, {ACC[3]} {CROSS}']
        else:
            t += [f'v_pk_fma_f32 {ACC[0]}, {l0}, {p0}, {ACC[0]}', f'v_pk_fma_f32 {ACC[1]}, {l0}, {p0}, {ACC[1]} {CROSS}',
                  f'v_pk_fma_f32 {ACC[2]}, {l1}, {p1}, {ACC[2]}', f'v_pk_fma_f32 {ACC[3]}, {l1}, {p1}, {ACC[3]} {CROSS}']
        t.append(f'.Lgg{tag}_{i}_%=:')
    t += [f'.Lgz{tag}_%=:', f's_mov_b64 exec, {SAVE}', f'v_mul_f32 {TT}, {HR}, {HR}', f'v_fma_f32 {TT}, {HI}, {HI}, {TT}']
    if variant == 4:       # one sum: butterflies over the four lane bits of a row, lane 0 of every row adds to Im G01
        dpp = 'row_mask:0xf bank_mask:0xf'
        return t + ['v_pk_add_f32 v[12:13], v[12:13], v[16:17]', 's_nop 0', 'v_sub_f32 v10, v13, v12',
                    's_nop 1', f'v_add_f32_dpp v10, v10, v10 row_ror:4 {dpp}',
                    's_nop 1', f'v_add_f32_dpp v10, v10, v10 row_ror:8 {dpp}',
                    's_nop 1', f'v_add_f32_dpp v10, v10, v10 quad_perm:[1,0,3,2] {dpp}',
                    's_nop 1', f'v_add_f32_dpp v10, v10, v10 quad_perm:[2,3,0,1] {dpp}',
                    's_nop 0', f'v_mul_f32 v10, v10, {TT}',
                    f'v_mov_b32 v32, {ACC_BASE - 32 + 12}', f'v_add_u32 v32, {GOFF}, v32',
                    's_mov_b32 exec_lo, 0x00010001', 's_mov_b32 exec_hi, 0x00010001',
                    'ds_add_f32 v32, v10',
                    's_mov_b64 exec, -1']
    if variant == 1:       # v10, v12, v14, v16 = Re G00, G01, G10, G11
        t += ['v_add_f32 v10, v10, v11', 'v_add_f32 v12, v12, v13', 'v_add_f32 v14, v14, v15', 'v_add_f32 v16, v16, v17']
        nval = 4
    elif variant == 2:     # v10 = Re (G00 + G11), v12 = Im (G01 + G10)
        t += ['v_pk_add_f32 v[10:11], v[10:11], v[14:15]', 'v_pk_add_f32 v[12:13], v[12:13], v[16:17]', 's_nop 0',
              'v_add_f32 v10, v10, v11', 'v_sub_f32 v12, v13, v12']
        nval = 2
    else:                  # v10, v12 = Re, Im G00; v14, v16 = Re, Im G11
        t += ['v_add_f32 v10, v10, v11', 'v_sub_f32 v12, v13, v12', 'v_add_f32 v14, v14, v15', 'v_sub_f32 v16, v17, v16']
        nval = 4
    dpp = 'row_mask:0xf'
    # lane bit 2: lanes with the bit clear take over the first sum of each pair, the others the second
    t += ['s_nop 1', f'v_add_f32_dpp v10, v10, v10 row_shl:4 {dpp} bank_mask:0x5']
    if nval == 4:
        t += [f'v_add_f32_dpp v14, v14, v14 row_shl:4 {dpp} bank_mask:0x5']
    t += [f'v_add_f32_dpp v10, v12, v12 row_shr:4 {dpp} bank_mask:0xa']
    if nval == 4:
        t += [f'v_add_f32_dpp v14, v16, v16 row_shr:4 {dpp} bank_mask:0xa',
              # lane bit 3: the lower half of a row keeps the first pair, the upper half the second
              's_nop 1', f'v_add_f32_dpp v10, v10, v10 row_shl:8 {dpp} bank_mask:0x3',
              's_nop 1', f'v_add_f32_dpp v10, v14, v14 row_shr:8 {dpp} bank_mask:0xc']
    else:
        t += ['s_nop 1', f'v_add_f32_dpp v10, v10, v10 row_ror:8 {dpp} bank_mask:0xf']
    t += ['s_nop 1', f'v_add_f32_dpp v10, v10, v10 quad_perm:[1,0,3,2] {dpp} bank_mask:0xf',
          's_nop 1', f'v_add_f32_dpp v10, v10, v10 quad_perm:[2,3,0,1] {dpp} bank_mask:0xf',
          's_nop 0', f'v_mul_f32 v10, v10, {TT}']
    # the lane with bits (b3, b2) holds sum number 2 b3 + b2 (variant 2: number b2); byte offset of its component
    if variant == 1:       # components 0, 2, 4, 6
        t += [f'v_bfe_u32 v32, {LANE}, 2, 2', 'v_lshlrev_b32 v32, 3, v32']
        ex = 0x11111111
    elif variant == 2:     # components 0, 3
        t += [f'v_bfe_u32 v32, {LANE}, 2, 1', 'v_mul_u32_u24 v32, 12, v32']
        ex = 0x00110011
    else:                  # components 0, 1, 6, 7
        t += [f'v_bfe_u32 v32, {LANE}, 2, 2', f'v_bfe_u32 v33, {LANE}, 3, 1', 'v_lshlrev_b32 v32, 2, v32', 'v_mad_u32_u24 v32, v33, 16, v32']
        ex = 0x11111111
    t += [f'v_add_u32 v32, {GOFF}, v32', f'v_add_u32 v32, {ACC_BASE - 32}, v32',
          f's_mov_b32 exec_lo, {hex(ex)}', f's_mov_b32 exec_hi, {hex(ex)}',
          'ds_add_f32 v32, v10',
          's_mov_b64 exec, -1']
    return t


G2ROW = [88, 40, 80, 72]         # SGPR base of matrix row r (eight dwords: four complex entries)


def far_next():
    """Back to the loop head from code that lies out of the reach of s_branch."""
    return ['s_sub_u32 vcc_lo, s54, .Ltable_%=-.Lnext_%=', 's_subb_u32 vcc_hi, s55, 0', 's_setpc_b64 vcc']


def gen2_groups(a, b):
    return [j for j in range(NA) if not (j >> a) & 1 and not (j >> b) & 1]


def gen2_body(a, b):
    """out[r] = sum_c M[r][c] in[c] on every group of four registers (r, c = 2 * bit b + bit a), groups by the mask."""
    out_ = []
    T = [T0, U0, T1, U1]
    for i, j in enumerate(gen2_groups(a, b)):
        rg = [A(j | (((r >> 1) & 1) << b) | ((r & 1) << a)) for r in range(4)]
        m = lambda r, c: f's[{G2ROW[r] + 2 * c}:{G2ROW[r] + 2 * c + 1}]'      # noqa: E731
        out_ += [f's_bitcmp1_b32 {STMP}, {i}', f's_cbranch_scc0 .Lg2s{a}{b}_{i}_%=']
        out_ += [f'v_pk_mul_f32 {T[r]}, {rg[0]}, {m(r, 0)} {RE2}' for r in range(4)]
        out_ += [f'v_pk_fma_f32 {T[r]}, {rg[0]}, {m(r, 0)}, {T[r]} {I3}' for r in range(4)]
        for c in range(1, 4):
            out_ += [f'v_pk_fma_f32 {T[r]}, {rg[c]}, {m(r, c)}, {T[r]} {RE3}' for r in range(4)]
            out_ += [f'v_pk_fma_f32 {T[r]}, {rg[c]}, {m(r, c)}, {T[r]} {I3}' for r in range(4)]
        out_ += [f'v_mov_b64 {rg[r]}, {T[r]}' for r in range(4)]
        out_.append(f'.Lg2s{a}{b}_{i}_%=:')
    return out_


def gen2_body_real(a, b):
    """`gen2_body` for a real matrix: out[r] = sum_c M[r][c].re in[c], one packed operation per entry (20 per group instead
    of 36).  No tests for exact zeros: a taken branch per zero entry and group costs more than the multiplication it
    saves (measured: the superoperator of a depolarizing channel, six non-zero entries of sixteen, ran no faster with
    them than the general body)."""
    out_ = []
    T = [T0, U0, T1, U1]
    for i, j in enumerate(gen2_groups(a, b)):
        rg = [A(j | (((r >> 1) & 1) << b) | ((r & 1) << a)) for r in range(4)]
        m = lambda r, c: f's[{G2ROW[r] + 2 * c}:{G2ROW[r] + 2 * c + 1}]'      # noqa: E731
        out_ += [f's_bitcmp1_b32 {STMP}, {i}', f's_cbranch_scc0 .Lg2r{a}{b}_{i}_%=']
        out_ += [f'v_pk_mul_f32 {T[r]}, {rg[0]}, {m(r, 0)} {RE2}' for r in range(4)]
        for c in range(1, 4):
            out_ += [f'v_pk_fma_f32 {T[r]}, {rg[c]}, {m(r, c)}, {T[r]} {RE3}' for r in range(4)]
        out_ += [f'v_mov_b64 {rg[r]}, {T[r]}' for r in range(4)]
        out_.append(f'.Lg2r{a}{b}_{i}_%=:')
    return out_


def gen2_body_xreal(a, b):
    """`gen2_body_real` for a matrix that is also X-shaped (index = 2 * bit b + bit a; entries (0,0) (0,3) (3,0) (3,3) and
    (1,1) (1,2) (2,1) (2,2) only): two independent real 2x2 products per group, the second output of each written in
    place -- eight packed operations and two moves.  (Not a test for zeros at run time -- `gen2_body_real` says why that
    does not pay -- but a body of its own, chosen by the host from the channel's CLASS.)"""
    out_ = []
    for i, j in enumerate(gen2_groups(a, b)):
        rg = [A(j | (((r >> 1) & 1) << b) | ((r & 1) << a)) for r in range(4)]
        m = lambda r, c: f's[{G2ROW[r] + 2 * c}:{G2ROW[r] + 2 * c + 1}]'      # noqa: E731
        out_ += [f's_bitcmp1_b32 {STMP}, {i}', f's_cbranch_scc0 .Lg2x{a}{b}_{i}_%=',
                 f'v_pk_mul_f32 {T0}, {rg[0]}, {m(0, 0)} {RE2}', f'v_pk_mul_f32 {T1}, {rg[0]}, {m(3, 0)} {RE2}',
                 f'v_pk_mul_f32 {U0}, {rg[1]}, {m(1, 1)} {RE2}', f'v_pk_mul_f32 {U1}, {rg[1]}, {m(2, 1)} {RE2}',
                 f'v_pk_fma_f32 {T0}, {rg[3]}, {m(0, 3)}, {T0} {RE3}', f'v_pk_fma_f32 {rg[3]}, {rg[3]}, {m(3, 3)}, {T1} {RE3}',
                 f'v_pk_fma_f32 {U0}, {rg[2]}, {m(1, 2)}, {U0} {RE3}', f'v_pk_fma_f32 {rg[2]}, {rg[2]}, {m(2, 2)}, {U1} {RE3}',
                 f'v_mov_b64 {rg[0]}, {T0}', f'v_mov_b64 {rg[1]}, {U0}',
                 f'.Lg2x{a}{b}_{i}_%=:']
    return out_


def gen2_body_xcplx(a, b):
    """`gen2_body` for an X-shaped matrix with complex entries: a complex 2x2 product on the registers (00, 11) and one on
    (01, 10) per group -- sixteen packed operations and four moves."""
    out_ = []
    T = [T0, U0, T1, U1]
    for i, j in enumerate(gen2_groups(a, b)):
        rg = [A(j | (((r >> 1) & 1) << b) | ((r & 1) << a)) for r in range(4)]
        m = lambda r, c: f's[{G2ROW[r] + 2 * c}:{G2ROW[r] + 2 * c + 1}]'      # noqa: E731
        out_ += [f's_bitcmp1_b32 {STMP}, {i}', f's_cbranch_scc0 .Lg2c{a}{b}_{i}_%=']
        first = {0: 0, 1: 1, 2: 1, 3: 0}       # output r reads the inputs r and 3 - r: the lower of the two first
        out_ += [f'v_pk_mul_f32 {T[r]}, {rg[first[r]]}, {m(r, first[r])} {RE2}' for r in range(4)]
        out_ += [f'v_pk_fma_f32 {T[r]}, {rg[first[r]]}, {m(r, first[r])}, {T[r]} {I3}' for r in range(4)]
        out_ += [f'v_pk_fma_f32 {T[r]}, {rg[3 - first[r]]}, {m(r, 3 - first[r])}, {T[r]} {RE3}' for r in range(4)]
        out_ += [f'v_pk_fma_f32 {T[r]}, {rg[3 - first[r]]}, {m(r, 3 - first[r])}, {T[r]} {I3}' for r in range(4)]
        out_ += [f'v_mov_b64 {rg[r]}, {T[r]}' for r in range(4)]
        out_.append(f'.Lg2c{a}{b}_{i}_%=:')
    return out_


def gen2_code():
    """Entry of every two-target dense record: controls; group mask -> STMP, first-target-on-the-lower-slot flag -> s70,
    pair number -> s71; rows 1 .. 3 of the 4x4 matrix (the look-ahead fetched row 0 as "the matrix") into s[40:47],
    s[80:87], s[72:79] -- the record itself and both look-ahead registers, fetched again afterwards; with the flag the
    two matrix index bits trade places (rows 1 <-> 2 by loading them crosswise, columns 1 <-> 2 by twelve s_mov); then
    the body of the slot pair through a second jump table (computed: the bodies lie out of the reach of s_branch)."""
    t = ['.Lgen2_%=:', 's_waitcnt lgkmcnt(0)',
         f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc1 .Lg2in_%='] + far_next()
    t += ['.Lg2in_%=:', f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
          's_cbranch_execnz .Lg2go_%=', f's_mov_b64 exec, {SAVE}'] + far_next()
    t += ['.Lg2go_%=:', f's_mov_b32 {STMP}, s{REC + 5}', f's_mov_b32 s70, s{REC + 6}', f's_sub_u32 s71, s{REC}, {ID_GEN2}',
          f's_sub_u32 s69, {MOFF}, 128', 's_cmp_eq_u32 s70, 0', 's_cbranch_scc0 .Lg2x_%=',
          f's_add_u32 s69, s69, 32', f's_load_dwordx8 s[40:47], {MB}, s69', f's_add_u32 s69, s69, 32', f's_load_dwordx8 s[80:87], {MB}, s69',
          's_branch .Lg2r3_%=', '.Lg2x_%=:',
          f's_add_u32 s69, s69, 32', f's_load_dwordx8 s[80:87], {MB}, s69', f's_add_u32 s69, s69, 32', f's_load_dwordx8 s[40:47], {MB}, s69',
          '.Lg2r3_%=:', f's_add_u32 s69, s69, 32', f's_load_dwordx8 s[72:79], {MB}, s69', 's_waitcnt lgkmcnt(0)',
          's_cmp_eq_u32 s70, 0', 's_cbranch_scc1 .Lg2j_%=']
    for r in range(4):
        b0 = G2ROW[r]
        t += [f's_mov_b64 vcc, s[{b0 + 2}:{b0 + 3}]', f's_mov_b64 s[{b0 + 2}:{b0 + 3}], s[{b0 + 4}:{b0 + 5}]', f's_mov_b64 s[{b0 + 4}:{b0 + 5}], vcc']
    t += ['.Lg2j_%=:', 's_getpc_b64 vcc', '.Lg2anchor_%=:', 's_lshl3_add_u32 vcc_lo, s71, vcc_lo', 's_addc_u32 vcc_hi, vcc_hi, 0',
          's_add_u32 vcc_lo, vcc_lo, .Lg2table_%=-.Lg2anchor_%=', 's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc', '.Lg2table_%=:']
    # eight bytes per entry: s_getpc + 64-bit add + s_setpc would not fit four; a long jump is s_getpc_b64 / s_add / s_setpc:
    # instead every entry is an s_branch to a trampoline that sits right behind the table, within reach of nothing but it
    for v in range(60):
        t += [f's_branch .Lg2t{v}_%=', 's_nop 0']
    for v in range(60):
        t += [f'.Lg2t{v}_%=:', 's_getpc_b64 vcc', f'.Lg2ta{v}_%=:', f's_sub_u32 vcc_lo, vcc_lo, .Lg2ta{v}_%=-.Lg2b{v}_%=',
              's_subb_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc']          # (the bodies lie in front of everything)
    return t


def gen2_bodies():
    t = []
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{v}_%=:'] + gen2_body(a, b) + [f's_mov_b64 exec, {SAVE}']
        t += prefetch(f'g2{v}') + far_next()
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{15 + v}_%=:'] + gen2_body_real(a, b) + [f's_mov_b64 exec, {SAVE}']
        t += prefetch(f'g2r{v}') + far_next()
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{30 + v}_%=:'] + gen2_body_xreal(a, b) + [f's_mov_b64 exec, {SAVE}']
        t += prefetch(f'g2x{v}') + far_next()
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{45 + v}_%=:'] + gen2_body_xcplx(a, b) + [f's_mov_b64 exec, {SAVE}']
        t += prefetch(f'g2c{v}') + far_next()
    return t


def expz_code():
    """DQ_FG_EXPZ: sum_i (-1)^popc(i & zmask) |a_i|^2 over the tile, added to component 0 of the record's accumulator.
    The parity of an amplitude splits into its register's (w5 / w7: bit j = sign of register j, made by the translator
    from the Z bits that are register slots), its lane's (w1 against the lane's tile-local index) and the tile's (w2:w3
    against the index bits outside the tile).  The registers lack the pass's deferred factor f: times |f|^2."""
    P, M_ = 'v[10:11]', 'v[12:13]'
    t = [f'v_mov_b32 v{r}, 0' for r in range(10, 14)]
    for j in range(NA):
        word = REC + 5 if j < 32 else REC + 7
        t += [f's_bitcmp1_b32 s{word}, {j % 32}', f's_cbranch_scc1 .Lezm{j}_%=',
              f'v_pk_fma_f32 {P}, {A(j)}, {A(j)}, {P}', f's_branch .Lezn{j}_%=',
              f'.Lezm{j}_%=:', f'v_pk_fma_f32 {M_}, {A(j)}, {A(j)}, {M_}', f'.Lezn{j}_%=:']
    t += ['v_sub_f32 v10, v10, v12', 'v_sub_f32 v11, v11, v13', 'v_add_f32 v10, v10, v11',
          # sign of the lane and of the tile
          f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_bcnt_u32_b32 {TT}, {TT}, 0',
          f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_bcnt1_i32_b64 {STMP}, vcc',
          f'v_add_u32 {TT}, {STMP}, {TT}', f'v_lshlrev_b32 {TT}, 31, {TT}', f'v_xor_b32 v10, {TT}, v10',
          f'v_mul_f32 {TT}, {HR}, {HR}', f'v_fma_f32 {TT}, {HI}, {HI}, {TT}', f'v_mul_f32 v10, v10, {TT}',
          f's_add_u32 {STMP}, {GOFF}, {ACC_BASE - 32}', f'v_mov_b32 v32, {STMP}',
          'ds_add_f32 v32, v10']
    return t


def gray_walk(op, base_operand, lane_operand, nt=False):
    """32 x (address = base + running slot offset + lane offset; op).  The running offset follows a Gray code over the
    slot bits 1..5, so each step is one 64-bit scalar add or subtract."""
    out_ = [f's_mov_b64 {RUN}, {base_operand}']
    for i in range(32):
        g = i ^ (i >> 1)
        if i:
            b = (i & -i).bit_length() - 1
            lo, hi = f's{40 + 2 * b}', f's{41 + 2 * b}'
            if (g >> b) & 1:
                out_ += [f's_add_u32 s50, s50, {lo}', f's_addc_u32 s51, s51, {hi}']
            else:
                out_ += [f's_sub_u32 s50, s50, {lo}', f's_subb_u32 s51, s51, {hi}']
        ad = ADDR[i % 4]
        out_.append(f'v_lshl_add_u64 {ad}, {RUN}, 0, {lane_operand}')
        regs = f'v[{AMP0 + 4 * g}:{AMP0 + 4 * g + 3}]'
        out_.append((f'global_load_dwordx4 {regs}, {ad}, off' if op == 'load' else f'global_store_dwordx4 {ad}, {regs}, off') + (' nt' if nt else ''))
    return out_


def zext_load(base_operand, lane_operand):
    """The loads of a pass whose input has index bits KNOWN TO BE |0> (dq_apply_fused_zext_c64: the circuit's own |0..0>
    and the passes right behind it).  flags bits 8..13 / 16..21 = the register slots / lane bits of the load layout
    that hold such bits: where one of them is 1 nothing is read -- the memory there is not even initialised -- and the
    registers are zero.  Falls through to the ordinary loads when no such bit is in the tile."""
    t = ['s_bfe_u32 s69, %[flags], 0x000e0008', 's_cmp_eq_u32 s69, 0', 's_cbranch_scc1 .Lldn_%=']
    t += [f'v_mov_b64 {A(j)}, 0' for j in range(NA)]
    t += ['s_lshr_b32 s70, s69, 8', 's_and_b32 s69, s69, 0x3f', f'v_and_b32 {TT}, s70, {LANE}', f'v_cmp_eq_u32 vcc, 0, {TT}',
          f's_and_saveexec_b64 {SAVE}, vcc', f's_mov_b64 {RUN}, {base_operand}']
    for i in range(32):
        g = i ^ (i >> 1)
        if i:
            b = (i & -i).bit_length() - 1
            lo, hi = f's{40 + 2 * b}', f's{41 + 2 * b}'
            if (g >> b) & 1:
                t += [f's_add_u32 s50, s50, {lo}', f's_addc_u32 s51, s51, {hi}']
            else:
                t += [f's_sub_u32 s50, s50, {lo}', f's_subb_u32 s51, s51, {hi}']
        if g:       # (the walk covers slots 1..5; slot 0 = index bit 0 inside the 16-byte piece, never such a bit)
            t += [f's_and_b32 s70, s69, {g << 1}', f's_cbranch_scc1 .Lzx{i}_%=']
        ad = ADDR[i % 4]
        t += [f'v_lshl_add_u64 {ad}, {RUN}, 0, {lane_operand}', f'global_load_dwordx4 v[{AMP0 + 4 * g}:{AMP0 + 4 * g + 3}], {ad}, off']
        if g:
            t.append(f'.Lzx{i}_%=:')
    return t + [f's_mov_b64 exec, {SAVE}', 's_branch .Lldd_%=', '.Lldn_%=:']


def kernel_body():
    h = handlers()
    ids = sorted(h)
    half = len(ids) // 2
    # split the handlers around the loop head so that every s_branch stays within its 16-bit reach: the trips (the
    # bulk of the code) go behind the table, the gates in front of it
    front = [i for i in ids if i < ID_TRIP0]
    back = [i for i in ids if i >= ID_TRIP0]
    nxt = ['s_branch .Lnext_%=']

    def emit(i):
        ctl, lines = h[i]
        out_ = [f'.Lh{i}_%=:']
        if ctl:
            out_ += [f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]',
                     's_cbranch_scc0 .Lnext_%=',
                     f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}',
                     f's_and_saveexec_b64 {SAVE}, vcc', 's_cbranch_execz .Lrestore_%=']
        out_ += lines
        if ctl:
            out_.append(f's_mov_b64 exec, {SAVE}')
        return out_ + nxt

    # the bodies of the two-target dense gates first (74 KB, jumped over; reached and left by computed jumps): behind
    # everything else they would push the last labels out of the reach of the store walks' s_branch
    text = ['s_getpc_b64 vcc', '.Ljs_%=:', 's_add_u32 vcc_lo, vcc_lo, .Lstart_%=-.Ljs_%=', 's_addc_u32 vcc_hi, vcc_hi, 0',
            's_setpc_b64 vcc'] + gen2_bodies() + ['.Lstart_%=:']      # (a computed jump: the bodies exceed the reach of s_branch)
    text += [f's_mov_b64 {KG}, %[kg]', f's_mov_b32 {GOFF}, 0', f's_mov_b32 {GEND}, %[gend]', f's_mov_b64 {MB}, %[mb]',
            f's_mov_b32 {MOFF}, %[moff]', f's_mov_b64 {TG}, %[tg]', f's_mov_b32 {LDSB}, %[ldsb]',
            # slot offsets of the load layout; byte shifts of the lane bits (load, store) and what they add to the
            # thread's tile-local base (WaveKernPass::load_off .. tb_contrib)
            f's_load_dwordx8 s[40:47], %[ks], 0', f's_load_dwordx2 s[48:49], %[ks], 32',
            f's_load_dwordx8 s[{REC}:{REC + 7}], %[ks], 80', f's_load_dwordx8 s[{REC2}:{REC2 + 7}], %[ks], 112',
            f's_load_dwordx2 s[{MAT}:{MAT + 1}], %[ks], 144',
            f'v_and_b32 {LANE}, 63, %[tid]', f'v_mov_b32 {HR}, 1.0', f'v_mov_b32 {HI}, 0',
            's_mov_b32 s56, 0xc0000000', 's_mov_b32 s57, 0xc0000000']
    text += [f'v_bfe_i32 {LB[b]}, {LANE}, {b}, 1' for b in range(6)]          # (sign-extended: 0 or all ones)
    text += ['v_mov_b32 v2, 0', 'v_mov_b32 v3, 0', 'v_mov_b32 v4, 0', 'v_mov_b32 v5, 0', 's_waitcnt lgkmcnt(0)']
    text += [f'v_and_b32 {TB}, s{REC + 12}, {LB[0]}'] + [f'v_and_or_b32 {TB}, {LB[b]}, s{REC + 12 + b}, {TB}' for b in range(1, 6)]
    for b in range(6):
        for lo, hi, sh in (('v2', 'v3', REC + b), ('v4', 'v5', REC + 6 + b)):
            text += [f'v_and_b32 v32, 1, {LB[b]}', 'v_mov_b32 v33, 0', f'v_lshlrev_b64 v[32:33], s{sh}, v[32:33]',
                     f'v_or_b32 {lo}, {lo}, v32', f'v_or_b32 {hi}, {hi}, v33']
    # streaming (non-temporal) loads and stores when the host says so (flags bit 0 / 1: states far bigger than the
    # caches; +10-15 % on the memory side, tools/experiments/mb_wavetile.hip), plain ones otherwise (small states live in
    # the Infinity Cache between passes; the pass that reads ONE shared input state relies on the L2)
    text += zext_load('%[inb]', LLD)
    text += ['s_bitcmp1_b32 %[flags], 0', 's_cbranch_scc0 .Lldp_%='] + gray_walk('load', '%[inb]', LLD, nt=True) + ['s_branch .Lldd_%=', '.Lldp_%=:']
    text += gray_walk('load', '%[inb]', LLD) + ['.Lldd_%=:']
    # the first record and its matrix arrive with the tile
    text += prefetch('first')
    text += [f's_getpc_b64 {TABLE}', '.Lanchor_%=:', 's_add_u32 s54, s54, .Ltable_%=-.Lanchor_%=', 's_addc_u32 s55, s55, 0',
             's_waitcnt vmcnt(0)', 's_branch .Lnext_%=']
    for i in front:
        text += emit(i)
    # loop head: the look-ahead record becomes the current one, the one behind it is requested, then the handler
    text += ['.Lrestore_%=:', f's_mov_b64 exec, {SAVE}', '.Lnext_%=:',
             f's_cmp_lt_u32 {GOFF}, {GEND}', 's_cbranch_scc0 .Lexit_%=', 's_waitcnt lgkmcnt(0)']
    text += [f's_mov_b64 s[{REC + 2 * i}:{REC + 2 * i + 1}], s[{REC2 + 2 * i}:{REC2 + 2 * i + 1}]' for i in range(4)]
    text += [f's_mov_b64 s[{MAT + 2 * i}:{MAT + 2 * i + 1}], s[{NMAT + 2 * i}:{NMAT + 2 * i + 1}]' for i in range(4)]
    text += [f's_add_u32 {GOFF}, {GOFF}, 32',
             f's_lshl3_add_u32 {MOFF}, s{REC + 4}, {MOFF}']
    text += prefetch('loop')
    text += [f's_lshl2_add_u32 vcc_lo, s{REC}, s54', 's_addc_u32 vcc_hi, s55, 0', 's_setpc_b64 vcc',
             '.Ltable_%=:']
    for i in range(NIDS):
        text.append(f's_branch .Lh{i}_%=' if i in h else ('s_branch .Lgen2_%=' if i >= ID_GEN2 else
                                                         's_branch .Ldiag_%=' if i >= ID_DIAG1 else 's_branch .Lnext_%='))
    # ---- epilogue: the pass's deferred factor, then the stores ----
    text += ['.Lexit_%=:', 's_load_dwordx8 s[40:47], %[ks], 40', 's_load_dwordx2 s[48:49], %[ks], 72',
             's_waitcnt lgkmcnt(0)',          # (the slot offsets -- and the LDS reads of a trip that ended the pass)
             f'v_readfirstlane_b32 {STMP}, {HI}', f'v_mov_b32 v10, {HR}', f'v_mov_b32 v11, {HR}',
             f's_cmp_eq_u32 {STMP}, 0', 's_cbranch_scc0 .Lcplx_%=']
    text += [f'v_pk_mul_f32 {A(j)}, {A(j)}, v[10:11]' for j in range(NA)]
    text += ['s_branch .Lstore_%=', '.Lcplx_%=:', f'v_mov_b32 v12, {HI}', f'v_mov_b32 v13, {HI}']
    for j in range(NA):
        t = 'v[14:15]' if j % 2 == 0 else 'v[16:17]'
        text += [f'v_pk_mul_f32 {t}, {A(j)}, v[12:13] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]',
                 f'v_pk_fma_f32 {A(j)}, {A(j)}, v[10:11], {t}']
    text += ['.Lstore_%=:']
    text += ['s_bitcmp1_b32 %[flags], 1', 's_cbranch_scc0 .Lstp_%='] + gray_walk('store', '%[outb]', LST, nt=True) + ['s_branch .Ldone_%=', '.Lstp_%=:']
    text += gray_walk('store', '%[outb]', LST)
    text += ['s_branch .Ldone_%=']
    text += gen2_code()          # (the entry: within reach of the table; the bodies at the far end, reached by computed jumps)
    text += diag_code()
    for i in back:
        text += emit(i)
    text += ['.Ldone_%=:']
    return text


out = ['// GENERATED by tools/gen_wave_asm.py -- do not edit by hand.', '// clang-format off',
       f'#define DQ_WAVE_NIDS {NIDS}', f'#define DQ_WAVE_MAXK {MAXK}',
       f'#define DQ_WID_GEN_U {ID_GEN_U}', f'#define DQ_WID_GEN_C {ID_GEN_C}', f'#define DQ_WID_GEN_R {ID_GEN_R}',
       f'#define DQ_WID_X_U {ID_X_U}', f'#define DQ_WID_X_C {ID_X_C}', f'#define DQ_WID_X_R {ID_X_R}', f'#define DQ_WID_X_R1 {ID_X_R1}',
       f'#define DQ_WID_TRIP0 {ID_TRIP0}', f'#define DQ_WID_TRIP {ID_TRIP}', f'#define DQ_WID_SWAP {ID_SWAP}',
       f'#define DQ_WID_DIAG1 {ID_DIAG1}', f'#define DQ_WID_DIAG2 {ID_DIAG2}', f'#define DQ_WID_GRAD {ID_GRAD}', f'#define DQ_WAVE_GRAD_VARIANTS {GRAD_VARIANTS}', f'#define DQ_WID_EXPZ {ID_EXPZ}', f'#define DQ_WID_GEN2 {ID_GEN2}', f'#define DQ_WID_GEN2R {ID_GEN2R}', f'#define DQ_WID_GEN2X {ID_GEN2X}', f'#define DQ_WID_GEN2XC {ID_GEN2XC}', f'#define DQ_WAVE_ACC_BASE {ACC_BASE}',
       '// trip handler id by slot mask (popcount 1..DQ_WAVE_MAXK), -1 otherwise; slot-swap handler id by (i < j)',
       'static const short kWaveTripId[64] = {' + ', '.join(str(ID_TRIP + TRIP_MASKS.index(m)) if m in TRIP_MASKS else '-1' for m in range(64)) + '};',
       'static const short kWaveSwapId[6][6] = {' + ', '.join('{' + ', '.join(str(ID_SWAP + SWAP_PAIRS.index((min(i, j), max(i, j)))) if i != j else '-1' for j in range(R)) + '}' for i in range(R)) + '};',
       '']
text = '\\n\\t"\n        "'.join(kernel_body())
clob = ', '.join(['"m0"'] + [f'"s{i}"' for i in range(40, 100)] + [f'"v{i}"' for i in range(1, AMP0 + 2 * NA)])
out += ['// kg = address of the records, gend = their size in bytes; mb + moff = address of the first matrix; tg = the index',
        '// bits this tile fixes; ks = address of WaveKernPass::load_off (slot offsets, lane shifts); inb / outb = tile bases;',
        '// ldsb = the wave\'s LDS region; tid = threadIdx.x',
        '__device__ __forceinline__ void wave_tile_body_f32(uint64_t kg, uint32_t gend, uint64_t mb, uint32_t moff, uint64_t tg,',
        '                                                   uint64_t ks, uint64_t inb, uint64_t outb, uint32_t ldsb, uint32_t tid, uint32_t flags) {',
        f'    asm volatile(\n        "{text}"',
        '        :',
        '        : [kg] "s"(kg), [gend] "s"(gend), [mb] "s"(mb), [moff] "s"(moff), [tg] "s"(tg), [ks] "s"(ks), [inb] "s"(inb),',
        '          [outb] "s"(outb), [ldsb] "s"(ldsb), [tid] "v"(tid), [flags] "s"(flags)',
        f'        : "vcc", "scc", "memory", {clob});',
        '}', '// clang-format on', '']
path = os.environ.get('DQ_ASM_OUT') or os.path.join(os.path.dirname(__file__), '..', 'deepquantum_amd', 'csrc', 'dq_wave_asm.inc')
open(path, 'w').write('\n'.join(out))
print('generated', len(out), 'lines;', NIDS, 'handler ids;', sum(len(v[1]) for v in handlers().values()), 'handler instructions')
