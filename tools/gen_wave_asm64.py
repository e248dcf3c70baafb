"""Generate csrc/dq_wave_asm64.inc: the body of the wave-tile pass kernel for complex128.

Same design as tools/gen_wave_asm.py (one wavefront per tile, no workgroup barrier, layout changes through a small
wave-private LDS buffer), with 16-byte amplitudes: 64 lanes x 32 amplitudes = an 11-bit tile in v[40:167] (amplitude j:
re = v[40 + 4j : 41 + 4j], im = v[42 + 4j : 43 + 4j]), five register-slot bits, v_fma_f64 bodies, ds_*_b128 trips of up
to three slots (8.6 KiB per wave).  Record format and handler families as in the complex64 generator; the matrix of a
gate is 16 dwords (s[80:95]), the Hadamard factor is deferred (a real f64), Rx-like gates keep their plain matrix.
"""
import os

R = 5
NA = 1 << R
AMP0 = 40
TMP = ['v[10:11]', 'v[12:13]', 'v[14:15]', 'v[16:17]']
TT, HS, TB, LANE, WB, RB = 'v18', 'v[20:21]', 'v22', 'v23', 'v24', 'v25'
LB = [f'v{26 + b}' for b in range(6)]            # the lane's bits as masks: 0 / 0xffffffff
ADDR = ['v[32:33]', 'v[34:35]', 'v[36:37]', 'v[38:39]']
LLD, LST = 'v[2:3]', 'v[4:5]'
RUN, SAVE, TABLE = 's[50:51]', 's[52:53]', 's[54:55]'
GOFF, GEND, MOFF, STMP = 's58', 's59', 's60', 's61'
MB, KG, TG, LDSB = 's[62:63]', 's[64:65]', 's[66:67]', 's68'
REC, REC2, MAT = 72, 80, 80
MNAMES = ['00r', '00i', '01r', '01i', '10r', '10i', '11r', '11i']
M = {nm: f's[{MAT + 2 * i}:{MAT + 2 * i + 1}]' for i, nm in enumerate(MNAMES)}
ELEM = 16


def RE(j):
    return f'v[{AMP0 + 4 * j}:{AMP0 + 4 * j + 1}]'


def IM(j):
    return f'v[{AMP0 + 4 * j + 2}:{AMP0 + 4 * j + 3}]'


def A(j):
    return f'v[{AMP0 + 4 * j}:{AMP0 + 4 * j + 3}]'


def pairs(q, cmask=0):
    return [(j, j | (1 << q)) for j in range(NA) if not (j >> q) & 1 and (j & cmask) == cmask]


def pair_body(mode, lo, hi):
    ar, ai, br, bi = RE(lo), IM(lo), RE(hi), IM(hi)
    t = TMP
    if mode == 3:      # s [[1, 1], [1, -1]], s deferred
        return [f'v_add_f64 {ar}, {ar}, {br}', f'v_add_f64 {ai}, {ai}, {bi}',
                f'v_fma_f64 {br}, -2.0, {br}, {ar}', f'v_fma_f64 {bi}, -2.0, {bi}, {ai}']
    last = [f'v_fma_f64 {ar}, {M["00r"]}, {ar}, {t[0]}', f'v_fma_f64 {ai}, {M["00r"]}, {ai}, {t[1]}',
            f'v_fma_f64 {br}, {M["11r"]}, {br}, {t[2]}', f'v_fma_f64 {bi}, {M["11r"]}, {bi}, {t[3]}']
    if mode == 1:      # real matrix
        return [f'v_mul_f64 {t[0]}, {M["01r"]}, {br}', f'v_mul_f64 {t[1]}, {M["01r"]}, {bi}',
                f'v_mul_f64 {t[2]}, {M["10r"]}, {ar}', f'v_mul_f64 {t[3]}, {M["10r"]}, {ai}'] + last
    if mode == 2:      # real diagonal, imaginary off-diagonal: (i s)(x + i y) = -s y + i s x
        return [f'v_mul_f64 {t[0]}, -{M["01i"]}, {bi}', f'v_mul_f64 {t[1]}, {M["01i"]}, {br}',
                f'v_mul_f64 {t[2]}, -{M["10i"]}, {ai}', f'v_mul_f64 {t[3]}, {M["10i"]}, {ar}'] + last
    return [f'v_mul_f64 {t[0]}, -{M["00i"]}, {ai}', f'v_mul_f64 {t[1]}, {M["00i"]}, {ar}',
            f'v_mul_f64 {t[2]}, {M["10r"]}, {ar}', f'v_mul_f64 {t[3]}, {M["10r"]}, {ai}',
            f'v_fma_f64 {t[0]}, {M["01r"]}, {br}, {t[0]}', f'v_fma_f64 {t[1]}, {M["01r"]}, {bi}, {t[1]}',
            f'v_fma_f64 {t[2]}, -{M["10i"]}, {ai}, {t[2]}', f'v_fma_f64 {t[3]}, {M["10i"]}, {ar}, {t[3]}',
            f'v_fma_f64 {t[0]}, -{M["01i"]}, {bi}, {t[0]}', f'v_fma_f64 {t[1]}, {M["01i"]}, {br}, {t[1]}',
            f'v_fma_f64 {t[2]}, -{M["11i"]}, {bi}, {t[2]}', f'v_fma_f64 {t[3]}, {M["11i"]}, {br}, {t[3]}'] + last


def body(mode, q):
    out_ = []
    for lo, hi in pairs(q):
        out_ += pair_body(mode, lo, hi)
    return out_


def swap_pair(lo, hi):
    t = TMP
    return [f'v_mov_b64 {t[0]}, {RE(lo)}', f'v_mov_b64 {t[1]}, {IM(lo)}', f'v_mov_b64 {RE(lo)}, {RE(hi)}', f'v_mov_b64 {IM(lo)}, {IM(hi)}',
            f'v_mov_b64 {RE(hi)}, {t[0]}', f'v_mov_b64 {IM(hi)}, {t[1]}']


def xlines(q, cmask=0):
    out_ = []
    for lo, hi in pairs(q, cmask):
        out_ += swap_pair(lo, hi)
    return out_


def masked(q, tag, per_pair):
    out_ = []
    for i, (lo, hi) in enumerate(pairs(q)):
        out_ += [f's_bitcmp1_b32 s{REC + 5}, {i}', f's_cbranch_scc0 .Lm{tag}_{i}_%='] + per_pair(lo, hi) + [f'.Lm{tag}_{i}_%=:']
    return out_


def slotswap(i, j):
    out_ = []
    for r in range(NA):
        if (r >> i) & 1 and not (r >> j) & 1:
            out_ += swap_pair(r, r ^ (1 << i) ^ (1 << j))
    return out_


def deposit(val, positions):
    return sum(((val >> i) & 1) << p for i, p in enumerate(positions))


def trip(k, mask):
    """As in the complex64 generator, 16-byte elements: slot (16-byte units) of element (x, y, z) = x * S + (y << a) + F(z),
    S = 64 + 2^a, a = 5 - k."""
    pre = [f's_load_dwordx8 s[{REC2}:{REC2 + 7}], {KG}, {GOFF}', f's_add_u32 {GOFF}, {GOFF}, 32']
    tbw = [REC + 1, REC + 2, REC + 3, REC + 5, REC + 6, REC + 7]
    pre += [f'v_and_b32 {TB}, s{tbw[0]}, {LB[0]}'] + [f'v_and_or_b32 {TB}, {LB[b]}, s{tbw[b]}, {TB}' for b in range(1, 6)]
    pre += ['s_waitcnt lgkmcnt(0)']
    pre += [f'v_and_b32 {WB}, s{REC2}, {LB[0]}'] + [f'v_and_or_b32 {WB}, {LB[b]}, s{REC2 + b}, {WB}' for b in range(1, 6)]
    pre += [f'v_lshrrev_b32 {RB}, 16, {WB}', f'v_and_b32 {WB}, 0xffff, {WB}', f'v_add_u32 {WB}, {LDSB}, {WB}', f'v_add_u32 {RB}, {LDSB}, {RB}']
    if k == 0:
        body_ = []
        for j in range(NA):
            body_ += [f'ds_write_b128 {WB}, {A(j)}', f'ds_read_b128 {A(j)}, {RB}']
        return pre + body_
    a = 5 - k
    S = 64 + (1 << a)
    moving = [s for s in range(R) if (mask >> s) & 1]
    staying = [s for s in range(R) if not (mask >> s) & 1]
    body_ = []
    for g in range(1 << (R - k)):
        base = deposit(g, staying)
        for x in range(1 << k):
            body_.append(f'ds_write_b128 {WB}, {A(base | deposit(x, moving))} offset:{ELEM * S * x}')
        for y in range(1 << k):
            body_.append(f'ds_read_b128 {A(base | deposit(y, moving))}, {RB} offset:{ELEM * (y << a)}')
    return pre + body_


MAXK = 3
TRIP_MASKS = [m for m in range(1, NA) if bin(m).count('1') <= MAXK]
SWAP_PAIRS = [(i, j) for i in range(R) for j in range(i + 1, R)]
NV = 2 * (R + 1)      # diag variants: 0 all, 1 + q by slot q, then the same with a register mask

ID_GEN_U = 0          # + 5 * mode + q
ID_GEN_C = 4 * R      # + q
ID_GEN_R = ID_GEN_C + R
ID_X_U = ID_GEN_R + R
ID_X_C = ID_X_U + R
ID_X_R = ID_X_C + R
ID_X_R1 = ID_X_R + R  # + 4 * q + c'
ID_TRIP0 = ID_X_R1 + R * (R - 1)
ID_TRIP = ID_TRIP0 + 1
ID_SWAP = ID_TRIP + len(TRIP_MASKS)
ID_DIAG1 = ID_SWAP + len(SWAP_PAIRS)
ID_DIAG2 = ID_DIAG1 + NV
# reduction of the adjoint method's reverse sweep: target on slot 1 + (id - ID_GRAD), psi / lambda told apart by slot 0
ID_GRAD = ID_DIAG2 + NV        # + (R - 1) * variant + (slot - 1); variants as in the complex64 generator
GRAD_VARIANTS = 5
ID_EXPZ = ID_GRAD + (R - 1) * GRAD_VARIANTS      # expectation value of a Z string (DQ_FG_EXPZ)
# dense gate on two register slots a < b (index in SWAP_PAIRS): a 4x4 matrix, index = 2 * (bit of slot b) + (bit of slot a)
ID_GEN2 = ID_EXPZ + 1
ID_GEN2R = ID_GEN2 + len(SWAP_PAIRS)      # ... for a matrix promised real (channel superoperators): half the operations
NIDS = ID_GEN2R + len(SWAP_PAIRS)
ACC_BASE = 4 * 8704       # LDS offset of the reduction accumulators (8 doubles per record): behind the staging buffers


def handlers():
    h = {}
    for q in range(R):
        for mode in (0, 1, 2):
            h[ID_GEN_U + R * mode + q] = (False, body(mode, q))
        h[ID_GEN_U + R * 3 + q] = (False, [f'v_mul_f64 {HS}, {HS}, {M["00r"]}'] + body(3, q))
        h[ID_GEN_C + q] = (True, body(0, q))
        h[ID_GEN_R + q] = (True, masked(q, f'g{q}', lambda lo, hi: pair_body(0, lo, hi)))
        h[ID_X_U + q] = (False, xlines(q))
        h[ID_X_C + q] = (True, xlines(q))
        h[ID_X_R + q] = (True, masked(q, f'x{q}', swap_pair))
        for c in range(R):
            if c != q:
                h[ID_X_R1 + (R - 1) * q + (c if c < q else c - 1)] = (True, xlines(q, 1 << c))
    h[ID_TRIP0] = (False, trip(0, 0))
    for i, m in enumerate(TRIP_MASKS):
        h[ID_TRIP + i] = (False, trip(bin(m).count('1'), m))
    for i, (a_, b_) in enumerate(SWAP_PAIRS):
        h[ID_SWAP + i] = (False, slotswap(a_, b_))
    for q in range(1, R):
        for v in range(GRAD_VARIANTS):
            h[ID_GRAD + (R - 1) * v + q - 1] = (False, grad_code(q, v))
    h[ID_EXPZ] = (False, expz_code())
    return h


PH0, PH1 = ('v[6:7]', 'v[8:9]'), ('v[32:33]', 'v[34:35]')        # (re, im) as f64 pairs


# ---- dense gates on two targets (DQ_FG_GEN2): a 4x4 complex128 matrix is 64 dwords -- more than the scalar registers the
# loop can spare -- so it passes through them two rows at a time.  Four register groups per batch: rows 0, 1 resident ->
# outputs 0, 1 of each group, parked in the wave's staging buffer (the inputs must stay whole for the other rows); rows
# 2, 3 resident -> outputs 2, 3 into their registers, outputs 0, 1 fetched back.  Two batches: four scalar-load phases.
G2A = [80, 84, 88, 92]         # SGPR base of entry c of the first resident row (re = +0:+1, im = +2:+3)
G2B = [40, 44, 72, 76]         # ... of the second one: s[40:47] and s[72:79] (the record's words are copied out first)
G2MASK, G2MOFF, G2FLAG, G2PAIR, G2OFFA, G2OFFB = STMP, 's69', 's70', 's71', 's56', 's57'
G2PARK = 'v8'                  # the lane's parking address: LDSB + 16 * lane (+ 1 KiB per parked amplitude)


def far_next():
    """Back to the loop head from code that lies out of the reach of s_branch."""
    return ['s_sub_u32 vcc_lo, s54, .Ltable_%=-.Lnext_%=', 's_subb_u32 vcc_hi, s55, 0', 's_setpc_b64 vcc']


def gen2_groups(a, b):
    return [j for j in range(NA) if not (j >> a) & 1 and not (j >> b) & 1]


def g2_load_rows(off_first, off_second):
    """Rows at byte offsets (SGPR or literal) `off_first`, `off_second` of this record's matrix -> s[80:95], s[40:47] +
    s[72:79]; with the swap flag the matrix index bits trade places: columns 1 <-> 2 of both rows."""
    t = [f's_add_u32 vcc_lo, {G2MOFF}, {off_first}', f's_load_dwordx16 s[80:95], {MB}, vcc_lo',
         f's_add_u32 vcc_lo, {G2MOFF}, {off_second}', f's_load_dwordx8 s[40:47], {MB}, vcc_lo',
         's_add_u32 vcc_lo, vcc_lo, 32', f's_load_dwordx8 s[72:79], {MB}, vcc_lo', 's_waitcnt lgkmcnt(0)',
         f's_cmp_eq_u32 {G2FLAG}, 0', 's_cbranch_scc1 .Lg2ns%=_{n}'.replace('{n}', str(g2_load_rows.n))]
    for c1, c2 in ((G2A[1], G2A[2]), (G2B[1], G2B[2])):
        for h_ in (0, 2):
            t += [f's_mov_b64 vcc, s[{c1 + h_}:{c1 + h_ + 1}]', f's_mov_b64 s[{c1 + h_}:{c1 + h_ + 1}], s[{c2 + h_}:{c2 + h_ + 1}]',
                  f's_mov_b64 s[{c2 + h_}:{c2 + h_ + 1}], vcc']
    t.append(f'.Lg2ns%=_{g2_load_rows.n}:')
    g2_load_rows.n += 1
    return t


g2_load_rows.n = 0


def g2_two_rows(regs, outs, real=False):
    """outs[0] (re, im pairs) = first resident row x inputs, outs[1] = second: four independent chains of eight (of four
    for a matrix promised real: its imaginary parts are never read)."""
    def m(base, c, part):
        return f's[{base[c] + (0 if part == "r" else 2)}:{base[c] + (1 if part == "r" else 3)}]'
    t = []
    for c in range(4):
        x_re, x_im = RE(regs[c]), IM(regs[c])
        for row, base in enumerate((G2A, G2B)):
            o_re, o_im = outs[row]
            if c == 0:
                t += [f'v_mul_f64 {o_re}, {m(base, c, "r")}, {x_re}', f'v_mul_f64 {o_im}, {m(base, c, "r")}, {x_im}']
            else:
                t += [f'v_fma_f64 {o_re}, {m(base, c, "r")}, {x_re}, {o_re}', f'v_fma_f64 {o_im}, {m(base, c, "r")}, {x_im}, {o_im}']
        for row, base in enumerate((G2A, G2B)):
            if real:
                continue
            o_re, o_im = outs[row]
            t += [f'v_fma_f64 {o_re}, -{m(base, c, "i")}, {x_im}, {o_re}', f'v_fma_f64 {o_im}, {m(base, c, "i")}, {x_re}, {o_im}']
    return t


def gen2_body(a, b, real=False):
    groups = gen2_groups(a, b)
    tag = 'r' if real else ''
    outs = [('v[10:11]', 'v[12:13]'), ('v[14:15]', 'v[16:17]')]
    t = []
    for batch in (0, 1):
        mine = list(enumerate(groups))[4 * batch:4 * batch + 4]
        t += g2_load_rows(0, G2OFFA)
        for gl, (i, j) in enumerate(mine):
            regs = [j | (((r >> 1) & 1) << b) | ((r & 1) << a) for r in range(4)]
            t += [f's_bitcmp1_b32 {G2MASK}, {i}', f's_cbranch_scc0 .Lg2a{tag}{a}{b}_{i}_%=']
            t += g2_two_rows(regs, outs, real)
            t += [f'ds_write_b128 {G2PARK}, v[10:13] offset:{1024 * (2 * gl)}', f'ds_write_b128 {G2PARK}, v[14:17] offset:{1024 * (2 * gl + 1)}']
            t.append(f'.Lg2a{tag}{a}{b}_{i}_%=:')
        t += g2_load_rows(G2OFFB, 192)
        for gl, (i, j) in enumerate(mine):
            regs = [j | (((r >> 1) & 1) << b) | ((r & 1) << a) for r in range(4)]
            t += [f's_bitcmp1_b32 {G2MASK}, {i}', f's_cbranch_scc0 .Lg2bb{tag}{a}{b}_{i}_%=']
            t += g2_two_rows(regs, outs, real)
            t += [f'v_mov_b64 {RE(regs[2])}, v[10:11]', f'v_mov_b64 {IM(regs[2])}, v[12:13]',
                  f'v_mov_b64 {RE(regs[3])}, v[14:15]', f'v_mov_b64 {IM(regs[3])}, v[16:17]',
                  f'ds_read_b128 {A(regs[0])}, {G2PARK} offset:{1024 * (2 * gl)}', f'ds_read_b128 {A(regs[1])}, {G2PARK} offset:{1024 * (2 * gl + 1)}']
            t.append(f'.Lg2bb{tag}{a}{b}_{i}_%=:')
        t.append('s_waitcnt lgkmcnt(0)')
    return t


def gen2_code():
    """Entry of every two-target dense record (within reach of the jump table): controls; what the body needs of the
    record -> registers of its own (s[72:79] will hold matrix entries); then the body of the slot pair through a second
    jump table whose entries lead to trampolines (the bodies lie in front of everything, out of the reach of s_branch)."""
    t = ['.Lgen2_%=:',
         f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=',
         f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
         's_cbranch_execz .Lrestore_%=',
         f's_mov_b32 {G2MASK}, s{REC + 5}', f's_mov_b32 {G2FLAG}, s{REC + 6}', f's_sub_u32 {G2PAIR}, s{REC}, {ID_GEN2}',
         f's_sub_u32 {G2MOFF}, {MOFF}, 256',
         # rows (0, 1 | 2, 3) of the matrix as the handler indexes it; with the swap flag rows 1 and 2 trade places
         f's_cmp_eq_u32 {G2FLAG}, 0', f's_cselect_b32 {G2OFFA}, 64, 128', f's_cselect_b32 {G2OFFB}, 128, 64',
         f'v_lshl_add_u32 {G2PARK}, {LANE}, 4, {LDSB}',
         's_getpc_b64 vcc', '.Lg2anchor_%=:', f's_lshl3_add_u32 vcc_lo, {G2PAIR}, vcc_lo', 's_addc_u32 vcc_hi, vcc_hi, 0',
         's_add_u32 vcc_lo, vcc_lo, .Lg2table_%=-.Lg2anchor_%=', 's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc', '.Lg2table_%=:']
    for v in range(2 * len(SWAP_PAIRS)):
        t += [f's_branch .Lg2t{v}_%=', 's_nop 0']
    for v in range(2 * len(SWAP_PAIRS)):
        t += [f'.Lg2t{v}_%=:', 's_getpc_b64 vcc', f'.Lg2ta{v}_%=:', f's_sub_u32 vcc_lo, vcc_lo, .Lg2ta{v}_%=-.Lg2b{v}_%=',
              's_subb_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc']
    return t


def gen2_bodies():
    t = []
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{v}_%=:'] + gen2_body(a, b) + [f's_mov_b64 exec, {SAVE}'] + far_next()
    for v, (a, b) in enumerate(SWAP_PAIRS):
        t += [f'.Lg2b{len(SWAP_PAIRS) + v}_%=:'] + gen2_body(a, b, real=True) + [f's_mov_b64 exec, {SAVE}'] + far_next()
    return t


def grad_groups(q):
    return [j for j in range(NA) if not (j >> q) & 1 and not j & 1]


def grad_code(q, variant=0):
    """(variant 1 / 2 / 3 -- DqFusedGate::loc of the record: the trainable gate's matrix is real / of the form a I + i b X
    / diagonal -- forms only the sums such a gate's gradient can need: Re G; Re (G00 + G11) in the place of Re G00 and
    Im (G01 + G10) in the place of Im G01; G00 and G11.  Eight operations per register group instead of sixteen; the
    other accumulators stay zero and take the same way through the reduction.  Variant 4: Im (G01 + G10) alone -- a unitary
    a I + i b X in a backward that records no graph, as in the complex64 generator.)

    DQ_FG_GRAD with the target on slot q, psi (0) / lambda (1) on slot 0: G[a][b] = sum lambda[target = a] conj(psi[target
    = b]) over the thread's register groups (w5 = mask of the groups whose register controls are set; lanes that fail the
    thread controls contribute nothing), eight float64 sums per lane: (G00, G01, G10, G11) x (re, im).  Across the lanes
    they go through the wave's staging buffer: every lane writes its eight sums (72-byte rows), lane L then adds sum
    number L & 7 of the eight lanes 8 (L >> 3) .. + 7, scales by the square of the pass's deferred factor and adds to the
    record's accumulator in LDS (ACC_BASE + 64 * record number + 8 * (L & 7)): eight lanes per address."""
    G = ['v[10:11]', 'v[12:13]', 'v[14:15]', 'v[16:17]', 'v[32:33]', 'v[34:35]', 'v[36:37]', 'v[38:39]']     # re, im of G00 G01 G10 G11
    t = [f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=']
    t += [f'v_mov_b32 v{r}, 0' for r in list(range(10, 18)) + list(range(32, 40))]
    t += [f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
          f's_cbranch_execz .Lgz{q}v{variant}_%=']
    tag = f'{q}v{variant}'
    for i, j in enumerate(grad_groups(q)):
        ps, ls = (j, j | (1 << q)), (j | 1, j | (1 << q) | 1)
        t += [f's_bitcmp1_b32 s{REC + 5}, {i}', f's_cbranch_scc0 .Lgg{tag}_{i}_%=']
        ab = [(a_, b_) for a_ in range(2) for b_ in range(2)]

        def re_(c, a_, b_):          # accumulator c += Re (lambda_a conj psi_b), as two terms (interleaved by the caller)
            return [f'v_fma_f64 {G[c]}, {RE(ls[a_])}, {RE(ps[b_])}, {G[c]}', f'v_fma_f64 {G[c]}, {IM(ls[a_])}, {IM(ps[b_])}, {G[c]}']

        def im_(c, a_, b_):          # accumulator c += Im (lambda_a conj psi_b)
            return [f'v_fma_f64 {G[c]}, {IM(ls[a_])}, {RE(ps[b_])}, {G[c]}', f'v_fma_f64 {G[c]}, -{RE(ls[a_])}, {IM(ps[b_])}, {G[c]}']

        if variant == 0:
            # term by term over the four entries: four independent chains per term
            t += [f'v_fma_f64 {G[2 * (2 * a_ + b_)]}, {RE(ls[a_])}, {RE(ps[b_])}, {G[2 * (2 * a_ + b_)]}' for a_, b_ in ab]
            t += [f'v_fma_f64 {G[2 * (2 * a_ + b_) + 1]}, {IM(ls[a_])}, {RE(ps[b_])}, {G[2 * (2 * a_ + b_) + 1]}' for a_, b_ in ab]
            t += [f'v_fma_f64 {G[2 * (2 * a_ + b_)]}, {IM(ls[a_])}, {IM(ps[b_])}, {G[2 * (2 * a_ + b_)]}' for a_, b_ in ab]
            t += [f'v_fma_f64 {G[2 * (2 * a_ + b_) + 1]}, -{RE(ls[a_])}, {IM(ps[b_])}, {G[2 * (2 * a_ + b_) + 1]}' for a_, b_ in ab]
        else:
            if variant == 1:
                chains = [re_(2 * (2 * a_ + b_), a_, b_) for a_, b_ in ab]
            elif variant == 2:       # two accumulators, each fed by two chains: (0, 0) then (1, 1); (0, 1) then (1, 0)
                chains = [re_(0, 0, 0) + re_(0, 1, 1), im_(3, 0, 1) + im_(3, 1, 0)]
            elif variant == 4:       # Im (G01 + G10) alone: two chains, the second in a scratch accumulator folded in below
                chains = [im_(3, 0, 1), im_(5, 1, 0)]
            else:
                chains = [re_(0, 0, 0), im_(1, 0, 0), re_(6, 1, 1), im_(7, 1, 1)]
            k_ = 0
            while any(k_ < len(ch) for ch in chains):
                t += [ch[k_] for ch in chains if k_ < len(ch)]
                k_ += 1
        t.append(f'.Lgg{tag}_{i}_%=:')
    t += [f'.Lgz{tag}_%=:', f's_mov_b64 exec, {SAVE}']
    if variant == 4:
        t += [f'v_add_f64 {G[3]}, {G[3]}, {G[5]}', 'v_mov_b32 v34, 0', 'v_mov_b32 v35, 0']
    t += [f'v_mul_f64 v[6:7], {HS}, {HS}',
          f'v_mul_u32_u24 v8, 72, {LANE}', f'v_add_u32 v8, {LDSB}, v8']
    t += [f'ds_write_b64 v8, {G[c]} offset:{8 * c}' for c in range(8)]
    t += [f'v_lshrrev_b32 v9, 3, {LANE}', 'v_mul_u32_u24 v9, 576, v9', f'v_and_b32 {TT}, 7, {LANE}', f'v_lshl_add_u32 v9, {TT}, 3, v9',
          f'v_add_u32 v9, {LDSB}, v9']
    t += [f'ds_read_b64 {G[c]}, v9 offset:{72 * c}' for c in range(8)]
    t += ['s_waitcnt lgkmcnt(0)',
          f'v_add_f64 {G[0]}, {G[0]}, {G[1]}', f'v_add_f64 {G[2]}, {G[2]}, {G[3]}', f'v_add_f64 {G[4]}, {G[4]}, {G[5]}', f'v_add_f64 {G[6]}, {G[6]}, {G[7]}',
          f'v_add_f64 {G[0]}, {G[0]}, {G[2]}', f'v_add_f64 {G[4]}, {G[4]}, {G[6]}', f'v_add_f64 {G[0]}, {G[0]}, {G[4]}',
          f'v_mul_f64 {G[0]}, {G[0]}, v[6:7]',
          f's_lshl_b32 {STMP}, {GOFF}, 1', f'v_lshlrev_b32 {TT}, 3, {TT}', f'v_add_u32 v9, {STMP}, {TT}', f'v_add_u32 v9, {ACC_BASE - 64}, v9',
          f'ds_add_f64 v9, {G[0]}']
    return t


def expz_code():
    """As in the complex64 generator, float64 sums: w5 = sign of register j (32 registers), w1 / w2:w3 the lane's and the
    tile's parity masks; eight-byte accumulator (component 0 of the record's row)."""
    P, M_ = 'v[10:11]', 'v[12:13]'
    t = [f'v_mov_b32 v{r}, 0' for r in range(10, 14)]
    for j in range(NA):
        t += [f's_bitcmp1_b32 s{REC + 5}, {j}', f's_cbranch_scc1 .Lezm{j}_%=',
              f'v_fma_f64 {P}, {RE(j)}, {RE(j)}, {P}', f'v_fma_f64 {P}, {IM(j)}, {IM(j)}, {P}', f's_branch .Lezn{j}_%=',
              f'.Lezm{j}_%=:', f'v_fma_f64 {M_}, {RE(j)}, {RE(j)}, {M_}', f'v_fma_f64 {M_}, {IM(j)}, {IM(j)}, {M_}', f'.Lezn{j}_%=:']
    t += [f'v_add_f64 {P}, {P}, -{M_}',
          f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_bcnt_u32_b32 {TT}, {TT}, 0',
          f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_bcnt1_i32_b64 {STMP}, vcc',
          f'v_add_u32 {TT}, {STMP}, {TT}', f'v_lshlrev_b32 {TT}, 31, {TT}', f'v_xor_b32 v11, {TT}, v11',
          f'v_mul_f64 v[6:7], {HS}, {HS}', f'v_mul_f64 {P}, {P}, v[6:7]',
          f's_lshl_b32 {STMP}, {GOFF}, 1', f's_add_u32 {STMP}, {STMP}, {ACC_BASE - 64}', f'v_mov_b32 v9, {STMP}',
          f'ds_add_f64 v9, {P}']
    return t


def cmul_inplace(j, ph):
    t = TMP[0] if j % 2 == 0 else TMP[1]
    return [f'v_mul_f64 {t}, {RE(j)}, {ph[1]}', f'v_mul_f64 {RE(j)}, {RE(j)}, {ph[0]}',
            f'v_fma_f64 {RE(j)}, -{IM(j)}, {ph[1]}, {RE(j)}', f'v_fma_f64 {IM(j)}, {IM(j)}, {ph[0]}, {t}']


def diag_body(variant):
    masked_, v = variant >= R + 1, variant % (R + 1)
    out_ = []
    for j in range(NA):
        ph = PH0 if v == 0 or not (j >> (v - 1)) & 1 else PH1
        if masked_:
            out_ += [f's_bitcmp1_b32 s{REC + 6}, {j}', f's_cbranch_scc0 .Ldg{variant}_{j}_%=']
        out_ += cmul_inplace(j, ph)
        if masked_:
            out_.append(f'.Ldg{variant}_{j}_%=:')
    return out_


def diag_code():
    """As in the complex64 generator.  The four phases (16 bytes each) sit in s[80:95] -- the diagonal of the 2x2 block the
    loop head loaded (entries 0, 3, twice), or of the 4x4 block that ends at the current matrix offset (entries 0, 5, 10,
    15) -- and the candidates of PH0 / PH1 go through s[40:43] into VGPRs."""
    W5 = f's{REC + 5}'
    t = ['.Ldiag_%=:',
         f's_and_b64 vcc, s[{REC + 2}:{REC + 3}], {TG}', f's_cmp_eq_u64 vcc, s[{REC + 2}:{REC + 3}]', 's_cbranch_scc0 .Lnext_%=',
         f'v_and_b32 {TT}, s{REC + 1}, {TB}', f'v_cmp_eq_u32 vcc, s{REC + 1}, {TT}', f's_and_saveexec_b64 {SAVE}, vcc',
         's_cbranch_execz .Lrestore_%=',
         f's_cmp_ge_u32 s{REC}, {ID_DIAG2}', 's_cbranch_scc1 .Ldiag4_%=',
         # 2x2 block: entry 0 = s[80:83] stays, entry 3 = s[92:95] -> phase 1 (s[84:87]); phases 2, 3 = copies
         's_mov_b64 s[84:85], s[92:93]', 's_mov_b64 s[86:87], s[94:95]', 's_mov_b64 s[88:89], s[80:81]', 's_mov_b64 s[90:91], s[82:83]',
         's_mov_b64 s[92:93], s[84:85]', 's_mov_b64 s[94:95], s[86:87]',
         's_branch .Ldiagsel_%=', '.Ldiag4_%=:', f's_sub_u32 s69, {MOFF}, 256']
    for k in range(4):
        t += [f's_load_dwordx4 s[{80 + 4 * k}:{83 + 4 * k}], {MB}, s69'] + (['s_add_u32 s69, s69, 80'] if k < 3 else [])
    t += ['s_waitcnt lgkmcnt(0)', '.Ldiagsel_%=:']
    for name, byte, dst in (('a', 0, 's[50:51]'), ('b', 8, 's[70:71]')):
        t += [f's_bfe_u32 s69, {W5}, {byte | (6 << 16)}', f's_bfe_u32 vcc_lo, {W5}, {(byte + 6) | (2 << 16)}', f's_mov_b64 {dst}, 0',
              's_cmp_eq_u32 vcc_lo, 1', f's_cbranch_scc0 .Lsel{name}o_%=',
              f'v_lshrrev_b32 {TT}, s69, {TB}', f'v_and_b32 {TT}, 1, {TT}', f'v_cmp_ne_u32 {dst}, 0, {TT}', f's_branch .Lsel{name}d_%=',
              f'.Lsel{name}o_%=:', 's_cmp_eq_u32 vcc_lo, 2', f's_cbranch_scc0 .Lsel{name}d_%=',
              f's_lshr_b64 vcc, {TG}, s69', 's_bitcmp1_b32 vcc_lo, 0', f's_cselect_b64 {dst}, -1, 0', f'.Lsel{name}d_%=:']

    def fetch(bitpos, vbase):      # candidate = phase number (2 bits of w5 at bitpos) -> 4 dwords in v[vbase : vbase + 3]
        return [f's_bfe_u32 s69, {W5}, {bitpos | (2 << 16)}', 's_lshl_b32 m0, s69, 2', 's_nop 0',
                's_movrels_b64 s[40:41], s[80:81]', 's_movrels_b64 s[42:43], s[82:83]'] + \
               [f'v_mov_b32 v{vbase + d}, s{40 + d}' for d in range(4)]

    # PH0 = selA ? (selB ? c3 : c2) : (selB ? c1 : c0): candidates in v[10:13], v[14:17], v[36:39], v[32:35]
    cv = [10, 14, 36, 32]
    for k in range(4):
        t += fetch(16 + 2 * k, cv[k])
    for d in range(4):
        t += [f'v_cndmask_b32 v{cv[0] + d}, v{cv[0] + d}, v{cv[1] + d}, s[70:71]', f'v_cndmask_b32 v{cv[2] + d}, v{cv[2] + d}, v{cv[3] + d}, s[70:71]',
              f'v_cndmask_b32 v{6 + d}, v{cv[0] + d}, v{cv[2] + d}, s[50:51]']
    # PH1 = selA ? c2 : c0 -> v[32:35]
    t += fetch(24, 10) + fetch(28, 14)
    t += [f'v_cndmask_b32 v{32 + d}, v{10 + d}, v{14 + d}, s[50:51]' for d in range(4)]
    t += [f's_sub_u32 s69, s{REC}, {ID_DIAG1}', f's_cmp_ge_u32 s69, {NV}', 's_cbranch_scc0 .Ldiagv_%=', f's_sub_u32 s69, s69, {NV}', '.Ldiagv_%=:',
          's_getpc_b64 vcc', '.Ldiaganchor_%=:', 's_lshl2_add_u32 vcc_lo, s69, vcc_lo', 's_addc_u32 vcc_hi, vcc_hi, 0',
          's_add_u32 vcc_lo, vcc_lo, .Ldiagtable_%=-.Ldiaganchor_%=', 's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc', '.Ldiagtable_%=:']
    t += [f's_branch .Ldiagb{v}_%=' for v in range(NV)]
    for v in range(NV):
        t += [f'.Ldiagb{v}_%=:'] + diag_body(v) + [f's_mov_b64 exec, {SAVE}', f's_cmp_lt_u32 {GOFF}, {GEND}', 's_cbranch_scc1 .Lloop_%=', 's_branch .Lexit_%=']
    return t


def gray_walk(op, base_operand, lane_operand, nt=False):
    """32 x (address = base + running slot offset + lane offset; op): a Gray code over the five slot bits."""
    out_ = [f's_mov_b64 {RUN}, {base_operand}']
    for i in range(NA):
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
        out_.append((f'global_load_dwordx4 {A(g)}, {ad}, off' if op == 'load' else f'global_store_dwordx4 {ad}, {A(g)}, off') + (' nt' if nt else ''))
    return out_


def zext_load(base_operand, lane_operand):
    """The loads of a pass whose input has index bits KNOWN TO BE |0> (dq_apply_fused_zext_c128; see tools/gen_wave_asm.py):
    flags bits 8..12 / 16..21 = the register slots / lane bits of the load layout that hold such bits."""
    t = ['s_bfe_u32 s69, %[flags], 0x000e0008', 's_cmp_eq_u32 s69, 0', 's_cbranch_scc1 .Lldn_%=']
    for j in range(NA):
        t += [f'v_mov_b64 {RE(j)}, 0', f'v_mov_b64 {IM(j)}, 0']
    t += ['s_lshr_b32 s70, s69, 8', 's_and_b32 s69, s69, 0x3f', f'v_and_b32 {TT}, s70, {LANE}', f'v_cmp_eq_u32 vcc, 0, {TT}',
          f's_and_saveexec_b64 {SAVE}, vcc', f's_mov_b64 {RUN}, {base_operand}']
    for i in range(NA):
        g = i ^ (i >> 1)
        if i:
            b = (i & -i).bit_length() - 1
            lo, hi = f's{40 + 2 * b}', f's{41 + 2 * b}'
            if (g >> b) & 1:
                t += [f's_add_u32 s50, s50, {lo}', f's_addc_u32 s51, s51, {hi}']
            else:
                t += [f's_sub_u32 s50, s50, {lo}', f's_subb_u32 s51, s51, {hi}']
        if g:
            t += [f's_and_b32 s70, s69, {g}', f's_cbranch_scc1 .Lzx{i}_%=']
        ad = ADDR[i % 4]
        t += [f'v_lshl_add_u64 {ad}, {RUN}, 0, {lane_operand}', f'global_load_dwordx4 {A(g)}, {ad}, off']
        if g:
            t.append(f'.Lzx{i}_%=:')
    return t + [f's_mov_b64 exec, {SAVE}', 's_branch .Lldd_%=', '.Lldn_%=:']


def kernel_body():
    h = handlers()
    ids = sorted(h)
    front = [i for i in ids if i < ID_TRIP0]
    back = [i for i in ids if i >= ID_TRIP0]
    nxt = [f's_cmp_lt_u32 {GOFF}, {GEND}', 's_cbranch_scc1 .Lloop_%=', 's_branch .Lexit_%=']

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

    text = ['s_getpc_b64 vcc', '.Ljs_%=:', 's_add_u32 vcc_lo, vcc_lo, .Lstart_%=-.Ljs_%=', 's_addc_u32 vcc_hi, vcc_hi, 0',
            's_setpc_b64 vcc'] + gen2_bodies() + ['.Lstart_%=:']
    text += [f's_mov_b64 {KG}, %[kg]', f's_mov_b32 {GOFF}, 0', f's_mov_b32 {GEND}, %[gend]', f's_mov_b64 {MB}, %[mb]',
            f's_mov_b32 {MOFF}, %[moff]', f's_mov_b64 {TG}, %[tg]', f's_mov_b32 {LDSB}, %[ldsb]',
            's_load_dwordx8 s[40:47], %[ks], 0', 's_load_dwordx2 s[48:49], %[ks], 32',
            f's_load_dwordx8 s[{REC}:{REC + 7}], %[ks], 80', 's_load_dwordx8 s[80:87], %[ks], 112',
            's_load_dwordx2 s[88:89], %[ks], 144',
            f'v_and_b32 {LANE}, 63, %[tid]', 'v_mov_b32 v20, 0', 'v_mov_b32 v21, 0x3ff00000']        # HS = 1.0
    text += [f'v_bfe_i32 {LB[b]}, {LANE}, {b}, 1' for b in range(6)]
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
    text += [f's_getpc_b64 {TABLE}', '.Lanchor_%=:', 's_add_u32 s54, s54, .Ltable_%=-.Lanchor_%=', 's_addc_u32 s55, s55, 0',
             's_waitcnt vmcnt(0)', 's_branch .Lloop_%=']
    for i in front:
        text += emit(i)
    text += ['.Lrestore_%=:', f's_mov_b64 exec, {SAVE}', '.Lnext_%=:'] + nxt
    text += ['.Lloop_%=:',
             f's_load_dwordx8 s[{REC}:{REC + 7}], {KG}, {GOFF}',
             f's_load_dwordx16 s[{MAT}:{MAT + 15}], {MB}, {MOFF}',
             's_waitcnt lgkmcnt(0)',
             f's_add_u32 {GOFF}, {GOFF}, 32',
             f's_lshl4_add_u32 {MOFF}, s{REC + 4}, {MOFF}',
             f's_lshl2_add_u32 vcc_lo, s{REC}, s54', 's_addc_u32 vcc_hi, s55, 0', 's_setpc_b64 vcc',
             '.Ltable_%=:']
    for i in range(NIDS):
        text.append(f's_branch .Lh{i}_%=' if i in h else ('s_branch .Lgen2_%=' if i >= ID_GEN2 else
                                                         's_branch .Ldiag_%=' if i >= ID_DIAG1 else 's_branch .Lnext_%='))
    text += ['.Lexit_%=:', 's_load_dwordx8 s[40:47], %[ks], 40', 's_load_dwordx2 s[48:49], %[ks], 72', 's_waitcnt lgkmcnt(0)']
    for j in range(NA):
        text += [f'v_mul_f64 {RE(j)}, {RE(j)}, {HS}', f'v_mul_f64 {IM(j)}, {IM(j)}, {HS}']
    text += ['s_bitcmp1_b32 %[flags], 1', 's_cbranch_scc0 .Lstp_%='] + gray_walk('store', '%[outb]', LST, nt=True) + ['s_branch .Ldone_%=', '.Lstp_%=:']
    text += gray_walk('store', '%[outb]', LST)
    text += ['s_branch .Ldone_%=']
    text += gen2_code()
    text += diag_code()
    for i in back:
        text += emit(i)
    text += ['.Ldone_%=:']
    return text


if __name__ == '__main__' or os.environ.get('DQ_ASM_OUT'):
    out = ['// GENERATED by tools/gen_wave_asm64.py -- do not edit by hand.', '// clang-format off',
           f'#define DQ_WAVE64_NIDS {NIDS}', f'#define DQ_WAVE64_MAXK {MAXK}',
           f'#define DQ_WID64_GEN_U {ID_GEN_U}', f'#define DQ_WID64_GEN_C {ID_GEN_C}', f'#define DQ_WID64_GEN_R {ID_GEN_R}',
           f'#define DQ_WID64_X_U {ID_X_U}', f'#define DQ_WID64_X_C {ID_X_C}', f'#define DQ_WID64_X_R {ID_X_R}', f'#define DQ_WID64_X_R1 {ID_X_R1}',
           f'#define DQ_WID64_TRIP0 {ID_TRIP0}', f'#define DQ_WID64_TRIP {ID_TRIP}', f'#define DQ_WID64_SWAP {ID_SWAP}',
           f'#define DQ_WID64_DIAG1 {ID_DIAG1}', f'#define DQ_WID64_DIAG2 {ID_DIAG2}', f'#define DQ_WID64_GRAD {ID_GRAD}', f'#define DQ_WAVE64_GRAD_VARIANTS {GRAD_VARIANTS}', f'#define DQ_WID64_EXPZ {ID_EXPZ}', f'#define DQ_WID64_GEN2 {ID_GEN2}', f'#define DQ_WID64_GEN2R {ID_GEN2R}', f'#define DQ_WAVE64_ACC_BASE {ACC_BASE}',
           'static const short kWave64TripId[32] = {' + ', '.join(str(ID_TRIP + TRIP_MASKS.index(m)) if m in TRIP_MASKS else '-1' for m in range(NA)) + '};',
           'static const short kWave64SwapId[5][5] = {' + ', '.join('{' + ', '.join(str(ID_SWAP + SWAP_PAIRS.index((min(i, j), max(i, j)))) if i != j else '-1' for j in range(R)) + '}' for i in range(R)) + '};',
           '']
    text = '\\n\\t"\n        "'.join(kernel_body())
    clob = ', '.join(['"m0"'] + [f'"s{i}"' for i in range(40, 96)] + [f'"v{i}"' for i in range(1, AMP0 + 4 * NA)])
    out += ['__device__ __forceinline__ void wave_tile_body_f64(uint64_t kg, uint32_t gend, uint64_t mb, uint32_t moff, uint64_t tg,',
            '                                                   uint64_t ks, uint64_t inb, uint64_t outb, uint32_t ldsb, uint32_t tid, uint32_t flags) {',
            f'    asm volatile(\n        "{text}"',
            '        :',
            '        : [kg] "s"(kg), [gend] "s"(gend), [mb] "s"(mb), [moff] "s"(moff), [tg] "s"(tg), [ks] "s"(ks), [inb] "s"(inb),',
            '          [outb] "s"(outb), [ldsb] "s"(ldsb), [tid] "v"(tid), [flags] "s"(flags)',
            f'        : "vcc", "scc", "memory", {clob});',
            '}', '// clang-format on', '']
    path = os.environ.get('DQ_ASM_OUT') or os.path.join(os.path.dirname(__file__), '..', 'deepquantum_amd', 'csrc', 'dq_wave_asm64.inc')
    open(path, 'w').write('\n'.join(out))
    print('generated', NIDS, 'handler ids;', sum(len(v[1]) for v in handlers().values()), 'handler instructions')
