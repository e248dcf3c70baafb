"""Generate csrc/dq_fused_asm.inc: straight-line inline-asm bodies of the fused kernel (complex64 with 16
amplitudes per thread, complex128 with 8).

One asm statement per (matrix structure MODE, target slot Q) updates all 16 register-resident amplitudes
of a thread in place (tied "+v" operands), 4 / 4 / 8 packed VALU ops per amplitude pair; X-type gates
are v_swap_b32 blocks per (Q, control-slot mask).  Generated code is committed; re-run after editing.
"""

R = 4
NA = 1 << R

def pairs(q, cmask=0):
    return [(j, j | (1 << q)) for j in range(NA) if not (j >> q) & 1 and (j & cmask) == cmask]

def operands():
    outs = ', '.join(f'"+v"(a[{j}])' for j in range(NA))
    outs += ', "=&v"(t0), "=&v"(u0), "=&v"(t1), "=&v"(u1)'
    return outs

T0, U0, T1, U1 = NA, NA + 1, NA + 2, NA + 3
S0 = NA + 4  # first scalar operand index
MATREGS = None  # fixed SGPR pairs instead of operands (the gate loop written in assembly)

def body(mode, q):
    """Scalar operands are the four raw matrix entries as 64-bit SGPR pairs (low half = re, high = im);
    op_sel / op_sel_hi pick the half, so no scalar instruction is spent on re-packing them."""
    lines = []
    P = MATREGS or [f'%{S0 + i}' for i in range(4)]  # m00, m01, m10, m11
    RE2, RE3 = 'op_sel_hi:[1,0]', 'op_sel_hi:[1,0,1]'                      # broadcast the low half (re)
    I2 = 'op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]'                       # (i * x) * im:  (-x.im, x.re) * im
    I3 = 'op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]'
    for k, (lo, hi) in enumerate(pairs(q)):
        t, u = (T0, U0) if k % 2 == 0 else (T1, U1)
        A, B = f'%{lo}', f'%{hi}'
        t, u = f'%{t}', f'%{u}'
        if mode == 3:    # s * [[1, 1], [1, -1]] with the factor s deferred to the end of the pass:
            # A' = A + B in place, then B' = A - B = A' - 2 B  (P[0] holds the pair (-2, -2))
            # (all sums first, then all differences: no packed op reads the result of its predecessor)
            lines.insert(k, f'v_pk_add_f32 {A}, {A}, {B}')
            lines.append(f'v_pk_fma_f32 {B}, {B}, {P[0]}, {A}')
        elif mode == 1:    # all entries real
            lines += [f'v_pk_mul_f32 {t}, {B}, {P[1]} {RE2}',
                      f'v_pk_mul_f32 {u}, {A}, {P[2]} {RE2}',
                      f'v_pk_fma_f32 {A}, {A}, {P[0]}, {t} {RE3}',
                      f'v_pk_fma_f32 {B}, {B}, {P[3]}, {u} {RE3}']
        elif mode == 2:  # real diagonal, imaginary off-diagonal
            lines += [f'v_pk_mul_f32 {t}, {B}, {P[1]} {I2}',
                      f'v_pk_mul_f32 {u}, {A}, {P[2]} {I2}',
                      f'v_pk_fma_f32 {A}, {A}, {P[0]}, {t} {RE3}',
                      f'v_pk_fma_f32 {B}, {B}, {P[3]}, {u} {RE3}']
        else:            # general complex; the two accumulation chains are interleaved so that no packed op
            # reads a register written by the instruction right before it (hipcc pads such pairs with
            # s_nop 0 on gfx950; inside an asm statement nobody would)
            lines += [f'v_pk_mul_f32 {t}, {B}, {P[1]} {RE2}',
                      f'v_pk_mul_f32 {u}, {A}, {P[2]} {RE2}',
                      f'v_pk_fma_f32 {t}, {B}, {P[1]}, {t} {I3}',
                      f'v_pk_fma_f32 {u}, {A}, {P[2]}, {u} {I3}',
                      f'v_pk_fma_f32 {t}, {A}, {P[0]}, {t} {I3}',
                      f'v_pk_fma_f32 {u}, {B}, {P[3]}, {u} {I3}',
                      f'v_pk_fma_f32 {A}, {A}, {P[0]}, {t} {RE3}',
                      f'v_pk_fma_f32 {B}, {B}, {P[3]}, {u} {RE3}']
    return lines


# Uncontrolled Rx-like gate  a I + i b X  with its scalar factor deferred like the Hadamards' (ids 8..11): the host
# hands over  f = the larger of a and i b,  t = the ratio of the other to it (|t| <= 1)  instead of the matrix
# (fusion.defer_rx):
#   form 0, f = a:    A' = A + (i t) B,  B' = B + (i t) A      t =  b / a
#   form 1, f = i b:  A' = B + (i t) A,  B' = A + (i t) B      t = -a / b
# one register copy + two packed FMAs per amplitude pair instead of two multiplications + two FMAs; the deferred factor
# (hr + i hi) is multiplied by f.  `fr`, `fi`, `flag` name 32-bit scalar sources, `it` the SGPR pair whose HIGH half is t.
def body_rx_deferred(q, it, fr, fi, flag):
    I3 = 'op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]'                 # (x, y) * (i t) + acc
    f0, f1 = [], []
    for k, (lo, hi) in enumerate(pairs(q)):
        t = f'%{T0 if k % 2 == 0 else T1}'
        A, B = f'%{lo}', f'%{hi}'
        f0 += [f'v_mov_b64 {t}, {A}', f'v_pk_fma_f32 {A}, {B}, {it}, {A} {I3}', f'v_pk_fma_f32 {B}, {t}, {it}, {B} {I3}']
        f1 += [f'v_mov_b64 {t}, {A}', f'v_pk_fma_f32 {A}, {A}, {it}, {B} {I3}', f'v_pk_fma_f32 {B}, {B}, {it}, {t} {I3}']
    # interleave the two dependency chains a little: copies of pair k + 1 before the FMAs of pair k are not needed --
    # the three instructions of a pair touch other registers than the next pair's
    return ([f's_cmp_eq_u32 {flag}, 0', 's_cbranch_scc0 .Ldqrx1_%=_Q'.replace('_Q', f'_{q}_@'),
             f'v_mul_f32 %[hr], %[hr], {fr}', f'v_mul_f32 %[hi], %[hi], {fr}'] + f0 +
            ['s_branch .Ldqrx2_%=_Q'.replace('_Q', f'_{q}_@'), '.Ldqrx1_%=_Q:'.replace('_Q', f'_{q}_@'),
             # (hr + i hi) * (i b) = -hi b + i hr b
             f'v_mul_f32 %[tt], %[hr], {fi}', f'v_mul_f32 %[hr], %[hi], {fi}', 'v_xor_b32 %[hr], 0x80000000, %[hr]',
             'v_mov_b32 %[hi], %[tt]'] + f1 + ['.Ldqrx2_%=_Q:'.replace('_Q', f'_{q}_@')])


out = ['// GENERATED by tools/gen_fused_asm.py -- do not edit by hand.',
       '// clang-format off',
       'using V2F = vec2<float>;', '']
for mode in (0, 1, 2, 3):
    for q in range(R):
        lines = body(mode, q)
        text = '\\n\\t"\n        "'.join(lines)
        ins = ', '.join(f'"s"(mq[{i}])' for i in range(4))
        out += [f'template <> __device__ __forceinline__ void gen1_block_f32<{mode}, {q}>(V2F (&a)[16], const uint64_t (&mq)[4]) {{',
                '    V2F t0, u0, t1, u1;',
                f'    asm volatile(\n        "{text}"',
                f'        : {operands()}',
                f'        : {ins});',
                '}', '']
# X-type: swap the pairs of slot q whose control slots (cmask) are all one.  64-bit register moves
# through one scratch pair (operands are whole amplitudes; sub-registers cannot be named in inline asm).
for q in range(R):
    for cmask in range(NA):
        if (cmask >> q) & 1:
            continue
        ps = pairs(q, cmask)
        lines = []
        for k, (lo, hi) in enumerate(ps):
            t = NA + (k % 2)
            lines += [f'v_mov_b64 %{t}, %{lo}', f'v_mov_b64 %{lo}, %{hi}', f'v_mov_b64 %{hi}, %{t}']
        text = '\\n\\t"\n        "'.join(lines)
        ops = ', '.join(f'"+v"(a[{j}])' for j in range(NA)) + ', "=&v"(t0), "=&v"(t1)'
        out += [f'template <> __device__ __forceinline__ void x1_block_f32<{q}, {cmask}>(V2F (&a)[16]) {{',
                '    V2F t0, t1;',
                f'    asm volatile(\n        "{text}"',
                f'        : {ops});',
                '}', '']
# ---------------------------------------------------------------------------------------------------
# complex64 dispatcher: ALL straight-line handlers in ONE asm statement, entered through a jump table.  hipcc lowers
# a C++ switch to a tree of s_cmp / s_cbranch (6-8 levels for this many handlers, each level two issue slots of the
# wave and most of them a taken branch); here the host-chosen handler id indexes a table of s_branch instructions:
#     s_getpc_b64 vcc ; vcc += 12 + 4 * id ; s_setpc_b64 vcc ; s_branch L0 ; s_branch L1 ; ...
# (12 = the three 4-byte instructions between the address s_getpc returns and the table).  Handler ids are the
# ones of include/dq_hip.h (DqFusedGate::fast) plus SKIP_ID for "an outside control is 0".  Handlers of
# controlled gates start with the thread-control test (exec mask; skipped entirely when no lane is selected).
NIDS = 76              # 0..51: gates (include/dq_hip.h); 52 + 6 * slot + lane: in-wave swap of a register slot with a lane bit
DS0 = NA + 9           # operand numbers: a[0..15], t0 u0 t1 u1, tt, save, st, hr, hi, then m00 m01 m10 m11 (SGPR pairs)


def dispatcher():
    global S0
    keep, S0 = S0, DS0
    handlers = {}
    for mode in (0, 1, 2, 3):
        for q in range(R):
            lines = body(mode, q)
            if mode == 3:   # Hadamard: the pair (-2, -2) lives in vcc, the deferred factor goes into hr + i hi
                lines = [ln.replace(f'%{DS0}', 'vcc') for ln in lines]
                lines = ['s_mov_b32 vcc_lo, 0xc0000000', 's_mov_b32 vcc_hi, 0xc0000000',
                         'v_mul_f32 %[hr], %[hr], %[m00]', 'v_mul_f32 %[hi], %[hi], %[m00]'] + lines
            if mode == 2:   # uncontrolled Rx-like: deferred form (the host rewrote the matrix block)
                lines = [ln.replace('@', 'd') for ln in body_rx_deferred(q, f'%{DS0 + 1}', '%[m00]', '%[m00i]', '%[m11]')]
            handlers[4 * mode + q] = (False, lines)
            if mode < 3:
                handlers[20 + 4 * mode + q] = (True, body(mode, q))

    def xlines(q, cmask):
        out_ = []
        for k, (lo, hi) in enumerate(pairs(q, cmask)):
            t = NA + (k % 2)
            out_ += [f'v_mov_b64 %{t}, %{lo}', f'v_mov_b64 %{lo}, %{hi}', f'v_mov_b64 %{hi}, %{t}']
        return out_

    for q in range(R):
        handlers[16 + q] = (False, xlines(q, 0))
        handlers[32 + q] = (True, xlines(q, 0))
        for c in range(R):
            if c != q:
                handlers[36 + 4 * q + c] = (True, xlines(q, 1 << c))
    S0 = keep
    text = ['s_lshl2_add_u32 %[st], %[id], 12', 's_getpc_b64 vcc', 's_add_u32 vcc_lo, vcc_lo, %[st]',
            's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc']
    for i in range(NIDS):
        text.append(f's_branch .Ldq{i}_%=' if i in handlers else 's_branch .Ldqend_%=')
    for i in sorted(handlers):
        ctl, lines = handlers[i]
        text.append(f'.Ldq{i}_%=:')
        if ctl:   # outside controls (uniform per workgroup): all 1 or the gate does nothing here; thread controls:
            # lanes whose tile-local base has all control bits set (g1 = loc2 | reg_cmask << 8 | thr_cmask << 16)
            text += ['s_and_b64 vcc, %[oc], %[tg]', 's_cmp_eq_u64 vcc, %[oc]', 's_cbranch_scc0 .Ldqend_%=',
                     's_lshr_b32 %[st], %[g1], 16', 'v_and_b32 %[tt], %[st], %[tb]',
                     'v_cmp_eq_u32 vcc, %[st], %[tt]', 's_and_saveexec_b64 %[save], vcc',
                     's_cbranch_execz .Ldqrestore_%=']
        text += lines
        text.append('s_branch .Ldqrestore_%=' if ctl else 's_branch .Ldqend_%=')
    text += ['.Ldqrestore_%=:', 's_mov_b64 exec, %[save]', '.Ldqend_%=:']
    return text


text = '\\n\\t"\n        "'.join(dispatcher())
amps = ', '.join(f'"+v"(a[{j}])' for j in range(NA))
out += ['// id = DqFusedGate::fast (< DQ_FAST32_IDS); g1 = word 1 of the gate record; oc = its outside-control mask;',
        '// tg = the global index bits of this workgroup; tb = the tile-local base of the thread',
        f'#define DQ_FAST32_IDS {NIDS}',
        '__device__ __forceinline__ void fast_dispatch_f32(V2F (&a)[16], const uint64_t (&mq)[4], uint32_t m00, uint32_t m00i,',
        '                                                  uint32_t m11, uint32_t id, uint32_t g1, uint64_t oc, uint64_t tg,',
        '                                                  uint32_t tb, float& hr, float& hi) {',
        '    V2F t0, u0, t1, u1;', '    uint32_t tt, st;', '    uint64_t save;',
        f'    asm volatile(\n        "{text}"',
        f'        : {amps}, "=&v"(t0), "=&v"(u0), "=&v"(t1), "=&v"(u1), [tt] "=&v"(tt), [save] "=&s"(save), [st] "=&s"(st), [hr] "+v"(hr), [hi] "+v"(hi)',
        '        : "s"(mq[0]), "s"(mq[1]), "s"(mq[2]), "s"(mq[3]), [m00] "s"(m00), [m00i] "s"(m00i), [m11] "s"(m11), [id] "s"(id), [g1] "s"(g1), [oc] "s"(oc), [tg] "s"(tg), [tb] "v"(tb)',
        '        : "vcc", "scc");',
        '}', '']
# In the gate loop the amplitudes are pinned to v[40:71] (physical-register constraints): their 32-bit halves can
# then be named, so an X is two v_swap_b32 per amplitude pair instead of three v_mov_b64 through a scratch pair, and
# the 16-byte global loads / stores of two neighbouring amplitudes hit aligned register quads without copies.
AMP0 = 40


# The whole gate loop of a round in ONE asm statement (rounds whose gates all have straight-line handlers; the host
# marks them): per gate  2 scalar loads (record -> s[84:91], matrix -> s[92:99]), the two running offsets, the jump
# through the table, the handler, and the loop test at the end of every handler -- 13 scalar-unit slots and three
# taken branches per gate instead of 18 + 4 with the loop in C++ around fast_dispatch_f32.
REC, MAT = 84, 92


# In-wave exchange of register slot `q` with lane bit `lane` (the wavefront butterfly): afterwards the tile bit that sat
# on the lane bit is a register slot and vice versa -- a change of layout without LDS and without a workgroup barrier.
# The amplitude (slot bit 1, lane bit 0) trades places with (slot bit 0, lane bit 1) of the partner lane, 32-bit
# register by register (A = slot bit 0, B = slot bit 1):
#   lane bit 5 / 4   v_permlane32_swap / v_permlane16_swap A, B: exactly this exchange, one instruction per register
#   lane bit 3 / 2   DPP row shifts by 8 / 4 lanes, the bank mask selects the lanes whose bit is set (A) or clear (B)
#   lane bit 1 / 0   DPP quad permutation + v_cndmask on a lane-parity mask in vcc
# Temporaries: v[72:79] (clobbered by the gate loop).  `s_nop 1` in front: a VALU write of an operand must be two wait
# states old before v_permlane* / a DPP source reads it (nothing pads inside an asm statement).
SWAP_TMP = 72


def swap_lines(q, lane):
    regs = []
    for lo, hi in pairs(q):
        for c in (0, 1):
            regs.append((f'v{AMP0 + 2 * lo + c}', f'v{AMP0 + 2 * hi + c}'))
    if lane >= 4:
        op = 'v_permlane32_swap_b32' if lane == 5 else 'v_permlane16_swap_b32'
        return ['s_nop 1'] + [f'{op} {a_}, {b_}' for a_, b_ in regs]
    tmp = [f'v{SWAP_TMP + i}' for i in range(8)]
    out_ = ['s_nop 1']
    if lane >= 2:
        sh = 1 << lane
        set_, clr = ('0xa', '0x5') if lane == 2 else ('0xc', '0x3')
        for b0 in range(0, 16, 8):
            batch = regs[b0:b0 + 8]
            out_ += [f'v_mov_b32 {tmp[i]}, {a_}' for i, (a_, b_) in enumerate(batch)]
            out_ += [f'v_mov_b32_dpp {a_}, {b_} row_shr:{sh} row_mask:0xf bank_mask:{set_}' for a_, b_ in batch]
            out_ += [f'v_mov_b32_dpp {b_}, {tmp[i]} row_shl:{sh} row_mask:0xf bank_mask:{clr}' for i, (a_, b_) in enumerate(batch)]
        return out_
    qp = '[1,0,3,2]' if lane == 0 else '[2,3,0,1]'
    clear = '0x55555555' if lane == 0 else '0x33333333'          # lanes whose bit is 0
    out_ += [f's_mov_b32 vcc_lo, {clear}', f's_mov_b32 vcc_hi, {clear}']
    for b0 in range(0, 16, 8):
        batch = regs[b0:b0 + 8]
        out_ += [f'v_mov_b32 {tmp[i]}, {a_}' for i, (a_, b_) in enumerate(batch)]
        # D = vcc ? src1 : dpp(src0): lanes with the bit clear keep A, the others take the partner's B
        out_ += [f'v_cndmask_b32_dpp {a_}, {b_}, {a_}, vcc quad_perm:{qp} row_mask:0xf bank_mask:0xf' for a_, b_ in batch]
        out_ += ['s_not_b64 vcc, vcc', 's_nop 1']
        out_ += [f'v_cndmask_b32_dpp {b_}, {tmp[i]}, {b_}, vcc quad_perm:{qp} row_mask:0xf bank_mask:0xf' for i, (a_, b_) in enumerate(batch)]
        out_ += ['s_not_b64 vcc, vcc', 's_nop 1']
    return out_


def gate_loop():
    global MATREGS
    MATREGS = [f's[{MAT + 2 * i}:{MAT + 2 * i + 1}]' for i in range(4)]
    handlers = {}
    for mode in (0, 1, 2, 3):
        for q in range(R):
            lines = body(mode, q)
            if mode == 3:   # the pair (-2, -2) sits in s[80:81] for the whole loop
                lines = [ln.replace(MATREGS[0], 's[80:81]') for ln in lines]
                lines = [f'v_mul_f32 %[hr], %[hr], s{MAT}', f'v_mul_f32 %[hi], %[hi], s{MAT}'] + lines
            if mode == 2:
                lines = [ln.replace('@', 'l') for ln in body_rx_deferred(q, f's[{MAT + 2}:{MAT + 3}]', f's{MAT}', f's{MAT + 1}', f's{MAT + 6}')]
            handlers[4 * mode + q] = (False, lines)
            if mode < 3:
                handlers[20 + 4 * mode + q] = (True, body(mode, q))
    MATREGS = None

    def xlines(q, cmask):
        # three 64-bit moves per amplitude pair through a scratch pair.  (Two v_swap_b32 per pair look cheaper and are
        # not: measured on the headline, 16.87 -> 16.62 ms per pass; with the swaps dropped altogether a pass takes
        # 15.9 ms -- the (C)NOTs of a pass are 1.3 ms of register traffic.)
        out_ = []
        for k, (lo, hi) in enumerate(pairs(q, cmask)):
            t = f'%{T0 if k % 2 == 0 else T1}'
            a_, b_ = f'v[{AMP0 + 2 * lo}:{AMP0 + 2 * lo + 1}]', f'v[{AMP0 + 2 * hi}:{AMP0 + 2 * hi + 1}]'
            out_ += [f'v_mov_b64 {t}, {a_}', f'v_mov_b64 {a_}, {b_}', f'v_mov_b64 {b_}, {t}']
        return out_

    for q in range(R):
        handlers[16 + q] = (False, xlines(q, 0))
        handlers[32 + q] = (True, xlines(q, 0))
        for c in range(R):
            if c != q:
                handlers[36 + 4 * q + c] = (True, xlines(q, 1 << c))
    for q in range(R):
        for lane in range(6):
            handlers[52 + 6 * q + lane] = (False, swap_lines(q, lane))
    # once per round: address of the branch table -> s[82:83], the Hadamard constant -> s[80:81]; per gate the jump is
    # then  vcc = table + 4 * id  (s_lshl2_add_u32 leaves the carry in SCC) ; s_setpc_b64 vcc
    # (requesting gate i + 1's record and matrix into a second register set while gate i runs -- every handler twice,
    # no register copies -- was measured in rounds 1 and 2: 16.78 vs 16.86 ms per pass, inside the noise; the scalar
    # round trip is not on the critical path)
    nxt = ['s_cmp_lt_u32 %[goff], %[gend]', 's_cbranch_scc1 .Ldqloop_%=', 's_branch .Ldqexit_%=']
    text = ['s_getpc_b64 s[82:83]', '.Ldqanchor_%=:',
            's_add_u32 s82, s82, .Ldqtable_%=-.Ldqanchor_%=', 's_addc_u32 s83, s83, 0',
            's_mov_b32 s80, 0xc0000000', 's_mov_b32 s81, 0xc0000000',
            '.Ldqloop_%=:',
            f's_load_dwordx8 s[{REC}:{REC + 7}], %[kg], %[goff]',
            f's_load_dwordx8 s[{MAT}:{MAT + 7}], %[mb], %[moff]',
            's_waitcnt lgkmcnt(0)',
            's_add_u32 %[goff], %[goff], 32',
            f's_lshl3_add_u32 %[moff], s{REC + 6}, %[moff]',
            f's_lshl2_add_u32 vcc_lo, s{REC + 3}, s82', 's_addc_u32 vcc_hi, s83, 0', 's_setpc_b64 vcc',
            '.Ldqtable_%=:']
    for i in range(NIDS):
        text.append(f's_branch .Ldq{i}_%=' if i in handlers else 's_branch .Ldqnext_%=')
    for i in sorted(handlers):
        ctl, lines = handlers[i]
        text.append(f'.Ldq{i}_%=:')
        if ctl:
            text += [f's_and_b64 vcc, s[{REC + 4}:{REC + 5}], %[tg]', f's_cmp_eq_u64 vcc, s[{REC + 4}:{REC + 5}]',
                     's_cbranch_scc0 .Ldqnext_%=',
                     f's_lshr_b32 s{REC + 7}, s{REC + 1}, 16', f'v_and_b32 %[tt], s{REC + 7}, %[tb]',
                     f'v_cmp_eq_u32 vcc, s{REC + 7}, %[tt]', 's_and_saveexec_b64 %[save], vcc',
                     's_cbranch_execz .Ldqrestore_%=']
        text += lines
        if ctl:
            text.append('s_mov_b64 exec, %[save]')
        text += nxt
    text += ['.Ldqrestore_%=:', 's_mov_b64 exec, %[save]', '.Ldqnext_%=:', 's_cmp_lt_u32 %[goff], %[gend]',
             's_cbranch_scc1 .Ldqloop_%=', '.Ldqexit_%=:']
    return text


text = '\\n\\t"\n        "'.join(gate_loop())
pinned = ', '.join(f'"+{{v[{AMP0 + 2 * j}:{AMP0 + 2 * j + 1}]}}"(a[{j}])' for j in range(NA))
clob = ', '.join([f'"s{i}"' for i in range(80, MAT + 8)] + [f'"v{SWAP_TMP + i}"' for i in range(8)])
out += ['// kg + goff = address of the first gate record of the round, gend = offset behind its last one; mb + moff = address',
        '// of the first matrix; both offsets are advanced.  Every gate of the round must have a handler id < DQ_FAST32_IDS.',
        '__device__ __forceinline__ void fast_gate_loop_f32(V2F (&a)[16], uint64_t kg, uint32_t& goff, uint32_t gend, uint64_t mb,',
        '                                                   uint32_t& moff, uint64_t tg, uint32_t tb, float& hr, float& hi) {',
        '    V2F t0, u0, t1, u1;', '    uint32_t tt;', '    uint64_t save;',
        f'    asm volatile(\n        "{text}"',
        f'        : {pinned}, "=&v"(t0), "=&v"(u0), "=&v"(t1), "=&v"(u1), [tt] "=&v"(tt), [save] "=&s"(save), [hr] "+v"(hr), [hi] "+v"(hi), [goff] "+s"(goff), [moff] "+s"(moff)',
        '        : [kg] "s"(kg), [gend] "s"(gend), [mb] "s"(mb), [tg] "s"(tg), [tb] "v"(tb)',
        f'        : "vcc", "scc", {clob});',
        '}', '']
# ---------------------------------------------------------------------------------------------------
# complex128, R = 3 (8 amplitudes per thread): operands are the re / im doubles of every amplitude (a 64-bit
# VGPR pair each), the 8 real numbers of the matrix are SGPR pairs (one scalar source per VOP3, as gfx9 allows).
# Every new component is  [three terms that read OTHER components, summed into a temporary]  +  the term in its
# own old value, so the final v_fma_f64 of a chain may overwrite in place: no register copies at all.
R64 = 3
NA64 = 1 << R64


def pairs64(q, cmask=0):
    return [(j, j | (1 << q)) for j in range(NA64) if not (j >> q) & 1 and (j & cmask) == cmask]


S64 = 2 * NA64 + 4                      # first matrix operand of the per-handler functions
MATREGS64 = None                        # fixed SGPR pairs instead of operands (gate loop in assembly)


def body64(mode, q):
    TMP = 2 * NA64                      # t0..t3
    S = S64                             # m00r m00i m01r m01i m10r m10i m11r m11i
    m = {name: (MATREGS64[i] if MATREGS64 else f'%{S + i}')
         for i, name in enumerate(['00r', '00i', '01r', '01i', '10r', '10i', '11r', '11i'])}
    t = [f'%{TMP + i}' for i in range(4)]
    lines = []
    for lo, hi in pairs64(q):
        ar, ai, br, bi = f'%{2 * lo}', f'%{2 * lo + 1}', f'%{2 * hi}', f'%{2 * hi + 1}'
        if mode == 3:      # Hadamard-like, factor deferred: A' = A + B, B' = A' - 2 B  (m["00r"] holds -2.0)
            nsum = sum(1 for ln in lines if ln.startswith('v_add_f64'))
            lines[nsum:nsum] = [f'v_add_f64 {ar}, {ar}, {br}', f'v_add_f64 {ai}, {ai}, {bi}']
            lines += [f'v_fma_f64 {br}, {m["00r"]}, {br}, {ar}', f'v_fma_f64 {bi}, {m["00r"]}, {bi}, {ai}']
        elif mode == 1:      # real matrix: re and im parts transform independently
            lines += [f'v_mul_f64 {t[0]}, {m["01r"]}, {br}', f'v_mul_f64 {t[1]}, {m["01r"]}, {bi}',
                      f'v_mul_f64 {t[2]}, {m["10r"]}, {ar}', f'v_mul_f64 {t[3]}, {m["10r"]}, {ai}',
                      f'v_fma_f64 {ar}, {m["00r"]}, {ar}, {t[0]}', f'v_fma_f64 {ai}, {m["00r"]}, {ai}, {t[1]}',
                      f'v_fma_f64 {br}, {m["11r"]}, {br}, {t[2]}', f'v_fma_f64 {bi}, {m["11r"]}, {bi}, {t[3]}']
        elif mode == 2:    # real diagonal, imaginary off-diagonal: (i s)(x + i y) = -s y + i s x
            lines += [f'v_mul_f64 {t[0]}, -{m["01i"]}, {bi}', f'v_mul_f64 {t[1]}, {m["01i"]}, {br}',
                      f'v_mul_f64 {t[2]}, -{m["10i"]}, {ai}', f'v_mul_f64 {t[3]}, {m["10i"]}, {ar}',
                      f'v_fma_f64 {ar}, {m["00r"]}, {ar}, {t[0]}', f'v_fma_f64 {ai}, {m["00r"]}, {ai}, {t[1]}',
                      f'v_fma_f64 {br}, {m["11r"]}, {br}, {t[2]}', f'v_fma_f64 {bi}, {m["11r"]}, {bi}, {t[3]}']
        else:              # general complex 2x2
            lines += [f'v_mul_f64 {t[0]}, -{m["00i"]}, {ai}', f'v_mul_f64 {t[1]}, {m["00i"]}, {ar}',
                      f'v_mul_f64 {t[2]}, {m["10r"]}, {ar}', f'v_mul_f64 {t[3]}, {m["10r"]}, {ai}',
                      f'v_fma_f64 {t[0]}, {m["01r"]}, {br}, {t[0]}', f'v_fma_f64 {t[1]}, {m["01r"]}, {bi}, {t[1]}',
                      f'v_fma_f64 {t[2]}, -{m["10i"]}, {ai}, {t[2]}', f'v_fma_f64 {t[3]}, {m["10i"]}, {ar}, {t[3]}',
                      f'v_fma_f64 {t[0]}, -{m["01i"]}, {bi}, {t[0]}', f'v_fma_f64 {t[1]}, {m["01i"]}, {br}, {t[1]}',
                      f'v_fma_f64 {t[2]}, -{m["11i"]}, {bi}, {t[2]}', f'v_fma_f64 {t[3]}, {m["11i"]}, {br}, {t[3]}',
                      f'v_fma_f64 {ar}, {m["00r"]}, {ar}, {t[0]}', f'v_fma_f64 {ai}, {m["00r"]}, {ai}, {t[1]}',
                      f'v_fma_f64 {br}, {m["11r"]}, {br}, {t[2]}', f'v_fma_f64 {bi}, {m["11r"]}, {bi}, {t[3]}']
    return lines


out += ['using V2D = vec2<double>;', '']
amp_ops = ', '.join(f'"+v"(a[{j}].x), "+v"(a[{j}].y)' for j in range(NA64))
for mode in (0, 1, 2, 3):
    for q in range(R64):
        text = '\\n\\t"\n        "'.join(body64(mode, q))
        ins = ', '.join(f'"s"(md[{i}])' for i in range(8))
        out += [f'template <> __device__ __forceinline__ void gen1_block_f64<{mode}, {q}>(V2D (&a)[8], const double (&md)[8]) {{',
                '    double t0, t1, t2, t3;',
                f'    asm volatile(\n        "{text}"',
                f'        : {amp_ops}, "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)',
                f'        : {ins});',
                '}', '']
for q in range(R64):
    for cmask in range(NA64):
        if (cmask >> q) & 1:
            continue
        lines = []
        for lo, hi in pairs64(q, cmask):
            for c in (0, 1):
                x, y, t = f'%{2 * lo + c}', f'%{2 * hi + c}', f'%{2 * NA64 + c}'
                lines += [f'v_mov_b64 {t}, {x}', f'v_mov_b64 {x}, {y}', f'v_mov_b64 {y}, {t}']
        text = '\\n\\t"\n        "'.join(lines)
        out += [f'template <> __device__ __forceinline__ void x1_block_f64<{q}, {cmask}>(V2D (&a)[8]) {{',
                '    double t0, t1;',
                f'    asm volatile(\n        "{text}"',
                f'        : {amp_ops}, "=&v"(t0), "=&v"(t1));',
                '}', '']


# complex128 dispatcher: same jump table as complex64 (ids with slot or control slot 3 do not exist here)
def dispatcher64():
    global S64
    keep, S64 = S64, 2 * NA64 + 8      # a (16 doubles), t0..t3, tt, save, st, hs, then the 8 matrix doubles
    handlers = {}
    for mode in (0, 1, 2, 3):
        for q in range(R64):
            lines = body64(mode, q)
            if mode == 3:   # Hadamard: -2.0 is an inline constant in double precision; the factor goes into hs
                lines = [ln.replace(f'%{S64},', '-2.0,') for ln in lines]
                lines = [f'v_mul_f64 %[hs], %[hs], %{S64}'] + lines
            handlers[4 * mode + q] = (False, lines)
            if mode < 3:
                handlers[20 + 4 * mode + q] = (True, body64(mode, q))

    def xlines(q, cmask):
        out_ = []
        for lo, hi in pairs64(q, cmask):
            for c in (0, 1):
                x, y, t = f'%{2 * lo + c}', f'%{2 * hi + c}', f'%{2 * NA64 + c}'
                out_ += [f'v_mov_b64 {t}, {x}', f'v_mov_b64 {x}, {y}', f'v_mov_b64 {y}, {t}']
        return out_

    for q in range(R64):
        handlers[16 + q] = (False, xlines(q, 0))
        handlers[32 + q] = (True, xlines(q, 0))
        for c in range(R64):
            if c != q:
                handlers[36 + 4 * q + c] = (True, xlines(q, 1 << c))
    S64 = keep
    text = ['s_lshl2_add_u32 %[st], %[id], 12', 's_getpc_b64 vcc', 's_add_u32 vcc_lo, vcc_lo, %[st]',
            's_addc_u32 vcc_hi, vcc_hi, 0', 's_setpc_b64 vcc']
    for i in range(NIDS):
        text.append(f's_branch .Ldq{i}_%=' if i in handlers else 's_branch .Ldqend_%=')
    for i in sorted(handlers):
        ctl, lines = handlers[i]
        text.append(f'.Ldq{i}_%=:')
        if ctl:
            text += ['s_and_b64 vcc, %[oc], %[tg]', 's_cmp_eq_u64 vcc, %[oc]', 's_cbranch_scc0 .Ldqend_%=',
                     's_lshr_b32 %[st], %[g1], 16', 'v_and_b32 %[tt], %[st], %[tb]',
                     'v_cmp_eq_u32 vcc, %[st], %[tt]', 's_and_saveexec_b64 %[save], vcc',
                     's_cbranch_execz .Ldqrestore_%=']
        text += lines
        text.append('s_branch .Ldqrestore_%=' if ctl else 's_branch .Ldqend_%=')
    text += ['.Ldqrestore_%=:', 's_mov_b64 exec, %[save]', '.Ldqend_%=:']
    return text


text = '\\n\\t"\n        "'.join(dispatcher64())
ins = ', '.join(f'"s"(md[{i}])' for i in range(8))
out += ['__device__ __forceinline__ void fast_dispatch_f64(V2D (&a)[8], const double (&md)[8], uint32_t id, uint32_t g1,',
        '                                                  uint64_t oc, uint64_t tg, uint32_t tb, double& hs) {',
        '    double t0, t1, t2, t3;', '    uint32_t tt, st;', '    uint64_t save;',
        f'    asm volatile(\n        "{text}"',
        f'        : {amp_ops}, "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), [tt] "=&v"(tt), [save] "=&s"(save), [st] "=&s"(st), [hs] "+v"(hs)',
        f'        : {ins}, [id] "s"(id), [g1] "s"(g1), [oc] "s"(oc), [tg] "s"(tg), [tb] "v"(tb)',
        '        : "vcc", "scc");',
        '}', '']


# complex128 gate loop in assembly: record -> s[76:83], matrix (8 doubles) -> s[84:99]
REC64, MAT64 = 76, 84


def gate_loop64():
    global MATREGS64
    MATREGS64 = [f's[{MAT64 + 2 * i}:{MAT64 + 2 * i + 1}]' for i in range(8)]
    handlers = {}
    for mode in (0, 1, 2, 3):
        for q in range(R64):
            lines = body64(mode, q)
            if mode == 3:
                lines = [ln.replace(f'{MATREGS64[0]},', '-2.0,') for ln in lines]
                lines = [f'v_mul_f64 %[hs], %[hs], {MATREGS64[0]}'] + lines
            handlers[4 * mode + q] = (False, lines)
            if mode < 3:
                handlers[20 + 4 * mode + q] = (True, body64(mode, q))
    MATREGS64 = None

    def xlines(q, cmask):
        out_ = []
        for lo, hi in pairs64(q, cmask):
            for c in (0, 1):
                x, y, t = f'%{2 * lo + c}', f'%{2 * hi + c}', f'%{2 * NA64 + c}'
                out_ += [f'v_mov_b64 {t}, {x}', f'v_mov_b64 {x}, {y}', f'v_mov_b64 {y}, {t}']
        return out_

    for q in range(R64):
        handlers[16 + q] = (False, xlines(q, 0))
        handlers[32 + q] = (True, xlines(q, 0))
        for c in range(R64):
            if c != q:
                handlers[36 + 4 * q + c] = (True, xlines(q, 1 << c))
    R_ = REC64
    nxt = ['s_cmp_lt_u32 %[goff], %[gend]', 's_cbranch_scc1 .Ldqloop_%=', 's_branch .Ldqexit_%=']
    text = ['s_getpc_b64 s[74:75]', '.Ldqanchor_%=:',
            's_add_u32 s74, s74, .Ldqtable_%=-.Ldqanchor_%=', 's_addc_u32 s75, s75, 0',
            '.Ldqloop_%=:',
            f's_load_dwordx8 s[{R_}:{R_ + 7}], %[kg], %[goff]',
            f's_load_dwordx16 s[{MAT64}:{MAT64 + 15}], %[mb], %[moff]',
            's_waitcnt lgkmcnt(0)',
            's_add_u32 %[goff], %[goff], 32',
            f's_lshl4_add_u32 %[moff], s{R_ + 6}, %[moff]',
            f's_lshl2_add_u32 vcc_lo, s{R_ + 3}, s74', 's_addc_u32 vcc_hi, s75, 0', 's_setpc_b64 vcc',
            '.Ldqtable_%=:']
    for i in range(NIDS):
        text.append(f's_branch .Ldq{i}_%=' if i in handlers else 's_branch .Ldqnext_%=')
    for i in sorted(handlers):
        ctl, lines = handlers[i]
        text.append(f'.Ldq{i}_%=:')
        if ctl:
            text += [f's_and_b64 vcc, s[{R_ + 4}:{R_ + 5}], %[tg]', f's_cmp_eq_u64 vcc, s[{R_ + 4}:{R_ + 5}]',
                     's_cbranch_scc0 .Ldqnext_%=',
                     f's_lshr_b32 s{R_ + 7}, s{R_ + 1}, 16', f'v_and_b32 %[tt], s{R_ + 7}, %[tb]',
                     f'v_cmp_eq_u32 vcc, s{R_ + 7}, %[tt]', 's_and_saveexec_b64 %[save], vcc',
                     's_cbranch_execz .Ldqrestore_%=']
        text += lines
        if ctl:
            text.append('s_mov_b64 exec, %[save]')
        text += nxt
    text += ['.Ldqrestore_%=:', 's_mov_b64 exec, %[save]', '.Ldqnext_%=:', 's_cmp_lt_u32 %[goff], %[gend]',
             's_cbranch_scc1 .Ldqloop_%=', '.Ldqexit_%=:']
    return text


text = '\\n\\t"\n        "'.join(gate_loop64())
clob = ', '.join(f'"s{i}"' for i in range(74, MAT64 + 16))
out += ['__device__ __forceinline__ void fast_gate_loop_f64(V2D (&a)[8], uint64_t kg, uint32_t& goff, uint32_t gend, uint64_t mb,',
        '                                                   uint32_t& moff, uint64_t tg, uint32_t tb, double& hs) {',
        '    double t0, t1, t2, t3;', '    uint32_t tt;', '    uint64_t save;',
        f'    asm volatile(\n        "{text}"',
        f'        : {amp_ops}, "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), [tt] "=&v"(tt), [save] "=&s"(save), [hs] "+v"(hs), [goff] "+s"(goff), [moff] "+s"(moff)',
        '        : [kg] "s"(kg), [gend] "s"(gend), [mb] "s"(mb), [tg] "s"(tg), [tb] "v"(tb)',
        f'        : "vcc", "scc", {clob});',
        '}', '']
out += ['// clang-format on', '']
import os as _os
if _os.environ.get('DQ_ABLATE_GATES'):   # timing experiments only (tools/ablate.sh): gate bodies without their VALU work
    drop = ('"v_pk_', '"v_swap_b32', '"v_mov_b64', '"v_mul_f32 %[hr]', '"v_mul_f32 %[hi]', '"v_mul_f64', '"v_fma_f64', '"v_add_f64')
    out = '\n'.join(out).split('\n')
    out = [(ln[:len(ln) - len(ln.lstrip())] + '""') if ln.strip().startswith(drop) else ln for ln in out]
open(_os.environ.get('DQ_ASM_OUT') or _os.path.join(_os.path.dirname(__file__), '..', 'deepquantum_amd', 'csrc', 'dq_fused_asm.inc'), 'w').write('\n'.join(out))
print('generated', len(out), 'lines')
