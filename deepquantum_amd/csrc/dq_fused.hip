// Fused gate pass for gfx950: one HBM read + one HBM write of the statevector applies a whole group
// of 1- and 2-qubit gates.  Replaces a run of consecutive Gate.forward calls of the reference
// (circuit.py:261 -> operation.py:274-289 -> qmath.py:485-506 / operation.py:203-219), each of which
// costs the reference >= 2 full read+write passes.
//
// Geometry.  A workgroup of 2^LOGT threads owns one tile of 2^M amplitudes, M = R + LOGT: every
// thread keeps 2^R amplitudes in VGPRs ("register slots").  The M tile bits are the low L index bits
// (contiguous -> every wave-level load/store instruction moves >= 512 contiguous bytes) plus
// h = M - L gathered high bits chosen per pass by the host scheduler.  A pass is a list of rounds; a
// round names which R tile bits are register slots, and its gates act on register slots only, so a
// gate costs VALU work but no memory traffic of any kind.  Between rounds the tile is re-distributed
// through LDS (one write + one read, XOR-swizzled).  Control qubits never need to be slots: a
// control on a thread bit is a per-lane predicate, a control outside the tile is a workgroup-uniform
// branch.  Diagonal gates act wherever their qubit happens to be.
//
// The first and last layouts are the "canonical" one used for global I/O: c64 keeps tile bit 0 and
// the top R-1 tile bits as slots (so a lane moves 16 B per instruction and a wave 1 KiB), c128 the
// top R tile bits.  A pass whose gates only touch those bits runs with no LDS traffic at all.
//
// Roofline: HBM-bound by construction.  Algorithmic bytes per launch = sum over the gates of the
// pass of 2 * 2^(n - nc) * sizeof(amp) * batch; actual traffic = 2 * 2^n * sizeof(amp) * batch.
#include "dq_common.hpp"
#include <type_traits>
#include <atomic>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef DQ_USE_ASM_BLOCKS
#define DQ_USE_ASM_BLOCKS 1
#endif

namespace dq {

// Amplitudes live in registers as 2-element native vectors: for complex64 that is one aligned 64-bit VGPR
// pair, the operand shape of the packed VALU ops and of the tied inline-asm operands below.
template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));
template <typename T> using amp = vec2<T>;
// Gate matrices are read through the CONSTANT address space: nothing in a pass writes them, and a constant pointer with
// a workgroup-uniform address is read by scalar loads (lgkmcnt).  As plain global pointers hipcc reads them with vector
// loads, whose s_waitcnt vmcnt(..) would also wait for the next tile's prefetch (one in-order counter).
template <typename T> using cmat = const __attribute__((address_space(4))) amp<T>*;
__device__ __forceinline__ uint64_t as_u64(const vec2<float> v) { return __builtin_bit_cast(uint64_t, v); }
__device__ __forceinline__ uint64_t as_u64(const vec2<double>) { return 0; }   // (never used: the f32 blocks only)

// ---- gate bodies on the register file ---------------------------------------------------------------
// MODE 0: general complex 2x2.  MODE 1: all four entries real (H, Ry, X, ...).  MODE 2: real diagonal,
// purely imaginary off-diagonal (Rx).  The mode is picked at run time from the (workgroup-uniform)
// matrix values, so multiplications by exact zeros are skipped: half the VALU work for H / Rx / Ry.
// Packed formulation: an amplitude (re, im) is one 2-vector, so every line below is one packed VALU op on
// gfx950 (v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers for the swizzled operand); for double
// the same source scalarises to v_fma_f64.

template <typename T, int MODE>
__device__ __forceinline__ void apply2x2(amp<T>& x0, amp<T>& x1, const amp<T> m00, const amp<T> m01, const amp<T> m10,
                                         const amp<T> m11) {
    using V2 = vec2<T>;
    const V2 a = {x0.x, x0.y}, b = {x1.x, x1.y};
    V2 r0, r1;
    if constexpr (MODE == 1) {
        // scalar source form: hipcc's SLP vectoriser turns each (x, y) pair into one packed op and keeps
        // the matrix entries as SGPR operands (32 packed VALU ops per 16 amplitudes, measured)
        amp<T> n0, n1;
        n0.x = fma(m01.x, x1.x, m00.x * x0.x);
        n0.y = fma(m01.x, x1.y, m00.x * x0.y);
        n1.x = fma(m11.x, x1.x, m10.x * x0.x);
        n1.y = fma(m11.x, x1.y, m10.x * x0.y);
        x0 = n0;
        x1 = n1;
        return;
    } else if constexpr (MODE == 2) {
        // i*y*(b.x + i b.y) = (-y b.y) + i (y b.x)
        amp<T> n0, n1;
        n0.x = fma(-m01.y, x1.y, m00.x * x0.x);
        n0.y = fma(m01.y, x1.x, m00.x * x0.y);
        n1.x = fma(-m10.y, x0.y, m11.x * x1.x);
        n1.y = fma(m10.y, x0.x, m11.x * x1.y);
        x0 = n0;
        x1 = n1;
        return;
    } else {
        const V2 bs = {-b.y, b.x}, as = {-a.y, a.x};
        V2 t = m01.x * b;
        t = __builtin_elementwise_fma((V2)(m01.y), bs, t);
        t = __builtin_elementwise_fma((V2)(m00.y), as, t);
        r0 = __builtin_elementwise_fma((V2)(m00.x), a, t);
        V2 u = m10.x * a;
        u = __builtin_elementwise_fma((V2)(m10.y), as, u);
        u = __builtin_elementwise_fma((V2)(m11.y), bs, u);
        r1 = __builtin_elementwise_fma((V2)(m11.x), b, u);
    }
    x0.x = r0.x;
    x0.y = r0.y;
    x1.x = r1.x;
    x1.y = r1.y;
}

// ---- complex64: straight-line inline-asm blocks (generated, csrc/dq_fused_asm.inc) -------------------
// One asm statement per (matrix structure, target slot) updates all 16 register-resident amplitudes in
// place: every amplitude is a tied "+v" 64-bit operand, matrix entries are SGPR pairs whose low half is
// broadcast (op_sel_hi 0), the "times i" swizzle (-im, re) is op_sel:[1,..] op_sel_hi:[0,..] + neg_lo.
// mq[] = the four matrix entries as raw 64-bit SGPR pairs (low half re, high half im).
template <int MODE, int Q> __device__ __forceinline__ void gen1_block_f32(vec2<float> (&a)[16], const uint64_t (&mq)[4]);
template <int Q, int CMASK> __device__ __forceinline__ void x1_block_f32(vec2<float> (&a)[16]);
// ---- complex128 (3 register slots = 8 amplitudes per thread): v_mul_f64 / v_fma_f64 chains on the re / im
// doubles, the 8 real numbers of the matrix as SGPR pairs; in-place final FMA, no register copies.
template <int MODE, int Q> __device__ __forceinline__ void gen1_block_f64(vec2<double> (&a)[8], const double (&md)[8]);
template <int Q, int CMASK> __device__ __forceinline__ void x1_block_f64(vec2<double> (&a)[8]);
#ifdef DQ_ASM_INC          // timing experiments (tools/ablate.sh): an alternative generated file
#include DQ_ASM_INC
#else
#include "dq_fused_asm.inc"
#endif

template <int MODE>
__device__ __forceinline__ void dispatch_gen1_block_f32(vec2<float> (&a)[16], int q, const uint64_t (&mq)[4]) {
    switch (q) {
        case 0: gen1_block_f32<MODE, 0>(a, mq); break;
        case 1: gen1_block_f32<MODE, 1>(a, mq); break;
        case 2: gen1_block_f32<MODE, 2>(a, mq); break;
        default: gen1_block_f32<MODE, 3>(a, mq); break;
    }
}

template <typename T, int R, int Q, int MODE, bool PRED>
__device__ __forceinline__ void gen1_body(amp<T> (&a)[1 << R], const amp<T> m00, const amp<T> m01, const amp<T> m10,
                                          const amp<T> m11, const unsigned reg_cmask, const bool thr_ok) {
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
        if ((j >> Q) & 1) continue;
        if constexpr (PRED) {
            if (thr_ok && ((j & reg_cmask) == reg_cmask)) apply2x2<T, MODE>(a[j], a[j | (1 << Q)], m00, m01, m10, m11);
        } else {
            apply2x2<T, MODE>(a[j], a[j | (1 << Q)], m00, m01, m10, m11);
        }
    }
}

// Register swap (C++ path, used for complex128): plain moves; the enclosing control is uniform or an
// exec-masked region.
template <typename V> __device__ __forceinline__ void swap_amp(V& a, V& b) {
    const V t = a;
    a = b;
    b = t;
}

template <typename T, int R, int Q>
__device__ __forceinline__ void x1_body(amp<T> (&a)[1 << R], const unsigned reg_cmask) {
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
        if ((j >> Q) & 1) continue;
        if ((j & reg_cmask) == reg_cmask) swap_amp(a[j], a[j | (1 << Q)]);
    }
}

template <typename T, int R, int Q, int Q2, typename MP>
__device__ __forceinline__ void gen2_body(amp<T> (&a)[1 << R], MP mp, const unsigned reg_cmask,
                                          const bool thr_ok) {
    amp<T> m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = mp[i];
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
        if (((j >> Q) & 1) || ((j >> Q2) & 1)) continue;
        if (thr_ok && ((j & reg_cmask) == reg_cmask)) {
            // matrix index = (bit of slot Q) * 2 + (bit of slot Q2)
            const int i0 = j, i1 = j | (1 << Q2), i2 = j | (1 << Q), i3 = j | (1 << Q) | (1 << Q2);
            const amp<T> x0 = a[i0], x1 = a[i1], x2 = a[i2], x3 = a[i3];
            a[i0] = cfma(m[3], x3, cfma(m[2], x2, cfma(m[1], x1, cmul(m[0], x0))));
            a[i1] = cfma(m[7], x3, cfma(m[6], x2, cfma(m[5], x1, cmul(m[4], x0))));
            a[i2] = cfma(m[11], x3, cfma(m[10], x2, cfma(m[9], x1, cmul(m[8], x0))));
            a[i3] = cfma(m[15], x3, cfma(m[14], x2, cfma(m[13], x1, cmul(m[12], x0))));
        }
    }
}

// 4x4 matrix promised REAL by the host (DqFusedMode REAL on a GEN2 gate: the superoperators of the noise
// channels): entry by entry, skipping the exact zeros with a uniform branch -- a depolarizing channel has 6
// non-zero entries of 16, amplitude damping 5 -- and one real-times-complex FMA per entry and amplitude group.
template <typename T, int R, int Q, int Q2, typename MP>
__device__ __forceinline__ void gen2_body_real(amp<T> (&a)[1 << R], MP mp,
                                               const unsigned reg_cmask, const bool thr_ok) {
    constexpr int NG = (1 << R) / 4;
    int base[NG];
    {
        int g = 0;
#pragma unroll
        for (int j = 0; j < (1 << R); ++j)
            if (!((j >> Q) & 1) && !((j >> Q2) & 1)) base[g++] = j;
    }
    constexpr int off[4] = {0, 1 << Q2, 1 << Q, (1 << Q) | (1 << Q2)};
    amp<T> y[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[g][r] = amp<T>{0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T mr = mp[r * 4 + c].x;
            if (mr != T(0)) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const amp<T> x = a[base[g] + off[c]];
                    y[g][r].x = fma(mr, x.x, y[g][r].x);
                    y[g][r].y = fma(mr, x.y, y[g][r].y);
                }
            }
        }
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (thr_ok && ((base[g] & reg_cmask) == reg_cmask)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[base[g] + off[r]] = y[g][r];
        }
}

template <typename T, int R, int MODE, bool PRED>
__device__ __forceinline__ void dispatch_gen1_q(amp<T> (&a)[1 << R], int q, const amp<T> m00, const amp<T> m01,
                                                const amp<T> m10, const amp<T> m11, unsigned reg_cmask, bool thr_ok) {
    switch (q) {
        case 0: gen1_body<T, R, 0, MODE, PRED>(a, m00, m01, m10, m11, reg_cmask, thr_ok); break;
        case 1: gen1_body<T, R, 1, MODE, PRED>(a, m00, m01, m10, m11, reg_cmask, thr_ok); break;
        case 2: gen1_body<T, R, 2, MODE, PRED>(a, m00, m01, m10, m11, reg_cmask, thr_ok); break;
        default:
            if constexpr (R > 3) gen1_body<T, R, 3, MODE, PRED>(a, m00, m01, m10, m11, reg_cmask, thr_ok);
            break;
    }
}

// `mode` = matrix structure promised by the host from the gate class (DqFusedMode): 0 general, 1 all
// entries real (H, Ry, X...), 2 real diagonal + imaginary off-diagonal (Rx).  Multiplications by the exact
// zeros are skipped: half the VALU work for the common gates.
template <typename T, int R, typename MP>
__device__ __forceinline__ void dispatch_gen1(amp<T> (&a)[1 << R], int q, MP mp,
                                              unsigned mode, unsigned reg_cmask, bool lane_pred, bool thr_ok) {
    if constexpr (sizeof(T) == 4 && R == 4 && DQ_USE_ASM_BLOCKS) {
        if (reg_cmask == 0) {
            // uncontrolled (or controlled only by thread / outside bits): straight-line asm block; a control
            // on a thread bit is ONE exec-masked region around it (asm is never if-converted)
            if (!lane_pred || thr_ok) {
                const uint64_t mq[4] = {as_u64(mp[0]), as_u64(mp[1]), as_u64(mp[2]), as_u64(mp[3])};
                if (mode == 1) dispatch_gen1_block_f32<1>(a, q, mq);
                else if (mode == 2) dispatch_gen1_block_f32<2>(a, q, mq);
                else dispatch_gen1_block_f32<0>(a, q, mq);
            }
            return;
        }
    }
    const amp<T> m00 = mp[0], m01 = mp[1], m10 = mp[2], m11 = mp[3];
    if (lane_pred || reg_cmask) {  // controlled gate: one predicated general body (code size)
        dispatch_gen1_q<T, R, 0, true>(a, q, m00, m01, m10, m11, reg_cmask, thr_ok);
    } else if (mode == 1) {
        dispatch_gen1_q<T, R, 1, false>(a, q, m00, m01, m10, m11, 0u, true);
    } else if (mode == 2) {
        dispatch_gen1_q<T, R, 2, false>(a, q, m00, m01, m10, m11, 0u, true);
    } else {
        dispatch_gen1_q<T, R, 0, false>(a, q, m00, m01, m10, m11, 0u, true);
    }
}

template <int Q>
__device__ __forceinline__ void dispatch_x1_block_f32(vec2<float> (&a)[16], unsigned cmask) {
    switch (cmask) {  // cmask never contains bit Q (host guarantees target != control)
        case 0: x1_block_f32<Q, 0>(a); break;
#define DQ_X1_CASE(C) case C: if constexpr (!((C >> Q) & 1)) x1_block_f32<Q, C>(a); break;
        DQ_X1_CASE(1) DQ_X1_CASE(2) DQ_X1_CASE(3) DQ_X1_CASE(4) DQ_X1_CASE(5) DQ_X1_CASE(6) DQ_X1_CASE(7)
        DQ_X1_CASE(8) DQ_X1_CASE(9) DQ_X1_CASE(10) DQ_X1_CASE(11) DQ_X1_CASE(12) DQ_X1_CASE(13) DQ_X1_CASE(14)
#undef DQ_X1_CASE
        default: break;
    }
}

template <int Q>
__device__ __forceinline__ void dispatch_x1_block_f64(vec2<double> (&a)[8], unsigned cmask) {
    switch (cmask) {
        case 0: x1_block_f64<Q, 0>(a); break;
#define DQ_X1_CASE(C) case C: if constexpr (!((C >> Q) & 1)) x1_block_f64<Q, C>(a); break;
        DQ_X1_CASE(1) DQ_X1_CASE(2) DQ_X1_CASE(3) DQ_X1_CASE(4) DQ_X1_CASE(5) DQ_X1_CASE(6)
#undef DQ_X1_CASE
        default: break;
    }
}

template <typename T, int R>
__device__ __forceinline__ void dispatch_x1(amp<T> (&a)[1 << R], int q, unsigned reg_cmask, bool lane_pred, bool thr_ok) {
    if (lane_pred && !thr_ok) return;  // per-lane control: exec-masked region around the swaps
    if constexpr (sizeof(T) == 4 && R == 4 && DQ_USE_ASM_BLOCKS) {
        switch (q) {
            case 0: dispatch_x1_block_f32<0>(a, reg_cmask); break;
            case 1: dispatch_x1_block_f32<1>(a, reg_cmask); break;
            case 2: dispatch_x1_block_f32<2>(a, reg_cmask); break;
            default: dispatch_x1_block_f32<3>(a, reg_cmask); break;
        }
    } else {
        switch (q) {
            case 0: x1_body<T, R, 0>(a, reg_cmask); break;
            case 1: x1_body<T, R, 1>(a, reg_cmask); break;
            case 2: x1_body<T, R, 2>(a, reg_cmask); break;
            default:
                if constexpr (R > 3) x1_body<T, R, 3>(a, reg_cmask);
                break;
        }
    }
}

template <typename T, int R, int Q, int Q2, typename MP>
__device__ __forceinline__ void gen2_any(amp<T> (&a)[1 << R], MP mp, unsigned mode,
                                         unsigned reg_cmask, bool thr_ok) {
    if (mode == DQ_MODE_REAL) gen2_body_real<T, R, Q, Q2>(a, mp, reg_cmask, thr_ok);
    else gen2_body<T, R, Q, Q2>(a, mp, reg_cmask, thr_ok);
}

template <typename T, int R, int Q, typename MP>
__device__ __forceinline__ void dispatch_gen2_q2(amp<T> (&a)[1 << R], int q2, MP mp,
                                                 unsigned mode, unsigned reg_cmask, bool thr_ok) {
    switch (q2) {
        case 0:
            if constexpr (Q != 0) gen2_any<T, R, Q, 0>(a, mp, mode, reg_cmask, thr_ok);
            break;
        case 1:
            if constexpr (Q != 1) gen2_any<T, R, Q, 1>(a, mp, mode, reg_cmask, thr_ok);
            break;
        case 2:
            if constexpr (Q != 2) gen2_any<T, R, Q, 2>(a, mp, mode, reg_cmask, thr_ok);
            break;
        default:
            if constexpr (R > 3 && Q != 3) gen2_any<T, R, Q, 3>(a, mp, mode, reg_cmask, thr_ok);
            break;
    }
}

template <typename T, int R, typename MP>
__device__ __forceinline__ void dispatch_gen2(amp<T> (&a)[1 << R], int q, int q2, MP mp,
                                              unsigned mode, unsigned reg_cmask, bool thr_ok) {
    switch (q) {
        case 0: dispatch_gen2_q2<T, R, 0>(a, q2, mp, mode, reg_cmask, thr_ok); break;
        case 1: dispatch_gen2_q2<T, R, 1>(a, q2, mp, mode, reg_cmask, thr_ok); break;
        case 2: dispatch_gen2_q2<T, R, 2>(a, q2, mp, mode, reg_cmask, thr_ok); break;
        default:
            if constexpr (R > 3) dispatch_gen2_q2<T, R, 3>(a, q2, mp, mode, reg_cmask, thr_ok);
            break;
    }
}

// ---- DQ_FG_GRAD: sum lambda (x) conj(psi) over a thread's registers, the wavefront, into LDS -----------------
// Eight per-lane sums g0..g7 -> for every row of 16 lanes, eight lanes that each hold ONE of the sums added up over the
// row: a reduce-scatter over lane bits 2, 3 and 0 (every step halves the values a lane is responsible for) and a plain
// add over lane bit 1 -- 16 DPP adds instead of the 32 a reduction of all eight values over a row takes, none of the
// hazard no-ops and register copies hipcc puts around __builtin_amdgcn_update_dpp.  The lane with bits (b0, b3, b2)
// ends up with sum number 4 b0 + 2 b3 + b2; lanes that differ in bit 1 hold the same value.
__device__ __forceinline__ float row_reduce_scatter8(float g0, float g1, float g2, float g3, float g4, float g5, float g6,
                                                     float g7) {
    float t0, t1;
    asm volatile(
        // lane bit 2: lanes with the bit clear take over the even-numbered sum of each pair, the others the odd one
        "v_add_f32_dpp %[g0], %[g0], %[g0] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[g2], %[g2], %[g2] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[g4], %[g4], %[g4] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[g6], %[g6], %[g6] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[g0], %[g1], %[g1] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[g2], %[g3], %[g3] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[g4], %[g5], %[g5] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[g6], %[g7], %[g7] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        // lane bit 3: (g0, g2) -> g0, (g4, g6) -> g4
        "v_add_f32_dpp %[g0], %[g0], %[g0] row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[g4], %[g4], %[g4] row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[g0], %[g2], %[g2] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[g4], %[g6], %[g6] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        // lane bit 0: (g0, g4) -> even lanes keep g0, odd lanes g4
        "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\t"
        "s_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[t0], %[g0], %[g0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[t1], %[g4], %[g4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %[g0], %[t0], %[t1], vcc\n\t"
        // lane bit 1: plain add
        "s_nop 1\n\t"
        "v_add_f32_dpp %[g0], %[g0], %[g0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : [g0] "+v"(g0), [g1] "+v"(g1), [g2] "+v"(g2), [g3] "+v"(g3), [g4] "+v"(g4), [g5] "+v"(g5), [g6] "+v"(g6),
          [g7] "+v"(g7), [t0] "=&v"(t0), [t1] "=&v"(t1)
        :
        : "vcc");
    return g0;
}

template <int R, int Q, int QS>
__device__ __forceinline__ void grad_body(const vec2<float> (&a)[1 << R], const unsigned reg_cmask, const bool lane_pred,
                                          const bool thr_ok, const float scale, const unsigned acc_byte) {
    using V2 = vec2<float>;
    V2 g[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};      // G[0][0], G[0][1], G[1][0], G[1][1] as (re, im)
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
        if (((j >> Q) & 1) || ((j >> QS) & 1)) continue;
        if ((j & reg_cmask) != reg_cmask) continue;      // (uniform)
        const V2 p0 = a[j], p1 = a[j | (1 << Q)], l0 = a[j | (1 << QS)], l1 = a[j | (1 << Q) | (1 << QS)];
        const V2 l0s = {l0.y, -l0.x}, l1s = {l1.y, -l1.x};
        // l * conj(p) = l * p.x + (l.y, -l.x) * p.y : two packed FMAs
        g[0] = __builtin_elementwise_fma(l0, (V2)(p0.x), g[0]);
        g[0] = __builtin_elementwise_fma(l0s, (V2)(p0.y), g[0]);
        g[1] = __builtin_elementwise_fma(l0, (V2)(p1.x), g[1]);
        g[1] = __builtin_elementwise_fma(l0s, (V2)(p1.y), g[1]);
        g[2] = __builtin_elementwise_fma(l1, (V2)(p0.x), g[2]);
        g[2] = __builtin_elementwise_fma(l1s, (V2)(p0.y), g[2]);
        g[3] = __builtin_elementwise_fma(l1, (V2)(p1.x), g[3]);
        g[3] = __builtin_elementwise_fma(l1s, (V2)(p1.y), g[3]);
    }
    if (lane_pred) {                                     // a control on a thread bit: such lanes contribute nothing
        const float w = thr_ok ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] *= w;
    }
    const float tot = row_reduce_scatter8(g[0].x, g[0].y, g[1].x, g[1].y, g[2].x, g[2].y, g[3].x, g[3].y) * scale;
    const unsigned lane = threadIdx.x;
    if ((lane & 2u) == 0u) {       // one lane of each pair; the four rows of the wavefront add on their own
        const unsigned idx = ((lane & 1u) << 2) | ((lane >> 2) & 2u) | ((lane >> 2) & 1u);
        __hip_atomic_fetch_add((__attribute__((address_space(3))) float*)(uintptr_t)(acc_byte + 4u * idx), tot,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int R, int Q>
__device__ __forceinline__ void dispatch_grad_qs(const vec2<float> (&a)[1 << R], int qs, unsigned reg_cmask, bool lane_pred,
                                                 bool thr_ok, float scale, unsigned acc_byte) {
    switch (qs) {
        case 0: if constexpr (Q != 0) grad_body<R, Q, 0>(a, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        case 1: if constexpr (Q != 1) grad_body<R, Q, 1>(a, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        case 2: if constexpr (Q != 2) grad_body<R, Q, 2>(a, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        default: if constexpr (R > 3 && Q != 3) grad_body<R, Q, 3>(a, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
    }
}

template <int R>
__device__ __forceinline__ void dispatch_grad(const vec2<float> (&a)[1 << R], int q, int qs, unsigned reg_cmask, bool lane_pred,
                                              bool thr_ok, float scale, unsigned acc_byte) {
    switch (q) {
        case 0: dispatch_grad_qs<R, 0>(a, qs, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        case 1: dispatch_grad_qs<R, 1>(a, qs, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        case 2: dispatch_grad_qs<R, 2>(a, qs, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
        default: if constexpr (R > 3) dispatch_grad_qs<R, 3>(a, qs, reg_cmask, lane_pred, thr_ok, scale, acc_byte); break;
    }
}
template <int R>
__device__ __forceinline__ void dispatch_grad(const vec2<double> (&)[1 << R], int, int, unsigned, bool, bool, float, unsigned) {}

// XOR swizzle of the LDS element index (fusion.lds_swizzle is the same function; tools/lds_conflicts.py counts the
// bank conflicts of a schedule under it): every higher group of 5 (8-byte elements: 32 slots per LDS row) / 4 index
// bits is folded onto the low group, so every tile bit moves the bank; for 8-byte elements bit 4 also toggles bit 0,
// which spreads the lanes of the global-I/O layout over the 16 slots a store group sees.  Bijective on [0, 2^M).
template <int ESZ> __device__ __forceinline__ unsigned lds_swz(unsigned e) {
    if constexpr (ESZ == 8)
        return e ^ ((e >> 5) & 31u) ^ ((e >> 10) & 31u) ^ ((e >> 4) & 1u);
    else
        return e ^ ((e >> 4) & 15u) ^ ((e >> 8) & 15u) ^ ((e >> 12) & 15u);
}

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// Kernel-side copy of a workgroup-tile pass: the layout these kernels decode word by word from the kernel-argument
// segment.  (The public DqFusedPass has room for the six slots of the wave-tile geometry; these kernels use four.)
struct KFusedRound {
    uint8_t rb[4];
    uint8_t tb[DQ_FUSED_MAX_TBITS];
    uint8_t flags, gate_begin, gate_end;
};
struct KFusedPass {
    uint8_t m, L, h, nrounds;
    uint8_t high_pos[DQ_FUSED_MAX_HIGH];
    uint8_t high_sorted[DQ_FUSED_MAX_HIGH];
    uint8_t load_rb[4];
    uint8_t store_rb[4];
    KFusedRound rounds[DQ_FUSED_MAX_ROUNDS];
    uint32_t mat_base;
    DqFusedGate gates[DQ_FUSED_MAX_GATES];
    uint64_t load_slot_off[4];
    uint64_t store_slot_off[4];
    uint16_t lds_tab[DQ_FUSED_MAX_ROUNDS + 2][16];
    uint8_t store_high_pos[DQ_FUSED_MAX_HIGH];
    uint8_t store_blk_pos[DQ_FUSED_MAX_BLK];
    uint8_t store_low_pos[DQ_FUSED_MAX_LOW];
    uint8_t store_tb[DQ_FUSED_MAX_TBITS];
};
static void repack(const DqFusedPass& s, KFusedPass& d) {
    memset(&d, 0, sizeof(d));
    d.m = s.m, d.L = s.L, d.h = s.h, d.nrounds = s.nrounds;
    memcpy(d.high_pos, s.high_pos, sizeof(d.high_pos));
    memcpy(d.high_sorted, s.high_sorted, sizeof(d.high_sorted));
    for (int i = 0; i < 4; ++i) {
        d.load_rb[i] = s.load_rb[i], d.store_rb[i] = s.store_rb[i];
        d.load_slot_off[i] = s.load_slot_off[i], d.store_slot_off[i] = s.store_slot_off[i];
    }
    for (int r = 0; r < DQ_FUSED_MAX_ROUNDS; ++r) {
        for (int i = 0; i < 4; ++i) d.rounds[r].rb[i] = s.rounds[r].rb[i];
        memcpy(d.rounds[r].tb, s.rounds[r].tb, sizeof(d.rounds[r].tb));
        d.rounds[r].flags = s.rounds[r].flags, d.rounds[r].gate_begin = s.rounds[r].gate_begin, d.rounds[r].gate_end = s.rounds[r].gate_end;
    }
    d.mat_base = s.mat_base;
    memcpy(d.gates, s.gates, sizeof(d.gates));
    memcpy(d.lds_tab, s.lds_tab, sizeof(d.lds_tab));
    memcpy(d.store_high_pos, s.store_high_pos, sizeof(d.store_high_pos));
    memcpy(d.store_blk_pos, s.store_blk_pos, sizeof(d.store_blk_pos));
    memcpy(d.store_low_pos, s.store_low_pos, sizeof(d.store_low_pos));
    memcpy(d.store_tb, s.store_tb, sizeof(d.store_tb));
}

// Mirror of the kernel's argument list: where the by-value descriptor sits in the kernarg segment.
struct FusedKernArgs {
    const void* in;
    void* out;
    const void* mats;
    int64_t mat_bstride;
    int64_t in_bstride;
    int n;
    int tpw;
    KFusedPass p;
    double* grads;          // GRAD kernels only: [batch, ngrads, 8], added to
    int64_t grad_bstride;   // = ngrads * 8
};

// PF: a workgroup walks `tpw` consecutive tiles and requests tile t + 1 from HBM (into spare VGPRs) before it starts
// the rounds of tile t, so the load latency of every tile but the first hides under the gates of its predecessor and
// the stores of tile t drain under the gates of tile t + 1.  Two 64-KiB workgroups per CU (what the tile buffer allows)
// cannot keep enough bytes in flight on their own: load -> gates -> store is strictly serial inside one workgroup.
//
// GRAD: the reverse sweep of the adjoint method (dq_apply_fused_grad_c64).  The pass's DQ_FG_GRAD records reduce
// sum lambda (x) conj(psi) over the thread's registers, then over the wavefront (DPP), and add the eight real sums to
// a per-record accumulator in LDS right behind the tile; after its last tile the workgroup adds the accumulators to
// `grads` (double, one atomic per record and component).  A separate instantiation: the forward kernels do not change.
template <typename T, int R, int LOGT, bool PF, bool GRAD = false>
__global__ __launch_bounds__(1 << LOGT) void fused_pass_kernel(const amp<T>* in, amp<T>* out,
                                                               const amp<T>* __restrict__ mats, int64_t mat_bstride,
                                                               int64_t in_bstride, int n, int tpw, const KFusedPass p,
                                                               double* grads, int64_t grad_bstride) {
    constexpr int M = R + LOGT;
    constexpr int NA = 1 << R;
    constexpr int VB = (sizeof(T) == 4) ? 1 : 0;  // low slots of the canonical layout (16 B per lane)
    using V = amp<T>;
    // The tile buffer is the kernel's only LDS (dynamic, no static LDS anywhere in this file), so it starts at LDS
    // address 0: forming the pointer from the byte offset alone saves the "+ base" VALU add the compiler would emit
    // per access for a relocatable symbol.  `dq_smem` stays declared so the launch's dynamic size has an owner.
    extern __shared__ __attribute__((aligned(16))) unsigned char dq_smem[];
    auto lds_at = [](unsigned byte_off) __attribute__((always_inline)) {
        return (__attribute__((address_space(3))) V*)(uintptr_t)byte_off;
    };
    (void)dq_smem;

    // ---- which tiles / which batch element (workgroup-uniform) ----
    // Normally blockIdx.x = group of `tpw` consecutive tiles, blockIdx.y = sample.  When all samples read ONE input
    // state (in_bstride == 0; the host launches that pass with tpw = 1) the B workgroups of a tile are instead made
    // neighbours in dispatch order AND placed on the same XCD (workgroups go round-robin over the 8 XCDs), so the
    // tile is fetched from HBM once and served to the other B - 1 from that XCD's L2.
    unsigned tile_id = blockIdx.x * (unsigned)tpw, sample = blockIdx.y;
#ifndef DQ_NO_XCD_REMAP
    if (in_bstride == 0 && (gridDim.x & 7u) == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.y;
        const unsigned group = lin / (8u * nb), r = lin % (8u * nb);
        sample = r >> 3;
        tile_id = group * 8u + (r & 7u);
    }
#endif
    // the raw 16-byte pieces of a tile as they come from memory: NP pieces per thread
    using Piece = typename std::conditional<VB == 1, float4, V>::type;
    constexpr int NP = VB == 1 ? NA / 2 : NA;
    V a[NA];
    uint64_t gt_store = 0;
    unsigned tbase0_keep = 0;
    constexpr unsigned ACC_BYTES0 = (unsigned)sizeof(V) << M;      // the accumulators of the DQ_FG_GRAD records
    if constexpr (GRAD) {
        for (unsigned i = threadIdx.x; i < DQ_FUSED_MAX_GATES * 8u; i += 1u << LOGT)
            *(__attribute__((address_space(3))) float*)(uintptr_t)(ACC_BYTES0 + 4u * i) = 0.0f;
        __syncthreads();
    }
  for (int tile_no = 0; tile_no < tpw; ++tile_no, ++tile_id) {
    // Everything a tile needs is derived INSIDE the loop from two laundered values (the kernel-argument pointer and the
    // thread id): hoisted out as loop invariants, the decoded header and the per-thread offsets would stay live across
    // the gate loop -- whose assembly block owns s[80:99] -- and spill (measured: 50 SGPR spills, 142 VGPRs).
    uint64_t karg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(karg));
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // (address space 4 = constant: keeps the descriptor reads scalar loads; a plain pointer made from an integer is
    // a flat one and would be read per lane)
    typedef const __attribute__((address_space(4))) uint32_t* KArgWords;
    const KArgWords hw = (KArgWords)(karg + offsetof(FusedKernArgs, p));
    static_assert(DQ_FAST32_IDS == DQ_FAST_IDS, "tools/gen_fused_asm.py and include/dq_hip.h disagree on the handler ids");
    static_assert(DQ_FUSED_MAX_HIGH == 12 && offsetof(KFusedPass, high_pos) == 4 && offsetof(KFusedPass, high_sorted) == 16 &&
                      offsetof(KFusedPass, load_rb) == 28 && offsetof(KFusedPass, store_rb) == 32,
                  "header word layout");
    const uint32_t hw0 = hw[0], hp0 = hw[1], hp1 = hw[2], hp2 = hw[3], hs0 = hw[4], hs1 = hw[5], hs2 = hw[6], lrb = hw[7],
                   srbw = hw[8];
    (void)srbw;
    const int L = (int)((hw0 >> 8) & 0xffu), h = (int)((hw0 >> 16) & 0xffu);
    auto byte_of = [](uint32_t w0, uint32_t w1, uint32_t w2, int i) __attribute__((always_inline)) -> unsigned {
        return ((i < 4 ? w0 : (i < 8 ? w1 : w2)) >> (8 * (i & 3))) & 0xffu;
    };

    // Are the gathered bits simply the bits right above the contiguous run, in order (high_pos[i] == L + i)?  With
    // permuted stores that is every pass but the first: the tile is one contiguous block, its base is a shift and a
    // tile-local index IS the offset.  (Three masked word compares; the general path deposits bit by bit.)
    bool contig;
    {
        const uint32_t want = 0x03020100u + (uint32_t)L * 0x01010101u;
        auto same = [&](uint32_t w, int k) __attribute__((always_inline)) -> bool {
            const int nb = h - 4 * k;   // bytes of word k that are in use
            const uint32_t mask = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
            return ((w ^ (want + 0x04040404u * (uint32_t)k)) & mask) == 0u;
        };
        contig = same(hp0, 0) && same(hp1, 1) && same(hp2, 2);
    }
    // global index bits a tile fixes (= offset of its first amplitude inside the state)
    auto tile_of = [&](unsigned id) __attribute__((always_inline)) -> uint64_t {
        if (contig) return (uint64_t)id << (L + h);
        uint64_t tl = (uint64_t)id << L;
        for (int i = 0; i < h; ++i) tl = insert_zero(tl, (int)byte_of(hs0, hs1, hs2, i));
        return tl;
    };
    // in_bstride = 2^n normally; 0 when every batch element starts from the same (single) input state
    const V* const pin0 = in + (uint64_t)sample * (uint64_t)in_bstride;
    // write side: block-index bit j goes to global bit store_blk_pos[j], tile bit L + i to store_high_pos[i]
    // (both equal to the read positions for an in-place pass; include/dq_hip.h)
    constexpr int SHP_W0 = offsetof(KFusedPass, store_high_pos) / 4, SBP_W0 = offsetof(KFusedPass, store_blk_pos) / 4;
    static_assert(offsetof(KFusedPass, store_high_pos) % 4 == 0 && offsetof(KFusedPass, store_blk_pos) % 4 == 0, "");
    auto tile_w_of = [&](unsigned id) __attribute__((always_inline)) -> uint64_t {
        // position bytes fetched once (six words), the loop unrolled with constant byte positions: no scalar load per
        // bit.  Block-index bits above n - m are zero, so whatever their position bytes hold contributes nothing.
        uint64_t tw = 0;
        const int nblk = n - L - h;
        uint32_t bw[DQ_FUSED_MAX_BLK / 4];
#pragma unroll
        for (int w = 0; w < DQ_FUSED_MAX_BLK / 4; ++w) bw[w] = hw[SBP_W0 + w];
#pragma unroll
        for (int w = 0; w < DQ_FUSED_MAX_BLK / 4; ++w) {
            if (4 * w < nblk) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tw |= (uint64_t)((id >> (4 * w + k)) & 1u) << ((bw[w] >> (8 * k)) & 0x3fu);
            }
        }
        return tw;
    };
    V* const pout0 = out + ((uint64_t)sample << n);
    constexpr int SLP_W0 = offsetof(KFusedPass, store_low_pos) / 4, STB_W0 = offsetof(KFusedPass, store_tb) / 4;
    static_assert(offsetof(KFusedPass, store_low_pos) % 4 == 0 && offsetof(KFusedPass, store_tb) % 4 == 0 &&
                      DQ_FUSED_MAX_LOW == 8, "");
    auto glob_w = [&](unsigned e) __attribute__((always_inline)) -> uint64_t {
        // tile bit i < L -> store_low_pos[i] (the low bits move like the others: include/dq_hip.h)
        const uint32_t lp0 = hw[SLP_W0], lp1 = hw[SLP_W0 + 1];
        uint64_t g = 0;
#pragma unroll
        for (int i = 0; i < DQ_FUSED_MAX_LOW; ++i)
            if (i < L) g |= (uint64_t)((e >> i) & 1u) << (((i < 4 ? lp0 : lp1) >> (8 * (i & 3))) & 0x3fu);
        for (int i = 0; i < h; ++i)
            g |= (uint64_t)((e >> (L + i)) & 1u) << ((hw[SHP_W0 + (i >> 2)] >> (8 * (i & 3))) & 0xffu);
        return g;
    };
    const V* mbase = mats + (int64_t)sample * mat_bstride;

    // tile-local index -> offset inside the state
    auto glob = [&](unsigned e) __attribute__((always_inline)) -> uint64_t {
        if (contig) return e;
        uint64_t g = e & ((1u << L) - 1u);
        for (int i = 0; i < h; ++i) g |= (uint64_t)((e >> (L + i)) & 1u) << byte_of(hp0, hp1, hp2, i);
        return g;
    };

    // ---- load layout: slots from the descriptor, thread bits = remaining tile bits ascending ----
    if (!PF || tile_no == 0) {     // (the same for every tile of the workgroup: carried, like the write offset below)
        unsigned b0 = tid;  // thread base (tile-local) in the load layout
#pragma unroll
        for (int s = 0; s < R; ++s) b0 = (unsigned)insert_zero(b0, (int)((lrb >> (8 * s)) & 0xffu));
        tbase0_keep = b0;
    }
    const unsigned tbase0 = tbase0_keep;
    unsigned tbase = tbase0;  // current thread base

    // per-thread offset of the load layout inside a tile, and what each register slot adds to it
    const uint64_t gt_load = glob(tbase0);
    uint64_t gs_load[R];
    {
        const __attribute__((address_space(4))) uint64_t* so = (const __attribute__((address_space(4))) uint64_t*)(hw + offsetof(KFusedPass, load_slot_off) / 4);
#pragma unroll
        for (int s = 0; s < R; ++s) gs_load[s] = so[s];
    }
    auto request = [&](Piece (&dst)[NP], const V* pin) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j = VB == 1 ? 2 * i : i;
            uint64_t o = gt_load;
#pragma unroll
            for (int s = VB; s < R; ++s)
                if ((j >> s) & 1) o += gs_load[s];
            dst[i] = *reinterpret_cast<const Piece*>(pin + o);
        }
    };
    auto unpack = [&](const Piece (&src)[NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if constexpr (VB == 1) {
                a[2 * i] = amp<T>{src[i].x, src[i].y};
                a[2 * i + 1] = amp<T>{src[i].z, src[i].w};
            } else {
                a[i] = src[i];
            }
        }
    };
    if (tile_no == 0) {
        Piece first[NP];
        request(first, pin0 + tile_of(tile_id));
        unpack(first);
        // complete HERE: were the first tile still in flight where this path joins the loop, hipcc's wait-count
        // bookkeeping would merge the two paths into an s_waitcnt vmcnt(0) in front of the gate loop, which every
        // later tile would then spend waiting for its successor's prefetch
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < NA; ++j) asm volatile("" : "+v"(a[j]));
        }
    }

    // LDS staging: a thread's byte address = swizzle(its base) * sizeof(V)  XOR  the host-made table entry of the
    // register-slot pattern (the swizzle is XOR-linear; include/dq_hip.h, lds_tab).  `ctab` = table of the layout
    // the registers are in right now.
    constexpr int TAB_W0 = offsetof(KFusedPass, lds_tab) / 4;   // 8 words = 16 entries per table
    constexpr int TAB_WORDS = (NA + 1) / 2;
    int cur_tab = 0;   // table of the layout the registers are in right now (0 = load layout, 1 + r = round r)
    auto tab_entry = [](const uint32_t (&tab)[TAB_WORDS], int j) __attribute__((always_inline)) -> unsigned {
        return (tab[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    };
    // Whether a trip is needed is the host's decision (DqFusedRound::flags, checked by dq_apply_fused): a comparison
    // of layouts here would be per lane (the thread base is a VGPR) and cost an exec-masked region per round.  Both
    // offset tables are (re)loaded from the descriptor -- scalar loads -- instead of being carried across the gates.
    auto transpose_to = [&](const unsigned ntbase, const int table) __attribute__((always_inline)) {
#ifdef DQ_ABLATE_LDS       // timing experiments only: wrong results, no LDS trip
        cur_tab = table;
        tbase = ntbase;
        return;
#endif
        uint32_t ctab[TAB_WORDS], ntab[TAB_WORDS];
#pragma unroll
        for (int w = 0; w < TAB_WORDS; ++w) {
            ctab[w] = hw[TAB_W0 + 8 * cur_tab + w];
            ntab[w] = hw[TAB_W0 + 8 * table + w];
        }
        const unsigned vw = lds_swz<sizeof(V)>(tbase) * (unsigned)sizeof(V);
        const unsigned vr = lds_swz<sizeof(V)>(ntbase) * (unsigned)sizeof(V);
#pragma unroll
        for (int j = 0; j < NA; ++j) *lds_at(vw ^ tab_entry(ctab, j)) = a[j];
#ifndef DQ_ABLATE_BARRIER   // timing experiments only (tools/ablate.sh)
        __syncthreads();
#endif
#pragma unroll
        for (int j = 0; j < NA; ++j) a[j] = *lds_at(vr ^ tab_entry(ntab, j));
#ifndef DQ_ABLATE_BARRIER
        __syncthreads();
#endif
        cur_tab = table;
        tbase = ntbase;
    };

    constexpr bool FAST32 = (sizeof(T) == 4 && R == 4 && DQ_USE_ASM_BLOCKS);
    constexpr bool FAST64 = (sizeof(T) == 8 && R == 3 && DQ_USE_ASM_BLOCKS);
    constexpr bool FAST = FAST32 || FAST64;
    // Gate records are fetched with explicit scalar loads from the kernel-argument segment (the descriptor
    // is passed by value; its address must not be taken through `&p`, that would force a private copy).
    const uint64_t kgates = karg + offsetof(FusedKernArgs, p) + offsetof(KFusedPass, gates);

    // The descriptor is read as 32-bit words (scalar loads; gfx950 has no sub-dword s_load) and decoded
    // with SALU bit ops, so no vector memory instruction is spent on it.
    const KArgWords pw = hw;
    constexpr int ROUND_W0 = offsetof(KFusedPass, rounds) / 4;
    constexpr int GATE_W0 = offsetof(KFusedPass, gates) / 4;
    const int nrounds = (int)(pw[0] >> 24);
    const uint64_t mbase_u = (uint64_t)mbase;
    const uint64_t tile_global = tile_of(tile_id);  // global index bits fixed for this tile (outside the tile)
    // tile t + 1 is requested now and consumed after the store of tile t
    Piece nx[NP];
    const bool more = PF && tile_no + 1 < tpw;
    if constexpr (PF) {
        if (more) request(nx, pin0 + tile_of(tile_id + 1u));
    }
    tbase = tbase0;
    cur_tab = 0;
    // running byte offset of the current gate's matrix from `mbase` (32 bits: SMEM takes base pair + SGPR offset)
    uint32_t moff = pw[offsetof(KFusedPass, mat_base) / 4] * (uint32_t)sizeof(V);
    T hscale = T(1);   // product of the deferred Hadamard factors of this pass (uniform)
    float hsr = 1.0f, hsi = 0.0f;   // complex64: the deferred factor is complex (Hadamards and Rx-like gates, dq_hip.h)
    bool had = false;
    unsigned last_flags = 0;
    for (int r = 0; r < nrounds; ++r) {
        const uint32_t rw1 = pw[ROUND_W0 + 4 * r + 1], rw2 = pw[ROUND_W0 + 4 * r + 2],
                       rw3 = pw[ROUND_W0 + 4 * r + 3];
        last_flags = (rw3 >> 8) & 0xffu;
        if (last_flags & (DQ_ROUND_TRANSPOSE | DQ_ROUND_SWAP)) {
            unsigned ntbase = 0;
#pragma unroll
            for (int i = 0; i < LOGT; ++i) {
                const uint32_t w = i < 4 ? rw1 : (i < 8 ? rw2 : rw3);
                ntbase |= ((tid >> i) & 1u) << ((w >> (8 * (i & 3))) & 0xffu);
            }
            if (last_flags & DQ_ROUND_TRANSPOSE) {
                transpose_to(ntbase, 1 + r);
            } else {        // the round's leading DQ_FG_SWAP records move the amplitudes inside the wavefronts
                tbase = ntbase;
                cur_tab = 1 + r;
            }
        }
        const int gbeg = (int)((rw3 >> 16) & 0x7fu), gend = (int)(rw3 >> 24);
        if constexpr (FAST) {
            if (rw3 & (DQ_ROUND_ALL_FAST << 16)) {   // the whole gate loop of the round in assembly (dq_fused_asm.inc)
                uint32_t goff = 32u * (unsigned)gbeg;
                if constexpr (FAST32)
                    fast_gate_loop_f32(a, kgates, goff, 32u * (unsigned)gend, mbase_u, moff, tile_global, tbase, hsr, hsi);
                else
                    fast_gate_loop_f64(a, kgates, goff, 32u * (unsigned)gend, mbase_u, moff, tile_global, tbase, hscale);
                continue;
            }
        }
        // byte offsets of the round's gate records from `kgates`; the loop runs on the offset (one add, one compare)
        const uint32_t goff_end = 32u * (unsigned)gend;
        for (uint32_t goff = 32u * (unsigned)gbeg; goff < goff_end; goff += 32u) {
            // ONE scalar-memory round trip per gate: the 32-byte record and -- from the running offset, no
            // decode needed because the host lays the matrices of a pass out in gate order -- its matrix.
            u32x8 rec;
            typename std::conditional<FAST64, u32x16, u32x8>::type mqv;
            if constexpr (FAST32) {
                asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx8 %1, %4, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(rec), "=&s"(mqv)
                             : "s"(kgates), "s"(goff), "s"(mbase_u), "s"(moff));
            } else if constexpr (FAST64) {
                asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx16 %1, %4, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(rec), "=&s"(mqv)
                             : "s"(kgates), "s"(goff), "s"(mbase_u), "s"(moff));
            } else {
                asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(rec) : "s"(kgates), "s"(goff));
            }
            if constexpr (sizeof(V) == 8) asm volatile("s_lshl3_add_u32 %0, %1, %0" : "+s"(moff) : "s"(rec[6]) : "scc");
            else asm volatile("s_lshl4_add_u32 %0, %1, %0" : "+s"(moff) : "s"(rec[6]) : "scc");
            const uint32_t g0 = rec[0], g1 = rec[1], gmat = rec[2], fast = rec[3];
            const uint64_t out_cmask = (uint64_t)rec[4] | ((uint64_t)rec[5] << 32);
            const unsigned reg_cmask = (g1 >> 8) & 0xffu, thr_cmask = g1 >> 16;
            if constexpr (FAST) {
                // Straight-line handlers picked by the host (include/dq_hip.h, DqFusedGate::fast), reached through the
                // jump table of fast_dispatch_f32 / _f64 (rounds that are not marked DQ_ROUND_ALL_FAST come here gate by
                // gate; marked rounds never leave the assembly loop above).
                if (__builtin_expect(fast < (uint32_t)DQ_FAST_IDS, 1)) {
#define DQ_PAIR(I) ((uint64_t)mqv[2 * (I)] | ((uint64_t)mqv[2 * (I) + 1] << 32))
                    if constexpr (FAST32) {
                        // every straight-line handler sits behind ONE jump table (dq_fused_asm.inc, fast_dispatch_f32):
                        // the cost of reaching it does not depend on which one it is; handlers of controlled gates test
                        // the outside and thread controls themselves.
                        const uint64_t mq[4] = {DQ_PAIR(0), DQ_PAIR(1), DQ_PAIR(2), DQ_PAIR(3)};
                        fast_dispatch_f32(a, mq, mqv[0], mqv[1], mqv[6], fast, g1, out_cmask, tile_global, tbase, hsr, hsi);
                    } else {
                        const double md[8] = {__longlong_as_double((long long)DQ_PAIR(0)), __longlong_as_double((long long)DQ_PAIR(1)),
                                              __longlong_as_double((long long)DQ_PAIR(2)), __longlong_as_double((long long)DQ_PAIR(3)),
                                              __longlong_as_double((long long)DQ_PAIR(4)), __longlong_as_double((long long)DQ_PAIR(5)),
                                              __longlong_as_double((long long)DQ_PAIR(6)), __longlong_as_double((long long)DQ_PAIR(7))};
                        fast_dispatch_f64(a, md, fast, g1, out_cmask, tile_global, tbase, hscale);
                    }
#undef DQ_PAIR
                    continue;
                }
            }
            if ((tile_global & out_cmask) != out_cmask) continue;  // uniform: a control outside the tile is 0
            const unsigned kind = g0 & 0xffu, q = (g0 >> 8) & 0xffu, q2 = (g0 >> 16) & 0xffu, loc = g0 >> 24;
            const unsigned loc2 = g1 & 0xffu;
            const bool thr_ok = (tbase & thr_cmask) == thr_cmask;
            // (the complex128 kernels keep plain global reads: sixteen 16-byte entries of a two-qubit matrix do not fit
            // the scalar registers, and they do not prefetch)
            using MatPtr = typename std::conditional<PF, cmat<T>, const V*>::type;
            const MatPtr mp = (MatPtr)mbase_u + gmat;
            switch (kind) {
                case DQ_FG_GEN1:  // (a Hadamard that is not on a straight-line handler is just a real matrix)
                    dispatch_gen1<T, R>(a, q, mp, loc == DQ_MODE_HAD ? (unsigned)DQ_MODE_REAL : loc, reg_cmask, thr_cmask != 0, thr_ok);
                    break;
                case DQ_FG_X1: dispatch_x1<T, R>(a, q, reg_cmask, thr_cmask != 0, thr_ok); break;
                case DQ_FG_GEN2: dispatch_gen2<T, R>(a, q, q2, mp, loc, reg_cmask, thr_ok); break;
                case DQ_FG_GRAD:
                    if constexpr (GRAD && sizeof(T) == 4) {
                        // the registers lack the pass's deferred factor f (both states alike): G_true = |f|^2 G
                        const float f2 = hsr * hsr + hsi * hsi;
                        dispatch_grad<R>(a, q, q2, reg_cmask, thr_cmask != 0, thr_ok, f2, ACC_BYTES0 + goff);   // 32 B per record, too
                    }
                    break;
                case DQ_FG_DIAG1: {
                    const V d0 = mp[0], d1 = mp[3];
                    int fixed = -1;  // target bit value when it is not a register slot
                    if (loc == DQ_LOC_THR) fixed = (tbase >> q) & 1u;
                    else if (loc == DQ_LOC_OUT) fixed = (int)((tile_global >> q) & 1ull);
#pragma unroll
                    for (int j = 0; j < NA; ++j) {
                        if (thr_ok && ((j & reg_cmask) == reg_cmask)) {
                            const int bit = (fixed >= 0) ? fixed : ((j >> q) & 1);
                            a[j] = cmul(bit ? d1 : d0, a[j]);
                        }
                    }
                    break;
                }
                default: {  // DQ_FG_DIAG2: index = bit(target1) * 2 + bit(target2)
                    const V d0 = mp[0], d1 = mp[5], d2 = mp[10], d3 = mp[15];
                    int f1 = -1, f2 = -1;
                    if (loc == DQ_LOC_THR) f1 = (tbase >> q) & 1u;
                    else if (loc == DQ_LOC_OUT) f1 = (int)((tile_global >> q) & 1ull);
                    if (loc2 == DQ_LOC_THR) f2 = (tbase >> q2) & 1u;
                    else if (loc2 == DQ_LOC_OUT) f2 = (int)((tile_global >> q2) & 1ull);
#pragma unroll
                    for (int j = 0; j < NA; ++j) {
                        if (thr_ok && ((j & reg_cmask) == reg_cmask)) {
                            const int b1 = (f1 >= 0) ? f1 : ((j >> q) & 1);
                            const int b2 = (f2 >= 0) ? f2 : ((j >> q2) & 1);
                            const V ph = b1 ? (b2 ? d3 : d2) : (b2 ? d1 : d0);
                            a[j] = cmul(ph, a[j]);
                        }
                    }
                    break;
                }
            }
        }
    }

    if constexpr (FAST32) {
        // a[j] *= hsr + i hsi (uniform per workgroup: the deferred factors of the Hadamards and Rx-like gates)
        const float ur = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hsr)));
        const float ui = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hsi)));
        if (ui == 0.0f) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                a[j].x *= ur;
                a[j].y *= ur;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const float x = a[j].x, y = a[j].y;
                a[j].x = fmaf(x, ur, -y * ui);
                a[j].y = fmaf(x, ui, y * ur);
            }
        }
    } else if (had || FAST) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            a[j].x *= hscale;
            a[j].y *= hscale;
        }
    }

    if (last_flags & DQ_ROUND_TRANSPOSE_AFTER) {  // ---- into the store layout ----
        unsigned stbase = 0;       // thread-index bit i sits on tile bit store_tb[i]
        const uint32_t sw0 = hw[STB_W0], sw1 = hw[STB_W0 + 1], sw2 = hw[STB_W0 + 2];
#pragma unroll
        for (int i = 0; i < LOGT; ++i) {
            const uint32_t w = i < 4 ? sw0 : (i < 8 ? sw1 : sw2);
            stbase |= ((tid >> i) & 1u) << ((w >> (8 * (i & 3))) & 0xffu);
        }
        transpose_to(stbase, DQ_FUSED_MAX_ROUNDS + 1);
    }

    {
        V* const pout = pout0 + tile_w_of(tile_id);
        // a thread's offset inside the written tile is the same for every tile of the pass: computed for the first tile
        // of a workgroup and carried in two VGPRs -- the deposit of m tile bits costs ~50 VALU issue slots per tile,
        // 0.5 ms of an 18.9-ms pass (A/B).  (Only the prefetching kernels walk several tiles.)
        if (!PF || tile_no == 0) gt_store = glob_w(tbase);
        const uint64_t gt = gt_store;
        const __attribute__((address_space(4))) uint64_t* so = (const __attribute__((address_space(4))) uint64_t*)(hw + offsetof(KFusedPass, store_slot_off) / 4);
        uint64_t gs[R];
#pragma unroll
        for (int s = 0; s < R; ++s) gs[s] = so[s];
        if constexpr (VB == 1) {
#pragma unroll
            for (int j = 0; j < NA; j += 2) {
                uint64_t o = gt;
#pragma unroll
                for (int s = 1; s < R; ++s)
                    if ((j >> s) & 1) o += gs[s];
                float4 v;
                v.x = a[j].x;
                v.y = a[j].y;
                v.z = a[j + 1].x;
                v.w = a[j + 1].y;
                *reinterpret_cast<float4*>(pout + o) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                uint64_t o = gt;
#pragma unroll
                for (int s = 0; s < R; ++s)
                    if ((j >> s) & 1) o += gs[s];
                pout[o] = a[j];
            }
        }
    }
    if constexpr (PF) {
        if (more) unpack(nx);
    } else {
        if (tile_no + 1 < tpw) {
            Piece nxt[NP];
            request(nxt, pin0 + tile_of(tile_id + 1u));
            unpack(nxt);
        }
    }
  }
    if constexpr (GRAD) {
        // one atomic per (record, component) and workgroup: the records' rows come from the descriptor, read from the
        // kernel-argument segment (constant address space; indexing `p` itself per lane would copy it to scratch)
        __syncthreads();
        typedef const __attribute__((address_space(4))) uint32_t* KWords;
        const KWords kw = (KWords)((uint64_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FusedKernArgs, p));
        const int nr = (int)(kw[0] >> 24);
        const unsigned ngates = kw[offsetof(KFusedPass, rounds) / 4 + 4 * (nr - 1) + 3] >> 24;
        double* const grow = grads + (uint64_t)sample * (uint64_t)grad_bstride;
        for (unsigned i = threadIdx.x; i < ngates * 8u; i += 1u << LOGT) {
            const KWords gw = kw + offsetof(KFusedPass, gates) / 4 + 8u * (i >> 3);
            if ((gw[0] & 0xffu) == (unsigned)DQ_FG_GRAD) {
                const float v = *(__attribute__((address_space(3))) float*)(uintptr_t)(ACC_BYTES0 + 4u * i);
                atomicAdd(grow + (uint64_t)gw[7] * 8u + (i & 7u), (double)v);
            }
        }
    }
}

struct FusedVariant {
    int m, slots, logt;
};
// variant 0 is the default geometry of each precision; complex64: the wave tile (csrc/dq_wave.hip), then the two
// workgroup tiles
static const FusedVariant kVariantsC64[] = {{12, 6, 6}, {13, 4, 9}, {12, 4, 8}};
static const FusedVariant kVariantsC128[] = {{11, 5, 6}, {12, 3, 9}, {11, 3, 8}};
static const int kNumVariantsC64 = 3, kNumVariantsC128 = 3;

// ngrads: rows of the caller's accumulator (dq_apply_fused_grad_*), -1 = a plain pass (DQ_FG_GRAD records refused)
template <typename T>
static int validate_pass(const DqFusedPass* p, int n, int slots, int logt, int64_t ngrads = -1) {
    const int m = slots + logt;
    if (p->m != m || p->L + p->h != m || p->h > DQ_FUSED_MAX_HIGH || p->L < 1) {
        set_error("dq_apply_fused: inconsistent geometry (m=%d L=%d h=%d)", p->m, p->L, p->h);
        return DQ_ERR_ARG;
    }
    if (n < m) {
        set_error("dq_apply_fused: n=%d smaller than tile m=%d", n, m);
        return DQ_ERR_ARG;
    }
    if (p->nrounds == 0 || p->nrounds > DQ_FUSED_MAX_ROUNDS) {
        set_error("dq_apply_fused: a pass has 1..%d rounds", DQ_FUSED_MAX_ROUNDS);
        return DQ_ERR_ARG;
    }
    uint64_t seen = 0;
    for (int i = 0; i < p->h; ++i) {
        const int hp = p->high_pos[i];
        if (hp < p->L || hp >= n || ((seen >> hp) & 1ull)) {
            set_error("dq_apply_fused: bad high bit %d", hp);
            return DQ_ERR_ARG;
        }
        seen |= 1ull << hp;
        if (i > 0 && p->high_sorted[i] <= p->high_sorted[i - 1]) {
            set_error("dq_apply_fused: high_sorted not ascending");
            return DQ_ERR_ARG;
        }
    }
    uint64_t seen2 = 0;
    for (int i = 0; i < p->h; ++i) seen2 |= 1ull << p->high_sorted[i];
    if (seen != seen2) {
        set_error("dq_apply_fused: high_sorted is not a permutation of high_pos");
        return DQ_ERR_ARG;
    }
    {   // load layout: tile bit 0 on slot 0 for complex64 (16 bytes per lane), otherwise gathered bits, ascending
        constexpr int vb = sizeof(T) == 4 ? 1 : 0;
        for (int s = 0; s < slots; ++s) {
            const bool ok = (s < vb) ? (p->load_rb[s] == s) : (p->load_rb[s] >= p->L && p->load_rb[s] < m);
            if (!ok || (s > 0 && p->load_rb[s] <= p->load_rb[s - 1])) {
                set_error("dq_apply_fused: load layout invalid at slot %d", s);
                return DQ_ERR_ARG;
            }
        }
    }
    if (p->L > DQ_FUSED_MAX_LOW) {
        set_error("dq_apply_fused: L = %d contiguous low bits, at most %d", p->L, DQ_FUSED_MAX_LOW);
        return DQ_ERR_UNSUPPORTED;
    }
    {   // write positions: a permutation of [0, n)
        const int nblk = n - m;
        if (nblk > DQ_FUSED_MAX_BLK) {
            set_error("dq_apply_fused: n - m = %d block bits, at most %d", nblk, DQ_FUSED_MAX_BLK);
            return DQ_ERR_UNSUPPORTED;
        }
        uint64_t wseen = 0;
        for (int i = 0; i < m + nblk; ++i) {
            const int pos = i < p->L ? p->store_low_pos[i]
                                     : (i < m ? p->store_high_pos[i - p->L] : p->store_blk_pos[i - m]);
            if (pos >= n || ((wseen >> pos) & 1ull)) {
                set_error("dq_apply_fused: write positions are not a permutation of [0, n) (entry %d = %d)", i, pos);
                return DQ_ERR_ARG;
            }
            wseen |= 1ull << pos;
        }
    }
    {   // store layout: explicit slots and thread bits covering the tile; complex64 stores two adjacent amplitudes per
        // lane, so the tile bit of slot 0 must be written to global bit 0
        unsigned used = 0;
        for (int s = 0; s < slots; ++s) {
            if (p->store_rb[s] >= m || ((used >> p->store_rb[s]) & 1u)) {
                set_error("dq_apply_fused: store layout invalid at slot %d", s);
                return DQ_ERR_ARG;
            }
            used |= 1u << p->store_rb[s];
        }
        for (int i = 0; i < logt; ++i) {
            if (p->store_tb[i] >= m || ((used >> p->store_tb[i]) & 1u)) {
                set_error("dq_apply_fused: store layout invalid at thread bit %d", i);
                return DQ_ERR_ARG;
            }
            used |= 1u << p->store_tb[i];
        }
        if (sizeof(T) == 4) {
            const int tb0 = p->store_rb[0];
            const int pos = tb0 < p->L ? p->store_low_pos[tb0] : p->store_high_pos[tb0 - p->L];
            if (pos != 0) {
                set_error("dq_apply_fused: the tile bit of store slot 0 (%d) is written to bit %d, not bit 0", tb0, pos);
                return DQ_ERR_ARG;
            }
        }
    }
    int next_gate = 0;
    uint32_t next_mat = p->mat_base;
    for (int r = 0; r < p->nrounds; ++r) {
        const DqFusedRound& rd = p->rounds[r];
        unsigned used = 0;
        for (int s = 0; s < slots; ++s) {
            if (rd.rb[s] >= m || ((used >> rd.rb[s]) & 1u)) {
                set_error("dq_apply_fused: round %d slot list invalid", r);
                return DQ_ERR_ARG;
            }
            used |= 1u << rd.rb[s];
        }
        for (int i = 0; i < logt; ++i) {
            if (rd.tb[i] >= m || ((used >> rd.tb[i]) & 1u)) {
                set_error("dq_apply_fused: round %d thread-bit list invalid", r);
                return DQ_ERR_ARG;
            }
            used |= 1u << rd.tb[i];
        }
        {   // the kernel trusts the layout flags: recompute them from the layouts
            const uint8_t* prb = r == 0 ? p->load_rb : p->rounds[r - 1].rb;
            // thread bits of an I/O layout: the tile bits that are not slots, ascending
            auto io_tb = [&](const uint8_t* iorb, uint8_t* tbo) {
                unsigned slotmask = 0;
                for (int s = 0; s < slots; ++s) slotmask |= 1u << iorb[s];
                for (int i = 0, q = 0; i < logt; ++i, ++q) {
                    while ((slotmask >> q) & 1u) ++q;
                    tbo[i] = (uint8_t)q;
                }
            };
            uint8_t ptb[DQ_FUSED_MAX_TBITS], stb[DQ_FUSED_MAX_TBITS];
            if (r == 0) io_tb(p->load_rb, ptb);
            else for (int i = 0; i < logt; ++i) ptb[i] = p->rounds[r - 1].tb[i];
            bool differs = false;
            for (int s = 0; s < slots; ++s) differs = differs || prb[s] != rd.rb[s];
            for (int i = 0; i < logt; ++i) differs = differs || ptb[i] != rd.tb[i];
            bool after = false;
            if (r == p->nrounds - 1) {
                for (int i = 0; i < logt; ++i) stb[i] = p->store_tb[i];
                for (int s = 0; s < slots; ++s) after = after || p->store_rb[s] != rd.rb[s];
                for (int i = 0; i < logt; ++i) {
                    after = after || rd.tb[i] != stb[i];
                }
            }
            // a layout change is an LDS trip or -- the host's choice, where it is possible -- in-wave exchanges
            const bool swaps = differs && (rd.flags & DQ_ROUND_SWAP) != 0;
            const unsigned want = (differs ? (swaps ? DQ_ROUND_SWAP : DQ_ROUND_TRANSPOSE) : 0u) |
                                  (after ? DQ_ROUND_TRANSPOSE_AFTER : 0u);
            if (rd.flags != want) {
                set_error("dq_apply_fused: round %d has layout flags %u, expected %u", r, rd.flags, want);
                return DQ_ERR_ARG;
            }
            // the leading DQ_FG_SWAP records of the round, applied to the previous layout, must give this one
            uint8_t erb[DQ_FUSED_MAX_SLOTS], etb[DQ_FUSED_MAX_TBITS];
            for (int s = 0; s < slots; ++s) erb[s] = prb[s];
            for (int i = 0; i < logt; ++i) etb[i] = ptb[i];
            int nswap = 0;
            for (int gi = rd.gate_begin & 0x7f; gi < rd.gate_end && p->gates[gi].kind == DQ_FG_SWAP; ++gi, ++nswap) {
                const DqFusedGate& g = p->gates[gi];
                if (!swaps || g.q >= slots || g.q2 >= 6 || g.q2 >= logt) {
                    set_error("dq_apply_fused: round %d: exchange record %d is out of place or malformed", r, gi);
                    return DQ_ERR_ARG;
                }
                const uint8_t t8 = erb[g.q];
                erb[g.q] = etb[g.q2];
                etb[g.q2] = t8;
            }
            if (swaps) {
                bool same = nswap > 0 && sizeof(T) == 4 && slots == 4 && (rd.gate_begin & DQ_ROUND_ALL_FAST);
                for (int s = 0; s < slots; ++s) same = same && erb[s] == rd.rb[s];
                for (int i = 0; i < logt; ++i) same = same && etb[i] == rd.tb[i];
                if (!same) {
                    set_error("dq_apply_fused: round %d: the exchange records do not produce the round's layout "
                              "(or the round is not an all-fast complex64 one)", r);
                    return DQ_ERR_ARG;
                }
            }
            for (int gi = (rd.gate_begin & 0x7f) + nswap; gi < rd.gate_end; ++gi) {
                if (p->gates[gi].kind == DQ_FG_SWAP) {
                    set_error("dq_apply_fused: round %d: exchange record %d does not lead the round", r, gi);
                    return DQ_ERR_ARG;
                }
            }
        }
        const int gate_begin = rd.gate_begin & 0x7f;
        const bool all_fast = (rd.gate_begin & DQ_ROUND_ALL_FAST) != 0;
        if (gate_begin > rd.gate_end || rd.gate_end > DQ_FUSED_MAX_GATES || (all_fast && gate_begin == rd.gate_end)) {
            set_error("dq_apply_fused: round %d gate range invalid", r);
            return DQ_ERR_ARG;
        }
        if (gate_begin != next_gate) {
            set_error("dq_apply_fused: round %d does not continue the gate list", r);
            return DQ_ERR_ARG;
        }
        next_gate = rd.gate_end;
        for (int gi = gate_begin; gi < rd.gate_end; ++gi) {
            const DqFusedGate& g = p->gates[gi];
            if (all_fast && g.fast == DQ_FAST_NONE) {
                set_error("dq_apply_fused: round %d is marked all-fast but gate %d has no handler id", r, gi);
                return DQ_ERR_ARG;
            }
            const bool slot_kind = g.kind == DQ_FG_GEN1 || g.kind == DQ_FG_X1 || g.kind == DQ_FG_GEN2;
            if (g.kind == DQ_FG_SWAP) {     // (position and layout were checked with the round)
                if (g.mat != next_mat || g.mat_advance != 0 || g.reg_cmask || g.thr_cmask || g.out_cmask ||
                    g.fast != 52u + 6u * g.q + g.q2) {
                    set_error("dq_apply_fused: exchange record %d malformed", gi);
                    return DQ_ERR_ARG;
                }
                continue;
            }
            if (g.kind == DQ_FG_GRAD) {
                if (ngrads < 0 || (sizeof(T) != 4 && logt != 6) || g.q >= slots || g.q2 >= slots || g.q == g.q2 || (g.reg_cmask >> slots) ||
                    ((g.reg_cmask >> g.q) & 1u) || ((g.reg_cmask >> g.q2) & 1u) || g.mat != next_mat || g.mat_advance != 0 ||
                    g.fast != DQ_FAST_NONE || (int64_t)g.reserved >= ngrads) {
                    set_error("dq_apply_fused: reduction record %d malformed, or not a dq_apply_fused_grad_* call (complex128: wave-tile geometry only)", gi);
                    return DQ_ERR_ARG;
                }
                continue;
            }
            if (g.kind == DQ_FG_EXPZ) {
                if (ngrads < 0 || logt != 6 || (g.reg_cmask >> slots) || g.mat != next_mat || g.mat_advance != 0 ||
                    g.fast != DQ_FAST_NONE || (int64_t)g.reserved >= ngrads) {
                    set_error("dq_apply_fused: expectation record %d malformed, or not a dq_apply_fused_grad_* call on a "
                              "wave-tile geometry", gi);
                    return DQ_ERR_ARG;
                }
                continue;
            }
            if (g.kind > DQ_FG_DIAG2 || (slot_kind && g.q >= slots) || ((g.kind == DQ_FG_GEN1 && g.loc > 3) || (g.kind == DQ_FG_GEN2 && g.loc > 1)) ||
                (g.kind == DQ_FG_GEN2 && (g.q2 >= slots || g.q2 == g.q)) || (g.reg_cmask >> slots)) {
                set_error("dq_apply_fused: gate %d malformed", gi);
                return DQ_ERR_ARG;
            }
            // the kernel walks the matrices with a running pointer: they must lie back to back in gate order
            static const uint32_t kSize[5] = {4, 0, 4, 16, 16};
            if (g.mat != next_mat || g.mat_advance != kSize[g.kind]) {
                set_error("dq_apply_fused: gate %d breaks the sequential matrix layout (mat=%u, expected %u)", gi,
                          g.mat, next_mat);
                return DQ_ERR_ARG;
            }
            next_mat += g.mat_advance;
            if (g.fast != DQ_FAST_NONE) {
                const bool free_ = g.reg_cmask == 0 && g.thr_cmask == 0 && g.out_cmask == 0;
                uint32_t want = DQ_FAST_NONE;
                if (g.kind == DQ_FG_X1) {
                    if (g.reg_cmask == 0) want = (free_ ? 16u : 32u) + g.q;
                    else if ((g.reg_cmask & (g.reg_cmask - 1)) == 0) want = 36u + 4u * g.q + (unsigned)__builtin_ctz(g.reg_cmask);
                } else if (g.kind == DQ_FG_GEN1 && g.reg_cmask == 0) {
                    want = free_ ? 4u * g.loc + g.q : 20u + 4u * (g.loc == DQ_MODE_HAD ? (unsigned)DQ_MODE_REAL : g.loc) + g.q;
                }
                if (g.fast != want) {
                    set_error("dq_apply_fused: gate %d has fast-handler id %u, expected %u", gi, g.fast, want);
                    return DQ_ERR_ARG;
                }
            }
        }
    }
    return DQ_OK;
}

// Tiles per workgroup of the prefetching variants (complex64): 0 = pick by size.  dq_fused_set_tiles_per_wg() is a
// tuning / A-B knob, not part of the data path's contract.
static std::atomic<int> g_tiles_per_wg{0};       // (relaxed atomic: a measurement knob, read once per launch)

template <typename T, int R, int LOGT, bool PF, bool GRAD = false>
static void launch_variant(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n,
                           int64_t batch, const DqFusedPass* pass, hipStream_t s, double* grads = nullptr,
                           int64_t ngrads = 0) {
    constexpr int M = R + LOGT;
    size_t lds_bytes = sizeof(amp<T>) << M;
    if (GRAD) lds_bytes += DQ_FUSED_MAX_GATES * 8 * sizeof(float);      // the reduction records' accumulators
    if (const char* pad = getenv("DQ_LDS_PAD_KB")) lds_bytes += (size_t)atoi(pad) << 10;   // occupancy experiments
    // (cheap; not cached: a process may drive several devices, and the attribute is per device)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_pass_kernel<T, R, LOGT, PF, GRAD>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    // tiles per workgroup: enough workgroups must remain to fill the chip several times over (256 CUs x 2..5
    // resident workgroups), and the pass that reads one shared input state keeps its XCD-aware order (one tile each)
    int tpw_log = 0;
    if ((PF || GRAD) && in_bstride != 0) {
        const int knob = g_tiles_per_wg.load(std::memory_order_relaxed);
        const int want = knob > 0 ? knob : 4;
        while ((2 << tpw_log) <= want && n - M - (tpw_log + 1) >= 0 &&
               ((int64_t)batch << (n - M - (tpw_log + 1))) >= 8192)
            ++tpw_log;
    }
    dim3 grid((unsigned)(1ull << (n - M - tpw_log)), (unsigned)batch);
    KFusedPass kp;
    repack(*pass, kp);
    hipLaunchKernelGGL((fused_pass_kernel<T, R, LOGT, PF, GRAD>), grid, dim3(1u << LOGT), lds_bytes, s,
                       static_cast<const amp<T>*>(in), static_cast<amp<T>*>(out), static_cast<const amp<T>*>(mats),
                       mat_bstride, in_bstride, n, 1 << tpw_log, kp, grads, ngrads * 8);
}

template <typename T>
static int fused_impl(const void* in, void* out, const void* mats, int64_t mat_bstride, int n, int64_t batch,
                      const DqFusedPass* pass, dq_stream_t stream, bool broadcast_in = false, double* grads = nullptr,
                      int64_t ngrads = -1) {
    const int64_t in_bstride = broadcast_in ? 0 : (int64_t)1 << n;
    if (broadcast_in && in == out) {
        set_error("dq_apply_fused_bcast: the shared input state cannot be the output buffer");
        return DQ_ERR_ARG;
    }
    if (!in || !out || !mats || !pass) {
        set_error("dq_apply_fused: null pointer");
        return DQ_ERR_ARG;
    }
    if (batch < 1 || batch > 65535) {
        set_error("dq_apply_fused: batch %lld out of range [1, 65535]", (long long)batch);
        return DQ_ERR_ARG;
    }
    constexpr bool is128 = sizeof(T) == 8;
    const FusedVariant* vars = is128 ? kVariantsC128 : kVariantsC64;
    const int want_slots = pass->slots ? pass->slots : (is128 ? 3 : 4);
    int vi = -1;
    for (int i = 0; i < (is128 ? kNumVariantsC128 : kNumVariantsC64); ++i)
        if (vars[i].m == pass->m && vars[i].slots == want_slots) vi = i;
    if (vi < 0) {
        set_error("dq_apply_fused: no kernel variant with m=%d and %d register slots", pass->m, want_slots);
        return DQ_ERR_UNSUPPORTED;
    }
    const FusedVariant v = vars[vi];
    int rc = validate_pass<T>(pass, n, v.slots, v.logt, ngrads);
    if (rc) return rc;
    if (in == out) {   // in place: every amplitude must be written where it was read
        bool same = true;
        for (int i = 0; i < pass->L; ++i) same = same && pass->store_low_pos[i] == i;
        for (int i = 0; i < pass->h; ++i) same = same && pass->store_high_pos[i] == pass->high_pos[i];
        uint64_t tilemask = 0;
        for (int i = 0; i < pass->h; ++i) tilemask |= 1ull << pass->high_pos[i];
        for (int j = 0, q = pass->L; j < n - v.m; ++j, ++q) {
            while ((tilemask >> q) & 1ull) ++q;
            same = same && pass->store_blk_pos[j] == q;
        }
        if (!same) {
            set_error("dq_apply_fused: a pass that writes to other index bits than it reads needs in != out");
            return DQ_ERR_ARG;
        }
    }
    if (n - v.m > 31) {
        set_error("dq_apply_fused: grid too large (n=%d)", n);
        return DQ_ERR_UNSUPPORTED;
    }
    hipStream_t s = as_stream(stream);
    if (v.logt == 6) {     // one wavefront per tile: csrc/dq_wave.hip
        if (ngrads >= 0) {
            if constexpr (is128) return wave_launch_grad_c128(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads);
            else return wave_launch_grad_c64(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads);
        }
        if constexpr (is128) return wave_launch_c128(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
        else return wave_launch_c64(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
    }
    if constexpr (!is128) {
        if (ngrads >= 0) {      // the reverse sweep: no next-tile prefetch (its registers go to the reductions)
            if (v.m == 12) launch_variant<float, 4, 8, false, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads);
            else launch_variant<float, 4, 9, false, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads);
            return check_launch("dq_apply_fused_grad");
        }
        if (v.m == 12) launch_variant<float, 4, 8, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
        else launch_variant<float, 4, 9, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
    } else {
        if (v.m == 11) launch_variant<double, 3, 8, false>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
        else launch_variant<double, 3, 9, false>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s);
    }
    return check_launch("dq_apply_fused");
}

}  // namespace dq

extern "C" int dq_fused_set_tiles_per_wg(int tiles) {
    if (tiles < 0 || tiles > 64 || (tiles & (tiles - 1))) {
        dq::set_error("dq_fused_set_tiles_per_wg: %d is not 0 (automatic) or a power of two <= 64", tiles);
        return DQ_ERR_ARG;
    }
    dq::g_tiles_per_wg.store(tiles, std::memory_order_relaxed);
    return DQ_OK;
}

extern "C" int dq_fused_geometry(int is_c128, int variant, int* m, int* slots, int* threads) {
    if (variant < 0 || variant >= (is_c128 ? dq::kNumVariantsC128 : dq::kNumVariantsC64)) {
        dq::set_error("dq_fused_geometry: variant %d out of range", variant);
        return DQ_ERR_ARG;
    }
    const dq::FusedVariant v = is_c128 ? dq::kVariantsC128[variant] : dq::kVariantsC64[variant];
    if (m) *m = v.m;
    if (slots) *slots = v.slots;
    if (threads) *threads = 1 << v.logt;
    return DQ_OK;
}

extern "C" int dq_apply_fused_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                  int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream);
}
extern "C" int dq_apply_fused_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                   int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream);
}

extern "C" int dq_apply_fused_grad_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                       int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads,
                                       dq_stream_t stream) {
    if (!grads || ngrads < 1) {
        dq::set_error("dq_apply_fused_grad_c64: no accumulator (grads = %p, ngrads = %lld)", (void*)grads, (long long)ngrads);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads);
}

extern "C" int dq_apply_fused_grad_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                        int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads,
                                        dq_stream_t stream) {
    if (!grads || ngrads < 1) {
        dq::set_error("dq_apply_fused_grad_c128: no accumulator (grads = %p, ngrads = %lld)", (void*)grads, (long long)ngrads);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads);
}

// Same pass, but `in` is ONE state (2^n amplitudes) shared by all `batch` outputs: the first pass of a batched
// circuit reads the circuit's initial state directly instead of `batch` materialised copies of it.
extern "C" int dq_apply_fused_bcast_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                        int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, true);
}
extern "C" int dq_apply_fused_bcast_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                         int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, true);
}
