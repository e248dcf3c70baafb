// Dense 2^k x 2^k gate (5 <= k <= 10) on the matrix cores: the one place on this path that is genuinely a GEMM
// (UAnyGate / LatentGate blocks on many wires, and through them QubitCircuit.get_unitary; reference:
// ArbitraryGate.get_unitary gate.py:318-330 -> evolve_state qmath.py:485-506, whose matmul has M = K = 2^k).
//
//     Y[r, c] = sum_j U[r, j] X[j, c]          r, j = patterns of the k target bits (matrix MSB = targets[0])
//                                              c    = every other index bit with the controls at 1 (x batch)
//
// 8 * 2^k flop per 16 bytes moved: HBM-bound for k = 5, 6 (16 .. 32 flop/B), MFMA-bound from k = 7 on.  The parity bar
// (1e-4 / 1e-10 against an f32 / f64 reference) rules out reduced-precision inputs, so the instructions are
// v_mfma_f32_16x16x4_f32 and v_mfma_f64_16x16x4_f64: exact f32 / f64 fused multiply-add chains at 64 flop/clk/SIMD
// (MI355X_MICROARCH.md: 157 TFLOP/s, what a packed-VALU kernel reaches only on paper).  The complex product is four
// real ones per 16x16x4 block: Yr += Ur Xr - Ui Xi, Yi += Ur Xi + Ui Xr.
//
// A workgroup of 4 waves owns a tile of (WM * 32) rows x (4 / WM * 32) columns; a wave owns 32 x 32 (2 x 2 MFMA
// blocks, re and im accumulators: 32 VGPRs).  The K loop walks the 2^k columns of U in chunks of 16: U's chunk and the
// gathered X chunk go through LDS as separate re / im planes (one ds_read per MFMA operand, padded rows: no bank
// conflicts), the NEXT chunk is already on its way from memory into registers while the MFMAs of this one run.
// Gather / scatter addressing (any target positions, controls as fixed ones) is done once per thread.
#include "dq_common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace dq {

struct DenseGeom {
    int n, k;
    int tpos[10];          // bit position of matrix index bit k-1-i  (targets[i], MSB first)
    BitList sorted;        // targets + controls ascending (for the column deposit)
    uint64_t cmask;        // control bits (set in every column)
    int colbits;           // n - k - nc
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    using acc_t = f32x4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D element `reg` of lane `l`: row, column inside the 16 x 16 block
    static __device__ __forceinline__ int row(int l, int reg) { return (l >> 4) * 4 + reg; }
};
template <> struct Mfma<double> {
    using acc_t = f64x4;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int l, int reg) { return (l >> 4) + 4 * reg; }   // (f64 has its own map)
};

// offset (in amplitudes) that row / column-of-U pattern `j` contributes: matrix index bit k-1-i -> position tpos[i]
__device__ __forceinline__ uint64_t target_offset(int j, const DenseGeom& g) {
    uint64_t o = 0;
    for (int i = 0; i < g.k; ++i) o |= (uint64_t)((j >> (g.k - 1 - i)) & 1) << g.tpos[i];
    return o;
}

// streaming (non-temporal) 8- / 16-byte accesses: a dense gate reads and writes the state once (DQ_DENSE_NT=0: plain)
template <typename T> struct NtVec;
template <> struct NtVec<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct NtVec<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T, bool NT> __device__ __forceinline__ cx<T> ld_amp(const cx<T>* p) {
    if constexpr (NT) {
        const typename NtVec<T>::type v = __builtin_nontemporal_load(reinterpret_cast<const typename NtVec<T>::type*>(p));
        return mk<T>(v.x, v.y);
    } else {
        return *p;
    }
}
template <typename T, bool NT> __device__ __forceinline__ void st_amp(cx<T>* p, T re, T im) {
    if constexpr (NT) {
        typename NtVec<T>::type v;
        v.x = re, v.y = im;
        __builtin_nontemporal_store(v, reinterpret_cast<typename NtVec<T>::type*>(p));
    } else {
        *p = mk<T>(re, im);
    }
}

// BM x BN = 16 x 16 blocks of a wave's tile: 2 x 2 (32 x 32; 64 x 64 per workgroup) or 4 x 4 (64 x 64; 128 x 128 per
// workgroup -- half the traffic from the caches per MFMA: every workgroup re-reads its rows of U and its columns of X)
template <typename T, int WM, bool NT, int BM = 2, int BN = 2>
__global__ __launch_bounds__(256) void apply_dense_mfma_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                               const cx<T>* __restrict__ mats, int64_t mat_bstride,
                                                               DenseGeom g, uint64_t ncols, int col_sample_shift) {
    constexpr int WN = 4 / WM;
    constexpr int TM = WM * BM * 16, TN = WN * BN * 16;      // workgroup tile: rows of U x columns
#ifndef DQ_DENSE_KC_F32
#define DQ_DENSE_KC_F32 16
#endif
    constexpr int KC = sizeof(T) == 4 ? DQ_DENSE_KC_F32 : 32;    // K chunk (complex columns of U per stage; measured: 32 costs f32 2 %, gains f64 5 %)
    constexpr int APAD = KC + 1, BPAD = TN + 1;     // padded row lengths of the LDS planes (in elements)
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    __shared__ T sAr[TM * APAD], sAi[TM * APAD], sBr[KC * BPAD], sBi[KC * BPAD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int D = 1 << g.k;
    // col_sample_shift bit 30 (host flag): row tiles vary fastest in dispatch order -- the workgroups that share a column
    // tile of X run back to back (X is then read from HBM once and from the caches otherwise; U lives in the Infinity Cache)
    const bool rows_fast = (col_sample_shift & 0x40000000) != 0 && col_sample_shift >= 0;
    if (col_sample_shift >= 0) col_sample_shift &= 0x3fffffff;
    const unsigned nrt = (unsigned)((1u << g.k) / TM);
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int row0 = (rows_fast ? (int)(lin % nrt) : (int)blockIdx.y) * TM;
    const uint64_t col0 = (uint64_t)(rows_fast ? lin / nrt : blockIdx.x) * TN;
    const int64_t zb = blockIdx.z;                  // sample (when every sample has its own matrix)
    const cx<T>* U = mats + zb * mat_bstride;

    // ---- staging assignment -----------------------------------------------------------------------------------
    // A (U chunk): TM rows x KC complex = TM * KC / 256 elements per thread, consecutive k of one row
    constexpr int A_PER = TM * KC / 256;            // 2 (WM = 1) or 4 (WM = 2)
    const int a_row = (tid * A_PER) / KC, a_k = (tid * A_PER) % KC;
    // B (X chunk): KC x TN complex = KC * TN / 256 per thread: column = tid % TN, k = tid / TN + i * (256 / TN)
    constexpr int B_PER = KC * TN / 256;            // 8 (TN = 128) or 4 (TN = 64)
    constexpr int B_KSTEP = 256 / TN;               // 2 or 4
    const int b_col = tid % TN, b_k0 = tid / TN;
    // where column col0 + b_col lives: sample (shared matrix: the batch is more columns) and amplitude base
    const uint64_t my_col = col0 + (uint64_t)b_col;
    const bool col_ok = my_col < ncols;
    uint64_t col_base = 0;
    {
        const uint64_t c = col_ok ? my_col : 0;
        const uint64_t sample = col_sample_shift >= 0 ? (c >> col_sample_shift) : (uint64_t)zb;
        const uint64_t within = col_sample_shift >= 0 ? (c & ((1ull << col_sample_shift) - 1ull)) : c;
        col_base = (sample << g.n) + (insert_zeros(within, g.sorted) | g.cmask);
    }

    constexpr int DEPTH = 1;      // chunks in flight (two: measured SLOWER, 104 vs 116 TFLOP/s at k = 10 -- a wave per SIMD less)
    cx<T> pa[DEPTH][A_PER], pb[DEPTH][B_PER];
    // offsets of the rows of X a thread fetches: the pattern's bits below KC once per thread, the bits above once per
    // chunk (uniform) -- computed per load (a loop over the k target bits in 64-bit arithmetic) the addressing cost half
    // as many issue slots as the MFMAs it feeds
    uint64_t boff[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) boff[i] = col_base + target_offset(b_k0 + i * B_KSTEP, g);
    auto fetch = [&](int k0, cx<T> (&qa)[A_PER], cx<T> (&qb)[B_PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) qa[i] = U[(int64_t)(row0 + a_row) * D + k0 + a_k + i];
        const uint64_t hi = target_offset(__builtin_amdgcn_readfirstlane(k0), g);      // (k0 is a multiple of KC)
#pragma unroll
        for (int i = 0; i < B_PER; ++i) qb[i] = col_ok ? ld_amp<T, NT>(in + (boff[i] | hi)) : mk<T>(0, 0);
    };
    auto stash = [&](const cx<T> (&qa)[A_PER], const cx<T> (&qb)[B_PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            sAr[a_row * APAD + a_k + i] = qa[i].x;
            sAi[a_row * APAD + a_k + i] = qa[i].y;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            sBr[(b_k0 + i * B_KSTEP) * BPAD + b_col] = qb[i].x;
            sBi[(b_k0 + i * B_KSTEP) * BPAD + b_col] = qb[i].y;
        }
    };

    acc_t cr[BM][BN], ci[BM][BN];
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
        for (int b = 0; b < BN; ++b) cr[a][b] = ci[a][b] = acc_t{0, 0, 0, 0};

    const int l15 = lane & 15, l4 = lane >> 4;
    fetch(0, pa[0], pb[0]);
    if constexpr (DEPTH == 2)
        if (KC < D) fetch(KC, pa[1], pb[1]);
    for (int k0 = 0, par = 0; k0 < D; k0 += KC, par ^= (DEPTH - 1)) {
#ifndef DQ_DENSE_ABL_NOBAR
        __syncthreads();                            // everybody is done with the previous chunk
#endif
        if (par == 0) stash(pa[0], pb[0]);
        else stash(pa[DEPTH - 1], pb[DEPTH - 1]);
#ifndef DQ_DENSE_ABL_NOBAR
        __syncthreads();
#endif
#ifndef DQ_DENSE_ABL_NOFETCH
        if (k0 + DEPTH * KC < D) {                  // DEPTH chunks ahead: in flight while the matrix cores work
            if (par == 0) fetch(k0 + DEPTH * KC, pa[0], pb[0]);
            else fetch(k0 + DEPTH * KC, pa[DEPTH - 1], pb[DEPTH - 1]);
        }
#endif
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            T ar[BM], ai[BM], nai[BM], br[BN], bi[BN];
#pragma unroll
            for (int a = 0; a < BM; ++a) {          // A[i = l & 15][k = l >> 4]
                const int r = wm * BM * 16 + a * 16 + l15;
#ifdef DQ_DENSE_ABL_NOLDS      // (timing experiment: operands from registers, wrong results)
                ar[a] = (T)(r + ks) * (T)1e-3, ai[a] = (T)l4;
#else
                ar[a] = sAr[r * APAD + ks + l4];
                ai[a] = sAi[r * APAD + ks + l4];
#endif
                nai[a] = -ai[a];
            }
#pragma unroll
            for (int b = 0; b < BN; ++b) {          // B[k = l >> 4][j = l & 15]
                const int c = wn * BN * 16 + b * 16 + l15;
#ifdef DQ_DENSE_ABL_NOLDS
                br[b] = (T)c * (T)1e-3, bi[b] = (T)(ks + l4);
#else
                br[b] = sBr[(ks + l4) * BPAD + c];
                bi[b] = sBi[(ks + l4) * BPAD + c];
#endif
            }
            // 16 MFMAs; consecutive ones never touch the same accumulator
#pragma unroll
            for (int a = 0; a < BM; ++a)
#pragma unroll
                for (int b = 0; b < BN; ++b) {
                    cr[a][b] = M::run(ar[a], br[b], cr[a][b]);
                    ci[a][b] = M::run(ar[a], bi[b], ci[a][b]);
                }
#pragma unroll
            for (int a = 0; a < BM; ++a)
#pragma unroll
                for (int b = 0; b < BN; ++b) {
                    cr[a][b] = M::run(nai[a], bi[b], cr[a][b]);
                    ci[a][b] = M::run(ai[a], br[b], ci[a][b]);
                }
        }
    }

    // ---- scatter: lane holds column (lane & 15) of each block, four rows per block ------------------------------
    const uint64_t off_rowlane = target_offset(M::row(lane, 0), g);
#pragma unroll
    for (int b = 0; b < BN; ++b) {
        const uint64_t c = col0 + (uint64_t)(wn * BN * 16 + b * 16 + l15);
        if (c >= ncols) continue;
        const uint64_t sample = col_sample_shift >= 0 ? (c >> col_sample_shift) : (uint64_t)zb;
        const uint64_t within = col_sample_shift >= 0 ? (c & ((1ull << col_sample_shift) - 1ull)) : c;
        cx<T>* po = out + (sample << g.n) + (insert_zeros(within, g.sorted) | g.cmask);
        // row = (row0 + 16 (wm BM + a)) + (lane part) + (reg part): three disjoint bit fields of the row pattern, and the offset
        // of a pattern is bitwise linear in it -- the lane part once per thread, the rest per WAVE in scalar registers
        // (round 6: the 16 full `target_offset(r)` per tile, ~60 64-bit vector operations each, were the fixed per-tile cost
        // that kept k = 6 / 7 at 0.50 / 0.61 of the MFMA peak: profiles/r06/exp_dense_ustat.txt)
#pragma unroll
        for (int a = 0; a < BM; ++a)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ru = __builtin_amdgcn_readfirstlane(row0 + wm * BM * 16 + a * 16) + (M::row(0, reg) - M::row(0, 0));
                const uint64_t ou = target_offset(ru, g);
                const uint64_t ous = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ou >> 32)) << 32) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ou);
                st_amp<T, NT>(po + (ous | off_rowlane), cr[a][b][reg], ci[a][b][reg]);
            }
    }
}

// ---- k = 5, 6: no LDS for the state, no barrier in the loop -----------------------------------------------------------
// 32 x 32 (64 x 64) is small enough for the MFMA operand layout to be filled straight from memory: lane (j = l & 15, q = l >> 4)
// of a 16x16x4 block wants B[k = 4 s + q][column j] -- for complex64 a 16-byte load brings columns 2 j and 2 j + 1 (any
// 16 columns can form a block: the even ones are block 0, the odd ones block 1), for complex128 one column.  A wave owns
// a group of 32 (16) columns: eight 16-byte loads per lane, 128 (64) MFMAs, eight 16-byte stores, the next group's loads
// already in flight; U sits in LDS as re / im planes (read per k step), written once per workgroup.  8 * 32 flop per 16
// bytes: the memory side has to stream at >= 4.4 TB/s while the matrix cores run at >= 0.45 of their f32 peak.
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int CPL = 2; };
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int CPL = 1; };

template <typename T, int KB, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 && KB == 5 ? 3 : 2, sizeof(T) == 4 && KB == 5 ? 3 : 2)))
void apply_dense56_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ mats,
                          int64_t mat_bstride, DenseGeom g, uint64_t ngroups, int col_sample_shift) {
    constexpr int D = 1 << KB, CPL = Vec16<T>::CPL, CG = 16 * CPL, PAD = D + 1;
    constexpr int RB = 2, KS = D / 4;               // row blocks of 16 per wave, k steps of 4
    constexpr int H = D / 32;                       // waves that share a column group (k = 6: two, 32 rows each)
    constexpr bool PREFETCH = KB == 5;              // (k = 6: the operands of one group fill the registers)
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    using V = typename Vec16<T>::type;
    __shared__ T sUr[D * PAD], sUi[D * PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t zb = blockIdx.y;
    const cx<T>* U = mats + zb * mat_bstride;
    for (int i = tid; i < D * D; i += 256) {
        const cx<T> u = U[i];
        sUr[(i >> KB) * PAD + (i & (D - 1))] = u.x;
        sUi[(i >> KB) * PAD + (i & (D - 1))] = u.y;
    }
    __syncthreads();
    const int l15 = lane & 15, l4 = lane >> 4;
    // offsets (in amplitudes) of the matrix-index patterns this lane touches
    const uint64_t off_l4 = target_offset(l4, g);                                  // k = 4 s + l4: the low two bits
    uint64_t off_s[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) off_s[s_] = target_offset(4 * s_, g);
    auto col_base = [&](uint64_t grp) __attribute__((always_inline)) {
        const uint64_t c = grp * CG + (uint64_t)l15 * CPL;
        const uint64_t sample = col_sample_shift >= 0 ? (c >> col_sample_shift) : (uint64_t)zb;
        const uint64_t within = col_sample_shift >= 0 ? (c & ((1ull << col_sample_shift) - 1ull)) : c;
        return (sample << g.n) + (insert_zeros(within, g.sorted) | g.cmask);
    };
    auto fetch = [&](V (&b)[KS], uint64_t base) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const V* p = reinterpret_cast<const V*>(in + base + (off_s[s_] | off_l4));
            if constexpr (NT) b[s_] = __builtin_nontemporal_load(p);
            else b[s_] = *p;
        }
    };
    const int row0 = (wave % H) * 32;
    const uint64_t off_rowlane56 = target_offset(M::row(lane, 0), g);
    const uint64_t stride = (uint64_t)gridDim.x * (4u / H);
    uint64_t grp = (uint64_t)blockIdx.x * (4u / H) + (uint64_t)(wave / H);
    V b[KS], bn[PREFETCH ? KS : 1];
    uint64_t base = 0;
    if (PREFETCH && grp < ngroups) {
        base = col_base(grp);
        fetch(b, base);
    }
    for (; grp < ngroups; grp += stride) {
        uint64_t nbase = 0;
        if constexpr (PREFETCH) {
            nbase = grp + stride < ngroups ? col_base(grp + stride) : 0;
            if (grp + stride < ngroups) fetch(bn, nbase);        // in flight while the matrix cores work on this group
        } else {
            base = col_base(grp);
            fetch(b, base);
        }
        acc_t cr[RB][CPL], ci[RB][CPL];
#pragma unroll
        for (int a = 0; a < RB; ++a)
#pragma unroll
            for (int c = 0; c < CPL; ++c) cr[a][c] = ci[a][c] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            T ar[RB], ai[RB];
#pragma unroll
            for (int a = 0; a < RB; ++a) {          // A[i = l & 15][k = 4 s + (l >> 4)]
                ar[a] = sUr[(row0 + a * 16 + l15) * PAD + 4 * s_ + l4];
                ai[a] = sUi[(row0 + a * 16 + l15) * PAD + 4 * s_ + l4];
            }
#pragma unroll
            for (int a = 0; a < RB; ++a)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const T br = b[s_][2 * c], bi = b[s_][2 * c + 1];
                    cr[a][c] = M::run(ar[a], br, cr[a][c]);
                    ci[a][c] = M::run(ar[a], bi, ci[a][c]);
                    cr[a][c] = M::run(-ai[a], bi, cr[a][c]);
                    ci[a][c] = M::run(ai[a], br, ci[a][c]);
                }
        }
#pragma unroll
        for (int a = 0; a < RB; ++a)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                V v;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    v[2 * c] = cr[a][c][reg];
                    v[2 * c + 1] = ci[a][c][reg];
                }
                // (the row pattern's offset: uniform part per wave in scalar registers, lane part once per thread -- see the
                // staged kernel's scatter)
                const uint64_t ou = target_offset(__builtin_amdgcn_readfirstlane(row0 + a * 16) + (M::row(0, reg) - M::row(0, 0)), g);
                const uint64_t ous = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ou >> 32)) << 32) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ou);
                V* p = reinterpret_cast<V*>(out + base + (ous | off_rowlane56));
                if constexpr (NT) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        if constexpr (PREFETCH) {
            base = nbase;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) b[s_] = bn[s_];
        }
    }
}

template <typename T>
int apply_dense_mfma(const cx<T>* in, cx<T>* out, const cx<T>* mats, int64_t mat_bstride, int n, const int* targets, int k,
                     const int* controls, int nc, const BitList& sorted, uint64_t cmask, int64_t batch, hipStream_t s) {
    DenseGeom g;
    g.n = n;
    g.k = k;
    for (int i = 0; i < k; ++i) g.tpos[i] = targets[i];
    g.sorted = sorted;
    g.cmask = cmask;
    g.colbits = n - k - nc;
    const int D = 1 << k;
    // one matrix for all samples: the batch is just more columns of X
    const bool shared = mat_bstride == 0;
    const uint64_t ncols = (shared ? (uint64_t)batch : 1ull) << g.colbits;
    const int shift = shared ? g.colbits : -1;
    const unsigned gz = shared ? 1u : (unsigned)batch;
    // states far beyond the Infinity Cache stream through (nothing is reused after the pass): non-temporal accesses
    static const int nt_env = [] { const char* e = getenv("DQ_DENSE_NT"); return e ? atoi(e) : -1; }();
    const bool nt = nt_env >= 0 ? nt_env != 0 : ((uint64_t)batch << n) * sizeof(cx<T>) >= (1ull << 30);
    static const int d5_env = [] { const char* e = getenv("DQ_DENSE5"); return e ? atoi(e) : 1; }();
    constexpr int CG = 16 * Vec16<T>::CPL;
    // (complex64: a 16-byte access is two neighbouring columns -- index bit 0 must be a column bit)
    // (k = 6 in complex128: the operands of 32 rows x 16 columns do not fit the registers -- the LDS-staged kernel stays)
    if ((D == 32 || (D == 64 && sizeof(T) == 4)) && d5_env && ncols % CG == 0 && (sizeof(T) == 8 || sorted.n == 0 || sorted.pos[0] != 0)) {
        const uint64_t ngroups = ncols / CG;
        static const int blk_env = [] { const char* e = getenv("DQ_DENSE5_BLOCKS"); return e ? atoi(e) : 0; }();
        // = the resident workgroups: every wave loops over its share of the column groups
        const uint64_t resident = 256ull * (sizeof(T) == 4 && D == 32 ? 3ull : 2ull);
        const uint64_t per_wg = D == 32 ? 4 : 2;       // column groups a workgroup works on at a time
        const unsigned blocks = (unsigned)std::min<uint64_t>((ngroups + per_wg - 1) / per_wg, blk_env > 0 ? (uint64_t)blk_env : resident);
        dim3 grid(blocks, gz);
        if (D == 32) {
            if (nt) hipLaunchKernelGGL((apply_dense56_kernel<T, 5, true>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ngroups, shift);
            else hipLaunchKernelGGL((apply_dense56_kernel<T, 5, false>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ngroups, shift);
        } else {
            if (nt) hipLaunchKernelGGL((apply_dense56_kernel<T, 6, true>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ngroups, shift);
            else hipLaunchKernelGGL((apply_dense56_kernel<T, 6, false>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ngroups, shift);
        }
    } else if (D == 32) {
        constexpr int TN = 128;
        dim3 grid((unsigned)((ncols + TN - 1) / TN), 1, gz);
        if (nt) hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 1, true>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift);
        else hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 1, false>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift);
    } else {
        static const int big_env = [] { const char* e = getenv("DQ_DENSE_BIG"); return e ? atoi(e) : 0; }();      // (measured: 100 vs 116 TFLOP/s at k = 10 -- two waves per SIMD hide less than the halved cache traffic saves)
        if (sizeof(T) == 4 && D >= 256 && big_env && ncols % 128 == 0) {      // complex64, k >= 8: 128 x 128 per workgroup
            dim3 grid((unsigned)(ncols / 128), (unsigned)(D / 128), gz);
            const int shift3 = shift >= 0 ? (shift | 0x40000000) : shift;
            if (nt) hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 2, true, 4, 4>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift3);
            else hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 2, false, 4, 4>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift3);
            (void)controls;
            return DQ_OK;
        }
        constexpr int TN = 64;
        dim3 grid((unsigned)((ncols + TN - 1) / TN), (unsigned)(D / 64), gz);
        static const int rf_env = [] { const char* e = getenv("DQ_DENSE_ROWS_FAST"); return e ? atoi(e) : 1; }();
        const int shift2 = (rf_env && shift >= 0) ? (shift | 0x40000000) : shift;
        if (nt) hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 2, true>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift2);
        else hipLaunchKernelGGL((apply_dense_mfma_kernel<T, 2, false>), grid, dim3(256), 0, s, in, out, mats, mat_bstride, g, ncols, shift2);
    }
    (void)controls;
    return DQ_OK;
}

template int apply_dense_mfma<float>(const cx<float>*, cx<float>*, const cx<float>*, int64_t, int, const int*, int, const int*,
                                     int, const BitList&, uint64_t, int64_t, hipStream_t);
template int apply_dense_mfma<double>(const cx<double>*, cx<double>*, const cx<double>*, int64_t, int, const int*, int,
                                      const int*, int, const BitList&, uint64_t, int64_t, hipStream_t);

}  // namespace dq
