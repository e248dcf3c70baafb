// Reductions over the statevector: Pauli-string expectation, inner product, probabilities,
// marginals and the gate-matrix gradient.  All are single-pass HBM-bound streams; accumulation is
// in double precision regardless of the state's precision.
//
// Replaces (reference, src/deepquantum/): qmath.expectation (qmath.py:830-860) together with
// Observable.forward (layer.py:127-165) -- there one full gate pass per Pauli factor plus a bmm;
// here one read of the state -- the |psi|^2 / permute / sum of qmath.measure (qmath.py:624-626),
// inner_product_dist's local part (distributed.py:288-291) and the matmul backward that autograd
// derives for qmath.py:504.
#include <cstring>
#include "dq_common.hpp"

namespace dq {

constexpr int RED_BLOCKS = 1024;  // partial sums per batch sample
constexpr int RED_THREADS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum (re, im) over the block; result valid in thread 0.
__device__ __forceinline__ void block_sum2(double& re, double& im) {
    __shared__ double sre[RED_THREADS / 64], sim[RED_THREADS / 64];
    re = wave_sum(re);
    im = wave_sum(im);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sre[w] = re;
        sim[w] = im;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int i = 0; i < RED_THREADS / 64; ++i) {
            a += sre[i];
            b += sim[i];
        }
        re = a;
        im = b;
    }
}

// ---- <psi|P|psi> ---------------------------------------------------------------------------------
// P|j> = i^ny (-1)^{popc(j & zmask)} |j ^ xmask>.  For xmask != 0 amplitudes are visited in pairs
// (i, k = i ^ xmask) so every amplitude is read exactly once:
//   term(i) + term(k) = s_k conj(psi_i) psi_k + s_i conj(psi_k) psi_i,   s_j = (-1)^{popc(j & zmask)}.
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void expect_pauli_kernel(const cx<T>* __restrict__ psi, uint64_t xmask,
                                                                    uint64_t zmask, int n, double* __restrict__ ws) {
    const int64_t b = blockIdx.y;
    const cx<T>* p = psi + ((uint64_t)b << n);
    double re = 0, im = 0;
    if (xmask == 0) {
        const uint64_t dim = 1ull << n;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim;
             i += (uint64_t)gridDim.x * blockDim.x) {
            const cx<T> a = p[i];
            const double pr = (double)a.x * a.x + (double)a.y * a.y;
            re += (__popcll(i & zmask) & 1) ? -pr : pr;
        }
    } else {
        const int low = __ffsll((long long)xmask) - 1;
        const uint64_t half = 1ull << (n - 1);
        for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < half;
             g += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t i = insert_zero(g, low);
            const uint64_t k = i ^ xmask;
            const cx<T> a = p[i], c = p[k];
            // conj(a) * c
            const double cr = (double)a.x * c.x + (double)a.y * c.y;
            const double ci = (double)a.x * c.y - (double)a.y * c.x;
            const double sk = (__popcll(k & zmask) & 1) ? -1.0 : 1.0;
            const double si = (__popcll(i & zmask) & 1) ? -1.0 : 1.0;
            // s_k * (cr + i ci) + s_i * (cr - i ci)
            re += (sk + si) * cr;
            im += (sk - si) * ci;
        }
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        double* w = ws + ((size_t)b * RED_BLOCKS + blockIdx.x) * 2;
        w[0] = re;
        w[1] = im;
    }
}

// mode 0: out[b] = Re(i^ny * S); mode 1: out[2b], out[2b+1] = S
__global__ __launch_bounds__(RED_THREADS) void finish_kernel(const double* __restrict__ ws, int nblocks, int ny,
                                                              int mode, double* __restrict__ out) {
    const int64_t b = blockIdx.x;
    double re = 0, im = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        re += ws[((size_t)b * RED_BLOCKS + i) * 2];
        im += ws[((size_t)b * RED_BLOCKS + i) * 2 + 1];
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        if (mode == 0) {
            double r;
            switch (ny & 3) {
                case 0: r = re; break;
                case 1: r = -im; break;  // Re(i (re + i im)) = -im
                case 2: r = -re; break;
                default: r = im; break;  // Re(-i (re + i im)) = im
            }
            out[b] = r;
        } else {
            out[2 * b] = re;
            out[2 * b + 1] = im;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(RED_THREADS) void inner_kernel(const cx<T>* __restrict__ bra, const cx<T>* __restrict__ ket,
                                                             uint64_t count, double* __restrict__ ws) {
    const int64_t b = blockIdx.y;
    const cx<T>* pa = bra + (uint64_t)b * count;
    const cx<T>* pb = ket + (uint64_t)b * count;
    double re = 0, im = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const cx<T> a = pa[i], c = pb[i];
        re += (double)a.x * c.x + (double)a.y * c.y;
        im += (double)a.x * c.y - (double)a.y * c.x;
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        double* w = ws + ((size_t)b * RED_BLOCKS + blockIdx.x) * 2;
        w[0] = re;
        w[1] = im;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void probs_kernel(const cx<T>* __restrict__ psi, T* __restrict__ probs, uint64_t count) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const cx<T> a = psi[i];
        probs[i] = a.x * a.x + a.y * a.y;
    }
}

// Marginals: a block owns a "chunk" of 2^c amplitudes (c = min(12, n)) whose index bits are the low L bits
// (1 KiB of contiguous state) plus UNMEASURED bits above them as far as there are any (then measured ones, lowest outcome
// bit first), so as much of the sum over the unmeasured bits as possible happens inside the block.  A thread holds the
// 16 amplitudes that differ in the chunk-local bits 8..11 (the unmeasured candidates go there first), issues all 16
// loads before it uses any, adds them up, and adds the sum into an LDS histogram over the measured chunk bits; the
// block then adds the histogram to the rows of `out` that the measured bits outside the chunk select.  The chunk NUMBER
// counts through the unmeasured bits outside the chunk first, wherever they are: a run of 2^run chunk numbers shares its
// rows of `out`, is taken by one block, and costs one set of 2^nlo atomics.
struct MargGeom {                       // unused entries are padded so that the kernel needs no guards (see below)
    int c, nlo, run, exclusive;
    unsigned qmask;                     // measured ones among the four chunk-local bits a thread holds itself
    uint8_t pos[12];                    // chunk-local bit i <-> index bit pos[i]                     (pad: 62)
    uint8_t cpos[28];                   // bit t of the chunk number <-> index bit cpos[t]: the unmeasured bits outside
                                        // the chunk first (a run of chunks shares its result rows)  (pad: 63)
    uint8_t lo_x[12], lo_out[12];       // measured chunk-local bit -> bit of the outcome index       (pad: 31, 0)
    uint8_t hi_pos[40], hi_out[40];     // measured index bit outside the chunk -> outcome bit        (pad: 63, 0)
};

template <typename T>
__global__ __launch_bounds__(256) void marginal_chunk_kernel(const cx<T>* __restrict__ psi, int n, int nw, MargGeom g,
                                                             double* __restrict__ out) {
    extern __shared__ double hist[];
    const unsigned nloc = 1u << g.nlo;
    for (unsigned j = threadIdx.x; j < nloc; j += 256) hist[j] = 0.0;
    __syncthreads();
    // every loop over the geometry has a constant trip count and no guard: the arrays are kernel arguments, and a
    // run-time index or a conditional access would turn them into one scalar load (and one wait) per byte
    // chunk-local index x = threadIdx.x | k << 8: the thread's 16 amplitudes differ in the chunk's top four bits
    const unsigned xt = threadIdx.x, nx = 1u << g.c;
    uint64_t off_t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) off_t |= (uint64_t)((xt >> i) & 1u) << g.pos[i];
    unsigned jt = 0;
#pragma unroll
    for (int t = 0; t < 12; ++t) jt |= ((xt >> g.lo_x[t]) & 1u) << t;
    uint64_t off_k[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        off_k[k] = 0;
#pragma unroll
        for (int i = 8; i < 12; ++i) off_k[k] |= (uint64_t)((k >> (i - 8)) & 1) << g.pos[i];
    }
    // 2^run consecutive chunks differ in unmeasured bits only: they go into the same rows of `out`, so one workgroup
    // takes them all and adds its histogram to `out` once
    uint64_t base0 = 0;
    double s = 0;
    for (uint64_t ci = (uint64_t)blockIdx.x << g.run; ci < ((uint64_t)blockIdx.x + 1) << g.run; ++ci) {
        uint64_t base = 0;
#pragma unroll
        for (int t = 0; t < 28; ++t) base |= ((ci >> t) & 1ull) << g.cpos[t];
        if (ci == (uint64_t)blockIdx.x << g.run) base0 = base;
        const cx<T>* p = psi + ((uint64_t)blockIdx.y << n) + base;
        cx<T> a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = p[(xt | (k << 8)) < nx ? off_t + off_k[k] : 0];
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            v[k] = (xt | (k << 8)) < nx ? (double)a[k].x * a[k].x + (double)a[k].y * a[k].y : 0.0;
        if (g.qmask == 0) {             // none of the thread's own bits is measured: one sum, one histogram bin
#pragma unroll
            for (int k = 0; k < 16; ++k) s += v[k];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned j = jt;
#pragma unroll
                for (int t = 0; t < 12; ++t) j |= ((((unsigned)k << 8) >> g.lo_x[t]) & 1u) << t;
                if ((xt | (k << 8)) < nx) unsafeAtomicAdd(&hist[j], v[k]);
            }
        }
    }
    if (g.qmask == 0) {
        if (g.nlo == 0) {               // no measured bit in the chunk at all: a plain block sum
            double dummy = 0;
            block_sum2(s, dummy);
            if (threadIdx.x == 0) hist[0] = s;
        } else {
            unsafeAtomicAdd(&hist[jt], s);
        }
    }
    __syncthreads();
    uint64_t hi = 0;
#pragma unroll
    for (int t = 0; t < 40; ++t) hi |= ((base0 >> g.hi_pos[t]) & 1ull) << g.hi_out[t];
    double* row = out + ((size_t)blockIdx.y << nw);
    for (unsigned j = threadIdx.x; j < nloc; j += 256) {
        uint64_t o = hi;
#pragma unroll
        for (int t = 0; t < 12; ++t) o |= (uint64_t)((j >> t) & 1u) << g.lo_out[t];
        if (g.exclusive) row[o] = hist[j];      // every bit outside the chunk is measured: nobody else adds to this row
        else unsafeAtomicAdd(row + o, hist[j]);
    }
}

struct GradGeom {
    BitList sorted;
    int tpos[2];
    uint64_t cmask;
    int n;
};

template <typename T, int K>
__global__ __launch_bounds__(RED_THREADS) void gate_grad_kernel(const cx<T>* __restrict__ x, const cx<T>* __restrict__ gy,
                                                                 GradGeom g, uint64_t groups, double* __restrict__ gU) {
    constexpr int D = 1 << K;
    const int64_t b = blockIdx.y;
    const cx<T>* px = x + ((uint64_t)b << g.n);
    const cx<T>* pg = gy + ((uint64_t)b << g.n);
    uint64_t offs[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        uint64_t o = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) o |= (uint64_t)((j >> (K - 1 - i)) & 1) << g.tpos[i];
        offs[j] = o;
    }
    double are[D * D], aim[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) are[i] = aim[i] = 0;
    for (uint64_t gi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups;
         gi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = insert_zeros(gi, g.sorted) | g.cmask;
        cx<T> xv[D], gv[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            xv[j] = px[base | offs[j]];
            gv[j] = pg[base | offs[j]];
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {  // gy[i] * conj(x[j])
                are[i * D + j] += (double)gv[i].x * xv[j].x + (double)gv[i].y * xv[j].y;
                aim[i * D + j] += (double)gv[i].y * xv[j].x - (double)gv[i].x * xv[j].y;
            }
    }
#pragma unroll
    for (int i = 0; i < D * D; ++i) {
        double re = are[i], im = aim[i];
        block_sum2(re, im);
        if (threadIdx.x == 0) {
            double* dst = gU + ((size_t)b * D * D + i) * 2;
            unsafeAtomicAdd(dst, re);
            unsafeAtomicAdd(dst + 1, im);
        }
    }
}

static int sort_bits(const int* bits, int nb, BitList& out) {
    out.n = nb;
    for (int i = 0; i < nb; ++i) out.pos[i] = bits[i];
    for (int i = 1; i < nb; ++i) {
        int v = out.pos[i], j = i - 1;
        while (j >= 0 && out.pos[j] > v) {
            out.pos[j + 1] = out.pos[j];
            --j;
        }
        out.pos[j + 1] = v;
    }
    return 0;
}

static unsigned red_blocks(uint64_t work) {
    uint64_t nb = (work + RED_THREADS - 1) / RED_THREADS;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

template <typename T>
static int expect_impl(const void* psi, uint64_t xmask, uint64_t zmask, int n, int64_t batch, double* out, void* ws,
                       dq_stream_t stream) {
    if (!psi || !out || !ws || n < 1 || n > 40 || batch < 1 || batch > 65535) {
        set_error("dq_expect_pauli: bad argument (n=%d batch=%lld)", n, (long long)batch);
        return DQ_ERR_ARG;
    }
    const uint64_t full = (n == 64) ? ~0ull : ((1ull << n) - 1ull);
    if ((xmask | zmask) & ~full) {
        set_error("dq_expect_pauli: mask has bits >= n");
        return DQ_ERR_ARG;
    }
    const uint64_t work = xmask ? (1ull << (n - 1)) : (1ull << n);
    const unsigned nb = red_blocks(work);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(expect_pauli_kernel<T>, dim3(nb, (unsigned)batch), dim3(RED_THREADS), 0, s,
                       static_cast<const cx<T>*>(psi), xmask, zmask, n, static_cast<double*>(ws));
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)batch), dim3(RED_THREADS), 0, s, static_cast<const double*>(ws),
                       (int)nb, __builtin_popcountll(xmask & zmask), 0, out);
    return check_launch("dq_expect_pauli");
}


// ---- several Z-type Pauli expectations from one read of the state ----------------------------------------
// <psi| Z..Z |psi> = sum_i |psi_i|^2 (-1)^{popc(i & zmask)}: a Hamiltonian made of many such strings (MaxCut /
// Ising cost functions: one ZZ term per edge) costs ONE pass instead of one per term.  Workgroups stride over the
// state and write one row of partial sums each; the caller adds the rows.
constexpr int ZM_MAX = 32;
struct ZMasks {
    int k;
    uint64_t m[ZM_MAX];
};

template <typename T>
__global__ __launch_bounds__(RED_THREADS) void expect_zmulti_kernel(const cx<T>* __restrict__ psi, ZMasks z, int n,
                                                                     double* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const cx<T>* p = psi + ((uint64_t)b << n);
    double acc[ZM_MAX];
#pragma unroll
    for (int k = 0; k < ZM_MAX; ++k) acc[k] = 0;
    const uint64_t dim = 1ull << n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (uint64_t)gridDim.x * blockDim.x) {
        const cx<T> a = p[i];
        const double pr = (double)a.x * a.x + (double)a.y * a.y;
#pragma unroll
        for (int k = 0; k < ZM_MAX; ++k)
            if (k < z.k) acc[k] += (__popcll(i & z.m[k]) & 1) ? -pr : pr;
    }
#pragma unroll
    for (int k = 0; k < ZM_MAX; k += 2) {
        if (k < z.k) {
            double u = acc[k], v = (k + 1 < ZM_MAX) ? acc[k + 1] : 0.0;
            block_sum2(u, v);
            if (threadIdx.x == 0) {
                double* dst = out + ((size_t)b * gridDim.x + blockIdx.x) * z.k;
                dst[k] = u;
                if (k + 1 < z.k) dst[k + 1] = v;
            }
        }
    }
}

// The same sums as a matrix product on the f64 matrix cores: for the 64 consecutive amplitudes i = i0 + 16 q + j a wave
// holds (lane 16 q + j), (-1)^{popc(i & m_k)} = s_k(i0 + 16 q) s_k(j), so
//     D[k][j] += sum_q A[k][q] B[q][j],   A[k][q] = s_k(i0 + 16 q) = +-1,   B[q][j] = |psi_i|^2
// is one v_mfma_f64_16x16x4_f64 per 16 strings (B is exactly what the lane computed from its own load; A costs a lane an
// and / popcount of ITS string's mask), and the factor s_k(j) is applied once, after the loop, before the sum over j.
// Exact +-1 factors and f64 accumulation: the same numbers as the loop above, without its select per string and
// amplitude -- 32 strings cost two matrix instructions (64 clocks) per 64 amplitudes.
typedef double zm_f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(RED_THREADS) void expect_zmulti_mfma_kernel(const cx<T>* __restrict__ psi, ZMasks z, int n,
                                                                          double* __restrict__ out) {
    __shared__ uint64_t sm[ZM_MAX];
    __shared__ double part[RED_THREADS / 64][ZM_MAX];
#pragma unroll
    for (int k = 0; k < ZM_MAX; ++k)
        if (threadIdx.x == k) sm[k] = z.m[k];
    __syncthreads();
    const int64_t b = blockIdx.y;
    const cx<T>* p = psi + ((uint64_t)b << n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const uint64_t m0 = sm[j] & ~15ull, m1 = sm[16 + j] & ~15ull;      // operand A: row = lane & 15
    const bool two = z.k > 16;
    zm_f64x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const uint64_t dim = 1ull << n, stride = (uint64_t)gridDim.x * RED_THREADS;
    constexpr int U = sizeof(T) == 4 ? 8 : 4;       // loads in flight per lane
    unsigned pu0 = 0, pu1 = 0;                      // bit u: parity of the slice number's bits under the lane's masks
#pragma unroll
    for (int u = 0; u < U; ++u) {
        pu0 |= (unsigned)(__popcll((uint64_t)(u * RED_THREADS) & m0) & 1) << u;
        pu1 |= (unsigned)(__popcll((uint64_t)(u * RED_THREADS) & m1) & 1) << u;
    }
    // (a workgroup reads U consecutive slices of 256 amplitudes, the grid a contiguous window per iteration: loads that
    // are in flight together stay within a few DRAM pages instead of U addresses a whole grid stride apart)
    for (uint64_t i0 = (uint64_t)blockIdx.x * (U * RED_THREADS) + wave * 64; i0 < dim; i0 += U * stride) {
        double pr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + u * RED_THREADS;
            if (i < dim) {
                const cx<T> a = p[i + lane];
                pr[u] = (double)a.x * a.x + (double)a.y * a.y;
            } else {
                pr[u] = 0.0;
            }
        }
        // +-1.0 with the sign bit = parity of (i & mask); the slice number u only touches bits 8.. of i0 (zero there)
        const uint64_t iq = i0 | (uint64_t)(q << 4);
        const unsigned par0 = __popcll(iq & m0), par1 = __popcll(iq & m1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double a0 = __hiloint2double((int)((((par0 ^ (pu0 >> u)) & 1u) << 31) | 0x3ff00000u), 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, pr[u], acc0, 0, 0, 0);
            if (two) {
                const double a1 = __hiloint2double((int)((((par1 ^ (pu1 >> u)) & 1u) << 31) | 0x3ff00000u), 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, pr[u], acc1, 0, 0, 0);
            }
        }
    }
    // D: register r of lane (q, j) = row q + 4 r, column j
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * h + q + 4 * r;
            double v = h ? acc1[r] : acc0[r];
            if (__popcll((uint64_t)j & sm[k]) & 1) v = -v;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (j == 0) part[wave][k] = v;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < z.k) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < RED_THREADS / 64; ++w) v += part[w][threadIdx.x];
        out[((size_t)b * gridDim.x + blockIdx.x) * z.k + threadIdx.x] = v;
    }
}

// out_i = psi_i * sum_k coef[b][k] (-1)^{popc(i & zmask_k)}: the backward of the above (and sum_k c_k Z-string |psi>)
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void scale_zsigns_kernel(const cx<T>* __restrict__ psi, cx<T>* __restrict__ out,
                                                                    ZMasks z, const double* __restrict__ coef, int n) {
    const int64_t b = blockIdx.y;
    const cx<T>* p = psi + ((uint64_t)b << n);
    cx<T>* q = out + ((uint64_t)b << n);
    double c[ZM_MAX];
#pragma unroll
    for (int k = 0; k < ZM_MAX; ++k) c[k] = (k < z.k) ? coef[(size_t)b * z.k + k] : 0.0;
    const uint64_t dim = 1ull << n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (uint64_t)gridDim.x * blockDim.x) {
        double w = 0;
#pragma unroll
        for (int k = 0; k < ZM_MAX; ++k)
            if (k < z.k) w += (__popcll(i & z.m[k]) & 1) ? -c[k] : c[k];
        const cx<T> a = p[i];
        q[i] = mk<T>((T)(a.x * w), (T)(a.y * w));
    }
}

// ... and its matrix form: for the 256 consecutive amplitudes i = i0 + 16 r + c a wave handles,
//     w[r][c] = sum_k A[r][k] B[k][c],   A[r][k] = coef_k s_k(i0 + 16 r),   B[k][c] = s_k(c)   (constant per lane),
// four strings per v_mfma_f64_16x16x4_f64; register t of lane l of the result is the factor of amplitude i0 + 64 t + l,
// i.e. what the lane's t-th (coalesced) load brought.
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void scale_zsigns_mfma_kernel(const cx<T>* __restrict__ psi, cx<T>* __restrict__ out,
                                                                         ZMasks z, const double* __restrict__ coef, int n) {
    __shared__ uint64_t sm[ZM_MAX];
    __shared__ double sc[ZM_MAX];
    const int64_t b = blockIdx.y;
#pragma unroll
    for (int k = 0; k < ZM_MAX; ++k)
        if (threadIdx.x == k) sm[k] = z.m[k];
    if (threadIdx.x < ZM_MAX) sc[threadIdx.x] = (int)threadIdx.x < z.k ? coef[(size_t)b * z.k + threadIdx.x] : 0.0;
    __syncthreads();
    const cx<T>* p = psi + ((uint64_t)b << n);
    cx<T>* po = out + ((uint64_t)b << n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int nt = (z.k + 3) >> 2;
    uint64_t m[ZM_MAX / 4];
    double c[ZM_MAX / 4], bs[ZM_MAX / 4];
#pragma unroll
    for (int t = 0; t < ZM_MAX / 4; ++t) {
        const uint64_t mk = sm[4 * t + q];
        m[t] = mk & ~15ull;
        c[t] = sc[4 * t + q];
        bs[t] = (__popcll(mk & (uint64_t)j) & 1) ? -1.0 : 1.0;
    }
    const uint64_t dim = 1ull << n;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * (RED_THREADS / 64) + wave) << 8; i0 < dim;
         i0 += (uint64_t)gridDim.x * (RED_THREADS / 64) << 8) {
        cx<T> a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = p[i0 + 64 * r + lane];
        const uint64_t i = i0 | (uint64_t)(j << 4);
        zm_f64x4 w = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < ZM_MAX / 4; ++t)
            if (t < nt) w = __builtin_amdgcn_mfma_f64_16x16x4f64((__popcll(i & m[t]) & 1) ? -c[t] : c[t], bs[t], w, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) po[i0 + 64 * r + lane] = mk<T>((T)(a[r].x * w[r]), (T)(a[r].y * w[r]));
    }
}

static int fill_zmasks(ZMasks& z, const uint64_t* zmasks, int k, int n, const char* who) {
    if (!zmasks || k < 1 || k > ZM_MAX) {
        set_error("%s: %d masks per call, supported 1..%d", who, k, ZM_MAX);
        return DQ_ERR_ARG;
    }
    const uint64_t full = (1ull << n) - 1ull;
    z.k = k;
    for (int i = 0; i < ZM_MAX; ++i) z.m[i] = i < k ? zmasks[i] : 0;
    for (int i = 0; i < k; ++i)
        if (zmasks[i] & ~full) {
            set_error("%s: mask %d has bits >= n", who, i);
            return DQ_ERR_ARG;
        }
    return DQ_OK;
}

template <typename T>
static int expect_zmulti_impl(const void* psi, const uint64_t* zmasks, int k, int n, int64_t batch, double* out,
                              int nblocks, dq_stream_t stream) {
    if (!psi || !out || n < 1 || n > 40 || batch < 1 || batch > 65535 || nblocks < 1 || nblocks > 65535) {
        set_error("dq_expect_zmulti: bad argument");
        return DQ_ERR_ARG;
    }
    ZMasks z;
    int rc = fill_zmasks(z, zmasks, k, n, "dq_expect_zmulti");
    if (rc) return rc;
    static const int loop_only = [] { const char* e = getenv("DQ_ZMULTI_MFMA"); return e && atoi(e) == 0; }();
    if (n >= 8 && !loop_only)
        hipLaunchKernelGGL(expect_zmulti_mfma_kernel<T>, dim3((unsigned)nblocks, (unsigned)batch), dim3(RED_THREADS), 0,
                           as_stream(stream), static_cast<const cx<T>*>(psi), z, n, out);
    else
        hipLaunchKernelGGL(expect_zmulti_kernel<T>, dim3((unsigned)nblocks, (unsigned)batch), dim3(RED_THREADS), 0,
                           as_stream(stream), static_cast<const cx<T>*>(psi), z, n, out);
    return check_launch("dq_expect_zmulti");
}

template <typename T>
static int scale_zsigns_impl(const void* psi, void* out, const uint64_t* zmasks, int k, const double* coef, int n,
                             int64_t batch, dq_stream_t stream) {
    if (!psi || !out || !coef || n < 1 || n > 40 || batch < 1 || batch > 65535) {
        set_error("dq_scale_zsigns: bad argument");
        return DQ_ERR_ARG;
    }
    ZMasks z;
    int rc = fill_zmasks(z, zmasks, k, n, "dq_scale_zsigns");
    if (rc) return rc;
    static const int loop_only = [] { const char* e = getenv("DQ_ZMULTI_MFMA"); return e && atoi(e) == 0; }();
    if (n >= 8 && !loop_only) {
        uint64_t nb = (1ull << n) >> 10;        // four waves of 256 amplitudes per workgroup and iteration
        nb = nb < 1 ? 1 : nb > 2048 ? 2048 : nb;
        hipLaunchKernelGGL(scale_zsigns_mfma_kernel<T>, dim3((unsigned)nb, (unsigned)batch), dim3(RED_THREADS), 0,
                           as_stream(stream), static_cast<const cx<T>*>(psi), static_cast<cx<T>*>(out), z, coef, n);
        return check_launch("dq_scale_zsigns");
    }
    const unsigned nb = red_blocks(1ull << n);
    hipLaunchKernelGGL(scale_zsigns_kernel<T>, dim3(nb, (unsigned)batch), dim3(RED_THREADS), 0, as_stream(stream),
                       static_cast<const cx<T>*>(psi), static_cast<cx<T>*>(out), z, coef, n);
    return check_launch("dq_scale_zsigns");
}

template <typename T>
static int inner_impl(const void* bra, const void* ket, int64_t count, int64_t batch, double* out, void* ws,
                      dq_stream_t stream) {
    if (!bra || !ket || !out || !ws || count < 1 || batch < 1 || batch > 65535) {
        set_error("dq_inner: bad argument");
        return DQ_ERR_ARG;
    }
    const unsigned nb = red_blocks((uint64_t)count);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(inner_kernel<T>, dim3(nb, (unsigned)batch), dim3(RED_THREADS), 0, s,
                       static_cast<const cx<T>*>(bra), static_cast<const cx<T>*>(ket), (uint64_t)count,
                       static_cast<double*>(ws));
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)batch), dim3(RED_THREADS), 0, s, static_cast<const double*>(ws),
                       (int)nb, 0, 1, out);
    return check_launch("dq_inner");
}

template <typename T>
static int probs_impl(const void* psi, void* probs, int64_t count, dq_stream_t stream) {
    if (!psi || !probs || count < 1) {
        set_error("dq_probs: bad argument");
        return DQ_ERR_ARG;
    }
    uint64_t nb = ((uint64_t)count + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(probs_kernel<T>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream),
                       static_cast<const cx<T>*>(psi), static_cast<T*>(probs), (uint64_t)count);
    return check_launch("dq_probs");
}

template <typename T>
static int marginal_impl(const void* psi, int n, const int* bits, int nw, int64_t batch, double* out,
                         dq_stream_t stream) {
    if (!psi || !out || batch < 1) {
        set_error("dq_marginal: bad argument");
        return DQ_ERR_ARG;
    }
    if (nw < 1 || nw > n || n > 40) {
        set_error("dq_marginal: nw=%d unsupported (1..n)", nw);
        return DQ_ERR_UNSUPPORTED;
    }
    int rc = validate_bits(n, bits, nw, nullptr, 0);
    if (rc) return rc;
    if (batch > 65535) {
        set_error("dq_marginal: batch %lld exceeds 65535", (long long)batch);
        return DQ_ERR_UNSUPPORTED;
    }
    {
        constexpr int VEC = sizeof(T) == 4 ? 2 : 1;
        const int c = n < 12 ? n : 12;
        const int low = n < 8 - VEC ? n : 8 - VEC;          // 1 KiB of contiguous state: 7 bits complex64, 6 complex128
        uint64_t measured = 0, in_chunk = 0;
        for (int i = 0; i < nw; ++i) measured |= 1ull << bits[i];
        MargGeom g{};
        g.c = c;
        memset(g.pos, 62, sizeof g.pos);
        memset(g.lo_x, 31, sizeof g.lo_x);
        memset(g.hi_pos, 63, sizeof g.hi_pos);
        memset(g.cpos, 63, sizeof g.cpos);
        int out_of[40];                                       // bits[i] <-> outcome bit nw - 1 - i
        for (int i = 0; i < nw; ++i) out_of[bits[i]] = nw - 1 - i;
        // candidates for the chunk's bits above the contiguous part: unmeasured ones (ascending), then measured ones by
        // outcome bit (so that what a workgroup adds to the result is as contiguous as it can be)
        int cand[40], ncand = 0;
        for (int b = low; b < n; ++b)
            if (!((measured >> b) & 1ull)) cand[ncand++] = b;
        for (int o = 0; o < nw; ++o)
            for (int b = low; b < n; ++b)
                if (((measured >> b) & 1ull) && out_of[b] == o) cand[ncand++] = b;
        // the four bits a thread holds itself (chunk-local 8..11) are served first: unmeasured there = one sum per thread
        int local_of[40], next = 0;
        for (int b = 0; b < low; ++b) {
            g.pos[b] = (uint8_t)b;
            local_of[b] = b;
            in_chunk |= 1ull << b;
        }
        for (int pass = 0; pass < 2; ++pass)
            for (int x = pass == 0 ? 8 : low; x < (pass == 0 ? c : (c < 8 ? c : 8)); ++x) {
                g.pos[x] = (uint8_t)cand[next];
                local_of[cand[next]] = x;
                in_chunk |= 1ull << cand[next++];
            }
        int nhi = 0;
        for (int o = 0; o < nw; ++o)                          // histogram bit t <-> the t-th lowest outcome bit of the chunk
            for (int i = 0; i < nw; ++i) {
                if (nw - 1 - i != o) continue;
                if ((in_chunk >> bits[i]) & 1ull) {
                    const int x = local_of[bits[i]];
                    g.lo_x[g.nlo] = (uint8_t)x;
                    g.lo_out[g.nlo++] = (uint8_t)o;
                    if (x >= 8) g.qmask |= 1u << (x - 8);
                } else {
                    g.hi_pos[nhi] = (uint8_t)bits[i];
                    g.hi_out[nhi++] = (uint8_t)o;
                }
            }
        g.exclusive = nhi == n - c;
        int run = 0, nt = 0;                                  // the chunk number: unmeasured bits first
        for (int pass = 0; pass < 2; ++pass)
            for (int b = 0; b < n; ++b)
                if (!((in_chunk >> b) & 1ull) && (int)((measured >> b) & 1ull) == pass) {
                    g.cpos[nt++] = (uint8_t)b;
                    run += pass == 0;
                }
        while (run > 0 && ((1ull << (n - c - run)) * (uint64_t)batch < 2048)) --run;      // keep the chip busy
        g.run = run;
        hipLaunchKernelGGL(marginal_chunk_kernel<T>, dim3((unsigned)(1ull << (n - c - run)), (unsigned)batch), dim3(256),
                           sizeof(double) << g.nlo, as_stream(stream), static_cast<const cx<T>*>(psi), n, nw, g, out);
        return check_launch("dq_marginal");
    }
}


// ---- gate gradients of SEVERAL single-target gates from one read of both states -------------------------
// The reverse sweep of the adjoint autograd node (executor._AdjointCircuit) needs, for every trainable gate g of a
// layer,  G_g[a][b] = sum over the groups where g's controls are 1 of  gy[.. a ..] conj(x[.. b ..])  on the SAME
// pair of states.  One launch of gate_grad_kernel per gate reads both states once per gate; here a workgroup stages
// a tile of x and of gy in LDS -- the low L index bits plus up to GM_HIGH gathered bits, chosen so that every
// listed target is a tile bit and therefore every amplitude pair sits in the tile -- and accumulates all the G_g.
// Workgroups stride over the tiles and keep their partial sums in registers; one block reduction per workgroup at
// the end writes its row of partial sums, the caller adds the rows up.
constexpr int GM_HIGH = 7;
struct GradMultiDesc {
    int n, L, h, ngates;
    uint8_t high_sorted[8];  // gathered global bit positions, ascending; tile bit L + i
    uint8_t tbit[8];         // tile-local target bit per gate
    uint16_t cin[8];         // tile-local control mask per gate
    uint64_t cout[8];        // controls outside the tile per gate (global positions)
};

template <typename T> __device__ __forceinline__ unsigned gm_swz(unsigned e) {
    if constexpr (sizeof(T) == 4) return e ^ ((e >> 5) & 31u);
    else return e ^ ((e >> 4) & 15u);
}

template <typename T, int M, int GM>
__global__ __launch_bounds__(RED_THREADS) void gate_grad_multi_kernel(const cx<T>* __restrict__ x,
                                                                       const cx<T>* __restrict__ gy, GradMultiDesc d,
                                                                       uint64_t ntiles, double* __restrict__ out) {
    using V = cx<T>;
    __shared__ V sx[1 << M], sy[1 << M];
    const unsigned tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const V* px = x + ((uint64_t)b << d.n);
    const V* py = gy + ((uint64_t)b << d.n);
    T acc[GM][8];
#pragma unroll
    for (int g = 0; g < GM; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[g][i] = 0;

    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint64_t base = t << d.L;
        for (int i = 0; i < d.h; ++i) base = insert_zero(base, d.high_sorted[i]);
        // stage the tile: 16 bytes per lane (two complex64 / one complex128), consecutive lanes on consecutive
        // amplitudes of the contiguous low run; all loads of a thread are issued before the first LDS write
        constexpr int PER = sizeof(T) == 4 ? 2 : 1;                 // amplitudes per 16-byte load
        constexpr int NLD = (1 << M) / (RED_THREADS * PER);
        struct alignas(16) Q { V v[PER]; };
        Q qx[NLD], qy[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const unsigned e = (unsigned)(j * RED_THREADS + tid) * PER;
            uint64_t off = e & ((1u << d.L) - 1u);
            for (int i = 0; i < d.h; ++i) off |= (uint64_t)((e >> (d.L + i)) & 1u) << d.high_sorted[i];
            qx[j] = *reinterpret_cast<const Q*>(px + (base | off));
            qy[j] = *reinterpret_cast<const Q*>(py + (base | off));
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const unsigned e = (unsigned)(j * RED_THREADS + tid) * PER;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                sx[gm_swz<T>(e + k)] = qx[j].v[k];
                sy[gm_swz<T>(e + k)] = qy[j].v[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g < d.ngates && (base & d.cout[g]) == d.cout[g]) {   // uniform
                const unsigned tb = d.tbit[g], cm = d.cin[g];
                for (unsigned p = tid; p < (1u << (M - 1)); p += RED_THREADS) {
                    const unsigned e0 = (unsigned)insert_zero(p, (int)tb), e1 = e0 | (1u << tb);
                    if ((e0 & cm) == cm) {
                        const V x0 = sx[gm_swz<T>(e0)], x1 = sx[gm_swz<T>(e1)];
                        const V y0 = sy[gm_swz<T>(e0)], y1 = sy[gm_swz<T>(e1)];
                        // gy[a] * conj(x[b])
                        acc[g][0] += y0.x * x0.x + y0.y * x0.y;
                        acc[g][1] += y0.y * x0.x - y0.x * x0.y;
                        acc[g][2] += y0.x * x1.x + y0.y * x1.y;
                        acc[g][3] += y0.y * x1.x - y0.x * x1.y;
                        acc[g][4] += y1.x * x0.x + y1.y * x0.y;
                        acc[g][5] += y1.y * x0.x - y1.x * x0.y;
                        acc[g][6] += y1.x * x1.x + y1.y * x1.y;
                        acc[g][7] += y1.y * x1.x - y1.x * x1.y;
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        if (g < d.ngates) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                double re = (double)acc[g][i], im = (double)acc[g][i + 1];
                block_sum2(re, im);
                if (threadIdx.x == 0) {   // one partial row per workgroup: no atomics, the caller sums the rows
                    double* dst = out + ((((size_t)b * gridDim.x + blockIdx.x) * d.ngates + g) * 4 + i / 2) * 2;
                    dst[0] = re;
                    dst[1] = im;
                }
            }
        }
    }
}

template <typename T>
static int gate_grad_multi_impl(const void* x, const void* gy, int n, int ngates, const int* targets,
                                const int* ctrl_begin, const int* ctrl_bits, int64_t batch, double* out, int nblocks,
                                dq_stream_t stream) {
    constexpr bool is128 = sizeof(T) == 8;
    constexpr int M = is128 ? 10 : 11, GM = is128 ? 4 : 8, L = is128 ? 3 : 4;
    if (!x || !gy || !out || !targets || !ctrl_begin || batch < 1 || batch > 65535) {
        set_error("dq_gate_grad_multi: bad argument");
        return DQ_ERR_ARG;
    }
    if (ngates < 1 || ngates > GM) {
        set_error("dq_gate_grad_multi: %d gates per call, supported 1..%d", ngates, GM);
        return DQ_ERR_UNSUPPORTED;
    }
    if (n < M) {
        set_error("dq_gate_grad_multi: n=%d smaller than the tile (%d bits)", n, M);
        return DQ_ERR_UNSUPPORTED;
    }
    GradMultiDesc d;
    d.n = n;
    d.L = L;
    d.h = M - L;
    d.ngates = ngates;
    // gathered bits: the high targets, then the lowest free bits above L
    uint64_t high = 0;
    int nh = 0;
    for (int g = 0; g < ngates; ++g) {
        const int t = targets[g];
        if (t < 0 || t >= n) {
            set_error("dq_gate_grad_multi: target %d out of range", t);
            return DQ_ERR_ARG;
        }
        if (t >= L && !((high >> t) & 1ull)) {
            high |= 1ull << t;
            ++nh;
        }
    }
    if (nh > d.h) {
        set_error("dq_gate_grad_multi: %d distinct targets above bit %d, the tile gathers %d", nh, L, d.h);
        return DQ_ERR_UNSUPPORTED;
    }
    for (int p = L; nh < d.h; ++p)
        if (!((high >> p) & 1ull)) {
            high |= 1ull << p;
            ++nh;
        }
    int local[64];
    for (int p = 0; p < L; ++p) local[p] = p;
    {
        int i = 0;
        for (int p = L; p < n; ++p)
            if ((high >> p) & 1ull) {
                d.high_sorted[i] = (uint8_t)p;
                local[p] = L + i;
                ++i;
            } else {
                local[p] = -1;
            }
    }
    for (int g = 0; g < ngates; ++g) {
        d.tbit[g] = (uint8_t)local[targets[g]];
        d.cin[g] = 0;
        d.cout[g] = 0;
        for (int c = ctrl_begin[g]; c < ctrl_begin[g + 1]; ++c) {
            const int q = ctrl_bits[c];
            if (q < 0 || q >= n || q == targets[g]) {
                set_error("dq_gate_grad_multi: bad control %d of gate %d", q, g);
                return DQ_ERR_ARG;
            }
            if (local[q] >= 0) d.cin[g] |= (uint16_t)(1u << local[q]);
            else d.cout[g] |= 1ull << q;
        }
    }
    const uint64_t ntiles = 1ull << (n - M);
    if (nblocks < 1 || (uint64_t)nblocks > ntiles) {
        set_error("dq_gate_grad_multi: nblocks=%d outside [1, %llu tiles]", nblocks, (unsigned long long)ntiles);
        return DQ_ERR_ARG;
    }
    dim3 grid((unsigned)nblocks, (unsigned)batch);       // workgroups stride over the tiles
    hipLaunchKernelGGL((gate_grad_multi_kernel<T, M, GM>), grid, dim3(RED_THREADS), 0, as_stream(stream),
                       static_cast<const cx<T>*>(x), static_cast<const cx<T>*>(gy), d, ntiles, out);
    return check_launch("dq_gate_grad_multi");
}

template <typename T>
static int gate_grad_impl(const void* x, const void* gy, int n, const int* targets, int k, const int* controls, int nc,
                          int64_t batch, double* gU, dq_stream_t stream) {
    if (!x || !gy || !gU || batch < 1 || batch > 65535) {
        set_error("dq_gate_grad: bad argument");
        return DQ_ERR_ARG;
    }
    if (k < 1 || k > 2) {
        set_error("dq_gate_grad: k=%d unsupported (1..2)", k);
        return DQ_ERR_UNSUPPORTED;
    }
    int rc = validate_bits(n, targets, k, controls, nc);
    if (rc) return rc;
    if (k + nc > 16) {
        set_error("dq_gate_grad: k+nc > 16");
        return DQ_ERR_UNSUPPORTED;
    }
    GradGeom g;
    g.n = n;
    g.cmask = 0;
    int all[16];
    for (int i = 0; i < k; ++i) {
        g.tpos[i] = targets[i];
        all[i] = targets[i];
    }
    for (int i = 0; i < nc; ++i) {
        g.cmask |= 1ull << controls[i];
        all[k + i] = controls[i];
    }
    sort_bits(all, k + nc, g.sorted);
    const uint64_t groups = 1ull << (n - k - nc);
    uint64_t nb = (groups + RED_THREADS - 1) / RED_THREADS;
    if (nb > 512) nb = 512;
    dim3 grid((unsigned)nb, (unsigned)batch);
    hipStream_t s = as_stream(stream);
    if (k == 1)
        hipLaunchKernelGGL((gate_grad_kernel<T, 1>), grid, dim3(RED_THREADS), 0, s, static_cast<const cx<T>*>(x),
                           static_cast<const cx<T>*>(gy), g, groups, gU);
    else
        hipLaunchKernelGGL((gate_grad_kernel<T, 2>), grid, dim3(RED_THREADS), 0, s, static_cast<const cx<T>*>(x),
                           static_cast<const cx<T>*>(gy), g, groups, gU);
    return check_launch("dq_gate_grad");
}

}  // namespace dq

extern "C" int64_t dq_reduce_ws_bytes(int64_t batch) {
    if (batch < 1) batch = 1;
    return batch * (int64_t)dq::RED_BLOCKS * 2 * (int64_t)sizeof(double);
}

#define DQ_DEFINE(SUFFIX, T)                                                                                          \
    extern "C" int dq_expect_pauli_##SUFFIX(const void* psi, uint64_t xmask, uint64_t zmask, int n, int64_t batch,    \
                                            double* out, void* ws, dq_stream_t stream) {                              \
        return dq::expect_impl<T>(psi, xmask, zmask, n, batch, out, ws, stream);                                      \
    }                                                                                                                 \
    extern "C" int dq_inner_##SUFFIX(const void* bra, const void* ket, int64_t count, int64_t batch, double* out,     \
                                     void* ws, dq_stream_t stream) {                                                  \
        return dq::inner_impl<T>(bra, ket, count, batch, out, ws, stream);                                            \
    }                                                                                                                 \
    extern "C" int dq_probs_##SUFFIX(const void* psi, void* probs, int64_t count, dq_stream_t stream) {               \
        return dq::probs_impl<T>(psi, probs, count, stream);                                                          \
    }                                                                                                                 \
    extern "C" int dq_marginal_##SUFFIX(const void* psi, int n, const int* bits, int nw, int64_t batch, double* out,  \
                                        dq_stream_t stream) {                                                         \
        return dq::marginal_impl<T>(psi, n, bits, nw, batch, out, stream);                                            \
    }                                                                                                                 \
    extern "C" int dq_gate_grad_##SUFFIX(const void* x, const void* gy, int n, const int* targets, int k,             \
                                         const int* controls, int nc, int64_t batch, double* gU,                      \
                                         dq_stream_t stream) {                                                        \
        return dq::gate_grad_impl<T>(x, gy, n, targets, k, controls, nc, batch, gU, stream);                          \
    }

#define DQ_DEFINE_Z(SUFFIX, T)                                                                                        \
    extern "C" int dq_expect_zmulti_##SUFFIX(const void* psi, const uint64_t* zmasks, int k, int n, int64_t batch,    \
                                             double* out, int nblocks, dq_stream_t stream) {                          \
        return dq::expect_zmulti_impl<T>(psi, zmasks, k, n, batch, out, nblocks, stream);                             \
    }                                                                                                                 \
    extern "C" int dq_scale_zsigns_##SUFFIX(const void* psi, void* out, const uint64_t* zmasks, int k,                \
                                            const double* coef, int n, int64_t batch, dq_stream_t stream) {          \
        return dq::scale_zsigns_impl<T>(psi, out, zmasks, k, coef, n, batch, stream);                                 \
    }
DQ_DEFINE_Z(c64, float)
DQ_DEFINE_Z(c128, double)

extern "C" int dq_gate_grad_multi_c64(const void* x, const void* gy, int n, int ngates, const int* targets,
                                      const int* ctrl_begin, const int* ctrl_bits, int64_t batch, double* out,
                                      int nblocks, dq_stream_t stream) {
    return dq::gate_grad_multi_impl<float>(x, gy, n, ngates, targets, ctrl_begin, ctrl_bits, batch, out, nblocks, stream);
}
extern "C" int dq_gate_grad_multi_c128(const void* x, const void* gy, int n, int ngates, const int* targets,
                                       const int* ctrl_begin, const int* ctrl_bits, int64_t batch, double* out,
                                       int nblocks, dq_stream_t stream) {
    return dq::gate_grad_multi_impl<double>(x, gy, n, ngates, targets, ctrl_begin, ctrl_bits, batch, out, nblocks, stream);
}

DQ_DEFINE(c64, float)
DQ_DEFINE(c128, double)
