// Single-gate application kernels (unfused path): psi' = (U on targets | controls = 1) psi.
// Replaces qmath.evolve_state (qmath.py:485-506) and Gate.op_state_control (operation.py:203-219)
// of the reference.  Used by the autograd path (one custom Function per gate), by states too small
// for the fused tile kernel, and for dense gates on more than two wires.
//
// HBM traffic per launch: 2 * 2^(n-nc) * sizeof(amp) * batch (each touched amplitude read once and
// written once); nothing is staged or copied around the controlled slice.
#include "dq_common.hpp"
#include <atomic>

namespace dq {

struct GateGeom {
    BitList sorted;      // targets U controls, ascending: zero-insertion positions
    int tpos[10];        // targets in matrix order (tpos[0] = matrix MSB)
    uint64_t cmask;      // OR of control bits
    int n, k, nc;
};

// One thread per amplitude group; the 2^K amplitudes of the group live in registers.
template <typename T, int K, bool WIDE>
__global__ __launch_bounds__(256) void apply_small_kernel(const cx<T>* in, cx<T>* out,
                                                          const cx<T>* __restrict__ mats, int64_t mat_bstride,
                                                          GateGeom g, uint64_t groups) {
    constexpr int D = 1 << K;
    __shared__ cx<T> sm[D * D];
    const int64_t b = blockIdx.y;
    const cx<T>* mp = mats + b * mat_bstride;
    for (int i = threadIdx.x; i < D * D; i += blockDim.x) sm[i] = mp[i];
    __syncthreads();

    uint64_t offs[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        uint64_t o = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) o |= (uint64_t)((j >> (K - 1 - i)) & 1) << g.tpos[i];
        offs[j] = o;
    }
    const uint64_t state_off = (uint64_t)b << g.n;
    const cx<T>* pin = in + state_off;
    cx<T>* pout = out + state_off;
    if constexpr (WIDE) {
        // complex64 with index bit 0 neither target nor control: a thread takes the two groups that differ in bit 0 --
        // neighbours in memory -- so every access moves 16 bytes per lane (a wave instruction 1 KiB) instead of 8.
        // `g.sorted` then lists bit 0 as well (the host adds it) and `groups` counts pairs.
        for (uint64_t gi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups;
             gi += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t base = insert_zeros(gi, g.sorted) | g.cmask;
            float4 a[D];
#pragma unroll
            for (int j = 0; j < D; ++j) a[j] = *reinterpret_cast<const float4*>(pin + (base | offs[j]));
#pragma unroll
            for (int i = 0; i < D; ++i) {
                cx<T> lo = cmul(sm[i * D], mk<T>(a[0].x, a[0].y)), hi = cmul(sm[i * D], mk<T>(a[0].z, a[0].w));
#pragma unroll
                for (int j = 1; j < D; ++j) {
                    lo = cfma(sm[i * D + j], mk<T>(a[j].x, a[j].y), lo);
                    hi = cfma(sm[i * D + j], mk<T>(a[j].z, a[j].w), hi);
                }
                *reinterpret_cast<float4*>(pout + (base | offs[i])) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        return;
    }
    for (uint64_t gi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups;
         gi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = insert_zeros(gi, g.sorted) | g.cmask;
        cx<T> a[D];
#pragma unroll
        for (int j = 0; j < D; ++j) a[j] = pin[base | offs[j]];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            cx<T> acc = cmul(sm[i * D], a[0]);
#pragma unroll
            for (int j = 1; j < D; ++j) acc = cfma(sm[i * D + j], a[j], acc);
            pout[base | offs[i]] = acc;
        }
    }
}

// When in != out the amplitudes whose controls are not all 1 must still be carried over.
template <typename T>
__global__ __launch_bounds__(256) void copy_uncontrolled_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                                uint64_t cmask, uint64_t total) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        if ((i & cmask) != cmask) out[i] = in[i];
    }
}

// One thread per OUTPUT amplitude, matrix streamed from global memory (L2 resident). k <= 10,
// out-of-place only.  Only reached by UAnyGate/LatentGate-style dense blocks on > 4 wires.
template <typename T>
__global__ __launch_bounds__(256) void apply_big_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                        const cx<T>* __restrict__ mats, int64_t mat_bstride,
                                                        GateGeom g) {
    const int64_t b = blockIdx.y;
    const cx<T>* mp = mats + b * mat_bstride;
    const uint64_t dim = 1ull << g.n;
    const int D = 1 << g.k;
    uint64_t tmask = 0;
    for (int i = 0; i < g.k; ++i) tmask |= 1ull << g.tpos[i];
    const cx<T>* pin = in + ((uint64_t)b << g.n);
    cx<T>* pout = out + ((uint64_t)b << g.n);
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dim;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        if ((idx & g.cmask) != g.cmask) {
            pout[idx] = pin[idx];
            continue;
        }
        int row = 0;
        for (int i = 0; i < g.k; ++i) row |= (int)((idx >> g.tpos[i]) & 1ull) << (g.k - 1 - i);
        const uint64_t base = idx & ~tmask;
        cx<T> acc = mk<T>(0, 0);
        for (int j = 0; j < D; ++j) {
            uint64_t o = 0;
            for (int i = 0; i < g.k; ++i) o |= (uint64_t)((j >> (g.k - 1 - i)) & 1) << g.tpos[i];
            acc = cfma(mp[(int64_t)row * D + j], pin[base | o], acc);
        }
        pout[idx] = acc;
    }
}

// csrc/dq_dense.hip: the same product on the matrix cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64)
template <typename T>
int apply_dense_mfma(const cx<T>* in, cx<T>* out, const cx<T>* mats, int64_t mat_bstride, int n, const int* targets, int k,
                     const int* controls, int nc, const BitList& sorted, uint64_t cmask, int64_t batch, hipStream_t s);

static std::atomic<int> g_dense_path{1};   // 1 = MFMA (default), 0 = one thread per output amplitude on the VALU (A/B, dq_set_dense_path)

template <typename T>
static int apply_gate_impl(const void* in, void* out, const void* mats, int64_t mat_bstride, int n,
                           const int* targets, int k, const int* controls, int nc, int64_t batch,
                           dq_stream_t stream) {
    if (!in || !out || !mats) {
        set_error("dq_apply_gate: null pointer");
        return DQ_ERR_ARG;
    }
    if (batch < 1 || batch > 65535) {
        set_error("dq_apply_gate: batch %lld out of range [1, 65535]", (long long)batch);
        return DQ_ERR_ARG;
    }
    if (k < 1 || k > 10) {
        set_error("dq_apply_gate: k=%d unsupported (1..10)", k);
        return DQ_ERR_UNSUPPORTED;
    }
    int rc = validate_bits(n, targets, k, controls, nc);
    if (rc) return rc;
    if (k + nc > 16) {
        set_error("dq_apply_gate: k+nc=%d > 16", k + nc);
        return DQ_ERR_UNSUPPORTED;
    }
    GateGeom g;
    g.n = n;
    g.k = k;
    g.nc = nc;
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
    const int na = k + nc;
    for (int i = 1; i < na; ++i) {  // insertion sort ascending
        int v = all[i], j = i - 1;
        while (j >= 0 && all[j] > v) {
            all[j + 1] = all[j];
            --j;
        }
        all[j + 1] = v;
    }
    g.sorted.n = na;
    for (int i = 0; i < na; ++i) g.sorted.pos[i] = all[i];

    hipStream_t s = as_stream(stream);
    using V = cx<T>;
    const V* pin = static_cast<const V*>(in);
    V* pout = static_cast<V*>(out);
    const V* pm = static_cast<const V*>(mats);
    const uint64_t total = (1ull << n) * (uint64_t)batch;

    if (k <= 4) {
        if (pin != pout && nc > 0) {
            uint64_t blocks = (total + 255) / 256;
            if (blocks > 65536) blocks = 65536;
            hipLaunchKernelGGL(copy_uncontrolled_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, pin, pout,
                               g.cmask, total);
        }
        uint64_t groups = 1ull << (n - na);
        // complex64, bit 0 free, k <= 3: pairs of neighbouring groups per thread (16-byte accesses)
        bool wide = false;
        if constexpr (sizeof(T) == 4) {
            if (k <= 3 && na < 16 && n - na >= 1 && g.sorted.pos[0] > 0) {
                wide = true;
                for (int i = na; i > 0; --i) g.sorted.pos[i] = g.sorted.pos[i - 1];
                g.sorted.pos[0] = 0;
                g.sorted.n = na + 1;
                groups >>= 1;
            }
        }
        uint64_t blocks = (groups + 255) / 256;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        dim3 grid((unsigned)blocks, (unsigned)batch);
#define DQ_SMALL(K, W) hipLaunchKernelGGL((apply_small_kernel<T, K, W>), grid, dim3(256), 0, s, pin, pout, pm, mat_bstride, g, groups)
        if constexpr (sizeof(T) == 4) {
            if (wide) {
                switch (k) {
                    case 1: DQ_SMALL(1, true); break;
                    case 2: DQ_SMALL(2, true); break;
                    default: DQ_SMALL(3, true); break;
                }
            }
        }
        if (!wide) {
            switch (k) {
                case 1: DQ_SMALL(1, false); break;
                case 2: DQ_SMALL(2, false); break;
                case 3: DQ_SMALL(3, false); break;
                default: DQ_SMALL(4, false); break;
            }
        }
#undef DQ_SMALL
    } else {
        if (pin == pout) {
            set_error("dq_apply_gate: k=%d > 4 requires out != in", k);
            return DQ_ERR_ARG;
        }
        if (g_dense_path.load(std::memory_order_relaxed) == 1) {
            // amplitudes whose controls are not all 1 are carried over; the MFMA kernel writes the controlled columns
            if (nc > 0) {
                uint64_t blocks = (total + 255) / 256;
                if (blocks > 65536) blocks = 65536;
                hipLaunchKernelGGL(copy_uncontrolled_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, pin, pout,
                                   g.cmask, total);
            }
            apply_dense_mfma<T>(pin, pout, pm, mat_bstride, n, targets, k, controls, nc, g.sorted, g.cmask, batch, s);
        } else {
            uint64_t blocks = ((1ull << n) + 255) / 256;
            if (blocks > (1u << 20)) blocks = 1u << 20;
            dim3 grid((unsigned)blocks, (unsigned)batch);
            hipLaunchKernelGGL(apply_big_kernel<T>, grid, dim3(256), 0, s, pin, pout, pm, mat_bstride, g);
        }
    }
    return check_launch("dq_apply_gate");
}

}  // namespace dq

extern "C" int dq_set_dense_path(int mfma) {
    if (mfma != 0 && mfma != 1) {
        dq::set_error("dq_set_dense_path: 0 (VALU) or 1 (MFMA)");
        return DQ_ERR_ARG;
    }
    dq::g_dense_path.store(mfma, std::memory_order_relaxed);
    return DQ_OK;
}

extern "C" int dq_apply_gate_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                 const int* targets, int k, const int* controls, int nc, int64_t batch,
                                 dq_stream_t stream) {
    return dq::apply_gate_impl<float>(in, out, mats, mat_batch_stride, n, targets, k, controls, nc, batch, stream);
}
extern "C" int dq_apply_gate_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                  const int* targets, int k, const int* controls, int nc, int64_t batch,
                                  dq_stream_t stream) {
    return dq::apply_gate_impl<double>(in, out, mats, mat_batch_stride, n, targets, k, controls, nc, batch, stream);
}
