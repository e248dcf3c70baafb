// Shard exchange helpers for the index-bit-partitioned state: gather the sub-cube of a shard that
// has to travel to the partner rank into a contiguous send buffer, and combine/scatter what came
// back.  Replaces the arange + boolean-mask gathers and the axpby on the received half in the
// reference (distributed.py:70, 109-126, 150-157).  HBM-bound streams; the exchange itself is done
// by the host through torch.distributed (RCCL over xGMI).
#include "dq_common.hpp"

namespace dq {

struct MaskGeom {
    BitList sorted;  // positions of the mask bits, ascending
    uint64_t value;
    int nl;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const cx<T>* __restrict__ amps, cx<T>* __restrict__ packed,
                                                    MaskGeom g, uint64_t count) {
    const int64_t b = blockIdx.y;
    const cx<T>* src = amps + ((uint64_t)b << g.nl);
    cx<T>* dst = packed + (uint64_t)b * count;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < count;
         c += (uint64_t)gridDim.x * blockDim.x)
        dst[c] = src[insert_zeros(c, g.sorted) | g.value];
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_axpby_kernel(cx<T>* amps, const cx<T>* x,
                                                            const cx<T>* y, const cx<T>* __restrict__ coef,
                                                            int64_t coef_bstride, MaskGeom g, uint64_t count) {
    const int64_t b = blockIdx.y;
    cx<T>* dst = amps + ((uint64_t)b << g.nl);
    const cx<T>* px = x + (uint64_t)b * count;
    if (y == nullptr) {
        for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < count;
             c += (uint64_t)gridDim.x * blockDim.x)
            dst[insert_zeros(c, g.sorted) | g.value] = px[c];
        return;
    }
    const cx<T>* py = y + (uint64_t)b * count;
    const cx<T> ca = coef[b * coef_bstride], cb = coef[b * coef_bstride + 1];
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < count;
         c += (uint64_t)gridDim.x * blockDim.x) {
        const cx<T> r = cfma(cb, py[c], cmul(ca, px[c]));
        dst[insert_zeros(c, g.sorted) | g.value] = r;
    }
}

struct PermGeom {
    int nl;
    int src_of_dst[40];  // bit p of the destination index comes from bit src_of_dst[p] of the source index
};

// out[b, i] = in[b, sigma(i)], sigma(i) = sum_p bit_p(i) << src_of_dst[p]: a pure re-labelling of the
// local qubits (one read + one write of the shard).  Used by the all-to-all qubit remap to move the bits
// that are about to become rank bits to the top of the local index, so that every peer's chunk is one
// contiguous slab, and to restore the canonical qubit order at the end.  Writes are fully coalesced;
// reads are coalesced as long as the low bits map to low bits.
template <typename T>
__global__ __launch_bounds__(256) void permute_bits_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                            PermGeom g) {
    const int64_t b = blockIdx.y;
    const cx<T>* src = in + ((uint64_t)b << g.nl);
    cx<T>* dst = out + ((uint64_t)b << g.nl);
    const uint64_t count = 1ull << g.nl;
    // identity prefix: the lowest bits that map to themselves can be copied as contiguous runs
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t sidx = 0;
        for (int p = 0; p < g.nl; ++p) sidx |= ((i >> p) & 1ull) << g.src_of_dst[p];
        dst[i] = src[sidx];
    }
}

// The same for states of >= 2^12 amplitudes without the per-element loop over the bits (it cost ~120 VALU instructions
// per 8 bytes: 3.4 TB/s for the identity): the index of an element is (block << (LV + 10)) | (piece << (LV + 8)) | (thread << LV) | e, so the
// source index is the OR of a per-thread part (once), a per-block part (uniform: scalar arithmetic) and, when index bit 0
// stays where it is (V = 2, complex64), nothing for e -- one 16-byte load and one 16-byte store per thread and block.
template <typename T, int V>
__global__ __launch_bounds__(256) void permute_bits_tiled_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                                  PermGeom g, int nt) {
    constexpr int LV = V == 2 ? 1 : 0;
    const int64_t b = blockIdx.y;
    const cx<T>* src = in + ((uint64_t)b << g.nl);
    cx<T>* dst = out + ((uint64_t)b << g.nl);
    const unsigned t = threadIdx.x;
    uint64_t st = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) st |= (uint64_t)((t >> p) & 1u) << g.src_of_dst[LV + p];
    // four 16-byte (8-byte) pieces per thread and block: index bits LV + 8, LV + 9 pick the piece (uniform offsets)
    uint64_t sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        sj[j] = ((uint64_t)(j & 1) << g.src_of_dst[LV + 8]) | ((uint64_t)(j >> 1) << g.src_of_dst[LV + 9]);
    const int nb_bits = g.nl - LV - 10;
    const uint64_t nblk = 1ull << nb_bits;
    typedef T vec_t __attribute__((ext_vector_type(2 * V)));
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        uint64_t sb = 0;
        for (int p = 0; p < nb_bits; ++p) sb |= ((blk >> p) & 1ull) << g.src_of_dst[LV + 10 + p];
        const uint64_t i = (blk << (LV + 10)) | ((uint64_t)t << LV);
        vec_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const vec_t* ps = reinterpret_cast<const vec_t*>(src + (sb | sj[j] | st));
            v[j] = nt ? __builtin_nontemporal_load(ps) : *ps;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vec_t* pd = reinterpret_cast<vec_t*>(dst + (i | ((uint64_t)j << (LV + 8))));
            if (nt) __builtin_nontemporal_store(v[j], pd);
            else *pd = v[j];
        }
    }
}

template <typename T>
static int permute_impl(const void* in, void* out, int nl, const int* src_of_dst, int64_t batch, dq_stream_t stream) {
    if (!in || !out || in == out || !src_of_dst || batch < 1 || batch > 65535 || nl < 0 || nl > 40) {
        set_error("dq_permute_bits: bad argument (in == out is not allowed)");
        return DQ_ERR_ARG;
    }
    PermGeom g;
    g.nl = nl;
    uint64_t seen = 0;
    for (int p = 0; p < nl; ++p) {
        const int sp = src_of_dst[p];
        if (sp < 0 || sp >= nl || ((seen >> sp) & 1ull)) {
            set_error("dq_permute_bits: src_of_dst is not a permutation of [0, %d)", nl);
            return DQ_ERR_ARG;
        }
        seen |= 1ull << sp;
        g.src_of_dst[p] = sp;
    }
    const uint64_t count = 1ull << nl;
    if (nl >= 12) {
        const bool pair = sizeof(T) == 4 && g.src_of_dst[0] == 0;       // complex64: two neighbours per 16-byte access
        const int lv = pair ? 1 : 0;
        uint64_t nblk = 1ull << (nl - lv - 10);
        if (nblk > 256ull * 16ull) nblk = 256ull * 16ull;                 // 16 workgroups per CU, looping
        const int nt = ((uint64_t)batch << nl) * sizeof(cx<T>) >= (1ull << 30);      // streaming accesses on big shards
        const dim3 grid((unsigned)nblk, (unsigned)batch);
        if (pair)
            hipLaunchKernelGGL((permute_bits_tiled_kernel<T, 2>), grid, dim3(256), 0, as_stream(stream),
                               static_cast<const cx<T>*>(in), static_cast<cx<T>*>(out), g, nt);
        else
            hipLaunchKernelGGL((permute_bits_tiled_kernel<T, 1>), grid, dim3(256), 0, as_stream(stream),
                               static_cast<const cx<T>*>(in), static_cast<cx<T>*>(out), g, nt);
        return check_launch("dq_permute_bits");
    }
    uint64_t nb = (count + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(permute_bits_kernel<T>, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, as_stream(stream),
                       static_cast<const cx<T>*>(in), static_cast<cx<T>*>(out), g);
    return check_launch("dq_permute_bits");
}

static int make_geom(int nl, uint64_t mask, uint64_t value, MaskGeom& g, const char* who) {
    if (nl < 0 || nl > 40) {
        set_error("%s: nl=%d out of range", who, nl);
        return DQ_ERR_ARG;
    }
    const uint64_t full = (1ull << nl) - 1ull;
    if ((mask & ~full) || (value & ~mask)) {
        set_error("%s: mask/value inconsistent with nl=%d", who, nl);
        return DQ_ERR_ARG;
    }
    g.nl = nl;
    g.value = value;
    g.sorted.n = 0;
    for (int p = 0; p < nl; ++p)
        if ((mask >> p) & 1ull) {
            if (g.sorted.n >= 16) {
                set_error("%s: more than 16 mask bits", who);
                return DQ_ERR_UNSUPPORTED;
            }
            g.sorted.pos[g.sorted.n++] = p;
        }
    return DQ_OK;
}

template <typename T>
static int pack_impl(const void* amps, void* packed, int nl, uint64_t mask, uint64_t value, int64_t batch,
                     dq_stream_t stream) {
    if (!amps || !packed || batch < 1 || batch > 65535) {
        set_error("dq_pack: bad argument");
        return DQ_ERR_ARG;
    }
    MaskGeom g;
    int rc = make_geom(nl, mask, value, g, "dq_pack");
    if (rc) return rc;
    const uint64_t count = 1ull << (nl - g.sorted.n);
    uint64_t nb = (count + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(pack_kernel<T>, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, as_stream(stream),
                       static_cast<const cx<T>*>(amps), static_cast<cx<T>*>(packed), g, count);
    return check_launch("dq_pack");
}

template <typename T>
static int unpack_impl(void* amps, const void* x, const void* y, const void* coef, int64_t coef_bstride, int nl,
                       uint64_t mask, uint64_t value, int64_t batch, dq_stream_t stream) {
    if (!amps || !x || (y && !coef) || batch < 1 || batch > 65535) {
        set_error("dq_unpack_axpby: bad argument");
        return DQ_ERR_ARG;
    }
    MaskGeom g;
    int rc = make_geom(nl, mask, value, g, "dq_unpack_axpby");
    if (rc) return rc;
    const uint64_t count = 1ull << (nl - g.sorted.n);
    uint64_t nb = (count + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(unpack_axpby_kernel<T>, dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, as_stream(stream),
                       static_cast<cx<T>*>(amps), static_cast<const cx<T>*>(x), static_cast<const cx<T>*>(y),
                       static_cast<const cx<T>*>(coef), coef_bstride, g, count);
    return check_launch("dq_unpack_axpby");
}

}  // namespace dq

extern "C" int dq_pack_c64(const void* amps, void* packed, int nl, uint64_t mask, uint64_t value, int64_t batch,
                           dq_stream_t stream) {
    return dq::pack_impl<float>(amps, packed, nl, mask, value, batch, stream);
}
extern "C" int dq_pack_c128(const void* amps, void* packed, int nl, uint64_t mask, uint64_t value, int64_t batch,
                            dq_stream_t stream) {
    return dq::pack_impl<double>(amps, packed, nl, mask, value, batch, stream);
}
extern "C" int dq_unpack_axpby_c64(void* amps, const void* x, const void* y, const void* coef, int64_t coef_batch_stride,
                                   int nl, uint64_t mask, uint64_t value, int64_t batch, dq_stream_t stream) {
    return dq::unpack_impl<float>(amps, x, y, coef, coef_batch_stride, nl, mask, value, batch, stream);
}
extern "C" int dq_unpack_axpby_c128(void* amps, const void* x, const void* y, const void* coef,
                                    int64_t coef_batch_stride, int nl, uint64_t mask, uint64_t value, int64_t batch,
                                    dq_stream_t stream) {
    return dq::unpack_impl<double>(amps, x, y, coef, coef_batch_stride, nl, mask, value, batch, stream);
}
extern "C" int dq_permute_bits_c64(const void* in, void* out, int nl, const int* src_of_dst, int64_t batch,
                                   dq_stream_t stream) {
    return dq::permute_impl<float>(in, out, nl, src_of_dst, batch, stream);
}
extern "C" int dq_permute_bits_c128(const void* in, void* out, int nl, const int* src_of_dst, int64_t batch,
                                    dq_stream_t stream) {
    return dq::permute_impl<double>(in, out, nl, src_of_dst, batch, stream);
}
