// Shard exchange helpers for the index-bit-partitioned state: gather the sub-cube of a shard that
// has to travel to the partner rank into a contiguous send buffer, and combine/scatter what came
// back.  Replaces the arange + boolean-mask gathers and the axpby on the received half in the
// reference (distributed.py:70, 109-126, 150-157).  HBM-bound streams; the exchange itself is done
// by the host through torch.distributed (RCCL over xGMI).
#include <cstring>
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

// A permutation that brings high source bits down to the low destination bits makes the kernel above gather 8 bytes at
// a time (3.3 - 3.5 TB/s).  Here a workgroup moves a TILE of 2^10 elements through LDS: the tile's destination bits are
// the low five, the destination bits fed by the low five SOURCE bits, and fill; it is read in source order (runs of 32
// elements), stored in LDS at its destination-tile coordinate (xor-swizzled by the higher coordinate bits), and written
// in destination order (runs of >= 32 elements).  Both sides of HBM see 256-byte (complex128: 512-byte) runs.
struct PermTileGeom {
    int nrest;                  // index bits outside the tile
    uint8_t ts[10];             // bit k of the source-order tile coordinate  <-> source index bit ts[k] (ascending)
    uint8_t pi[10];             // ... and the bit of the destination-order coordinate it becomes
    uint8_t td[10];             // bit k of the destination-order coordinate <-> destination index bit td[k] (ascending)
    uint8_t rd[30], rs[30];     // bit t of the tile number <-> destination bit rd[t] / source bit rs[t]   (pad: 63)
};

__device__ __forceinline__ unsigned perm_swz(unsigned e) { return e ^ ((e >> 4) & 15u) ^ ((e >> 8) & 3u); }

template <typename T>
__global__ __launch_bounds__(256) void permute_bits_lds_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, int nl,
                                                                PermTileGeom g) {
    __shared__ cx<T> tile[1024];
    const int64_t b = blockIdx.y;
    const cx<T>* src = in + ((uint64_t)b << nl);
    cx<T>* dst = out + ((uint64_t)b << nl);
    const unsigned t = threadIdx.x;
    uint64_t st = 0, dt = 0;
    unsigned et = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        st |= (uint64_t)((t >> k) & 1u) << g.ts[k];
        et |= ((t >> k) & 1u) << g.pi[k];
        dt |= (uint64_t)((t >> k) & 1u) << g.td[k];
    }
    uint64_t sj[4], dj[4];
    unsigned ej[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sj[j] = ((uint64_t)(j & 1) << g.ts[8]) | ((uint64_t)(j >> 1) << g.ts[9]);
        ej[j] = ((unsigned)(j & 1) << g.pi[8]) | ((unsigned)(j >> 1) << g.pi[9]);
        dj[j] = ((uint64_t)(j & 1) << g.td[8]) | ((uint64_t)(j >> 1) << g.td[9]);
    }
    const uint64_t ntile = 1ull << g.nrest;
    for (uint64_t blk = blockIdx.x; blk < ntile; blk += gridDim.x) {
        uint64_t sb = 0, db = 0;
#pragma unroll
        for (int p = 0; p < 30; ++p) {
            sb |= ((blk >> p) & 1ull) << g.rs[p];
            db |= ((blk >> p) & 1ull) << g.rd[p];
        }
        cx<T> v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = src[sb | sj[j] | st];
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[perm_swz(et | ej[j])] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[perm_swz(t | (j << 8))];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[db | dj[j] | dt] = v[j];
        __syncthreads();
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
    static const int lds_off = [] { const char* e = getenv("DQ_PERMUTE_LDS"); return e && atoi(e) == 0; }();
    bool low_in_place = true;       // do the low five destination bits come from the low five source bits?
    for (int p = 0; p < 5 && p < nl; ++p) low_in_place = low_in_place && g.src_of_dst[p] < 5;
    if (nl >= 12 && !low_in_place && !lds_off) {
        PermTileGeom tg{};
        memset(tg.rd, 63, sizeof tg.rd);
        memset(tg.rs, 63, sizeof tg.rs);
        int inv[40];
        for (int p = 0; p < nl; ++p) inv[g.src_of_dst[p]] = p;
        uint64_t in_tile = 0;       // destination bits of the tile
        for (int p = 0; p < 5; ++p) in_tile |= (1ull << p) | (1ull << inv[p]);
        for (int p = 5; p < nl && __builtin_popcountll(in_tile) < 10; ++p) in_tile |= 1ull << p;
        int idx_of_dst[40], nd = 0, nr = 0;
        uint64_t src_tile = 0;
        for (int p = 0; p < nl; ++p) {
            if ((in_tile >> p) & 1ull) {
                idx_of_dst[p] = nd;
                tg.td[nd++] = (uint8_t)p;
                src_tile |= 1ull << g.src_of_dst[p];
            } else {
                tg.rd[nr] = (uint8_t)p;
                tg.rs[nr++] = (uint8_t)g.src_of_dst[p];
            }
        }
        tg.nrest = nr;
        for (int q = 0, k = 0; q < nl; ++q)
            if ((src_tile >> q) & 1ull) {
                tg.ts[k] = (uint8_t)q;
                tg.pi[k++] = (uint8_t)idx_of_dst[inv[q]];
            }
        uint64_t nblk = 1ull << nr;
        if (nblk > 256ull * 16ull) nblk = 256ull * 16ull;
        hipLaunchKernelGGL(permute_bits_lds_kernel<T>, dim3((unsigned)nblk, (unsigned)batch), dim3(256), 0,
                           as_stream(stream), static_cast<const cx<T>*>(in), static_cast<cx<T>*>(out), nl, tg);
        return check_launch("dq_permute_bits");
    }
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

// ---- two states side by side along a NEW index bit 0: the (psi, lambda) pair of a fused reverse sweep -------------------
// out[2 i] = a[i], out[2 i + 1] = b[i] (interleave) and back (which = 0 / 1 picks the half).  16-byte pieces in, 32-byte
// pieces out per thread: every access a full, coalesced line -- torch.stack([a, b], dim=-1) of two 2-GiB states ran at
// 1.75 TB/s (CatArrayBatchedCopy, 4.9 ms of the 71-ms training step at n = 28).
template <typename T>
__global__ __launch_bounds__(256) void interleave_kernel(const cx<T>* __restrict__ a, const cx<T>* __restrict__ b,
                                                         cx<T>* __restrict__ out, uint64_t count) {
    constexpr int V = 16 / sizeof(cx<T>);       // amplitudes per 16-byte piece: 2 (complex64) or 1 (complex128)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    const f4* pa = reinterpret_cast<const f4*>(a);
    const f4* pb = reinterpret_cast<const f4*>(b);
    f4* po = reinterpret_cast<f4*>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count / V; i += stride) {
        const f4 x = __builtin_nontemporal_load(pa + i), y = __builtin_nontemporal_load(pb + i);
        if constexpr (V == 2) {
            f4 lo, hi;
            lo.x = x.x, lo.y = x.y, lo.z = y.x, lo.w = y.y;
            hi.x = x.z, hi.y = x.w, hi.z = y.z, hi.w = y.w;
            __builtin_nontemporal_store(lo, po + 2 * i);
            __builtin_nontemporal_store(hi, po + 2 * i + 1);
        } else {
            __builtin_nontemporal_store(x, po + 2 * i);
            __builtin_nontemporal_store(y, po + 2 * i + 1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void deinterleave_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, uint64_t count,
                                                           int which) {
    constexpr int V = 16 / sizeof(cx<T>);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    const f4* pi = reinterpret_cast<const f4*>(in);
    f4* po = reinterpret_cast<f4*>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count / V; i += stride) {
        if constexpr (V == 2) {
            const f4 lo = __builtin_nontemporal_load(pi + 2 * i), hi = __builtin_nontemporal_load(pi + 2 * i + 1);
            f4 r;
            if (which) r.x = lo.z, r.y = lo.w, r.z = hi.z, r.w = hi.w;
            else r.x = lo.x, r.y = lo.y, r.z = hi.x, r.w = hi.y;
            __builtin_nontemporal_store(r, po + i);
        } else {
            __builtin_nontemporal_store(__builtin_nontemporal_load(pi + 2 * i + which), po + i);
        }
    }
}

template <typename T>
static int interleave_impl(const void* a, const void* b, void* out, int64_t count, dq_stream_t stream) {
    if (!a || !b || !out || count < 2 || (count & 1) || a == out || b == out) {
        set_error("dq_interleave: bad argument (count = %lld amplitudes per input: even, >= 2; out of place)", (long long)count);
        return DQ_ERR_ARG;
    }
    if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15u) {
        set_error("dq_interleave: the buffers must be 16-byte aligned (the kernel moves 16-byte pieces)");
        return DQ_ERR_ARG;
    }
    uint64_t nb = ((uint64_t)count / (16 / sizeof(cx<T>)) + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(interleave_kernel<T>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), static_cast<const cx<T>*>(a),
                       static_cast<const cx<T>*>(b), static_cast<cx<T>*>(out), (uint64_t)count);
    return check_launch("dq_interleave");
}

template <typename T>
static int deinterleave_impl(const void* in, void* out, int64_t count, int which, dq_stream_t stream) {
    if (!in || !out || count < 2 || (count & 1) || in == out || which < 0 || which > 1) {
        set_error("dq_deinterleave: bad argument (count = %lld amplitudes per output: even, >= 2; which = %d)", (long long)count, which);
        return DQ_ERR_ARG;
    }
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) {
        set_error("dq_deinterleave: the buffers must be 16-byte aligned (the kernel moves 16-byte pieces)");
        return DQ_ERR_ARG;
    }
    uint64_t nb = ((uint64_t)count / (16 / sizeof(cx<T>)) + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(deinterleave_kernel<T>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), static_cast<const cx<T>*>(in),
                       static_cast<cx<T>*>(out), (uint64_t)count, which);
    return check_launch("dq_deinterleave");
}

}  // namespace dq

extern "C" int dq_interleave_c64(const void* a, const void* b, void* out, int64_t count, dq_stream_t stream) {
    return dq::interleave_impl<float>(a, b, out, count, stream);
}
extern "C" int dq_interleave_c128(const void* a, const void* b, void* out, int64_t count, dq_stream_t stream) {
    return dq::interleave_impl<double>(a, b, out, count, stream);
}
extern "C" int dq_deinterleave_c64(const void* in, void* out, int64_t count, int which, dq_stream_t stream) {
    return dq::deinterleave_impl<float>(in, out, count, which, stream);
}
extern "C" int dq_deinterleave_c128(const void* in, void* out, int64_t count, int which, dq_stream_t stream) {
    return dq::deinterleave_impl<double>(in, out, count, which, stream);
}

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
