// Shared device/host helpers for libdqhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dq_hip.h"

namespace dq {

// ---- complex arithmetic on HIP vector types (float2 / double2) --------------------------------
template <typename T> struct C2;
template <> struct C2<float> { using type = float2; };
template <> struct C2<double> { using type = double2; };
template <typename T> using cx = typename C2<T>::type;

template <typename T> __device__ __forceinline__ cx<T> mk(T re, T im) {
    cx<T> r;
    r.x = re;
    r.y = im;
    return r;
}
// a*b
template <typename V> __device__ __forceinline__ V cmul(V a, V b) {
    V r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
// acc + a*b
template <typename V> __device__ __forceinline__ V cfma(V a, V b, V acc) {
    V r;
    r.x = fma(a.x, b.x, fma(-a.y, b.y, acc.x));
    r.y = fma(a.x, b.y, fma(a.y, b.x, acc.y));
    return r;
}
template <typename V> __device__ __forceinline__ V cadd(V a, V b) {
    V r;
    r.x = a.x + b.x;
    r.y = a.y + b.y;
    return r;
}

// ---- index-bit helpers -------------------------------------------------------------------------
// Insert a zero bit at position p (bits >= p move up by one).
__host__ __device__ __forceinline__ uint64_t insert_zero(uint64_t i, int p) {
    const uint64_t lo = i & ((1ull << p) - 1ull);
    return ((i >> p) << (p + 1)) | lo;
}

// Host-side list of sorted bit positions handed to kernels by value.
struct BitList {
    int n;
    int pos[16];
};

__host__ __device__ __forceinline__ uint64_t insert_zeros(uint64_t i, const BitList& bl) {
    for (int s = 0; s < bl.n; ++s) i = insert_zero(i, bl.pos[s]);
    return i;
}

// Deposit the bits of i into the zero positions of `mask` complement, i.e. software pdep over ~mask,
// then OR `value` (value must be a subset of mask).
__host__ __device__ __forceinline__ uint64_t expand_bits(uint64_t i, uint64_t mask, uint64_t value, int nbits) {
    uint64_t r = 0;
    int src = 0;
    for (int p = 0; p < nbits; ++p) {
        if (!((mask >> p) & 1ull)) {
            r |= ((i >> src) & 1ull) << p;
            ++src;
        }
    }
    return r | value;
}

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

inline hipStream_t as_stream(dq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// The wave-tile geometry of complex64 passes (csrc/dq_wave.hip); `pass` has been validated by dq_pass.hip.
// (`known_zero`: index bits, read side, known to be |0> in the input -- dq_apply_fused_zext_*; 0 = none)
int wave_launch_c64(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                    const DqFusedPass* pass, hipStream_t s, uint64_t known_zero = 0, uint64_t slice_mask = 0, uint64_t slice_value = 0);
int wave_launch_grad_c64(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                         const DqFusedPass* pass, hipStream_t s, double* grads, int64_t ngrads, const void* ext_rec = nullptr,
                         int64_t ext_bytes = 0);
int wave_launch_grad_c128(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                          const DqFusedPass* pass, hipStream_t s, double* grads, int64_t ngrads, const void* ext_rec = nullptr,
                          int64_t ext_bytes = 0);
int wave_launch_c128(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                     const DqFusedPass* pass, hipStream_t s, uint64_t known_zero = 0, uint64_t slice_mask = 0, uint64_t slice_value = 0);

// Validate target/control bit lists: in range, pairwise distinct.
int validate_bits(int n, const int* targets, int k, const int* controls, int nc);

}  // namespace dq
